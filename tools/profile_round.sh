#!/bin/bash
# Collects the judged evidence of a round on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes.
# usage: tools/profile_round.sh r01   (run through gpurun; copies summaries under gpurun_out/<tag>/)
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $out/bench_n1.json 2> $out/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
cat $out/bench_n1.json
