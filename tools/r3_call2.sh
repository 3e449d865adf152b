#!/bin/bash
# round 3, GPU call 2 (run through tools/gpu.sh): multi-handle tests, true-cycle microbenchmark, 3-waves A/B, instruction-cache counters, PC sampling probe
out=$RR_OUT
python -m pytest tests/test_gpu_multi.py tests/test_abi.py -x -q > $out/pytest_multi.txt 2>&1; tail -3 $out/pytest_multi.txt
./build/valu_issue > $out/valu_issue.txt 2>&1 || echo "microbench failed"
tools/ab.sh "" build/lib_base.so build/lib_w3.so build/lib_c3.so build/lib_s3.so build/lib_base.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene helmet_syn --spp 64" build/lib_base.so build/lib_w3.so > $out/ab_helmet.txt 2>&1
cat $out/ab_sponza.txt $out/ab_helmet.txt
tools/pmc_adhoc.sh icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH > $out/icache.log 2>&1 || true
tools/pmc_adhoc.sh ifl SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY > $out/ifl.log 2>&1 || true
cp $GRAFT_REPO_ROOT/gpurun_out/adhoc/icache.txt $GRAFT_REPO_ROOT/gpurun_out/adhoc/ifl.txt $out/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d $out/pcs -- python3 $RR_CODE_ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/pcs.json 2> $out/pcs.err || { echo "pc sampling (stochastic) failed"; tail -5 $out/pcs.err; }
ls -R $out/pcs 2>/dev/null | head -20
