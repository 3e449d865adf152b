#!/bin/bash
# Collects the judged evidence of a round on the GPU box: bench line, rocprofv3 kernel stats, and the PMC passes
# (HBM traffic, SQ issue / wait counters, TCP counters), each PMC set in its own run with --kernel-trace only.
# usage (through gpurun): tools/profile_round.sh <tag> [quick|full] [bench args]   -> gpurun_out/<tag>/ ; then, back in the container:
#        python tools/summarize_profiles.py <tag> <name> [note] [scene]   -> profiles/<name>_*_<scene>.{json,csv}
# "quick": SQ + kernel stats only (before / after comparisons of one kernel change).
# bench args select another workload, e.g.  tools/profile_round.sh r04_helmet full --scene helmet_syn --spp 64
tag=$1; mode=$2; shift; shift
R=${RR_CODE_ROOT:-$GRAFT_REPO_ROOT}   # the code (a frozen copy under tools/gpu.sh); output always goes to the real gpurun_out/
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras $*"
if [ "$mode" != "quick" ]; then
  python3 $R/bench.py "$@" > $out/bench_n1.json 2> $out/bench_n1.err || { echo "bench failed"; tail -5 $out/bench_n1.err; exit 1; }
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 3 --warmup 1 > $out/stats_bench.json 2> $out/stats.err || { echo "stats pass failed"; exit 1; }
pass() { # name, counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$name -- $B --steps 1 --warmup 0 > $out/pmc_$name.json 2> $out/pmc_$name.err || { echo "pmc pass $name failed (see pmc_$name.err)"; tail -3 $out/pmc_$name.err; return 1; }
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS || exit 1
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS || exit 1
if [ "$mode" != "quick" ]; then
  pass fetch FETCH_SIZE || exit 1
  pass write WRITE_SIZE || exit 1
  # the sets below are side information: a set the profiler rejects ("Unable to find all counters") is skipped, never retried
  pass grbm GRBM_GUI_ACTIVE
  pass tcp1 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum
  pass tcp2 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
fi
echo "profile $tag done"; cat $out/stats_bench.json
