#!/bin/bash
# Developer profile: vector-memory path counters per kernel (TA / TCP busy and stalls, UTCL1), separate passes.
# usage (through gpurun): tools/pmc_mem.sh <tag> [bench args]
# (a pass with TA_ADDR_STALLED_* / TD_* / TCP_*_LATENCY counters hung on this pool and is left out)
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/p$i.err || echo "pass $i failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in tot.items():
    if not k.startswith("k_"): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:42s} total {v:14.5g}   per launch {v / cnt[k][c]:12.5g}")
PY
