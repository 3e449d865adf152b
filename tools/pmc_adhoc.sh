#!/bin/bash
# Developer tool: one extra PMC pass over one bench frame.  usage (through gpurun): tools/pmc_adhoc.sh <name> COUNTER...  -> gpurun_out/adhoc/<name>.txt
name=$1; shift
R=${RR_CODE_ROOT:-$GRAFT_REPO_ROOT}   # the code (a frozen copy under tools/gpu.sh); output always goes to the real gpurun_out/
out=$GRAFT_REPO_ROOT/gpurun_out/adhoc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $out/$name.json 2> $out/$name.err || { echo "pass $name failed"; tail -n 3 $out/$name.err; exit 1; }
python3 - "$name" <<'PY' > $out/$name.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, {c: x for c, x in sorted(v.items())})
PY
cat $out/$name.txt
