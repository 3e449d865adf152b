#!/bin/bash
# Developer profile: per-kernel time of one scene, level 1 (k_*<true>) against the deeper levels (k_*<false>), from rocprofv3 --kernel-trace --stats.
# usage (through gpurun): tools/level_split.sh <tag> <bench args>
tag=$1; shift
R=${RR_CODE_ROOT:-$GRAFT_REPO_ROOT}   # the code (a frozen copy under tools/gpu.sh); output always goes to the real gpurun_out/
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > $out/bench.json 2> $out/err.txt || { echo failed; tail -n 3 $out/err.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, os, json
f = max(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
b = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print(b["config"]["workload"][:60], "| rays per frame", b["rays_per_frame"], "| ms", round(b["ms_per_step"], 2))
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("void k_", "k_")):
        print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):4d}  per frame {float(r["TotalDurationNs"]) / 4e6:8.3f} ms  avg {float(r["AverageNs"]) / 1e6:8.3f} ms')
PY
