"""Turns gpurun_out/<tag>/ (made by tools/profile_round.sh on the GPU box) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]  # e.g. r01c r01
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "bench_n1.json"), f"profiles/{name}_bench_sponza_syn_n1.json")
stats = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)  # gpurun merges into older local copies
shutil.copy(stats, f"profiles/{name}_kernel_stats_sponza_syn.csv")
data = {}
for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = max(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    data[cname] = agg
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (two separate passes)",
       "workload": "sponza_syn 1280x720 128spp monte_carlo=1, one frame", "unit": "bytes",
       "correction": "FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE is used as read. "
                     "Calibration in this access pattern: k_raygen WRITE_SIZE per launch equals rays x 40 B exactly. The ray-queue reads of the trace kernels are 16-B/lane streams "
                     "(the x2 case); BVH node / triangle gathers mostly hit L2 / Infinity Cache and are uncalibrated.",
       "kernels": {}}
for k in ("k_trace_closest", "k_trace_shadow", "k_shade", "k_raygen", "k_resolve"):
    n = data["FETCH_SIZE"][k][0]
    fr, wr = data["FETCH_SIZE"][k][1] * 1024, data["WRITE_SIZE"][k][1] * 1024
    out["kernels"][k] = {"launches_per_frame": n, "FETCH_SIZE_raw_bytes_per_frame": fr, "WRITE_SIZE_bytes_per_frame": wr,
                         "hbm_bytes_per_launch_corrected": (2 * fr + wr) / max(n, 1)}
out["k_trace_closest_bytes_per_launch"] = out["kernels"]["k_trace_closest"]["hbm_bytes_per_launch_corrected"]
json.dump(out, open(f"profiles/{name}_hbm_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in out["kernels"].items()}))
print(open(f"profiles/{name}_kernel_stats_sponza_syn.csv").read()[:900])
