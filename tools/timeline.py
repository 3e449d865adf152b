#!/usr/bin/env python3
"""Developer probe: the dispatch TIMELINE of one frame (every kernel launch with its start offset, duration and the idle gap before
it), from `rocprofv3 --kernel-trace`.  Shows what a frame pays besides kernel time: level read-backs, launch gaps, small launches.
usage (GPU box): python tools/timeline.py run <n_ranks_share> [scene spp]   -> renders rank 0's share of an n-way tiling 4 times
                 rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/timeline.py run 8 ; python tools/timeline.py show DIR"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch
    import bench
    from rustray_amd import capi
    from rustray_amd.renderer import TiledFrame, render_region_torch
    n = int(sys.argv[2])
    scene = sys.argv[3] if len(sys.argv) > 3 else "sponza_syn"
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    fs, cam, cfg = bench.build_workload(scene, 1280, 720, spp, 1)
    ds = capi.DeviceScene(fs, 0)
    camc = cam.c_struct()
    tf = TiledFrame(1280, 720, 0, n, 32, 8)
    for _ in range(4):
        render_region_torch(ds, camc, cfg, tf, aux=True)
        torch.cuda.synchronize()
    ds.close()
else:
    f = max(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last frame = from the last k_trace_closest<true> launch to the k_resolve after it
    names = [r["Kernel_Name"] for r in rows]
    first = max(i for i, nm in enumerate(names) if "k_trace_closest<true>" in nm)
    last = next(i for i in range(first, len(rows)) if "k_resolve" in names[i])
    # include the memsets before the level-1 launch (same frame): walk back while the gap is small
    t0 = int(rows[first]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    print(f"{'kernel':44s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}")
    for r in rows[first:last + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
        print(f"{nm[:44]:44s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
        busy += e - s
        prev_end = e
    total = prev_end - t0
    print(f"frame (level-1 launch to resolve): {total / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, gaps {(total - busy) / 1e3:.1f} us, {last - first + 1} dispatches")
