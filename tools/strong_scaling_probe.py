#!/usr/bin/env python3
"""Developer probe: time ONE rank's share of the bench frame for N = 1, 2, 4, 8 on a single GPU
(tiles tile_index % N == 0), to see how far per-frame fixed costs eat into strong scaling."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rustray_amd import capi
from rustray_amd.renderer import TiledFrame, render_region_torch

args = argparse.Namespace(scene="sponza_syn", width=1280, height=720, spp=128, monte_carlo=1)
fs, cam, cfg = bench.build_workload(args.scene, args.width, args.height, args.spp, args.monte_carlo)
ds = capi.DeviceScene(fs, 0)
ds.set_profiling(True)
camc = cam.c_struct()
base = None
for n in (1, 2, 4, 8):
    tf = TiledFrame(args.width, args.height, 0, n, 32, 8)
    tf.world_size_for_gather = 1
    for _ in range(2):
        render_region_torch(ds, camc, cfg, tf, aux=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        render_region_torch(ds, camc, cfg, tf, aux=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 200.0
    st = ds.stats()
    base = base or ms
    print(f"N={n}: rank-0 share {ms:7.2f} ms   ideal {base / n:7.2f} ms   efficiency bound {base / n / ms:5.2f}   "
          f"last frame: device {st['ms_total']:.2f} closest {st['ms_trace_closest']:.2f} shadow {st['ms_trace_shadow']:.2f} shade {st['ms_shade']:.2f} "
          f"launches {st['launches_trace_closest']}/{st['launches_trace_shadow']}/{st['launches_shade']}")
ds.close()
