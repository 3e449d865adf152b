#!/usr/bin/env python3
"""Developer probe: the slowest rank's share of the bench frame at N = 8 (and 4) for different tile sizes of the interleaved
tiling, all ranks measured one after the other on a single GPU: locality per rank against balance between ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rustray_amd import capi
from rustray_amd.renderer import TiledFrame, render_region_torch
fs, cam, cfg = bench.build_workload("sponza_syn", 1280, 720, 128, 1)
ds = capi.DeviceScene(fs, 0)
camc = cam.c_struct()
for n in (8, 4):
    for tw, th in ((32, 8), (64, 16), (64, 32), (128, 32), (128, 72), (160, 90)):
        worst, total = 0.0, 0.0
        for r in range(n):
            tf = TiledFrame(1280, 720, r, n, tw, th)
            tf.world_size_for_gather = 1
            render_region_torch(ds, camc, cfg, tf, aux=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                render_region_torch(ds, camc, cfg, tf, aux=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 3
            worst = max(worst, ms); total += ms
        print(f"N={n} tiles {tw:3d}x{th:2d}: slowest rank {worst:6.2f} ms, mean {total / n:6.2f} ms")
ds.close()
