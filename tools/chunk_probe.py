#!/usr/bin/env python3
"""Developer tool: the contract frame with different shade chunk sizes (rr_tuning::shade_chunk_rays): does keeping a chunk's
shadow rays in the Infinity Cache between k_shade and k_trace_shadow pay for the smaller launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rustray_amd import capi
fs, cam, cfg = bench.build_workload("sponza_syn", 1280, 720, 128, 1)
with capi.DeviceScene(fs, 0) as ds:
    ds.set_tuning(kernel_timing=1)
    for chunk in (0, 32 << 20, 16 << 20, 8 << 20, 4 << 20, 2 << 20, 0):
        ds.set_tuning(shade_chunk_rays=chunk)
        ds.render(cam.c_struct(), cfg)
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); ds.render(cam.c_struct(), cfg); t.append((time.perf_counter() - t0) * 1e3)
        st = ds.stats()
        print(f"chunk {chunk >> 20:3d} M rays: frame {min(t):6.2f} ms  closest {st['ms_trace_closest']:.2f} shade {st['ms_shade']:.2f} shadow {st['ms_trace_shadow']:.2f}")
