#!/usr/bin/env python3
"""Developer probe: host-side latency of rr_render for the small BASELINE configs (C1 spheres 256x256x1 mc=0, C2 monkey 800x600x16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first, see tests/conftest.py)
from rustray_amd import capi
from rustray_amd.flat import make_config
from tests.helpers import camera_for, load_scene
for name, w, h, spp, mc in (("spheres", 256, 256, 1, False), ("monkey", 800, 600, 16, True), ("spheres", 1280, 720, 1, False)):
    fs = load_scene(name); cam = camera_for(fs, w, h).c_struct(); cfg = make_config(samples=spp, monte_carlo=mc, seed=0)
    with capi.DeviceScene(fs, 0) as ds:
        ds.set_profiling(True)
        for _ in range(3): ds.render(cam, cfg)
        t0 = time.perf_counter()
        for _ in range(20): ds.render(cam, cfg)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        st = ds.stats()
        print(f"{name} {w}x{h}x{spp}: rr_render {dt:.3f} ms host, device frame {st['ms_total']:.3f} ms, kernels closest {st['ms_trace_closest']:.3f} shadow {st['ms_trace_shadow']:.3f} shade {st['ms_shade']:.3f}, launches {st['launches_trace_closest']}/{st['launches_shade']}/{st['launches_trace_shadow']}, rays {st['primary_rays']+st['secondary_rays']+st['shadow_rays']}")
