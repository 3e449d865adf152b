#!/usr/bin/env python3
"""Developer tool: host wall time of rr_scene_update_transforms (median of 9 calls with the items' own matrices) and of rr_pick on a scene.
usage: [RUSTRAY_HIP_LIB=build/lib_x.so] python tools/update_probe.py [scene]"""
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rustray_amd import capi

if os.environ.get("RR_PROBE_ABI"):   # an older library (tools/build_rev.sh): speak its ABI version
    import rustray_amd.flat as _flat
    _flat.RR_ABI_VERSION = int(os.environ["RR_PROBE_ABI"])
scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_syn"
fs, cam, cfg = bench.build_workload(scene, 1280, 720, 1, 1)
t = np.stack([np.asarray(it.trans, np.float32) for it in fs.items])
ti = np.stack([np.asarray(it.trans_inv, np.float32) for it in fs.items])
with capi.DeviceScene(fs, 0) as ds:
    ds.render(cam.c_struct(), cfg, aux=False)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); ds.update_transforms(t, ti); ts.append((time.perf_counter() - t0) * 1e3)
    ps = []
    for k in range(10):
        t0 = time.perf_counter(); ds.pick(cam.c_struct(), 600 + k, 360); ps.append((time.perf_counter() - t0) * 1e3)
    print(f"{scene}: {len(fs.items)} items, {fs.n_triangles_instanced()} instanced triangles: rr_scene_update_transforms {statistics.median(ts[1:]):.3f} ms "
          f"(min {min(ts[1:]):.3f}), rr_pick {statistics.median(ps[1:]):.3f} ms   [{os.environ.get('RUSTRAY_HIP_LIB', 'shipped build')}]")
