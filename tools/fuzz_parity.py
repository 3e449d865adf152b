#!/usr/bin/env python3
"""Developer tool: HIP path against the oracle on N seeded random scenes (the generator of tests/test_gpu_random.py) with
random frame sizes, sample counts (up to 64), recursion depths (up to 9), fog / gamma / depth of field, and with the
binning and small-arena paths switched on for some seeds.  Counts frames that differ by more than 1 LSB or whose path
trees differ.  usage (on the GPU box): python tools/fuzz_parity.py [N]"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa
import numpy as np
from oracle import binding as ob
from rustray_amd import capi
from rustray_amd.flat import make_config
from tests.helpers import camera_for, compare_frames
from tests.test_gpu_random import _random_scene
bad = 0
t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(100, 100 + N):
    fs = _random_scene(5000 + seed)
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(40, 110)), int(rng.integers(30, 90))
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=int(rng.choice([1, 2, 3, 4, 6, 16, 64])), monte_carlo=bool(seed % 3), seed=seed, max_recursion=int(rng.choice([1, 2, 4, 6, 9])),
                      fog_density=float(rng.choice([0.0, 0.02])), gamma_correction=bool(seed % 5 == 0),
                      aperture_size=float(rng.choice([1.0, 1.0, 8.0])), focal_length=float(rng.choice([1.0, 6.0])))
    with capi.DeviceScene(fs, 0) as ds:
        if seed % 4 == 0:
            ds.set_tuning(bin_min_rays=1)
        if seed % 7 == 0:
            ds.set_tuning(queue_budget_bytes=1, shade_chunk_rays=65536)
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = ob.render(fs.c_struct(), cam, cfg, n_threads=14, want_counters=True)
    r = compare_frames(out, ref)
    c = ref["counters"]
    ok = (r["n_rgb_over"] == 0 and r["n_id_diff"] == 0 and r["nan_mismatch"] == 0 and r["max_depth_rel"] < 1e-4 and st["primary_rays"] == c["rays_primary"]
          and st["secondary_rays"] == c["rays_secondary"] and st["shaded_hits"] == c["shaded_hits"] and st["shadow_rays"] <= c["rays_shadow"])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, r, {k: st[k] for k in ("primary_rays", "secondary_rays", "shaded_hits", "shadow_rays")}, {k: c[k] for k in ("rays_primary", "rays_secondary", "shaded_hits", "rays_shadow")})
print(f"{N} random scenes, {bad} mismatches, {time.time() - t0:.0f} s")
