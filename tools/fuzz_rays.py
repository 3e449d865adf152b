#!/usr/bin/env python3
"""Developer tool: rr_trace_rays against the oracle's Raytracing::trace on random rays into random scenes, a third of them
with special values in some components (NaN, infinite origin components, +-0, denormals): found / item / face id / toi must agree bit
for bit (a NaN toi as NaN).  usage (on the GPU box): python tools/fuzz_rays.py [scenes] [first seed] [far]"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa
import numpy as np
from oracle import binding as ob
from rustray_amd import capi
from tools.fuzz_parity import rich_scene, far_and_scaled

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
FAR = len(sys.argv) > 3 and sys.argv[3] == "far"   # the scenes moved far from the origin and scaled (fuzz_parity's far mode)
# Special values as a render can produce them: NaN anywhere (a NaN normal), an infinite ORIGIN component (a hit point that overflowed),
# zeros and denormals.  Not covered, and not claimed: infinite DIRECTION components (directions are normalised vectors or their
# reflections: a triangle then reports toi = t * (1 / inf) = 0 in the reference, which the non-finite path here does not reproduce) and
# coordinates whose squares overflow (1e30).
SPECIAL_O = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 1.0], np.float32)
SPECIAL_D = np.array([np.nan, 0.0, -0.0, 1e-42, -1e-42, 1.0], np.float32)
bad = 0
t0 = time.time()
for seed in range(first, first + n_scenes):
    fs = rich_scene(9000 + seed)
    scale = 1.0
    if FAR:
        fs, scale = far_and_scaled(fs, seed)
    rng = np.random.default_rng(seed)
    n = 3000
    eye = np.asarray(fs.meta["camera"]["eye_pos"], np.float32)
    o = (eye[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(rng.choice([0.01, 1.0, 5.0]) * scale)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)          # half normalised, half not (trace does not normalise)
    k = n // 3                                                                 # special values in one to three components of the first third
    for i in range(k):
        for _ in range(int(rng.integers(1, 4))):
            if rng.random() < 0.5: o[i, int(rng.integers(0, 3))] = SPECIAL_O[int(rng.integers(0, len(SPECIAL_O)))]
            else: d[i, int(rng.integers(0, 3))] = SPECIAL_D[int(rng.integers(0, len(SPECIAL_D)))]
    depth = int(rng.integers(1, 3))
    with capi.DeviceScene(fs, 0) as ds:
        g = ds.trace_rays(o, d, depth)
    r = ob.trace_rays(fs.c_struct(), o, d, depth, brute_force=True)   # (the all-items form: DESIGN D10)
    both = g[0] & r[0]
    same_toi = (g[3].view(np.uint32) == r[3].view(np.uint32)) | (np.isnan(g[3]) & np.isnan(r[3]))
    diff = (g[0] != r[0]) | (both & ((g[1] != r[1]) | (g[2] != r[2]) | ~same_toi))
    if diff.any():
        bad += 1
        i = int(np.nonzero(diff)[0][0])
        print("MISMATCH seed", seed, "rays", int(diff.sum()), "first:", i, o[i], d[i], "depth", depth, "hip", g[0][i], g[1][i], g[2][i], g[3][i], "oracle", r[0][i], r[1][i], r[2][i], r[3][i], flush=True)
    if (seed - first + 1) % 50 == 0:
        print(f"... {seed - first + 1} scenes, {bad} with differing rays, {time.time() - t0:.0f} s", flush=True)
print(f"{n_scenes} scenes x 3000 rays, {bad} scenes with differing rays, {time.time() - t0:.0f} s")
