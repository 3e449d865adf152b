#!/usr/bin/env python3
"""Developer tool: one seed of tools/fuzz_parity.py again, against BOTH forms of the oracle (its item tree and all items), with the differing
pixels.  usage (GPU box): [RR_PROBE_ABI=2 RUSTRAY_HIP_LIB=build/lib_x.so] python tools/replay_seed.py <seed> <basic|rich|far|farbasic>"""
import os, sys
sys.path.insert(0, '.')
import torch, numpy as np
if os.environ.get("RR_PROBE_ABI"):
    import rustray_amd.flat as _flat
    _flat.RR_ABI_VERSION = int(os.environ["RR_PROBE_ABI"])
from oracle import binding as ob
from rustray_amd import capi
from rustray_amd.flat import make_config
from tests.helpers import camera_for, compare_frames
from tools.fuzz_parity import case
seed, mode = int(sys.argv[1]), sys.argv[2]
fs, w, h, kw, brute, _ = case(seed, mode)
cam = camera_for(fs, w, h).c_struct(); cfg = make_config(**kw)
with capi.DeviceScene(fs, 0) as ds:
    if seed % 4 == 0: ds.set_tuning(bin_min_rays=1)
    if seed % 7 == 0 and mode != "rich": ds.set_tuning(queue_budget_bytes=1, shade_chunk_rays=65536)
    out = ds.render(cam, cfg); st = ds.stats()
for bf in (False, True):
    ref = ob.render(fs.c_struct(), cam, cfg, n_threads=14, want_counters=True, brute_force=bf)
    r = compare_frames(out, ref)
    d = np.abs(out["rgba"][..., :3].astype(int) - ref["rgba"][..., :3].astype(int)).max(-1)
    ys, xs = np.nonzero(d > 1)
    print("brute" if bf else "tree ", {k: r[k] for k in ("max_rgb_diff", "n_rgb_over", "n_id_diff", "nan_mismatch")}, "pixels", list(zip(xs.tolist(), ys.tolist()))[:5],
          [(out["rgba"][y, x, :3].tolist(), ref["rgba"][y, x, :3].tolist()) for x, y in list(zip(xs.tolist(), ys.tolist()))[:3]],
          "counts dev", st["secondary_rays"], st["shaded_hits"], "oracle", ref["counters"]["rays_secondary"], ref["counters"]["shaded_hits"])
print("items", len(fs.items), "w h", w, h, kw, os.environ.get("RUSTRAY_HIP_LIB", "shipped"))
