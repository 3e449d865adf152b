#!/usr/bin/env python3
"""Developer tool (GPU): per-pixel statistical test of the README renderings against K seeds of the HIP path (the exploration
behind tests/test_ref_shots.py::test_reference_rendering_is_one_more_draw_of_the_hip_path).  Prints, per shot, semantic and
sampling model, mean z, the tail fractions, the 16x16-block bias map's extremes, and the same for a CONTROL (render K+1 pushed
through the test); saves mean / sd / control arrays to $RR_OUT (or gpurun_out/refz) for offline analysis.
  model "fixed":  every render uses the built-in sub-sample table (the seed only moves the jitter)
  model "random": every render draws its own table (a numpy shuffle of the cell_size^2 cells), so that sd holds the variance of
                  the sampling pattern too -- the model under which the reference is "one more draw" if its binary did not use
                  the seed-0 table of the source at HEAD."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustray_amd import capi  # noqa: E402
from rustray_amd.flat import make_config  # noqa: E402
from tests.helpers import camera_for  # noqa: E402
from tests.test_ref_shots import SHOTS, box2, load_shot, scene_of_2022  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
OUT = os.environ.get("RR_OUT") or os.path.join(ROOT, "gpurun_out", "refz")
os.makedirs(OUT, exist_ok=True)


def zstats(r, mu, sd, keep, label):
    z = (r - mu) / np.sqrt(sd * sd * (1.0 + 1.0 / K) + 1.0 / 12.0)
    zz = z[keep]
    d = (r - mu)
    H, W = keep.shape[:2]
    bh, bw = H // 16, W // 16
    db = np.where(keep, d, np.nan)[: bh * 16, : bw * 16].reshape(bh, 16, bw, 16, -1)
    full = np.isfinite(db).all(axis=(1, 3, 4))
    with np.errstate(invalid="ignore"):
        bmf = np.where(full, np.nanmean(db, axis=(1, 3, 4)), np.nan)
    worst = np.dstack(np.unravel_index(np.argsort(-np.abs(np.nan_to_num(bmf)), axis=None)[:4], bmf.shape))[0]
    print(f"  {label:30s} n={zz.size:7d} mean z {zz.mean():+.4f} sd z {zz.std():.3f} |z|>3 {np.mean(np.abs(zz) > 3):.5f} |z|>4 {np.mean(np.abs(zz) > 4):.6f} "
          f"|z|>6 {np.mean(np.abs(zz) > 6):.6f} bias {d[keep].mean():+.4f} block max {np.nanmax(np.abs(bmf)):.3f} "
          f"worst {[(int(a), int(b), round(float(bmf[a, b]), 2)) for a, b in worst]}", flush=True)


def cell_size_of(samples):
    if samples <= 1:
        return 1
    v, p = samples + 2, 1
    while p < v:
        p <<= 1
    return p // 2


for name in SHOTS:
    ref, mask, meta = load_shot(name)
    fs = scene_of_2022(name)
    cam = camera_for(fs, 1280, 720).c_struct()
    r = ref.astype(np.float64)
    spp = meta["samples"]
    cs = cell_size_of(spp)
    cells = np.stack(np.meshgrid(np.arange(cs), np.arange(cs), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.uint16)
    for compat in ((1, 0) if mask.any() else (1,)):
        for model in ("fixed", "random"):
            rng = np.random.default_rng(777)
            with capi.DeviceScene(fs, 0) as ds:
                ds.set_compat(compat)
                frames = []
                for seed in range(K + 1):
                    cfg = make_config(samples=spp, monte_carlo=True, seed=1000 + seed)
                    table = None if model == "fixed" else np.ascontiguousarray(cells[rng.permutation(len(cells))[:spp]])
                    frames.append(box2(ds.render(cam, cfg, aux=False, sample_xy=table)["rgba"][..., :3]).astype(np.float32))
            frames = np.stack(frames)
            mu, sd = frames[:K].mean(axis=0).astype(np.float64), frames[:K].std(axis=0, ddof=1).astype(np.float64)
            ctrl = frames[K].astype(np.float64)
            np.savez_compressed(os.path.join(OUT, f"{name}_compat{compat}_{model}.npz"), mu=mu.astype(np.float32), sd=sd.astype(np.float32), ctrl=frames[K].astype(np.uint8))
            keep_all = np.ones(ref.shape, bool)
            print(f"{name} ({spp} spp, K={K}) semantic={'2022 (occluder alpha)' if compat else 'HEAD (receiver alpha)'} tables={model}")
            zstats(r, mu, sd, keep_all, "reference, whole frame")
            zstats(ctrl, mu, sd, keep_all, "control, whole frame")
            if mask.any():
                keep_unmasked = np.repeat(~mask[..., None], 3, axis=2)
                zstats(r, mu, sd, keep_unmasked, "reference, outside era_mask")
                zstats(ctrl, mu, sd, keep_unmasked, "control, outside era_mask")
