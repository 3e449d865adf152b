"""The oracle (and the HIP path) against the reference's OWN renderings.

The reference ships renderings with the exact command line that made them (Readme.md:33-46); three of those scenes
use only assets that are local (tests/golden/ref_shots/make_ref_shots.py).  They were made in 2022-05 with the
un-seeded `thread_rng` jitter (src/raytracing.rs:616-618), so agreement is statistical: PSNR / mean |d| / mean signed
bias on the 2x box-downsampled frame.

The shots also dated the binary: it fetched texels NEAREST (expressible: `texture_filtering_nearest` on every
material) and attenuated shadows by the OCCLUDER's alpha where HEAD takes the receiver's (src/raytracing.rs:898;
not expressible as input — the oracle has a test-only switch, and `era_mask` marks the pixels that semantic touches).
What these tests pin against the real binary is therefore everything else on the path: the camera and ray
generation, item / triangle / sphere intersection, shadow-ray order semantics, point / directional falloff, Phong
terms, reflection, refraction, fresnel, alpha compositing, nearest texel addressing and uv conventions, clamp and
`as u8`.  NOT pinned by any reference output: the bilinear texel path, the receiver-alpha shadow semantic (both
follow the source text at HEAD), fog, DOF, gamma, normal / roughness / AO / reflectivity maps.

CPU tests: the oracle at 1280x720 / 4 spp (about a minute on 8 cores).  GPU tests: the HIP path at the README's own size and sample
count, plus HIP == oracle on a tile at that sample count.
Thresholds are set from this repository's own runs (noted beside each), a few tenths of a dB under the measured value.
"""
import json
import os

import numpy as np
import pytest

from rustray_amd.flat import make_config
from tests.helpers import GOLDEN, camera_for, compare_frames, load_scene

SHOTS = ("room_spheres", "room_kbert", "floor_monkey")


def load_shot(name):
    z = np.load(os.path.join(GOLDEN, "ref_shots", name + ".npz"))
    ref = z["rgb_half"]
    mask = np.unpackbits(z["era_mask"])[: ref.shape[0] * ref.shape[1]].reshape(ref.shape[:2]).astype(bool)
    return ref, mask, json.loads(str(z["meta"]))


def box2(rgb):
    h, w = rgb.shape[:2]
    s = rgb.astype(np.uint16).reshape(h // 2, 2, w // 2, 2, -1).sum(axis=(1, 3))
    return ((s + 2) // 4).astype(np.uint8)


def score(ours, ref, mask=None):
    """PSNR (dB), mean |d|, mean signed d over the pixels NOT in `mask`."""
    d = ours.astype(np.float64) - ref.astype(np.float64)
    if mask is not None:
        d = d[~mask]
    mse = float(np.mean(d * d))
    return 10.0 * np.log10(255.0 ** 2 / max(mse, 1e-12)), float(np.abs(d).mean()), float(d.mean())


def scene_of_2022(name):
    fs = load_scene(name)
    for m in fs.materials:
        m.texture_filtering_nearest = True  # finding 1 (module docstring)
    return fs


# ---------------------------------------------------------------------------
# CPU: the oracle against the shots
# ---------------------------------------------------------------------------
# measured here at 1280x720, 4 spp, seed 0, box-downsampled (PSNR dB / mean |d| / bias), 2022 semantics, whole frame:
#   room_spheres 40.12 / 1.016 / -0.055   room_kbert 38.39 / 1.265 / +0.008   floor_monkey 45.19 / 0.164 / +0.002
# (at the shots' own 128 / 64 / 32 spp: room_kbert 43.75 / 0.717 / -0.01, floor_monkey 48.62 / 0.106 / +0.014)
CPU_2022 = {"room_spheres": (39.5, 1.15, 0.2), "room_kbert": (37.8, 1.4, 0.15), "floor_monkey": (44.5, 0.2, 0.1)}
# HEAD semantics (receiver alpha) outside era_mask: room_spheres 41.04 (bias -0.25: the mask keeps differences <= 1 LSB),
# floor_monkey 49.72; room_kbert has no pixel in the mask (every material there has alpha 1)
CPU_HEAD_UNMASKED = {"room_spheres": (40.4, 0.4), "floor_monkey": (49.0, 0.1)}


_CACHE = {}


def oracle_half(oracle, name, spp, era, nearest=True):
    """Oracle frame of a shot scene at 1280x720, box-downsampled like the stored shot."""
    key = (name, spp, era, nearest)
    if key not in _CACHE:
        fs = scene_of_2022(name) if nearest else load_scene(name)
        cam = camera_for(fs, 1280, 720).c_struct()
        cfg = make_config(samples=spp, monte_carlo=True, seed=0)
        oracle.lib().rro_set_shot_era(era)
        try:
            _CACHE[key] = box2(oracle.render(fs.c_struct(), cam, cfg, n_threads=8)["rgba"][..., :3])
        finally:
            oracle.lib().rro_set_shot_era(0)
    return _CACHE[key]


@pytest.mark.parametrize("name", SHOTS)
def test_oracle_matches_the_reference_rendering(oracle, name):
    ref, mask, meta = load_shot(name)
    assert meta["width"] == 1280 and meta["height"] == 720
    psnr_min, mad_max, bias_max = CPU_2022[name]
    ours = oracle_half(oracle, name, 4, era=1 if mask.any() else 0)
    psnr, mad, bias = score(ours, ref)
    assert psnr >= psnr_min and mad <= mad_max and abs(bias) <= bias_max, (name, psnr, mad, bias)
    if not mask.any():
        return  # the two shadow-alpha semantics coincide on this scene
    # the source at HEAD (receiver's alpha): same agreement wherever that one semantic has no footprint ...
    head = oracle_half(oracle, name, 4, era=0)
    psnr_h, _, bias_h = score(head, ref, mask)
    assert psnr_h >= CPU_HEAD_UNMASKED[name][0] and abs(bias_h) <= CPU_HEAD_UNMASKED[name][1], (name, psnr_h, bias_h)
    # ... and the pin is sharp enough to tell the two semantics apart where it has one
    in_mask_2022 = score(ours, ref, ~mask)[0]   # 39.8 / 36.6 dB (room_spheres / floor_monkey)
    in_mask_head = score(head, ref, ~mask)[0]   # 34.1 / 27.5 dB
    assert in_mask_2022 >= in_mask_head + 4.0, (name, in_mask_2022, in_mask_head)


def test_bilinear_default_is_what_the_2022_binary_did_not_do(oracle):
    """room_kbert with HEAD's bilinear default is far further from the shot than with nearest texels
    (4 spp: 29.6 dB against 38.4 dB; 8 spp: 29.3 against 39.4)."""
    ref, _, _ = load_shot("room_kbert")
    near = score(oracle_half(oracle, "room_kbert", 4, era=0), ref)[0]
    bil = score(oracle_half(oracle, "room_kbert", 4, era=0, nearest=False), ref)[0]
    assert near >= bil + 6.0, (near, bil)


# ---------------------------------------------------------------------------
# GPU: the HIP path against the shots, at the README's own size and sample count
# ---------------------------------------------------------------------------
# measured on MI355X at 1280x720 and the shot's spp (HEAD semantics, nearest texels), outside era_mask (PSNR / mean |d| / bias):
#   room_spheres (29 % of the frame) 51.31 / 0.271 / -0.154 (the mask keeps differences <= 1 LSB: hence the bias)
#   room_kbert (100 %) 43.75 / 0.717 / -0.008   floor_monkey (91 %) 52.64 / 0.042 / +0.008
GPU_HEAD_UNMASKED = {"room_spheres": (50.5, 0.25), "room_kbert": (43.2, 0.1), "floor_monkey": (51.8, 0.1)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHOTS)
def test_hip_matches_the_reference_rendering(hip, oracle, name):
    ref, mask, meta = load_shot(name)
    fs = scene_of_2022(name)
    cam = camera_for(fs, 1280, 720).c_struct()
    cfg = make_config(samples=meta["samples"], monte_carlo=True, seed=0)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    psnr, mad, bias = score(box2(out["rgba"][..., :3]), ref, mask)
    psnr_min, bias_max = GPU_HEAD_UNMASKED[name]
    print(f"{name}: HIP vs README rendering outside era_mask ({100 * (1 - mask.mean()):.0f} % of the frame): "
          f"PSNR {psnr:.2f} dB, mean |d| {mad:.3f}, bias {bias:+.3f}")
    assert psnr >= psnr_min and abs(bias) <= bias_max, (name, psnr, mad, bias)
    # and the HIP frame IS the oracle's frame at this sample count (tile: the oracle takes seconds per 64x32x128)
    win = (608, 344, 672, 376)
    o = oracle.render(fs.c_struct(), cam, cfg, window=win, n_threads=8)
    x0, y0, x1, y1 = win
    res = compare_frames({k: v[y0:y1, x0:x1] for k, v in out.items()}, {k: v[y0:y1, x0:x1] for k, v in o.items()})
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


# ---------------------------------------------------------------------------
# GPU: the reference rendering as ONE MORE DRAW of the HIP path -- a per-pixel statistical test (round 3)
# ---------------------------------------------------------------------------
# Whole-frame PSNR is dominated by the reference's own Monte-Carlo noise and averages a local systematic error away.  Here
# every pixel is tested by itself: K renders of the HIP path at the shot's own size and sample count (frames cost
# milliseconds) give a per-pixel mean mu and per-render sd of the box-downsampled value; the stored reference pixel r is
# tested as one more draw, z = (r - mu) / sqrt(sd^2 (1 + 1/K) + 1/12) (1/12: the u8 quantisation both sides went through),
# and render K + 1 is pushed through the same test as the CONTROL -- the reference must not look worse than our own next draw.
#
# What the test found (finding 3): with the built-in sub-sample table in every render (the seed only
# moves the jitter) the reference is NOT a draw -- 0.2 % / 9.5 % / 1.4 % of the pixel channels sit beyond 6 sigma, all on
# geometric and texel edges, where sd is zero and the value is decided by which cells of the pixel the samples fall in.  With
# a table drawn at random per render (sd then holds the variance of the sampling pattern) the reference is statistically
# indistinguishable from the control on the WHOLE frame of all three shots.  So the binary of 2022-05 did not sample the
# seed-0 table the source at HEAD builds (src/raytracing.rs:290-313) or this library's restatement of it; what the shots pin is
# the distribution of the offsets (cells of the cell_size grid, [0, 1) px right of and above the pixel centre), not the table.
# (tools/ref_shot_tables.py: eight other plausible fixed tables are rejected the same way, and the reference's edge residuals
# are uncorrelated between neighbouring pixels like a table drawn per pixel, not like one table per frame.)
# The table is an INPUT of the boundary (rr_render's sample_xy): a Rust host passes its own.
#
# The 2022 shadow semantic (finding 2) is a compatibility switch of the PRODUCT (rr_scene_set_compat), so the shipped
# kernels are compared with the reference over 100 % of each frame; HEAD's semantic is then checked on the pixels where the two
# semantics produce bit-identical frames in every one of the K renders.
Z_K = 32
# measured on MI355X (deterministic: frames are bit-identical on every box), reference | control, tables drawn per render, 2022 semantic, whole frame:
#   room_spheres  mean z -0.0010 | +0.0012   sd z 0.589 | 0.565   |z|>4 3.5e-5 | 2.3e-5   |z|>6 1e-6 | 1e-6    bias -0.0007 LSB   16x16-block |bias| max 0.20 | 0.23
#   room_kbert    mean z -0.0021 | +0.0012   sd z 0.685 | 0.625   |z|>4 1.7e-4 | 1.2e-5   |z|>6 1.7e-5 | 0    bias -0.0013 LSB   block max 0.25 | 0.88
#   floor_monkey  mean z +0.0006 | +0.0016   sd z 0.221 | 0.162   |z|>4 3.9e-5 | 1e-6     |z|>6 1.2e-5 | 1e-6 bias +0.0010 LSB   block max 0.42 | 1.09
# (the few pixels beyond 4.5 sigma are isolated: channels that saturate at 255 in all K renders, and edge pixels whose
#  discrete coverage distribution 32 draws under-sample; none are neighbours.)  Same test with the built-in table in every render:
#   |z|>6  2.3e-3 | 0     9.5e-2 | 0     1.4e-2 | 0


def _cell_size_of(samples):
    if samples <= 1:
        return 1
    v, p = samples + 2, 1
    while p < v:
        p <<= 1
    return p // 2


def _z_stats(r, mu, sd, keep, k):
    z = (r - mu) / np.sqrt(sd * sd * (1.0 + 1.0 / k) + 1.0 / 12.0)
    zz, d = z[keep], (r - mu)
    H, W = keep.shape[:2]
    bh, bw = H // 16, W // 16
    db = np.where(keep, d, np.nan)[: bh * 16, : bw * 16].reshape(bh, 16, bw, 16, -1)
    full = np.isfinite(db).all(axis=(1, 3, 4))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # blocks with masked pixels are all-NaN slices
        bm = np.where(full, np.nanmean(db, axis=(1, 3, 4)), np.nan)
    worst = np.dstack(np.unravel_index(np.argsort(-np.abs(np.nan_to_num(bm)), axis=None)[:4], bm.shape))[0]
    return dict(n=int(zz.size), mean_z=float(zz.mean()), sd_z=float(zz.std()), gt4=float(np.mean(np.abs(zz) > 4)), gt6=float(np.mean(np.abs(zz) > 6)),
                bias=float(d[keep].mean()), block_max=float(np.nanmax(np.abs(bm))) if full.any() else 0.0,
                worst_blocks=[(int(a), int(b), round(float(bm[a, b]), 2)) for a, b in worst if full[a, b]])


def _draws(hip, fs, cam, spp, compat, k, tables):
    """k + 1 renders (seeds 1000 ...) of one scene: full-resolution RGB frames.  tables: list of sample tables (None = built-in)."""
    frames = []
    with hip.DeviceScene(fs, 0) as ds:
        ds.set_compat(compat)
        for i in range(k + 1):
            cfg = make_config(samples=spp, monte_carlo=True, seed=1000 + i)
            frames.append(ds.render(cam, cfg, aux=False, sample_xy=tables[i])["rgba"][..., :3].copy())
    return frames


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHOTS)
def test_reference_rendering_is_one_more_draw_of_the_hip_path(hip, name):
    ref, era_mask, meta = load_shot(name)
    fs = scene_of_2022(name)
    cam = camera_for(fs, 1280, 720).c_struct()
    spp, k = meta["samples"], Z_K
    cs = _cell_size_of(spp)
    cells = np.stack(np.meshgrid(np.arange(cs), np.arange(cs), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.uint16)
    rng = np.random.default_rng(777)
    tables = [np.ascontiguousarray(cells[rng.permutation(len(cells))[:spp]]) for _ in range(k + 1)]
    r = ref.astype(np.float64)
    everywhere = np.ones(ref.shape, bool)

    def mu_sd_ctrl(frames):
        half = np.stack([box2(f).astype(np.float32) for f in frames])
        return half[:k].mean(axis=0).astype(np.float64), half[:k].std(axis=0, ddof=1).astype(np.float64), half[k].astype(np.float64)

    # ---- the product with the 2022 shadow semantic, a table drawn per render: the WHOLE frame
    f2022 = _draws(hip, fs, cam, spp, 1, k, tables)
    mu, sd, ctrl = mu_sd_ctrl(f2022)
    a, c = _z_stats(r, mu, sd, everywhere, k), _z_stats(ctrl, mu, sd, everywhere, k)
    print(f"{name} ({spp} spp, K={k}, whole frame, 2022 semantic, tables drawn per render)\n  reference {a}\n  control   {c}")
    assert abs(a["mean_z"]) <= 0.02 and abs(a["bias"]) <= 0.02, a                # measured |mean z| <= 0.0021, |bias| <= 0.0013 LSB
    assert a["sd_z"] <= 1.5 * c["sd_z"], (a, c)                                   # measured ratio 1.04 / 1.10 / 1.36
    assert a["gt4"] <= max(20.0 * c["gt4"], 5e-4) and a["gt6"] <= 1e-4, (a, c)    # measured 3.5e-5 / 1.7e-4 / 3.9e-5 and <= 1.7e-5
    assert a["block_max"] <= 0.6, a                                               # measured 0.20 / 0.25 / 0.42 LSB (control: 0.23 / 0.88 / 1.09)

    # ---- finding 3: the same test with the built-in table in every render rejects the reference on the edges (and only the reference)
    kf = 8
    ffix = _draws(hip, fs, cam, spp, 1, kf, [None] * (kf + 1))
    half = np.stack([box2(f).astype(np.float32) for f in ffix])
    muf, sdf = half[:kf].mean(axis=0).astype(np.float64), half[:kf].std(axis=0, ddof=1).astype(np.float64)
    af, cf = _z_stats(r, muf, sdf, everywhere, kf), _z_stats(half[kf].astype(np.float64), muf, sdf, everywhere, kf)
    print(f"  built-in table in every render (K={kf}): reference |z|>6 {af['gt6']:.2e}, control {cf['gt6']:.2e}")
    assert af["gt6"] >= 20.0 * max(a["gt6"], 2e-5) and cf["gt6"] <= 1e-3, (af, cf)

    # ---- the source at HEAD (receiver's alpha), on the pixels where the two semantics' MEANS over the same K renders (same seeds,
    # same tables) differ by at most 0.02 LSB: what the one line that separates them cannot explain there is tested as above
    if not era_mask.any():
        return   # every material of this scene has alpha 1: the two semantics are the same frame
    fhead = _draws(hip, fs, cam, spp, 0, k, tables)
    mu_h, sd_h, ctrl_h = mu_sd_ctrl(fhead)
    touched = np.abs(mu_h - mu).max(axis=-1) > 0.02
    p = np.pad(touched, 1)
    touched = np.logical_or.reduce([p[1 + dy:p.shape[0] - 1 + dy, 1 + dx:p.shape[1] - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
    keep = np.repeat(~touched[..., None], 3, axis=2)
    share = float(keep.mean())
    if share < 0.05:   # room_spheres: a closed room around three translucent spheres -- their shadows reach every pixel through the bounces
        print(f"  HEAD semantic: the two semantics agree to 0.02 LSB on {100.0 * share:.1f} % of the frame only; nothing to test apart from the 2022 form")
        return
    a, c = _z_stats(r, mu_h, sd_h, keep, k), _z_stats(ctrl_h, mu_h, sd_h, keep, k)
    print(f"  HEAD semantic on the {100.0 * share:.0f} % of the frame where the two semantics agree to 0.02 LSB\n  reference {a}\n  control   {c}")
    assert abs(a["mean_z"]) <= 0.04 and abs(a["bias"]) <= 0.04 and a["sd_z"] <= 1.5 * c["sd_z"], (a, c)
    assert a["gt4"] <= max(20.0 * c["gt4"], 5e-4) and a["gt6"] <= 1e-4 and a["block_max"] <= 0.6, (a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["room_spheres", "floor_monkey"])
def test_compat_occluder_alpha_is_the_oracles_2022_switch(hip, oracle, name):
    """rr_scene_set_compat(RR_COMPAT_OCCLUDER_ALPHA_SHADOWS) on the product == the oracle's rro_set_shot_era(1), +-1 LSB with equal
    ray counts; switching it off again restores HEAD's frame bit for bit; unknown flags are refused."""
    fs = scene_of_2022(name)
    cam = camera_for(fs, 320, 180).c_struct()
    cfg = make_config(samples=4, monte_carlo=True, seed=11)
    with hip.DeviceScene(fs, 0) as ds:
        head = ds.render(cam, cfg)
        ds.set_compat(1)
        era = ds.render(cam, cfg)
        st = ds.stats()
        ds.set_compat(0)
        again = ds.render(cam, cfg)
        with pytest.raises(hip.RustrayHipError):
            ds.set_compat(2)
    assert np.array_equal(head["rgba"], again["rgba"])
    assert not np.array_equal(head["rgba"], era["rgba"])   # both scenes hold translucent occluders
    oracle.lib().rro_set_shot_era(1)
    try:
        o = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True)
    finally:
        oracle.lib().rro_set_shot_era(0)
    res = compare_frames(era, o)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res
    c = o["counters"]
    assert st["primary_rays"] + st["secondary_rays"] == c["rays_primary"] + c["rays_secondary"] and st["shaded_hits"] == c["shaded_hits"]
