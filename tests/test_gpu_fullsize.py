"""Full-size frames (BASELINE.json sizes) checked through size-independent properties: the oracle would
need minutes for these, so they are checked by determinism, tiling invariance, transform round trips and
a one-tile oracle spot check."""
import numpy as np
import pytest

from rustray_amd.flat import make_config
from tests.helpers import camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def test_c2_monkey_full_size(hip, oracle):
    """BASELINE C2: monkey 800x600 16 spp monte_carlo=1."""
    fs = load_scene("monkey")
    cam = camera_for(fs, 800, 600).c_struct()
    cfg = make_config(samples=16, monte_carlo=True, seed=0)
    with hip.DeviceScene(fs, 0) as ds:
        a = ds.render(cam, cfg)
        b = ds.render(cam, cfg)
        st = ds.stats()
    assert np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["depth"], b["depth"])        # idempotent, bit for bit
    assert st["primary_rays"] == 800 * 600 * 16
    win = (368, 268, 432, 332)                                                                     # 64x64 tile through the oracle
    ref = oracle.render(fs.c_struct(), cam, cfg, window=win, n_threads=16)
    x0, y0, x1, y1 = win
    r = compare_frames({k: v[y0:y1, x0:x1] for k, v in a.items()}, {k: v[y0:y1, x0:x1] for k, v in ref.items()})
    assert r["n_rgb_over"] == 0 and r["n_id_diff"] == 0, r


def test_c4_sponza_syn_full_frame_properties(hip, oracle):
    """BASELINE C4 stand-in at 1280x720 (8 spp to bound the run): region tiling for 8 ranks reproduces the
    single-call frame bit for bit, and one tile matches the oracle."""
    import torch
    from rustray_amd import synthetic
    from rustray_amd.camera import Camera
    from rustray_amd.renderer import TiledFrame, region_pixels, render_region_torch
    fs = synthetic.sponza_syn()
    cam = Camera.from_state(fs.meta["camera"]).c_struct()
    cfg = make_config(samples=8, monte_carlo=True, seed=0)
    with hip.DeviceScene(fs, 0) as ds:
        whole = ds.render(cam, cfg, aux=False)
        frame = torch.zeros((720 * 1280, 4), dtype=torch.uint8, device="cuda")
        for r in range(8):
            tf = TiledFrame(1280, 720, r, 8, 32, 8)
            part = render_region_torch(ds, cam, cfg, tf)["rgba"]
            xy = region_pixels(1280, 720, 32, 8, 8, r)
            frame[torch.from_numpy(xy[:, 1] * 1280 + xy[:, 0]).cuda()] = part
        torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(720, 1280, 4)
    assert np.array_equal(got, whole["rgba"])
    win = (608, 400, 672, 432)
    ref = oracle.render(fs.c_struct(), cam, cfg, window=win, n_threads=16)
    x0, y0, x1, y1 = win
    d = np.abs(whole["rgba"][y0:y1, x0:x1, :3].astype(int) - ref["rgba"][y0:y1, x0:x1, :3].astype(int))
    assert d.max() <= 1


# ---------------------------------------------------------------------------
# BASELINE configs C3, C4, C5 at their full size and sample count (synthetic stand-ins: the .glb assets are absent offline)
# ---------------------------------------------------------------------------
def _full_size_config(hip, oracle, fs, spp, tile_win, n_ranks=8):
    """Same recipe as C2: determinism bit for bit, the primary-ray count, an n_ranks tiling through rr_render_multi
    (handles on device 0) that reproduces the single-call frame bit for bit, and one 64x32 tile through the oracle AT
    THE CONFIG'S REAL SAMPLE COUNT."""
    from rustray_amd.camera import Camera
    st = dict(fs.meta["camera"]); st["width"], st["height"] = 1280, 720
    cam = Camera.from_state(st).c_struct()
    cfgd = fs.meta.get("config") or {}
    cfg = make_config(samples=spp, monte_carlo=True, seed=0, max_recursion=6,
                      focal_length=cfgd.get("focal_length", 1.0), aperture_size=cfgd.get("aperture_size", 1.0))
    with hip.DeviceScene(fs, 0) as ds:
        a = ds.render(cam, cfg)
        st_a = ds.stats()
        b = ds.render(cam, cfg)
        assert st_a["primary_rays"] == 1280 * 720 * spp
        for k in ("rgba", "depth", "object_id"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["normal"], b["normal"], equal_nan=True)
        # the frame tiled over n_ranks device slots (all on this GPU) through the one-process multi-GPU entry
        others = [hip.DeviceScene(fs, 0) for _ in range(n_ranks - 1)]
        try:
            m = hip.render_multi([ds] + others, cam, cfg)
        finally:
            for o in others:
                o.close()
        for k in ("rgba", "depth", "object_id"):
            assert np.array_equal(a[k], m[k]), k
    x0, y0, x1, y1 = tile_win
    ref = oracle.render(fs.c_struct(), cam, cfg, window=tile_win, n_threads=16)
    r = compare_frames({k: v[y0:y1, x0:x1] for k, v in a.items()}, {k: v[y0:y1, x0:x1] for k, v in ref.items()})
    assert r["n_rgb_over"] == 0 and r["n_id_diff"] == 0 and r["nan_mismatch"] == 0, r
    return st_a


def test_c3_helmet_syn_full_size(hip, oracle):
    """BASELINE C3 stand-in: helmet_syn 1280x720, 64 spp, monte_carlo=1 (base + normal + roughness + AO maps, bilinear)."""
    from rustray_amd import synthetic
    st = _full_size_config(hip, oracle, synthetic.helmet_syn(), 64, (600, 330, 664, 362))
    assert st["secondary_rays"] > 0 and st["shadow_rays"] > 0


def test_c4_sponza_syn_full_size_128spp(hip, oracle):
    """BASELINE C4 stand-in at its real sample count: sponza_syn 1280x720, 128 spp, monte_carlo=1 (194 items: top-level tree)."""
    from rustray_amd import synthetic
    _full_size_config(hip, oracle, synthetic.sponza_syn(), 128, (608, 400, 672, 432))


def test_c5_lotus_syn_full_size_512spp_dof(hip, oracle):
    """BASELINE C5 stand-in: lotus_syn 1280x720, 512 spp, monte_carlo=1, depth of field (glass + reflective floor: deep trees)."""
    from rustray_amd import synthetic
    fs = synthetic.lotus_syn()
    assert fs.meta["config"]["aperture_size"] > 1.0 and fs.meta["config"]["focal_length"] > 1.0
    st = _full_size_config(hip, oracle, fs, 512, (608, 420, 672, 452), n_ranks=4)
    assert st["secondary_rays"] > st["primary_rays"] // 4
