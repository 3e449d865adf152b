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
