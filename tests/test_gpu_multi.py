"""rr_render_multi: one frame over several scene handles from ONE host process (the form a Rust host would call,
reference src/renderer.rs:105-172 is one process).  A one-GPU box rehearses it with several handles on device 0: the
tiling, the peer copies into slot 0's concatenation, the de-interleave and the host copy are the same code."""
import numpy as np
import pytest

from rustray_amd.flat import make_config
from tests.helpers import camera_for, load_scene

pytestmark = pytest.mark.gpu


def _same(a, b):
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["normal"], b["normal"], equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_multi_handle_frame_equals_rr_render(hip, n):
    fs = load_scene("spheres_room")
    cam = camera_for(fs, 200, 113).c_struct()   # not a multiple of the 32x8 tile: clipped tiles at the right and bottom borders
    cfg = make_config(samples=3, monte_carlo=True, seed=5)
    scenes = [hip.DeviceScene(fs, 0) for _ in range(n)]
    try:
        ref = scenes[0].render(cam, cfg)
        out = hip.render_multi(scenes, cam, cfg)
        _same(out, ref)
        rgba_only = hip.render_multi(scenes, cam, cfg, aux=False)
        assert np.array_equal(rgba_only["rgba"], ref["rgba"])
        _same(hip.render_multi(scenes[::-1], cam, cfg), ref)   # any handle can be slot 0
    finally:
        for s in scenes:
            s.close()


def test_multi_rejects_bad_arguments(hip):
    fs = load_scene("spheres")
    cam = camera_for(fs, 32, 32).c_struct()
    cfg = make_config(samples=1)
    with hip.DeviceScene(fs, 0) as ds:
        with pytest.raises(hip.RustrayHipError):
            hip.render_multi([ds, ds], cam, cfg)        # the same handle twice
        with pytest.raises(hip.RustrayHipError):
            hip.render_multi([], cam, cfg)
