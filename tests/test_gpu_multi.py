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


def test_multi_reports_how_the_buffers_travelled_and_concurrent_callers_do_not_deadlock(hip):
    """rr_scene_last_stats(scenes[0]) after rr_render_multi: handles, direct (peer / same device) and host-staged links.
    Two threads calling with the SAME handles in opposite orders serialise (address-ordered locks) and both get the frame."""
    import threading
    fs = load_scene("spheres")
    cam = camera_for(fs, 96, 64).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=3)
    scenes = [hip.DeviceScene(fs, 0) for _ in range(3)]
    try:
        ref = scenes[0].render(cam, cfg)
        hip.render_multi(scenes, cam, cfg)
        st = scenes[0].stats()
        assert st["multi_devices"] == 3 and st["multi_peer_links"] == 2 and st["multi_staged_links"] == 0
        assert st["ms_multi_exchange"] > 0.0
        scenes[0].render(cam, cfg)
        assert scenes[0].stats()["multi_devices"] == 0   # a single-handle frame resets it
        # the path between devices WITHOUT peer access (device -> pinned host -> device 0), forced here where one GPU has to play both
        scenes[0].set_tuning(multi_force_staged=1)
        staged = hip.render_multi(scenes, cam, cfg)
        st = scenes[0].stats()
        assert st["multi_peer_links"] == 0 and st["multi_staged_links"] == 2
        _same(staged, ref)
        scenes[0].set_tuning(multi_force_staged=0)
        results, errors = {}, []

        def run(tag, order):
            try:
                for _ in range(4):
                    results[tag] = hip.render_multi(order, cam, cfg)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        ts = [threading.Thread(target=run, args=("fwd", scenes)), threading.Thread(target=run, args=("rev", scenes[::-1]))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts), "rr_render_multi deadlocked on handles passed in opposite orders"
        assert not errors, errors
        _same(results["fwd"], ref)
        _same(results["rev"], ref)
    finally:
        for s in scenes:
            s.close()
