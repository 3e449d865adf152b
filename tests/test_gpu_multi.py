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


def test_packed_deinterleave_equals_the_per_buffer_form(hip):
    """rr_deinterleave_packed_device (all buffers of all ranks' packs in one launch, what rank 0 runs on its gather target) against
    rr_render's frame, for 1 and 3 ranks whose packs are laid out exactly as TiledFrame lays them out."""
    import torch
    from rustray_amd.renderer import TiledFrame, render_region_torch
    fs = load_scene("spheres")
    w, h = 100, 61
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=8)
    with hip.DeviceScene(fs, 0) as ds:
        whole = ds.render(cam, cfg)
        for world in (1, 3):
            tfs = [TiledFrame(w, h, r, world, 32, 8) for r in range(world)]
            for tf in tfs:
                render_region_torch(ds, cam, cfg, tf, aux=True)
            torch.cuda.synchronize()
            tf0 = tfs[0]
            packs = torch.stack([tf._pack for tf in tfs]).contiguous()       # what dist.gather leaves on rank 0
            names = [sp[0] for sp in TiledFrame._SPEC]
            hip.deinterleave_packed_device(w, h, 32, 8, world, packs.data_ptr(), packs.stride(0), [tf0._section[n][0] for n in names],
                                           [tf0._section[n][1] for n in names], [tf0._frame[n].data_ptr() for n in names], 0,
                                           torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            for key in ("rgba", "normal", "depth", "object_id"):
                got = tf0._frame[key].cpu().numpy().reshape(whole[key].shape)
                assert np.array_equal(got.view(np.uint8), whole[key].view(np.uint8)), (world, key)
            # and through TiledFrame.gather itself for the one-rank case
            if world == 1:
                out = tf0.gather(tf0._parts, use_device_kernel=True)
                assert np.array_equal(out["rgba"].cpu().numpy().reshape(h, w, 4), whole["rgba"])
