"""The C++ host layer (include/rustray_host.hpp: Camera, RaytracingConfig, Raytracing, RendererManager restated from
the reference's Rust host code) driven through rustray_amd/csrc/host_shim.cpp."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from rustray_amd import capi
from rustray_amd.camera import Camera
from rustray_amd.flat import make_config, rr_config, rr_flat_scene
from tests.helpers import camera_for, load_scene

SHIM = os.path.join(os.path.dirname(capi.LIB_PATH), "librustray_host_shim.so")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        pytest.fail(f"{SHIM} is missing: run `make -C rustray_amd/csrc`")
    capi.lib()  # librustray_hip.so first (the shim links against it)
    L = C.CDLL(SHIM)
    F3 = C.c_float * 3
    L.rh_camera.argtypes = [C.c_float, F3, F3, F3, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                            C.POINTER(C.c_int), F3, C.POINTER(C.c_int)]
    L.rh_config_apply.argtypes = [C.POINTER(rr_config), C.POINTER(rr_config), C.POINTER(rr_config)]
    L.rh_render.argtypes = [C.POINTER(rr_flat_scene), C.c_int, C.c_float, F3, F3, F3, C.c_float, C.c_float, C.POINTER(rr_config),
                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


def _cam_args(cam: Camera):
    F3 = C.c_float * 3
    return (C.c_float(cam.fov), F3(*map(float, cam.eye_pos)), F3(*map(float, cam.up)), F3(*map(float, cam.dir)),
            C.c_float(cam.clipping_near), C.c_float(cam.clipping_far))


@pytest.mark.parametrize("state", [
    dict(fov=math.radians(90.0), eye_pos=[0, 0, 0], up=[0, 1, 0], dir=[0, 0, -1], clipping_near=0.001, clipping_far=1000.0),
    dict(fov=math.radians(47.5), eye_pos=[1.5, 2.25, 7.0], up=[0, 1, 0], dir=[-0.3, -0.2, -1.0], clipping_near=0.1, clipping_far=100.0),
    dict(fov=math.radians(70.0), eye_pos=[-4.0, 0.5, -2.0], up=[0.1, 1, 0], dir=[1.0, 0.1, 0.4], clipping_near=0.05, clipping_far=500.0),
])
def test_camera_matrices_equal_the_python_mirror(shim, state):
    """Camera::init / init_matrices (reference src/camera.rs:69-90): both host mirrors hand the library the same bits."""
    st = dict(state, width=1280, height=720)
    st["fov"] = float(np.float32(st["fov"]))
    cam = Camera.from_state(st)
    ref = cam.c_struct()
    pi, vi = (C.c_float * 16)(), (C.c_float * 16)()
    isdef, inside = C.c_int(), C.c_int()
    pt = np.asarray(cam.eye_pos, np.float64) + 3.0 * np.asarray(cam.dir, np.float64)
    shim.rh_camera(*_cam_args(cam), 1280, 720, pi, vi, C.byref(isdef), (C.c_float * 3)(*map(float, pt)), C.byref(inside))
    assert list(pi) == list(ref.projection_inverse) and list(vi) == list(ref.view_inverse)
    assert bool(isdef.value) == cam.is_default_cam()
    assert bool(inside.value) == cam.points_in_frustum(np.asarray([pt])) is True


def test_config_apply_takes_over_only_non_default_fields(shim):
    """RaytracingConfig::apply (reference src/raytracing.rs:129-185)."""
    base = make_config(samples=16, monte_carlo=True, focal_length=8.0, aperture_size=4.0, fog_density=0.1, max_recursion=3, gamma_correction=True)
    base.fog_color[:] = [0.1, 0.2, 0.3]
    new = make_config(samples=1, monte_carlo=False, focal_length=1.0, aperture_size=2.0, fog_density=0.0, max_recursion=6, gamma_correction=False)
    new.fog_color[:] = [0.4, 0.4, 0.4]                     # the default colour: not taken over
    out = rr_config()
    shim.rh_config_apply(C.byref(base), C.byref(new), C.byref(out))
    assert (out.samples, out.monte_carlo, out.max_recursion, out.gamma_correction) == (16, 1, 3, 1)   # defaults in `new` do not overwrite
    assert out.focal_length == 8.0 and out.aperture_size == 2.0 and abs(out.fog_density - 0.1) < 1e-7
    assert [round(v, 6) for v in out.fog_color] == [0.1, 0.2, 0.3]
    new2 = make_config(samples=4, monte_carlo=True, max_recursion=2, gamma_correction=True, fog_density=0.5)
    new2.fog_color[:] = [0.9, 0.4, 0.4]
    shim.rh_config_apply(C.byref(make_config()), C.byref(new2), C.byref(out))
    assert (out.samples, out.monte_carlo, out.max_recursion, out.gamma_correction) == (4, 1, 2, 1)
    assert abs(out.fog_density - 0.5) < 1e-7 and [round(v, 6) for v in out.fog_color] == [0.9, 0.4, 0.4]


def _render(shim, fs, cam, cfg, w, h, min_passes, stop_after=0, pick=None):
    rgba = np.zeros((h, w, 4), np.uint8); normal = np.zeros((h, w, 3), np.float32)
    depth = np.zeros((h, w), np.float32); ids = np.zeros((h, w), np.uint32)
    stats = np.zeros(6, np.uint64); pick_out = np.zeros(3, np.float32)
    cs = fs.c_struct()
    px, py = pick if pick else (0, 0)
    rc = shim.rh_render(C.byref(cs), 0, *_cam_args(cam), C.byref(cfg), w, h, min_passes, stop_after, rgba.ctypes.data, normal.ctypes.data,
                        depth.ctypes.data, ids.ctypes.data, stats.ctypes.data, px, py, pick_out.ctypes.data)
    return rc, dict(rgba=rgba, normal=normal, depth=depth, object_id=ids), stats, pick_out


@pytest.mark.gpu
def test_renderer_manager_frame_equals_the_frame_level_call(shim):
    """RendererManager::start / is_done / get_rendered_pixels / drain (reference src/renderer.rs:105-251) on a worker thread:
    same bits as rr_render through the Python binding; Raytracing::pick agrees too."""
    fs = load_scene("spheres_room")
    w, h = 96, 64
    cam = camera_for(fs, w, h)
    cfg = make_config(samples=16, monte_carlo=True, seed=4)
    with capi.DeviceScene(fs, 0) as ds:
        ref = ds.render(cam.c_struct(), cfg)
        pk = ds.pick(cam.c_struct(), 48, 40)
    rc, out, stats, pick = _render(shim, fs, cam, cfg, w, h, min_passes=4, pick=(48, 40))
    assert rc == 0
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(out["normal"], ref["normal"], equal_nan=True)
    passes, rendered, done, drained, _, was_running = [int(v) for v in stats]
    assert was_running == 1 and done == 1 and rendered == w * h and passes >= 3
    assert drained >= w * h and drained % (w * h) == 0          # whole frames per pass, the last one always delivered
    assert bool(pick[0]) == bool(pk.hit) and int(pick[1]) == pk.object_id and abs(pick[2] - pk.distance) < 1e-6


@pytest.mark.gpu
def test_renderer_manager_stop_ends_the_frame_early(shim):
    """RendererManager::stop (reference src/renderer.rs:174-198): the frame ends after the current pass, is_done stays
    false and the buffers keep the last finished pass."""
    fs = load_scene("spheres_room")
    w, h = 96, 64
    cam = camera_for(fs, w, h)
    cfg = make_config(samples=64, monte_carlo=True, seed=4)
    rc, out, stats, _ = _render(shim, fs, cam, cfg, w, h, min_passes=16, stop_after=2)
    assert rc == 0
    passes, rendered, done = int(stats[0]), int(stats[1]), int(stats[2])
    assert done == 0 and 2 <= passes < 16 and 0 < rendered < w * h
    assert out["rgba"][..., 3].min() == 255 and out["rgba"][..., :3].max() > 0


def test_animation_frames_equal_the_python_mirror(shim):
    """Animation (reference src/animation.rs) + Scene::apply_frame (src/scene.rs:1695-1713): keyframe selection, the
    NaN factor at the last keyframe, interpolation and T * Rz * Ry * Rx * S agree with rustray_amd/animation.py."""
    from rustray_amd.animation import Animation, Frame, Keyframe
    from rustray_amd.flat import FlatScene, Item
    F3 = C.c_float * 3
    shim.rh_animation_frame.argtypes = [C.c_uint32, C.c_uint64, F3, F3, F3, F3, F3, F3, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    tr0, rot0, sc0 = (0.0, 0.0, -10.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    tr1, rot1, sc1 = (1.0, 0.5, -9.0), (0.3, 2.0, -0.7), (1.4, 0.8, 1.1)
    fs = FlatScene()
    base = np.eye(4, dtype=np.float32); base[:3, 3] = (0.5, -1.0, 2.0)
    for i in range(3):
        fs.items.append(Item(kind=0, id=i + 1, material=0, material_cache=0, radius=1.0, trans=base.copy(), trans_inv=np.linalg.inv(base).astype(np.float32),
                             bbox_min=(-1, -1, -1), bbox_max=(1, 1, 1), name=f"item{i}"))
    an = Animation(True, 7, [Keyframe(0, [Frame("item1", tr0, rot0, sc0)]), Keyframe(1500, [Frame("item1", tr1, rot1, sc1)])])
    for frame in (0, 3, 9, 10, 11):
        trans = np.stack([base.T.copy() for _ in range(3)]).astype(np.float32)      # column-major, as the ABI takes them
        inv = np.zeros_like(trans)
        amount, exists = C.c_uint64(), C.c_int()
        touched = shim.rh_animation_frame(7, 1500, F3(*tr0), F3(*rot0), F3(*sc0), F3(*tr1), F3(*rot1), F3(*sc1), 3, 1, frame,
                                          trans.ctypes.data, inv.ctypes.data, C.byref(amount), C.byref(exists))
        ref = an.frame_transforms(fs, frame)
        assert amount.value == an.get_frames_amount_to_render() == 10 and bool(exists.value) == an.frame_exists(frame)
        assert bool(touched) == (ref is not None)
        if ref is None:
            continue
        rt, ri = ref
        got_t = np.transpose(trans, (0, 2, 1)); got_i = np.transpose(inv, (0, 2, 1))
        assert np.allclose(got_t, rt, rtol=2e-6, atol=2e-6, equal_nan=True), frame
        assert np.allclose(got_i, ri, rtol=2e-5, atol=2e-5, equal_nan=True), frame
        assert np.array_equal(got_t[0], base) and np.array_equal(got_t[2], base)        # items without keyframes keep their matrix


@pytest.mark.gpu
def test_renderer_manager_post_processing_equals_the_abi_call(shim):
    """Run::post_processing (reference src/run.rs:588-600) through the C++ layer == rr_post_process on the same frame."""
    fs = load_scene("spheres_room")
    w, h = 96, 64
    cam = camera_for(fs, w, h)
    cfg = make_config(samples=4, monte_carlo=True, seed=4)
    with capi.DeviceScene(fs, 0) as ds:
        ref = ds.render(cam.c_struct(), cfg)
    want = capi.post_process(ref["rgba"], ref["normal"], ref["object_id"], cavity=True, outline=True)
    rc, out, _, _ = _render(shim, fs, cam, cfg, w, h, min_passes=2, stop_after=-1)
    assert rc == 0 and np.array_equal(out["rgba"], want) and not np.array_equal(want, ref["rgba"])
