"""Nothing unwinds across the ABI, on the device path itself: the test-only fault hook makes the real entry
points throw -- on the calling thread, in a mesh-tree worker of rr_scene_create, in a device worker of rr_render_multi -- and every
call must come back with a status code, leave the handles usable and leak no device memory."""
import ctypes as C

import numpy as np
import pytest

from rustray_amd import capi
from rustray_amd.flat import make_config
from tests.helpers import camera_for, load_scene

pytestmark = pytest.mark.gpu


def _fault(point, kind, skip=0):
    L = capi.lib()
    L.rr_test_fault.argtypes = [C.c_char_p, C.c_int, C.c_int]
    assert L.rr_test_fault(point.encode(), kind, skip) == 0


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_scene_create_survives_exceptions_and_frees_what_it_holds():
    fs = load_scene("kbert_room")
    with capi.DeviceScene(fs, 0):
        pass   # first use of the device: runtime allocations settle
    free0 = _free_bytes()
    for point, kind, code in (("scene_create.host", 1, -5), ("scene_create.host", 2, -4), ("scene_create.mesh_worker", 1, -5),
                              ("scene_create.mesh_worker", 2, -4), ("scene_create.mesh_worker", 3, -4)):
        _fault(point, kind, 1 if "worker" in point else 0)
        with pytest.raises(capi.RustrayHipError) as e:
            capi.DeviceScene(fs, 0)
        assert e.value.code == code, (point, kind, str(e.value))
    _fault("", 0)
    assert abs(_free_bytes() - free0) < (8 << 20)
    with capi.DeviceScene(fs, 0) as ds:   # and the next create works
        cam = camera_for(fs, 64, 48).c_struct()
        assert ds.render(cam, make_config(samples=1, monte_carlo=False))["rgba"][..., 3].min() == 255


def test_frame_entry_points_survive_exceptions():
    fs = load_scene("kbert_room")
    cam = camera_for(fs, 96, 64).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=3)
    with capi.DeviceScene(fs, 0) as a, capi.DeviceScene(fs, 0) as b, capi.DeviceScene(fs, 0) as c:
        ref = a.render(cam, cfg)
        # a device worker of rr_render_multi throws: bad_alloc, then a runtime_error in the second worker to start
        for kind, skip, code in ((1, 0, -5), (2, 1, -4), (3, 2, -4)):
            _fault("render_multi.worker", kind, skip)
            with pytest.raises(capi.RustrayHipError) as e:
                capi.render_multi([a, b, c], cam, cfg)
            assert e.value.code == code and "device slot" in str(e.value)
        _fault("", 0)
        out = capi.render_multi([a, b, c], cam, cfg)   # every handle was unlocked and is usable
        assert np.array_equal(out["rgba"], ref["rgba"])
        # host vectors of rr_trace_rays and rr_scene_update_transforms
        o = np.zeros((16, 3), np.float32); d = np.tile(np.array([[0, 0, -1]], np.float32), (16, 1))
        _fault("trace_rays.host", 1)
        with pytest.raises(capi.RustrayHipError) as e:
            a.trace_rays(o, d)
        assert e.value.code == -5 and "rr_trace_rays" in str(e.value)
        a.trace_rays(o, d)
        t = np.stack([np.asarray(it.trans, np.float32) for it in fs.items])
        ti = np.stack([np.asarray(it.trans_inv, np.float32) for it in fs.items])
        _fault("update_transforms.host", 2)
        with pytest.raises(capi.RustrayHipError) as e:
            a.update_transforms(t, ti)
        assert e.value.code == -4
        _fault("", 0)
        a.update_transforms(t, ti)
        assert np.array_equal(a.render(cam, cfg)["rgba"], ref["rgba"])
