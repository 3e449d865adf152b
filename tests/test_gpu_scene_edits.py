"""Scene life cycle through the C ABI on the GPU: degenerate scenes (empty meshes, no items), repeated create / destroy
without leaking HBM, material edits between frames (rr_scene_update_materials vs a freshly created scene)."""
import copy
import ctypes as C

import numpy as np
import pytest

from rustray_amd.flat import Item, MeshData, make_config
from tests.helpers import camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def _same(a, b):
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["normal"], b["normal"], equal_nan=True)


def test_empty_mesh_item_is_legal_and_invisible(hip, oracle):
    """A mesh without faces (an OBJ group or glTF primitive with no triangles) used to crash the host-side BVH collapse."""
    fs = load_scene("spheres")
    base = copy.deepcopy(fs)
    fs.meshes.append(MeshData(positions=np.zeros((0, 3), np.float32), indices=np.zeros((0, 3), np.uint32)))
    it = copy.deepcopy(fs.items[0])
    it.kind, it.mesh, it.id, it.name = 1, len(fs.meshes) - 1, 999, "empty"
    it.bbox_min, it.bbox_max = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    fs.items.append(it)
    cam = camera_for(fs, 96, 96).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=4)
    with hip.DeviceScene(fs, 0) as ds, hip.DeviceScene(base, 0) as ds0:
        out, ref = ds.render(cam, cfg), ds0.render(cam, cfg)
        _same(out, ref)
    res = compare_frames(out, oracle.render(fs.c_struct(), cam, cfg, n_threads=4))
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_scene_without_items_renders_black_and_accepts_updates(hip):
    fs = load_scene("spheres")
    fs.items = []
    cam = camera_for(fs, 64, 48).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=1)
    with hip.DeviceScene(fs, 0) as ds:
        ds.update_transforms(np.zeros((0, 4, 4), np.float32), np.zeros((0, 4, 4), np.float32))
        out = ds.render(cam, cfg)
        st = ds.stats()
    assert (out["rgba"][..., :3] == 0).all() and (out["rgba"][..., 3] == 255).all() and (out["object_id"] == 0).all()
    assert st["primary_rays"] == 64 * 48 * 2 and st["shaded_hits"] == 0 and st["shadow_rays"] == 0


def test_create_destroy_does_not_leak_device_memory(hip):
    """The BVH4 nodes and the precomputed triangles (the largest scene buffers) were not freed by rr_scene_destroy."""
    rt = C.CDLL("libamdhip64.so")   # the HIP runtime the library itself is linked against

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert rt.hipDeviceSynchronize() == 0 and rt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    fs = load_scene("monkey")
    cam = camera_for(fs, 64, 48).c_struct()
    cfg = make_config(samples=1, monte_carlo=False, seed=0)

    def cycle():
        with hip.DeviceScene(fs, 0) as ds:
            ds.render(cam, cfg)
    cycle()  # first use: code objects, the de-interleave maps, torch's context
    lost = []
    for _ in range(3):   # a leak loses memory in EVERY window; the HIP runtime growing one of its own pools (seen: 16 MiB, once) shows in one
        free0 = free_bytes()
        for _ in range(12):
            cycle()
        lost.append(free0 - free_bytes())
    assert min(lost) < (4 << 20), f"{[round(v / 2**20, 1) for v in lost]} MiB lost over three windows of 12 create / destroy cycles"


def test_material_edits_in_place_equal_a_fresh_scene(hip, oracle):
    fs = load_scene("spheres_room")
    cam = camera_for(fs, 128, 72).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=9)
    edited = copy.deepcopy(fs)
    touched = 0
    for i, it in enumerate(edited.items):   # edit the full material AND its cache, as Material::apply_diff + update_material_cache do
        for idx in (it.material, it.material_cache):
            m = edited.materials[idx]
            if i % 3 == 0:
                m.alpha, m.refraction_index = 0.6, 1.3
            elif i % 3 == 1:
                m.reflectivity, m.base_color = 0.35, (0.9, 0.4, 0.2)
            else:
                m.cast_shadow = False
            touched += 1
    assert touched >= 6
    with hip.DeviceScene(fs, 0) as ds, hip.DeviceScene(edited, 0) as fresh:
        before = ds.render(cam, cfg)
        ds.update_materials(edited.materials)
        after = ds.render(cam, cfg)
        ref = fresh.render(cam, cfg)
        _same(after, ref)
        assert not np.array_equal(before["rgba"], after["rgba"])
        ds.update_materials(fs.materials)   # and back
        _same(ds.render(cam, cfg), before)
        with pytest.raises(hip.RustrayHipError):
            ds.update_materials(edited.materials[:-1])
    res = compare_frames(after, oracle.render(edited.c_struct(), cam, cfg, n_threads=8))
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_tuning_is_validated(hip):
    from rustray_amd.flat import rr_tuning
    with hip.DeviceScene(load_scene("spheres"), 0) as ds:
        t = rr_tuning()
        assert hip.lib().rr_scene_set_tuning(ds._h, C.byref(t)) == -1          # struct_size 0
        t.struct_size = C.sizeof(rr_tuning); t.sample_group = 3
        assert hip.lib().rr_scene_set_tuning(ds._h, C.byref(t)) == -1          # not a power of two
        ds.set_tuning(sample_group=4, kernel_timing=1)
        g = rr_tuning()
        assert hip.lib().rr_scene_get_tuning(ds._h, C.byref(g)) == 0 and g.sample_group == 4 and g.kernel_timing == 1
