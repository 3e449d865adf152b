"""Edges of the frame call on the GPU: smallest and most lopsided frames, the largest accepted recursion depth and sample
count, scenes without lights / with many lights, every light disabled, and argument errors that must come back as codes."""
import copy

import numpy as np
import pytest

from rustray_amd.flat import Light, make_config
from tests.helpers import camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def _check(hip, oracle, fs, w, h, cfg, threads=8):
    cam = camera_for(fs, w, h).c_struct()
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=threads, want_counters=True)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    c = ref["counters"]
    assert st["primary_rays"] == c["rays_primary"] and st["secondary_rays"] == c["rays_secondary"] and st["shaded_hits"] == c["shaded_hits"]
    return out, st


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (1, 1, 64), (3, 1, 5), (1, 7, 2), (257, 2, 3), (2, 129, 1)])
def test_tiny_and_lopsided_frames(hip, oracle, w, h, spp):
    _check(hip, oracle, load_scene("spheres"), w, h, make_config(samples=spp, monte_carlo=True, seed=2))


def test_deepest_recursion_the_abi_accepts(hip, oracle):
    """max_recursion 16 on the branching spheres scene (two glass spheres facing each other keep every path alive to the last
    level), and RR_MAX_RECURSION = 30 (31 depth levels: five bits of depth in a shadow record) on mirrors that spawn one child per
    hit; 31 is refused."""
    fs = load_scene("spheres")
    out, st = _check(hip, oracle, fs, 48, 48, make_config(samples=1, monte_carlo=False, seed=0, max_recursion=16))
    assert st["secondary_rays"] > 4 * st["primary_rays"]
    mirrors = load_scene("spheres_room")              # a closed room: no ray leaves
    for m in mirrors.materials:
        m.alpha, m.reflectivity = 1.0, 0.9        # no refraction child: chains of 30 reflections, not trees
    out, st = _check(hip, oracle, mirrors, 24, 24, make_config(samples=2, monte_carlo=True, seed=1, max_recursion=30))
    assert st["secondary_rays"] == 30 * st["primary_rays"]
    with hip.DeviceScene(fs, 0) as ds:
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(camera_for(fs, 8, 8).c_struct(), make_config(samples=1, max_recursion=31))
        assert e.value.code == -2


def test_many_samples_on_a_small_frame(hip, oracle):
    """4096 samples per pixel: a 4096-cell table shuffled out of 4096^2 / 4 cells, 64-sample packets of one pixel."""
    fs = load_scene("spheres")
    cam = camera_for(fs, 8, 6).c_struct()
    cfg = make_config(samples=4096, monte_carlo=True, seed=5)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        assert ds.stats()["primary_rays"] == 8 * 6 * 4096
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=16383))          # beyond the BUILT-IN table's limit (RR_MAX_SAMPLES)
        assert e.value.code == -2
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_the_references_own_sample_limit_with_the_callers_table(hip, oracle):
    """With the caller's sub-sample table the reference's own limit applies: 32766 samples (the u16 `(samples + 2).next_power_of_two()`
    overflows from 32767 on, src/raytracing.rs:297).  4x3 pixels, cell_size 16384, a table of random cells; 32767 is refused."""
    fs = load_scene("spheres")
    cam = camera_for(fs, 4, 3).c_struct()
    n = 32766
    rng = np.random.default_rng(8)
    table = rng.integers(0, 16384, (n, 2)).astype(np.uint16)
    cfg = make_config(samples=n, monte_carlo=True, seed=9, max_recursion=2)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg, sample_xy=table)
        assert ds.stats()["primary_rays"] == 12 * n
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=32767), sample_xy=np.zeros((32767, 2), np.uint16))
        assert e.value.code == -2
    ref = oracle.render(fs.c_struct(), cam, cfg, sample_xy=table, n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_lights_none_many_and_disabled(hip, oracle):
    base = load_scene("monkey_room")
    cfg = make_config(samples=2, monte_carlo=True, seed=6)
    dark = copy.deepcopy(base); dark.lights = []
    out, st = _check(hip, oracle, dark, 64, 36, cfg)
    assert st["shadow_rays"] == 0
    off = copy.deepcopy(base)
    for l in off.lights:
        l.enabled = False
    out_off, st_off = _check(hip, oracle, off, 64, 36, cfg)
    assert st_off["shadow_rays"] == 0 and np.array_equal(out_off["rgba"], out["rgba"])
    many = copy.deepcopy(base)
    rng = np.random.default_rng(3)
    many.lights = [Light(pos=tuple(rng.uniform(-4, 4, 3) + np.array([0, 3, -8])), dir=(0.0, -1.0, 0.0), color=tuple(rng.uniform(0.2, 1.0, 3)),
                         intensity=float(rng.uniform(5, 30)), light_type=int(i % 3), max_angle=0.9, enabled=bool(i % 5)) for i in range(24)]
    out_many, st_many = _check(hip, oracle, many, 64, 36, cfg)
    assert st_many["shadow_rays"] > 5 * st_many["shaded_hits"]   # 19 enabled lights; a light with a zero term casts no ray (D7)


def test_bad_frame_arguments_come_back_as_codes(hip):
    fs = load_scene("spheres")
    with hip.DeviceScene(fs, 0) as ds:
        cam = camera_for(fs, 16, 16).c_struct()
        for bad in (dict(samples=0),):
            with pytest.raises(hip.RustrayHipError) as e:
                ds.render(cam, make_config(**bad))
            assert e.value.code == -1
        cam.width = 0
        with pytest.raises(hip.RustrayHipError):
            ds.render(cam, make_config(samples=1))
        cam = camera_for(fs, 16, 16).c_struct()
        cam.view_inverse[5] = float("nan")
        with pytest.raises(hip.RustrayHipError):
            ds.render(cam, make_config(samples=1))
        ok = ds.render(camera_for(fs, 16, 16).c_struct(), make_config(samples=1))   # the handle survives every rejected call
        assert ok["rgba"].shape == (16, 16, 4)


def _many_items_scene(n_spheres, n_quads, seed=3):
    """n_spheres small spheres + n_quads two-triangle meshes (all sharing ONE mesh: instancing) on a jittered grid over a floor."""
    from rustray_amd.flat import FlatScene, Item, Material, MeshData
    from rustray_amd.scene import Scene, get_transformation, inverse_affine
    rng = np.random.default_rng(seed)
    fs = FlatScene()
    fs.name = f"many{n_spheres + n_quads}"
    eye4 = np.eye(4, dtype=np.float32)
    mats = []
    for k in range(6):
        m = Material(base_color=tuple(rng.uniform(0.2, 1.0, 3)), specular_color=(0.3, 0.3, 0.3), shininess=40.0)
        if k == 1:
            m.reflectivity = 0.5
        if k == 2:
            m.alpha, m.refraction_index = 0.5, 1.3
        fs.materials.append(m); fs.materials.append(Scene._cache_of(m))
        mats.append((len(fs.materials) - 2, len(fs.materials) - 1))
    p = np.asarray([[-60, -1, 10], [60, -1, 10], [60, -1, -120], [-60, -1, -120]], np.float32)
    fs.meshes.append(MeshData(positions=p, indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32)))
    q = np.asarray([[-0.3, 0, -0.3], [0.3, 0, -0.3], [0.3, 0.6, 0.3], [-0.3, 0.6, 0.3]], np.float32)
    fs.meshes.append(MeshData(positions=q, indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32)))
    fs.items.append(Item(kind=1, id=1, material=mats[0][0], material_cache=mats[0][1], mesh=0, trans=eye4.copy(), trans_inv=eye4.copy(),
                         bbox_min=tuple(p.min(0)), bbox_max=tuple(p.max(0)), name="floor"))
    n = n_spheres + n_quads
    side = int(np.ceil(np.sqrt(n)))
    for i in range(n):
        gx, gz = i % side, i // side
        pos = (-40.0 + 80.0 * (gx + 0.5) / side + float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.6, 0.4)), -8.0 - 100.0 * (gz + 0.5) / side)
        mi, ci = mats[int(rng.integers(0, 6))]
        if i < n_spheres:
            r = float(rng.uniform(0.15, 0.35))
            t = get_transformation(eye4, pos, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
            fs.items.append(Item(kind=0, id=2 + i, material=mi, material_cache=ci, radius=r, trans=t, trans_inv=inverse_affine(t), bbox_min=(-r, -r, -r), bbox_max=(r, r, r)))
        else:
            t = get_transformation(eye4, pos, (1.0, 1.0, 1.0), (0.0, float(rng.uniform(0, 6.28)), 0.0))
            fs.items.append(Item(kind=1, id=2 + i, material=mi, material_cache=ci, mesh=1, trans=t, trans_inv=inverse_affine(t), bbox_min=tuple(q.min(0)), bbox_max=tuple(q.max(0))))
    fs.lights = [Light(pos=(5.0, 30.0, -20.0), intensity=900.0)]
    fs.meta = {"camera": dict(width=64, height=48, fov=float(np.float32(np.radians(55.0))), eye_pos=[0.0, 6.0, 6.0], up=[0.0, 1.0, 0.0], dir=[0.0, -0.25, -1.0],
                              clipping_near=0.1, clipping_far=300.0)}
    return fs


def test_more_items_than_the_top_level_used_to_hold(hip, oracle):
    """The reference's `items` is a Vec (src/scene.rs:69-83).  Up to round 3 item 4 097 made rr_scene_create fail (one item per leaf
    under a 12-level top tree); now the top level takes ceil(log2 n) levels out of the per-mesh trees' share of the traversal stack.
    5 000 items (13 levels) against the oracle, which walks its own item tree."""
    fs = _many_items_scene(3000, 1999)
    assert len(fs.items) == 5000
    _check(hip, oracle, fs, 96, 64, make_config(samples=2, monte_carlo=True, seed=4, max_recursion=3))


def test_item_count_beyond_rr_max_items_is_refused_with_a_message(hip):
    """RR_MAX_ITEMS = 2^20 (include/rustray_hip.h): one more is RR_ERR_UNSUPPORTED, not a crash and not a truncated scene."""
    import ctypes as C
    from rustray_amd.flat import rr_flat_scene, rr_item, rr_material
    n = (1 << 20) + 1
    items = (rr_item * n)()
    mats = (rr_material * 1)()
    for k in range(8):
        mats[0].texture[k] = -1
    eye = [1.0 if i % 5 == 0 else 0.0 for i in range(16)]
    proto = rr_item()
    proto.kind, proto.mesh, proto.radius, proto.visible = 0, -1, 1.0, 1
    for i in range(16):
        proto.trans[i] = proto.trans_inv[i] = eye[i]
    for i in range(3):
        proto.bbox_min[i], proto.bbox_max[i] = -1.0, 1.0
    C.memmove(items, bytes(proto) * n, C.sizeof(rr_item) * n)
    cs = rr_flat_scene()
    cs.abi_version, cs.n_items, cs.n_materials = 3, n, 1
    cs.items, cs.materials = C.cast(items, C.POINTER(rr_item)), C.cast(mats, C.POINTER(rr_material))
    h = C.c_void_p(None)
    rc = hip.lib().rr_scene_create(C.byref(cs), 0, C.byref(h))
    assert rc == -2 and not h and "RR_MAX_ITEMS" in hip.lib().rr_last_error().decode()
