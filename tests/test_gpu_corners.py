"""Adversarial scenes for order-dependent and rarely taken paths of the reference's trace loop, GPU vs oracle."""
import numpy as np
import pytest

from rustray_amd.flat import FlatScene, Item, Light, Material, MeshData, make_config
from rustray_amd.scene import Scene
from tests.helpers import camera_for, compare_frames, load_scene
from tests.test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu
EYE = np.eye(4, dtype=np.float32)


def _mat(fs, m):
    fs.materials.append(m); fs.materials.append(Scene._cache_of(m))
    return len(fs.materials) - 2, len(fs.materials) - 1


def _quad(y, half, uv=True):
    p = np.asarray([[-half, y, half], [half, y, half], [half, y, -half], [-half, y, -half]], np.float32)
    md = MeshData(positions=p, indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32))
    if uv:
        md.uvs = np.asarray([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
        md.uv_indices = np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32)
    return md


def _mesh_item(fs, mesh, mat, idn, name, bbox=None, trans=None, trans_inv=None):
    p = fs.meshes[mesh].positions
    mi, ci = _mat(fs, mat)
    lo, hi = (tuple(p.min(0)), tuple(p.max(0))) if bbox is None else bbox
    fs.items.append(Item(kind=1, id=idn, material=mi, material_cache=ci, mesh=mesh, trans=(EYE if trans is None else trans).copy(),
                         trans_inv=(EYE if trans_inv is None else trans_inv).copy(), bbox_min=lo, bbox_max=hi, name=name))


def _cam(fs, eye=(0.0, 6.0, 9.0), direction=(0.0, -0.6, -1.0), fov=60.0):
    fs.meta = {"camera": dict(width=64, height=64, fov=float(np.float32(np.radians(fov))), eye_pos=list(eye), up=[0.0, 1.0, 0.0],
                              dir=list(direction), clipping_near=0.1, clipping_far=100.0)}


def _check(hip, oracle, fs, w=96, h=96, **cfg):
    cam = camera_for(fs, w, h).c_struct()
    c = make_config(**({"samples": 2, "monte_carlo": True, "seed": 3} | cfg))
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, c)
    ref = oracle.render(fs.c_struct(), cam, c, n_threads=8)
    assert_parity(out, ref)
    return out, ref


def test_equal_toi_goes_to_the_smaller_bbox_distance_then_index(hip, oracle):
    """Two items share ONE mesh (bit-equal toi).  The stable sort by bbox distance + strict `<` of the reference
    (src/raytracing.rs:466-476) gives the hit to the item with the smaller bbox distance, index breaking ties."""
    fs = FlatScene()
    fs.meshes = [_quad(0.0, 5.0)]
    _mesh_item(fs, 0, Material(base_color=(1.0, 0.1, 0.1)), 3, "red")
    _mesh_item(fs, 0, Material(base_color=(0.1, 1.0, 0.1)), 6, "green")
    fs.lights = [Light(pos=(0.0, 8.0, 0.0), intensity=80.0)]
    _cam(fs)
    out, _ = _check(hip, oracle, fs)
    assert set(np.unique(out["object_id"])) == {0, 3}                       # same key: the lower index wins
    fs.items[1].bbox_min, fs.items[1].bbox_max = (-5.0, -1.0, -5.0), (5.0, 1.0, 5.0)   # a thicker declared box is entered earlier
    out, _ = _check(hip, oracle, fs)
    assert set(np.unique(out["object_id"])) == {0, 6}


def test_alpha_textured_occluder_and_short_uv_lists(hip, oracle):
    """Occluder alpha maps (src/raytracing.rs:894-913: receiver's alpha, receiver's get_uv, occluder's face id) and
    Mesh::get_uv's bounds test for faces without uv indices (src/shape/mesh.rs:116-120)."""
    fs = FlatScene()
    rng = np.random.default_rng(4)
    alpha = np.zeros((16, 16, 4), np.uint8); alpha[..., :3] = (rng.integers(0, 2, (16, 16, 1)) * 255).astype(np.uint8); alpha[..., 3] = 255
    base = np.full((8, 8, 4), 255, np.uint8); base[::2, ::2, :3] = 60
    fs.textures = [alpha, base]
    floor, cover = _quad(0.0, 10.0), _quad(3.0, 2.5)
    cover.uv_indices = cover.uv_indices[:1]                                  # second face has no uv indices -> uv (0,0)
    fs.meshes = [floor, cover]
    fm = Material(base_color=(0.9, 0.9, 0.9)); fm.texture[0] = 1
    _mesh_item(fs, 0, fm, 3, "floor")
    cm = Material(base_color=(0.2, 0.3, 0.9), alpha=0.7, refraction_index=1.2); cm.texture[4] = 0; cm.texture[0] = 1
    _mesh_item(fs, 1, cm, 6, "cover")
    fs.lights = [Light(pos=(1.0, 9.0, 2.0), intensity=90.0), Light(pos=(-3.0, 6.0, -1.0), color=(1.0, 0.6, 0.3), intensity=50.0)]
    _cam(fs)
    for nearest in (False, True):
        fs.materials[fs.items[1].material].texture_filtering_nearest = nearest
        _check(hip, oracle, fs, samples=3)


def test_camera_inside_solid_and_hollow_spheres(hip, oracle):
    """Ball::cast_local_ray from inside: a solid ball answers toi = 0, a non-solid one its far side."""
    for alpha, cull in ((1.0, True), (1.0, False), (0.6, True)):
        fs = FlatScene()
        m = Material(base_color=(0.7, 0.8, 0.9), alpha=alpha, backface_cullig=cull, reflectivity=0.2, refraction_index=1.3)
        mi, ci = _mat(fs, m)
        fs.items = [Item(kind=0, id=3, material=mi, material_cache=ci, radius=6.0, bbox_min=(-6.0,) * 3, bbox_max=(6.0,) * 3, name="shell")]
        m2 = Material(base_color=(0.9, 0.4, 0.1))
        mi2, ci2 = _mat(fs, m2)
        t = EYE.copy(); t[:3, 3] = (0.5, -0.5, -3.0); ti = EYE.copy(); ti[:3, 3] = (-0.5, 0.5, 3.0)
        fs.items.append(Item(kind=0, id=6, material=mi2, material_cache=ci2, radius=1.0, trans=t, trans_inv=ti, bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3, name="ball"))
        fs.lights = [Light(pos=(1.0, 2.0, 1.0), intensity=30.0)]
        _cam(fs, eye=(0.0, 0.0, 0.0), direction=(0.0, 0.0, -1.0), fov=80.0)
        _check(hip, oracle, fs, samples=2)


def test_projective_inverse_and_non_uniform_scale(hip, oracle):
    """get_inverse_ray divides the transformed origin by w (Point3::from_homogeneous, src/shape/mod.rs:757-760): a w row
    other than (0,0,0,1) must be honoured; normals use `trans`, not its inverse transpose (Appendix A 13)."""
    fs = FlatScene()
    fs.meshes = [_quad(0.0, 4.0)]
    _mesh_item(fs, 0, Material(base_color=(0.8, 0.8, 0.8), reflectivity=0.3), 3, "floor")
    s = np.diag(np.asarray([1.5, 0.6, 1.0, 1.0], np.float32)); s[:3, 3] = (0.0, 1.5, -1.0)
    si = np.linalg.inv(s.astype(np.float64)).astype(np.float32)
    si[3, :] = (0.0, 0.0, 0.0, 2.0)                                          # homogeneous scale: origin' = (M x) / 2
    si[:3, :] *= 2.0                                                        # ... compensated in the first three rows
    m = Material(base_color=(0.2, 0.7, 0.3), alpha=0.5, refraction_index=1.4, reflectivity=0.3)
    mi, ci = _mat(fs, m)
    fs.items.append(Item(kind=0, id=6, material=mi, material_cache=ci, radius=1.0, trans=s, trans_inv=si, bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3, name="ellipsoid"))
    fs.lights = [Light(pos=(2.0, 7.0, 3.0), intensity=70.0)]
    _cam(fs, eye=(0.0, 3.0, 6.0), direction=(0.0, -0.35, -1.0))
    _check(hip, oracle, fs, samples=2)


def test_deep_mesh_tree_keeps_within_the_traversal_stack(hip, oracle):
    """A geometric progression of nested triangles makes SAH peel one primitive per level, so the per-mesh tree reaches
    the builder's depth limit; the 4-wide collapse must then stay binary where the LDS stack budget is tight
    (rustray_amd/csrc/rr_bvh.cpp) and every triangle must remain reachable."""
    n = 600
    s = (4.0 * 0.97 ** np.arange(n)).astype(np.float32)
    z = (-0.004 * np.arange(n)).astype(np.float32)
    p = np.zeros((n, 3, 3), np.float32)
    p[:, 1, 0] = s; p[:, 2, 1] = s
    p[:, :, 2] = z[:, None]
    p[:, :, :2] -= 1.0
    fs = FlatScene()
    fs.meshes = [MeshData(positions=p.reshape(-1, 3), indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3)), _quad(-1.5, 6.0, uv=False)]
    _mesh_item(fs, 0, Material(base_color=(0.9, 0.6, 0.2), reflectivity=0.2), 2, "fan")
    _mesh_item(fs, 1, Material(base_color=(0.5, 0.5, 0.6)), 4, "floor")
    fs.lights = [Light(pos=(2.0, 3.0, 5.0), intensity=60.0)]
    _cam(fs, eye=(0.5, 0.8, 5.0), direction=(-0.1, -0.15, -1.0))
    out, _ = _check(hip, oracle, fs, w=128, h=128)
    assert (out["object_id"] == 2).sum() > 1000


@pytest.mark.parametrize("name", ["fuzz_alpha", "fuzz_234", "fuzz_568", "fuzz_6601"])
def test_non_finite_samples_reach_the_pixel_as_in_the_reference(hip, oracle, name):
    """Scenes reduced from tools/fuzz_parity.py mismatches (tests/golden/fuzz_*.npz).  The reference's f32 sums carry a
    NaN sample to the pixel (NaN.min(1.0) = 1.0 -> 255, src/raytracing.rs:406-417): fixed-point sums cannot, so k_shade /
    k_trace_shadow flag the pixel and k_resolve reproduces the value.
      fuzz_alpha: a sphere with flipped normals shadowed by a mesh with an alpha map.  The light term is exactly 0, but the
                  map is sampled at the RECEIVER's uv of the occluder's hit point (:905): acos(> 1) = NaN, 0 * NaN = NaN;
      fuzz_234:   a reflectivity map sampled at a NaN uv at a sphere's pole, no light: color * (1 - NaN) with color = 0;
      fuzz_568:   several such items, semi-transparent and invisible ones, refraction index below 1;
      fuzz_6601:  a normal map gives a sphere a NaN normal, the SHADOW ray starts at NaN: in the reference every candidate
                  sphere then reports Some(NaN), `in_light = toi > len` is false, and the occluder's alpha map sampled at a
                  NaN uv under the bilinear filter makes the attenuation NaN (rr_kernels.hip: trace_shadow_nonfinite)."""
    from rustray_amd.flat import FlatScene
    import os
    from tests.helpers import GOLDEN
    fs = FlatScene.load(os.path.join(GOLDEN, name + ".npz"))
    w, h = fs.meta["wh"]
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(**fs.meta["kw"])
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    assert int((ref["rgba"][..., :3] == 255).all(axis=-1).sum()) >= 1   # the white pixels are there


def test_zero_light_term_still_reaches_the_pixel_through_a_nan_uv_and_only_then(hip, oracle):
    """A light term that is exactly zero (light behind the receiver, no specular colour) is NaN in the reference when the occluder
    has an alpha map and the RECEIVER's get_uv of the occluder's hit point is not finite (src/raytracing.rs:898-912: 0 * NaN): the pixel
    turns white.  Receivers whose uv can be non-finite: every sphere, and a mesh with a zero-area face (0 / 0 area weights,
    src/shape/mesh.rs:127-143) -- the occluder's face id picks the receiver's face.  For every other receiver the zero term stays zero,
    and the device does not trace its shadow ray at all (D7): same frame, fewer shadow rays than the reference traces."""
    rng = np.random.default_rng(9)
    alpha = np.zeros((8, 8, 4), np.uint8); alpha[..., :3] = rng.integers(40, 255, (8, 8, 1)).astype(np.uint8); alpha[..., 3] = 255

    def scene(degenerate):
        fs = FlatScene()
        fs.textures = [alpha]
        floor = _quad(0.0, 6.0)
        # a third face: degenerate (three collinear vertices) or a proper sliver beside the quad
        third = [[7.0, 0.0, 0.0], [8.0, 0.0, 0.0], [9.0, 0.0, 0.0]] if degenerate else [[7.0, 0.0, 0.0], [8.0, 0.0, 0.0], [8.0, 0.0, -1.0]]
        floor.positions = np.concatenate([floor.positions, np.asarray(third, np.float32)])
        floor.indices = np.concatenate([floor.indices, np.asarray([[4, 5, 6]], np.uint32)])
        floor.uvs = np.concatenate([floor.uvs, np.asarray([[0.2, 0.2], [0.8, 0.3], [0.5, 0.9]], np.float32)])
        floor.uv_indices = np.concatenate([floor.uv_indices, np.asarray([[4, 5, 6]], np.uint32)])
        # the occluder BELOW the floor, three faces so that face id 2 exists: the receiver's face 2 is the third one
        p = np.asarray([[-4, -2, 4], [4, -2, 4], [4, -2, -4], [-4, -2, -4], [0, -2, 0]], np.float32)
        cover = MeshData(positions=p, indices=np.asarray([[0, 1, 4], [1, 2, 4], [2, 3, 0]], np.uint32),
                         uvs=np.asarray([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]], np.float32), uv_indices=np.asarray([[0, 1, 4], [1, 2, 4], [2, 3, 0]], np.uint32))
        fs.meshes = [floor, cover]
        fm = Material(base_color=(0.6, 0.6, 0.6), specular_color=(0.0, 0.0, 0.0), ambient_color=(0.2, 0.1, 0.05), cast_shadow=False)
        _mesh_item(fs, 0, fm, 3, "floor")
        cm = Material(base_color=(0.3, 0.3, 0.9)); cm.texture[4] = 0     # alpha map, bilinear (the default filter)
        _mesh_item(fs, 1, cm, 6, "cover")
        fs.lights = [Light(pos=(0.5, -9.0, -0.5), intensity=60.0)]        # BELOW the floor: dot(normal, to_light) < 0, the term is exactly zero
        _cam(fs)
        return fs

    white = {}
    for degenerate in (True, False):
        fs = scene(degenerate)
        cam = camera_for(fs, 96, 96).c_struct()
        cfg = make_config(samples=2, monte_carlo=False, seed=3)
        with hip.DeviceScene(fs, 0) as ds:
            out = ds.render(cam, cfg)
            st = ds.stats()
        ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True)
        assert_parity(out, ref)
        white[degenerate] = int((out["rgba"][..., :3] == 255).all(axis=-1).sum())
        if degenerate:
            assert st["shadow_rays"] > 0                                   # traced although every term is zero: the NaN has to be found
        else:
            assert st["shadow_rays"] == 0 and ref["counters"]["rays_shadow"] > 0   # (D7) nothing to find: not traced
    assert white[True] > 50 and white[False] == 0, white


def _scaled_world(fs, S):
    """The same scene with the world S times larger (item transforms, lights, camera): hit distances scale by S, colours stay
    (point-light falloff is I / (4 pi d): the intensity scales with S)."""
    import copy
    from rustray_amd.scene import inverse_affine
    out = copy.deepcopy(fs)
    sc = np.diag([S, S, S, 1.0]).astype(np.float32)
    for it in out.items:
        it.trans = (sc @ np.asarray(it.trans, np.float32)).astype(np.float32)
        it.trans_inv = inverse_affine(it.trans)
    for l in out.lights:
        l.pos = tuple(float(v) * S for v in l.pos)
        if l.light_type != 0:
            l.intensity = float(l.intensity) * S
    cam = dict(out.meta["camera"])
    cam["eye_pos"] = [float(v) * S for v in cam["eye_pos"]]
    cam["clipping_far"] = float(cam.get("clipping_far", 100.0)) * S
    out.meta = dict(out.meta); out.meta["camera"] = cam
    return out


def test_depth_of_far_hits_is_merged_per_pixel(hip, oracle):
    """a root hit farther than 512 units does not fit the 32-bit lane sum of the depth accumulator (depth * 2^16 >= 2^25)
    and was added as one unmerged 64-bit atomic per lane -- 64 lanes of a level-1 packet to one word.  Such terms are now merged per
    pixel in 64 bits (accum_depth_wide_merged).  spheres_room scaled 200 x (hit distances 600 .. 4000 units) against the oracle, and
    timed against the unscaled scene: the far form must not cost more than a fraction of the frame."""
    near = load_scene("spheres_room")
    far = _scaled_world(near, 200.0)
    cfg = make_config(samples=64, monte_carlo=True, seed=6, max_recursion=2)
    cam_f = camera_for(far, 160, 96).c_struct()
    with hip.DeviceScene(far, 0) as ds:
        out = ds.render(cam_f, cfg)
        assert float(np.nanmax(out["depth"])) > 512.0
        ds.set_profiling(True)
        ds.render(cam_f, cfg); t_far = ds.stats()["ms_shade"]
    ref = oracle.render(far.c_struct(), cam_f, make_config(samples=64, monte_carlo=True, seed=6, max_recursion=2), n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["max_depth_rel"] < 1e-4, res
    with hip.DeviceScene(near, 0) as ds:
        ds.set_profiling(True)
        cam_n = camera_for(near, 160, 96).c_struct()
        ds.render(cam_n, cfg); ds.render(cam_n, cfg); t_near = ds.stats()["ms_shade"]
    assert t_far < 1.5 * t_near + 0.2, (t_far, t_near)
