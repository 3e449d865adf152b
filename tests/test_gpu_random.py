"""Randomised parity: seeded random scenes (spheres, triangle soups, mixed materials, several light kinds, random
cameras and configs) rendered by the HIP path and by the oracle must agree within +-1 LSB and build the same path tree."""
import numpy as np
import pytest

from rustray_amd.flat import FlatScene, Item, Light, Material, MeshData, make_config
from rustray_amd.scene import Scene, get_transformation, inverse_affine
from tests.helpers import camera_for
from tests.test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu


def _random_scene(seed: int) -> FlatScene:
    rng = np.random.default_rng(seed)
    fs = FlatScene()
    fs.name = f"random{seed}"
    # one shared checker texture (nearest or bilinear per material)
    tex = np.zeros((16, 16, 4), np.uint8)
    tex[..., :3] = rng.integers(40, 255, size=(16, 16, 3))
    tex[..., 3] = np.where(rng.random((16, 16)) < 0.2, 128, 255)
    fs.textures = [tex]

    def material():
        m = Material(base_color=tuple(rng.uniform(0.1, 1.0, 3)), specular_color=tuple(rng.uniform(0.0, 0.9, 3)),
                     ambient_color=tuple(rng.uniform(0.0, 0.05, 3)), shininess=float(rng.uniform(5.0, 200.0)))
        kind = rng.integers(0, 5)
        if kind == 1:
            m.reflectivity = float(rng.uniform(0.2, 0.9)); m.roughness = float(rng.choice([0.0, 0.02]))
        elif kind == 2:
            m.alpha = float(rng.uniform(0.1, 0.7)); m.refraction_index = float(rng.uniform(1.0, 1.6)); m.reflectivity = float(rng.uniform(0.0, 0.4))
        elif kind == 3:
            m.texture = [0, -1, -1, -1, -1, -1, -1, -1]; m.texture_filtering_nearest = bool(rng.integers(0, 2))
        m.smooth_shading = bool(rng.integers(0, 2))
        m.shadow_softness = float(rng.choice([0.0, 0.01, 0.05]))
        m.backface_cullig = bool(rng.integers(0, 2))
        m.cast_shadow = bool(rng.random() < 0.9); m.receive_shadow = bool(rng.random() < 0.9)
        return m

    def add(item, m):
        fs.materials.append(m); fs.materials.append(Scene._cache_of(m))
        item.material, item.material_cache = len(fs.materials) - 2, len(fs.materials) - 1
        fs.items.append(item)

    # floor
    p = np.asarray([[-12, -2, 12], [12, -2, 12], [12, -2, -12], [-12, -2, -12]], np.float32)
    fs.meshes.append(MeshData(positions=p, indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32),
                              uvs=np.asarray([[0, 0], [4, 0], [4, 4], [0, 4]], np.float32), uv_indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32)))
    eye4 = np.eye(4, dtype=np.float32)
    add(Item(kind=1, id=1, material=0, material_cache=0, mesh=0, trans=eye4.copy(), trans_inv=eye4.copy(), bbox_min=tuple(p.min(0)), bbox_max=tuple(p.max(0)), name="floor"), material())
    nid = 2
    for _ in range(int(rng.integers(2, 6))):           # spheres with random affine transforms
        r = float(rng.uniform(0.4, 1.3))
        t = get_transformation(eye4, tuple(rng.uniform(-4, 4, 3) * (1, 0.4, 1) + (0, 0, -7)), tuple(rng.uniform(0.7, 1.5, 3)), tuple(rng.uniform(-1, 1, 3)))
        add(Item(kind=0, id=nid, material=0, material_cache=0, radius=r, trans=t, trans_inv=inverse_affine(t), bbox_min=(-r, -r, -r), bbox_max=(r, r, r), name=f"s{nid}"), material())
        nid += 1
    for _ in range(int(rng.integers(1, 4))):           # triangle soups, instanced with a transform
        n = int(rng.integers(8, 120))
        c = rng.uniform(-1.0, 1.0, (n, 1, 3)).astype(np.float32)
        tri = (c + rng.uniform(-0.45, 0.45, (n, 3, 3))).astype(np.float32)
        md = MeshData(positions=tri.reshape(-1, 3), indices=np.arange(3 * n, dtype=np.uint32).reshape(n, 3))
        if rng.random() < 0.6:
            md.uvs = rng.uniform(-1.5, 2.5, (3 * n, 2)).astype(np.float32); md.uv_indices = md.indices.copy()
        if rng.random() < 0.6:
            nn = rng.normal(size=(3 * n, 3)); md.normals = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32); md.normal_indices = md.indices.copy()
        fs.meshes.append(md)
        t = get_transformation(eye4, tuple(rng.uniform(-3, 3, 3) * (1, 0.3, 1) + (0, 0.5, -7)), tuple(rng.uniform(0.8, 1.6, 3)), tuple(rng.uniform(-2, 2, 3)))
        pp = md.positions
        add(Item(kind=1, id=nid, material=0, material_cache=0, mesh=len(fs.meshes) - 1, trans=t, trans_inv=inverse_affine(t), bbox_min=tuple(pp.min(0)), bbox_max=tuple(pp.max(0)),
                 name=f"m{nid}", flip_normals=bool(rng.random() < 0.2)), material())
        nid += 1
    fs.lights = [Light(pos=tuple(rng.uniform(-5, 5, 3) * (1, 0, 1) + (0, 7, -3)), intensity=float(rng.uniform(80, 250)))]
    if rng.random() < 0.5:
        fs.lights.append(Light(dir=tuple(rng.uniform(-0.5, 0.5, 3) + (0, -1, 0)), intensity=float(rng.uniform(0.2, 0.6)), light_type=0, color=(1.0, 0.9, 0.8)))
    if rng.random() < 0.4:
        fs.lights.append(Light(pos=(0.0, 6.0, 2.0), dir=(0.0, -1.0, -1.2), intensity=120.0, light_type=2, max_angle=float(rng.uniform(0.3, 0.8))))
    fs.meta = {"camera": dict(width=64, height=64, fov=float(np.float32(np.radians(rng.uniform(40, 80)))), eye_pos=[float(np.float32(v)) for v in rng.uniform(-1, 1, 3) + (0, 1.0, 3)],
                              up=[0.0, 1.0, 0.0], dir=[float(np.float32(v)) for v in rng.uniform(-0.2, 0.2, 3) + (0, -0.15, -1)], clipping_near=0.1, clipping_far=100.0)}
    return fs


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_oracle(hip, oracle, seed):
    fs = _random_scene(1000 + seed)
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(40, 90)), int(rng.integers(30, 70))
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=int(rng.choice([1, 2, 4, 6, 16])), monte_carlo=bool(seed % 3), seed=seed, max_recursion=int(rng.choice([2, 4, 6])),
                      fog_density=float(rng.choice([0.0, 0.02])), gamma_correction=bool(seed % 5 == 0),
                      aperture_size=float(rng.choice([1.0, 1.0, 8.0])), focal_length=float(rng.choice([1.0, 6.0])))
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=14, want_counters=True)
    assert_parity(out, ref, fs.name)
    c = ref["counters"]
    assert st["primary_rays"] == c["rays_primary"] and st["secondary_rays"] == c["rays_secondary"] and st["shaded_hits"] == c["shaded_hits"]
