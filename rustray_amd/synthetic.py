"""Synthetic stand-ins for the BASELINE configs whose assets are not available offline.

`scene/helmet.json`, `scene/sponza.json` and `scene/lotus.json` of the reference
download `.glb` files at load time (reference src/scene.rs:470-493,
src/helper.rs:22-33); `data/temp/` is empty in this pipeline and there is no
network.  The scenes below are assembled ONLY from geometry and
textures that ship inside the reference (scenes/assets.npz, made by
scenes/make_scenes.py) and are deterministic (numpy default_rng(1234)).  Every
result obtained on them is labelled "synthetic".

  sponza_syn   C4 stand-in: closed textured room (nearest filtering as sponza.json:15 asks),
               a grid of un-instanced meshes (> 50 items, so the top-level structure is used,
               reference src/raytracing.rs:23,434), reflection-only environment sphere,
               the reference's default point light (src/scene.rs:1386-1401).
  helmet_syn   C3 stand-in: one merged ~80k-triangle mesh with base + normal + roughness + AO maps,
               environment sphere, one point light of intensity 100 (helmet.json:29-38).
  lotus_syn    C5 stand-in: sponza_syn geometry with half of the objects made of glass
               (alpha 0.3, ior 1.5, reflectivity 0.5), the reflective floor of
               scene/floor_reflective.json (reflectivity 0.8, roughness 0.015) and depth of field
               (focal_length 20, aperture_size 16 — build-defined).
"""
from __future__ import annotations

import math
import os

import numpy as np

from .camera import Camera
from .flat import (FlatScene, Item, Light, Material, MeshData, RR_ITEM_MESH, RR_ITEM_SPHERE, RR_LIGHT_POINT)
from .scene import Scene, get_transformation, inverse_affine

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "assets.npz")
F32 = np.float32


def _assets() -> FlatScene:
    return FlatScene.load(_ASSETS)


def _add_material(fs: FlatScene, m: Material):
    fs.materials.append(m)
    fs.materials.append(Scene._cache_of(m))
    return len(fs.materials) - 2, len(fs.materials) - 1


def _mesh_item(fs: FlatScene, mesh_index: int, mat: Material, item_id: int, translation, rotation_deg, scale, name):
    trans = get_transformation(np.eye(4, dtype=F32), translation, scale,
                               tuple(float(F32(math.radians(r))) for r in rotation_deg))
    p = fs.meshes[mesh_index].positions
    mi, ci = _add_material(fs, mat)
    fs.items.append(Item(kind=RR_ITEM_MESH, id=item_id, material=mi, material_cache=ci, mesh=mesh_index,
                         trans=trans, trans_inv=inverse_affine(trans),
                         bbox_min=tuple(p.min(axis=0).tolist()), bbox_max=tuple(p.max(axis=0).tolist()), name=name))


def _plane(fs: FlatScene, pts, mat: Material, item_id: int, name: str):
    md = MeshData(positions=np.asarray(pts, F32), indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32),
                  uvs=np.asarray([[0, 0], [1, 0], [1, 1], [0, 1]], F32),
                  uv_indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32))
    fs.meshes.append(md)
    _mesh_item(fs, len(fs.meshes) - 1, mat, item_id, (0, 0, 0), (0, 0, 0), (1, 1, 1), name)


def _env_sphere(fs: FlatScene, tex_env: int, item_id: int):
    # scene/environment.json: radius 100, base black, ambient white * env map, reflection_only
    m = Material(base_color=(0.0, 0.0, 0.0), ambient_color=(1.0, 1.0, 1.0), reflection_only=True)
    m.texture[1] = tex_env
    mi, ci = _add_material(fs, m)
    fs.items.append(Item(kind=RR_ITEM_SPHERE, id=item_id, material=mi, material_cache=ci, radius=100.0,
                         bbox_min=(-100.0,) * 3, bbox_max=(100.0,) * 3, name="environment"))


def _unique_copy(md: MeshData, rng, amount=0.02) -> MeshData:
    """A per-object copy with a baked non-uniform stretch, so that no two objects share memory."""
    s = (1.0 + amount * (rng.random(3) * 2.0 - 1.0)).astype(F32)
    return MeshData(positions=(md.positions * s).astype(F32), indices=md.indices.copy(), uvs=md.uvs.copy(),
                    uv_indices=md.uv_indices.copy(), normals=md.normals.copy(), normal_indices=md.normal_indices.copy())


def _camera(eye, direction, fov_deg, width, height) -> Camera:
    c = Camera()
    c.eye_pos = np.asarray(eye, float)
    c.dir = np.asarray(direction, float)
    c.fov = float(F32(math.radians(fov_deg)))
    c.clipping_near, c.clipping_far = 0.1, 1000.0
    c.init(width, height)
    return c


def sponza_syn(grid: int = 10, glass: bool = False, seed: int = 1234) -> FlatScene:
    a = _assets()
    tex, mesh_id = a.meta["textures"], a.meta["meshes"]
    rng = np.random.default_rng(seed)
    fs = FlatScene()
    fs.name = "lotus_syn" if glass else "sponza_syn"
    fs.textures = list(a.textures)
    next_id = [0]

    def nid():
        next_id[0] += 3  # the reference's loader hands out three ids per object (src/scene.rs:299-300,:440,:541)
        return next_id[0]

    _env_sphere(fs, tex["env"], nid())
    # room: x in [-22, 22], y in [-2, 16], z in [-46, 8]
    X, Y0, Y1, Z0, Z1 = 22.0, -2.0, 16.0, -46.0, 8.0
    wall = Material(base_color=(0.9, 0.9, 0.9), specular_color=(0.2, 0.2, 0.2), texture_filtering_nearest=True)
    wall.texture[0], wall.texture[3] = tex["wall_base"], tex["wall_normal"]
    floor = Material(base_color=(0.8, 0.8, 0.8), specular_color=(0.3, 0.3, 0.3), texture_filtering_nearest=True)
    floor.texture[0] = tex["checker"]
    if glass:
        floor = Material(base_color=(0.2, 0.2, 0.2), specular_color=(0.16, 0.16, 0.16), reflectivity=0.8, roughness=0.015)
    import copy
    _plane(fs, [(-X, Y0, Z1), (X, Y0, Z1), (X, Y0, Z0), (-X, Y0, Z0)], floor, nid(), "floor")
    _plane(fs, [(-X, Y1, Z0), (X, Y1, Z0), (X, Y1, Z1), (-X, Y1, Z1)], copy.deepcopy(wall), nid(), "ceiling")
    _plane(fs, [(-X, Y0, Z0), (X, Y0, Z0), (X, Y1, Z0), (-X, Y1, Z0)], copy.deepcopy(wall), nid(), "back")
    _plane(fs, [(X, Y0, Z1), (-X, Y0, Z1), (-X, Y1, Z1), (X, Y1, Z1)], copy.deepcopy(wall), nid(), "front")
    _plane(fs, [(-X, Y0, Z1), (-X, Y0, Z0), (-X, Y1, Z0), (-X, Y1, Z1)], copy.deepcopy(wall), nid(), "left")
    _plane(fs, [(X, Y0, Z0), (X, Y0, Z1), (X, Y1, Z1), (X, Y1, Z0)], copy.deepcopy(wall), nid(), "right")
    # objects on a grid x grid raster over x in [-18, 18], z in [-42, -6]
    k = 0
    for gi in range(grid):
        for gj in range(grid):
            x = -18.0 + 36.0 * (gi + 0.5) / grid + float(rng.uniform(-0.6, 0.6))
            z = -42.0 + 36.0 * (gj + 0.5) / grid + float(rng.uniform(-0.6, 0.6))
            roty = float(rng.uniform(0.0, 360.0))
            kind = k % 8
            col = tuple(float(v) for v in rng.uniform(0.15, 0.9, 3))
            m = Material(base_color=col, specular_color=(0.5, 0.5, 0.5), shininess=float(rng.uniform(30.0, 300.0)),
                         ambient_color=tuple(c * 0.01 for c in col))
            u = float(rng.random())
            if glass:
                if k % 2 == 0:
                    m.alpha, m.refraction_index, m.reflectivity = 0.3, 1.5, 0.5
            else:
                if u < 0.15:
                    m.reflectivity = 0.4
                elif u < 0.23:
                    m.alpha, m.refraction_index = 0.4, 1.5
            if kind == 0:      # monkey, smooth shaded, leather maps
                md = _unique_copy(a.meshes[mesh_id["monkey"]], rng)
                fs.meshes.append(md)
                s = float(rng.uniform(1.3, 1.9))
                if not glass and u >= 0.23 and u < 0.6:
                    m.texture[0], m.texture[3] = tex["leather_base"], tex["leather_normal"]
                    m.base_color = (1.0, 1.0, 1.0)
                _mesh_item(fs, len(fs.meshes) - 1, m, nid(), (x, Y0 + 1.0 * s, z), (0.0, roty, 0.0), (s, s, s), f"monkey_{k}")
            else:
                big = kind == 1
                pa, pb = ("kbert_bevel_a", "kbert_bevel_b") if big else ("kbert_a", "kbert_b")
                s = float(rng.uniform(0.5, 0.8))
                m.smooth_shading = False
                for part, suffix in ((pa, "a"), (pb, "b")):
                    md = _unique_copy(a.meshes[mesh_id[part]], np.random.default_rng(seed + 7 * k))  # both parts share the stretch
                    fs.meshes.append(md)
                    mm = copy.deepcopy(m)
                    if suffix == "b" and not glass:
                        mm.texture[0] = tex["man"]
                    _mesh_item(fs, len(fs.meshes) - 1, mm, nid(), (x, Y0, z), (0.0, roty, 0.0), (s, s, s), f"kbert_{k}{suffix}")
            k += 1
    # the reference's default light (src/scene.rs:1386-1401)
    fs.lights.append(Light(pos=(-2.0, 10.0, 5.0), dir=(0.0, -1.0, 0.0), color=(1.0, 1.0, 1.0), intensity=200.0,
                           light_type=RR_LIGHT_POINT))
    cam = _camera((0.0, 5.0, 6.5), (0.0, -0.22, -1.0), 80.0, 1280, 720)
    fs.meta = {"camera": cam.state(), "synthetic": True,
               "config": dict(samples=512, monte_carlo=True, focal_length=20.0, aperture_size=16.0) if glass
               else dict(samples=128, monte_carlo=True)}
    return fs


def lotus_syn(grid: int = 10) -> FlatScene:
    return sponza_syn(grid=grid, glass=True)


def helmet_syn() -> FlatScene:
    a = _assets()
    tex, mesh_id = a.meta["textures"], a.meta["meshes"]
    fs = FlatScene()
    fs.name = "helmet_syn"
    fs.textures = list(a.textures)
    _env_sphere(fs, tex["env"], 3)
    # four copies of the bevelled figure merged into ONE mesh (~80k triangles), as glTF primitives are
    parts_p, parts_i, parts_uv, parts_n = [], [], [], []
    base = 0
    offsets = [(-1.6, 0.0, 0.0), (1.6, 0.0, 0.0), (0.0, 0.0, -1.8), (0.0, 0.0, 1.8)]
    for off in offsets:
        for part in ("kbert_bevel_a", "kbert_bevel_b"):
            md = a.meshes[mesh_id[part]]
            parts_p.append(md.positions + np.asarray(off, F32))
            parts_i.append(md.indices + base)
            parts_uv.append(md.uvs if len(md.uvs) == len(md.positions) else np.zeros((len(md.positions), 2), F32))
            parts_n.append(md.normals if len(md.normals) == len(md.positions) else np.tile(np.asarray([[0, 1, 0]], F32), (len(md.positions), 1)))
            base += len(md.positions)
    idx = np.concatenate(parts_i).astype(np.uint32)
    merged = MeshData(positions=np.concatenate(parts_p).astype(F32), indices=idx, uvs=np.concatenate(parts_uv).astype(F32),
                      uv_indices=idx.copy(), normals=np.concatenate(parts_n).astype(F32), normal_indices=idx.copy())
    fs.meshes.append(merged)
    m = Material(base_color=(1.0, 1.0, 1.0), specular_color=(0.5, 0.5, 0.5), reflectivity=0.25, roughness=0.02)
    m.texture[0], m.texture[3] = tex["leather_base"], tex["leather_normal"]
    m.texture[5], m.texture[6] = tex["wall_roughness"], tex["wall_ao"]
    _mesh_item(fs, 0, m, 6, (0.0, -2.0, -9.0), (0.0, 25.0, 0.0), (1.0, 1.0, 1.0), "helmet_syn")
    fs.lights.append(Light(pos=(3.0, 6.0, -2.0), dir=(0.0, -1.0, 0.0), color=(1.0, 1.0, 1.0), intensity=100.0,
                           light_type=RR_LIGHT_POINT))
    cam = _camera((0.0, 1.5, 0.0), (0.0, -0.2, -1.0), 60.0, 1280, 720)
    fs.meta = {"camera": cam.state(), "synthetic": True, "config": dict(samples=64, monte_carlo=True)}
    return fs
