"""Flat scene: the POD form of rustray's `Scene` that crosses the C ABI.

ctypes mirrors of every struct in include/rustray_hip.h plus `FlatScene`, a
numpy-backed container that (a) builds the `rr_flat_scene` view handed to
`rr_scene_create` and (b) round-trips through a compressed .npz file, which is
also the on-disk fixture format of this repository.

Reference types flattened here: `Scene` (reference src/scene.rs:69-83),
`ShapeBasics` (src/shape/mod.rs:661-680), `Material` (src/shape/mod.rs:95-134),
`Mesh` (src/shape/mesh.rs:10-21), `Sphere` (src/shape/sphere.rs:10-15),
`Light` (src/scene.rs:40-51).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

RR_ABI_VERSION = 3
RR_TEX_COUNT = 8
TEX_NAMES = ["base", "ambient", "specular", "normal", "alpha", "roughness",
             "ambient_occlusion", "reflectivity"]  # TextureType order, src/shape/mod.rs:633-643
RR_ITEM_SPHERE, RR_ITEM_MESH = 0, 1
RR_LIGHT_DIRECTIONAL, RR_LIGHT_POINT, RR_LIGHT_SPOT = 0, 1, 2


class rr_texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p)]


class rr_material(C.Structure):
    _fields_ = [
        ("ambient_color", C.c_float * 3), ("base_color", C.c_float * 3), ("specular_color", C.c_float * 3),
        ("alpha", C.c_float), ("shininess", C.c_float), ("reflectivity", C.c_float),
        ("refraction_index", C.c_float), ("normal_map_strength", C.c_float),
        ("shadow_softness", C.c_float), ("roughness", C.c_float),
        ("texture", C.c_int32 * RR_TEX_COUNT),
        ("texture_filtering_nearest", C.c_uint8), ("cast_shadow", C.c_uint8), ("receive_shadow", C.c_uint8),
        ("monte_carlo", C.c_uint8), ("smooth_shading", C.c_uint8), ("reflection_only", C.c_uint8),
        ("backface_cullig", C.c_uint8), ("_pad", C.c_uint8),
    ]


class rr_mesh(C.Structure):
    _fields_ = [
        ("positions", C.c_void_p), ("indices", C.c_void_p), ("uvs", C.c_void_p), ("uv_indices", C.c_void_p),
        ("normals", C.c_void_p), ("normal_indices", C.c_void_p),
        ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("n_uvs", C.c_uint32),
        ("n_uv_faces", C.c_uint32), ("n_normals", C.c_uint32), ("n_normal_faces", C.c_uint32),
    ]


class rr_item(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("id", C.c_uint32), ("material", C.c_int32), ("material_cache", C.c_int32),
        ("mesh", C.c_int32), ("radius", C.c_float),
        ("trans", C.c_float * 16), ("trans_inv", C.c_float * 16),
        ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
        ("visible", C.c_uint8), ("flip_normals", C.c_uint8), ("_pad", C.c_uint8 * 2),
    ]


class rr_light(C.Structure):
    _fields_ = [
        ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("color", C.c_float * 3),
        ("intensity", C.c_float), ("max_angle", C.c_float), ("light_type", C.c_uint32),
        ("enabled", C.c_uint8), ("_pad", C.c_uint8 * 3),
    ]


class rr_flat_scene(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_items", C.c_uint32), ("n_meshes", C.c_uint32),
        ("n_materials", C.c_uint32), ("n_textures", C.c_uint32), ("n_lights", C.c_uint32),
        ("items", C.POINTER(rr_item)), ("meshes", C.POINTER(rr_mesh)),
        ("materials", C.POINTER(rr_material)), ("textures", C.POINTER(rr_texture)),
        ("lights", C.POINTER(rr_light)),
    ]


class rr_camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("projection_inverse", C.c_float * 16), ("view_inverse", C.c_float * 16)]


class rr_config(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("focal_length", C.c_float), ("aperture_size", C.c_float),
        ("fog_density", C.c_float), ("fog_color", C.c_float * 3),
        ("samples", C.c_uint16), ("max_recursion", C.c_uint16),
        ("monte_carlo", C.c_uint8), ("gamma_correction", C.c_uint8), ("_pad", C.c_uint8 * 2),
    ]


class rr_frame(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("normal", C.c_void_p), ("depth", C.c_void_p), ("object_id", C.c_void_p)]


class rr_region(C.Structure):
    _fields_ = [("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("n_ranks", C.c_uint32), ("rank", C.c_uint32)]


class rr_pick_result(C.Structure):
    _fields_ = [("hit", C.c_uint32), ("object_id", C.c_uint32), ("item_index", C.c_uint32), ("distance", C.c_float)]


class rr_frame_stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("shaded_hits", C.c_uint64), ("ms_total", C.c_double), ("ms_trace_closest", C.c_double),
        ("ms_trace_shadow", C.c_double), ("ms_shade", C.c_double),
        ("launches_trace_closest", C.c_uint64), ("launches_trace_shadow", C.c_uint64),
        ("launches_shade", C.c_uint64), ("batches", C.c_uint64), ("sliced_levels", C.c_uint64),
        ("binned_rays", C.c_uint64), ("ms_binning", C.c_double),
        ("ms_trace_closest_level1", C.c_double), ("launches_trace_closest_level1", C.c_uint64),
        ("multi_devices", C.c_uint32), ("multi_peer_links", C.c_uint32), ("multi_staged_links", C.c_uint32), ("_pad", C.c_uint32),
        ("ms_multi_exchange", C.c_double),
        ("ms_shade_level1", C.c_double), ("launches_shade_level1", C.c_uint64),
        ("ms_trace_shadow_level1", C.c_double), ("launches_trace_shadow_level1", C.c_uint64),
    ]


class rr_tuning(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("sample_group", C.c_uint32), ("queue_budget_bytes", C.c_uint64),
                ("shade_chunk_rays", C.c_uint64), ("kernel_timing", C.c_uint32), ("multi_force_staged", C.c_uint32), ("bin_min_rays", C.c_uint64)]


# ---------------------------------------------------------------------------
# numpy-side description
# ---------------------------------------------------------------------------
@dataclass
class Material:
    """Defaults of Material::new (reference src/shape/mod.rs:138-180)."""
    ambient_color: tuple = (0.0, 0.0, 0.0)
    base_color: tuple = (1.0, 1.0, 1.0)
    specular_color: tuple = (0.8, 0.8, 0.8)
    alpha: float = 1.0
    shininess: float = 150.0
    reflectivity: float = 0.0
    refraction_index: float = 1.0
    normal_map_strength: float = 1.0
    shadow_softness: float = 0.01
    roughness: float = 0.0
    texture: list = field(default_factory=lambda: [-1] * RR_TEX_COUNT)
    texture_filtering_nearest: bool = False
    cast_shadow: bool = True
    receive_shadow: bool = True
    monte_carlo: bool = True
    smooth_shading: bool = True
    reflection_only: bool = False
    backface_cullig: bool = True

    _FLOATS = ["alpha", "shininess", "reflectivity", "refraction_index", "normal_map_strength",
               "shadow_softness", "roughness"]
    _BOOLS = ["texture_filtering_nearest", "cast_shadow", "receive_shadow", "monte_carlo",
              "smooth_shading", "reflection_only", "backface_cullig"]

    def c_struct(self) -> "rr_material":
        c = rr_material()
        c.ambient_color[:] = [np.float32(v) for v in self.ambient_color]
        c.base_color[:] = [np.float32(v) for v in self.base_color]
        c.specular_color[:] = [np.float32(v) for v in self.specular_color]
        for f in Material._FLOATS:
            setattr(c, f, np.float32(getattr(self, f)))
        c.texture[:] = self.texture
        for b in Material._BOOLS:
            setattr(c, b, 1 if getattr(self, b) else 0)
        return c

    def to_row(self) -> np.ndarray:
        r = list(self.ambient_color) + list(self.base_color) + list(self.specular_color)
        r += [getattr(self, f) for f in self._FLOATS]
        r += [float(t) for t in self.texture]
        r += [1.0 if getattr(self, b) else 0.0 for b in self._BOOLS]
        return np.asarray(r, dtype=np.float64)

    @classmethod
    def from_row(cls, r) -> "Material":
        m = cls()
        m.ambient_color, m.base_color, m.specular_color = tuple(r[0:3]), tuple(r[3:6]), tuple(r[6:9])
        k = 9
        for f in cls._FLOATS:
            setattr(m, f, float(r[k])); k += 1
        m.texture = [int(v) for v in r[k:k + RR_TEX_COUNT]]; k += RR_TEX_COUNT
        for b in cls._BOOLS:
            setattr(m, b, bool(r[k] != 0.0)); k += 1
        return m


@dataclass
class MeshData:
    positions: np.ndarray  # (V,3) f32
    indices: np.ndarray    # (T,3) u32
    uvs: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    uv_indices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint32))
    normals: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    normal_indices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint32))


@dataclass
class Item:
    kind: int
    id: int
    material: int
    material_cache: int
    mesh: int = -1
    radius: float = 0.0
    trans: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))      # math layout [row, col]
    trans_inv: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    bbox_min: tuple = (0.0, 0.0, 0.0)
    bbox_max: tuple = (0.0, 0.0, 0.0)
    visible: bool = True
    flip_normals: bool = False
    name: str = ""


@dataclass
class Light:
    pos: tuple = (0.0, 0.0, 0.0)
    dir: tuple = (0.0, -1.0, 0.0)
    color: tuple = (1.0, 1.0, 1.0)
    intensity: float = 1.0
    max_angle: float = float(np.float32(np.pi) / np.float32(2.0))
    light_type: int = RR_LIGHT_POINT
    enabled: bool = True


def _colmajor(m: np.ndarray):
    """4x4 in math layout -> 16 floats column-major (nalgebra storage)."""
    return np.asarray(m, dtype=np.float32).T.reshape(16)


class FlatScene:
    """numpy-backed flat scene; `c_struct()` gives the rr_flat_scene view."""

    def __init__(self):
        self.items: List[Item] = []
        self.meshes: List[MeshData] = []
        self.materials: List[Material] = []
        self.textures: List[np.ndarray] = []  # (H,W,4) uint8
        self.lights: List[Light] = []
        self.name = ""
        self.meta: dict = {}  # JSON-serialisable extras: camera state, scene-file config overrides
        self._keep = None

    # -- ABI view -----------------------------------------------------------
    def c_struct(self) -> rr_flat_scene:
        keep = []
        texs = (rr_texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            assert t.ndim == 3 and t.shape[2] == 4
            keep.append(t)
            texs[i].width, texs[i].height = t.shape[1], t.shape[0]
            texs[i].rgba8 = t.ctypes.data
        mats = (rr_material * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            mats[i] = m.c_struct()
        meshes = (rr_mesh * max(1, len(self.meshes)))()
        for i, md in enumerate(self.meshes):
            c = meshes[i]

            def arr(a, dt, cols):
                a = np.ascontiguousarray(a, dtype=dt).reshape(-1, cols)
                keep.append(a)
                return a
            p = arr(md.positions, np.float32, 3); ix = arr(md.indices, np.uint32, 3)
            uv = arr(md.uvs, np.float32, 2); uvi = arr(md.uv_indices, np.uint32, 3)
            n = arr(md.normals, np.float32, 3); ni = arr(md.normal_indices, np.uint32, 3)
            c.positions, c.indices = p.ctypes.data, ix.ctypes.data
            c.uvs = uv.ctypes.data if len(uv) else None
            c.uv_indices = uvi.ctypes.data if len(uvi) else None
            c.normals = n.ctypes.data if len(n) else None
            c.normal_indices = ni.ctypes.data if len(ni) else None
            c.n_vertices, c.n_triangles = len(p), len(ix)
            c.n_uvs, c.n_uv_faces, c.n_normals, c.n_normal_faces = len(uv), len(uvi), len(n), len(ni)
        items = (rr_item * max(1, len(self.items)))()
        for i, it in enumerate(self.items):
            c = items[i]
            c.kind, c.id, c.material, c.material_cache, c.mesh = it.kind, it.id, it.material, it.material_cache, it.mesh
            c.radius = np.float32(it.radius)
            c.trans[:] = _colmajor(it.trans).tolist()
            c.trans_inv[:] = _colmajor(it.trans_inv).tolist()
            c.bbox_min[:] = [np.float32(v) for v in it.bbox_min]
            c.bbox_max[:] = [np.float32(v) for v in it.bbox_max]
            c.visible, c.flip_normals = int(it.visible), int(it.flip_normals)
        lights = (rr_light * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            c = lights[i]
            c.pos[:] = [np.float32(v) for v in l.pos]
            c.dir[:] = [np.float32(v) for v in l.dir]
            c.color[:] = [np.float32(v) for v in l.color]
            c.intensity, c.max_angle = np.float32(l.intensity), np.float32(l.max_angle)
            c.light_type, c.enabled = l.light_type, int(l.enabled)
        fs = rr_flat_scene()
        fs.abi_version = RR_ABI_VERSION
        fs.n_items, fs.n_meshes, fs.n_materials = len(self.items), len(self.meshes), len(self.materials)
        fs.n_textures, fs.n_lights = len(self.textures), len(self.lights)
        fs.items = C.cast(items, C.POINTER(rr_item)); fs.meshes = C.cast(meshes, C.POINTER(rr_mesh))
        fs.materials = C.cast(mats, C.POINTER(rr_material)); fs.textures = C.cast(texs, C.POINTER(rr_texture))
        fs.lights = C.cast(lights, C.POINTER(rr_light))
        keep += [texs, mats, meshes, items, lights]
        self._keep = keep  # the struct borrows these buffers
        return fs

    # -- statistics ----------------------------------------------------------
    def n_triangles_instanced(self) -> int:
        return sum(len(self.meshes[it.mesh].indices) for it in self.items if it.kind == RR_ITEM_MESH)

    # -- npz round trip --------------------------------------------------------
    def save(self, path: str) -> None:
        import json
        d = {"name": np.asarray(self.name), "meta": np.asarray(json.dumps(self.meta))}
        d["n"] = np.asarray([len(self.items), len(self.meshes), len(self.materials), len(self.textures), len(self.lights)])
        for i, t in enumerate(self.textures):
            t = np.asarray(t, dtype=np.uint8)
            # store RGB only when alpha is all 255 (smaller files)
            d[f"tex{i}"] = t[:, :, :3] if bool((t[:, :, 3] == 255).all()) else t
        for i, md in enumerate(self.meshes):
            d[f"mesh{i}_p"] = np.asarray(md.positions, np.float32); d[f"mesh{i}_i"] = np.asarray(md.indices, np.uint32)
            d[f"mesh{i}_uv"] = np.asarray(md.uvs, np.float32); d[f"mesh{i}_uvi"] = np.asarray(md.uv_indices, np.uint32)
            d[f"mesh{i}_n"] = np.asarray(md.normals, np.float32); d[f"mesh{i}_ni"] = np.asarray(md.normal_indices, np.uint32)
        d["materials"] = np.stack([m.to_row() for m in self.materials]) if self.materials else np.zeros((0, 31))
        it_rows, it_mats, it_names = [], [], []
        for it in self.items:
            it_rows.append([it.kind, it.id, it.material, it.material_cache, it.mesh, int(it.visible), int(it.flip_normals)])
            it_mats.append(np.concatenate([_colmajor(it.trans), _colmajor(it.trans_inv),
                                           np.asarray(it.bbox_min, np.float32), np.asarray(it.bbox_max, np.float32),
                                           np.asarray([it.radius], np.float32)]))
            it_names.append(it.name)
        d["items_i"] = np.asarray(it_rows, dtype=np.int64).reshape(-1, 7)
        d["items_f"] = np.asarray(it_mats, dtype=np.float32).reshape(-1, 39)
        d["items_name"] = np.asarray(it_names, dtype=str)
        d["lights"] = np.asarray([[*l.pos, *l.dir, *l.color, l.intensity, l.max_angle, l.light_type, int(l.enabled)]
                                  for l in self.lights], dtype=np.float64).reshape(-1, 13)
        np.savez_compressed(path, **d)

    @classmethod
    def load(cls, path: str) -> "FlatScene":
        z = np.load(path, allow_pickle=False)
        s = cls()
        import json
        s.name = str(z["name"])
        s.meta = json.loads(str(z["meta"])) if "meta" in z.files else {}
        ni, nm, nmat, nt, nl = [int(v) for v in z["n"]]
        for i in range(nt):
            t = z[f"tex{i}"]
            if t.shape[2] == 3:
                t = np.concatenate([t, np.full(t.shape[:2] + (1,), 255, np.uint8)], axis=2)
            s.textures.append(np.ascontiguousarray(t))
        for i in range(nm):
            s.meshes.append(MeshData(z[f"mesh{i}_p"], z[f"mesh{i}_i"], z[f"mesh{i}_uv"], z[f"mesh{i}_uvi"],
                                     z[f"mesh{i}_n"], z[f"mesh{i}_ni"]))
        for r in z["materials"]:
            s.materials.append(Material.from_row(r))
        names = z["items_name"]
        for k in range(ni):
            ii, ff = z["items_i"][k], z["items_f"][k]
            s.items.append(Item(kind=int(ii[0]), id=int(ii[1]), material=int(ii[2]), material_cache=int(ii[3]),
                                mesh=int(ii[4]), visible=bool(ii[5]), flip_normals=bool(ii[6]),
                                trans=ff[0:16].reshape(4, 4).T.copy(), trans_inv=ff[16:32].reshape(4, 4).T.copy(),
                                bbox_min=tuple(ff[32:35]), bbox_max=tuple(ff[35:38]), radius=float(ff[38]),
                                name=str(names[k])))
        for r in z["lights"]:
            s.lights.append(Light(pos=tuple(r[0:3]), dir=tuple(r[3:6]), color=tuple(r[6:9]), intensity=float(r[9]),
                                  max_angle=float(r[10]), light_type=int(r[11]), enabled=bool(r[12])))
        return s


def make_config(samples=1, monte_carlo=False, seed=0, max_recursion=6, focal_length=1.0, aperture_size=1.0,
                fog_density=0.0, fog_color=(0.4, 0.4, 0.4), gamma_correction=False) -> rr_config:
    """RaytracingConfig::new defaults (reference src/raytracing.rs:110-127)."""
    c = rr_config()
    c.seed = seed
    c.focal_length, c.aperture_size, c.fog_density = focal_length, aperture_size, fog_density
    c.fog_color[:] = list(fog_color)
    c.samples, c.max_recursion = samples, max_recursion
    c.monte_carlo, c.gamma_correction = int(monte_carlo), int(gamma_correction)
    return c
