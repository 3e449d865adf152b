"""glTF 2.0 / GLB reader producing what the reference's `Scene::load_gltf` consumes.

The reference loads glTF through the `easy-gltf` crate (reference src/scene.rs:722-978), which
flattens the node hierarchy: vertices come back in WORLD space (node transforms applied), one
"model" per mesh primitive, triangles de-indexed, materials with the metallic-roughness texture
split into a metallic (B) and a roughness (G) image, lights and cameras with their node
transform applied.  This module restates that behaviour [recalled: easy-gltf 1.1 is not vendored
in the reference tree] on top of a plain GLB/JSON parser; images are decoded with PIL.

Only what `load_gltf` reads is produced: positions, normals, uv set 0, the material fields and
maps it maps (src/scene.rs:909-962, :980-1124), KHR_lights_punctual lights, the first camera.
"""
from __future__ import annotations

import base64
import io
import json
import math
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


@dataclass
class GltfMaterial:
    name: str = "default"
    base_color_factor: tuple = (1.0, 1.0, 1.0, 1.0)
    metallic_factor: float = 0.0   # easy-gltf's Default for a primitive without material [recalled]
    roughness_factor: float = 0.0
    emissive_factor: tuple = (0.0, 0.0, 0.0)
    occlusion_factor: float = 1.0
    base_color_texture: Optional[np.ndarray] = None   # (H,W,4) u8
    normal_texture: Optional[np.ndarray] = None       # (H,W,3)
    metallic_texture: Optional[np.ndarray] = None     # (H,W) = B channel of the MR texture
    roughness_texture: Optional[np.ndarray] = None    # (H,W) = G channel
    occlusion_texture: Optional[np.ndarray] = None    # (H,W) = R channel
    emissive_texture: Optional[np.ndarray] = None     # (H,W,3)
    index: int = -1


@dataclass
class GltfModel:
    name: str
    material: GltfMaterial
    positions: np.ndarray                 # (T*3, 3) world space, de-indexed
    normals: Optional[np.ndarray] = None  # (T*3, 3)
    tex_coords: Optional[np.ndarray] = None  # (T*3, 2)


@dataclass
class GltfLight:
    kind: str
    position: tuple = (0.0, 0.0, 0.0)
    direction: tuple = (0.0, 0.0, -1.0)
    color: tuple = (1.0, 1.0, 1.0)
    intensity: float = 1.0
    outer_cone_angle: float = math.pi / 4.0
    name: Optional[str] = None


@dataclass
class GltfCamera:
    transform: np.ndarray
    yfov: float
    znear: float
    zfar: float


@dataclass
class GltfScene:
    models: List[GltfModel] = field(default_factory=list)
    lights: List[GltfLight] = field(default_factory=list)
    cameras: List[GltfCamera] = field(default_factory=list)


class _Doc:
    def __init__(self, path: str):
        self.dir = os.path.dirname(path)
        raw = open(path, "rb").read()
        self.bin: Optional[bytes] = None
        if raw[:4] == b"glTF":
            _, _, total = struct.unpack("<4sII", raw[:12])
            off = 12
            self.json = None
            while off < total:
                clen, ctype = struct.unpack("<I4s", raw[off:off + 8])
                chunk = raw[off + 8:off + 8 + clen]
                if ctype == b"JSON":
                    self.json = json.loads(chunk.decode("utf-8"))
                elif ctype[:3] == b"BIN":
                    self.bin = chunk
                off += 8 + clen
        else:
            self.json = json.loads(raw.decode("utf-8"))
        self._buffers: Dict[int, bytes] = {}
        self._images: Dict[int, np.ndarray] = {}

    def buffer(self, i: int) -> bytes:
        if i not in self._buffers:
            b = self.json["buffers"][i]
            uri = b.get("uri")
            if uri is None:
                self._buffers[i] = self.bin
            elif uri.startswith("data:"):
                self._buffers[i] = base64.b64decode(uri.split(",", 1)[1])
            else:
                self._buffers[i] = open(os.path.join(self.dir, uri), "rb").read()
        return self._buffers[i]

    def accessor(self, i: int) -> np.ndarray:
        a = self.json["accessors"][i]
        dt, nc = _COMP[a["componentType"]], _NCOMP[a["type"]]
        count = a["count"]
        if "bufferView" not in a:
            return np.zeros((count, nc), dt)
        bv = self.json["bufferViews"][a["bufferView"]]
        buf = self.buffer(bv["buffer"])
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        item = np.dtype(dt).itemsize * nc
        stride = bv.get("byteStride") or item
        if stride == item:
            arr = np.frombuffer(buf, dtype=dt, count=count * nc, offset=start).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(buf, dtype=dt, count=nc, offset=start + k * stride) for k in range(count)])
        if a.get("normalized") and dt != np.float32:
            info = np.iinfo(dt)
            arr = np.maximum(arr.astype(np.float32) / float(info.max), -1.0)
        return arr

    def image(self, tex_index: int) -> np.ndarray:
        """RGBA8 pixels of texture `tex_index` (decoded once)."""
        src = self.json["textures"][tex_index]["source"]
        if src not in self._images:
            from PIL import Image
            im = self.json["images"][src]
            if "bufferView" in im:
                bv = self.json["bufferViews"][im["bufferView"]]
                o = bv.get("byteOffset", 0)
                data = self.buffer(bv["buffer"])[o:o + bv["byteLength"]]
            elif im["uri"].startswith("data:"):
                data = base64.b64decode(im["uri"].split(",", 1)[1])
            else:
                data = open(os.path.join(self.dir, im["uri"]), "rb").read()
            self._images[src] = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"), dtype=np.uint8))
        return self._images[src]


def _node_matrix(n: dict) -> np.ndarray:
    if "matrix" in n:
        return np.asarray(n["matrix"], np.float64).reshape(4, 4).T  # glTF stores column-major
    t = np.asarray(n.get("translation", (0, 0, 0)), np.float64)
    x, y, z, w = n.get("rotation", (0, 0, 0, 1))
    s = np.asarray(n.get("scale", (1, 1, 1)), np.float64)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)
    m = np.eye(4)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m


def load(path: str) -> List[GltfScene]:
    doc = _Doc(path)
    js = doc.json
    materials: Dict[int, GltfMaterial] = {}
    default_material = GltfMaterial()

    def material(i: Optional[int]) -> GltfMaterial:
        if i is None:
            return default_material
        if i in materials:
            return materials[i]
        m = js["materials"][i]
        pbr = m.get("pbrMetallicRoughness", {})
        g = GltfMaterial(name=m.get("name") or "default", index=i,
                         base_color_factor=tuple(float(np.float32(v)) for v in pbr.get("baseColorFactor", (1, 1, 1, 1))),
                         metallic_factor=float(np.float32(pbr.get("metallicFactor", 1.0))),
                         roughness_factor=float(np.float32(pbr.get("roughnessFactor", 1.0))),
                         emissive_factor=tuple(float(np.float32(v)) for v in m.get("emissiveFactor", (0, 0, 0))))
        if "baseColorTexture" in pbr:
            g.base_color_texture = doc.image(pbr["baseColorTexture"]["index"])
        if "metallicRoughnessTexture" in pbr:
            mr = doc.image(pbr["metallicRoughnessTexture"]["index"])
            g.roughness_texture, g.metallic_texture = mr[:, :, 1].copy(), mr[:, :, 2].copy()
        if "normalTexture" in m:
            g.normal_texture = doc.image(m["normalTexture"]["index"])[:, :, :3].copy()
        if "occlusionTexture" in m:
            g.occlusion_texture = doc.image(m["occlusionTexture"]["index"])[:, :, 0].copy()
            g.occlusion_factor = float(np.float32(m["occlusionTexture"].get("strength", 1.0)))
        if "emissiveTexture" in m:
            g.emissive_texture = doc.image(m["emissiveTexture"]["index"])[:, :, :3].copy()
        materials[i] = g
        return g

    lights_def = (js.get("extensions", {}).get("KHR_lights_punctual", {}) or {}).get("lights", [])
    out: List[GltfScene] = []
    for sc in js.get("scenes", []):
        scene = GltfScene()

        def visit(ni: int, parent: np.ndarray):
            n = js["nodes"][ni]
            m = parent @ _node_matrix(n)
            if "mesh" in n:
                mesh = js["meshes"][n["mesh"]]
                for prim in mesh["primitives"]:
                    if prim.get("mode", 4) != 4:
                        continue
                    attr = prim["attributes"]
                    pos = doc.accessor(attr["POSITION"]).astype(np.float64)
                    idx = doc.accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(len(pos))
                    idx = idx[: (len(idx) // 3) * 3]
                    wpos = (np.concatenate([pos, np.ones((len(pos), 1))], axis=1) @ m.T)
                    wpos = (wpos[:, :3] / wpos[:, 3:4]).astype(np.float32)
                    model = GltfModel(name=mesh.get("name") or "unknown", material=material(prim.get("material")), positions=wpos[idx])
                    if "NORMAL" in attr:
                        nrm = doc.accessor(attr["NORMAL"]).astype(np.float64) @ m[:3, :3].T
                        ln = np.linalg.norm(nrm, axis=1, keepdims=True)
                        model.normals = (nrm / np.where(ln > 0, ln, 1.0)).astype(np.float32)[idx]
                    if "TEXCOORD_0" in attr:
                        model.tex_coords = doc.accessor(attr["TEXCOORD_0"]).astype(np.float32)[idx]
                    scene.models.append(model)
            ext = (n.get("extensions") or {}).get("KHR_lights_punctual")
            if ext is not None:
                ld = lights_def[ext["light"]]
                d = m[:3, :3] @ np.array([0.0, 0.0, -1.0])
                d = d / np.linalg.norm(d)
                scene.lights.append(GltfLight(kind=ld["type"], position=tuple(np.float32(m[:3, 3]).tolist()), direction=tuple(np.float32(d).tolist()),
                                              color=tuple(float(np.float32(v)) for v in ld.get("color", (1, 1, 1))),
                                              intensity=float(np.float32(ld.get("intensity", 1.0))),
                                              outer_cone_angle=float(np.float32((ld.get("spot") or {}).get("outerConeAngle", math.pi / 4.0))),
                                              name=ld.get("name")))
            if "camera" in n:
                c = js["cameras"][n["camera"]]
                if c.get("type") == "perspective":
                    p = c["perspective"]
                    scene.cameras.append(GltfCamera(transform=m.copy(), yfov=float(np.float32(p["yfov"])), znear=float(np.float32(p["znear"])),
                                                    zfar=float(np.float32(p.get("zfar", 1000.0)))))
            for ch in n.get("children", []):
                visit(ch, m)

        for root in sc.get("nodes", []):
            visit(root, np.eye(4))
        out.append(scene)
    return out
