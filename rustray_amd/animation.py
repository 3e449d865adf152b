"""Host-side mirror of rustray's keyframe animation (reference src/animation.rs).

Only item transforms change between frames (`Scene::apply_frame`, reference src/scene.rs:1695-1713:
`apply_mat` REPLACES the item's matrix by the interpolated T * Rz * Ry * Rx * S), so a frame step on the
device is `rr_scene_update_transforms` + a top-level rebuild; meshes and their trees stay resident.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .scene import get_transformation, inverse_affine

F32 = np.float32


@dataclass
class Frame:  # src/animation.rs:9-30
    object_name: str
    translation: Optional[tuple] = None
    rotation: Optional[tuple] = None   # radians
    scale: Optional[tuple] = None


@dataclass
class Keyframe:  # src/animation.rs:35-52
    time: int
    objects: List[Frame] = field(default_factory=list)


class Animation:
    def __init__(self, enabled: bool = False, fps: int = 25, keyframes: Optional[List[Keyframe]] = None):
        self.enabled, self.fps, self.keyframes = enabled, fps, keyframes or []

    # -- (de)serialisation used by FlatScene.meta["animation"] ------------------------------
    @classmethod
    def from_json(cls, a: Optional[dict]) -> "Animation":
        """The `animation` block of a scene file (reference src/scene.rs:549-628); rotations are degrees there."""
        an = cls()
        if not a:
            return an
        if a.get("fps") is not None:
            an.fps = int(a["fps"])
        if a.get("enabled") is not None:
            an.enabled = bool(a["enabled"])

        def vec(t, key, to_rad=False):
            v = (t or {}).get(key)
            if isinstance(v, dict) and all(v.get(k) is not None for k in "xyz"):
                out = tuple(float(F32(v[k])) for k in "xyz")
                return tuple(float(F32(math.radians(c))) for c in out) if to_rad else out
            return None
        for kf in a.get("keyframes") or []:
            if kf.get("time") is None:
                continue
            frames = []
            for o in kf.get("objects") or []:
                t = o.get("transformation")
                frames.append(Frame(o["name"], vec(t, "translation"), vec(t, "rotation", True), vec(t, "scale")))
            an.keyframes.append(Keyframe(int(kf["time"]), frames))
        return an

    # -- src/animation.rs:79-130 --------------------------------------------------------------
    def has_initial_keyframe(self) -> bool:
        return bool(self.keyframes) and self.keyframes[0].time == 0

    def get_frames_amount_to_render(self) -> int:
        last = self.keyframes[-1].time if self.keyframes else 0
        return int(math.floor(float(self.fps) * (last / 1000.0)))

    def has_animation(self) -> bool:
        return self.enabled and self.get_frames_amount_to_render() > 0 and self.has_initial_keyframe() and len(self.keyframes) >= 2

    def get_keyframes_for_frame(self, frame: int):
        timestamp = int(math.floor((1000.0 / self.fps) * frame))
        first = last = self.keyframes[0]
        for i, k in enumerate(self.keyframes):
            if k.time <= timestamp:
                first = k
                last = self.keyframes[i] if i + 1 >= len(self.keyframes) else self.keyframes[i + 1]
        pos, diff = timestamp - first.time, last.time - first.time
        with np.errstate(divide="ignore", invalid="ignore"):
            factor = float(np.float64(1.0) / np.float64(diff) * np.float64(pos))   # 1/0 * 0 = NaN at the last keyframe, as in Rust
        return first, last, factor

    def get_trans_for_frame(self, frame: int, object_name: str) -> Optional[np.ndarray]:
        first, last, factor = self.get_keyframes_for_frame(frame)
        a = next((o for o in first.objects if o.object_name == object_name), None)
        b = next((o for o in last.objects if o.object_name == object_name), None)
        if a is None or b is None:
            return None
        f = F32(factor)

        def lerp(p, q, default):
            if p is None or q is None:
                return default
            return tuple(float(F32(x) + f * (F32(y) - F32(x))) for x, y in zip(p, q))   # helper::interpolate
        translation = lerp(a.translation, b.translation, (0.0, 0.0, 0.0))
        scale = lerp(a.scale, b.scale, (1.0, 1.0, 1.0))
        rotation = lerp(a.rotation, b.rotation, (0.0, 0.0, 0.0))
        return get_transformation(np.eye(4, dtype=F32), translation, scale, rotation)

    def frame_exists(self, frame: int) -> bool:
        """Scene::frame_exists (reference src/scene.rs:1690-1693)."""
        return self.has_animation() and frame < self.get_frames_amount_to_render()

    def frames_to_render(self, start: int = 0):
        """Frames the reference's loop visits (src/run.rs:421-465): `start`, then every next frame that exists."""
        frames = [start]
        while self.frame_exists(frames[-1] + 1):
            frames.append(frames[-1] + 1)
        return frames

    # -- Scene::apply_frame (reference src/scene.rs:1695-1713) on a flat scene ----------------------
    def frame_transforms(self, flat_scene, frame: int):
        """(trans, trans_inv) arrays (n_items, 4, 4) for `frame`, or None when the reference would not touch the scene."""
        if not self.has_animation() or frame > self.get_frames_amount_to_render():
            return None
        trans = np.stack([np.asarray(it.trans, F32) for it in flat_scene.items]) if flat_scene.items else np.zeros((0, 4, 4), F32)
        for i, it in enumerate(flat_scene.items):
            m = self.get_trans_for_frame(frame, it.name)
            if m is not None:
                trans[i] = m
        inv = np.stack([inverse_affine(t) for t in trans]) if len(trans) else trans.copy()
        return trans, inv

    def to_meta(self) -> dict:
        return {"enabled": self.enabled, "fps": self.fps,
                "keyframes": [{"time": k.time, "objects": [{"name": o.object_name, "translation": o.translation, "rotation": o.rotation,
                                                           "scale": o.scale} for o in k.objects]} for k in self.keyframes]}

    @classmethod
    def from_meta(cls, m: Optional[dict]) -> "Animation":
        an = cls()
        if not m:
            return an
        an.enabled, an.fps = bool(m["enabled"]), int(m["fps"])
        for k in m["keyframes"]:
            an.keyframes.append(Keyframe(int(k["time"]), [Frame(o["name"], tuple(o["translation"]) if o["translation"] else None,
                                                             tuple(o["rotation"]) if o["rotation"] else None,
                                                             tuple(o["scale"]) if o["scale"] else None) for o in k["objects"]]))
        return an
