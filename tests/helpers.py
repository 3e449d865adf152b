"""Shared helpers for the test suite: fixture scenes, cameras, comparisons."""
from __future__ import annotations

import os

import numpy as np

from rustray_amd.camera import Camera
from rustray_amd.flat import FlatScene, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_scene(name: str) -> FlatScene:
    return FlatScene.load(os.path.join(SCENES, name + ".npz"))


def camera_for(fs: FlatScene, width: int, height: int) -> Camera:
    st = dict(fs.meta["camera"])
    st["width"], st["height"] = width, height
    return Camera.from_state(st)


def compare_frames(a: dict, b: dict, rgb_tol: int = 1):
    """Returns a dict of mismatch statistics between two renders (a = candidate, b = oracle)."""
    ra, rb = a["rgba"].astype(np.int32), b["rgba"].astype(np.int32)
    diff = np.abs(ra[..., :3] - rb[..., :3]).max(axis=-1)
    res = dict(max_rgb_diff=int(diff.max()), n_rgb_over=int((diff > rgb_tol).sum()), n_pixels=int(diff.size),
               alpha_ok=bool((ra[..., 3] == 255).all()))
    if "object_id" in a and "object_id" in b:
        res["n_id_diff"] = int((a["object_id"] != b["object_id"]).sum())
    if "depth" in a and "depth" in b:
        da, db = a["depth"].astype(np.float64), b["depth"].astype(np.float64)
        res["max_depth_rel"] = float((np.abs(da - db) / np.maximum(np.abs(db), 1e-3)).max())
    if "normal" in a and "normal" in b:
        na, nb = a["normal"], b["normal"]
        both_nan = np.isnan(na) & np.isnan(nb)
        d = np.where(both_nan, 0.0, np.abs(na.astype(np.float64) - nb.astype(np.float64)))
        res["nan_mismatch"] = int((np.isnan(na) != np.isnan(nb)).sum())
        res["max_normal_abs"] = float(np.nanmax(d)) if d.size else 0.0
    return res
