"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under rustray_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rustray_amd.flat import (rr_camera, rr_config, rr_flat_scene, rr_frame, rr_pick_result, rr_texture)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class rro_counters(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes", C.c_uint64 * 2), ("leaf_prims", C.c_uint64 * 2), ("items", C.c_uint64 * 2),
                ("shaded_hits", C.c_uint64), ("texels", C.c_uint64)]

    def as_dict(self):
        return dict(rays_primary=self.rays_primary, rays_secondary=self.rays_secondary, rays_shadow=self.rays_shadow,
                    nodes=list(self.nodes), leaf_prims=list(self.leaf_prims), items=list(self.items),
                    shaded_hits=self.shaded_hits, texels=self.texels)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "oracle_math.h")] + \
           [os.path.join(_HERE, "..", "include", "rustray_hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.rro_render.restype = C.c_int
        _LIB.rro_render_scene.restype = C.c_int
        _LIB.rro_scene_create.restype = C.c_void_p
        _LIB.rro_scene_destroy.argtypes = [C.c_void_p]
        _LIB.rro_render_scene.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        _LIB.rro_wrap.restype = C.c_uint32
        _LIB.rro_wrap.argtypes = [C.c_float, C.c_uint32]
        _LIB.rro_fresnel.restype = C.c_float
        _LIB.rro_fresnel.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _LIB.rro_approx_equal.argtypes = [C.c_float, C.c_float]
        _LIB.rro_ray_ball.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _LIB.rro_jitter.argtypes = [C.c_void_p, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p]
        _LIB.rro_transmission.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        _LIB.rro_tex_interpolate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _LIB.rro_stdrng_u32.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        _LIB.rro_chacha_block.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        _LIB.rro_set_shot_era.argtypes = [C.c_int]
        _LIB.rro_set_shot_era.restype = None
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sample_table(samples: int):
    xy = np.zeros((max(samples, 1), 2), np.uint16)
    cs = C.c_uint32(0)
    lib().rro_sample_table(C.c_uint16(samples), _p(xy), C.byref(cs))
    return xy[:samples], int(cs.value)


class PreparedScene:
    """Acceleration structures built once; `fs_struct` must outlive the handle."""

    def __init__(self, fs_struct: rr_flat_scene, brute_force: bool = False):
        lib().rro_scene_create.restype = C.c_void_p
        self._fs = fs_struct
        self._h = C.c_void_p(lib().rro_scene_create(C.byref(fs_struct), 1 if brute_force else 0))

    def close(self):
        if self._h:
            lib().rro_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def render(fs_struct, cam: rr_camera, cfg: rr_config, sample_xy=None, window=None,
           n_threads: int = 1, brute_force: bool = False, want_counters: bool = False):
    """Render with the CPU restatement.  `fs_struct`: an rr_flat_scene or a PreparedScene.
    Returns dict(rgba, normal, depth, object_id[, counters])."""
    w, h = cam.width, cam.height
    rgba = np.zeros((h, w, 4), np.uint8)
    normal = np.zeros((h, w, 3), np.float32)
    depth = np.zeros((h, w), np.float32)
    oid = np.zeros((h, w), np.uint32)
    fr = rr_frame(rgba.ctypes.data, normal.ctypes.data, depth.ctypes.data, oid.ctypes.data)
    x0, y0, x1, y1 = window if window is not None else (0, 0, w, h)
    cnt = rro_counters()
    sxy = None
    if sample_xy is not None:
        sample_xy = np.ascontiguousarray(sample_xy, np.uint16)
        sxy = _p(sample_xy)
    if isinstance(fs_struct, PreparedScene):
        rc = lib().rro_render_scene(fs_struct._h, C.byref(cam), C.byref(cfg), sxy, C.byref(fr),
                                    C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), C.c_int(n_threads),
                                    C.byref(cnt) if want_counters else None)
    else:
        rc = lib().rro_render(C.byref(fs_struct), C.byref(cam), C.byref(cfg), sxy, C.byref(fr),
                              C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), C.c_int(n_threads),
                              C.c_int(1 if brute_force else 0), C.byref(cnt) if want_counters else None)
    if rc != 0:
        raise RuntimeError(f"rro_render failed: {rc}")
    out = dict(rgba=rgba, normal=normal, depth=depth, object_id=oid)
    if want_counters:
        out["counters"] = cnt.as_dict()
    return out


def post_process(rgba, normal, object_id, cavity: bool, outline: bool):
    h, w = rgba.shape[:2]
    src = np.ascontiguousarray(rgba, np.uint8)
    out = np.zeros_like(src)
    nrm = np.ascontiguousarray(normal, np.float32)
    ids = np.ascontiguousarray(object_id, np.uint32)
    lib().rro_post_process(C.c_uint32(w), C.c_uint32(h), C.c_int(int(cavity)), C.c_int(int(outline)), _p(src), _p(nrm), _p(ids), _p(out))
    return out


def trace_rays(fs_struct, origins, dirs, depth: int = 2, for_shadow: bool = False, brute_force: bool = False):
    """Raytracing::trace for given rays: (found, item, face, toi) arrays."""
    o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
    out = np.zeros((len(o), 4), np.uint32)
    lib().rro_trace_rays(C.byref(fs_struct), _p(o), _p(d), C.c_uint32(len(o)), C.c_uint32(depth), C.c_int(int(for_shadow)), C.c_int(int(brute_force)), _p(out))
    return out[:, 0].astype(bool), out[:, 1].astype(np.int32), out[:, 2].copy(), out[:, 3].copy().view(np.float32)


class ray_log:
    """with ray_log(cap) as log: render(..., n_threads=1) -> log.rays() = every trace() call of the render, in order."""
    def __init__(self, cap=1 << 16):
        self.buf = np.zeros((cap, 12), np.uint32); self.n = C.c_uint32(0)
    def __enter__(self):
        lib().rro_set_ray_log(_p(self.buf), C.c_uint32(len(self.buf)), C.byref(self.n)); return self
    def __exit__(self, *a):
        lib().rro_set_ray_log(None, C.c_uint32(0), None)
    def rays(self):
        b = self.buf[:self.n.value]
        return dict(origin=b[:, 0:3].copy().view(np.float32), dir=b[:, 3:6].copy().view(np.float32), depth=b[:, 6].copy(), for_shadow=b[:, 7].astype(bool),
                    found=b[:, 8].astype(bool), item=b[:, 9].astype(np.int32), face=b[:, 10].copy(), toi=b[:, 11].copy().view(np.float32))


def pick(fs_struct, cam, x, y):
    r = rr_pick_result()
    lib().rro_pick(C.byref(fs_struct), C.byref(cam), C.c_int(x), C.c_int(y), C.byref(r))
    return r


def algorithmic_bytes(counters: dict, width: int, height: int) -> dict:
    """SURVEY.md 8d byte model: 64 B per BVH2 node, 36 B per leaf primitive, 128 B per instance
    record, 224 B per shaded hit, 4 B per texel, 24 B per output pixel."""
    c = counters
    closest = c["nodes"][0] * 64 + c["leaf_prims"][0] * 36 + c["items"][0] * 128
    shadow = c["nodes"][1] * 64 + c["leaf_prims"][1] * 36 + c["items"][1] * 128
    shade = c["shaded_hits"] * 224 + c["texels"] * 4
    total = closest + shadow + shade + width * height * 24
    n_closest = c["rays_primary"] + c["rays_secondary"]
    n_rays = n_closest + c["rays_shadow"]
    return dict(total=total, closest=closest, shadow=shadow, shade=shade, rays=n_rays,
                bytes_per_ray=total / max(n_rays, 1),
                bytes_per_closest_ray=closest / max(n_closest, 1),
                bytes_per_shadow_ray=shadow / max(c["rays_shadow"], 1))
