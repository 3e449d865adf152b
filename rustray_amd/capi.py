"""ctypes binding of librustray_hip.so (the C ABI of include/rustray_hip.h).

This is the only way Python reaches the trace loop: there is no CPU or PyTorch
fallback.  If the HIP library has not been built, or no GPU is present, the
calls below raise — they never silently compute something else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .flat import (RR_ABI_VERSION, rr_camera, rr_config, rr_flat_scene, rr_frame, rr_frame_stats, rr_material, rr_pick_result, rr_region, rr_tuning)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RUSTRAY_HIP_LIB") or os.path.join(_HERE, "librustray_hip.so")  # override: developer A/B builds
_LIB = None

# every symbol include/rustray_hip.h declares (tests/test_abi.py checks the list against the header)
EXPORTS = ["rr_abi_version", "rr_device_count", "rr_last_error", "rr_scene_create", "rr_scene_destroy", "rr_scene_update_transforms",
           "rr_scene_update_materials", "rr_scene_set_tuning", "rr_scene_get_tuning", "rr_scene_set_compat",
           "rr_sample_table", "rr_render", "rr_render_multi", "rr_multi_lock_order", "rr_render_progressive", "rr_render_progressive_tiles", "rr_region_pixel_count", "rr_render_region_device",
           "rr_deinterleave_device", "rr_deinterleave_packed_device", "rr_pick", "rr_trace_rays", "rr_scene_last_stats", "rr_post_process", "rr_post_process_device"]


class RustrayHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rustray_hip error {code}: {msg}")
        self.code = code


PASS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64)


def source_id() -> str:
    """Identity of the kernel sources next to the library (sha256 over csrc/ + the header, 16 hex digits): what ties a committed
    counter profile (profiles/*_sq_counters.json) to the build a bench run measures."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(_HERE, "csrc", f) for f in ("rr_api.hip", "rr_kernels.hip", "rr_bvh.cpp", "rr_bvh.h", "rr_device.h", "rr_math.h")]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "rustray_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False) -> str:
    """Compile librustray_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", csrc], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C rustray_amd/csrc` — there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.rr_last_error.restype = C.c_char_p
        got = 1
        if hasattr(L, "rr_abi_version"):
            L.rr_abi_version.restype = C.c_uint32
            got = L.rr_abi_version()
        if got != RR_ABI_VERSION and not os.environ.get("RUSTRAY_HIP_LIB"):  # (a developer A/B build of an older revision is the caller's business)
            raise RuntimeError(f"{LIB_PATH} speaks ABI version {got}, this binding {RR_ABI_VERSION}: rebuild the library")
        L.rr_region_pixel_count.restype = C.c_uint64
        L.rr_region_pixel_count.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(rr_region)]
        L.rr_scene_create.argtypes = [C.POINTER(rr_flat_scene), C.c_int, C.POINTER(C.c_void_p)]
        L.rr_scene_destroy.argtypes = [C.c_void_p]
        L.rr_scene_destroy.restype = None
        L.rr_scene_update_transforms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rr_sample_table.argtypes = [C.c_uint16, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rr_render.argtypes = [C.c_void_p, C.POINTER(rr_camera), C.POINTER(rr_config), C.c_void_p, C.POINTER(rr_frame), C.c_void_p]
        L.rr_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(rr_camera), C.POINTER(rr_config), C.c_void_p, C.POINTER(rr_frame), C.c_void_p]
        L.rr_render_progressive.argtypes = [C.c_void_p, C.POINTER(rr_camera), C.POINTER(rr_config), C.c_void_p, C.POINTER(rr_frame),
                                            C.c_uint32, PASS_FN, C.c_void_p, C.c_void_p]
        if hasattr(L, "rr_render_progressive_tiles") or not os.environ.get("RUSTRAY_HIP_LIB"):   # (a developer A/B build of an older revision may lack it)
            L.rr_render_progressive_tiles.argtypes = [C.c_void_p, C.POINTER(rr_camera), C.POINTER(rr_config), C.c_void_p, C.POINTER(rr_frame),
                                                      C.c_uint32, PASS_FN, C.c_void_p, C.c_void_p]
        L.rr_render_region_device.argtypes = [C.c_void_p, C.POINTER(rr_camera), C.POINTER(rr_config), C.c_void_p,
                                              C.POINTER(rr_region), C.POINTER(rr_frame), C.c_void_p, C.c_void_p]
        L.rr_deinterleave_device.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rr_pick.argtypes = [C.c_void_p, C.POINTER(rr_camera), C.c_int, C.c_int, C.POINTER(rr_pick_result)]
        L.rr_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rr_scene_last_stats.argtypes = [C.c_void_p, C.POINTER(rr_frame_stats)]
        L.rr_scene_update_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.rr_scene_set_tuning.argtypes = [C.c_void_p, C.POINTER(rr_tuning)]
        L.rr_scene_get_tuning.argtypes = [C.c_void_p, C.POINTER(rr_tuning)]
        L.rr_post_process.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.rr_post_process_device.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rr_math_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_int]
        _LIB = L
    return _LIB


def _check(rc: int):
    if rc != 0:
        raise RustrayHipError(rc, lib().rr_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(lib().rr_device_count())


def sample_table(samples: int):
    """Built-in sub-sample table of the library (reference src/raytracing.rs:290-313)."""
    xy = np.zeros((max(samples, 1), 2), np.uint16)
    cs = C.c_uint32(0)
    _check(lib().rr_sample_table(C.c_uint16(samples), xy.ctypes.data_as(C.c_void_p), C.byref(cs)))
    return xy[:samples], int(cs.value)


def region_pixel_count(width: int, height: int, tile_w: int, tile_h: int, n_ranks: int, rank: int) -> int:
    rg = rr_region(tile_w, tile_h, n_ranks, rank)
    return int(lib().rr_region_pixel_count(width, height, C.byref(rg)))


def _sxy(sample_xy):
    if sample_xy is None:
        return None, None
    a = np.ascontiguousarray(sample_xy, np.uint16)
    return a, a.ctypes.data_as(C.c_void_p)


class DeviceScene:
    """Owns one `rr_scene*` (scene uploaded to one GPU, acceleration structures built)."""

    def __init__(self, flat_scene, device: int = 0):
        self._h = C.c_void_p(None)
        self.device = device
        self._flat = flat_scene               # keeps host arrays alive during the call
        fs = flat_scene.c_struct() if hasattr(flat_scene, "c_struct") else flat_scene
        _check(lib().rr_scene_create(C.byref(fs), device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rr_scene_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- whole frame into host arrays -------------------------------------------
    def render(self, cam: rr_camera, cfg: rr_config, sample_xy=None, aux: bool = True):
        w, h = cam.width, cam.height
        rgba = np.zeros((h, w, 4), np.uint8)
        out = dict(rgba=rgba)
        fr = rr_frame(rgba.ctypes.data, None, None, None)
        if aux:
            out["normal"] = np.zeros((h, w, 3), np.float32)
            out["depth"] = np.zeros((h, w), np.float32)
            out["object_id"] = np.zeros((h, w), np.uint32)
            fr = rr_frame(rgba.ctypes.data, out["normal"].ctypes.data, out["depth"].ctypes.data, out["object_id"].ctypes.data)
        keep, p = _sxy(sample_xy)
        _check(lib().rr_render(self._h, C.byref(cam), C.byref(cfg), p, C.byref(fr), None))
        return out

    def render_progressive(self, cam: rr_camera, cfg: rr_config, on_pass, min_passes: int = 8, sample_xy=None, aux: bool = True, tiles: bool = False):
        """rr_render_progressive: `on_pass(out, samples_done, samples_total)` sees the frame resolved over the samples
        finished so far after every device batch; a truthy return stops the frame (RustrayHipError, code -6).
        tiles=True: rr_render_progressive_tiles -- `min_passes` passes of interleaved 32x8 tiles, every pixel final when it appears."""
        w, h = cam.width, cam.height
        rgba = np.zeros((h, w, 4), np.uint8)
        out = dict(rgba=rgba)
        fr = rr_frame(rgba.ctypes.data, None, None, None)
        if aux:
            out["normal"] = np.zeros((h, w, 3), np.float32)
            out["depth"] = np.zeros((h, w), np.float32)
            out["object_id"] = np.zeros((h, w), np.uint32)
            fr = rr_frame(rgba.ctypes.data, out["normal"].ctypes.data, out["depth"].ctypes.data, out["object_id"].ctypes.data)
        keep, p = _sxy(sample_xy)
        errors = []

        def _cb(user, done, total):
            try:
                return 1 if on_pass(out, int(done), int(total)) else 0
            except BaseException as e:  # an exception must not unwind through the C frames
                errors.append(e)
                return 1
        cb = PASS_FN(_cb)
        fn = lib().rr_render_progressive_tiles if tiles else lib().rr_render_progressive
        rc = fn(self._h, C.byref(cam), C.byref(cfg), p, C.byref(fr), int(min_passes), cb, None, None)
        if errors:
            raise errors[0]
        _check(rc)
        return out

    # -- one rank's region into device (torch) tensors ----------------------------
    def render_region_device(self, cam, cfg, region: rr_region, out_ptrs, stream_ptr=None, sample_xy=None):
        """out_ptrs: (rgba8, normal, depth, object_id) device pointers (ints, None allowed for aux)."""
        fr = rr_frame(*[C.c_void_p(p) if p else None for p in out_ptrs])
        keep, p = _sxy(sample_xy)
        _check(lib().rr_render_region_device(self._h, C.byref(cam), C.byref(cfg), p, C.byref(region), C.byref(fr),
                                             C.c_void_p(stream_ptr) if stream_ptr else None, None))

    def update_transforms(self, trans: np.ndarray, trans_inv: np.ndarray):
        """trans / trans_inv: (n_items, 4, 4) in math layout."""
        t = np.ascontiguousarray(np.transpose(np.asarray(trans, np.float32), (0, 2, 1)))
        ti = np.ascontiguousarray(np.transpose(np.asarray(trans_inv, np.float32), (0, 2, 1)))
        _check(lib().rr_scene_update_transforms(self._h, t.ctypes.data_as(C.c_void_p), ti.ctypes.data_as(C.c_void_p)))

    def pick(self, cam: rr_camera, x: int, y: int) -> rr_pick_result:
        r = rr_pick_result()
        _check(lib().rr_pick(self._h, C.byref(cam), x, y, C.byref(r)))
        return r

    def update_materials(self, materials):
        """rr_scene_update_materials: `materials` = the flat scene's material list (same length and order), edited."""
        arr = (rr_material * len(materials))(*[m.c_struct() if hasattr(m, "c_struct") else m for m in materials])
        _check(lib().rr_scene_update_materials(self._h, arr, len(materials)))

    def set_tuning(self, **kw):
        """rr_scene_set_tuning: sample_group, queue_budget_bytes, shade_chunk_rays, kernel_timing, multi_force_staged, bin_min_rays (others keep their value)."""
        t = rr_tuning()
        _check(lib().rr_scene_get_tuning(self._h, C.byref(t)))
        for k, v in kw.items():
            if k not in ("sample_group", "queue_budget_bytes", "shade_chunk_rays", "kernel_timing", "multi_force_staged", "bin_min_rays"):
                raise TypeError(f"unknown tuning field {k}")
            setattr(t, k, int(v) & 0xffffffffffffffff)
        t.struct_size = C.sizeof(rr_tuning)
        _check(lib().rr_scene_set_tuning(self._h, C.byref(t)))

    def trace_rays(self, origins, dirs, depth: int = 2):
        """rr_trace_rays: closest hits of caller-supplied rays -> (found, item, face, toi) arrays."""
        o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
        n = len(o)
        out = np.zeros((n, 5), np.uint32)
        _check(lib().rr_trace_rays(self._h, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(depth), out.ctypes.data_as(C.c_void_p)))
        return out[:, 0].astype(bool), out[:, 1].astype(np.int32), out[:, 3].copy(), out[:, 4].copy().view(np.float32)

    def set_profiling(self, on: bool):
        self.set_tuning(kernel_timing=1 if on else 0)

    def set_compat(self, flags: int):
        """rr_scene_set_compat: behaviours of earlier reference binaries (1 = shadows attenuated by the occluder's alpha)."""
        lib().rr_scene_set_compat.argtypes = [C.c_void_p, C.c_uint32]
        _check(lib().rr_scene_set_compat(self._h, C.c_uint32(flags)))

    def stats(self) -> dict:
        st = rr_frame_stats()
        _check(lib().rr_scene_last_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in rr_frame_stats._fields_ if not k.startswith("_")}


def render_multi(device_scenes, cam: rr_camera, cfg: rr_config, sample_xy=None, aux: bool = True):
    """rr_render_multi: one frame over several DeviceScene handles (one per GPU) from this one process, into host arrays."""
    w, h = cam.width, cam.height
    rgba = np.zeros((h, w, 4), np.uint8)
    out = dict(rgba=rgba)
    fr = rr_frame(rgba.ctypes.data, None, None, None)
    if aux:
        out["normal"] = np.zeros((h, w, 3), np.float32)
        out["depth"] = np.zeros((h, w), np.float32)
        out["object_id"] = np.zeros((h, w), np.uint32)
        fr = rr_frame(rgba.ctypes.data, out["normal"].ctypes.data, out["depth"].ctypes.data, out["object_id"].ctypes.data)
    keep, p = _sxy(sample_xy)
    handles = (C.c_void_p * len(device_scenes))(*[ds._h for ds in device_scenes])
    _check(lib().rr_render_multi(handles, len(device_scenes), C.byref(cam), C.byref(cfg), p, C.byref(fr), None))
    return out


def deinterleave_device(width, height, tile_w, tile_h, n_ranks, elem_bytes, src_ptr, dst_ptr, device, stream_ptr=None):
    _check(lib().rr_deinterleave_device(width, height, tile_w, tile_h, n_ranks, elem_bytes, C.c_void_p(src_ptr),
                                        C.c_void_p(dst_ptr), device, C.c_void_p(stream_ptr) if stream_ptr else None))


def deinterleave_packed_device(width, height, tile_w, tile_h, n_ranks, packs_ptr, pack_stride, section_offset, elem_bytes, dst_ptrs, device, stream_ptr=None):
    """rr_deinterleave_packed_device: the gathered packs of all ranks -> the frame-order buffers, one launch.  section_offset / elem_bytes /
    dst_ptrs: 4 entries (rgba, normal, depth, object id); elem_bytes 0 = absent."""
    so = (C.c_uint64 * 4)(*[int(v) for v in section_offset])
    eb = (C.c_uint32 * 4)(*[int(v) for v in elem_bytes])
    dp = (C.c_void_p * 4)(*[C.c_void_p(int(v)) if v else None for v in dst_ptrs])
    L = lib()
    L.rr_deinterleave_packed_device.argtypes = [C.c_uint32] * 5 + [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                                C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    _check(L.rr_deinterleave_packed_device(width, height, tile_w, tile_h, n_ranks, C.c_void_p(packs_ptr), C.c_uint64(pack_stride), so, eb, dp,
                                           device, C.c_void_p(stream_ptr) if stream_ptr else None))


def post_process(rgba: np.ndarray, normal, object_id, cavity: bool, outline: bool, device: int = 0) -> np.ndarray:
    """run_post_processing (reference src/post_processing.rs:123-181) on the GPU, host arrays in and out."""
    h, w = rgba.shape[:2]
    src = np.ascontiguousarray(rgba, np.uint8)
    out = np.zeros_like(src)
    nrm = np.ascontiguousarray(normal, np.float32) if normal is not None else None
    ids = np.ascontiguousarray(object_id, np.uint32) if object_id is not None else None
    _check(lib().rr_post_process(w, h, int(cavity), int(outline), src.ctypes.data_as(C.c_void_p),
                                 nrm.ctypes.data_as(C.c_void_p) if nrm is not None else None,
                                 ids.ctypes.data_as(C.c_void_p) if ids is not None else None,
                                 out.ctypes.data_as(C.c_void_p), device))
    return out


def post_process_device(w, h, cavity, outline, rgba_ptr, normal_ptr, id_ptr, out_ptr, device=0, stream_ptr=None):
    _check(lib().rr_post_process_device(w, h, int(cavity), int(outline), C.c_void_p(rgba_ptr), C.c_void_p(normal_ptr) if normal_ptr else None,
                                        C.c_void_p(id_ptr) if id_ptr else None, C.c_void_p(out_ptr), device,
                                        C.c_void_p(stream_ptr) if stream_ptr else None))


def math_probe(op: int, a, b=None, c=None, seed: int = 0, device: int = 0):
    a = np.ascontiguousarray(a, np.float32)
    n = len(a)
    b = np.ascontiguousarray(b, np.float32) if b is not None else None
    c = np.ascontiguousarray(c, np.float32) if c is not None else None
    outs = [np.zeros(n, np.float32) for _ in range(3)]
    _check(lib().rr_math_probe(op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
                               c.ctypes.data_as(C.c_void_p) if c is not None else None, n,
                               *[o.ctypes.data_as(C.c_void_p) for o in outs], C.c_uint64(seed), device))
    return outs
