"""Host-side mirror of the reference's frame scheduler, on top of the C ABI.

`Raytracing` and `RendererManager` keep the names and call surface of the
reference (src/raytracing.rs:205-273, src/renderer.rs:38-251) so that host code
and tests read like the reference's; what changes is that `start()` issues ONE
frame-level call into librustray_hip.so instead of spawning num_cpus-2 worker
threads over a shuffled queue of 2x2-pixel cells (src/renderer.rs:105-172).

`TiledFrame` is the multi-GPU form (no reference counterpart):
one process per GPU, the frame cut into interleaved tiles, every rank renders
its tiles into a compact device buffer, one gather to rank 0 over RCCL, one
de-interleave kernel.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np

from . import capi
from .camera import Camera
from .flat import FlatScene, make_config, rr_config, rr_region


class Raytracing:
    """Scene + RaytracingConfig, as reference `Raytracing` (src/raytracing.rs:205-224)."""

    def __init__(self, flat_scene: FlatScene, camera: Camera, device: int = 0):
        self.flat_scene = flat_scene
        self.camera = camera
        self.config: rr_config = make_config()
        cfg = flat_scene.meta.get("config") or {}
        # a scene file's "config" block overrides whatever the caller set before load (src/scene.rs:180-198)
        self.apply_config(**{k: v for k, v in cfg.items() if k in
                             ("samples", "monte_carlo", "focal_length", "aperture_size", "fog_density",
                              "max_recursion", "gamma_correction")})
        self.device_scene = capi.DeviceScene(flat_scene, device)

    def apply_config(self, **kw):
        for k, v in kw.items():
            if k == "fog_color":
                self.config.fog_color[:] = list(v)
            elif k in ("monte_carlo", "gamma_correction"):
                setattr(self.config, k, int(bool(v)))
            else:
                setattr(self.config, k, v)

    def render_frame(self, sample_xy=None, aux: bool = True) -> dict:
        """All pixels of `Raytracing::render(x, y)` (src/raytracing.rs:275-427) in one call."""
        return self.device_scene.render(self.camera.c_struct(), self.config, sample_xy=sample_xy, aux=aux)

    def pick(self, x: int, y: int):
        """Raytracing::pick (src/raytracing.rs:237-273): Some((id, distance)) or None."""
        r = self.device_scene.pick(self.camera.c_struct(), x, y)
        return (int(r.object_id), float(r.distance)) if r.hit else None

    def close(self):
        self.device_scene.close()


class RendererManager:
    """Call surface of reference `RendererManager` (src/renderer.rs:63-251)."""

    def __init__(self, width: int, height: int, raytracing: Raytracing):
        self.width, self.height = width, height
        self.raytracing = raytracing
        self.thread_amount = 1  # one frame-level device call replaces the worker threads
        self._running = False
        self._pixels_rendered = 0
        self._start = time.time()
        self._done_ms = 0
        self.image = self.normals = self.depth = self.objects = None

    def update_resolution(self, width: int, height: int):
        self.width, self.height = width, height

    def start(self, on_pass=None, min_passes: int = 8):
        """One frame.  With `on_pass(manager)` the frame is rendered progressively: after every device batch the
        image buffers hold the frame over the samples finished so far (what Run::apply_pixels shows while the
        reference renders, src/run.rs:506-545) and the callback may call `stop()` to end the frame early
        (RendererManager::stop, src/renderer.rs:174-198)."""
        self._start = time.time()
        self._done_ms = 0
        self._pixels_rendered = 0
        self._running = True
        self.raytracing.camera.init(self.width, self.height)
        if on_pass is not None:
            def _pass(out, done, total):
                self.image, self.normals, self.depth, self.objects = out["rgba"], out["normal"], out["depth"], out["object_id"]
                # the reference counts finished pixels; a pass finishes a share of every pixel's samples
                self._pixels_rendered = (self.width * self.height * done) // total
                on_pass(self)
                return not self._running
            try:
                out = self.raytracing.device_scene.render_progressive(self.raytracing.camera.c_struct(), self.raytracing.config,
                                                                      _pass, min_passes=min_passes)
            except capi.RustrayHipError as e:
                if e.code != -6:
                    raise
                self._done_ms = int((time.time() - self._start) * 1000.0)
                return  # stopped: the buffers keep the last preview
        else:
            out = self.raytracing.render_frame()
        # what Run::apply_pixels stores per PixelData (src/run.rs:519-541)
        self.image, self.normals, self.depth, self.objects = out["rgba"], out["normal"], out["depth"], out["object_id"]
        self._pixels_rendered = self.width * self.height
        self._done_ms = int((time.time() - self._start) * 1000.0)

    def stop(self):
        self._running = False

    def restart(self, width: int, height: int):
        self.stop()
        self.update_resolution(width, height)
        self.start()

    def is_running(self) -> bool:
        return self._running

    def is_done(self) -> bool:
        return self._pixels_rendered == self.width * self.height

    def get_rendered_pixels(self) -> int:
        return self._pixels_rendered

    def check_and_get_elapsed_time(self) -> int:
        return self._done_ms if self._done_ms > 0 else int((time.time() - self._start) * 1000.0)


# ---------------------------------------------------------------------------
# animation: the frame loop, one frame per GPU at a time
# ---------------------------------------------------------------------------
class AnimationRun:
    """The reference's animation loop (Run::render_next_frame_if_possible, src/run.rs:421-465: apply_frame, restart,
    wait for completion, next frame) over ONE resident device scene: only the item transforms change per frame
    (Scene::apply_frame, src/scene.rs:1695-1713 -> rr_scene_update_transforms), geometry, per-mesh trees and textures
    stay in HBM.  With world_size > 1 the frames are dealt round-robin to the ranks (one process per GPU, scene
    replicated): frames are independent, so the data path needs no collective; `gather()` brings the finished
    RGBA8 frames to rank 0 if one process is to write them out."""

    def __init__(self, raytracing: Raytracing, animation, rank: int = 0, world_size: int = 1, start_frame: int = 0):
        self.raytracing, self.animation = raytracing, animation
        self.rank, self.world_size = rank, world_size
        self.frames = animation.frames_to_render(start_frame)

    def my_frames(self, rank: Optional[int] = None):
        r = self.rank if rank is None else rank
        return self.frames[r::self.world_size]

    def render(self, on_frame=None, render_fn=None) -> dict:
        """{frame: out} for this rank's frames.  render_fn(frame) replaces the device call (host-logic tests)."""
        done = {}
        for f in self.my_frames():
            if render_fn is not None:
                out = render_fn(f)
            else:
                tr = self.animation.frame_transforms(self.raytracing.flat_scene, f)
                if tr is not None:
                    self.raytracing.device_scene.update_transforms(*tr)
                out = self.raytracing.render_frame()
            done[f] = out
            if on_frame is not None:
                on_frame(f, out)
        return done

    def gather(self, done: dict, via_cpu: bool = False):
        """Rank 0: list of (frame, rgba) in frame order; other ranks: None.  One gather of the stacked RGBA8 frames."""
        import torch
        import torch.distributed as dist
        mine = self.my_frames()
        if self.world_size == 1:
            return [(f, done[f]["rgba"]) for f in mine]
        on_gpu = (not via_cpu) and torch.cuda.is_available() and dist.get_backend() == "nccl"  # RCCL moves device tensors only
        h, w = (done[mine[0]]["rgba"].shape[:2]) if mine else (0, 0)
        shape = torch.tensor([h, w], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(shape, op=dist.ReduceOp.MAX)  # a rank without frames still needs the shape
        h, w = int(shape[0]), int(shape[1])
        n_max = max(len(self.my_frames(r)) for r in range(self.world_size))
        pad = torch.zeros((n_max, h, w, 4), dtype=torch.uint8)
        for i, f in enumerate(mine):
            pad[i] = torch.from_numpy(np.ascontiguousarray(done[f]["rgba"]))
        if on_gpu:
            pad = pad.cuda()
        gl = [torch.empty_like(pad) for _ in range(self.world_size)] if self.rank == 0 else None
        dist.gather(pad, gl, dst=0)
        if self.rank != 0:
            return None
        out = []
        for r in range(self.world_size):
            for i, f in enumerate(self.my_frames(r)):
                out.append((f, gl[r][i].cpu().numpy()))
        return sorted(out, key=lambda t: t[0])


# ---------------------------------------------------------------------------
# multi-GPU tiling
# ---------------------------------------------------------------------------
def region_pixels(width: int, height: int, tile_w: int, tile_h: int, n_ranks: int, rank: int) -> np.ndarray:
    """(n, 2) array of (x, y) in the order `rr_region` defines (include/rustray_hip.h)."""
    tx, ty = (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h
    out = []
    for t in range(rank, tx * ty, n_ranks):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        x1, y1 = min(x0 + tile_w, width), min(y0 + tile_h, height)
        ys, xs = np.mgrid[y0:y1, x0:x1]
        out.append(np.stack([xs.ravel(), ys.ravel()], axis=1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros((0, 2), np.int64)


class TiledFrame:
    """One rank of a tiled multi-GPU frame.

    render_fn(region, n_pixels) must return a dict of compact per-rank torch tensors
    {"rgba": (n,4) uint8, ["normal": (n,3) f32, "depth": (n,) f32, "object_id": (n,) int32]}.
    `gather()` collects them on rank 0 (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
    on CPU) and returns full frames there, None elsewhere.
    """

    def __init__(self, width: int, height: int, rank: int, world_size: int, tile_w: int = 32, tile_h: int = 8):
        self.width, self.height = width, height
        self.rank, self.world_size = rank, world_size
        self.tile_w, self.tile_h = tile_w, tile_h
        self.counts = [len(region_pixels(width, height, tile_w, tile_h, world_size, r)) for r in range(world_size)]
        self.max_count = max(self.counts) if self.counts else 0
        self._index = None

    def region(self) -> rr_region:
        return rr_region(self.tile_w, self.tile_h, self.world_size, self.rank)

    def n_pixels(self) -> int:
        return self.counts[self.rank]

    def _frame_index(self, device):
        """index[y*W + x] = position of that pixel in the rank-ordered concatenation of compact buffers."""
        import torch
        if self._index is None or self._index.device != device:
            idx = np.zeros(self.width * self.height, np.int64)
            base = 0
            for r in range(self.world_size):
                xy = region_pixels(self.width, self.height, self.tile_w, self.tile_h, self.world_size, r)
                idx[xy[:, 1] * self.width + xy[:, 0]] = base + np.arange(len(xy))
                base += len(xy)
            self._index = torch.from_numpy(idx).to(device)
        return self._index

    # bytes per pixel of the buffers a rank can render, in packing order
    _SPEC = (("rgba", "uint8", 4, 4), ("normal", "float32", 3, 12), ("depth", "float32", 1, 4), ("object_id", "int32", 1, 4))

    def alloc(self, device, aux: bool, via_cpu: bool = False) -> dict:
        """Persistent buffers of this rank, allocated ONCE (outside any timed region) and reused every frame:
        `pack` = one byte buffer with a section of max_count pixels per rendered buffer (every section starts 4-byte
        aligned on every rank); the returned compact views into it are what `rr_render_region_device` fills, so the
        gather sends the pack as it is.  Rank 0 also holds the gather target (world x pack), the rank-ordered
        concatenation per buffer and the frame-order outputs."""
        import torch
        key = (str(device), bool(aux), bool(via_cpu))
        if getattr(self, "_alloc_key", None) == key:
            return self._parts
        spec = [sp for sp in self._SPEC if aux or sp[0] == "rgba"]
        pack_bytes = self.max_count * sum(sp[3] for sp in spec)
        self._pack = torch.zeros(max(pack_bytes, 4), dtype=torch.uint8, device=device)
        self._parts, self._section, off = {}, {}, 0
        n = self.n_pixels()
        for name, dt, comps, width in spec:
            self._section[name] = (off, width, getattr(torch, dt), comps)
            view = self._pack[off: off + n * width].view(getattr(torch, dt))
            self._parts[name] = view.reshape(n, comps) if comps > 1 else view.reshape(n)
            off += self.max_count * width
        self._pack_cpu = torch.zeros_like(self._pack, device="cpu").pin_memory() if (via_cpu and torch.cuda.is_available()) else (torch.zeros_like(self._pack, device="cpu") if via_cpu else None)
        self._gbuf = self._gbuf_dev = self._frame = None
        if self.world_size > 1 and self.rank == 0:
            self._gbuf = torch.zeros((self.world_size, max(pack_bytes, 4)), dtype=torch.uint8, device="cpu" if via_cpu else device)
            if via_cpu and torch.device(device).type == "cuda":
                self._gbuf_dev = torch.zeros_like(self._gbuf, device=device)   # gloo rehearsal: the gathered packs copied to the card once
        if self.rank == 0:
            self._frame = {name: torch.zeros((self.height * self.width, comps), dtype=getattr(torch, dt), device=device) for name, dt, comps, _ in spec}
        self._alloc_key = key
        return self._parts

    def gather(self, parts: dict, use_device_kernel: bool = False, via_cpu: bool = False) -> Optional[dict]:
        """ONE collective per frame: every rank sends its pack (RGBA8 and whichever aux buffers were rendered, one byte
        buffer), rank 0 gathers them (backend "nccl" = RCCL over xGMI, device tensors; `via_cpu` stages through host
        memory for "gloo", which cannot gather device tensors) and de-interleaves each buffer into frame order.
        With `parts` from `alloc()` nothing is allocated or packed here: the views ARE the pack's sections.

        ALIASING (device-kernel path): the returned tensors are reshaped VIEWS of this object's persistent frame buffers; the next
        `gather` overwrites them in place.  A caller that keeps a frame across calls -- to compare two consecutive frames, say --
        must `.clone()` it first (comparing two returned dicts without a copy always compares a buffer with itself)."""
        import torch
        import torch.distributed as dist
        keys = list(parts.keys())
        persistent = getattr(self, "_alloc_key", None) is not None and all(parts[k] is self._parts.get(k) for k in keys)
        if not persistent:   # ad-hoc tensors (tests, one-off frames): pack them into freshly allocated buffers
            dev = parts[keys[0]].device
            self.alloc(dev, aux=len(keys) > 1, via_cpu=via_cpu)
            for k in keys:
                self._parts[k].copy_(parts[k].reshape(self._parts[k].shape))
        dev = self._pack.device
        gathered, packs = None, self._pack   # world_size 1: this rank's own pack is the only one
        if self.world_size > 1:
            send = self._pack
            if via_cpu:
                self._pack_cpu.copy_(self._pack)
                send = self._pack_cpu
            gl = list(self._gbuf.unbind(0)) if self.rank == 0 else None
            dist.gather(send, gl, dst=0)
            if self.rank != 0:
                return None
            packs = self._gbuf
            if via_cpu and use_device_kernel and dev.type == "cuda":
                self._gbuf_dev.copy_(self._gbuf)
                packs = self._gbuf_dev
        out = {}
        if use_device_kernel and packs.is_cuda:
            # ONE launch for all buffers, straight from the gather target (rr_deinterleave_packed_device): no concatenation pass
            names = [sp[0] for sp in self._SPEC]
            so = [self._section[n][0] if n in keys else 0 for n in names]
            eb = [self._section[n][1] if n in keys else 0 for n in names]
            dp = [self._frame[n].data_ptr() if n in keys else 0 for n in names]
            capi.deinterleave_packed_device(self.width, self.height, self.tile_w, self.tile_h, self.world_size, packs.data_ptr(),
                                            packs.stride(0) if packs.dim() == 2 else packs.numel(), so, eb, dp,
                                            packs.device.index or 0, torch.cuda.current_stream().cuda_stream)
            for k in keys:
                out[k] = self._frame[k].reshape(self.height, self.width, -1)
            return out
        for k in keys:
            off, width, dt, comps = self._section[k]
            if self.world_size > 1:
                rows = [packs[r, off: off + self.counts[r] * width] for r in range(self.world_size)]
                cat = torch.cat(rows).view(dt).reshape(-1, comps)
            else:
                cat = self._parts[k].reshape(self.n_pixels(), comps)
            out[k] = cat.index_select(0, self._frame_index(cat.device)).reshape(self.height, self.width, -1)
        return out


def render_region_torch(device_scene: capi.DeviceScene, cam, cfg, tf: TiledFrame, aux: bool = False, sample_xy=None, via_cpu: bool = False) -> dict:
    """Render this rank's tiles on torch's current stream, straight into the sections of the rank's persistent pack
    buffer (TiledFrame.alloc): nothing is allocated per frame and `tf.gather` sends the pack as it is.

    ALIASING: the returned tensors are VIEWS of that persistent pack; the next call renders over them.  `.clone()` what must outlive
    the next frame."""
    import torch
    dev = torch.device("cuda", device_scene.device)
    parts = tf.alloc(dev, aux, via_cpu=via_cpu)
    ptrs = [parts["rgba"].data_ptr(), None, None, None]
    if aux:
        ptrs = [parts["rgba"].data_ptr(), parts["normal"].data_ptr(), parts["depth"].data_ptr(), parts["object_id"].data_ptr()]
    device_scene.render_region_device(cam, cfg, tf.region(), ptrs, torch.cuda.current_stream(dev).cuda_stream, sample_xy)
    return parts
