#!/usr/bin/env python3
"""bench.py — Mrays/s and ms/frame of the trace loop on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU).  One "step" = one whole frame of the
workload BASELINE.json quotes its metric on: sponza 1280x720, 128 spp,
monte_carlo=1.  The real Sponza .glb is downloaded by the reference at load time and is
absent offline, so the frame is `sponza_syn`, the synthetic stand-in of
rustray_amd/synthetic.py (labelled as such in `data` and `config`).

The scene is resident in HBM before the timed region; every rank renders its
interleaved tiles, rank 0 gathers the compact RGBA8 buffers over RCCL and
de-interleaves them; the timed region is bracketed by barrier +
torch.cuda.synchronize() and the MAX over ranks is taken.  A ray = one
Raytracing::trace call of the reference (primary, reflection, refraction, shadow).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel: k_trace_closest,
algorithmic bytes per SURVEY.md 8d from the instrumented oracle, launch time from
HIP events on the launch stream) and `cpu_baseline` (the C++ restatement in oracle/
timed on the host cores over a bounded sample of the same frame).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_workload(args):
    from rustray_amd import synthetic
    from rustray_amd.camera import Camera
    from rustray_amd.flat import make_config
    from tests.helpers import load_scene
    if args.scene == "sponza_syn":
        fs = synthetic.sponza_syn()
    elif args.scene == "lotus_syn":
        fs = synthetic.lotus_syn()
    elif args.scene == "helmet_syn":
        fs = synthetic.helmet_syn()
    else:
        fs = load_scene(args.scene)
    st = dict(fs.meta["camera"])
    st["width"], st["height"] = args.width, args.height
    cam = Camera.from_state(st)
    cfgd = fs.meta.get("config") or {}
    cfg = make_config(samples=args.spp, monte_carlo=bool(args.monte_carlo), seed=0, max_recursion=6,
                      focal_length=cfgd.get("focal_length", 1.0), aperture_size=cfgd.get("aperture_size", 1.0))
    return fs, cam, cfg


def usable_cpus() -> int:
    """CPUs this process may actually use: the smallest of the online count, the affinity mask and the cgroup quota
    (a GPU box shows all host CPUs but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(fs, cam, args):
    """Oracle (C++ restatement of the reference algorithm, oracle/) on the host cores, bounded sample:
    the same frame at `cpu_spp` samples per pixel.  Also yields the algorithmic-byte model."""
    from oracle import binding as ob
    from rustray_amd.flat import make_config
    threads = max(1, usable_cpus() - 2)  # num_cpus - 2, reference src/renderer.rs:67-71
    cfg = make_config(samples=args.cpu_spp, monte_carlo=bool(args.monte_carlo), seed=0, max_recursion=6)
    cs = fs.c_struct()
    t0 = time.time()
    prepared = ob.PreparedScene(cs)  # BVH build is outside the frame, as Scene::update is in the reference
    t_build = time.time() - t0
    t0 = time.time()
    out = ob.render(prepared, cam.c_struct(), cfg, n_threads=threads, want_counters=True)
    dt = time.time() - t0
    prepared.close()
    c = out["counters"]
    ab = ob.algorithmic_bytes(c, cam.width, cam.height)
    return dict(value=ab["rays"] / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample=f"same frame ({fs.name} {cam.width}x{cam.height}) at {args.cpu_spp} spp instead of {args.spp}; "
                       f"{ab['rays']} rays in {dt:.2f} s; C++ restatement of the reference algorithm, not the Rust binary",
                ms_per_frame_scaled=dt * 1000.0 * args.spp / args.cpu_spp, bvh_build_s=t_build), ab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sponza_syn")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--monte-carlo", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile", default="32x8")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rustray_amd import capi
    from rustray_amd.renderer import TiledFrame, render_region_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(n_dev, 1)   # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)

    fs, cam, cfg = build_workload(args)
    tw, th = [int(v) for v in args.tile.split("x")]
    tf = TiledFrame(args.width, args.height, rank, world, tw, th)
    ds = capi.DeviceScene(fs, local_rank)  # scene replicated on every GPU, resident before timing
    ds.set_profiling(True)
    camc = cam.c_struct()

    def step():
        parts = render_region_torch(ds, camc, cfg, tf, aux=False)
        return tf.gather(parts, use_device_kernel=True, via_cpu=(args.dist_backend == "gloo"))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    acc = dict(primary_rays=0, secondary_rays=0, shadow_rays=0, shaded_hits=0, ms_trace_closest=0.0, ms_trace_shadow=0.0,
               ms_shade=0.0, launches_trace_closest=0, launches_trace_shadow=0, launches_shade=0, ms_total=0.0)
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step()
        st = ds.stats()  # waits for this rank's frame events (inside the timed region, part of the cost)
        for k in acc:
            acc[k] += st[k]
    fence()
    elapsed = time.perf_counter() - t0
    # MAX over ranks of the wall time; SUM over ranks of the work
    rdev = "cpu" if args.dist_backend == "gloo" else "cuda"
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    work = torch.tensor([acc["primary_rays"], acc["secondary_rays"], acc["shadow_rays"], acc["shaded_hits"]], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    primary, secondary, shadow, shaded = [float(v) for v in work.tolist()]
    rays = primary + secondary + shadow

    if rank == 0:
        ms_per_step = elapsed * 1000.0 / args.steps
        result = {
            "metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{fs.name} {args.width}x{args.height} {args.spp}spp monte_carlo={args.monte_carlo} max_recursion=6 "
                                   "(synthetic stand-in for scene/sponza.json: the .glb asset is not available offline)",
                       "items": len(fs.items), "triangles": fs.n_triangles_instanced(),
                       "tiling": f"{tw}x{th} tiles interleaved over {world} rank(s), RGBA8 gather to rank 0",
                       "ray_definition": "one Raytracing::trace call: primary + reflection + refraction + shadow"},
            "ms_per_frame": ms_per_step,
            "primary_samples_per_s": primary / elapsed,
            "rays_per_frame": {"primary": primary / args.steps, "secondary": secondary / args.steps, "shadow": shadow / args.steps},
        }
        ab = None
        if not args.no_cpu_baseline and world == 1:
            cb, ab = cpu_baseline(fs, cam, args)
            result["cpu_baseline"] = cb
        elif not args.no_cpu_baseline:
            # N > 1: no CPU timing, only the byte model (per-ray figures) from a small oracle sample
            from rustray_amd.camera import Camera as _Cam
            st = dict(fs.meta["camera"]); st["width"], st["height"] = 320, 180
            small = argparse.Namespace(**vars(args)); small.cpu_spp = 1
            _, ab = cpu_baseline(fs, _Cam.from_state(st), small)
        # roofline of the dominant kernel (rank 0's launches): algorithmic bytes / launch time
        n_closest_r0 = acc["primary_rays"] + acc["secondary_rays"]
        if ab is not None and acc["launches_trace_closest"] > 0:
            bpr = ab["bytes_per_closest_ray"]
            launches = acc["launches_trace_closest"]
            avg_ms = acc["ms_trace_closest"] / launches
            bytes_per_launch = bpr * n_closest_r0 / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            import glob
            tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
            if tpaths:   # PMC passes are separate rocprofv3 runs (tools/profile_round.sh); the newest committed summary is quoted
                try:
                    traffic = json.load(open(tpaths[-1])).get("k_trace_closest_bytes_per_launch")
                except Exception:  # noqa: BLE001
                    traffic = None
            result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                  "kernel": "k_trace_closest", "launches": launches, "avg_launch_ms": avg_ms,
                                  "algorithmic_bytes_per_ray": bpr, "rays_per_launch": n_closest_r0 / launches,
                                  "whole_frame_bytes_per_ray": ab["bytes_per_ray"],
                                  "whole_frame_achieved_gbs": ab["bytes_per_ray"] * rays / elapsed / 1e9}
        result["kernel_ms_per_frame"] = {"k_trace_closest": acc["ms_trace_closest"] / args.steps,
                                         "k_trace_shadow": acc["ms_trace_shadow"] / args.steps,
                                         "k_shade": acc["ms_shade"] / args.steps,
                                         "frame_device_ms": acc["ms_total"] / args.steps}
        if frame is not None:
            result["frame_checksum"] = int(frame["rgba"].to(torch.int64).sum().item())
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ds.close()


if __name__ == "__main__":
    main()
