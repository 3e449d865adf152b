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

The timed frame is the CONTRACT frame: RGBA8 plus the aux buffers PixelData carries
(normal, depth, object id: reference src/raytracing.rs:57-70), resident in HBM at the end.
Extra keys (rank 0, N = 1): `rgba_only_ms` (aux buffers not requested), `rr_render_host_ms`
(the same frame through rr_render into host memory, PCIe included), `update_transforms_ms` / `pick_ms` (the animation step and
the pick query on the resident scene), and `other_configs`: every other BASELINE config at its own size and sample count
(C1 spheres 256x256x1 mc=0, C2 monkey 800x600x16, and the stand-ins helmet_syn 1280x720x64, lotus_syn 1280x720x512 + DOF for
C3 / C5), each with its device time, launches and -- where profiles/ holds SQ counters of the same build for it -- its own `roofline`.
`roofline.kernels` prices EVERY kernel build of the frame (k_trace_closest / k_shade / k_trace_shadow, level 1 `<true>` and the deeper
levels `<false>`) from its live launch times.

N > 1 additionally reports what shows that RCCL really saw N ranks (`dist_backend`, `world_size` as torch.distributed
reports it, every rank's device), `frame_checksum_matches_single_gpu` (rank 0 renders the whole frame alone after the
timed region and compares), `gather_ms` / `render_only_ms` (timed apart, after the timed region), and `one_process`:
the same frame through rr_render_multi over the same N devices from ONE host process (the form the reference host is,
src/renderer.rs:105-172), run by a helper process that rank 0 starts before it touches the GPU.  `--one-process` runs
that form alone.

Rank 0 prints ONE JSON line with `roofline` and `cpu_baseline` (the C++ restatement in
oracle/ timed on the host cores over a bounded sample of the same frame).

`roofline`: the dominant kernel is the level-1 build of the closest-hit kernel,
`k_trace_closest<true>` (8.7 of the frame's 25.4 ms; its row in the rocprofv3 kernel stats
under profiles/ carries the same average duration).  It walks a BVH that lives in L2 / Infinity
Cache (130 MB), so HBM is NOT what bounds it (measured: 3 % of HBM peak); what it runs out
of is vector-instruction issue slots and the latency that keeps them empty.  `bound` is
therefore "valu_issue": achieved = VALU wave-instructions per ray (SQ_INSTS_VALU of the
committed profiles/r*_sq_counters.json, a property of code + scene) x rays per launch /
launch duration measured LIVE with HIP events on the launch stream; peak = 1024 SIMDs x
2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md: SIMD-32).  The SURVEY 8d
algorithmic bytes and the PMC-measured HBM bytes are kept beside it as `algorithmic_gbs`,
`traffic` and `hbm_measured_frac`, never as `frac`.
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
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # 1024 SIMD-32 x 2.4 GHz / 2 issue cycles per wave64 VALU instruction = 1228.8 G wave-inst/s


STAND_INS = {"sponza": "sponza_syn", "helmet": "helmet_syn", "lotus": "lotus_syn"}


def build_workload(scene, width, height, spp, monte_carlo=1, scene_root=None):
    """`scene`: a stand-in name (sponza_syn, helmet_syn, lotus_syn), a fixture under scenes/, or one of the reference's scene files
    (e.g. scene/sponza.json) resolved against `scene_root` (a tree laid out like the reference's: scene/*.json, data/...).
    A scene file is loaded for REAL when every asset it names is present there (the .glb of sponza / helmet / lotus under
    data/temp/, which the reference downloads at load time: SURVEY.md 8d (i)); nothing is ever downloaded here.  When an asset
    is missing the stand-in of the same name is rendered and labelled so.  fs.meta["data"] says which it was."""
    from rustray_amd import synthetic
    from rustray_amd.camera import Camera
    from rustray_amd.flat import make_config
    from tests.helpers import load_scene
    fs = None
    if scene.endswith(".json"):
        from rustray_amd import scene as scene_mod
        root = scene_root or os.environ.get("RUSTRAY_SCENE_ROOT") or os.getcwd()
        why = None
        try:
            sc = scene_mod.load_scene([scene], width, height, root=root)
            fs = sc.flatten()
            fs.name = os.path.splitext(os.path.basename(scene))[0]
            fs.meta = {"camera": sc.cam.state(), "config": sc.raytracing_config, "data": "real", "source": os.path.join(root, scene)}
        except Exception as e:  # noqa: BLE001  (a missing asset, or a file the loader mirror cannot read)
            why = f"{type(e).__name__}: {e}"
        if fs is None:
            base = os.path.splitext(os.path.basename(scene))[0]
            scene = STAND_INS.get(base)
            if scene is None:
                raise SystemExit(f"{base}: cannot load the scene file ({why}) and there is no stand-in for it")
            sys.stderr.write(f"bench.py: {base}: {why}; rendering the stand-in {scene}\n")
    if fs is None:
        if scene == "sponza_syn":
            fs = synthetic.sponza_syn()
        elif scene == "lotus_syn":
            fs = synthetic.lotus_syn()
        elif scene == "helmet_syn":
            fs = synthetic.helmet_syn()
        else:
            fs = load_scene(scene)
            fs.name = scene   # (the fixture files carry the reference scene files' own titles)
        fs.meta = dict(fs.meta)
        fs.meta["data"] = "synthetic" if fs.meta.get("synthetic") else "fixture"
    st = dict(fs.meta["camera"])
    st["width"], st["height"] = width, height
    cam = Camera.from_state(st)
    cfgd = fs.meta.get("config") or {}
    cfg = make_config(samples=spp, monte_carlo=bool(monte_carlo), seed=0, max_recursion=6,
                      focal_length=cfgd.get("focal_length", 1.0), aperture_size=cfgd.get("aperture_size", 1.0))
    return fs, cam, cfg


def newest_profile(pattern):
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not paths:
        return None, None
    try:
        return json.load(open(paths[-1])), os.path.relpath(paths[-1], ROOT)
    except Exception:  # noqa: BLE001
        return None, None


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


KERNEL_BUILDS = (  # kernel build -> (its ms / launches fields in rr_frame_stats)
    ("k_trace_closest<true>", "ms_trace_closest_level1", "launches_trace_closest_level1"),
    ("k_shade<true>", "ms_shade_level1", "launches_shade_level1"),
    ("k_trace_shadow<true>", "ms_trace_shadow_level1", "launches_trace_shadow_level1"),
    ("k_trace_closest<false>", None, None), ("k_shade<false>", None, None), ("k_trace_shadow<false>", None, None))


def kernel_rooflines(scene_name, width, height, spp, acc, frames, source_id):
    """`roofline` of one workload from the LIVE per-launch HIP-event times in `acc` (rr_frame_stats summed over `frames` frames) and the
    committed SQ counters of the same build and workload (profiles/rNN_sq_counters[_<scene>].json, tools/profile_round.sh): for every
    kernel build, achieved = VALU wave-instructions per frame (a property of code + scene) / live kernel time per frame, against the
    VALU issue peak.  The headline entry is the level-1 closest-hit build; `kernels` has all of them.  None when nothing was timed."""
    if acc.get("launches_trace_closest_level1", 0) <= 0:
        return None
    launches = acc["launches_trace_closest_level1"]
    roof = {"bound": "valu_issue", "peak": VALU_PEAK_GINST, "unit": "Ginst/s", "kernel": "k_trace_closest<true>", "launches": launches,
            "avg_launch_ms": acc["ms_trace_closest_level1"] / launches, "rays_per_launch": acc["primary_rays"] / launches,
            "achieved": None, "frac": None, "traffic": None}
    sfx = "" if scene_name == "sponza_syn" else "_" + scene_name
    sq, sq_path = newest_profile(f"r[0-9][0-9]_sq_counters{sfx}.json")   # the round's final profile (rNNa / rNNb are experiments)
    same_workload = bool(sq) and sq.get("workload", "").startswith(f"{scene_name} {width}x{height} {spp}spp")
    same_build = bool(sq) and sq.get("source_id") == source_id
    if sq and not (same_workload and same_build):
        # the instruction count is a property of code + scene: a count taken from another build or workload is not mixed with this run's time
        roof["frac_reason"] = (f"{sq_path} was collected on source_id {sq.get('source_id')} / workload {sq.get('workload', '')[:40]!r}; "
                               f"this run is source_id {source_id}: re-run tools/profile_round.sh")
    elif not sq:
        roof["frac_reason"] = f"no profiles/rNN_sq_counters{sfx}.json for this workload"
    ms_frame = {"k_trace_closest<true>": acc["ms_trace_closest_level1"], "k_shade<true>": acc.get("ms_shade_level1", 0.0), "k_trace_shadow<true>": acc.get("ms_trace_shadow_level1", 0.0),
                "k_trace_closest<false>": acc["ms_trace_closest"] - acc["ms_trace_closest_level1"], "k_shade<false>": acc["ms_shade"] - acc.get("ms_shade_level1", 0.0),
                "k_trace_shadow<false>": acc["ms_trace_shadow"] - acc.get("ms_trace_shadow_level1", 0.0)}
    n_launch = {"k_trace_closest<true>": launches, "k_shade<true>": acc.get("launches_shade_level1", 0), "k_trace_shadow<true>": acc.get("launches_trace_shadow_level1", 0),
                "k_trace_closest<false>": acc["launches_trace_closest"] - launches, "k_shade<false>": acc["launches_shade"] - acc.get("launches_shade_level1", 0),
                "k_trace_shadow<false>": acc["launches_trace_shadow"] - acc.get("launches_trace_shadow_level1", 0)}
    kernels = {}
    for name, _, _ in KERNEL_BUILDS:
        if n_launch[name] <= 0:
            continue
        ms = ms_frame[name] / frames
        k = {"launches_per_frame": n_launch[name] / frames, "ms_per_frame": ms, "avg_launch_ms": ms_frame[name] / n_launch[name], "frac": None}
        c = (sq or {}).get("kernels", {}).get(name) if (same_workload and same_build) else None
        if c and c.get("SQ_INSTS_VALU") and ms > 0:
            k["valu_wave_insts_per_frame"] = c["SQ_INSTS_VALU"]
            k["achieved"] = c["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
            k["frac"] = k["achieved"] / VALU_PEAK_GINST
            k["wait_any_frac"] = c.get("wait_any_frac")            # share of wave cycles parked in s_waitcnt
            k["wait_inst_any_frac"] = c.get("wait_inst_any_frac")  # share stalled at issue
            k["avg_active_lanes_per_valu"] = c.get("avg_active_lanes_per_valu")
        kernels[name] = k
    k1 = kernels.get("k_trace_closest<true>", {})
    if k1.get("frac") is not None:
        roof["achieved"], roof["frac"] = k1["achieved"], k1["frac"]
        roof["valu_wave_insts_per_ray"] = k1["valu_wave_insts_per_frame"] / (acc["primary_rays"] / frames)
        roof["sq_counters"] = sq_path
        roof["sq_wait_any_frac_of_wave_cycles"] = k1.get("wait_any_frac")
        roof["sq_wait_inst_any_frac_of_wave_cycles"] = k1.get("wait_inst_any_frac")
    roof["kernels"] = kernels
    all_l = acc["launches_trace_closest"]
    roof["all_levels"] = {"launches": all_l, "avg_launch_ms": acc["ms_trace_closest"] / max(all_l, 1),
                          "rays_per_launch": (acc["primary_rays"] + acc["secondary_rays"]) / max(all_l, 1)}
    return roof


def extras(args, ds, camc, cfg, step, fence):
    """Rank 0, N = 1: the same frame without aux buffers, the same frame through rr_render into host memory, and one
    frame each of the stand-ins for BASELINE configs C3 / C5."""
    from rustray_amd import capi
    out = {}

    def timed(fn, reps=2):
        fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        fence()
        return (time.perf_counter() - t0) * 1000.0 / reps
    out["rgba_only_ms"] = timed(lambda: step(aux=False))
    out["rr_render_host_ms"] = timed(lambda: ds.render(camc, cfg, aux=True))   # RGBA8 + aux over PCIe into pageable host arrays
    out["rr_render_host_rgba_only_ms"] = timed(lambda: ds.render(camc, cfg, aux=False))
    # the frame's side doors, on the resident bench scene: an animation step that moves nothing (every item's own matrices back in:
    # rr_scene_update_transforms re-derives world normals, surface boxes and the top level whatever the matrices are), and one pick
    import statistics
    from rustray_amd import capi as _capi
    t = np.stack([np.asarray(it.trans, np.float32) for it in ds._flat.items])
    ti = np.stack([np.asarray(it.trans_inv, np.float32) for it in ds._flat.items])
    ts = []
    for _ in range(6):
        fence(); t0 = time.perf_counter(); ds.update_transforms(t, ti); ts.append((time.perf_counter() - t0) * 1000.0)
    out["update_transforms_ms"] = statistics.median(ts[1:])   # host wall time of the whole call, median of 5 (the first call allocates)
    ts = []
    for k in range(6):
        t0 = time.perf_counter(); ds.pick(camc, args.width // 2 + k, args.height // 2); ts.append((time.perf_counter() - t0) * 1000.0)
    out["pick_ms"] = statistics.median(ts[1:])
    # an animation on the resident scene (the frame loop of src/run.rs:421-465: apply_frame -> restart -> wait): 8 frames, every item turned a
    # little further about y each frame (rr_scene_update_transforms), the contract frame rendered after each step
    def roty(a):
        c, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
        return np.asarray([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]], np.float32)
    n_frames = 8
    step(); fence()
    t0 = time.perf_counter()
    for f in range(1, n_frames + 1):
        r = roty(0.02 * f)
        ds.update_transforms((t @ r).astype(np.float32), (r.T @ ti).astype(np.float32))
        step()
    fence()
    dt = time.perf_counter() - t0
    ds.update_transforms(t, ti)   # back to the scene as it was
    out["animation"] = {"frames": n_frames, "frames_per_s": n_frames / dt, "ms_per_frame": dt * 1000.0 / n_frames,
                        "what": "rr_scene_update_transforms (all items turned about y) + the contract frame, per frame, on the resident scene"}
    # every other BASELINE config at its own size and sample count, one frame each (C1 spheres, C2 monkey, C3 / C5 stand-ins)
    other = {}
    for scene, w, h, spp, mc in (("spheres", 256, 256, 1, 0), ("monkey", 800, 600, 16, 1), ("helmet_syn", args.width, args.height, 64, 1), ("lotus_syn", args.width, args.height, 512, 1)):
        fs2, cam2, cfg2 = build_workload(scene, w, h, spp, mc)
        with _capi.DeviceScene(fs2, ds.device) as d2:
            d2.set_profiling(True)
            c2 = cam2.c_struct()
            ms = timed(lambda: d2.render(c2, cfg2, aux=True), reps=1)
            st = d2.stats()
        rays = st["primary_rays"] + st["secondary_rays"] + st["shadow_rays"]
        entry = {"ms_per_frame_host": ms, "ms_per_frame_device": st["ms_total"], "mrays_per_s": rays / (st["ms_total"] * 1e-3) / 1e6,
                 "rays": rays, "items": len(fs2.items), "triangles": fs2.n_triangles_instanced(),
                 "launches": st["launches_trace_closest"] + st["launches_shade"] + st["launches_trace_shadow"],
                 "kernel_ms_per_frame": {"k_trace_closest": st["ms_trace_closest"], "k_shade": st["ms_shade"], "k_trace_shadow": st["ms_trace_shadow"]}}
        r2 = kernel_rooflines(scene, w, h, spp, st, 1, _capi.source_id())
        if r2 is not None:
            entry["roofline"] = r2
        other[f"{scene} {w}x{h} {spp}spp monte_carlo={mc}" + (" +DOF" if cfg2.aperture_size > 1.0 else "")] = entry
    out["other_configs"] = other
    return out


def one_process_main(args):
    """rr_render_multi over --gpus devices from ONE host process (no torch, no torch.distributed): one handle per device,
    one host thread per device inside the call, peer-to-peer copies into device 0, one host copy.  As a helper of an N-rank
    run it prepares the scene on the host, then waits for "go" on stdin: rank 0 says it when the N ranks' run is over and
    the other ranks have left their GPUs."""
    from rustray_amd import capi
    n = args.gpus
    fs, cam, cfg = build_workload(args.scene, args.width, args.height, args.spp, args.monte_carlo, args.scene_root)   # (host work only)
    camc = cam.c_struct()
    if args.one_process_helper:
        # nothing of this process touches a GPU before "go": the N ranks' timed region sees no allocation, upload or build of ours
        line = sys.stdin.readline()
        if line.strip() != "go":
            return
    n_dev = capi.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py --one-process needs a GPU")
    devices = [0] * n if args.same_device else list(range(n))
    if max(devices) >= n_dev:
        print(json.dumps({"error": f"{n} devices asked, {n_dev} visible"}), flush=True)
        return
    scenes = [capi.DeviceScene(fs, d) for d in devices]
    try:
        for _ in range(max(args.warmup, 1)):
            capi.render_multi(scenes, camc, cfg, aux=not args.rgba_only)
        t0 = time.perf_counter()
        rays = 0
        for _ in range(args.steps):
            out = capi.render_multi(scenes, camc, cfg, aux=not args.rgba_only)
            for ds in scenes:
                st = ds.stats()
                rays += st["primary_rays"] + st["secondary_rays"] + st["shadow_rays"]
        elapsed = time.perf_counter() - t0
        st0 = scenes[0].stats()
        res = {"metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": max(args.warmup, 1),
               "ms_per_step": elapsed * 1000.0 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": {"workload": f"{fs.name} {args.width}x{args.height} {args.spp}spp monte_carlo={args.monte_carlo} max_recursion=6",
                                               "form": "rr_render_multi: ONE host process, one handle and one host thread per device, frame into host memory (PCIe included)",
                                               "devices": devices},
               "frame_checksum": int(out["rgba"].astype(np.int64).sum()),
               "multi_devices": st0["multi_devices"], "multi_peer_links": st0["multi_peer_links"], "multi_staged_links": st0["multi_staged_links"],
               "ms_multi_exchange": st0["ms_multi_exchange"], "source_id": capi.source_id()}
        print(json.dumps(res), flush=True)
    finally:
        for ds in scenes:
            ds.close()


def start_one_process_helper(args, world):
    """Started by rank 0 BEFORE it initialises the GPU (a process that has may not exec another program)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--one-process-helper", "--gpus", str(world), "--steps", str(max(1, min(args.steps, 5))),
           "--warmup", "1", "--scene", args.scene, *(["--scene-root", args.scene_root] if args.scene_root else []), "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp),
           "--monte-carlo", str(args.monte_carlo)]
    if args.rgba_only:
        cmd.append("--rgba-only")
    if args.dist_backend == "gloo":
        cmd.append("--same-device")   # rehearsal: the ranks share a card
    try:
        return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except OSError as e:
        return e


def wait_for_ranks_to_leave(pids, limit_s=20.0):
    """Rank 0, after destroy_process_group: waits until the other ranks' processes are gone (they exit right after the barrier; a rank
    still alive holds its GPU context and would share the device with the one-process leg).  Bounded; says what happened."""
    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        try:   # a zombie (exited, not yet reaped by the launcher) has left its GPU
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False
    t0 = time.perf_counter()
    left = [p for p in pids if alive(p)]
    while left and time.perf_counter() - t0 < limit_s:
        time.sleep(0.05)
        left = [p for p in left if alive(p)]
    return {"waited_s": time.perf_counter() - t0, "ranks_gone": not left, "still_alive": left}


def finish_one_process_helper(helper, timeout=300):
    """Rank 0, after the collective is torn down (the other ranks are leaving): let the helper build its handles and run its frames, return its line."""
    import subprocess
    if not hasattr(helper, "communicate"):
        return {"error": f"helper did not start: {helper}"}
    try:
        out, err = helper.communicate("go\n", timeout=timeout)
    except subprocess.TimeoutExpired:
        helper.kill()   # this exact child, by handle
        helper.communicate()
        return {"error": f"rr_render_multi helper did not finish within {timeout} s"}
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": "no result line", "stderr_tail": err[-400:]}
    try:
        return json.loads(lines[-1])
    except ValueError:
        return {"error": "unparsable result line", "stdout_tail": out[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sponza_syn", help="stand-in or fixture name, or a reference scene file such as scene/sponza.json (see --scene-root)")
    ap.add_argument("--scene-root", default=None, help="tree laid out like the reference's (scene/*.json, data/temp/*.glb): a scene file is rendered for real when "
                                                       "its assets are present there, else its stand-in; nothing is downloaded")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--monte-carlo", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip rgba_only / rr_render_host / other_configs (profiling runs)")
    ap.add_argument("--rgba-only", action="store_true", help="developer A/B: time the frame without the aux buffers")
    ap.add_argument("--binning", action="store_true", help="developer A/B: bin deeper levels by (origin cell, direction octant) before tracing")
    ap.add_argument("--tile", default="32x8")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--one-process", action="store_true", help="time rr_render_multi over --gpus devices from this ONE process (no torch.distributed)")
    ap.add_argument("--same-device", action="store_true", help="--one-process rehearsal: all handles on device 0")
    ap.add_argument("--no-one-process", action="store_true", help="N > 1: skip the rr_render_multi leg")
    ap.add_argument("--one-process-helper", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.one_process or args.one_process_helper:
        return one_process_main(args)

    import torch
    import torch.distributed as dist
    from rustray_amd import capi
    from rustray_amd.renderer import TiledFrame, render_region_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the one-process leg runs in a helper that must exist BEFORE this process touches the GPU (a GPU process may not exec)
    helper = start_one_process_helper(args, world) if (world > 1 and rank == 0 and not args.no_one_process) else None
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

    fs, cam, cfg = build_workload(args.scene, args.width, args.height, args.spp, args.monte_carlo, args.scene_root)
    tw, th = [int(v) for v in args.tile.split("x")]
    tf = TiledFrame(args.width, args.height, rank, world, tw, th)
    ds = capi.DeviceScene(fs, local_rank)  # scene replicated on every GPU, resident before timing
    ds.set_profiling(True)
    if args.binning:
        ds.set_tuning(bin_min_rays=1 << 18)
    camc = cam.c_struct()
    via_cpu = args.dist_backend == "gloo"

    def step(aux=not args.rgba_only):
        parts = render_region_torch(ds, camc, cfg, tf, aux=aux, via_cpu=via_cpu)   # into the rank's persistent pack buffer
        return tf.gather(parts, use_device_kernel=True, via_cpu=via_cpu)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    acc = dict(primary_rays=0, secondary_rays=0, shadow_rays=0, shaded_hits=0, ms_trace_closest=0.0, ms_trace_shadow=0.0,
               ms_shade=0.0, launches_trace_closest=0, launches_trace_shadow=0, launches_shade=0, ms_total=0.0, ms_binning=0.0, binned_rays=0,
               ms_trace_closest_level1=0.0, launches_trace_closest_level1=0, ms_shade_level1=0.0, launches_shade_level1=0,
               ms_trace_shadow_level1=0.0, launches_trace_shadow_level1=0)
    t0 = time.perf_counter()
    frame = None
    for it in range(args.steps):
        frame = step()
        if world == 1:
            st = ds.stats()  # waits for this rank's frame events (inside the timed region, part of the cost): per-launch times for the roofline
            for k in acc:
                acc[k] += st[k]
        elif it == args.steps - 1:
            # N > 1: every frame is the same frame (same rays, bit for bit); the counters and launch times of the LAST one stand for all of
            # them, so that the timed loop holds nothing but render + gather
            st = ds.stats()
            for k in acc:
                acc[k] = st[k] * args.steps
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

    # ---- N > 1, outside the timed region: what shows the collective really ran over N ranks, and the gather timed apart
    dist_info = None
    if world > 1:
        me = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(),
              "device": torch.cuda.get_device_name(torch.cuda.current_device()), "pid": os.getpid(), "region_pixels": tf.n_pixels()}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        t_render = t_gather = 0.0
        reps = 3
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            parts = render_region_torch(ds, camc, cfg, tf, aux=not args.rgba_only, via_cpu=via_cpu)
            fence()
            t1 = time.perf_counter()
            tf.gather(parts, use_device_kernel=True, via_cpu=via_cpu)
            fence()
            t_render += t1 - t0
            t_gather += time.perf_counter() - t1
        tt = torch.tensor([t_render, t_gather], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist_info = {"dist_backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": everyone,
                     "render_only_ms": float(tt[0]) * 1000.0 / reps, "gather_ms": float(tt[1]) * 1000.0 / reps,
                     "gather_bytes_per_rank": int(tf._pack.numel())}
        if rank == 0:   # the same frame on this rank's GPU alone: the tiled frame must be bit-identical to it
            whole = ds.render(camc, cfg, aux=False)
            dist_info["frame_checksum_single_gpu"] = int(whole["rgba"].astype(np.int64).sum())
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed * 1000.0 / args.steps
        result = {
            "metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": fs.meta.get("data", "synthetic"),
            "config": {"workload": f"{fs.name} {args.width}x{args.height} {args.spp}spp monte_carlo={args.monte_carlo} max_recursion=6 "
                                   + ("(the reference's own scene file and assets)" if fs.meta.get("data") == "real" else
                                      "(synthetic stand-in for scene/sponza.json: the .glb asset is not available offline)" if fs.name == "sponza_syn" else f"({fs.meta.get('data')})"),
                       "outputs": "RGBA8 only (--rgba-only)" if args.rgba_only else "RGBA8 + normal + depth + object_id (PixelData), resident in HBM",
                       "items": len(fs.items), "triangles": fs.n_triangles_instanced(),
                       "tiling": f"{tw}x{th} tiles interleaved over {world} rank(s), one packed gather (RGBA8 + aux) to rank 0",
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
        # roofline of the dominant kernel (rank 0's launches), see the module docstring: the level-1 build of the closest-hit
        # kernel, `k_trace_closest<true>` in the rocprofv3 kernel stats (one launch per batch, the frame is one batch) -- and, beside
        # it under `kernels`, every kernel build of the frame with its own live launch time
        roof = kernel_rooflines(fs.name, args.width, args.height, args.spp, acc, args.steps, capi.source_id())
        if roof is not None:
            hbm, hbm_path = newest_profile("r[0-9][0-9]_hbm_traffic.json")
            h1 = (hbm or {}).get("kernels", {}).get("k_trace_closest<true>")
            if h1 and roof.get("avg_launch_ms"):
                roof["traffic"] = h1["hbm_bytes_per_launch_corrected"]
                roof["traffic_source"] = hbm_path
                roof["hbm_measured_frac"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if ab is not None and roof.get("avg_launch_ms"):
                bpr = ab["bytes_per_closest_ray"]
                roof["algorithmic_bytes_per_ray"] = bpr
                roof["algorithmic_gbs"] = bpr * roof["rays_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9   # cache-served: may exceed HBM peak
                roof["whole_frame_bytes_per_ray"] = ab["bytes_per_ray"]
            result["roofline"] = roof
        result["kernel_ms_per_frame"] = {"k_trace_closest": acc["ms_trace_closest"] / args.steps,
                                         "k_trace_shadow": acc["ms_trace_shadow"] / args.steps,
                                         "k_shade": acc["ms_shade"] / args.steps, "k_bin_*": acc["ms_binning"] / args.steps,
                                         "frame_device_ms": acc["ms_total"] / args.steps}
        if frame is not None:
            result["frame_checksum"] = int(frame["rgba"].to(torch.int64).sum().item())
        result["source_id"] = capi.source_id()
        if dist_info is not None:
            dist_info["frame_checksum_matches_single_gpu"] = dist_info.get("frame_checksum_single_gpu") == result.get("frame_checksum")
            result.update(dist_info)
        if world == 1 and not args.no_extras:
            result.update(extras(args, ds, camc, cfg, step, fence))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ds.close()
    if rank == 0:
        if helper is not None:
            # the one-process leg runs now: the collective is torn down, this rank's handle is closed and the other ranks are
            # exiting, so rr_render_multi has the N devices to itself (a rank waiting in an RCCL barrier would spin on its GPU)
            result["one_process_wait"] = wait_for_ranks_to_leave([r["pid"] for r in (dist_info or {}).get("ranks", []) if r["pid"] != os.getpid()])
            result["one_process"] = finish_one_process_helper(helper)
        print(json.dumps(result))


if __name__ == "__main__":
    main()
