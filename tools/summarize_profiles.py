"""Turns gpurun_out/<tag>/ (made by tools/profile_round.sh on the GPU box) into the committed summaries under profiles/:
    <name>_bench_sponza_syn_n1.json   the bench line of the run
    <name>_kernel_stats_sponza_syn.csv   rocprofv3 --kernel-trace --stats
    <name>_hbm_traffic.json           FETCH_SIZE / WRITE_SIZE per kernel (separate passes, gfx950 correction applied)
    <name>_sq_counters.json           SQ instruction / wait counters per kernel + the derived issue fractions bench.py quotes
    <name>_tcp_counters.json          vector-L1 (TCP) and L2 (TCC) counters per kernel
usage: python tools/summarize_profiles.py <tag> <name> [build note] [scene]   e.g. r02c r02; scene (default sponza_syn) only names the files"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
scene = sys.argv[4] if len(sys.argv) > 4 else "sponza_syn"
sfx = "" if scene == "sponza_syn" else "_" + scene   # the headline workload keeps the names bench.py looks for
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
KERNELS = ("k_trace_closest", "k_trace_closest<true>", "k_trace_closest<false>", "k_stream_prepare", "k_stream_walk", "k_trace_shadow", "k_trace_shadow<true>", "k_trace_shadow<false>", "k_shade", "k_shade<true>", "k_shade<false>", "k_resolve")


def kname(raw):
    """'void k_shade<true>(DSceneView, ...)' -> 'k_shade<true>'"""
    n = raw.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


def with_totals(d, combine):
    """adds 'k_x' = combine over 'k_x<true>' and 'k_x<false>' (level 1 and the deeper levels of one kernel)"""
    for base in ("k_trace_closest", "k_trace_shadow", "k_shade"):
        parts = [d[k] for k in (base + "<true>", base + "<false>") if k in d]
        if parts and base not in d:
            d[base] = combine(parts)
    return d

N_SIMD, CLOCK_GHZ = 1024, 2.4


def newest(pattern):
    c = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(c, key=os.path.getmtime) if c else None  # gpurun merges into older local copies


def counters(sub):
    """{kernel: {counter: [launches, sum]}} of one PMC pass."""
    f = newest(os.path.join(sub, "**", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if f:
        for r in csv.DictReader(open(f)):
            a = agg[kname(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])

    def combine(parts):
        out = collections.defaultdict(lambda: [0, 0.0])
        for p in parts:
            for c, (n, v) in p.items():
                out[c][0] += n
                out[c][1] += v
        return out
    return with_totals(agg, combine)


bench = None
for cand in ("bench_n1.json", "stats_bench.json"):
    p = os.path.join(src, cand)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        bench = json.loads(open(p).read().strip().splitlines()[-1])
        if cand == "bench_n1.json":
            shutil.copy(p, f"profiles/{name}_bench_{scene}_n1.json")
        break
workload = bench["config"]["workload"] if bench else "?"
rays_closest = (bench["rays_per_frame"]["primary"] + bench["rays_per_frame"]["secondary"]) if bench else None
rays_shadow = bench["rays_per_frame"]["shadow"] if bench else None

stats = newest(os.path.join("stats", "**", "*_kernel_stats.csv"))
kernel_ns = {}
if stats:
    shutil.copy(stats, f"profiles/{name}_kernel_stats_{scene}.csv")
    for r in csv.DictReader(open(stats)):
        kernel_ns[kname(r["Name"])] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]))
    with_totals(kernel_ns, lambda parts: dict(calls=sum(p["calls"] for p in parts), total_ns=sum(p["total_ns"] for p in parts),
                                              avg_ns=sum(p["total_ns"] for p in parts) / max(sum(p["calls"] for p in parts), 1)))

# ---- HBM traffic
fetch, write = counters("pmc_fetch"), counters("pmc_write")
if fetch and write:
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras (two separate passes)",
           "workload": workload + ", one frame", "unit": "bytes",
           "correction": "FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE is used as read. "
                         "The ray-queue reads of the trace kernels are 16-B/lane streams (the x2 case); BVH node / triangle gathers mostly hit L2 / Infinity Cache and are uncalibrated.",
           "kernels": {}}
    for k in KERNELS:
        if k not in fetch:
            continue
        n = fetch[k]["FETCH_SIZE"][0]
        fr, wr = fetch[k]["FETCH_SIZE"][1] * 1024, write[k]["WRITE_SIZE"][1] * 1024
        out["kernels"][k] = {"launches_per_frame": n, "FETCH_SIZE_raw_bytes_per_frame": fr, "WRITE_SIZE_bytes_per_frame": wr,
                             "hbm_bytes_per_launch_corrected": (2 * fr + wr) / max(n, 1)}
    out["k_trace_closest_bytes_per_launch"] = out["kernels"]["k_trace_closest"]["hbm_bytes_per_launch_corrected"]
    json.dump(out, open(f"profiles/{name}_hbm_traffic{sfx}.json", "w"), indent=1)

# ---- SQ counters
sq = collections.defaultdict(dict)
for sub in ("pmc_sq1", "pmc_sq2"):
    for k, d in counters(sub).items():
        for c, (n, v) in d.items():
            sq[k][c] = v
            sq[k]["launches"] = n
if sq:
    out = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras (one run per counter set, tools/profile_round.sh)",
           "workload": workload + ", one frame; counters summed over the launches of the frame",
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count wave-instructions; SQ_BUSY_CYCLES is summed over the shader engines",
           "valu_issue_peak": f"{N_SIMD} SIMD-32 x {CLOCK_GHZ} GHz / 2 cycles per wave64 VALU instruction = {N_SIMD * CLOCK_GHZ / 2:.1f} G wave-inst/s",
           "source_id": (bench or {}).get("source_id"),   # rustray_amd.capi.source_id() of the build that was profiled: bench.py quotes valu_insts_per_ray only for the same sources
           "build_note": sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "the shipped build (rustray_amd/csrc/Makefile flags)",
           "frame_checksum": (bench or {}).get("frame_checksum"),
           "kernels": {}}
    for k in KERNELS:
        if k not in sq:
            continue
        d = dict(sq[k])
        rays = {"k_trace_closest": rays_closest, "k_trace_shadow": rays_shadow,
                "k_trace_closest<true>": bench["rays_per_frame"]["primary"] if bench else None,
                "k_trace_closest<false>": bench["rays_per_frame"]["secondary"] if bench else None,
                "k_trace_shadow<true>": (bench or {}).get("rays_per_frame", {}).get("shadow_level1"),
                "k_trace_shadow<false>": (bench or {}).get("rays_per_frame", {}).get("shadow_deeper")}.get(k)
        if rays and "SQ_INSTS_VALU" in d:
            d["rays_per_frame"] = rays
            d["valu_insts_per_ray"] = d["SQ_INSTS_VALU"] / rays
            d["vmem_insts_per_ray"] = d.get("SQ_INSTS_VMEM", 0.0) / rays
        if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
            d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
            d["wait_inst_any_frac"] = d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
            d["active_inst_any_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if k in kernel_ns and "SQ_INSTS_VALU" in d:
            # the stats run renders 4 frames (1 warm-up + 3 timed); the counter runs render one
            per_frame_ns = kernel_ns[k]["total_ns"] / 4.0
            d["kernel_ms_per_frame_unprofiled"] = per_frame_ns / 1e6
            d["valu_issue_frac"] = d["SQ_INSTS_VALU"] / (per_frame_ns * 1e-9) / (N_SIMD * CLOCK_GHZ * 1e9 / 2)
        if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"]:
            d["avg_active_lanes_per_valu"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]  # of 64
        out["kernels"][k] = d
    json.dump(out, open(f"profiles/{name}_sq_counters{sfx}.json", "w"), indent=1)

# ---- TCP / TCC / GRBM counters
tc = collections.defaultdict(dict)
for sub in ("pmc_tcp1", "pmc_tcp2", "pmc_tcc", "pmc_grbm"):
    for k, d in counters(sub).items():
        for c, (n, v) in d.items():
            tc[k][c] = v
            tc[k]["launches"] = n
if tc:
    out = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras (one run per counter set)",
           "workload": workload + ", one frame", "kernels": {}}
    for k in KERNELS:
        if k not in tc:
            continue
        d = dict(tc[k])
        if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if d.get("TCP_GATE_EN1_sum") and d.get("TCP_PENDING_STALL_CYCLES_sum") is not None:
            d["tcp_pending_stall_frac"] = d["TCP_PENDING_STALL_CYCLES_sum"] / d["TCP_GATE_EN1_sum"]
        if d.get("GRBM_GUI_ACTIVE") and k in kernel_ns:
            d["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (kernel_ns[k]["total_ns"] / 4.0)
        out["kernels"][k] = d
    json.dump(out, open(f"profiles/{name}_tcp_counters{sfx}.json", "w"), indent=1)
print("wrote", sorted(glob.glob(f"profiles/{name}_*")))
