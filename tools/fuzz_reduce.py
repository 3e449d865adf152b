#!/usr/bin/env python3
"""Developer tool: shrink a scene on which tools/fuzz_parity.py found a HIP / oracle mismatch (delta debugging over config,
items and lights) and print what is left.  usage (GPU box): MODE=rich|far|basic|farbasic SEED=n python tools/fuzz_reduce.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODE = os.environ.get("MODE", "rich"); seed = int(os.environ.get("SEED", "0"))
import torch  # noqa
import numpy as np
from oracle import binding as ob
from rustray_amd import capi
from rustray_amd.flat import make_config
from tests.helpers import camera_for, compare_frames
from tools.fuzz_parity import case, depth_ok
fs, w, h, kw, brute, scale = case(seed, MODE)
print("scale", scale, "eye", fs.meta["camera"]["eye_pos"])


def bad(f, kw):
    cam = camera_for(f, w, h).c_struct(); cfg = make_config(**kw)
    with capi.DeviceScene(f, 0) as ds:
        out = ds.render(cam, cfg); st = ds.stats()
    ref = ob.render(f.c_struct(), cam, cfg, n_threads=8, want_counters=True, brute_force=brute)
    r = compare_frames(out, ref); c = ref["counters"]
    n = r["n_rgb_over"] + r["n_id_diff"] + r["nan_mismatch"] + (0 if depth_ok(out["depth"], ref["depth"]) else 1)
    out["st"] = st; ref["c"] = c
    return n, out, ref


n, out, ref = bad(fs, kw)
print("start", kw, "items", len(fs.items), "lights", len(fs.lights), "bad", n)
for trial in (dict(samples=1), dict(monte_carlo=False), dict(aperture_size=1.0), dict(fog_density=0.0), dict(gamma_correction=False), dict(max_recursion=1), dict(max_recursion=2)):
    k2 = dict(kw); k2.update(trial)
    if bad(fs, k2)[0] > 0: kw = k2
i = 0
while i < len(fs.items):
    f2 = copy.copy(fs); f2.items = fs.items[:i] + fs.items[i + 1:]
    if bad(f2, kw)[0] > 0: fs = f2
    else: i += 1
i = 0
while i < len(fs.lights):
    f2 = copy.copy(fs); f2.lights = fs.lights[:i] + fs.lights[i + 1:]
    if bad(f2, kw)[0] > 0: fs = f2
    else: i += 1
n, out, ref = bad(fs, kw)
print("reduced: items", len(fs.items), "lights", len(fs.lights), kw, "bad", n, {k: out["st"][k] for k in ("primary_rays", "secondary_rays", "shaded_hits", "shadow_rays")},
      {k: ref["c"][k] for k in ("rays_primary", "rays_secondary", "shaded_hits", "rays_shadow")})
for it in fs.items:
    m = fs.materials[it.material]
    print(" item kind", it.kind, "id", it.id, "radius", round(float(it.radius), 4), "tris", (len(fs.meshes[it.mesh].indices) if it.kind == 1 else 0), "visible", it.visible, "flip", it.flip_normals, "tex", m.texture, "alpha", m.alpha, "refl", m.reflectivity,
          "ior", m.refraction_index, "rough", m.roughness, "bf", m.backface_cullig, "cast", m.cast_shadow, "recv", m.receive_shadow, "refl_only", m.reflection_only, "smooth", m.smooth_shading)
    print("   trans", np.asarray(it.trans).tolist())
for l in fs.lights: print(" light", l)
d = np.abs(out["rgba"][..., :3].astype(int) - ref["rgba"][..., :3].astype(int)).max(axis=-1) + (out["object_id"] != ref["object_id"])
ys, xs = np.nonzero(d > 1)
for y, x in list(zip(ys, xs))[:5]:
    print("  px", x, y, "hip", out["rgba"][y, x, :3], "oracle", ref["rgba"][y, x, :3], "depth", out["depth"][y, x], ref["depth"][y, x], "id", out["object_id"][y, x], ref["object_id"][y, x], "normal", out["normal"][y, x], ref["normal"][y, x])
fs.meta["kw"] = kw; fs.meta["wh"] = [w, h]; fs.meta["brute"] = brute
fs.save(f"gpurun_out/reduced_{MODE}_{seed}.npz")

# the reduced case as a fixture (the format of tests/golden/fuzz_*.npz), and the rays of its first differing pixel replayed one by one
fs.meta["wh"] = [w, h]; fs.meta["kw"] = kw; fs.meta["brute"] = brute
os.makedirs("gpurun_out", exist_ok=True)
path = f"gpurun_out/reduced_{MODE}_{seed}.npz"
fs.save(path)
d = (np.abs(out["rgba"].astype(int) - ref["rgba"].astype(int)).max(axis=-1) > 1) | (out["object_id"] != ref["object_id"])
print("saved", path, "differing pixels", int(d.sum()))
if d.any():
    py, px = [int(v[0]) for v in np.nonzero(d)]
    cam = camera_for(fs, w, h).c_struct(); cfg = make_config(**kw)
    with ob.ray_log() as log:
        one = ob.render(fs.c_struct(), cam, cfg, window=(px, py, px + 1, py + 1), n_threads=1, brute_force=brute)
        L = log.rays()
    print("pixel", (px, py), "hip", out["rgba"][py, px], out["object_id"][py, px], "oracle", ref["rgba"][py, px], ref["object_id"][py, px], "oracle rays", len(L["toi"]))
    cl = ~L["for_shadow"]
    with capi.DeviceScene(fs, 0) as ds:
        for dep in sorted(set(L["depth"][cl].tolist())):
            m = cl & (L["depth"] == dep)
            g = ds.trace_rays(L["origin"][m], L["dir"][m], int(dep))
            for j, i in enumerate(np.nonzero(m)[0][:64]):
                same = g[0][j] == L["found"][i] and (not g[0][j] or (g[1][j] == L["item"][i] and g[2][j] == L["face"][i] and g[3][j].view(np.uint32) == L["toi"][i].view(np.uint32)))
                if not same or len(L["toi"]) <= 16:
                    print(" depth", dep, "ray", i, L["origin"][i], L["dir"][i], "oracle", L["found"][i], L["item"][i], L["face"][i], L["toi"][i], "hip", g[0][j], g[1][j], g[2][j], g[3][j], "" if same else "  <<< differs")
