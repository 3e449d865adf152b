#!/usr/bin/env python3
"""Developer tool: cost breakdown of a kernel by ABLATION (the pool refuses rocprofv3 PC sampling).  Builds the library from a patched
copy of csrc/ in which one piece of work is removed (the frame is then WRONG: only the kernel times of tools/ab.sh are read) into
build/lib_abl_<name>.so.  usage: tools/ablate.py <name> [<name> ...] | all"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = "rr_kernels.hip"
PATCHES = {
    # k_shade
    "nojit": [(K, "    if (spread <= 0.0f) return dir;\n    f3 b3 = normalize3(dir);", "    return dir;\n    f3 b3 = normalize3(dir);")],
    "nopow": [(K, "const float light_power = powf(spec_dot, m.shininess);", "const float light_power = spec_dot;")],
    "noacc": [(K, "        accum_merged(acc, sum_pix, sum_r, sum_g, sum_b);\n        if ((acc.normal || acc.depth) && __ballot(aux_pix != 0xffffffffu) != 0ull) accum_aux_merged(acc, aux_pix, aux_nx, aux_ny, aux_nz, aux_d);",
               "        if (sum_r == 0x7fffffffffffll) acc.flags[0] = (uint32_t)(sum_g + sum_b + aux_nx + aux_ny + aux_nz + aux_d + sum_pix + aux_pix);")],
    "noaux": [(K, "        if ((acc.normal || acc.depth) && __ballot(aux_pix != 0xffffffffu) != 0ull) accum_aux_merged(acc, aux_pix, aux_nx, aux_ny, aux_nz, aux_d);",
               "        if (aux_d == 0x7fffffffffffll) acc.flags[0] = (uint32_t)(aux_nx + aux_ny + aux_nz + aux_pix);")],
    "nosq": [(K, "                sq_wrote |= 1u << lk;\n", ""),
             (K, "                sq.s0[si] = make_float4(so.x, so.y, so.z, limit);\n                sq.s1[si] = make_float4(sd.x, sd.y, sd.z, __uint_as_float((uint32_t)item_idx | (depth << 27)));\n                sq.s2[si] = make_float4(cr, cg, cb, __uint_as_float(pix));\n",
              "                if (so.x + sd.x + cr + cg + cb + limit == 12345.678f) sq.s0[si] = make_float4(so.y, so.z, sd.y, sd.z);\n")],
    "notex": [(K, "    if (!(m.flags & (RR_MF_TEX_SLOT0 << slot)) || !has_uv) return false; // slot bit = index >= 0 and width > 0", "    return false;")],
    "nochild": [(K, "        spawn_refl = reflectivity > 0.0f && may_recurse;", "        spawn_refl = false;"),
                (K, "        if (alpha < 1.0f && may_recurse) {\n            // create_transmission", "        if (alpha < -1.0f && may_recurse) {\n            // create_transmission")],
    "nolight": [(K, "        for (uint32_t li = 0; li < sc.n_lights; li++) {\n            const DLight& L = rr_global(sc.lights)[li];\n            if (L.type & 0x80u) continue; // disabled",
                 "        for (uint32_t li = 0; li < sc.n_lights; li++) {\n            const DLight& L = rr_global(sc.lights)[li];\n            if (L.type != 0x7fu) continue;")],
}


def build(name):
    d = tempfile.mkdtemp(prefix="abl_")
    try:
        os.makedirs(os.path.join(d, "rustray_amd", "csrc"))
        os.makedirs(os.path.join(d, "include"))
        for f in ("rr_api.hip", "rr_kernels.hip", "rr_bvh.cpp", "rr_bvh.h", "rr_device.h", "rr_math.h"):
            shutil.copy(os.path.join(ROOT, "rustray_amd", "csrc", f), os.path.join(d, "rustray_amd", "csrc", f))
        shutil.copy(os.path.join(ROOT, "include", "rustray_hip.h"), os.path.join(d, "include", "rustray_hip.h"))
        for part in name.split("+"):
            for f, old, new in PATCHES[part]:
                p = os.path.join(d, "rustray_amd", "csrc", f)
                s = open(p).read()
                if s.count(old) != 1:
                    raise SystemExit(f"{name}: pattern of {part} found {s.count(old)} times in {f}")
                open(p, "w").write(s.replace(old, new))
        out = os.path.join(ROOT, "build", f"lib_abl_{name}.so")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-shared", "-o", out, "rr_api.hip", "rr_bvh.cpp"],
                              cwd=os.path.join(d, "rustray_amd", "csrc"))
        print("built", out)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    names = sys.argv[1:]
    if names == ["all"]:
        names = list(PATCHES)
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    for n in names:
        build(n)
