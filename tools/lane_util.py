#!/usr/bin/env python3
"""Developer tool: executed wave-steps and their average active lanes, by step kind and by ray kind (closest-hit level 1,
closest-hit deeper levels, shadow), from a -DRR_EXP_UTIL build of the library.
usage: RUSTRAY_HIP_LIB=build/lib_util.so python tools/lane_util.py [scene] [spp] [bin]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rustray_amd import capi

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_syn"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
binning = len(sys.argv) > 3 and sys.argv[3] == "bin"
fs, cam, cfg = bench.build_workload(scene, 1280, 720, spp, 1)
lib = capi.lib()
buf = (C.c_ulonglong * 64)()
steps = {0: "top-level node step", 1: "item set-up", 2: "mesh node step", 3: "triangle test", 4: "mesh walk entry"}
kinds = ["closest-hit, level 1", "closest-hit, deeper levels", "shadow rays"]
with capi.DeviceScene(fs, 0) as ds:
    ds.set_tuning(bin_min_rays=(1 << 18) if binning else 0)
    lib.rr_exp_util.argtypes = [C.c_void_p, C.c_int]
    lib.rr_exp_util(None, 1)
    ds.render(cam.c_struct(), cfg, aux=False)
    st = ds.stats()
    lib.rr_exp_util(buf, 0)
    rays = [st["primary_rays"], st["secondary_rays"], st["shadow_rays"]]
    print(f"{scene} {spp} spp binning={'on' if binning else 'off'}: rays {rays}")
    for k, kn in enumerate(kinds):
        for slot, nm in steps.items():
            lanes, n = buf[10 * k + 2 * slot], buf[10 * k + 2 * slot + 1]
            if n:
                uni, uni2 = buf[30 + 10 * k + 2 * slot], buf[30 + 10 * k + 2 * slot + 1]
                print(f"  {kn:28s} {nm:30s} wave-steps {n:12d}  lane-steps per ray {lanes / max(rays[k], 1):7.2f}  avg active lanes {lanes / n:5.1f} / 64"
                      + (f"  same address in all active lanes {uni / n:5.1%}, and same direction signs {uni2 / n:5.1%}" if slot in (0, 2, 3) else ""))
    if buf[62]:
        print(f"  node steps in which no lane has more than one child hit {buf[60] / buf[62]:5.1%}, more than two {1 - buf[61] / buf[62]:5.1%} (all ray kinds)")
