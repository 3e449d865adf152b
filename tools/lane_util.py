#!/usr/bin/env python3
"""Developer tool: average active lanes per executed step, by kind, from a -DRR_EXP_UTIL build of the library.
usage: RUSTRAY_HIP_LIB=build/lib_util.so python tools/lane_util.py [scene] [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
from rustray_amd import capi

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_syn"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
args = argparse.Namespace(scene=scene, width=1280, height=720, spp=spp, monte_carlo=1)
fs, cam, cfg = bench.build_workload(args)
lib = capi.lib()
buf = (C.c_ulonglong * 32)()
names = ["top-level node step", "item setup", "mesh node step", "triangle test", "mesh walk entry"]
with capi.DeviceScene(fs, 0) as ds:
    for kernel in ("all",):
        lib.rr_exp_util.argtypes = [C.c_void_p, C.c_int]; lib.rr_exp_util(None, 1)
        ds.render(cam.c_struct(), cfg, aux=False)
        lib.rr_exp_util(buf, 0)
        for i, nm in enumerate(names):
            lanes, steps = buf[2 * i], buf[2 * i + 1]
            if steps:
                print(f"{scene} {nm:22s} wave-steps {steps:12d}  lane-steps {lanes:14d}  avg active lanes {lanes / steps:5.1f} / 64")
