#!/usr/bin/env python3
"""Developer probe: k_shade time of the bench frame with parts of the scene switched off."""
import argparse
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rustray_amd import capi

args = argparse.Namespace(scene=sys.argv[1] if len(sys.argv) > 1 else "sponza_syn", width=1280, height=720, spp=64, monte_carlo=1)
fs, cam, cfg = bench.build_workload(args.scene, args.width, args.height, args.spp, args.monte_carlo)
camc = cam.c_struct()


def run(tag, f, c):
    with capi.DeviceScene(f, 0) as ds:
        ds.set_profiling(True)
        ds.render(camc, c, aux=False)
        ds.render(camc, c, aux=False)
        st = ds.stats()
    print(f"{tag:34s} frame {st['ms_total']:7.2f}  closest {st['ms_trace_closest']:6.2f}  shadow {st['ms_trace_shadow']:6.2f}  shade {st['ms_shade']:6.2f}  "
          f"hits {st['shaded_hits'] / 1e6:6.1f} M  shadow rays {st['shadow_rays'] / 1e6:6.1f} M  secondary {st['secondary_rays'] / 1e6:5.1f} M")


run("as is", fs, cfg)
f2 = copy.copy(fs); f2.lights = []
run("no lights", f2, cfg)
f3 = copy.copy(fs); f3.materials = [copy.copy(m) for m in fs.materials]
for m in f3.materials:
    m.texture = [-1] * 8
run("no textures", f3, cfg)
f4 = copy.copy(fs); f4.materials = [copy.copy(m) for m in fs.materials]
for m in f4.materials:
    m.receive_shadow = False
run("lights, nothing receives shadows", f4, cfg)
f5 = copy.copy(fs); f5.materials = [copy.copy(m) for m in fs.materials]
for m in f5.materials:
    m.shadow_softness = 0.0; m.roughness = 0.0
run("no softness / roughness", f5, cfg)
c2 = bench.make_config(samples=args.spp, monte_carlo=False, seed=0, max_recursion=6) if hasattr(bench, "make_config") else None
if c2 is None:
    from rustray_amd.flat import make_config
    c2 = make_config(samples=args.spp, monte_carlo=False, seed=0, max_recursion=6)
run("monte_carlo off", fs, c2)
f6 = copy.copy(fs); f6.materials = [copy.copy(m) for m in fs.materials]
for m in f6.materials:
    m.reflectivity = 0.0; m.alpha = 1.0
run("no reflection / transmission", f6, cfg)
