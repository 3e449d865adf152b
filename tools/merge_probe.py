#!/usr/bin/env python3
"""Developer probe: how much does the two-level structure cost?  Renders a synthetic scene as it is and with all of its
mesh items merged into ONE world-space mesh (one material: the colours are wrong, the geometry and so the closest-hit
and shadow walks are the same) and prints the kernel times of both."""
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rustray_amd import capi
from rustray_amd.flat import Item, MeshData, RR_ITEM_MESH

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_syn"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fs, cam, cfg = bench.build_workload(scene, 1280, 720, spp, 1)
camc = cam.c_struct()


def run(tag, f):
    with capi.DeviceScene(f, 0) as ds:
        ds.set_profiling(True)
        ds.render(camc, cfg, aux=False)
        ds.render(camc, cfg, aux=False)
        st = ds.stats()
    print(f"{tag:12s} items {len(f.items):4d}  frame {st['ms_total']:7.2f}  closest {st['ms_trace_closest']:6.2f}  shadow {st['ms_trace_shadow']:6.2f}  shade {st['ms_shade']:6.2f}  "
          f"rays p/s/sh {st['primary_rays'] / 1e6:.1f} / {st['secondary_rays'] / 1e6:.1f} / {st['shadow_rays'] / 1e6:.1f} M")


run("as is", fs)
m = copy.copy(fs)
pos, idx, base = [], [], 0
keep = []
first_mesh_item = None
for it in fs.items:
    if it.kind != RR_ITEM_MESH:
        keep.append(it)
        continue
    if first_mesh_item is None:
        first_mesh_item = it
    md = fs.meshes[it.mesh]
    p = np.concatenate([md.positions.astype(np.float64), np.ones((len(md.positions), 1))], axis=1) @ np.asarray(it.trans, np.float64).T
    pos.append(p[:, :3].astype(np.float32))
    idx.append(md.indices.astype(np.uint32) + base)
    base += len(md.positions)
P, I = np.concatenate(pos), np.concatenate(idx)
m.meshes = [MeshData(positions=P, indices=I)]
eye = np.eye(4, dtype=np.float32)
mat = copy.copy(fs.materials[first_mesh_item.material]); mat.texture = [-1] * 8; mat.reflectivity = 0.0; mat.alpha = 1.0
cache = copy.copy(fs.materials[first_mesh_item.material_cache]); cache.reflectivity = 0.0; cache.alpha = 1.0
m.materials = list(fs.materials) + [mat, cache]
merged = Item(kind=RR_ITEM_MESH, id=7, material=len(m.materials) - 2, material_cache=len(m.materials) - 1, mesh=0, trans=eye, trans_inv=eye.copy(),
              bbox_min=tuple(P.min(axis=0).tolist()), bbox_max=tuple(P.max(axis=0).tolist()), name="merged")
m.items = keep + [merged]
run("merged", m)
f2 = copy.copy(fs); f2.materials = [copy.copy(x) for x in fs.materials]
for x in f2.materials:
    x.texture = [-1] * 8; x.reflectivity = 0.0; x.alpha = 1.0
run("as is, plain", f2)
