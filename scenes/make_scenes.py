"""Regenerates scenes/*.npz from the reference's scene files and assets.

Run in the build container (where /root/reference exists):
    python scenes/make_scenes.py
The .npz files are flat scenes (rustray_amd.flat.FlatScene): decoded textures,
de-referenced meshes, materials, lights and the camera parameters.  They are
DATA: inputs of the trace loop.  The GPU box has no /root/reference, so tests
and bench.py read these files only.

  spheres.npz     scene/spheres.json  (BASELINE config C1)
  monkey.npz      scene/monkey.json   (BASELINE config C2)
  kbert.npz       scene/kbert.json    (spot light, flat shading, two meshes, one base texture)
  earth_room.npz  scene/earth_in_room.json if loadable (textured sphere inside planes)
Synthetic stand-ins for C3-C5 (assets absent offline) are built from
these by rustray_amd/synthetic.py at run time.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rustray_amd.scene import load_scene  # noqa: E402

REF = os.environ.get("RUSTRAY_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def export(name, scene_json, w, h):
    sc = load_scene(scene_json, w, h, root=REF)
    fs = sc.flatten()
    fs.meta = {"camera": sc.cam.state(), "config": sc.raytracing_config, "source": scene_json}
    fs.save(os.path.join(OUT, name + ".npz"))
    print(name, "items", len(fs.items), "meshes", len(fs.meshes), "tris", fs.n_triangles_instanced(),
          "textures", [t.shape for t in fs.textures], "lights", len(fs.lights))


if __name__ == "__main__":
    export("spheres", "scene/spheres.json", 256, 256)
    export("monkey", "scene/monkey.json", 800, 600)
    export("kbert", "scene/kbert.json", 1280, 720)
    for extra, path in (("earth_room", "scene/earth_in_room.json"), ("spheres_room", "scene/spheres_in_room.json"),
                        ("monkey_room", "scene/monkey_in_room.json")):
        try:
            export(extra, path, 1280, 720)
        except Exception as e:  # noqa: BLE001
            print("skipped", extra, "->", repr(e))
