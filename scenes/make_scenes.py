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
  monkey_glb.npz  scene/models/monkey/monkey.glb through the glTF path (two glTF lights, glTF camera, textured plane)
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
                        ("monkey_room", "scene/monkey_in_room.json"), ("kbert_room", "scene/kbert_in_room.json"),
                        ("earth", "scene/earth.json"), ("floor", "scene/floor.json")):
        try:
            export(extra, path, 1280, 720)
        except Exception as e:  # noqa: BLE001
            print("skipped", extra, "->", repr(e))


def export_assets():
    """assets.npz: base meshes and textures the synthetic C3-C5 stand-ins are assembled from."""
    import numpy as np
    from PIL import Image
    from rustray_amd.flat import FlatScene
    from rustray_amd.scene import Scene
    sc = Scene(REF)
    sc.load_wavefront("scene/models/monkey/monkey.obj")
    sc.load_wavefront("scene/models/kBert/kBert_thumbsup_bevel.obj")
    sc.load_wavefront("scene/models/kBert/kBert_thumbsup.obj")
    fs = FlatScene()
    fs.name = "assets"
    fs.meshes = list(sc.meshes)
    names = ["monkey", "kbert_bevel_a", "kbert_bevel_b", "kbert_a", "kbert_b"]
    tex = {}
    # 1024^2 maps are stored at 512^2 (box filter) to keep the repository small; the env map keeps its size
    for key, path, size in (("leather_base", "scene/textures/leather/Leather_Weave_006_basecolor.jpg", 512),
                            ("leather_normal", "scene/textures/leather/Leather_Weave_006_normal.jpg", 512),
                            ("wall_base", "scene/textures/wall/Wall_Stone_022_basecolor.jpg", 512),
                            ("wall_normal", "scene/textures/wall/Wall_Stone_022_normal.jpg", 512),
                            ("wall_roughness", "scene/textures/wall/Wall_Stone_022_roughness.jpg", 512),
                            ("wall_ao", "scene/textures/wall/Wall_Stone_022_ambientOcclusion.jpg", 512),
                            ("env", "scene/textures/environment/footprint_court.jpg", None),
                            ("checker", "scene/textures/checkerboard.png", 256),
                            ("man", "scene/models/kBert/man.png", None)):
        im = Image.open(os.path.join(REF, path)).convert("RGBA")
        if size:
            im = im.resize((size, size), Image.BOX)
        tex[key] = len(fs.textures)
        fs.textures.append(np.ascontiguousarray(np.asarray(im, dtype=np.uint8)))
    fs.meta = {"meshes": {n: i for i, n in enumerate(names)}, "textures": tex}
    fs.save(os.path.join(OUT, "assets.npz"))
    print("assets", [len(m.indices) for m in fs.meshes], {k: fs.textures[v].shape for k, v in tex.items()})


if __name__ == "__main__":
    export_assets()
    # glTF path (src/scene.rs:722-1124): the only .glb that ships inside the reference
    export("monkey_glb", "scene/models/monkey/monkey.glb", 1280, 720)
