"""Generates the golden frames under tests/golden/ with the CPU oracle (oracle/).

The reference holds no golden vectors for this path and cannot be run offline, so
these are SELF-GENERATED regression pins: they freeze the oracle's output so that any later change
to the restatement (or to the scene flattening) is caught, and they give the GPU parity tests a
second, file-based target.  Regenerate only deliberately:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import binding as ob  # noqa: E402
from rustray_amd.flat import make_config  # noqa: E402
from tests.helpers import camera_for, load_scene  # noqa: E402

CASES = {
    # BASELINE config C1: scene/spheres.json 256x256 samples=1 monte_carlo=0
    "spheres_c1": dict(scene="spheres", w=256, h=256, spp=1, mc=False, seed=0, window=None),
    # BASELINE config C2 (scene/monkey.json 800x600 monte_carlo=1) at 4 spp, 128x96 crop around the head
    "monkey_c2_crop": dict(scene="monkey", w=800, h=600, spp=4, mc=True, seed=0, window=(336, 252, 464, 348)),
    # textured room: planes, spheres, 4 lights, bilinear maps
    "spheres_room": dict(scene="spheres_room", w=160, h=90, spp=2, mc=True, seed=3, window=None),
}


def render_case(c):
    fs = load_scene(c["scene"])
    cam = camera_for(fs, c["w"], c["h"]).c_struct()
    cfg = make_config(samples=c["spp"], monte_carlo=c["mc"], seed=c["seed"])
    out = ob.render(fs.c_struct(), cam, cfg, window=c["window"], n_threads=8)
    if c["window"]:
        x0, y0, x1, y1 = c["window"]
        out = {k: v[y0:y1, x0:x1] for k, v in out.items()}
    return out


if __name__ == "__main__":
    for name, c in CASES.items():
        out = render_case(c)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out["rgba"], normal=out["normal"], depth=out["depth"],
                            object_id=out["object_id"])
        print(name, out["rgba"].shape, int(out["rgba"][..., :3].sum()))
