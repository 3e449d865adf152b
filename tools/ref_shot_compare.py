"""Developer tool: render a README shot scene with the CPU oracle and write reference / ours / |diff| PNGs.
usage: python tools/ref_shot_compare.py NAME [spp] [outdir]   (NAME in room_spheres, room_kbert, floor_monkey)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob
from rustray_amd.flat import make_config
from tests.helpers import camera_for, load_scene
from tests.golden.ref_shots.make_ref_shots import box2

def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / max(np.mean(d * d), 1e-12))

if __name__ == "__main__":
    from PIL import Image
    name = sys.argv[1]; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/shots"
    shot = np.load(os.path.join(ROOT, "tests/golden/ref_shots", name + ".npz"))
    ref = shot["rgb_half"]
    fs = load_scene(name)
    cam = camera_for(fs, 1280, 720).c_struct()
    cfg = make_config(samples=spp, monte_carlo=True, seed=0)
    t0 = time.time()
    r = ob.render(fs.c_struct(), cam, cfg, n_threads=8)
    ours = box2(r["rgba"][..., :3])
    d = np.abs(ours.astype(int) - ref.astype(int))
    print(name, "spp", spp, "time %.1fs" % (time.time() - t0), "PSNR %.2f dB" % psnr(ours, ref), "mean|d| %.3f" % d.mean(), "median", np.median(d), "max", d.max(),
          "signed mean per channel", (ours.astype(float) - ref.astype(float)).mean(axis=(0, 1)))
    Image.fromarray(ours).save(os.path.join(out, name + "_ours.png"))
    Image.fromarray(ref).save(os.path.join(out, name + "_ref.png"))
    Image.fromarray(np.clip(d * 8, 0, 255).astype(np.uint8)).save(os.path.join(out, name + "_diff8.png"))
    np.save(os.path.join(out, name + "_ours.npy"), ours)
