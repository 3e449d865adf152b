"""Developer check: GPU render vs oracle on the fixture scenes (run on a GPU box)."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import load_scene, camera_for, compare_frames
from rustray_amd.flat import make_config
from rustray_amd import capi
from oracle import binding as ob

cases = [("spheres", 256, 256, 1, False), ("monkey", 200, 150, 4, True), ("kbert", 160, 90, 2, True),
         ("earth_room", 160, 90, 2, True), ("spheres_room", 160, 90, 2, True), ("monkey_room", 160, 90, 2, True)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, w, h, spp, mc in cases:
    fs = load_scene(name)
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=spp, monte_carlo=mc, seed=7)
    t = time.time(); ref = ob.render(fs.c_struct(), cam, cfg, n_threads=16, want_counters=True); t_cpu = time.time() - t
    with capi.DeviceScene(fs, 0) as ds:
        t = time.time(); out = ds.render(cam, cfg); t_gpu = time.time() - t
        st = ds.stats()
    cmp_ = compare_frames(out, ref)
    print(name, w, h, spp, json.dumps(cmp_), "cpu %.2fs gpu %.3fs" % (t_cpu, t_gpu))
    print("   gpu rays", st["primary_rays"], st["secondary_rays"], st["shadow_rays"], st["shaded_hits"],
          "oracle", ref["counters"]["rays_primary"], ref["counters"]["rays_secondary"], ref["counters"]["rays_shadow"], ref["counters"]["shaded_hits"])
    if cmp_["n_rgb_over"]:
        d = np.abs(out["rgba"].astype(int) - ref["rgba"].astype(int))[..., :3].max(-1)
        ys, xs = np.nonzero(d > 1)
        for y, x in list(zip(ys, xs))[:8]:
            print("   px", x, y, out["rgba"][y, x], ref["rgba"][y, x])
    from PIL import Image
    Image.fromarray(out["rgba"]).save(f"gpurun_out/{name}_gpu.png")
    Image.fromarray(ref["rgba"]).save(f"gpurun_out/{name}_ref.png")
