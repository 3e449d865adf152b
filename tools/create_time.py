import sys, time
sys.path.insert(0, '/root/repo')
import torch  # noqa
import bench
from rustray_amd import capi
for sc in ("sponza_syn", "helmet_syn"):
    fs, cam, cfg = bench.build_workload(sc, 1280, 720, 16, 1)
    for _ in range(2):
        t0 = time.perf_counter()
        ds = capi.DeviceScene(fs, 0)
        t = time.perf_counter() - t0
        ds.close()
    print(sc, "rr_scene_create %.3f s" % t)
