"""Level 1 of a scene whose top level has a packet form (17 .. 512 items) keeps its shadow rays in FIXED slots
(enabled light x hit, one validity word per 64) and traces them with the packet form; deeper levels and small scenes use the
dense sharded queue.  Both against the oracle, with several lights (one of them disabled), chunked and unchunked."""
import numpy as np
import pytest

from rustray_amd.flat import make_config
from tests.helpers import camera_for, compare_frames

pytestmark = pytest.mark.gpu


def _scene(seed):
    from tools.fuzz_parity import rich_scene   # the fuzzer's generator: texture maps of every kind, alpha, several lights
    return rich_scene(seed)


@pytest.mark.parametrize("seed", [9110, 9119, 9128])
def test_many_items_several_lights_fixed_shadow_slots(hip, oracle, seed):
    fs = _scene(seed)
    assert len(fs.items) >= 17 and sum(1 for l in fs.lights if l.enabled) >= 2
    cam = camera_for(fs, 72, 48).c_struct()
    cfg = make_config(samples=4, monte_carlo=True, seed=seed, max_recursion=4)
    with hip.DeviceScene(fs, 0) as ds:
        a = ds.render(cam, cfg)
        st = ds.stats()
        ds.set_tuning(shade_chunk_rays=65536, queue_budget_bytes=1)   # several shade chunks per level, sliced levels
        b = ds.render(cam, cfg)
        ds.set_tuning(sample_group=1)                                  # a packet = 64 pixels x one sample: partly filled shadow packets
        c = ds.render(cam, cfg)
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True)
    res = compare_frames(a, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    assert st["shadow_rays"] > 0 and st["shadow_rays"] <= ref["counters"]["rays_shadow"]


def test_more_than_32_enabled_lights_is_refused(hip):
    import copy
    fs = _scene(9119)
    fs.lights = [copy.copy(fs.lights[0]) for _ in range(33)]
    for l in fs.lights:
        l.enabled = True
    with pytest.raises(hip.RustrayHipError) as e:
        hip.DeviceScene(fs, 0)
    assert "32" in str(e.value)
    fs.lights = fs.lights[:32]
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(camera_for(fs, 32, 24).c_struct(), make_config(samples=1, monte_carlo=False, seed=0, max_recursion=1))
    assert out["rgba"].shape == (24, 32, 4)
