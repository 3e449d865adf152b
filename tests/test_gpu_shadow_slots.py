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


def test_any_number_of_lights(hip, oracle):
    """The reference loops over scene.lights without a limit (src/raytracing.rs:814).  Up to 32 enabled lights level 1 keeps fixed
    shadow slots (one validity bit per light and lane); 40 enabled lights (of 44: every eleventh disabled, its slot still counts
    as an RNG stream) take the dense queue of the deeper levels.  Both against the oracle, and 32 | 33 lights against each other
    on the lights they share."""
    import copy
    fs = _scene(9119)
    assert len(fs.items) >= 17
    rng = np.random.default_rng(40)
    proto = fs.lights[0]
    lights = []
    for i in range(44):
        l = copy.copy(proto)
        l.pos = tuple(float(v) for v in (np.asarray(proto.pos) + rng.uniform(-3.0, 3.0, 3)))
        l.color = tuple(float(v) for v in rng.uniform(0.2, 1.0, 3))
        l.intensity = float(proto.intensity) / 12.0
        l.light_type = int(i % 3)
        l.max_angle = 1.0
        l.enabled = (i % 11) != 10
        lights.append(l)
    cam = camera_for(fs, 48, 32).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=77, max_recursion=3)
    for n_lights in (44, 35):   # 40 and 32 enabled
        fs.lights = lights[:n_lights]
        n_on = sum(1 for l in fs.lights if l.enabled)
        with hip.DeviceScene(fs, 0) as ds:
            a = ds.render(cam, cfg)
            st = ds.stats()
            ds.set_tuning(shade_chunk_rays=65536, queue_budget_bytes=1)
            b = ds.render(cam, cfg)
        assert np.array_equal(a["rgba"], b["rgba"])
        ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True)
        res = compare_frames(a, ref)
        assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, (n_on, res)
        assert 0 < st["shadow_rays"] <= ref["counters"]["rays_shadow"] and st["shaded_hits"] == ref["counters"]["shaded_hits"], n_on


def test_hundreds_of_unshadowed_lights_do_not_overflow_the_lane_sums(hip, oracle):
    """the 32-bit fixed-point lane sums of k_shade are flushed every 32 enabled lights; the flush sat inside the branch of
    a light with a non-zero term, so a back-facing light number 31 or 63 skipped it.  140 enabled lights on materials that do NOT
    receive shadows (every term is added in k_shade itself), bright enough that 96 unflushed terms would pass 2^31 (each term ~ 1.5
    at scale 2^24), every light whose ordinal is 31 mod 32 placed BELOW the floor so that its term is exactly zero."""
    import copy
    fs = _scene(9119)
    for m in fs.materials:
        m.receive_shadow = False
    rng = np.random.default_rng(41)
    proto = fs.lights[0]
    lights = []
    for i in range(140):
        l = copy.copy(proto)
        l.enabled = True
        l.light_type = 0   # directional: intensity is the term's scale, whatever the distance
        l.color = (1.0, 1.0, 1.0)
        l.intensity = 1.5
        down = (i % 32) == 31
        l.dir = (float(rng.uniform(-0.2, 0.2)), 1.0 if down else -1.0, float(rng.uniform(-0.2, 0.2)))   # `down`: shines upwards, from below every surface that faces up
        lights.append(l)
    fs.lights = lights
    cam = camera_for(fs, 48, 32).c_struct()
    cfg = make_config(samples=2, monte_carlo=False, seed=1, max_recursion=1)
    with hip.DeviceScene(fs, 0) as ds:
        a = ds.render(cam, cfg)
        assert ds.stats()["shadow_rays"] == 0
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    res = compare_frames(a, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    assert (a["rgba"][..., :3] == 255).mean() > 0.1   # the sums really are large: most lit pixels saturate
