"""Edges of the frame call on the GPU: smallest and most lopsided frames, the largest accepted recursion depth and sample
count, scenes without lights / with many lights, every light disabled, and argument errors that must come back as codes."""
import copy

import numpy as np
import pytest

from rustray_amd.flat import Light, make_config
from tests.helpers import camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def _check(hip, oracle, fs, w, h, cfg, threads=8):
    cam = camera_for(fs, w, h).c_struct()
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=threads, want_counters=True)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    c = ref["counters"]
    assert st["primary_rays"] == c["rays_primary"] and st["secondary_rays"] == c["rays_secondary"] and st["shaded_hits"] == c["shaded_hits"]
    return out, st


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (1, 1, 64), (3, 1, 5), (1, 7, 2), (257, 2, 3), (2, 129, 1)])
def test_tiny_and_lopsided_frames(hip, oracle, w, h, spp):
    _check(hip, oracle, load_scene("spheres"), w, h, make_config(samples=spp, monte_carlo=True, seed=2))


def test_deepest_recursion_the_abi_accepts(hip, oracle):
    """max_recursion 16 on the branching spheres scene (two glass spheres facing each other keep every path alive to the last
    level), and RR_MAX_RECURSION = 30 (31 depth levels: five bits of depth in a shadow record) on mirrors that spawn one child per
    hit; 31 is refused."""
    fs = load_scene("spheres")
    out, st = _check(hip, oracle, fs, 48, 48, make_config(samples=1, monte_carlo=False, seed=0, max_recursion=16))
    assert st["secondary_rays"] > 4 * st["primary_rays"]
    mirrors = load_scene("spheres_room")              # a closed room: no ray leaves
    for m in mirrors.materials:
        m.alpha, m.reflectivity = 1.0, 0.9        # no refraction child: chains of 30 reflections, not trees
    out, st = _check(hip, oracle, mirrors, 24, 24, make_config(samples=2, monte_carlo=True, seed=1, max_recursion=30))
    assert st["secondary_rays"] == 30 * st["primary_rays"]
    with hip.DeviceScene(fs, 0) as ds:
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(camera_for(fs, 8, 8).c_struct(), make_config(samples=1, max_recursion=31))
        assert e.value.code == -2


def test_many_samples_on_a_small_frame(hip, oracle):
    """4096 samples per pixel: a 4096-cell table shuffled out of 4096^2 / 4 cells, 64-sample packets of one pixel."""
    fs = load_scene("spheres")
    cam = camera_for(fs, 8, 6).c_struct()
    cfg = make_config(samples=4096, monte_carlo=True, seed=5)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        assert ds.stats()["primary_rays"] == 8 * 6 * 4096
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=16383))          # beyond the BUILT-IN table's limit (RR_MAX_SAMPLES)
        assert e.value.code == -2
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_the_references_own_sample_limit_with_the_callers_table(hip, oracle):
    """With the caller's sub-sample table the reference's own limit applies: 32766 samples (the u16 `(samples + 2).next_power_of_two()`
    overflows from 32767 on, src/raytracing.rs:297).  4x3 pixels, cell_size 16384, a table of random cells; 32767 is refused."""
    fs = load_scene("spheres")
    cam = camera_for(fs, 4, 3).c_struct()
    n = 32766
    rng = np.random.default_rng(8)
    table = rng.integers(0, 16384, (n, 2)).astype(np.uint16)
    cfg = make_config(samples=n, monte_carlo=True, seed=9, max_recursion=2)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg, sample_xy=table)
        assert ds.stats()["primary_rays"] == 12 * n
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=32767), sample_xy=np.zeros((32767, 2), np.uint16))
        assert e.value.code == -2
    ref = oracle.render(fs.c_struct(), cam, cfg, sample_xy=table, n_threads=8)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0, res


def test_lights_none_many_and_disabled(hip, oracle):
    base = load_scene("monkey_room")
    cfg = make_config(samples=2, monte_carlo=True, seed=6)
    dark = copy.deepcopy(base); dark.lights = []
    out, st = _check(hip, oracle, dark, 64, 36, cfg)
    assert st["shadow_rays"] == 0
    off = copy.deepcopy(base)
    for l in off.lights:
        l.enabled = False
    out_off, st_off = _check(hip, oracle, off, 64, 36, cfg)
    assert st_off["shadow_rays"] == 0 and np.array_equal(out_off["rgba"], out["rgba"])
    many = copy.deepcopy(base)
    rng = np.random.default_rng(3)
    many.lights = [Light(pos=tuple(rng.uniform(-4, 4, 3) + np.array([0, 3, -8])), dir=(0.0, -1.0, 0.0), color=tuple(rng.uniform(0.2, 1.0, 3)),
                         intensity=float(rng.uniform(5, 30)), light_type=int(i % 3), max_angle=0.9, enabled=bool(i % 5)) for i in range(24)]
    out_many, st_many = _check(hip, oracle, many, 64, 36, cfg)
    assert st_many["shadow_rays"] > 5 * st_many["shaded_hits"]   # 19 enabled lights; a light with a zero term casts no ray (D7)


def test_bad_frame_arguments_come_back_as_codes(hip):
    fs = load_scene("spheres")
    with hip.DeviceScene(fs, 0) as ds:
        cam = camera_for(fs, 16, 16).c_struct()
        for bad in (dict(samples=0),):
            with pytest.raises(hip.RustrayHipError) as e:
                ds.render(cam, make_config(**bad))
            assert e.value.code == -1
        cam.width = 0
        with pytest.raises(hip.RustrayHipError):
            ds.render(cam, make_config(samples=1))
        cam = camera_for(fs, 16, 16).c_struct()
        cam.view_inverse[5] = float("nan")
        with pytest.raises(hip.RustrayHipError):
            ds.render(cam, make_config(samples=1))
        ok = ds.render(camera_for(fs, 16, 16).c_struct(), make_config(samples=1))   # the handle survives every rejected call
        assert ok["rgba"].shape == (16, 16, 4)
