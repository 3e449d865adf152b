"""Ray-level parity: every Raytracing::trace call the oracle makes while rendering a window (its test-only ray log) is
replayed through rr_trace_rays, and found / item / face id / toi must agree bit for bit -- the closest-hit kernel
checked below the frame level, where a wrong hit can hide behind a small colour weight."""
import os

import numpy as np
import pytest

from rustray_amd.flat import FlatScene, make_config
from tests.helpers import GOLDEN, camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def _replay(hip, oracle, fs, cam, cfg, window):
    with oracle.ray_log(1 << 18) as log:
        ref = oracle.render(fs.c_struct(), cam, cfg, window=window, n_threads=1)
        rays = log.rays()
    closest = ~rays["for_shadow"]
    n_checked = 0
    with hip.DeviceScene(fs, 0) as ds:
        for depth in sorted(set(rays["depth"][closest].tolist())):
            m = closest & (rays["depth"] == depth)
            found, item, face, toi = ds.trace_rays(rays["origin"][m], rays["dir"][m], int(depth))
            assert np.array_equal(found, rays["found"][m]), ("found", depth)
            f = found
            assert np.array_equal(item[f], rays["item"][m][f]), ("item", depth)
            assert np.array_equal(face[f], rays["face"][m][f]), ("face", depth)
            assert np.array_equal(toi[f].view(np.uint32), rays["toi"][m][f].view(np.uint32)), ("toi", depth)
            n_checked += int(m.sum())
    return ref, rays, n_checked


def test_rays_of_a_rendered_window_hit_the_same_things(hip, oracle):
    fs = load_scene("spheres_room")
    cam = camera_for(fs, 96, 64).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=3, max_recursion=4)
    _, rays, n = _replay(hip, oracle, fs, cam, cfg, (24, 16, 72, 48))
    assert n > 48 * 32 * 2 and rays["depth"].max() >= 3


def test_rays_into_a_mesh_scene_report_the_reference_face_ids(hip, oracle):
    fs = load_scene("monkey")
    cam = camera_for(fs, 80, 60).c_struct()
    cfg = make_config(samples=1, monte_carlo=False, seed=0, max_recursion=3)
    _, rays, n = _replay(hip, oracle, fs, cam, cfg, (20, 10, 60, 50))
    assert n >= 40 * 40 and rays["face"][rays["found"]].max() > 0


def test_non_finite_rays_hit_the_first_sphere_as_in_the_reference(hip, oracle):
    """tests/golden/fuzz_far_822.npz: reduced from a tools/fuzz_parity.py `far` mismatch (scene scaled by 100, 1e6 from the
    origin).  A normal-mapped sphere yields a NaN normal, the reflection ray is NaN in origin and direction, and in the
    reference every candidate sphere then reports Some(NaN) (ray_toi_with_ball's comparisons are all false), the first in
    bbox order is kept and shades a finite ambient colour.  The top-level tree would cull such rays."""
    fs = FlatScene.load(os.path.join(GOLDEN, "fuzz_far_822.npz"))
    w, h = fs.meta["wh"]
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(**fs.meta["kw"])
    ref, rays, _ = _replay(hip, oracle, fs, cam, cfg, (27, 33, 28, 34))
    nan_rays = np.isnan(rays["origin"]).any(axis=1) & ~rays["for_shadow"]
    assert nan_rays.any() and rays["found"][nan_rays].all() and np.isnan(rays["toi"][nan_rays]).all()
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    full = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    res = compare_frames(out, full)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res


@pytest.mark.parametrize("name", ["fuzz_far_2514", "fuzz_far_3025"])
def test_far_from_the_origin_the_top_level_keeps_every_candidate(hip, oracle, name):
    """Reduced from tools/fuzz_parity.py `far` mismatches.  The reference has no world-space test: it moves the ray into
    an item's space with the item's f32 inverse matrix and tests there.  1e5 .. 1e8 from the origin, with a sheared
    transform (cond ~ 3000), that local ray sits up to 15 world units from the true one, so world boxes padded by float
    spacing cull items the reference hits.  The boxes are padded by a derived bound instead (rr_api.hip: padded_world_box),
    for the camera's distance as well (the top level is rebuilt when a camera moves far outside the scene)."""
    fs = FlatScene.load(os.path.join(GOLDEN, name + ".npz"))
    w, h = fs.meta["wh"]
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(**fs.meta["kw"])
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True, brute_force=True)
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res
    assert st["secondary_rays"] == ref["counters"]["rays_secondary"] and st["shaded_hits"] == ref["counters"]["shaded_hits"]
    assert (out["object_id"] != 0).sum() > 0


def test_camera_far_outside_the_scene_and_back(hip, oracle):
    """The top level is padded for the ray origins' distance from the origin: a camera 3e6 away from a unit-sized scene
    makes it rebuild (and again when the camera returns); both frames match the oracle."""
    from rustray_amd.camera import Camera
    fs = load_scene("spheres")
    near = camera_for(fs, 64, 48)
    st = dict(fs.meta["camera"]); st["width"], st["height"] = 64, 48
    d = np.asarray(st["dir"], np.float64); d /= np.linalg.norm(d)
    st["eye_pos"] = [float(v) for v in (np.asarray(st["eye_pos"], np.float64) - d * 3.0e6)]
    st["clipping_far"] = 1e7
    far = Camera.from_state(st)
    cfg = make_config(samples=1, monte_carlo=False, seed=0, max_recursion=2)
    with hip.DeviceScene(fs, 0) as ds:
        a0 = ds.render(near.c_struct(), cfg)
        b = ds.render(far.c_struct(), cfg)
        a1 = ds.render(near.c_struct(), cfg)
    assert np.array_equal(a0["rgba"], a1["rgba"]) and np.array_equal(a0["object_id"], a1["object_id"])
    for got, cam in ((a0, near), (b, far)):
        ref = oracle.render(fs.c_struct(), cam.c_struct(), cfg, n_threads=8)
        res = compare_frames(got, ref)
        assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, res


@pytest.mark.parametrize("far", [False, True])
def test_random_rays_with_special_values_hit_the_same_things(hip, oracle, far):
    """tools/fuzz_rays.py in small: random rays into the fuzzer's rich scenes, a third with NaN components, infinite origin
    components, signed zeros or denormals, against the oracle's all-items form -- bit for bit (a NaN toi as NaN).  A point
    with a non-finite component takes the DIVIDING form of the inverse transform even for an affine inverse."""
    from tools.fuzz_parity import rich_scene, far_and_scaled
    special_o = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 1.0], np.float32)
    special_d = np.array([np.nan, 0.0, -0.0, 1e-42, -1e-42, 1.0], np.float32)
    n_special_found = 0
    # far: the scenes moved up to 1e8 from the origin and scaled by up to 1e4.  Seed 419 holds a ray that starts 2e4 units from a
    # sphere, grazes past it, and "hits" it in the reference 7 units in FRONT of its box: the discriminant of ray_toi_with_ball is
    # rounding noise there, so the top level must not prune by distance more tightly than sqrt(u) (RR_TOI_SLACK).
    for seed in (range(412, 422) if far else range(390, 400)):
        fs = rich_scene(9000 + seed)
        scale = 1.0
        if far:
            fs, scale = far_and_scaled(fs, seed)
        rng = np.random.default_rng(seed)
        n = 3000 if far else 1200
        eye = np.asarray(fs.meta["camera"]["eye_pos"], np.float32)
        o = (eye[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(rng.choice([0.01, 1.0, 5.0]) * scale)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)
        for i in range(n // 3):
            for _ in range(int(rng.integers(1, 4))):
                if rng.random() < 0.5:
                    o[i, int(rng.integers(0, 3))] = special_o[int(rng.integers(0, len(special_o)))]
                else:
                    d[i, int(rng.integers(0, 3))] = special_d[int(rng.integers(0, len(special_d)))]
        depth = int(rng.integers(1, 3))
        with hip.DeviceScene(fs, 0) as ds:
            g = ds.trace_rays(o, d, depth)
        r = oracle.trace_rays(fs.c_struct(), o, d, depth, brute_force=True)
        assert np.array_equal(g[0], r[0]), seed
        f = g[0]
        assert np.array_equal(g[1][f], r[1][f]) and np.array_equal(g[2][f], r[2][f]), seed
        same = (g[3][f].view(np.uint32) == r[3][f].view(np.uint32)) | (np.isnan(g[3][f]) & np.isnan(r[3][f]))
        assert same.all(), seed
        n_special_found += int(np.isnan(g[3][f]).sum())
    assert far or n_special_found > 0      # NaN hits (non-finite rays against spheres) did occur


def test_trace_rays_argument_checks(hip):
    fs = load_scene("spheres")
    with hip.DeviceScene(fs, 0) as ds:
        found, item, face, toi = ds.trace_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 1)
        assert len(found) == 0
        with pytest.raises(hip.RustrayHipError):
            ds.trace_rays(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32), 0)
        # a ray that points away from everything
        found, item, face, toi = ds.trace_rays(np.array([[0, 1e6, 0]], np.float32), np.array([[0, 1, 0]], np.float32), 1)
        assert not found[0] and item[0] == -1
