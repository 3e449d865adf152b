"""Host logic around the hot path: flat-scene round trip, loader semantics, camera, tiling."""
import math
import os

import numpy as np
import pytest

from rustray_amd.camera import Camera, approx_equal
from rustray_amd.flat import FlatScene, Material, make_config
from rustray_amd.renderer import region_pixels
from tests.helpers import SCENES, load_scene

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def test_flat_scene_npz_round_trip(tmp_path):
    fs = load_scene("spheres_room")
    p = str(tmp_path / "x.npz")
    fs.save(p)
    g = FlatScene.load(p)
    assert len(g.items) == len(fs.items) and len(g.textures) == len(fs.textures) and g.meta == fs.meta
    for a, b in zip(fs.items, g.items):
        assert a.id == b.id and a.kind == b.kind and a.name == b.name and np.array_equal(a.trans, b.trans)
        assert np.array_equal(a.trans_inv, b.trans_inv) and tuple(np.float32(a.bbox_min)) == tuple(np.float32(b.bbox_min))
    for a, b in zip(fs.materials, g.materials):
        assert a == b
    for a, b in zip(fs.textures, g.textures):
        assert np.array_equal(a, b)


def test_spheres_scene_semantics():
    """What Scene::load leaves in memory for scene/spheres.json."""
    fs = load_scene("spheres")
    assert [it.id for it in fs.items] == [3, 6, 9, 12, 15, 18, 21, 24]       # ids assigned twice (src/scene.rs:440,:541)
    assert [it.visible for it in fs.items] == [True, False, True, True, True, True, True, False]
    assert len(fs.lights) == 1 and fs.lights[0].pos == (-2.0, 10.0, 5.0) and fs.lights[0].intensity == 200.0
    m = fs.materials[fs.items[0].material]
    assert m.base_color == (1.0, 0.0, 1.0) and m.specular_color == tuple(np.float32(c) * np.float32(0.8) for c in m.base_color)
    tex = fs.materials[fs.items[5].material]
    assert tex.texture[0] == 0 and fs.textures[0].shape == (1024, 2048, 4) and tex.roughness == float(np.float32(0.05))
    cache = fs.materials[fs.items[5].material_cache]
    assert cache.texture == [-1] * 8 and cache.alpha == tex.alpha            # cache never holds textures
    cam = fs.meta["camera"]
    assert abs(cam["fov"] - math.radians(90.0)) < 1e-6 and cam["clipping_near"] == float(np.float32(0.1))


def test_monkey_scene_semantics():
    fs = load_scene("monkey")
    assert len(fs.items) == 1 and fs.items[0].id == 3 and len(fs.meshes[0].indices) == 15744
    m = fs.materials[fs.items[0].material]
    # .mtl values, then ambient = base * 0.01 (src/scene.rs:1284), then the JSON diff (refl .5, alpha .5, ior 1.5)
    assert m.shininess == float(np.float32(323.999994)) and m.reflectivity == 0.5 and m.alpha == 0.5 and m.refraction_index == 1.5
    assert abs(m.ambient_color[2] - 0.008) < 1e-6
    t = fs.items[0].trans
    assert abs(t[2, 3] + 10.0) < 1e-6 and abs(t[1, 1] - 1.3) < 1e-6                 # T * Ry(20 deg) * S(1.3)
    assert abs(t[0, 0] - 1.3 * math.cos(math.radians(20))) < 1e-5 and abs(t[0, 2] - 1.3 * math.sin(math.radians(20))) < 1e-5
    assert np.allclose(t.astype(np.float64) @ fs.items[0].trans_inv.astype(np.float64), np.eye(4), atol=1e-5)


@needs_reference
@pytest.mark.parametrize("name,path,w,h", [("spheres", "scene/spheres.json", 256, 256), ("monkey", "scene/monkey.json", 800, 600)])
def test_loader_reproduces_committed_fixture(name, path, w, h):
    from rustray_amd.scene import load_scene as load_ref
    sc = load_ref(path, w, h, root=REF)
    fs, g = sc.flatten(), load_scene(name)
    assert len(fs.items) == len(g.items)
    for a, b in zip(fs.items, g.items):
        assert a.id == b.id and np.array_equal(a.trans, b.trans) and a.material == b.material
    for a, b in zip(fs.materials, g.materials):
        assert a == b
    for a, b in zip(fs.meshes, g.meshes):
        assert np.array_equal(a.positions, b.positions) and np.array_equal(a.indices, b.indices)


@needs_reference
def test_auto_camera_and_spot_light_for_kbert():
    """kbert.json has no camera block: find_optimal_camera_pos runs (src/scene.rs:1426-1547)."""
    fs = load_scene("kbert")
    cam = Camera.from_state(fs.meta["camera"])
    assert not cam.is_default_cam()
    d = np.asarray(cam.dir) / np.linalg.norm(cam.dir)
    assert np.allclose(d, -np.asarray([-0.5, 0.5, 1.0]) / np.linalg.norm([-0.5, 0.5, 1.0]), atol=1e-6)
    assert fs.lights[0].light_type == 2 and abs(fs.lights[0].max_angle - math.radians(22.5)) < 1e-6
    assert [m.smooth_shading for m in fs.materials[::2]] == [False, False]


def test_camera_matrices():
    c = Camera(); c.eye_pos = np.array([1.0, 2.0, 3.0]); c.dir = np.array([0.0, 0.0, -1.0]); c.init(200, 100)
    assert np.allclose(c.projection @ c.projection_inverse, np.eye(4), atol=1e-9)
    assert np.allclose(c.view @ c.view_inverse, np.eye(4), atol=1e-12)
    # fov 90, aspect 2: x scale = 1 / (2 * tan 45)
    assert abs(c.projection[0, 0] - 0.5) < 1e-7 and abs(c.projection[1, 1] - 1.0) < 1e-7
    assert np.allclose(c.view_inverse[:3, 3], [1, 2, 3])
    assert approx_equal(1.0, 1.0000001) and not approx_equal(1.0, 1.00001)


def test_material_defaults_are_the_references():
    m = Material()  # src/shape/mod.rs:138-180
    assert (m.alpha, m.shininess, m.reflectivity, m.refraction_index, m.shadow_softness, m.roughness) == (1.0, 150.0, 0.0, 1.0, 0.01, 0.0)
    assert m.specular_color == (0.8, 0.8, 0.8) and m.cast_shadow and m.receive_shadow and m.monte_carlo and m.smooth_shading
    cfg = make_config()  # src/raytracing.rs:110-127
    assert (cfg.samples, cfg.max_recursion, cfg.monte_carlo, cfg.gamma_correction) == (1, 6, 0, 0)
    assert abs(cfg.fog_color[0] - 0.4) < 1e-7 and cfg.focal_length == 1.0 and cfg.aperture_size == 1.0


@pytest.mark.parametrize("w,h,tw,th,n", [(1280, 720, 32, 8, 8), (100, 37, 32, 8, 3), (7, 5, 8, 8, 2), (64, 64, 8, 8, 1)])
def test_region_pixels_partition(w, h, tw, th, n):
    seen = np.zeros((h, w), np.int32)
    for r in range(n):
        xy = region_pixels(w, h, tw, th, n, r)
        seen[xy[:, 1], xy[:, 0]] += 1
    assert (seen == 1).all()
    # first tile of rank 0 is row-major inside the tile
    xy = region_pixels(w, h, tw, th, n, 0)
    assert xy[0].tolist() == [0, 0] and xy[1].tolist() == ([1, 0] if min(tw, w) > 1 else [0, 1])


def test_synthetic_scenes_are_deterministic_and_sized():
    from rustray_amd import synthetic
    a, b = synthetic.sponza_syn(grid=4), synthetic.sponza_syn(grid=4)
    assert len(a.items) == len(b.items) and all(np.array_equal(x.trans, y.trans) for x, y in zip(a.items, b.items))
    full = synthetic.sponza_syn()
    assert len(full.items) > 50 and full.n_triangles_instanced() > 250000          # large BVH, top-level structure in use
    h = synthetic.helmet_syn()
    assert len(h.items) == 2 and 70000 < h.n_triangles_instanced() < 90000
    m = h.materials[h.items[1].material]
    assert m.texture[0] >= 0 and m.texture[3] >= 0 and m.texture[5] >= 0 and m.texture[6] >= 0
    lo = synthetic.lotus_syn(grid=4)
    assert lo.meta["config"]["aperture_size"] == 16.0 and lo.meta["config"]["focal_length"] == 20.0


@needs_reference
def test_gltf_loader_semantics():
    """monkey.glb through the glTF path (reference src/scene.rs:722-978)."""
    from rustray_amd.scene import load_scene as load_ref
    sc = load_ref("scene/models/monkey/monkey.glb", 320, 180, root=REF)
    fs = sc.flatten()
    assert [it.id for it in fs.items] == [3, 5]                 # lights take ids 1, 2; then (object, material) pairs
    assert len(fs.meshes[0].indices) == 15744 and len(fs.meshes[0].positions) == 3 * 15744   # de-indexed (:853-892)
    assert (fs.meshes[0].indices.reshape(-1) == np.arange(3 * 15744)).all()
    assert len(fs.lights) == 2 and fs.lights[0].intensity == 100.0 and fs.lights[1].intensity == 500.0   # point: intensity / 10 (:747)
    m = fs.materials[fs.items[0].material]
    assert abs(m.base_color[2] - 0.8) < 1e-6 and m.reflectivity == 0.0
    assert abs(m.roughness - 0.4 / (2 * math.pi)) < 1e-6           # (1/pi/2) * roughness_factor (:915)
    assert all(abs(s - 0.8 * b) < 1e-6 for s, b in zip(m.specular_color, m.base_color))
    assert abs(sc.cam.fov - 0.39959648) < 1e-6 and not sc.cam.is_default_cam()
    # world-space vertices: the node scale 3.024 is baked in, item transforms stay identity
    obj = load_ref("scene/models/monkey/monkey.obj", 64, 64, root=REF).flatten()
    assert np.allclose(fs.meshes[0].positions.max(0), obj.meshes[0].positions.max(0), rtol=1e-4)
    assert np.array_equal(fs.items[0].trans, np.eye(4, dtype=np.float32))
    uv = fs.meshes[1].uvs
    assert uv.min() >= -1e-6 and uv.max() <= 1.0 + 1e-6            # v flipped to 1 - v (:871)
    g = load_scene("monkey_glb")
    assert len(g.items) == 2 and np.array_equal(g.meshes[0].positions, fs.meshes[0].positions)


def test_animation_mirror():
    """Keyframe interpolation of reference src/animation.rs (the helmet turntable of scene/helmet.json:55-92)."""
    from rustray_amd.animation import Animation
    an = Animation.from_json({"fps": 25, "enabled": True, "keyframes": [
        {"time": 0, "objects": [{"name": "helmet", "transformation": {"rotation": {"x": -25.0, "y": 15.0, "z": 0.0},
                                                                      "scale": {"x": 1.25, "y": 1.25, "z": 1.25}, "translation": {"x": 0.3, "y": 0.2, "z": 0.0}}}]},
        {"time": 6000, "objects": [{"name": "helmet", "transformation": {"rotation": {"x": -25.0, "y": 375.0, "z": 0.0},
                                                                         "scale": {"x": 1.25, "y": 1.25, "z": 1.25}, "translation": {"x": 0.3, "y": 0.2, "z": 0.0}}}]}]})
    assert an.has_animation() and an.get_frames_amount_to_render() == 150
    first, last, f = an.get_keyframes_for_frame(75)
    assert (first.time, last.time) == (0, 6000) and abs(f - 0.5) < 1e-12
    m = an.get_trans_for_frame(75, "helmet")
    # T * Rz * Ry * Rx * S with ry = 195 deg, rx = -25 deg, s = 1.25
    ry, rx = math.radians(195.0), math.radians(-25.0)
    assert abs(m[0, 0] - 1.25 * math.cos(ry)) < 1e-5 and abs(m[1, 1] - 1.25 * math.cos(rx)) < 1e-5
    assert abs(m[0, 3] - 0.3) < 1e-6 and abs(m[1, 3] - 0.2) < 1e-6
    assert an.get_trans_for_frame(75, "nobody") is None
    assert not Animation.from_json({"fps": 25, "enabled": True, "keyframes": [{"time": 0, "objects": []}]}).has_animation()
    back = Animation.from_meta(an.to_meta())
    assert np.array_equal(back.get_trans_for_frame(10, "helmet"), an.get_trans_for_frame(10, "helmet"))


def test_bench_scene_file_is_real_when_its_assets_are_present_and_the_stand_in_otherwise(tmp_path, capsys):
    """bench.py --scene <scene file> --scene-root <tree>: a reference scene file is rendered for real when every
    asset it names is present under the root; when the .glb the reference would download is missing, the stand-in of the same
    name is rendered and labelled synthetic.  Nothing is downloaded either way."""
    import json
    import bench
    (tmp_path / "scene").mkdir()
    (tmp_path / "scene" / "mini.json").write_text(json.dumps({
        "name": "mini", "camera": {"pos": {"x": 0, "y": 0, "z": 5}, "fov": 60},
        "lights": [{"light_type": "point", "pos": {"x": 0, "y": 4, "z": 2}, "color": {"r": 1, "g": 1, "b": 1}, "intensity": 50.0}],
        "objects": [{"type": "sphere", "name": "ball", "radius": 1.0, "pos": {"x": 0, "y": 0, "z": 0}, "reflectivity": 0.3}]}))
    (tmp_path / "scene" / "sponza.json").write_text(json.dumps({
        "name": "Sponza", "objects": [{"name": "sponza", "type": "gltf", "url": "https://example.invalid/Sponza_fixed.glb",
                                      "path": "data/temp/Sponza_fixed.glb", "texture_filtering_nearest": True}]}))
    fs, cam, cfg = bench.build_workload("scene/mini.json", 64, 36, 2, 1, str(tmp_path))
    assert fs.meta["data"] == "real" and fs.name == "mini" and len(fs.items) == 1 and cam.width == 64 and cfg.samples == 2
    fs, cam, cfg = bench.build_workload("scene/sponza.json", 64, 36, 2, 1, str(tmp_path))
    assert fs.meta["data"] == "synthetic" and fs.name == "sponza_syn" and len(fs.items) > 50
    assert "Sponza_fixed.glb" in capsys.readouterr().err


def test_bench_prices_every_kernel_build_and_never_mixes_builds(monkeypatch):
    """bench.kernel_rooflines: `roofline.kernels` holds every kernel build of the frame with its live time; `frac` is quoted only against
    SQ counters of the SAME sources and workload (profiles/rNN_sq_counters*.json carry the source_id they were taken on), and the line
    says why when it is not."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(SCENES))
    import bench
    prof = json.load(open(os.path.join(os.path.dirname(SCENES), "profiles", "r04_sq_counters.json")))
    k1 = prof["kernels"]["k_trace_closest<true>"]
    acc = dict(primary_rays=3 * 117964800, secondary_rays=3 * 4.5e6, shadow_rays=3e8, shaded_hits=3e8, ms_trace_closest=3 * 6.6, ms_trace_shadow=3 * 6.0, ms_shade=3 * 6.6,
               launches_trace_closest=6, launches_trace_shadow=9, launches_shade=9, ms_total=58.0, ms_binning=0.0, binned_rays=0,
               ms_trace_closest_level1=3 * 6.2, launches_trace_closest_level1=3, ms_shade_level1=3 * 6.3, launches_shade_level1=6,
               ms_trace_shadow_level1=3 * 5.5, launches_trace_shadow_level1=6)
    roof = bench.kernel_rooflines("sponza_syn", 1280, 720, 128, acc, 3, prof["source_id"])
    assert roof["kernel"] == "k_trace_closest<true>" and roof["bound"] == "valu_issue" and abs(roof["avg_launch_ms"] - 6.2) < 1e-9
    assert abs(roof["frac"] - k1["SQ_INSTS_VALU"] / 6.2e-3 / 1e9 / bench.VALU_PEAK_GINST) < 1e-9 and 0.3 < roof["frac"] < 0.6
    assert set(roof["kernels"]) == {"k_trace_closest<true>", "k_shade<true>", "k_trace_shadow<true>", "k_trace_closest<false>", "k_shade<false>", "k_trace_shadow<false>"}
    for name, k in roof["kernels"].items():
        assert k["frac"] is not None and 0.0 < k["frac"] < 1.0 and k["ms_per_frame"] > 0.0, name
    assert abs(roof["kernels"]["k_trace_shadow<false>"]["ms_per_frame"] - 0.5) < 1e-9 and roof["kernels"]["k_shade<true>"]["launches_per_frame"] == 2
    # another build of the library: times are still reported, the instruction counts are not borrowed
    other = bench.kernel_rooflines("sponza_syn", 1280, 720, 128, acc, 3, "0123456789abcdef")
    assert other["frac"] is None and "re-run tools/profile_round.sh" in other["frac_reason"]
    assert all(k["frac"] is None and k["ms_per_frame"] > 0.0 for k in other["kernels"].values())
    # another workload of the same scene, and a scene nobody profiled
    assert bench.kernel_rooflines("sponza_syn", 640, 360, 128, acc, 3, prof["source_id"])["frac"] is None
    assert "no profiles" in bench.kernel_rooflines("kbert_room", 1280, 720, 128, acc, 3, prof["source_id"])["frac_reason"]
    assert bench.kernel_rooflines("sponza_syn", 1280, 720, 128, dict(acc, launches_trace_closest_level1=0), 3, prof["source_id"]) is None


def test_bench_waits_for_the_other_ranks_by_pid():
    """N > 1: before the one-process leg starts, rank 0 waits until the other ranks' processes are gone -- bounded, and the line says
    what happened (it used to sleep two seconds and hope)."""
    import subprocess
    import sys
    import time
    sys.path.insert(0, os.path.dirname(SCENES))
    import bench
    p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(0.6)"])
    t0 = time.perf_counter()
    r = bench.wait_for_ranks_to_leave([p.pid], limit_s=10.0)
    dt = time.perf_counter() - t0
    p.wait()
    assert r["ranks_gone"] and r["still_alive"] == [] and 0.3 < dt < 5.0   # a zombie (exited, not yet reaped) counts as gone
    q = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(5)"])
    r = bench.wait_for_ranks_to_leave([q.pid], limit_s=0.3)
    assert not r["ranks_gone"] and r["still_alive"] == [q.pid] and r["waited_s"] < 2.0
    q.kill(); q.wait()
    assert bench.wait_for_ranks_to_leave([], limit_s=1.0)["ranks_gone"]
