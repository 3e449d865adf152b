"""Parity of the HIP trace loop with the oracle, through the C ABI (run with -m gpu on an MI355X).

Bar (BASELINE.json north_star): pixel RGB within +-1 LSB of the fixed-seed CPU reference; object ids
equal; depth / normal within 1e-4 relative (they are f32 sums taken in a different order).
"""
import math
import os

import numpy as np
import pytest

from rustray_amd.flat import make_config, rr_region
from tests.golden.make_golden import CASES
from tests.helpers import GOLDEN, camera_for, compare_frames, load_scene

pytestmark = pytest.mark.gpu


def assert_parity(out, ref, what=""):
    r = compare_frames(out, ref)
    assert r["alpha_ok"], what
    assert r["n_rgb_over"] == 0, f"{what}: {r}"
    assert r["n_id_diff"] == 0, f"{what}: {r}"
    assert r["nan_mismatch"] == 0, f"{what}: {r}"
    assert r["max_depth_rel"] < 1e-4 and r["max_normal_abs"] < 1e-4, f"{what}: {r}"
    return r


SMALL = [  # scene, w, h, spp, mc, seed
    ("spheres", 256, 256, 1, False, 0),        # BASELINE C1 exactly
    ("spheres", 96, 96, 4, True, 5),
    ("monkey", 200, 150, 4, True, 7),          # BASELINE C2 geometry/material at reduced size
    ("kbert", 160, 90, 2, True, 1),            # spot light, flat shading, base texture, auto camera
    ("earth_room", 160, 90, 2, True, 2),       # 5 lights, textured sphere, planes
    ("spheres_room", 160, 90, 2, True, 3),
    ("monkey_room", 128, 72, 2, True, 4),
    ("monkey_glb", 160, 90, 3, True, 6),
    ("kbert_room", 160, 90, 2, True, 8),       # two meshes inside a textured room, 5 lights
    ("earth", 128, 128, 2, True, 9),           # sphere with base + specular + normal maps, auto camera
    ("floor", 160, 90, 2, True, 10),           # one textured plane (bump map as normal map), 4 lights       # glTF path: de-indexed world-space mesh, glTF lights and camera, roughness jitter
]


@pytest.mark.parametrize("name,w,h,spp,mc,seed", SMALL)
def test_fixture_scenes_match_oracle(hip, oracle, name, w, h, spp, mc, seed):
    fs = load_scene(name)
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=spp, monte_carlo=mc, seed=seed)
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=16, want_counters=True)
    assert_parity(out, ref, name)
    c = ref["counters"]
    # identical path trees: same number of closest-hit rays and shaded hits as the recursive reference
    assert st["primary_rays"] == c["rays_primary"] and st["secondary_rays"] == c["rays_secondary"]
    assert st["shaded_hits"] == c["shaded_hits"]
    assert st["shadow_rays"] <= c["rays_shadow"]  # zero-weight shadow rays are not traced on the device


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames(hip, name):
    c = CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    fs = load_scene(c["scene"])
    cam = camera_for(fs, c["w"], c["h"]).c_struct()
    cfg = make_config(samples=c["spp"], monte_carlo=c["mc"], seed=c["seed"])
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    if c["window"]:
        x0, y0, x1, y1 = c["window"]
        out = {k: v[y0:y1, x0:x1] for k, v in out.items()}
    assert_parity(out, {k: g[k] for k in g.files}, name)


@pytest.mark.parametrize("which", ["sponza_syn", "helmet_syn", "lotus_syn"])
def test_synthetic_stand_ins_match_oracle(hip, oracle, which):
    """C3-C5 stand-ins at reduced size: top-level structure (> 50 items), nearest and bilinear maps, normal /
    roughness / AO maps, glass + reflective floor, depth of field."""
    from rustray_amd import synthetic
    from rustray_amd.camera import Camera
    fs = {"sponza_syn": lambda: synthetic.sponza_syn(grid=6), "helmet_syn": synthetic.helmet_syn,
          "lotus_syn": lambda: synthetic.lotus_syn(grid=6)}[which]()
    st = dict(fs.meta["camera"]); st["width"], st["height"] = 160, 90
    cam = Camera.from_state(st).c_struct()
    cd = fs.meta["config"]
    cfg = make_config(samples=4, monte_carlo=True, seed=12, focal_length=cd.get("focal_length", 1.0), aperture_size=cd.get("aperture_size", 1.0))
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=16)
    assert_parity(out, ref, which)


def test_config_variants(hip, oracle):
    """fog, gamma, max_recursion 0 / 2, directional light, disabled light, no shadows, flip normals."""
    fs = load_scene("spheres_room")
    fs.lights[0].light_type = 0          # directional
    fs.lights[0].dir = (0.3, -1.0, -0.2)
    fs.lights[0].intensity = 0.8
    fs.lights[1].enabled = False
    fs.materials[fs.items[2].material].receive_shadow = False
    fs.materials[fs.items[3].material].cast_shadow = False
    fs.materials[fs.items[3].material_cache].cast_shadow = False
    fs.items[8].flip_normals = True
    cam = camera_for(fs, 128, 72).c_struct()
    for kw in (dict(fog_density=0.02, fog_color=(0.2, 0.3, 0.5)), dict(gamma_correction=True), dict(max_recursion=0),
               dict(max_recursion=2), dict(monte_carlo=False), dict(samples=1)):
        args = dict(samples=3, monte_carlo=True, seed=21)
        args.update(kw)
        cfg = make_config(**args)
        with hip.DeviceScene(fs, 0) as ds:
            out = ds.render(cam, cfg)
        assert_parity(out, oracle.render(fs.c_struct(), cam, cfg, n_threads=16), str(kw))


def test_explicit_sample_table_equals_builtin(hip):
    fs = load_scene("spheres")
    cam = camera_for(fs, 64, 64).c_struct()
    cfg = make_config(samples=8, monte_carlo=True, seed=1)
    xy, _ = hip.sample_table(8)
    with hip.DeviceScene(fs, 0) as ds:
        a = ds.render(cam, cfg)
        b = ds.render(cam, cfg, sample_xy=xy)
        c = ds.render(cam, cfg, sample_xy=xy[::-1].copy())  # a different table must change MC pixels
    assert (a["rgba"] == b["rgba"]).all()
    assert (a["rgba"] != c["rgba"]).any()


def test_batching_and_chunking_do_not_change_a_single_bit(hip):
    """Fixed-point accumulation: any batch size / shade chunk gives the same bits (depth and normals too)."""
    fs = load_scene("monkey_room")
    cam = camera_for(fs, 96, 54).c_struct()
    cfg = make_config(samples=6, monte_carlo=True, seed=33)
    outs = []
    stats = []
    for budget_mb, chunk in ((16384, 0), (4, 65536), (1, 65536), (0, 65536)):
        with hip.DeviceScene(fs, 0) as ds:
            ds.set_tuning(queue_budget_bytes=max(budget_mb << 20, 1), shade_chunk_rays=chunk)  # rr_tuning: 0 would mean automatic
            outs.append(ds.render(cam, cfg))
            stats.append(ds.stats())
    assert stats[0]["batches"] == 1 and stats[-1]["batches"] > 4
    for o in outs[1:]:
        assert (o["rgba"] == outs[0]["rgba"]).all() and np.array_equal(o["depth"], outs[0]["depth"])
        assert np.array_equal(o["normal"], outs[0]["normal"], equal_nan=True) and (o["object_id"] == outs[0]["object_id"]).all()


def test_branching_scene_in_a_small_ray_arena_is_sliced_depth_first(hip):
    """Every hit on glass spawns two children; with a ray arena of a few thousand rays the deeper levels do not fit
    behind their parents at once, so levels are shaded in slices whose subtrees finish first (rr_api.hip run_level).
    Same bits as the unconstrained frame."""
    from rustray_amd.flat import FlatScene, Item, Light, Material
    from rustray_amd.scene import Scene
    fs = FlatScene()
    eye4 = np.eye(4, dtype=np.float32)
    for i, (x, y, z, r) in enumerate([(0, 0, -6, 2.0), (2.5, 0.5, -8, 2.0), (-2.5, -0.5, -8, 2.0), (0, 2.5, -9, 2.0), (0, -3, -7, 2.0)]):
        m = Material(base_color=(0.2 + 0.15 * i, 0.5, 0.9 - 0.15 * i), alpha=0.4, reflectivity=0.5, refraction_index=1.4)
        fs.materials.append(m); fs.materials.append(Scene._cache_of(m))
        t = eye4.copy(); t[:3, 3] = (x, y, z)
        ti = eye4.copy(); ti[:3, 3] = (-x, -y, -z)
        fs.items.append(Item(kind=0, id=i + 1, material=2 * i, material_cache=2 * i + 1, radius=r, trans=t, trans_inv=ti,
                             bbox_min=(-r, -r, -r), bbox_max=(r, r, r), name=f"glass{i}"))
    fs.lights = [Light(pos=(3.0, 6.0, 2.0), intensity=150.0)]
    fs.meta = {"camera": dict(width=64, height=64, fov=float(np.float32(np.radians(70.0))), eye_pos=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0],
                              dir=[0.0, 0.0, -1.0], clipping_near=0.1, clipping_far=100.0)}
    cam = camera_for(fs, 96, 96).c_struct()
    cfg = make_config(samples=2, monte_carlo=False, seed=1, max_recursion=6)
    with hip.DeviceScene(fs, 0) as ds:
        first = ds.render(cam, cfg)
        st_first = ds.stats()
        for _ in range(3):   # a frame that had to slice doubles the arena of the next one
            ref = ds.render(cam, cfg)
        st0 = ds.stats()
    assert st_first["sliced_levels"] > 0 and np.array_equal(first["rgba"], ref["rgba"])
    with hip.DeviceScene(fs, 0) as ds:
        ds.set_tuning(queue_budget_bytes=1, shade_chunk_rays=65536)
        out = ds.render(cam, cfg)
        st = ds.stats()
    assert st0["sliced_levels"] == 0 and st["sliced_levels"] > 0 and st["batches"] > 1
    assert st["secondary_rays"] == st0["secondary_rays"] > 3 * st["primary_rays"]
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(out["normal"], ref["normal"], equal_nan=True)


def test_tiled_regions_reassemble_bit_exactly(hip):
    """rr_render_region_device for 3 ranks + rr_deinterleave_device == rr_render of the whole frame."""
    import torch
    from rustray_amd.renderer import TiledFrame, render_region_torch
    fs = load_scene("spheres")
    w, h = 100, 61
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=3, monte_carlo=True, seed=8)
    with hip.DeviceScene(fs, 0) as ds:
        whole = ds.render(cam, cfg)
        world = 3
        parts = []
        for r in range(world):
            tf = TiledFrame(w, h, r, world, 32, 8)
            parts.append(render_region_torch(ds, cam, cfg, tf, aux=True))
        torch.cuda.synchronize()
        tf0 = TiledFrame(w, h, 0, world, 32, 8)
        for key, eb in (("rgba", 4), ("normal", 12), ("depth", 4), ("object_id", 4)):
            cat = torch.cat([p[key].reshape(p[key].shape[0], -1) for p in parts], dim=0).contiguous()
            frame = torch.empty((h * w, cat.shape[1]), dtype=cat.dtype, device=cat.device)
            hip.deinterleave_device(w, h, 32, 8, world, eb, cat.data_ptr(), frame.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = frame.cpu().numpy().reshape(whole[key].shape)
            ref_idx = cat.index_select(0, tf0._frame_index(cat.device)).cpu().numpy().reshape(whole[key].shape)
            assert np.array_equal(got, ref_idx, equal_nan=True)             # device kernel == torch index path
            assert np.array_equal(got.view(np.uint8), whole[key].view(np.uint8)), key  # == single full-frame render, bit for bit


def test_pick_matches_oracle(hip, oracle):
    fs = load_scene("spheres")
    cam = camera_for(fs, 256, 256).c_struct()
    with hip.DeviceScene(fs, 0) as ds:
        for x, y in ((128, 128), (5, 5), (80, 128), (185, 150), (40, 80)):
            a, b = ds.pick(cam, x, y), oracle.pick(fs.c_struct(), cam, x, y)
            assert (a.hit, a.object_id, a.item_index) == (b.hit, b.object_id, b.item_index)
            assert a.distance == b.distance


def test_update_transforms_equals_new_scene(hip):
    """rr_scene_update_transforms (animation frames) == creating the scene with those transforms."""
    from rustray_amd.scene import get_transformation, inverse_affine
    fs = load_scene("monkey_room")
    cam = camera_for(fs, 96, 54).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=4)
    trans = np.stack([it.trans for it in fs.items]).copy()
    trans[0] = get_transformation(trans[0], (0.5, 0.2, -1.0), (1.1, 1.1, 1.1), (0.0, 0.6, 0.0))
    inv = np.stack([inverse_affine(t) for t in trans])
    with hip.DeviceScene(fs, 0) as ds:
        before = ds.render(cam, cfg)
        ds.update_transforms(trans, inv)
        moved = ds.render(cam, cfg)
    for it, t, ti in zip(fs.items, trans, inv):
        it.trans, it.trans_inv = t, ti
    with hip.DeviceScene(fs, 0) as ds2:
        fresh = ds2.render(cam, cfg)
    assert (moved["rgba"] == fresh["rgba"]).all() and (moved["rgba"] != before["rgba"]).any()


def test_edge_inputs(hip, oracle):
    """1x1 and ragged frames, an empty scene, a scene whose only mesh has two triangles, samples not a power of two."""
    fs = load_scene("spheres")
    for w, h, spp in ((1, 1, 1), (3, 2, 5), (65, 7, 3)):
        cam = camera_for(fs, w, h).c_struct()
        cfg = make_config(samples=spp, monte_carlo=True, seed=2)
        with hip.DeviceScene(fs, 0) as ds:
            out = ds.render(cam, cfg)
        assert_parity(out, oracle.render(fs.c_struct(), cam, cfg), f"{w}x{h}")
    from rustray_amd.flat import FlatScene
    empty = FlatScene(); empty.meta = fs.meta
    cam = camera_for(fs, 16, 8).c_struct()
    with hip.DeviceScene(empty, 0) as ds:
        out = ds.render(cam, make_config(samples=2))
    assert (out["rgba"][..., :3] == 0).all() and (out["rgba"][..., 3] == 255).all() and (out["object_id"] == 0).all()
    assert np.isnan(out["normal"]).all() and (out["depth"] == 0).all()


def test_errors_do_not_abort(hip):
    fs = load_scene("spheres")
    cam = camera_for(fs, 16, 16).c_struct()
    with hip.DeviceScene(fs, 0) as ds:
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=0))
        assert e.value.code == -1
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render(cam, make_config(samples=1, max_recursion=40))
        assert e.value.code == -2
        with pytest.raises(hip.RustrayHipError):
            ds.pick(cam, 99, 0)
        out = ds.render(cam, make_config(samples=1))  # the handle is still usable
        assert out["rgba"].shape == (16, 16, 4)


def test_animation_frames_match_oracle(hip, oracle):
    """A keyframed turntable (reference src/animation.rs, Scene::apply_frame): every frame rendered through
    rr_scene_update_transforms on ONE resident scene equals the oracle on a scene built with that frame's matrices."""
    from rustray_amd.animation import Animation, Frame, Keyframe
    fs = load_scene("monkey_room")
    name = fs.items[0].name
    an = Animation(True, 5, [Keyframe(0, [Frame(name, (0.0, 0.0, -10.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))]),
                             Keyframe(1000, [Frame(name, (1.0, 0.5, -9.0), (0.0, math.pi, 0.3), (1.4, 1.4, 1.4))])])
    assert an.get_frames_amount_to_render() == 5
    cam = camera_for(fs, 96, 54).c_struct()
    cfg = make_config(samples=2, monte_carlo=True, seed=9)
    with hip.DeviceScene(fs, 0) as ds:
        for frame in (0, 2, 4):
            trans, inv = an.frame_transforms(fs, frame)
            ds.update_transforms(trans, inv)
            out = ds.render(cam, cfg)
            for it, t, ti in zip(fs.items, trans, inv):
                it.trans, it.trans_inv = t, ti
            assert_parity(out, oracle.render(fs.c_struct(), cam, cfg, n_threads=16), f"frame {frame}")


def _blocker_scene():
    """Corner of the reference's shadow semantics (src/raytracing.rs:466-487, :886-892): candidates are tried in
    bbox-distance order and the FIRST item that is hit decides.  Item Y (an occluder quad plus a far triangle that
    stretches Y's box past the light, so the shadow origin lies INSIDE Y's box and Y's key is the box EXIT distance)
    would shadow the floor, but the sphere X behind the light has a smaller key (its box entry), is hit beyond the
    light and therefore makes the receiver lit."""
    from rustray_amd.flat import FlatScene, Item, Light, Material, MeshData
    from rustray_amd.scene import Scene
    fs = FlatScene()

    def add_mat(m):
        fs.materials.append(m); fs.materials.append(Scene._cache_of(m))
        return len(fs.materials) - 2, len(fs.materials) - 1
    eye4 = np.eye(4, dtype=np.float32)
    floor = MeshData(positions=np.asarray([[-20, 0, 20], [20, 0, 20], [20, 0, -20], [-20, 0, -20]], np.float32),
                     indices=np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32))
    # Y: a 2x2 occluder at height 3 above the origin region, and a far sliver at height 40 that extends the box
    ypos = np.asarray([[-1, 3, 1], [1, 3, 1], [1, 3, -1], [-1, 3, -1], [30, 40, 30], [31, 40, 30], [30, 40, 31],
                       [-30, -1, -30], [-31, -1, -30], [-30, -1, -31]], np.float32)
    ymesh = MeshData(positions=ypos, indices=np.asarray([[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9]], np.uint32))
    fs.meshes = [floor, ymesh]
    m0, c0 = add_mat(Material(base_color=(0.8, 0.8, 0.8)))
    m1, c1 = add_mat(Material(base_color=(0.9, 0.2, 0.2)))
    m2, c2 = add_mat(Material(base_color=(0.2, 0.9, 0.2)))
    fs.items = [Item(kind=1, id=3, material=m0, material_cache=c0, mesh=0, trans=eye4.copy(), trans_inv=eye4.copy(),
                     bbox_min=(-20, 0, -20), bbox_max=(20, 0, 20), name="floor"),
                Item(kind=1, id=6, material=m1, material_cache=c1, mesh=1, trans=eye4.copy(), trans_inv=eye4.copy(),
                     bbox_min=tuple(ypos.min(0)), bbox_max=tuple(ypos.max(0)), name="Y"),
                Item(kind=0, id=9, material=m2, material_cache=c2, radius=2.0, name="X",
                     trans=np.asarray([[1, 0, 0, 0], [0, 1, 0, 12], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32),
                     trans_inv=np.asarray([[1, 0, 0, 0], [0, 1, 0, -12], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32),
                     bbox_min=(-2, -2, -2), bbox_max=(2, 2, 2))]
    fs.lights = [Light(pos=(0.0, 8.0, 0.0), color=(1.0, 1.0, 1.0), intensity=60.0, light_type=1)]
    fs.meta = {"camera": dict(width=64, height=64, fov=float(np.float32(np.radians(60.0))), eye_pos=[0.0, 6.0, 9.0], up=[0.0, 1.0, 0.0],
                              dir=[0.0, -0.6, -1.0], clipping_near=0.1, clipping_far=100.0)}
    return fs


def test_shadow_blocker_beyond_the_light(hip, oracle):
    fs = _blocker_scene()
    cam = camera_for(fs, 96, 96).c_struct()
    cfg = make_config(samples=1, monte_carlo=False)
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8)
    # the semantics under test really occur: removing X darkens floor pixels under the occluder
    fs2 = _blocker_scene(); fs2.items[2].visible = False
    ref_without_x = oracle.render(fs2.c_struct(), cam, cfg, n_threads=8)
    floor_px = ref["object_id"] == 3
    assert (ref["rgba"][floor_px].astype(int).sum(-1) > ref_without_x["rgba"][floor_px].astype(int).sum(-1) + 30).sum() > 20
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
    assert_parity(out, ref, "blocker beyond the light")


def test_progressive_tiles_fill_the_frame_with_final_pixels(hip):
    """rr_render_progressive_tiles: the reference's own kind of progress (shuffled 2x2 cells, each rendered with all of its samples,
    src/renderer.rs:125-172): every pass adds an interleaved subset of 32x8 tiles, a pixel that has appeared never changes again, the
    finished frame and its statistics equal rr_render's, and the callback can stop the frame."""
    from rustray_amd.renderer import region_pixels
    fs = load_scene("spheres_room")
    w, h = 200, 75   # 7 x 10 tiles, the right column and the bottom row clipped
    cam = camera_for(fs, w, h).c_struct()
    cfg = make_config(samples=4, monte_carlo=True, seed=4)
    with hip.DeviceScene(fs, 0) as ds:
        ref = ds.render(cam, cfg)
        ref_stats = ds.stats()
        snaps = []

        def on_pass(out, done, total):
            snaps.append((done, total, out["rgba"].copy(), out["object_id"].copy()))
            return False
        P = 5
        out = ds.render_progressive(cam, cfg, on_pass, min_passes=P, tiles=True)
        st = ds.stats()
        for k in ("rgba", "normal", "depth", "object_id"):
            assert np.array_equal(out[k], ref[k], equal_nan=True), k
        for k in ("primary_rays", "secondary_rays", "shadow_rays", "shaded_hits"):
            assert st[k] == ref_stats[k], k
        assert len(snaps) == P - 1 and all(t == w * h * 4 for _, t, _, _ in snaps)
        covered = np.zeros((h, w), bool)
        for k, (done, _, rgba, ids) in enumerate(snaps):
            xy = region_pixels(w, h, 32, 8, P, k)
            covered[xy[:, 1], xy[:, 0]] = True
            assert done == int(covered.sum()) * 4
            assert np.array_equal(rgba[covered], ref["rgba"][covered]) and np.array_equal(ids[covered], ref["object_id"][covered])   # final when they appear
            assert not rgba[~covered].any()                                                                                     # not rendered yet
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render_progressive(cam, cfg, lambda out, done, total: True, min_passes=P, tiles=True)
        assert e.value.code == -6
        assert np.array_equal(ds.render(cam, cfg)["rgba"], ref["rgba"])


def test_progressive_passes_refine_towards_the_one_shot_frame(hip):
    """rr_render_progressive (the reference shows the frame filling in while it renders, src/run.rs:506-545): every pass
    reports more samples, previews are whole frames, the finished frame is bit-identical to rr_render's, and a callback
    can stop the frame (RendererManager::stop, src/renderer.rs:174-198)."""
    fs = load_scene("spheres_room")
    cam = camera_for(fs, 96, 64).c_struct()
    cfg = make_config(samples=16, monte_carlo=True, seed=4)
    with hip.DeviceScene(fs, 0) as ds:
        ref = ds.render(cam, cfg)
        seen, errs = [], []

        def on_pass(out, done, total):
            seen.append((done, total))
            # a preview is the mean over the finished sample slices: already close to the final frame
            errs.append(float(np.abs(out["rgba"][..., :3].astype(np.int32) - ref["rgba"][..., :3].astype(np.int32)).mean()))
            assert out["rgba"][..., 3].min() == 255
            return False
        out = ds.render_progressive(cam, cfg, on_pass, min_passes=4)
        for k in ("rgba", "normal", "depth", "object_id"):
            assert np.array_equal(out[k], ref[k]), k
        assert len(seen) >= 3 and all(t == 96 * 64 * 16 for _, t in seen)
        assert [d for d, _ in seen] == sorted(set(d for d, _ in seen)) and all(d % (96 * 64) == 0 for d, _ in seen)
        assert errs[-1] <= errs[0] and errs[-1] < 12.0
        # stopping after the first pass keeps that preview and reports the cancellation
        with pytest.raises(hip.RustrayHipError) as e:
            ds.render_progressive(cam, cfg, lambda out, done, total: True, min_passes=4)
        assert e.value.code == -6
        # the scene is still usable
        assert np.array_equal(ds.render(cam, cfg)["rgba"], ref["rgba"])


def test_animation_run_equals_frame_by_frame_updates(hip):
    """AnimationRun (frame loop of src/run.rs:421-465 on one resident scene) gives the frames that separate
    update_transforms + render calls give; RendererManager.start(on_pass=...) can stop a frame early."""
    from rustray_amd.animation import Animation, Frame, Keyframe
    from rustray_amd.renderer import AnimationRun, Raytracing, RendererManager
    fs = load_scene("monkey_room")
    name = fs.items[0].name
    an = Animation(True, 3, [Keyframe(0, [Frame(name, (0.0, 0.0, -10.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))]),
                             Keyframe(1000, [Frame(name, (1.0, 0.5, -9.0), (0.0, 2.0, 0.3), (1.4, 1.4, 1.4))])])
    rt = Raytracing(fs, camera_for(fs, 64, 48))
    rt.apply_config(samples=2, monte_carlo=True, seed=2)
    try:
        run = AnimationRun(rt, an)
        assert run.frames == [0, 1, 2]
        frames = run.render()
        for f in run.frames:
            rt.device_scene.update_transforms(*an.frame_transforms(fs, f))
            assert np.array_equal(frames[f]["rgba"], rt.render_frame()["rgba"]), f
        assert not np.array_equal(frames[0]["rgba"], frames[2]["rgba"])
        mgr = RendererManager(64, 48, rt)
        rt.apply_config(samples=16)
        calls = []
        mgr.start(on_pass=lambda m: (calls.append(m.get_rendered_pixels()), m.stop()), min_passes=4)
        assert len(calls) == 1 and 0 < calls[0] < 64 * 48 and not mgr.is_done() and mgr.image is not None
        mgr.start()
        assert mgr.is_done()
    finally:
        rt.close()


def test_cancel_flag_stops_a_frame(hip):
    """`cancel` of rr_render (RendererManager::stop in the reference, src/renderer.rs:174-198): a flag that is already set
    ends the call with RR_ERR_CANCELLED and the scene stays usable."""
    import ctypes as C
    from rustray_amd.flat import rr_frame
    fs = load_scene("spheres")
    cam = camera_for(fs, 64, 64).c_struct()
    cfg = make_config(samples=4, monte_carlo=True, seed=1)
    with hip.DeviceScene(fs, 0) as ds:
        rgba = np.zeros((64, 64, 4), np.uint8)
        fr = rr_frame(rgba.ctypes.data, None, None, None)
        flag = C.c_int(1)
        rc = hip.lib().rr_render(ds._h, C.byref(cam), C.byref(cfg), None, C.byref(fr), C.byref(flag))
        assert rc == -6 and b"cancel" in hip.lib().rr_last_error()
        flag.value = 0
        assert hip.lib().rr_render(ds._h, C.byref(cam), C.byref(cfg), None, C.byref(fr), C.byref(flag)) == 0
        assert np.array_equal(rgba, ds.render(cam, cfg, aux=False)["rgba"])


def test_binned_deeper_levels_do_not_change_a_single_bit(hip):
    """rr_tuning::bin_min_rays: deeper levels re-ordered by (origin cell, direction octant) before they are traced
    (bounds -> histogram -> prefix -> scatter).  Order never changes the frame: fixed-point accumulators."""
    fs = load_scene("monkey_room")
    cam = camera_for(fs, 160, 90).c_struct()
    cfg = make_config(samples=8, monte_carlo=True, seed=21)
    with hip.DeviceScene(fs, 0) as ds:
        ref = ds.render(cam, cfg)
        st0 = ds.stats()
        ds.set_tuning(bin_min_rays=1)
        out = ds.render(cam, cfg)
        st = ds.stats()
    # levels whose sorted copy does not fit behind them in the arena stay in spawn order
    assert st0["binned_rays"] == 0 and st["secondary_rays"] >= st["binned_rays"] > 100000
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(out["normal"], ref["normal"], equal_nan=True)
    for k in ("primary_rays", "secondary_rays", "shadow_rays", "shaded_hits"):
        assert st[k] == st0[k], k
