"""The packet form of the top level tests closest-hit packets against the items' SURFACE boxes (the world box of a mesh's
transformed vertices: much smaller than the box of its turned local box) and shadow packets against the boxes of the local
boxes' corners, whose entry distances bound the shadow query's order (rr_api.hip build_tlas, exact_world_box).  A scene of turned
instances, a ball whose arithmetic overflows (ray_ball answers Some(NaN) for every ray through its LOCAL box: it must keep the
corner box) and a stretched one, against the oracle in both of its candidate forms; then the same after rr_scene_update_transforms."""
import math

import numpy as np
import pytest

from rustray_amd.flat import Item, Material, RR_ITEM_SPHERE, make_config
from tests.helpers import camera_for, compare_frames

pytestmark = pytest.mark.gpu


def _turn(fs, turn_seed):
    """Turns the objects about all three axes (the generator turns them about y only), from the generator's own placement."""
    from rustray_amd.scene import get_transformation, inverse_affine
    rng = np.random.default_rng(turn_seed)
    for it in fs.items:
        if it.name.startswith(("monkey_", "kbert_")):
            if not hasattr(it, "_placed"):
                it._placed = it.trans.copy()
            rot = tuple(float(np.float32(math.radians(v))) for v in rng.uniform(-60.0, 60.0, 3))
            it.trans = get_transformation(it._placed, (0.0, float(rng.uniform(0.0, 2.0)), 0.0), (1.0, 1.0, 1.0), rot)
            it.trans_inv = inverse_affine(it.trans)


def _turned(seed, extra_balls=True):
    from rustray_amd.scene import get_transformation, inverse_affine
    from rustray_amd.synthetic import _add_material, sponza_syn
    fs = sponza_syn(grid=5, seed=seed)
    _turn(fs, seed)
    if extra_balls:
        mi, ci = _add_material(fs, Material(base_color=(0.9, 0.2, 0.2), reflectivity=0.3))
        # a ball of radius 1e12 under a transform that scales it to 1 unit: local coordinates ~1e12, b * b overflows in ray_toi_with_ball
        s = np.float32(1e-12)
        t = get_transformation(np.eye(4, dtype=np.float32), (3.0, 1.0, -12.0), (float(s),) * 3, (0.3, 0.4, 0.5))
        fs.items.append(Item(kind=RR_ITEM_SPHERE, id=9001, material=mi, material_cache=ci, radius=1e12, trans=t, trans_inv=inverse_affine(t),
                             bbox_min=(-1e12,) * 3, bbox_max=(1e12,) * 3, name="overflowing_ball"))
        t = get_transformation(np.eye(4, dtype=np.float32), (-4.0, 1.5, -14.0), (2.5, 0.6, 1.2), (0.7, 0.2, 1.1))
        fs.items.append(Item(kind=RR_ITEM_SPHERE, id=9004, material=mi, material_cache=ci, radius=1.0, trans=t, trans_inv=inverse_affine(t),
                             bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3, name="stretched_ball"))
    return fs


def _check(out, ref, what):
    res = compare_frames(out, ref)
    assert res["n_rgb_over"] == 0 and res["n_id_diff"] == 0 and res["nan_mismatch"] == 0, (what, res)


@pytest.mark.parametrize("seed", [41, 42])
def test_turned_instances_against_the_oracle(hip, oracle, seed):
    fs = _turned(seed)
    assert 17 <= len(fs.items) <= 512   # the packet form of the top level is in use
    cam = camera_for(fs, 72, 40).c_struct()
    cfg = make_config(samples=64, monte_carlo=True, seed=seed, max_recursion=3)   # 64 samples: a packet is one pixel
    with hip.DeviceScene(fs, 0) as ds:
        out = ds.render(cam, cfg)
        st = ds.stats()
    ref = oracle.render(fs.c_struct(), cam, cfg, n_threads=8, want_counters=True, brute_force=True)   # every item is a candidate
    _check(out, ref, "all-items form")
    c = ref["counters"]
    assert (st["primary_rays"], st["secondary_rays"], st["shaded_hits"]) == (c["rays_primary"], c["rays_secondary"], c["shaded_hits"])
    assert st["shadow_rays"] > 0 and st["secondary_rays"] > 0


def test_turned_instances_after_a_transform_update(hip, oracle):
    """The surface boxes are rebuilt from the meshes' vertices when the transforms change."""
    a = _turned(43, extra_balls=False)
    cam = camera_for(a, 64, 36).c_struct()
    cfg = make_config(samples=64, monte_carlo=True, seed=5, max_recursion=2)
    before = [(it.trans.copy(), it.trans_inv.copy()) for it in a.items]
    _turn(a, 44)   # other turns of the same objects
    trans = np.stack([it.trans for it in a.items])
    inv = np.stack([it.trans_inv for it in a.items])
    for it, (t, ti) in zip(a.items, before):
        it.trans, it.trans_inv = t, ti
    with hip.DeviceScene(a, 0) as ds:
        first = ds.render(cam, cfg)
        ds.update_transforms(trans, inv)
        moved = ds.render(cam, cfg)
    for it, t, ti in zip(a.items, trans, inv):
        it.trans, it.trans_inv = t, ti
    with hip.DeviceScene(a, 0) as ds2:
        fresh = ds2.render(cam, cfg)
    assert (moved["rgba"] != first["rgba"]).any()
    for k in ("rgba", "depth", "object_id"):
        assert np.array_equal(moved[k], fresh[k]), k
    _check(moved, oracle.render(a.c_struct(), cam, cfg, n_threads=8, brute_force=True), "after the update")
