"""Hand-derived unit cases for the restated primitives (parry3d 0.13 formulas, reference
src/raytracing.rs helpers).  Expected values are derived by hand from the formulas the oracle cites."""
import ctypes as C
import math

import numpy as np
import pytest

from rustray_amd.flat import rr_texture

F = np.float32


def _v(*a):
    return np.asarray(a, np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ray_aabb(oracle, mins, maxs, o, d, solid):
    toi = C.c_float(0)
    hit = oracle.lib().rro_ray_aabb(_p(_v(*mins)), _p(_v(*maxs)), _p(_v(*o)), _p(_v(*d)), int(solid), C.byref(toi))
    return (bool(hit), toi.value)


def test_aabb_outside_hit_and_miss(oracle):
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), True) == (True, 4.0)
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), False) == (True, 4.0)
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, 1), True)[0] is False  # pointing away
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (3, 0, 5), (0, 0, -1), True)[0] is False  # parallel, outside slab


def test_aabb_origin_inside_solid_vs_not(oracle):
    # origin inside: tmin == 0 -> solid returns 0, non-solid returns the exit distance
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 0), (0, 0, -1), True) == (True, 0.0)
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 0), (0, 0, -1), False) == (True, 1.0)


def test_aabb_unnormalised_direction_scales_toi(oracle):
    # get_inverse_ray does not renormalise (src/shape/mod.rs:755-761): toi is in units of |dir|
    assert ray_aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -2), True) == (True, 2.0)


def ray_tri(oracle, a, b, c, o, d):
    toi, n, back = C.c_float(0), np.zeros(3, np.float32), C.c_int(0)
    hit = oracle.lib().rro_ray_triangle(_p(_v(*a)), _p(_v(*b)), _p(_v(*c)), _p(_v(*o)), _p(_v(*d)), C.byref(toi), _p(n), C.byref(back))
    return bool(hit), toi.value, n.tolist(), back.value


def test_triangle_front_and_back(oracle):
    tri = ((0, 0, 0), (1, 0, 0), (0, 1, 0))  # n = +z
    hit, toi, n, back = ray_tri(oracle, *tri, (0.25, 0.25, 1), (0, 0, -1))
    assert hit and toi == 1.0 and n == [0.0, 0.0, 1.0] and back == 0
    hit, toi, n, back = ray_tri(oracle, *tri, (0.25, 0.25, -2), (0, 0, 1))
    assert hit and toi == 2.0 and n == [-0.0, -0.0, -1.0] and back == 1  # two-sided, normal faces the ray


def test_triangle_edges_parallel_and_behind(oracle):
    tri = ((0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert ray_tri(oracle, *tri, (0.5, 0.5, 1), (0, 0, -1))[0]        # on the hypotenuse: v + w == d is accepted
    assert ray_tri(oracle, *tri, (0, 0, 1), (0, 0, -1))[0]            # on a vertex
    assert not ray_tri(oracle, *tri, (0.75, 0.75, 1), (0, 0, -1))[0]  # outside
    assert not ray_tri(oracle, *tri, (0.25, 0.25, 1), (1, 0, 0))[0]   # parallel: d == 0
    assert not ray_tri(oracle, *tri, (0.25, 0.25, 1), (0, 0, 1))[0]   # plane behind the origin


def ray_ball(oracle, r, o, d, solid):
    toi, n = C.c_float(0), np.zeros(3, np.float32)
    hit = oracle.lib().rro_ray_ball(C.c_float(r), _p(_v(*o)), _p(_v(*d)), int(solid), C.byref(toi), _p(n))
    return bool(hit), toi.value, n.tolist()


def test_ball_outside_inside_tangent(oracle):
    assert ray_ball(oracle, 1.0, (0, 0, 5), (0, 0, -1), True) == (True, 4.0, [0.0, 0.0, 1.0])
    # inside: solid -> toi 0, otherwise the far root and the inward normal
    hit, toi, n = ray_ball(oracle, 2.0, (0, 0, 0.5), (0, 0, -1), False)
    assert hit and toi == 2.5 and n == [-0.0, -0.0, 1.0]
    hit, toi, _ = ray_ball(oracle, 2.0, (0, 0, 0.5), (0, 0, -1), True)
    assert hit and toi == 0.0
    assert not ray_ball(oracle, 1.0, (0, 0, 5), (0, 0, 1), True)[0]      # c > 0 and b > 0
    assert not ray_ball(oracle, 1.0, (2, 0, 5), (0, 0, -1), True)[0]     # misses
    hit, toi, _ = ray_ball(oracle, 1.0, (1, 0, 5), (0, 0, -1), True)     # tangent: delta == 0
    assert hit and toi == 5.0


def test_wrap(oracle):
    w = oracle.lib().rro_wrap
    assert w(0.0, 8) == 0 and w(0.5, 8) == 4 and w(0.999, 8) == 7
    assert w(1.0, 8) == 0 and w(1.25, 8) == 2          # wraps
    assert w(-0.25, 8) == 6 and w(-1.0, 8) == 0         # negative: (-2 % 8) + 8, (-8 % 8) == 0
    assert w(-0.01, 8) == 0                             # (-0.08 as i32) == 0
    assert w(float("nan"), 8) == 0


def _tex(arr):
    a = np.ascontiguousarray(arr, np.uint8)
    return a, rr_texture(a.shape[1], a.shape[0], a.ctypes.data)


def test_bilinear(oracle):
    # 2x2 texture, red channel 0, 255 / 255, 0
    arr, t = _tex([[[0, 0, 0, 255], [255, 0, 0, 255]], [[255, 0, 0, 255], [0, 0, 0, 255]]])
    out = np.zeros(4, np.float32)
    f = oracle.lib().rro_tex_interpolate
    f(C.byref(t), 0.0, 0.0, _p(out)); assert out.tolist() == [0.0, 0.0, 0.0, 1.0]
    f(C.byref(t), 0.25, 0.0, _p(out)); assert out[0] == 0.5          # x = 0.5 between texel 0 and 1
    f(C.byref(t), 0.5, 0.0, _p(out)); assert out[0] == 1.0           # x = 1.0 exactly: floor == ceil == 1
    f(C.byref(t), 0.75, 0.0, _p(out)); assert out[0] == 1.0          # x1 = ceil(1.5) = 2 clamps to 1 (no wrap)
    f(C.byref(t), -0.25, 0.0, _p(out)); assert out[0] == 1.0         # negative: += width once -> x = 1.5
    f(C.byref(t), 0.25, -0.75, _p(out)); assert out[0] == 0.5        # y = -1.5 + 2 = 0.5: average of both rows at x=.5
    f(C.byref(t), 5.0, 5.0, _p(out)); assert out[0] == 0.0           # beyond 1: saturates to the last texel (1,1)


def test_fresnel(oracle):
    fr = oracle.lib().rro_fresnel
    n = _v(0, 0, 1)
    # normal incidence from outside; cos_i = |cos_t| as the reference writes it (src/raytracing.rs:557-558)
    kr = fr(_p(_v(0, 0, -1)), _p(n), 1.5)
    assert abs(kr - 0.04) < 1e-6
    assert fr(_p(_v(0, 0, -1)), _p(n), 1.0) == 0.0
    # total internal reflection: from inside (i.n > 0) at a grazing angle
    d = _v(math.sin(1.2), 0, math.cos(1.2))
    assert fr(_p(d), _p(n), 1.5) == 1.0


def test_transmission(oracle):
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    t = oracle.lib().rro_transmission
    assert t(_p(_v(0, 0, 1)), _p(_v(0, 0, -1)), _p(_v(0, 0, 0)), 1.5, _p(o), _p(d)) == 1
    assert d.tolist() == [0.0, 0.0, -1.0] and abs(o[2] + 0.001) < 1e-9   # origin pushed below the surface
    g = _v(math.sin(1.2), 0, math.cos(1.2))
    assert t(_p(_v(0, 0, 1)), _p(g), _p(_v(0, 0, 0)), 1.5, _p(o), _p(d)) == 0  # TIR: None


def test_jitter_cone_and_determinism(oracle):
    j = oracle.lib().rro_jitter
    d = _v(0.3, -0.5, 0.8)
    dn = d / np.linalg.norm(d)
    out, out2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
    spread = 0.05
    for i in range(200):
        j(_p(d), spread, 42, i, i % 7, 1 + i % 5, i % 3, _p(out))
        assert abs(np.linalg.norm(out) - 1.0) < 1e-6
        assert float(out @ dn) >= math.cos(spread * math.pi) - 1e-6
    j(_p(d), spread, 42, 5, 1, 1, 0, _p(out)); j(_p(d), spread, 42, 5, 1, 1, 0, _p(out2))
    assert out.tolist() == out2.tolist()
    j(_p(d), spread, 43, 5, 1, 1, 0, _p(out2))
    assert out.tolist() != out2.tolist()
    j(_p(d), 0.0, 42, 5, 1, 1, 0, _p(out))
    assert out.tolist() == d.tolist()  # spread <= 0 returns dir unchanged (not normalised)


def test_approx_equal(oracle):
    ae = oracle.lib().rro_approx_equal
    assert ae(1.0, 1.0000001) and not ae(1.0, 1.00001) and ae(0.0, 0.0000004) and not ae(0.0, 0.000002)


@pytest.mark.parametrize("samples", [1, 16])
def test_primary_ray_quirks(oracle, samples):
    """Appendix A 1-3: origin on the z = -1 view plane, no perspective divide, offsets from the pixel centre."""
    from rustray_amd.camera import Camera
    from rustray_amd.flat import make_config
    cam = Camera(); cam.clipping_near, cam.clipping_far = 0.1, 100.0; cam.init(4, 4)
    cfg = make_config(samples=samples)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    oracle.lib().rro_primary_ray(C.byref(cam.c_struct()), C.byref(cfg), 1, 2, 0, 0, _p(o), _p(d))
    # fov 90, aspect 1: proj_inv scales by tan(45 deg) = 1; sensor = ((1.5/4)*2-1, 1-(2.5/4)*2) = (-0.25, -0.25)
    assert np.allclose(o, [-0.25, -0.25, -1.0], atol=1e-6) and np.allclose(d, [-0.25, -0.25, -1.0], atol=1e-6)
