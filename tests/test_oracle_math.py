"""Accuracy of the oracle's pinned transcendentals against float64 (they stand in for Rust's f32::sin etc.)."""
import ctypes as C

import numpy as np


def _ulp_err(got, want64):
    want = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-45)


def test_sincos_within_2ulp(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-50, 50, 50000), [0.0, np.pi / 4, -np.pi / 2]]).astype(np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    oracle.lib().rro_sincos(x.ctypes.data_as(C.c_void_p), len(x), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    x64 = x.astype(np.float64)
    # relative to ulp of the result, except near zeros of the function where absolute error matters
    es = np.abs(s - np.sin(x64)); ec = np.abs(c - np.cos(x64))
    assert es.max() < 2.5e-7 and ec.max() < 2.5e-7
    big = np.abs(np.sin(x64)) > 0.1
    assert _ulp_err(s[big], np.sin(x64[big])).max() <= 2.0
    big = np.abs(np.cos(x64)) > 0.1
    assert _ulp_err(c[big], np.cos(x64[big])).max() <= 2.0


def test_acos_within_2ulp(oracle):
    x = np.concatenate([np.random.default_rng(1).uniform(-1, 1, 200000), [-1.0, 1.0, 0.0, 0.5, -0.5]]).astype(np.float32)
    y = np.zeros_like(x)
    oracle.lib().rro_acos(x.ctypes.data_as(C.c_void_p), len(x), y.ctypes.data_as(C.c_void_p))
    want = np.arccos(x.astype(np.float64))
    ok = want > 1e-3
    assert _ulp_err(y[ok], want[ok]).max() <= 2.5
    assert np.abs(y - want).max() < 5e-7
    z = np.asarray([1.5, -1.5, np.nan], np.float32); out = np.zeros_like(z)
    oracle.lib().rro_acos(z.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p))
    assert np.isnan(out).all()


def test_atan2_within_2ulp_and_quadrants(oracle):
    rng = np.random.default_rng(2)
    yy = rng.uniform(-10, 10, 200000).astype(np.float32)
    xx = rng.uniform(-10, 10, 200000).astype(np.float32)
    out = np.zeros_like(xx)
    oracle.lib().rro_atan2(yy.ctypes.data_as(C.c_void_p), xx.ctypes.data_as(C.c_void_p), len(xx), out.ctypes.data_as(C.c_void_p))
    want = np.arctan2(yy.astype(np.float64), xx.astype(np.float64))
    assert np.abs(out - want).max() < 6e-7
    sp_y = np.asarray([0.0, 0.0, 1.0, -1.0, -0.0], np.float32); sp_x = np.asarray([1.0, -1.0, 0.0, 0.0, -1.0], np.float32)
    o = np.zeros(5, np.float32)
    oracle.lib().rro_atan2(sp_y.ctypes.data_as(C.c_void_p), sp_x.ctypes.data_as(C.c_void_p), 5, o.ctypes.data_as(C.c_void_p))
    assert np.allclose(o, [0.0, np.pi, np.pi / 2, -np.pi / 2, -np.pi], atol=1e-7)


def test_host_build_of_rr_cos_equals_the_oracles(oracle):
    """make_dmaterial evaluates jitter()'s cone bound cos(spread * pi) once per material on the HOST with the kernels' own rr_cos
    (rr_math.h, host + device).  Its host build against the oracle's cosine, bit for bit, over the spreads a material can hold."""
    import ctypes as C
    from rustray_amd import capi
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0.0, 1.0, 200000), rng.uniform(-4.0, 40.0, 50000), [0.0, 0.01, 0.015, 0.02, 0.5, 1.0, 1e-9]]).astype(np.float32)
    got = np.zeros_like(x)
    L = capi.lib()
    assert L.rr_math_probe(6, x.ctypes.data_as(C.c_void_p), None, None, len(x), got.ctypes.data_as(C.c_void_p), None, None, C.c_uint64(0), 0) == 0
    arg = (x * np.float32(np.pi)).astype(np.float32)   # the same single f32 multiply
    s, c = np.zeros_like(arg), np.zeros_like(arg)
    oracle.lib().rro_sincos(arg.ctypes.data_as(C.c_void_p), len(arg), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    assert np.array_equal(got.view(np.uint32), c.view(np.uint32))


def test_host_build_of_the_triangle_constants_is_the_ieee_sequence():
    """DTri::v1.w (area = |cross(a - b, a - c)|) and v3 (normalize(cross(b - a, c - a))) are evaluated on the host (rr_api.hip
    tri_shading_constants).  numpy's float32 arithmetic is the same correctly rounded IEEE sequence: bit for bit, NaN for NaN.
    (tests/test_gpu_math.py holds the device build to the host build.)"""
    from rustray_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(21)
    t = 30000
    a = rng.uniform(-3.0, 3.0, (t, 3)); b = a + rng.uniform(-1.0, 1.0, (t, 3)); c = a + rng.uniform(-1.0, 1.0, (t, 3))
    k = t // 5
    c[:k] = a[:k] + (b[:k] - a[:k]) * rng.uniform(0.0, 2.0, (k, 1)) + rng.uniform(-1e-6, 1e-6, (k, 3))
    s = 10.0 ** rng.uniform(-12.0, 9.0, (k, 1))
    a[k:2 * k] *= s; b[k:2 * k] *= s; c[k:2 * k] *= s
    c[2 * k:2 * k + 40] = b[2 * k:2 * k + 40]
    a, b, c = (np.ascontiguousarray(v, np.float32) for v in (a, b, c))
    ng, area = np.zeros(3 * t, np.float32), np.zeros(3 * t, np.float32)
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    assert L.rr_math_probe(11, p(a), p(b), p(c), 3 * t, p(ng), p(area), None, C.c_uint64(0), 0) == 0

    def cross(u, v):
        return np.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2], u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], axis=1)

    def norm(u):
        return np.sqrt((u[:, 0] * u[:, 0] + u[:, 1] * u[:, 1]) + u[:, 2] * u[:, 2])
    with np.errstate(all="ignore"):
        want_area = norm(cross(a - b, a - c))
        x = cross(b - a, c - a)
        want_ng = x / norm(x)[:, None]
    assert want_area.dtype == np.float32 and want_ng.dtype == np.float32
    assert np.array_equal(area.reshape(t, 3)[:, 0].view(np.uint32), want_area.view(np.uint32))
    got = ng.reshape(t, 3)
    same = (got.view(np.uint32) == want_ng.view(np.uint32)) | (np.isnan(got) & np.isnan(want_ng))
    assert same.all()
    assert np.isnan(got[2 * k:2 * k + 40]).all()
