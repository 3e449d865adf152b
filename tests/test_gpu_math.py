"""The arithmetic contract on the device: gfx950 evaluates rr_math.h bit-identically to the oracle's
oracle_math.h (sin/cos/acos/atan2 sequences, IEEE divide and sqrt, jitter)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_transcendentals_bit_exact(hip, oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 100000), rng.uniform(-100, 100, 20000), [0.0, -0.0, 1e-20, 3.14159274]]).astype(np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    oracle.lib().rro_sincos(_p(x), len(x), _p(s), _p(c))
    gs, gc, _ = hip.math_probe(0, x)
    assert np.array_equal(gs.view(np.uint32), s.view(np.uint32)) and np.array_equal(gc.view(np.uint32), c.view(np.uint32))
    u = np.concatenate([rng.uniform(-1, 1, 100000), [-1.0, 1.0, 0.5, -0.5, 0.0, 1.5]]).astype(np.float32)
    a = np.zeros_like(u)
    oracle.lib().rro_acos(_p(u), len(u), _p(a))
    ga, _, _ = hip.math_probe(1, u)
    assert np.array_equal(ga.view(np.uint32), a.view(np.uint32))
    yy, xx = rng.uniform(-5, 5, 100000).astype(np.float32), rng.uniform(-5, 5, 100000).astype(np.float32)
    t = np.zeros_like(xx)
    oracle.lib().rro_atan2(_p(yy), _p(xx), len(xx), _p(t))
    gt, _, _ = hip.math_probe(2, yy, xx)
    assert np.array_equal(gt.view(np.uint32), t.view(np.uint32))


def test_divide_and_sqrt_are_correctly_rounded(hip):
    rng = np.random.default_rng(4)
    a = (rng.standard_normal(200000) * 10 ** rng.uniform(-20, 20, 200000)).astype(np.float32)
    b = (rng.standard_normal(200000) * 10 ** rng.uniform(-20, 20, 200000)).astype(np.float32)
    b[b == 0] = 1.0
    with np.errstate(over="ignore", under="ignore"):
        want = (a / b).astype(np.float32)
    got, _, _ = hip.math_probe(3, a, b)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    p = np.abs(a)
    got, _, _ = hip.math_probe(4, p)
    assert np.array_equal(got.view(np.uint32), np.sqrt(p).astype(np.float32).view(np.uint32))


def test_jitter_bit_exact(hip, oracle):
    rng = np.random.default_rng(6)
    n = 5000
    d = rng.standard_normal((n, 3)).astype(np.float32)
    seed = 0x1234567890
    gx, gy, gz = hip.math_probe(5, d[:, 0].copy(), d[:, 1].copy(), d[:, 2].copy(), seed=seed)
    out = np.zeros(3, np.float32)
    for i in range(0, n, 37):
        oracle.lib().rro_jitter(_p(d[i].copy()), 0.05, seed, i, i & 7, 1 + i % 5, i % 3, _p(out))
        assert (out[0], out[1], out[2]) == (gx[i], gy[i], gz[i])


def test_host_and_device_builds_of_rr_cos_agree(hip):
    """DMaterial::cos_* (host) is used where jitter() computed cos(spread * pi) per call (device): the two builds, bit for bit."""
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(0.0, 1.0, 200000), rng.uniform(-4.0, 40.0, 50000), [0.0, 0.01, 0.015, 0.02, 0.5, 1.0]]).astype(np.float32)
    host, _, _ = hip.math_probe(6, x)
    dev, _, _ = hip.math_probe(7, x)
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))


def test_host_and_device_builds_of_the_triangle_constants_agree(hip):
    """DTri::v1.w (area) and v3 (the flat normal) are evaluated once per triangle on the host where k_shade evaluated them per hit:
    the two builds of the same IEEE sequence, bit for bit -- ordinary, sliver, huge, tiny and degenerate triangles."""
    rng = np.random.default_rng(14)
    t = 60000
    a = rng.uniform(-3.0, 3.0, (t, 3)); b = a + rng.uniform(-1.0, 1.0, (t, 3)); c = a + rng.uniform(-1.0, 1.0, (t, 3))
    k = t // 6
    c[:k] = a[:k] + (b[:k] - a[:k]) * rng.uniform(0.0, 2.0, (k, 1)) + rng.uniform(-1e-6, 1e-6, (k, 3))    # slivers
    s = 10.0 ** rng.uniform(-12.0, 9.0, (k, 1))
    a[k:2 * k] *= s; b[k:2 * k] *= s; c[k:2 * k] *= s                                                  # tiny and huge
    c[2 * k:2 * k + 50] = b[2 * k:2 * k + 50]                                                          # zero area: 0 / 0
    a[2 * k + 50:2 * k + 60] = 0.0; b[2 * k + 50:2 * k + 60] = 0.0; c[2 * k + 50:2 * k + 60] = 0.0
    fa, fb, fc = (np.ascontiguousarray(v, np.float32).reshape(-1) for v in (a, b, c))
    dn, da, _ = hip.math_probe(10, fa, fb, fc)
    hn, ha, _ = hip.math_probe(11, fa, fb, fc)
    assert np.array_equal(dn.view(np.uint32), hn.view(np.uint32))
    assert np.array_equal(da.view(np.uint32), ha.view(np.uint32))
    assert np.isnan(hn.reshape(-1, 3)[2 * k:2 * k + 60]).all()   # the degenerate ones: NaN on both sides, as in the reference
