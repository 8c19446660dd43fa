"""CPU build of the product's math layer: correctly rounded against mpmath on a sample, and
within 1 ulp of glibc with a mismatch rate of the order of glibc's own misrounding rate."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from mp3common import ROOT


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dm") / "libdmath_host.so")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                    os.path.join(ROOT, "tests", "dmath_host.cpp")], check=True)
    return ctypes.CDLL(out)


def call1(dm, name, x):
    y = np.empty_like(x)
    getattr(dm, "t_dm_" + name)(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(y.ctypes.data), ctypes.c_size_t(len(x)))
    return y


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


@pytest.mark.parametrize("name,lo,hi", [("log", None, None), ("exp", -50, 50), ("sin", -25, 25), ("cos", -25, 25)])
def test_unary_against_glibc_and_mpmath(dm, name, lo, hi):
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.default_rng(11)
    n = 200000
    x = np.exp(rng.uniform(-40, 40, n)) if name == "log" else rng.uniform(lo, hi, n)
    y = call1(dm, name, x)
    libm = ctypes.CDLL("libm.so.6")
    f = getattr(libm, name)
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double]
    g = np.array([f(float(v)) for v in x[:50000]])
    d = ulps(y[:50000], g)
    assert d.max() <= 1
    assert (d != 0).mean() < 5e-3
    cr = np.array([float(getattr(mp, name)(mp.mpf(float(v)))) for v in x[:4000]])
    assert (ulps(y[:4000], cr) != 0).sum() == 0


def test_atan2_against_glibc_and_mpmath(dm):
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.default_rng(12)
    n = 50000
    a = (rng.standard_normal(n) * rng.choice([1e-3, 1, 1e3], n)).astype(np.float32).astype(np.float64)
    b = (rng.standard_normal(n) * rng.choice([1e-3, 1, 1e3], n)).astype(np.float32).astype(np.float64)
    y = np.empty_like(a)
    dm.t_dm_atan2(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(y.ctypes.data), ctypes.c_size_t(n))
    g = np.arctan2(a, b)
    assert ulps(y, g).max() <= 1
    cr = np.array([float(mp.atan2(mp.mpf(float(u)), mp.mpf(float(v)))) for u, v in zip(a[:4000], b[:4000])])
    assert (ulps(y[:4000], cr) != 0).sum() == 0
    for u, v in [(0.0, 1.0), (0.0, -1.0), (-0.0, 1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (2.0, -0.0)]:
        o, ua, va = np.empty(1), np.array([u]), np.array([v])
        dm.t_dm_atan2(ctypes.c_void_p(ua.ctypes.data), ctypes.c_void_p(va.ctypes.data), ctypes.c_void_p(o.ctypes.data), ctypes.c_size_t(1))
        assert o[0] == np.arctan2(u, v) and np.signbit(o[0]) == np.signbit(np.arctan2(u, v))


def test_log_exp_special_values(dm):
    x = np.array([1.0, 0.0, np.inf, 2.0 ** -1060, 1e300])
    y = call1(dm, "log", x)
    assert y[0] == 0.0 and y[1] == -np.inf and y[2] == np.inf
    assert abs(y[3] - np.log(2.0 ** -1060)) <= abs(np.log(2.0 ** -1060)) * 2 ** -52
    e = call1(dm, "exp", np.array([0.0, 800.0, -800.0, 1.0]))
    assert e[0] == 1.0 and e[1] == np.inf and e[2] == 0.0 and e[3] == np.e


def test_log_fast_error_bound(dm):
    """dm_log_fast (k_prep's first tier) stays within its documented absolute error of the
    correctly rounded dm_log: 2^-50 * max(1, |log x|)."""
    rng = np.random.default_rng(13)
    x = np.concatenate([np.exp(rng.uniform(-60, 60, 1 << 20)), 1.0 + rng.uniform(-1e-3, 1e-3, 1 << 16),
                        np.ldexp(1.0 + np.arange(128) / 128.0, 3)])
    exact, fast = call1(dm, "log", x), call1(dm, "log_fast", x)
    err = np.abs(fast - exact)
    assert (err <= 2.0 ** -50 * np.maximum(1.0, np.abs(exact))).all()
    d = ulps(fast[np.abs(exact) > 1e-2], exact[np.abs(exact) > 1e-2])
    assert d.max() <= 2


def test_atan2_fast_error_bound_and_rounding_guard(dm):
    """dm_atan2_fast (k_cw's first tier) stays within 2^-50 relative of the correctly rounded dm_atan2,
    and dm_float_rounding_safe rejects every argument whose float rounding the two tiers could disagree on."""
    rng = np.random.default_rng(14)
    n = 1 << 20
    a = (rng.standard_normal(n) * rng.choice([1e-6, 1e-3, 1.0, 1e3, 1e6], n)).astype(np.float32).astype(np.float64)
    b = (rng.standard_normal(n) * rng.choice([1e-6, 1e-3, 1.0, 1e3, 1e6], n)).astype(np.float32).astype(np.float64)
    ok = (a != 0) & (b != 0)
    a, b = a[ok], b[ok]
    n = len(a)
    exact, fast = np.empty(n), np.empty(n)
    dm.t_dm_atan2(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(exact.ctypes.data), ctypes.c_size_t(n))
    dm.t_dm_atan2_fast(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(fast.ctypes.data), ctypes.c_size_t(n))
    assert (np.abs(fast - exact) <= 2.0 ** -50 * np.abs(exact)).all()
    dm.t_dm_float_rounding_safe.argtypes = [ctypes.c_double]
    differ = np.nonzero(fast.astype(np.float32) != exact.astype(np.float32))[0]
    for i in differ:  # wherever the float roundings differ the guard must have said "not safe"
        assert dm.t_dm_float_rounding_safe(float(fast[i])) == 0
    safe = np.array([dm.t_dm_float_rounding_safe(float(v)) for v in fast[:20000]])
    assert safe.mean() > 0.999  # and it almost never cries wolf
    # a float midpoint itself and its neighbourhood are rejected, plain floats accepted
    f = np.float32(0.7853982)
    mid = (float(f) + float(np.nextafter(f, np.float32(1)))) / 2
    assert dm.t_dm_float_rounding_safe(mid) == 0 and dm.t_dm_float_rounding_safe(float(f)) == 1


def test_exp_fast_error_bound(dm):
    """dm_exp_fast (k_psy's first tier) stays within 2^-50 relative of the correctly rounded dm_exp."""
    rng = np.random.default_rng(15)
    x = np.concatenate([rng.uniform(-60, 60, 1 << 20), rng.uniform(-8, 0, 1 << 18), np.arange(-64, 65) * (np.log(2) / 64)])
    exact, fast = call1(dm, "exp", x), call1(dm, "exp_fast", x)
    assert (np.abs(fast - exact) <= 2.0 ** -50 * exact).all()


def test_sincos_fast_error_bound(dm):
    """dm_sincos_fast (k_cw's first tier) stays within 2^-51 absolute of the correctly rounded dm_sin / dm_cos
    over the phases it is given: floats in [-pi, pi] and 2 phi0 - phi2 in [-3 pi, 3 pi]."""
    rng = np.random.default_rng(16)
    ph = rng.uniform(-np.pi, np.pi, 1 << 20).astype(np.float32).astype(np.float64)
    pp = 2.0 * rng.uniform(-np.pi, np.pi, 1 << 20).astype(np.float32).astype(np.float64) - rng.uniform(-np.pi, np.pi, 1 << 20).astype(np.float32).astype(np.float64)
    k = np.arange(-6, 7) * (np.pi / 2)
    edge = np.concatenate([k, np.nextafter(k, 10), np.nextafter(k, -10), k + np.pi / 4, np.array([0.0, 1e-30, -1e-30, 1e-9])])
    x = np.concatenate([ph, pp, edge])
    for name in ("sin", "cos"):
        exact, fast = call1(dm, name, x), call1(dm, name + "_fast", x)
        assert np.abs(fast - exact).max() <= 2.0 ** -51


def test_sincos_fast_on_the_arguments_k_cw_really_has(dm):
    """dm_sincos_fast is only ever called with phi2 = a float phase in [-pi, pi] and phi' = 2 phi0 - phi2 (|.| <= 3 pi),
    both float-derived (k_fft.hip, cw_record).  Instead of random doubles: (a) float phases on a stride through ALL
    floats of [0, pi] by their bit patterns (every 509th: ~2.1 M), both signs, with the phi' they form with another one;
    (b) a dense grid around every multiple of pi/2 up to 3 pi, where the argument reduction cancels most; (c) the
    floats next to those multiples.  The bound part_cw_safe builds on is 2^-51 absolute (it assumes twelve times that)."""
    import mpmath
    rng = np.random.default_rng(23)
    hi = int(np.float32(np.pi).view(np.int32))
    pos = np.arange(0, hi + 1, 509, dtype=np.int64).astype(np.int32).view(np.float32).astype(np.float64)
    phases = np.concatenate([pos, -pos])
    partner = rng.permutation(phases)
    args = [phases, 2.0 * partner - phases]
    edges = []
    for k in range(-6, 7):
        c = k * np.pi / 2
        args.append(c + np.linspace(-2.0 ** -9, 2.0 ** -9, 20001))
        f = np.float32(c)
        near = np.array([np.nextafter(f, np.float32(np.inf)), f, np.nextafter(f, np.float32(-np.inf))], dtype=np.float32).astype(np.float64)
        edges += [near, 2.0 * near - near[::-1]]
    x = np.concatenate(args + edges)
    x = x[np.abs(x) <= 3 * np.pi + 1e-6]
    for name in ("sin", "cos"):
        exact, fast = call1(dm, name, x), call1(dm, name + "_fast", x)
        assert np.abs(fast - exact).max() <= 2.0 ** -51, name
    # the judge itself (dm_sin / dm_cos, correctly rounded) against mpmath where cancellation is worst
    mpmath.mp.prec = 200
    e = np.concatenate(edges)
    got = call1(dm, "sin", e)
    worst = max(abs(float(mpmath.sin(mpmath.mpf(float(v))) - mpmath.mpf(float(g)))) / max(abs(float(g)), 2.0 ** -1000) for v, g in zip(e, got))
    assert worst <= 2.0 ** -53, worst


def test_cos_fast_alone_error_bound(dm):
    """dm_cos_fast (the one cosine of k12_psy's first tier): |error| < 2^-51 absolute against mpmath over the arguments the
    kernel has -- |phi - phi'| <= 4 pi --, densely around every multiple of pi / 2 (where the cosine vanishes and only an
    ABSOLUTE bound can hold), near 0, and out to 1000"""
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(-13, 13, 40000), (np.arange(-8, 9)[:, None] * np.pi / 2 + rng.uniform(-1e-3, 1e-3, (17, 600))).reshape(-1),
                        rng.uniform(-1e-6, 1e-6, 500), rng.uniform(-1000, 1000, 4000), np.array([0.0, np.pi, -np.pi, 4 * np.pi])])
    y = call1(dm, "cos_only_fast", x)
    err = max(abs(float(mp.cos(mp.mpf(float(a))) - mp.mpf(float(b)))) for a, b in zip(x, y))
    assert err < 2.0 ** -51, err
    assert call1(dm, "cos_only_fast", np.array([0.0]))[0] == 1.0


def test_sin_fast_rel_error_bound(dm):
    """dm_sin_fast_rel (the one sine of k12_psy's first tier): RELATIVE error < 2^-49 against mpmath wherever it does not
    flag its argument -- uniformly over [-2 pi, 2 pi], for differences of floats halved (what the kernel passes), for tiny
    arguments down to 2^-60, and near every multiple of pi down to 2^-29.5 away, inside of which it must flag"""
    import mpmath as mp
    mp.mp.prec = 300
    rng = np.random.default_rng(22)
    parts = [rng.uniform(-6.4, 6.4, 12000)]
    for k in range(-2, 3):
        for e in (-3, -8, -15, -22, -28, -29.5, -33):
            parts.append(k * np.pi + rng.uniform(-1, 1, 200) * 2.0 ** e)
    parts.append(rng.uniform(-1, 1, 600) * 2.0 ** rng.uniform(-60, -1, 600))
    a = rng.uniform(-np.pi, np.pi, 6000).astype(np.float32).astype(np.float64)
    b = rng.uniform(-3 * np.pi, 3 * np.pi, 6000).astype(np.float32).astype(np.float64)
    parts.append((a - b) / 2)
    x = np.concatenate(parts)
    y = call1(dm, "sin_fast_rel", x)  # (NaN where the function flags its argument)
    flagged = np.isnan(y)
    k = np.rint(x / np.pi)
    near = (k != 0) & (np.abs(x - k * np.pi) < 2.0 ** -31)
    assert flagged[near].all() and not flagged[np.abs(x - k * np.pi) > 2.0 ** -29].any()
    worst = 0.0
    for xi, yi in zip(x[~flagged], y[~flagged]):
        t = mp.sin(mp.mpf(float(xi)))
        if t == 0:
            assert yi == 0
            continue
        worst = max(worst, float(abs((mp.mpf(float(yi)) - t) / t)))
    assert worst < 2.0 ** -49, worst
