"""The device math layer must agree bit for bit with the host build of the same header, and
the hardware f64/f32 sqrt and divide must be correctly rounded (= host IEEE results)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from mp3common import ROOT

pytestmark = pytest.mark.gpu


def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dm") / "libdmath_host.so")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                    os.path.join(ROOT, "tests", "dmath_host.cpp")], check=True)
    return ctypes.CDLL(out)


def test_device_math_equals_host_build(product, tmp_path_factory):
    h = host_lib(tmp_path_factory)
    rng = np.random.default_rng(7)
    n = 1 << 20
    L = product.lib
    L.mp3mi_debug_dmath.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]

    def dev(fn, x, y=None):
        out = np.empty_like(x)
        rc = L.mp3mi_debug_dmath(fn, x.ctypes.data, y.ctypes.data if y is not None else None, out.ctypes.data, len(x))
        assert rc == 0
        return out

    def host1(name, x):
        out = np.empty_like(x)
        getattr(h, "t_dm_" + name)(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(len(x)))
        return out

    cases = {
        0: ("log", np.concatenate([np.exp(rng.uniform(-60, 60, n)), 1 + rng.uniform(-1e-3, 1e-3, n // 4)])),
        1: ("exp", rng.uniform(-60, 60, n)),
        2: ("sin", rng.uniform(-30, 30, n)),
        3: ("cos", rng.uniform(-30, 30, n)),
    }
    for fn, (name, x) in cases.items():
        d, hh = dev(fn, x), host1(name, x)
        assert np.array_equal(d.view(np.int64), hh.view(np.int64)), name
    a = rng.standard_normal(n).astype(np.float32).astype(np.float64)
    b = rng.standard_normal(n).astype(np.float32).astype(np.float64)
    out = np.empty_like(a)
    h.t_dm_atan2(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(n))
    assert np.array_equal(dev(4, a, b).view(np.int64), out.view(np.int64))
    # sincos halves equal the separate functions
    x = rng.uniform(-30, 30, n)
    assert np.array_equal(dev(9, x).view(np.int64), dev(2, x).view(np.int64))
    assert np.array_equal(dev(10, x).view(np.int64), dev(3, x).view(np.int64))
    # IEEE basics on the device: sqrt, divide (f64 and f32), and no FMA contraction
    xp = np.exp(rng.uniform(-100, 100, n))
    yp = np.exp(rng.uniform(-100, 100, n))
    assert np.array_equal(dev(5, xp), np.sqrt(xp))
    assert np.array_equal(dev(6, xp, yp), xp / yp)
    xf = rng.uniform(0, 1e6, n).astype(np.float32)
    yf = rng.uniform(1e-3, 1e3, n).astype(np.float32)
    assert np.array_equal(dev(7, xf.astype(np.float64)), np.sqrt(xf).astype(np.float64))
    assert np.array_equal(dev(8, xf.astype(np.float64), yf.astype(np.float64)), (xf / yf).astype(np.float64))
    u = rng.uniform(-2, 2, n)
    v = rng.uniform(-2, 2, n)
    assert np.array_equal(dev(11, u, v), u * v + 1.0)
    uf, vf = u.astype(np.float32), v.astype(np.float32)
    assert np.array_equal(dev(12, uf.astype(np.float64), vf.astype(np.float64)), (uf * vf + np.float32(1.0)).astype(np.float64))


def test_wave_reductions_on_device(product):
    """the DPP reductions / readlane helpers of mp3mi_dev.h against numpy, every lane active"""
    L = product.lib
    L.mp3mi_debug_dmath.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(5)
    n = 1 << 16
    x = rng.integers(0, 30000, n).astype(np.float64)
    lanes = np.repeat(rng.integers(0, 64, n // 64), 64).astype(np.float64)
    out = np.empty(n)
    for fn, ref in ((20, x.reshape(-1, 64).sum(1)), (21, x.reshape(-1, 64).max(1)),
                    (22, x.reshape(-1, 64)[np.arange(n // 64), lanes.reshape(-1, 64)[:, 0].astype(int)])):
        assert L.mp3mi_debug_dmath(fn, x.ctypes.data, lanes.ctypes.data, out.ctypes.data, n) == 0
        assert np.array_equal(out.reshape(-1, 64), np.repeat(ref[:, None], 64, 1)), fn
