"""include/pt_math.h and include/pt_rng.h, probed through the oracle library (host build of the same
headers the kernels include).  No GPU involved.

pt_math.h: accuracy against float64 references (the functions stand in for the platform libm the
reference calls; <= ~2 ulp keeps the substitution far below the image tolerance).
pt_rng.h:  Philox4x32-10 known-answer vectors (Random123 kat_vectors), and the reference's
distributions (rand 0.8.5 Standard / Uniform, rand_distr 0.4.3 UnitSphere).
"""
import math

import numpy as np
import pytest

from tests.oracle import FN_ACOS, FN_ASIN, FN_CBRT, FN_EXP, FN_LOG, FN_POW1P5, FN_PPF, FN_SIN


def ulp_err(got, want64):
    got = np.asarray(got, dtype=np.float32)
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny))
    return np.abs(got.astype(np.float64) - want64) / ulp


@pytest.mark.parametrize("fn,ref,xs,max_ulp", [
    (FN_EXP, np.exp, np.concatenate([np.linspace(-87, 88, 4001), np.linspace(-1, 1, 2001)]), 2.0),
    (FN_LOG, np.log, np.concatenate([np.geomspace(1e-37, 1e37, 4001), np.linspace(0.5, 2.0, 2001)]), 2.0),
    (FN_SIN, np.sin, np.linspace(-20, 20, 8001), 2.5),
    (FN_CBRT, np.cbrt, np.concatenate([np.geomspace(1e-30, 1e30, 3001), -np.geomspace(1e-3, 1e3, 301)]), 2.0),
    (FN_POW1P5, lambda x: x ** 1.5, np.geomspace(1e-6, 1e6, 3001), 2.0),
])
def test_pt_math_accuracy(oracle, fn, ref, xs, max_ulp):
    xs = xs.astype(np.float32)
    got = oracle.math(fn, xs)
    err = ulp_err(got, ref(xs.astype(np.float64)))
    assert err.max() <= max_ulp, f"max error {err.max():.2f} ulp at x={xs[err.argmax()]}"


def test_pt_acos_asin_accuracy(oracle):
    xs = np.concatenate([np.linspace(-1, 1, 8001), [1.0, -1.0, 0.0, 0.5, -0.5, 0.999999, -0.999999]]).astype(np.float32)
    for fn, ref in ((FN_ACOS, np.arccos), (FN_ASIN, np.arcsin)):
        got = oracle.math(fn, xs).astype(np.float64)
        want = ref(xs.astype(np.float64))
        # absolute bound: near acos(x)=0 / asin(x)=0 a relative ulp bound is not meaningful
        assert np.abs(got - want).max() <= 4e-7 * max(1.0, np.abs(want).max())
    assert oracle.math(FN_ACOS, [1.0])[0] == 0.0
    assert np.isnan(oracle.math(FN_ACOS, [1.0000001, -1.5, np.nan])).all()


def test_pt_math_special_values(oracle):
    inf = np.float32(np.inf)
    assert oracle.math(FN_EXP, [0.0])[0] == 1.0
    assert oracle.math(FN_EXP, [-np.inf, -200.0]).tolist() == [0.0, 0.0]
    assert oracle.math(FN_EXP, [np.inf, 100.0]).tolist() == [inf, inf]
    assert np.isnan(oracle.math(FN_EXP, [np.nan])[0])
    assert oracle.math(FN_LOG, [1.0])[0] == 0.0 and oracle.math(FN_LOG, [0.0])[0] == -inf
    assert np.isnan(oracle.math(FN_LOG, [-1.0])[0])
    assert oracle.math(FN_CBRT, [0.0, 8.0, -27.0, 1.0]).tolist() == [0.0, 2.0, -3.0, 1.0]
    assert np.isnan(oracle.math(FN_POW1P5, [-1.0])[0]) and oracle.math(FN_POW1P5, [4.0])[0] == 8.0
    assert oracle.math(FN_SIN, [0.0])[0] == 0.0


def test_norm_ppf(oracle):
    from scipy.stats import norm
    ps = np.concatenate([np.linspace(1e-6, 1 - 1e-6, 4001), [(0 + 0.5) * 2.0 ** -23, 1 - 0.5 * 2.0 ** -23]]).astype(np.float32)
    got = oracle.math(FN_PPF, ps).astype(np.float64)
    want = norm.ppf(ps.astype(np.float64))
    assert np.abs(got - want).max() <= 2e-5 * np.maximum(1.0, np.abs(want)).max()
    assert oracle.math(FN_PPF, [0.5])[0] == 0.0


# ---- Philox4x32-10, Random123 known-answer vectors ---------------------------------------------

@pytest.mark.parametrize("ctr,key,want", [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
])
def test_philox_kat(oracle, ctr, key, want):
    assert tuple(oracle.philox(ctr, key)) == want


def test_stream_is_a_pure_function_of_seed_pixel_sample_index(oracle):
    a, ia = oracle.rng_draw(7, 123, 5, 0, 16)
    # word i of the stream = Philox(counter = (i >> 2, pixel, sample, domain), key = seed)[i & 3]
    blk = oracle.philox((2, 123, 5, 0x70747263), (7, 0))
    assert a[8:12].tolist() == blk and ia == 16
    b, _ = oracle.rng_draw(7, 123, 6, 0, 16)
    c, _ = oracle.rng_draw(7, 124, 5, 0, 16)
    d, _ = oracle.rng_draw(8, 123, 5, 0, 16)
    for other in (b, c, d):
        assert not np.array_equal(a, other)
    e, _ = oracle.rng_draw((1 << 32) | 7, 123, 5, 0, 4)     # high seed word is key[1]
    assert e.tolist() != a[:4].tolist()


def test_standard_f32(oracle):
    bits, _ = oracle.rng_draw(1, 2, 3, 0, 4096)
    f, _ = oracle.rng_draw(1, 2, 3, 1, 4096)
    # rand 0.8.5 Standard: (u32 >> 8) * 2^-24
    assert np.array_equal(f, (bits >> 8).astype(np.float32) * np.float32(2.0 ** -24))
    assert f.min() >= 0.0 and f.max() < 1.0 and abs(f.mean() - 0.5) < 0.02


def test_uniform_f32(oracle):
    bits, _ = oracle.rng_draw(9, 0, 0, 0, 4096)
    lo, hi = 0.0, (830.0 - 360.0) / 8.0
    u, _ = oracle.rng_draw(9, 0, 0, 3, 4096, lo, hi)
    # rand 0.8.5 UniformFloat: value in [1,2) from 23 mantissa bits, minus 1, times scale, plus low
    v12 = ((bits >> 9) | 0x3F800000).astype(np.uint32).view(np.float32)
    assert np.array_equal(u, (v12 - np.float32(1.0)) * np.float32(hi - lo) + np.float32(lo))
    assert u.min() >= lo and u.max() < hi
    s, _ = oracle.rng_draw(9, 0, 0, 3, 4096, -1.0, 1.0)
    assert s.min() >= -1.0 and s.max() < 1.0


def test_unit_sphere(oracle):
    v, idx = oracle.rng_draw(5, 1, 1, 4, 20000)
    v = v.reshape(-1, 3).astype(np.float64)
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1.0, atol=5e-7)
    assert np.abs(v.mean(axis=0)).max() < 0.02                  # uniform on the sphere
    assert abs((v[:, 2] ** 2).mean() - 1.0 / 3.0) < 0.01
    # Marsaglia rejection: acceptance pi/4, two uniforms per attempt
    assert abs(idx / (2 * 20000) - 4.0 / math.pi) < 0.03


def test_standard_normal(oracle):
    z, idx = oracle.rng_draw(11, 0, 0, 2, 50000)
    assert idx == 50000                                         # one u32 per variate
    z = z.astype(np.float64)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    from scipy.stats import kstest
    assert kstest(z, "norm").pvalue > 1e-3
