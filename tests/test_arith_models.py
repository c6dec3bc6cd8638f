"""The arithmetic model of DESIGN.md §4 on bulk operands.

CPU: the oracle's restated complex division equals the compiler's own _Complex float division (libgcc __divsc3, what the
reference's std::complex<float> operator/ compiles to in the surveyed build), and glibc hypotf equals the double-sqrt form the
kernels use, on a million operands each including denormals and extreme ratios.
GPU (-m gpu): the device primitives (c8p_selftest) equal the oracle's on the same operands, bit for bit."""
import numpy as np
import pytest

from c8p_testlib import oracle


def operands(n, k, seed):
    """(n, k) float32: Gaussian mantissas with exponents spread over the whole float range, plus blocks of special cases:
    denormals, zeros, huge / tiny ratios, equal magnitudes, powers of two"""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, k)).astype(np.float32)
    e = rng.integers(-30, 31, size=(n, k))
    x = (x.astype(np.float64) * 10.0 ** e).astype(np.float32)
    q = n // 16
    x[0 * q:1 * q] = (rng.normal(size=(q, k)) * 1e-41).astype(np.float32)              # denormals
    x[1 * q:2 * q, 0] = 0.0                                                           # one component zero
    x[2 * q:3 * q, 0] = (rng.normal(size=q) * 1e30).astype(np.float32)                 # huge against tiny
    x[2 * q:3 * q, 1] = (rng.normal(size=q) * 1e-30).astype(np.float32)
    x[3 * q:4 * q, 1] = x[3 * q:4 * q, 0]                                              # equal magnitudes
    x[4 * q:5 * q] = np.float32(2.0) ** rng.integers(-60, 60, size=(q, k)).astype(np.float32)
    x[5 * q:6 * q] = (rng.normal(size=(q, k)) * 1e-3).astype(np.float32)               # the demodulator's working range
    return x


def test_hypotf_equals_the_double_sqrt_form_on_a_million_operands():
    xy = operands(1 << 20, 2, 1)
    model = np.sqrt(xy[:, 0].astype(np.float64) ** 2 + xy[:, 1].astype(np.float64) ** 2).astype(np.float32)
    assert np.array_equal(oracle().cabsf_n(xy), model)


def _div_operands(n, seed):
    ab = operands(n, 4, seed)
    ab = np.clip(ab, -1e18, 1e18)                              # keep c^2 + d^2 and the products finite in float-double range
    den = ab[:, 2].astype(np.float64) ** 2 + ab[:, 3].astype(np.float64) ** 2
    return ab[(den > 1e-60) & np.isfinite(den)]


def test_restated_complex_division_equals_libgcc_on_a_million_operands():
    """std::complex<float> / std::complex<float> = libgcc __divsc3 = ((ac+bd)/(c^2+d^2), (bc-ad)/(c^2+d^2)) evaluated
    in double and rounded once — on operands whose quotient is finite and normal (the scaling branches of __divsc3 for
    overflowing / vanishing denominators are outside the equaliser's range: |H|^2 of a received LTF)"""
    ab = _div_operands(1 << 20, 2)
    mine = oracle().cdiv_n(ab)
    gcc = oracle().cdiv_n(ab, libgcc=True)
    ok = np.isfinite(gcc).all(axis=1) & (np.abs(gcc).max(axis=1) > 1e-30)
    assert ok.sum() > 0.8 * len(ab)
    assert np.array_equal(mine[ok], gcc[ok])


@pytest.mark.gpu
def test_device_cabs_equals_hypotf():
    from gr_ieee80211_amd import engine
    xy = operands(1 << 21, 2, 3)
    e = engine.Engine()
    got = e.selftest(0, xy)
    e.close()
    assert np.array_equal(got, oracle().cabsf_n(xy))


@pytest.mark.gpu
def test_device_complex_division_equals_the_oracle():
    from gr_ieee80211_amd import engine
    ab = _div_operands(1 << 21, 4)
    e = engine.Engine()
    got = e.selftest(1, ab)
    e.close()
    assert np.array_equal(got.view(np.uint32), oracle().cdiv_n(ab).view(np.uint32))


@pytest.mark.gpu
def test_device_detection_quotient_is_correctly_rounded():
    """presiso's ac = sqrtf(p.re^2 + p.im^2) / e with IEEE sqrt and division, as the oracle (and VOLK / GNU Radio) compute it"""
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(5)
    n = 1 << 21
    t = np.zeros((n, 3), np.float32)
    t[:, :2] = (rng.normal(size=(n, 2)) * 10.0 ** rng.integers(-6, 3, size=(n, 1))).astype(np.float32)
    t[:, 2] = (np.abs(rng.normal(size=n)) * 10.0 ** rng.integers(-6, 3, size=n) + 1e-12).astype(np.float32)
    # a block of quotients within a few ulp of the 0.3 threshold
    q = n // 4
    ee = t[:q, 2].astype(np.float64)
    t[:q, 0] = (0.3 * ee * (1.0 + rng.integers(-8, 9, size=q) * 2.0 ** -24)).astype(np.float32)
    t[:q, 1] = 0.0
    e = engine.Engine()
    got = e.selftest(2, t)
    e.close()
    s = (t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1]).astype(np.float32)
    want = (np.sqrt(s).astype(np.float32) / t[:, 2]).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (np.abs(want[:q] - np.float32(0.3)) < 3e-7).sum() > q // 2
