"""GPU: the device builds of csrc/fmd_math.h swept directly against what the reference's CPU build
calls: glibc atan2f (bit-identical), the x87 fsincos instruction (rounding-boundary cases only,
probability ~2^-28 per value), IEEE division, the RTL-SDR byte conversion.  The CPU test
(test_device_math_cpu.py) sweeps the host build of the same source; this one covers what only
exists on the device: reciprocal-based division, wave ballots, LDS tables."""
import ctypes as C

import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
NSWEEP = 4_000_000


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _atan2f_host(oracle, y, x):
    f = oracle.lib().fmo_atan2f
    return np.array([f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)


def _args(rng, n):
    """y, x pairs: PLL-like magnitudes, wide exponent spreads, range edges, specials."""
    parts_y, parts_x = [], []
    m = n // 4
    parts_y.append(rng.uniform(-2, 2, m)), parts_x.append(rng.uniform(-2, 2, m))
    parts_y.append(rng.uniform(-1e-3, 1e-3, m)), parts_x.append(rng.uniform(-1, 1, m))
    e = rng.integers(-40, 40, m)
    parts_y.append(np.ldexp(rng.uniform(-1, 1, m), e)), parts_x.append(rng.uniform(-1, 1, m))
    # quotients next to the range thresholds 7/16, 11/16, 19/16, 39/16 and to 2^-29, 2^25
    thr = np.array([7 / 16, 11 / 16, 19 / 16, 39 / 16, 2.0 ** -29, 2.0 ** 25], dtype=np.float64)
    t = thr[rng.integers(0, thr.size, m)] * (1 + rng.integers(-4, 5, m) * 2.0 ** -24)
    xs = rng.uniform(0.01, 2, m) * rng.choice([-1.0, 1.0], m)
    parts_y.append(t * xs * rng.choice([-1.0, 1.0], m)), parts_x.append(xs)
    y = np.concatenate(parts_y).astype(np.float32)
    x = np.concatenate(parts_x).astype(np.float32)
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-45, -1e-45, 3e38, 1e-38],
                  dtype=np.float32)
    yy, xx = np.meshgrid(sp, sp)
    return np.concatenate([y, yy.ravel()]), np.concatenate([x, xx.ravel()])


def test_atan2f_forms_equal_glibc(pkg, oracle):
    rng = np.random.default_rng(1)
    y, x = _args(rng, 400_000)
    ref = _atan2f_host(oracle, y, x)
    for what in (0, 1):  # table form used in the kernels, literal fdlibm restatement
        got, _ = pkg.debug_math(what, y, x)
        same = (_bits(got) == _bits(ref)) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), (what, y[~same][:5], x[~same][:5], got[~same][:5], ref[~same][:5])


def test_midrange_division_is_ieee(pkg):
    rng = np.random.default_rng(2)
    n = NSWEEP
    den = np.concatenate([rng.uniform(0.4375, 8, n // 2),
                          np.ldexp(rng.uniform(1, 2, n // 2), rng.integers(-1, 25, n // 2))])
    num = np.concatenate([rng.uniform(-2, 2, n // 2),
                          np.ldexp(rng.uniform(-2, 2, n // 2), rng.integers(-30, 25, n // 2))])
    num[:1000] = 0.0
    num, den = num.astype(np.float32), den.astype(np.float32)
    got, _ = pkg.debug_math(4, num, den)
    assert np.array_equal(_bits(got), _bits(num / den))  # numpy float32 division is IEEE


def test_sincos_forms_equal_fsincos(pkg, oracle):
    rng = np.random.default_rng(3)
    ph = np.concatenate([rng.uniform(-0.5, 6.8, 300_000), rng.uniform(-40, 40, 50_000),
                         np.float32(2 * np.pi) * rng.integers(-3, 4, 1000)]).astype(np.float32)
    s_ref = np.empty_like(ph)
    c_ref = np.empty_like(ph)
    f = oracle.lib().fmo_sincos_x87
    sv, cv = C.c_float(), C.c_float()
    for i, p in enumerate(ph):
        f(float(p), C.byref(sv), C.byref(cv))
        s_ref[i], c_ref[i] = sv.value, cv.value
    for what in (2, 3):
        s, c = pkg.debug_math(what, ph)
        bad = int((_bits(s) != _bits(s_ref)).sum() + (_bits(c) != _bits(c_ref)).sum())
        assert bad <= 1, (what, bad)  # expected ~0.003 double-rounding cases in 700k values


def test_sincos_exact_reduction_equals_fsincos(pkg, oracle):
    """fmd_sincos_p256 (the form the serial stage's two NCOs use: phase in [0, 2 pi], split into
    k / 256 + r without rounding error) against the x87 instruction: a dense sweep of the range, every
    float next to the table's grid points, and the floats around 0 and 2 pi."""
    rng = np.random.default_rng(7)
    twopi = np.float32(2 * np.pi)
    grid = (np.arange(0, 1610, dtype=np.float32) / np.float32(256.0))
    near = np.concatenate([np.nextafter(grid, np.float32(10)), grid, np.nextafter(grid, np.float32(-1)),
                           grid + np.float32(1.0 / 512), np.nextafter(grid + np.float32(1.0 / 512), np.float32(10))])
    edge = np.array([0.0, 1e-45, 1e-38, 1e-20, 1e-7, twopi, np.nextafter(twopi, np.float32(0)),
                     np.nextafter(twopi, np.float32(10)), 6.2831855, 6.29, 7.99], dtype=np.float32)
    ph = np.concatenate([rng.uniform(0, 6.2832, 400_000).astype(np.float32), near[near >= 0], edge])
    s_ref = np.empty_like(ph)
    c_ref = np.empty_like(ph)
    f = oracle.lib().fmo_sincos_x87
    sv, cv = C.c_float(), C.c_float()
    for i, p in enumerate(ph):
        f(float(p), C.byref(sv), C.byref(cv))
        s_ref[i], c_ref[i] = sv.value, cv.value
    s, c = pkg.debug_math(7, ph)
    bad = int((_bits(s) != _bits(s_ref)).sum() + (_bits(c) != _bits(c_ref)).sum())
    assert bad <= 1, bad  # double-rounding cases: probability ~2^-28 per value
    # non-finite phases give NaN like the instruction does
    s, c = pkg.debug_math(7, np.array([np.nan, np.inf], dtype=np.float32))
    assert np.isnan(s).all() and np.isnan(c).all()


def test_byte_conversion_all_values(pkg, oracle):
    b = np.arange(256, dtype=np.float32)
    got, _ = pkg.debug_math(5, b)
    ref = oracle.convert_u8(np.arange(256, dtype=np.uint8))
    assert np.array_equal(_bits(got), _bits(ref))


def test_rds_arctan2_equals_oracle(pkg, oracle):
    rng = np.random.default_rng(4)
    y = rng.uniform(-2, 2, 200_000).astype(np.float32)
    x = rng.uniform(-2, 2, 200_000).astype(np.float32)
    y[:100] = 0.0
    x[50:150] = 0.0
    f = oracle.lib().fmo_rds_arctan2
    ref = np.array([f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    got, _ = pkg.debug_math(6, y, x)
    assert np.array_equal(_bits(got), _bits(ref))
