"""The numerics contract (DESIGN.md "Numerics"): IEEE-754 + - * fma, the fp16 / UNORM codecs, and the five transcendental instructions of gfx950 that the
oracle reproduces from measured tables. CPU part pins the oracle against numpy; GPU part pins the HIP device functions against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from oracle import driver as oracle_driver


def _vec(fn, xs, restype=np.float32):
    return np.array([fn(float(x)) for x in xs], dtype=restype)


def test_oracle_fp16_codec_matches_ieee():
    lib = oracle_driver.load()
    # every half value decodes like numpy's binary16 and re-encodes to itself
    halfs = np.arange(65536, dtype=np.uint16)
    ref = halfs.view(np.float16).astype(np.float32)
    finite = np.isfinite(ref)
    dec = np.array([lib.oracle_f16tof32(int(h)) for h in halfs], dtype=np.float32)
    assert np.array_equal(dec[finite].view(np.uint32), ref[finite].view(np.uint32))
    enc = np.array([lib.oracle_f32tof16(float(v)) for v in ref[finite]], dtype=np.uint16)
    assert np.array_equal(enc, halfs[finite])
    # round-to-nearest-even on random floats incl. denormal halfs, overflow and ties
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000).astype(np.float32),
                         np.array([65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 6.1e-5, 0.0, -0.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11],
                                  dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.oracle_f32tof16(float(v)) for v in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_oracle_transcendentals_are_accurate():
    lib = oracle_driver.load()
    rng = np.random.default_rng(5)
    x = rng.uniform(-30, 30, 4000).astype(np.float32)
    assert np.max(np.abs(_vec(lib.oracle_exp2, x) / np.exp2(x.astype(np.float64)) - 1.0)) < 4e-7
    p = np.exp(rng.uniform(-40, 40, 4000)).astype(np.float32)
    assert np.max(np.abs(_vec(lib.oracle_log2, p) - np.log2(p.astype(np.float64)))) < 2e-6 * np.maximum(1.0, np.abs(np.log2(p))).max()
    a = np.concatenate([rng.uniform(-5, 5, 3000), rng.uniform(-1e4, 1e4, 500), [0.0, 0.41421356, 2.41421356, 1.0]]).astype(np.float32)
    assert np.max(np.abs(_vec(lib.oracle_atan, a) - np.arctan(a.astype(np.float64)))) < 3e-7
    assert lib.oracle_exp2(0.0) == 1.0 and lib.oracle_exp2(3.0) == 8.0 and lib.oracle_log2(8.0) == 3.0 and lib.oracle_log2(1.0) == 0.0
    assert lib.oracle_pow(0.0, 2.0) == 0.0 and abs(lib.oracle_pow(0.5, 4.0) - 0.0625) < 1e-7
    # round 5: the forms for non-positive arguments (oracle/hlsl.h Exp2NonPos / SatExp2 / ExpNegAbs, ml.h Pow01)
    xn = -np.concatenate([np.abs(rng.standard_normal(4000)) * 10.0 ** rng.integers(-10, 2, 4000), [0.0, 1.0, 2.0, 100.0]]).astype(np.float32)
    for op, exact in ((5, np.exp2(xn.astype(np.float64))), (6, np.exp2(xn.astype(np.float64))), (7, np.exp(xn.astype(np.float64)))):
        out = np.empty_like(xn)
        lib.oracle_eval_hw(op, xn.ctypes.data, out.ctypes.data, xn.size)
        ok = exact > 2.4e-38  # (the instruction computes 2^(x - 1): results below 2^-125 are flushed to zero)
        # x - 1 is exact for |x| >= 1 except where it crosses into the next binade (half an ulp of |x|: a relative error of |x| * 2^-24 * ln 2 in 2^x)
        # (e^-|w| multiplies by the fp32 constant log2 e first: the product carries half an ulp of |w| * 1.44 as well, as the plain exp() always did)
        assert np.all(np.abs(out[ok] / exact[ok] - 1.0) < 4e-7 + (1.2e-7 if op == 7 else 5e-8) * np.abs(xn[ok].astype(np.float64))), op
    pos = np.array([0.5, 3.0, 1e30], dtype=np.float32)
    out = np.empty_like(pos)
    lib.oracle_eval_hw(6, pos.ctypes.data, out.ctypes.data, pos.size)
    assert np.all(out == 1.0)  # saturate(2^x) = 1 for x > 0
    assert lib.oracle_pow01(0.5, 4.0) == 0.0625 and lib.oracle_pow01(-1.0, 2.0) == 0.0 and lib.oracle_pow01(1.0, 7.0) == 1.0 and lib.oracle_pow01(3.0, 2.0) == 1.0


@pytest.mark.gpu
def test_hip_numerics_bit_exact_vs_oracle():
    import torch

    from raytracingdenoiser_amd import api

    lib = api.load_library()
    ora = oracle_driver.load()
    rng = np.random.default_rng(11)
    n = 20000
    cases = {
        0: (rng.uniform(-40, 40, n), None, ora.oracle_exp2),
        1: (np.exp(rng.uniform(-60, 60, n)), None, ora.oracle_log2),
        2: (np.concatenate([rng.uniform(-6, 6, n - 1000), rng.uniform(-1e5, 1e5, 1000)]), None, ora.oracle_atan),
        3: (rng.uniform(0, 1, n), rng.uniform(0.1, 40, n), ora.oracle_pow),
        4: (rng.standard_normal(n) * 10.0 ** rng.integers(-9, 6, n), None, lambda v: ora.oracle_f16tof32(ora.oracle_f32tof16(v))),
    }
    for op, (a, b, fn) in cases.items():
        a32 = a.astype(np.float32)
        ta = torch.from_numpy(a32).cuda()
        tb = torch.from_numpy(b.astype(np.float32)).cuda() if b is not None else None
        out = torch.empty_like(ta)
        r = lib.nrdHipEvalNumerics(op, ta.data_ptr(), tb.data_ptr() if tb is not None else None, out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        assert r == 0
        got = out.cpu().numpy()
        if b is None:
            want = np.array([fn(float(v)) for v in a32], dtype=np.float32)
        else:
            want = np.array([fn(float(v), float(w)) for v, w in zip(a32, b.astype(np.float32))], dtype=np.float32)
        ok = np.isfinite(want)
        bad = np.nonzero(got[ok].view(np.uint32) != want[ok].view(np.uint32))[0]
        assert bad.size == 0, "op %d: %d / %d differ from the oracle, e.g. in=%r got=%r want=%r" % (op, bad.size, ok.sum(), a32[ok][bad[:4]], got[ok][bad[:4]], want[ok][bad[:4]])
    # a / b of the contract = a * v_rcp_f32(b): the device's Div against the oracle's reciprocal (table emulation) and one fp32 multiplication
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)
    ta, tb, out = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.empty(n, device="cuda")
    lib.nrdHipEvalNumerics(5, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    rb = np.empty_like(b)
    ora.oracle_eval_hw(2, b.ctypes.data, rb.ctypes.data, n)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), (a * rb).view(np.uint32))
    assert np.abs(rb.view(np.int32).astype(np.int64) - (1.0 / b.astype(np.float64)).astype(np.float32).view(np.int32)).max() <= 1
    # div_edge_cases (the contract is spelled out at nrdmath.h Div): denominators that v_rcp_f32 flushes (denormals -> +-inf, |b| > 2^126 -> +-0),
    # zeros, infinities, NaN against numerators 0, +-1, tiny, huge: the device and the oracle agree bit for bit (NaN for NaN), including 0 * inf = NaN
    num = np.array([0.0, -0.0, 1.0, -1.0, 1e-38, 3e38, 1e-45], dtype=np.float32)
    den = np.array([1e-39, -1e-39, 1e-45, 0.0, -0.0, 2.0 ** 126, 1.8e38, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan, 1.17549435e-38, 3.0], dtype=np.float32)
    ea, eb = np.repeat(num, den.size).astype(np.float32), np.tile(den, num.size).astype(np.float32)
    ta, tb, out = torch.from_numpy(ea).cuda(), torch.from_numpy(eb).cuda(), torch.empty(ea.size, device="cuda")
    lib.nrdHipEvalNumerics(5, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), ea.size, torch.cuda.current_stream().cuda_stream)
    erb = np.empty_like(eb)
    ora.oracle_eval_hw(2, eb.ctypes.data, erb.ctypes.data, eb.size)
    with np.errstate(invalid="ignore", over="ignore"):
        want = ea * erb
    got = out.cpu().numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), list(zip(ea[~same], eb[~same], got[~same], want[~same]))[:8]
    specials = np.array([0.0, -0.0, np.inf, 1e-45, 1e-39, -1e-39, 1.17549435e-38, 3.4e38, 1.0, 2.0, 4.0, 0.25, -1.0, -np.inf], dtype=np.float32)
    wide = (np.abs(rng.standard_normal(200000)) * 10.0 ** rng.integers(-37, 38, 200000)).astype(np.float32)  # the whole exponent range
    pa = np.concatenate([np.abs(a), wide, specials]).astype(np.float32)
    ta, out = torch.from_numpy(pa).cuda(), torch.empty(pa.size, device="cuda")
    for op, hw in ((6, 0), (7, 1), (20, 2)):  # Sqrt, Rsqrt, Rcp over the whole exponent range, zeros, infinities, denormals
        src = np.concatenate([pa, -pa]).astype(np.float32) if hw == 2 else pa
        ta, out = torch.from_numpy(src).cuda(), torch.empty(src.size, device="cuda")
        lib.nrdHipEvalNumerics(op, ta.data_ptr(), None, out.data_ptr(), src.size, torch.cuda.current_stream().cuda_stream)
        got, want = out.cpu().numpy(), np.empty_like(src)
        ora.oracle_eval_hw(hw, src.ctypes.data, want.ctypes.data, src.size)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "op %d: %d differ, e.g. in=%r got=%r want=%r" % (op, (~same).sum(), src[~same][:6], got[~same][:6], want[~same][:6])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            x64 = src.astype(np.float64)
            exact = (np.sqrt(x64) if hw == 0 else 1.0 / np.sqrt(x64) if hw == 1 else 1.0 / x64).astype(np.float32)
        normal = np.isfinite(exact) & (np.abs(src) >= np.float32(1.17549435e-38)) & np.isfinite(src) & (np.abs(exact) >= np.float32(1.17549435e-38))
        assert np.abs(got[normal].view(np.int32).astype(np.int64) - exact[normal].view(np.int32)).max() <= 1  # within 1 ulp of the correctly rounded result
    # round 5: 2^x for x <= 0 as 2 * v_exp_f32(x - 1) (nrdmath.h Exp2NonPos / SatExp2 / ExpNegAbs / Pow01): the instruction on arguments <= -1 against the oracle's
    # sign-magnitude model of it (oracle/hw_math.h HwExp2OnNegative, table oracle/hw_exp2neg.i8.z) -- dense over every binade the chain can produce, and the specials
    neg = -np.concatenate([np.abs(rng.standard_normal(400000)) * 10.0 ** rng.integers(-12, 3, 400000), rng.uniform(0, 130, 200000), [0.0, 1.0, 2.0, 124.9, 125.0, 125.5, 126.0, 127.0, 149.0, 1e30, np.inf, 1e-45, 1e-39]]).astype(np.float32)
    anysign = np.concatenate([neg, -neg[:200000], [np.nan]]).astype(np.float32)
    for op, hw, src in ((21, 5, neg), (22, 6, anysign), (23, 7, anysign)):
        ta, out = torch.from_numpy(src).cuda(), torch.empty(src.size, device="cuda")
        assert lib.nrdHipEvalNumerics(op, ta.data_ptr(), None, out.data_ptr(), src.size, torch.cuda.current_stream().cuda_stream) == 0
        got, want = out.cpu().numpy(), np.empty_like(src)
        ora.oracle_eval_hw(hw, src.ctypes.data, want.ctypes.data, src.size)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "op %d: %d differ, e.g. in=%r got=%r want=%r" % (op, (~same).sum(), src[~same][:6], got[~same][:6], want[~same][:6])
        with np.errstate(over="ignore", invalid="ignore"):
            x64 = src.astype(np.float64)
            exact = np.exp2(x64) if op == 21 else np.minimum(np.exp2(x64), 1.0) if op == 22 else np.exp(-np.abs(x64))
        normal = np.isfinite(exact) & (exact >= 2.4e-38)  # (the instruction computes 2^(x - 1): results below 2^-125 are flushed to zero)
        assert np.all(np.abs(got[normal] / exact[normal] - 1.0) < 4e-7 + (1.2e-7 if op == 23 else 5e-8) * np.abs(x64[normal]))  # (see test_oracle_transcendentals_are_accurate)
    pa01, pb01 = rng.uniform(-0.2, 1.2, n).astype(np.float32), rng.uniform(0.0, 40, n).astype(np.float32)
    ta, tb, out = torch.from_numpy(pa01).cuda(), torch.from_numpy(pb01).cuda(), torch.empty(n, device="cuda")
    assert lib.nrdHipEvalNumerics(24, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
    want = np.array([ora.oracle_pow01(float(v), float(w)) for v, w in zip(pa01, pb01)], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # v_cvt_pk_f16_f32 (StoreRGBA16F): each half is the round-to-nearest-even conversion of its own operand
    xs = np.concatenate([rng.standard_normal(n) * 10.0 ** rng.integers(-9, 6, n), [65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 6.1e-5, 0.0, -0.0]]).astype(np.float32)
    ys = xs[::-1].copy()
    ta, tb, out = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda(), torch.empty(xs.size, device="cuda")
    lib.nrdHipEvalNumerics(19, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), xs.size, torch.cuda.current_stream().cuda_stream)
    packed = out.cpu().numpy().view(np.uint32)
    with np.errstate(over="ignore"):
        lo, hi = xs.astype(np.float16).view(np.uint16).astype(np.uint32), ys.astype(np.float16).view(np.uint16).astype(np.uint32)
    assert np.array_equal(packed, lo | (hi << 16))


@pytest.mark.gpu
def test_hip_constant_division_and_gaussian_constants():
    """planes.h: k / c via q0 = k*R, r = fma(-q0,c,k), q = fma(r,R,q0) equals the IEEE quotient for every numerator the codecs
    produce; reblur_device.h: the two baked Gaussian tap weights equal Exp(-0.66 z^2) evaluated on the device."""
    import torch

    from raytracingdenoiser_amd import api

    lib = api.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    for op, c, n in ((8, 1023.0, 1024), (9, 255.0, 256), (10, 63.0, 64), (11, 15.0, 16), (12, 3.0, 4), (14, 65535.0, 65536)):
        k = np.arange(n, dtype=np.float32)
        t, out = torch.from_numpy(k).cuda(), torch.empty(n, device="cuda")
        assert lib.nrdHipEvalNumerics(op, t.data_ptr(), None, out.data_ptr(), n, stream) == 0
        assert np.array_equal(out.cpu().numpy().view(np.uint32), (k / np.float32(c)).view(np.uint32))
    k = np.arange(-32768, 32768, dtype=np.float32)  # SNORM16 decode (negative numerators too)
    t, out = torch.from_numpy(k).cuda(), torch.empty(k.size, device="cuda")
    assert lib.nrdHipEvalNumerics(15, t.data_ptr(), None, out.data_ptr(), k.size, stream) == 0
    assert np.array_equal(out.cpu().numpy().view(np.uint32), (k / np.float32(32767.0)).view(np.uint32))
    z = np.array([1.0, 0.5, 0.3], dtype=np.float32)  # offset.z of g_Special8 (1, 0.5) and g_Special6 (1, 0.3)
    t, out = torch.from_numpy(z).cuda(), torch.empty(3, device="cuda")
    assert lib.nrdHipEvalNumerics(13, t.data_ptr(), None, out.data_ptr(), 3, stream) == 0
    assert out.cpu().numpy().view(np.uint32).tolist() == [0x3F04505F, 0x3F590F8F, 0x3F713C86]
    ora = oracle_driver.load()
    assert np.float32(ora.oracle_exp2(float(np.float32(np.float32(-0.66) * np.float32(1.0) * np.float32(1.0)) * np.float32(1.44269504)))).view(np.uint32) == 0x3F04505F
    # RELAX pre-pass: the z column of g_Poisson8 (reference Shaders/Include/Poisson.hlsli:40-50) against kernels_relax_spatial.hip g_Poisson8Gaussian
    z = np.array([0.6461146, 0.9542373, 0.5335386, 0.6520134, 0.6695386, 0.3149309, 0.8895339, 0.8346850], dtype=np.float32)
    t, out = torch.from_numpy(z).cuda(), torch.empty(8, device="cuda")
    assert lib.nrdHipEvalNumerics(13, t.data_ptr(), None, out.data_ptr(), 8, stream) == 0
    assert out.cpu().numpy().view(np.uint32).tolist() == [0x3F425921, 0x3F0C5BDB, 0x3F5426BA, 0x3F415E51, 0x3F3E6F61, 0x3F6FC76C, 0x3F17DB60, 0x3F21A332]


def test_oracle_division_contract_edge_cases():
    """a / b = a * v_rcp_f32(b): where that is not the IEEE quotient (documented at nrdmath.h Div and oracle/hlsl.h)"""
    ora = oracle_driver.load()
    prev = oracle_driver.set_ieee_mode(False)
    try:
        den = np.array([1e-39, 3.4e38, 2.0 ** 126, 2.0 ** 126 * 1.5, 3.0, np.inf, 0.0], dtype=np.float32)
        r = np.empty_like(den)
        ora.oracle_eval_hw(2, den.ctypes.data, r.ctypes.data, den.size)
    finally:
        oracle_driver.set_ieee_mode(prev)
    assert np.isposinf(r[0])  # a denormal denominator is flushed: rcp = inf, so 0 / denormal = 0 * inf = NaN where IEEE gives 0
    assert r[1] == 0.0 and r[3] == 0.0  # 1 / b would be denormal: flushed to 0 (IEEE: 2.9e-39)
    assert r[2] == np.float32(2.0 ** -126)  # the smallest normal result survives
    assert abs(float(r[4]) * 3.0 - 1.0) < 2e-7 and r[5] == 0.0 and np.isposinf(r[6])
    with np.errstate(invalid="ignore"):
        assert np.isnan(np.float32(0.0) * r[0])


def test_hw_tables_are_sane():
    """oracle/hw_{rcp,sqrt,rsq,exp2,log2}.i8.z (the measured deviation of gfx950's v_rcp / v_sqrt / v_rsq / v_exp / v_log from the reference results of
    oracle/hw_ref.h): every entry is -1, 0 or +1 ulp (checked at load), exact squares / powers of two stay exact, the emulation is within 1 ulp of the correctly
    rounded value over the whole exponent range, flushes denormals and follows the instructions on zeros / infinities / negative inputs."""
    ora = oracle_driver.load()
    rng = np.random.default_rng(5)
    x = (np.abs(rng.standard_normal(400000)) * 10.0 ** rng.integers(-37, 38, 400000)).astype(np.float32)
    x = x[(x >= np.float32(1.17549435e-38)) & np.isfinite(x)]
    for op, exact in ((0, np.sqrt(x.astype(np.float64))), (1, 1.0 / np.sqrt(x.astype(np.float64))), (2, 1.0 / x.astype(np.float64))):
        out = np.empty_like(x)
        ora.oracle_eval_hw(op, x.ctypes.data, out.ctypes.data, x.size)
        ok = exact >= 1.17549435e-38
        d = out.view(np.int32).astype(np.int64)[ok] - exact.astype(np.float32).view(np.int32)[ok]
        assert np.abs(d).max() <= 1 and 0.8 < np.mean(d == 0) < 0.95
    squares = (np.arange(1, 4096, dtype=np.float32) ** 2).astype(np.float32)
    out = np.empty_like(squares)
    ora.oracle_eval_hw(0, squares.ctypes.data, out.ctypes.data, squares.size)
    assert np.array_equal(out, np.arange(1, 4096, dtype=np.float32))
    pow2 = (2.0 ** np.arange(-100, 100)).astype(np.float32)
    ora.oracle_eval_hw(2, pow2.ctypes.data, (out := np.empty_like(pow2)).ctypes.data, pow2.size)
    assert np.array_equal(out, (1.0 / pow2.astype(np.float64)).astype(np.float32))
    special = np.array([0.0, -0.0, np.inf, 1e-45, 1e-39, -1e-39, -1.0], dtype=np.float32)
    s, r, q = np.empty_like(special), np.empty_like(special), np.empty_like(special)
    ora.oracle_eval_hw(0, special.ctypes.data, s.ctypes.data, special.size)
    ora.oracle_eval_hw(1, special.ctypes.data, r.ctypes.data, special.size)
    ora.oracle_eval_hw(2, special.ctypes.data, q.ctypes.data, special.size)
    assert s[:6].tolist() == [0.0, 0.0, np.inf, 0.0, 0.0, 0.0] and np.signbit(s[1]) and np.signbit(s[5]) and np.isnan(s[6])
    assert r[:6].tolist() == [np.inf, -np.inf, 0.0, np.inf, np.inf, -np.inf] and np.isnan(r[6])
    assert q.tolist() == [np.inf, -np.inf, 0.0, np.inf, np.inf, -np.inf, -1.0]
    # exp2 / log2 of the contract (range reduction + the instruction on one binade): accuracy over the ranges the passes use, exact at the integers
    xs = rng.uniform(-120, 120, 200000).astype(np.float32)
    ora.oracle_eval_hw(3, xs.ctypes.data, (e := np.empty_like(xs)).ctypes.data, xs.size)
    assert np.max(np.abs(e.astype(np.float64) / np.exp2(xs.astype(np.float64)) - 1.0)) < 2.5e-7
    ints = np.arange(-125, 126, dtype=np.float32)
    ora.oracle_eval_hw(3, ints.ctypes.data, (e := np.empty_like(ints)).ctypes.data, ints.size)
    assert np.array_equal(e, np.exp2(ints.astype(np.float64)).astype(np.float32))
    ps = np.exp(rng.uniform(-80, 80, 200000)).astype(np.float32)
    ora.oracle_eval_hw(4, ps.ctypes.data, (l := np.empty_like(ps)).ctypes.data, ps.size)
    assert np.max(np.abs(l - np.log2(ps.astype(np.float64)))) < 1.5e-5  # absolute: one fp32 addition e + log2(m) with |e| <= 127
    near1 = (1.0 + rng.uniform(0, 1e-3, 1000)).astype(np.float32)
    ora.oracle_eval_hw(4, near1.ctypes.data, (l := np.empty_like(near1)).ctypes.data, near1.size)
    assert np.max(np.abs(l / np.log2(near1.astype(np.float64)) - 1.0)[near1 > 1.0]) < 3e-7  # e = 0: the relative accuracy of the instruction


