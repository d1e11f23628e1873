"""Known-answer tests for the oracle, read straight off the reference source.

The reference ships no tests or golden vectors ("parity unpinned"); these hand-derived answers and
the independent numpy restatement (test_oracle_vs_numpy.py) are what pin the C oracle.
"""
import numpy as np
import pytest

U64 = np.uint64


def test_ord32_known_answers(oracle):
    # src/ord32.rs:12-17
    assert oracle.ord32_from_f32(0.0) == 0
    assert oracle.ord32_from_f32(1.0) == 0x3F800000
    assert oracle.ord32_from_f32(-1.0) == -1065353217  # 0xC07FFFFF
    assert oracle.ord32_from_f32(-0.0) == -1
    xs = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 2.0, 3e38, np.inf], dtype=np.float32)
    keys = [oracle.ord32_from_f32(x) for x in xs]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)  # monotone bijection
    for x in xs:
        assert oracle.ord32_to_f32(oracle.ord32_from_f32(x)).tobytes() == np.float32(x).tobytes()


def test_bitplanes_and_asymmetric_dot(oracle):
    # D=64, q[i] = i & 15  =>  planes 0xAAAA.., 0xCCCC.., 0xF0F0.., 0xFF00FF00.. (simd.rs:103)
    q = (np.arange(64) & 15).astype(np.uint8)
    planes = oracle.vector_binarize_query(q)
    assert [int(p) for p in planes] == [0xAAAAAAAAAAAAAAAA, 0xCCCCCCCCCCCCCCCC, 0xF0F0F0F0F0F0F0F0,
                                        0xFF00FF00FF00FF00]
    # code = all ones  =>  asymmetric dot == sum(q) == 480 (utils.rs:113-135)
    assert oracle.asymmetric_binary_dot_product(np.array([2**64 - 1], dtype=U64), planes) == 480 == int(q.sum())
    assert oracle.asymmetric_binary_dot_product(np.array([0], dtype=U64), planes) == 0
    # single bit 5 set: q[5] = 5
    assert oracle.asymmetric_binary_dot_product(np.array([1 << 5], dtype=U64), planes) == 5


def test_binarize_u64_strictly_positive(oracle):
    # utils.rs:56: bit set iff v > 0.0; zero, negatives (and NaN) -> 0
    v = np.zeros(64, dtype=np.float32)
    v[:4] = [1.0, -1.0, 0.0, 2.0]
    assert int(oracle.vector_binarize_u64(v)[0]) == 0b1001
    v2 = np.array([np.nan, -0.0, 1e-45, np.inf] + [0.0] * 60, dtype=np.float32)
    assert int(oracle.vector_binarize_u64(v2)[0]) == 0b1100


def test_scalar_quantize_rne_ties(oracle):
    # simd.rs:214-215: cvtps_epi32 rounds half to even: 0.5->0, 1.5->2, 2.5->2, 3.5->4; no bias
    v = np.array([0.5, 1.5, 2.5, 3.5, 0.49, 14.5, 15.0, 0.0], dtype=np.float32)
    q, s = oracle.scalar_quantize(v, 0.0, 1.0)
    assert q.tolist() == [0, 2, 2, 4, 0, 14, 15, 0] and s == 37
    # subtract-then-multiply, not fused
    v = np.full(8, 3.0, dtype=np.float32)
    q, s = oracle.scalar_quantize(v, 1.0, 2.0)
    assert q.tolist() == [4] * 8 and s == 32


def test_scalar_quantize_degenerate_delta(oracle):
    # delta == 0  =>  multiplier = inf, (v - lo) * inf = NaN  =>  cvtps_epi32 "indefinite" 0x80000000:
    # low byte 0, wrapped sum 0 for an even number of lanes (rabitq.rs:307-315 edge case)
    v = np.zeros(64, dtype=np.float32)
    q, s = oracle.scalar_quantize(v, 0.0, np.inf)
    assert not q.any() and s == 0


def test_l2_and_dot_reduction_order(oracle):
    # simd.rs:52-63: ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)); choose values where the order matters
    a = np.array([1e8, 1.0, -1e8, 1.0, 1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    ones = np.ones(8, dtype=np.float32)
    f = np.float32
    expect = f(f(f(a[0] + a[4]) + f(a[1] + a[5])) + f(f(a[2] + a[6]) + f(a[3] + a[7])))
    assert oracle.vector_dot_product(a, ones) == expect
    # l2: diff rounded first, then fused square-accumulate over chunks in index order
    x = np.arange(16, dtype=np.float32) * f(0.1)
    y = np.zeros(16, dtype=np.float32)
    lanes = [np.float64(x[i]) ** 2 for i in range(8)]
    lanes = [f(v) for v in lanes]
    lanes = [f(np.float64(x[8 + i]) ** 2 + np.float64(lanes[i])) for i in range(8)]
    expect = f(f(f(lanes[0] + lanes[4]) + f(lanes[1] + lanes[5])) + f(f(lanes[2] + lanes[6]) + f(lanes[3] + lanes[7])))
    assert oracle.l2_squared_distance(x, y) == expect


def test_constants(oracle):
    # consts.rs:10 SCALAR = 1/15; rabitq.rs:220 error_base
    f = np.float32
    assert f(1.0) / f(15.0) == f(0.06666667)
    assert abs(float(f(2.0) * f(1.9) / np.sqrt(f(127.0))) - 0.337195) < 1e-6
    assert abs(float(f(2.0) * f(1.9) / np.sqrt(f(767.0))) - 0.137210) < 1e-6


def test_min_max_residual(oracle):
    x = np.linspace(-3, 5, 128).astype(np.float32)
    y = np.linspace(2, -7, 128).astype(np.float32)
    res, lo, hi = oracle.min_max_residual(x, y)
    assert np.array_equal(res, x - y) and lo == (x - y).min() and hi == (x - y).max()


def test_binary_dot_product_both_branches(oracle):
    rng = np.random.default_rng(0)
    for words in (1, 2, 3, 4, 5, 8, 12, 13):  # <4: scalar branch (simd.rs:333-339); >=4: AVX2 LUT
        x = rng.integers(0, 2**64, size=words, dtype=np.uint64)
        y = rng.integers(0, 2**64, size=words, dtype=np.uint64)
        expect = sum(bin(int(a) & int(b)).count("1") for a, b in zip(x, y))
        assert oracle.binary_dot_product(x, y) == expect


def test_recall(oracle):
    assert oracle.calculate_recall([1, 2, 3, 4], [4, 9, 1, 7], 4) == 0.5
