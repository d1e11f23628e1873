"""The C/AVX2 oracle against the independent numpy / pure-Python restatement (oracle/raw_numpy.py).

Integer work must agree exactly; float reductions to rounding; the orchestration (build, query,
both re-rankers, Rust BinaryHeap) must agree exactly when both run on the same float kernels.
"""
import numpy as np
import pytest

from oracle import raw_numpy as rn
from tests import synth


@pytest.mark.parametrize("dim", [64, 128, 256, 768])
def test_bitplanes_match_raw(oracle, dim):
    rng = np.random.default_rng(dim)
    q = rng.integers(0, 16, size=dim, dtype=np.uint8)
    assert np.array_equal(oracle.vector_binarize_query(q), rn.vector_binarize_query_raw(q))


@pytest.mark.parametrize("dim", [64, 128, 256, 320, 768])
def test_asymmetric_dot_matches_raw(oracle, dim):
    rng = np.random.default_rng(dim + 1)
    for _ in range(20):
        x = rng.integers(0, 2**64, size=dim // 64, dtype=np.uint64)
        q = rng.integers(0, 16, size=dim, dtype=np.uint8)
        planes = oracle.vector_binarize_query(q)
        got = oracle.asymmetric_binary_dot_product(x, planes)
        assert got == rn.asymmetric_binary_dot_product_raw(x, planes)
        # and it is the integer inner product <bits(x), q>
        bits = np.array([(int(x[i // 64]) >> (i % 64)) & 1 for i in range(dim)])
        assert got == int((bits * q).sum())


def test_min_max_and_sign_pack_match_raw(oracle):
    rng = np.random.default_rng(3)
    for dim in (64, 128, 768):
        x = rng.standard_normal(dim).astype(np.float32)
        y = rng.standard_normal(dim).astype(np.float32)
        res, lo, hi = oracle.min_max_residual(x, y)
        res2, lo2, hi2 = rn.min_max_raw(x, y)
        assert np.array_equal(res, res2) and lo == lo2 and hi == hi2
        r = res.copy()
        r[::7] = 0.0
        assert np.array_equal(oracle.vector_binarize_u64(r), rn.vector_binarize_u64(r))


def test_scalar_quantize_matches_rne_restatement_not_floor(oracle):
    rng = np.random.default_rng(4)
    differs = 0
    for _ in range(50):
        v = rng.standard_normal(128).astype(np.float32)
        lo, hi = v.min(), v.max()
        delta = np.float32(np.float32(hi - lo) * rn.SCALAR)
        mult = np.float32(1.0) / delta
        q, s = oracle.scalar_quantize(v, lo, mult)
        q2, s2 = rn.scalar_quantize_rne(v, lo, mult)
        assert np.array_equal(q, q2) and s == s2 and q.max() <= 15
        # the scalar fallback (floor + random bias, utils.rs:204) is a different quantiser
        q3, _ = rn.scalar_quantize_raw(v, rng.random(128).astype(np.float32), lo, mult)
        differs += int(not np.array_equal(q, q3))
    assert differs > 0


def test_float_reductions_lane_order(oracle):
    rng = np.random.default_rng(5)
    exact = total = 0
    for dim in (64, 128, 768, 100):
        for _ in range(50):
            a = rng.standard_normal(dim).astype(np.float32)
            b = rng.standard_normal(dim).astype(np.float32)
            for got, ref, f64 in (
                (oracle.l2_squared_distance(a, b), rn.l2_squared_distance_lanes(a, b),
                 ((a.astype(np.float64) - b) ** 2).sum()),
                (oracle.vector_dot_product(a, b), rn.vector_dot_product_lanes(a, b),
                 (a.astype(np.float64) * b).sum())):
                total += 1
                exact += int(np.float32(got) == ref)
                assert abs(got - f64) <= dim * 2.0 ** -24 * max(1.0, abs(f64)) * 4  # stated f32 tolerance
    assert exact >= total - 2  # the numpy FMA emulation can double-round in rare ties


def test_rust_binary_heap_model():
    import heapq
    rng = np.random.default_rng(6)
    h = rn.RustBinaryHeap()
    ref = []
    for step in range(2000):
        if ref and rng.random() < 0.4:
            got = h.pop()
            want = -heapq.heappop(ref)
            assert got[0] == want
        else:
            key = int(rng.integers(-50, 50))
            h.push((key, step))
            heapq.heappush(ref, -key)
        assert len(h.data) == len(ref)
        if h.data:
            assert h.data[0][0] == -ref[0]


@pytest.mark.parametrize("heuristic", [False, True])
def test_orchestration_matches_python_model(oracle, heuristic):
    n, d, k = 600, 128, 8
    x, centres, _ = synth.mixture(n, d, k, sigma=0.6, seed=7)
    P = synth.random_orthogonal(d, seed=11)
    idx = oracle.OracleIndex.build(x, centres, P)
    py = rn.PyRaBitQ(x, centres, P, l2=oracle.l2_squared_distance, dot=oracle.vector_dot_product)
    assert np.array_equal(idx.offsets, py.offsets)
    assert np.array_equal(idx.map_ids, py.map_ids)
    assert np.array_equal(idx.codes, py.codes)
    assert np.array_equal(idx.centroids, py.centroids)
    np.testing.assert_array_equal(idx.factors, py.factors)
    assert np.array_equal(idx.base, py.base)
    queries, _, _ = synth.mixture(12, d, k, sigma=0.6, seed=8)
    for q in queries:
        for probe, topk in ((3, 10), (8, 5), (100, 1)):
            oracle.metrics_reset()
            dist, ids = idx.query(q, probe, topk, heuristic)
            out, cnt = py.query(q, probe, topk, heuristic)
            m = oracle.metrics()
            assert m["rough"] == cnt["rough"] and m["precise"] == cnt["precise"] and m["query"] == 1
            assert [i for _, i in out] == ids.tolist()          # same order, not just the same set
            assert [np.float32(a).tobytes() for a, _ in out] == [np.float32(a).tobytes() for a in dist]
    idx.close()


def test_index_invariants_and_recall(oracle):
    n, d, k = 4000, 128, 16
    x, centres, _ = synth.mixture(n, d, k, sigma=0.5, seed=21)
    P = synth.random_orthogonal(d, seed=22)
    idx = oracle.OracleIndex.build(x, centres, P)
    off, ids, codes, fac = idx.offsets, idx.map_ids, idx.codes, idx.factors
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off.astype(np.int64)) >= 0)
    assert np.array_equal(np.sort(ids), np.arange(n))                     # permutation
    assert np.array_equal(idx.base, x[ids])                               # base re-ordered, un-rotated
    cent = idx.centroids
    xr = oracle.project_rows(x, P)
    for c in range(k):                                                    # rabitq.rs:232-238
        seg = ids[off[c]:off[c + 1]]
        dist = np.array([oracle.l2_squared_distance(cent[c], xr[i]) for i in seg])
        assert np.all(np.diff(dist) >= 0)
        same = np.where(np.diff(dist) == 0)[0]
        assert np.all(seg[same] < seg[same + 1])                          # stable: ties keep id order
    pop = np.array([sum(bin(int(w)).count("1") for w in row) for row in codes])
    np.testing.assert_array_equal(fac[:, 1], fac[:, 0] * (2 * pop - d).astype(np.float32))  # rabitq.rs:228
    # rotation is orthogonal: distances survive it to f32 rounding
    assert np.allclose((xr ** 2).sum(1), (x ** 2).sum(1), rtol=1e-4)
    # the bound is probabilistic: epsilon = 1.9 leaves a one-sided Gaussian tail P(Z > 1.9) ~ 2.9 % of
    # candidates with rough > accurate -- which is why id parity needs the ordered threshold replay
    queries, _, _ = synth.mixture(20, d, k, sigma=0.5, seed=23)
    gt = synth.brute_force_topk(x, queries, 10)
    hits = viol = tot = 0
    for qi, q in enumerate(queries):
        dist, rid = idx.query(q, k, 10)
        hits += len(set(rid.tolist()) & set(gt[qi].tolist()))
        y = idx.rotate_query(q)
        cl, cd = idx.coarse_rank(y, 2)
        for c, ycd in zip(cl, cd):
            lo, delta, s, planes = idx.query_prep(y, int(c))
            rough = idx.scan_cluster(int(c), ycd, planes, lo, np.float32(s), delta)
            acc = ((idx.base[off[c]:off[c + 1]] - q) ** 2).sum(1)
            viol += int((rough > acc).sum())
            tot += rough.size
    assert hits / (10 * len(queries)) >= 0.95
    assert 0.002 < viol / tot < 0.06
    idx.close()


def test_dump_load_roundtrip_and_layout(oracle, tmp_path):
    n, d, k = 300, 100, 5    # d = 100 pads to 128
    x, centres, _ = synth.mixture(n, d, k, sigma=0.5, seed=31)
    P = synth.random_orthogonal(128, seed=32)
    idx = oracle.OracleIndex.build(x, centres, P)
    assert idx.dim == 128 and np.all(idx.base[:, 100:] == 0)
    idx.dump_to_dir(str(tmp_path / "idx"))
    raw = (tmp_path / "idx" / "centroids.fvecs").read_bytes()
    # rotated + transposed: dim records of k floats
    assert len(raw) == 128 * (4 + 4 * k) and int.from_bytes(raw[:4], "little") == k
    first = np.frombuffer(raw[4:4 + 4 * k], dtype=np.float32)
    assert np.array_equal(first, idx.centroids[:, 0])
    raw = (tmp_path / "idx" / "factors.fvecs").read_bytes()
    assert int.from_bytes(raw[:4], "little") == 4 * n and len(raw) == 4 + 16 * n
    raw = (tmp_path / "idx" / "x_binary_vec.u64vecs").read_bytes()
    assert int.from_bytes(raw[:4], "little") == 2 * n and len(raw) == 4 + 16 * n
    raw = (tmp_path / "idx" / "offsets_ids.ivecs").read_bytes()
    assert len(raw) == 4 + 4 * (k + 1) + 4 + 4 * n
    idx2 = oracle.OracleIndex.load_from_dir(str(tmp_path / "idx"))
    for name in ("base", "orthogonal", "centroids", "offsets", "map_ids", "codes", "factors"):
        assert np.array_equal(getattr(idx, name), getattr(idx2, name)), name
    q = x[5]
    assert idx.query(q, 3, 7)[1].tolist() == idx2.query(q, 3, 7)[1].tolist()


def test_query_error_paths(oracle):
    x, centres, _ = synth.mixture(50, 64, 2, seed=41)
    idx = oracle.OracleIndex.build(x, centres, np.eye(64, dtype=np.float32))
    with pytest.raises(RuntimeError):
        idx.query(np.zeros(65, dtype=np.float32), 2, 5)      # rabitq.rs:275 assert_eq!
    with pytest.raises(RuntimeError):
        idx.query(x[0], 0, 5)                                # rabitq.rs:295 length - 1 underflow
    # query == a centroid with P = I: residual all zero => delta = 0 edge (finite rough distances)
    d, ids = idx.query(centres[0], 2, 5)
    assert len(ids) == 5 and np.all(np.isfinite(d))
