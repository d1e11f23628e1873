"""BASELINE.json configs[3] at FULL size on one GPU: 100M x 768 (307 GB of raw vectors, more than the HBM) through
the streamed two-pass builder (rq_builder_*), raw vectors never resident: every chunk is generated twice from its own
seed.  The re-ranker gathers raw rows (src/rerank.rs:85-90) that no longer fit HBM, so the per-list HBM / pinned-host
tiers are what this test is about.  No oracle at this size: size-independent properties
(tier split, permutation, list sizes, sampled code bits and factors, batch == single query, exact distances of the
returned ids) and recall@10 against a streamed f64 brute force over 256 queries.
Reference: src/rabitq.rs:188-189 (rotation), :232-247 (cluster ordering), src/rerank.rs:85-90 (gather)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def test_100m_x_768_streamed_build_and_query():
    import torch
    import rabitq_amd as rq
    from rabitq_amd import _lib
    _lib.check(_lib.lib().rq_init(0))
    dev = torch.device("cuda", 0)
    n, d, k, probe, topk, nq, sigma = 100_000_000, 768, 4096, 64, 10, 256, 0.5
    centres = synth.device_centres(k, d, dev, 1.0, seed=31)
    q = synth.device_queries(centres, nq, sigma, dev, seed=32)
    P = synth.random_orthogonal(d, seed=33)
    chunk = (512 << 20) // d
    chunks = [(ci, i0, min(chunk, n - i0)) for ci, i0 in enumerate(range(0, n, chunk))]

    def gen(ci, i0, m):
        return synth.device_mixture_chunk(centres, i0, m, sigma, ci, seed_base=7000)

    sample = np.arange(0, n, n // 500)[:500]           # rows whose codes / factors are checked bit for bit
    sample_rows = torch.empty((len(sample), d), device=dev)
    counts = torch.zeros(k, dtype=torch.int64, device=dev)
    qd = q.double()
    qn = (qd * qd).sum(1, keepdim=True)
    best = torch.full((nq, topk), float("inf"), device=dev, dtype=torch.float64)
    besti = torch.full((nq, topk), -1, device=dev, dtype=torch.int64)
    b = rq.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=P)          # automatic HBM budget
    for ci, i0, m in chunks:                                                   # ---- pass 1
        x, u = gen(ci, i0, m)
        counts += torch.bincount(u, minlength=k)
        sel = np.nonzero((sample >= i0) & (sample < i0 + m))[0]
        if len(sel):
            sample_rows[torch.from_numpy(sel).to(dev)] = x[torch.from_numpy(sample[sel] - i0).to(dev)]
        for j0 in range(0, m, 350_000):                                        # streamed f64 ground truth
            xb = x[j0:j0 + 350_000].double()
            d2 = qn - 2.0 * (qd @ xb.T) + (xb * xb).sum(1)[None, :]
            cd, cj = torch.topk(d2, topk, dim=1, largest=False)
            alld, alli = torch.cat([best, cd], 1), torch.cat([besti, cj + i0 + j0], 1)
            s2 = torch.topk(alld, topk, dim=1, largest=False).indices
            best, besti = torch.gather(alld, 1, s2), torch.gather(alli, 1, s2)
            del xb, d2
        torch.cuda.synchronize()
        b.assign_chunk(x.data_ptr(), i0, m)
        del x, u
    gt = besti.cpu().numpy()
    del best, besti
    torch.cuda.empty_cache()        # the engine sizes the HBM tier by what is free now
    b.order()
    st = b.stats()
    assert st["rows_assigned"] == n and st["rows_in_hbm"] + st["rows_in_host_memory"] == n
    assert 0 < st["rows_in_host_memory"] < n // 2, st      # 307 GB do not fit: a host tier exists, the bulk stays in HBM
    for ci, i0, m in chunks:                                                   # ---- pass 2
        x, _ = gen(ci, i0, m)
        x = x.contiguous()
        torch.cuda.synchronize()
        b.place_chunk(x.data_ptr(), i0, m)
        del x
    idx = b.finish()
    torch.cuda.empty_cache()
    assert (idx.n, idx.dim, idx.k) == (n, d, k) and idx.n_hbm == st["rows_in_hbm"]

    off, ids = idx.offsets.astype(np.int64), idx.map_ids
    lens = np.diff(off)
    assert off[0] == 0 and off[-1] == n and np.all(lens >= 0)
    assert np.array_equal(lens, counts.cpu().numpy())                       # well separated mixture: list = generating centre
    # tier split: list c keeps floor(len_c * budget_rows / n) members in HBM for ONE budget; the sum is what the index reports
    lo_b, hi_b = idx.n_hbm, n
    while lo_b < hi_b:                                                         # smallest budget giving at least n_hbm rows
        mid = (lo_b + hi_b) // 2
        if int((lens * mid // n).sum()) >= idx.n_hbm:
            hi_b = mid
        else:
            lo_b = mid + 1
    assert int((lens * lo_b // n).sum()) == idx.n_hbm
    ids_d = torch.from_numpy(ids.view(np.int32)).to(dev)                       # n < 2^31
    assert torch.equal(torch.sort(ids_d).values, torch.arange(n, device=dev, dtype=torch.int32))   # a permutation
    pos_of = torch.empty(n, dtype=torch.int64, device=dev)
    pos_of[ids_d.long()] = torch.arange(n, device=dev)
    spos = pos_of[torch.from_numpy(sample).to(dev)].cpu().numpy()
    del ids_d, pos_of
    codes, fac = idx.codes, idx.factors
    pop = np.zeros(2000, np.int64)
    for w in range(codes.shape[1]):
        pop += np.array([bin(int(v)).count("1") for v in codes[:2000, w]])
    np.testing.assert_array_equal(fac[:2000, 1], fac[:2000, 0] * (2 * pop - d).astype(np.float32))   # factor_ppc = factor_ip * (2 popcount - D)
    xs = rq.ops.rotate(sample_rows.cpu().numpy(), P)
    lab = np.searchsorted(off, spos, side="right") - 1
    r = xs - idx.centroids[lab]
    want = np.zeros((len(sample), d // 64), np.uint64)
    for j in range(d):
        want[:, j // 64] |= (r[:, j] > 0).astype(np.uint64) << np.uint64(j % 64)
    assert np.array_equal(codes[spos], want)                                   # sign bits of the rotated residual
    np.testing.assert_allclose(fac[spos, 3], (r.astype(np.float64) ** 2).sum(1), rtol=1e-5)
    # within a list, members are ordered by centre distance (src/rabitq.rs:232-238): check three lists
    for c in (0, k // 2, k - 1):
        cds = fac[off[c]:off[c + 1], 3]
        assert np.all(np.diff(cds) >= 0)
    del codes, fac

    # ---- queries: batch == one at a time, recall, exact distances of the returned ids ---------------------------------
    qh = q.cpu().numpy()
    dist, got, cnt = idx.query_batch(qh, probe, topk)
    for j in (0, 7, 100):
        single = idx.query(qh[j], probe, topk)
        assert [i for _, i in single] == got[j, :cnt[j]].tolist()
        assert np.array_equal(np.array([v for v, _ in single], np.float32).view(np.uint32), dist[j, :cnt[j]].view(np.uint32))
    hd, hg, hc = idx.query_batch(qh[:64], probe, topk, heuristic_rank=True)
    recall = np.mean([len(set(got[j, :topk].tolist()) & set(gt[j].tolist())) / topk for j in range(nq)])
    assert recall >= 0.95, recall
    assert np.mean([len(set(hg[j, :topk].tolist()) & set(gt[j].tolist())) / topk for j in range(64)]) >= 0.95
    want_rows = {}
    for j in range(4):                                                         # regenerate the chunks of the returned ids
        for e in range(int(cnt[j])):
            want_rows.setdefault(int(got[j, e]) // chunk, []).append((j, e, int(got[j, e])))
    for ci, hits in want_rows.items():
        i0 = ci * chunk
        x, _ = gen(ci, i0, min(chunk, n - i0))
        for j, e, gid in hits:
            exact = float(((x[gid - i0].double() - qd[j]) ** 2).sum())
            np.testing.assert_allclose(dist[j, e], exact, rtol=1e-5)
        del x
    idx.close()
