"""Round-4 experiment (CPU, numpy): how loose does the matrix-core gate of the dim-128 scan get if the rank-5 threshold
S*(c, q) = sum_r u'_c[r] v'_q[r] + v4_q is replaced by an ADDITIVE lower bound  R_q + G_c(T)  (so that the bf16 threshold
MFMA can go)?  Bench geometry: centres N(0, I_128), vectors = centre + 0.5 N(0, I), 4096 lists of ~24.4 k, nprobe 64,
top-10.  One target list is materialised (codes + factors, cluster order), the queries of a 65 536-query batch that probe
it as a NON-nearest list are quantised against its centroid exactly as src/rabitq.rs:304-317 does, their settled
thresholds are sampled from the own-list distance distribution, and the fraction of 32 x 32 sub-tile steps that would take
the exact path is counted for the exact threshold and for several additive forms / query orders.

Not product code, not a parity artefact: a sizing experiment for DESIGN.md.
"""
import sys
import numpy as np

D, K, NPROBE, TOPK, SIGMA = 128, 4096, 64, 10, 0.5
LIST_LEN = 24414
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 65536

centres = rng.standard_normal((K, D)).astype(np.float32)

# ---- candidates of the target list: residuals, codes, factors (src/rabitq.rs:205-229), cluster order (:232-243)
res = (SIGMA * rng.standard_normal((LIST_LEN, D))).astype(np.float32)
cds = (res.astype(np.float64) ** 2).sum(1)
order = np.argsort(cds, kind="stable")
res, cds = res[order], cds[order]
norm = np.sqrt(cds)
bits = res > 0
ip = np.abs(res).sum(1) / (norm * np.sqrt(D))
xc_over_ip = norm / ip
eb = (2 * 1.9 / np.sqrt(D - 1)) * np.sqrt(np.maximum(xc_over_ip ** 2 - cds, 0))
fip = -2 / np.sqrt(D) * xc_over_ip
pop = bits.sum(1)
ssum = 2 * pop - D
ppc = fip * ssum
U = np.stack([1 / fip, cds / fip, ppc / fip, eb / fip], 1)  # u'_c (n, 4)

# ---- queries: mixture draws; keep those whose nprobe nearest centroids include the target as a non-nearest one
own = rng.integers(0, K, NQ)
qn = rng.standard_normal((NQ, D)).astype(np.float32)
Y = centres[own] + SIGMA * qn
d2 = ((Y ** 2).sum(1)[:, None] - 2 * Y @ centres.T + (centres ** 2).sum(1)[None, :])
kth = np.partition(d2, NPROBE - 1, axis=1)[:, NPROBE - 1]
cnt = (d2 <= kth[:, None]).sum(0)
want = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5   # quantile of the per-list probe count the target list sits at
tgt = int(np.argsort(cnt)[int(want * (K - 1))])
print(f"pairs per list: min {cnt.min()} median {int(np.median(cnt))} mean {cnt.mean():.0f} p90 {int(np.percentile(cnt, 90))} max {cnt.max()}; target list {tgt}: {cnt[tgt]} pairs, |c|^2 = {(centres[tgt] ** 2).sum():.1f}")
sel = (d2[:, tgt] <= kth) & (d2.argmin(1) != tgt)
Yq = Y[sel].astype(np.float64)
nq = Yq.shape[0]
# settled threshold = 10th smallest of LIST_LEN own-list distances sigma^2 |n - n'|^2, sampled through its scalar law
qn2 = (qn[sel].astype(np.float64) ** 2).sum(1)
thr = np.empty(nq)
for i in range(nq):
    n2 = rng.chisquare(D, LIST_LEN)
    dot = rng.standard_normal(LIST_LEN) * np.sqrt(qn2[i])
    dd = SIGMA ** 2 * (qn2[i] + n2 - 2 * dot)
    thr[i] = np.partition(dd, TOPK - 1)[TOPK - 1]
print(f"queries probing the list as a non-nearest one: {nq} of {NQ}; thr mean {thr.mean():.1f}")

r = Yq - centres[tgt].astype(np.float64)
ycd = (r ** 2).sum(1)
lower, upper = r.min(1), r.max(1)
delta = (upper - lower) / 15.0
qc = np.rint((r - lower[:, None]) / delta[:, None]).astype(np.int64)
sumq = qc.sum(1)
inv2d = 0.5 / delta
V = np.stack([(thr - ycd) * inv2d, -inv2d, -lower * inv2d, np.sqrt(ycd) * inv2d], 1)  # v'_q (nq, 4)
qb = (np.abs(thr - ycd) + cds.max() + np.abs(lower) * np.abs(ppc).max() + np.sqrt(ycd) * eb.max()) * np.abs(1 / fip).max() * inv2d
margin = 2 + qb / 8192 + (qb + sumq) / 262144
v4 = 0.5 * sumq - margin
print(f"ycd mean {ycd.mean():.0f}  delta mean {delta.mean():.3f}  v' means {V.mean(0)}  v' std {V.std(0)}  margin mean {margin.mean():.2f}")
print(f"u' means {U.mean(0)}  u' std {U.std(0)}")

s = bits.astype(np.float32) @ qc.T.astype(np.float32)  # (n, nq) exact integers
Sstar = U @ V.T + v4[None, :]
gap = Sstar - s
print(f"cells {s.size}: s > S* in {np.count_nonzero(gap < 0)} cells; gap mean {gap.mean():.1f} std {gap.std():.1f} min {gap.min():.1f}")

NST = LIST_LEN // 32  # whole sub-tiles only
def step_rate(LB, perm, label):
    """fraction of (sub-tile, query tile) steps with any cell s > LB, queries taken in order `perm`"""
    nqt = nq // 32
    fl = (s[: NST * 32][:, perm[: nqt * 32]] > LB[: NST * 32][:, perm[: nqt * 32]])
    fl = fl.reshape(NST, 32, nqt, 32).any(axis=(1, 3))
    print(f"  {label:58s} flagged steps {fl.mean():.2e}  ({fl.sum()} of {fl.size})")
    return fl.mean()

ident = np.arange(nq)
print("exact rank-5 threshold (what the bf16 MFMA computes):")
step_rate(Sstar, ident, "arrival order")

def additive(perm, nseg, label, tile_ref="mid"):
    """LB(c, q) = sum_r [U0_seg[r] v_q[r] + d_c[r] v0_T[r] - |d_c[r]| dv_T[r]] + v4_q with d_c = u_c - U0_seg (a reference per
    list segment: nseg segments of the cds-ordered list), v0_T / dv_T = centre / half-width of v over query tile T"""
    nqt = nq // 32
    P = perm[: nqt * 32]
    Vt = V[P].reshape(nqt, 32, 4)
    v0 = 0.5 * (Vt.max(1) + Vt.min(1))
    dv = 0.5 * (Vt.max(1) - Vt.min(1))
    LB = np.empty((NST * 32, nqt * 32))
    seg_edges = np.linspace(0, NST * 32, nseg + 1).astype(int)
    for g in range(nseg):
        a, b = seg_edges[g], seg_edges[g + 1]
        U0 = 0.5 * (U[a:b].max(0) + U[a:b].min(0))
        U0[2] = 0.0
        d = U[a:b] - U0
        base = V[P] @ U0 + v4[P]                      # per query (the C operand)
        cand = d @ v0.T - np.abs(d) @ dv.T             # (cands, tiles): the per-lane compare operand of a step
        LB[a:b] = base[None, :] + np.repeat(cand, 32, axis=1)
    nflag = (s[: NST * 32][:, P] > LB).reshape(NST, 32, nqt, 32).any(axis=(1, 3))
    slack = (Sstar[: NST * 32][:, P] - LB)
    print(f"  {label:58s} flagged steps {nflag.mean():.2e}  ({nflag.sum()} of {nflag.size})  slack mean {slack.mean():.1f} p99 {np.percentile(slack, 99):.1f}")

def order_by(keys, buckets):
    """lexicographic bucketing: keys[0] cut into buckets[0] quantile groups, inside each keys[1] into buckets[1], ..."""
    idx = [np.arange(nq)]
    for k, b in zip(keys, buckets):
        nxt = []
        for grp in idx:
            o = grp[np.argsort(k[grp], kind="stable")]
            nxt += list(np.array_split(o, b))
        idx = nxt
    return np.concatenate(idx)

print("additive bound, one reference per LIST:")
additive(ident, 1, "arrival order")
additive(np.argsort(V[:, 0]), 1, "sorted by v'0 = (thr-ycd)/(2 delta)")
additive(np.argsort(V[:, 2]), 1, "sorted by v'2 = -lower/(2 delta)")
for b0, b2 in ((4, 8), (8, 4), (6, 6)):
    additive(order_by([V[:, 0], V[:, 2]], [b0, b2]), 1, f"{b0} buckets of v'0 x {b2} buckets of v'2 (then v'0 again)")
print("additive bound, a reference per list SEGMENT (cds order):")
for nseg in (4, 16, 64):
    additive(ident, nseg, f"{nseg} segments, arrival order")
    additive(np.argsort(V[:, 2]), nseg, f"{nseg} segments, sorted by v'2")
    additive(order_by([V[:, 0], V[:, 2]], [4, 8]), nseg, f"{nseg} segments, 4 x 8 buckets")
print("additive bound, a reference per 96-candidate wave (= exact McCormick-style centre per wave):")
additive(ident, NST // 3, "per wave, arrival order")
additive(np.argsort(V[:, 2]), NST // 3, "per wave, sorted by v'2")

def additive_list_level(perm, label, tile_r=()):
    """references on BOTH sides per list: U0 = mean of u' over the list, V0 / DV = centre / half-range of v' over ALL the list's
    pairs; the rows r in tile_r take per-tile v0_T / dv_T instead"""
    nqt = nq // 32
    P = perm[: nqt * 32]
    U0 = U.mean(0)
    U0[2] = 0.0
    d = U[: NST * 32] - U0
    V0 = 0.5 * (V.max(0) + V.min(0))
    DV = 0.5 * (V.max(0) - V.min(0))
    Vt = V[P].reshape(nqt, 32, 4)
    v0 = np.repeat((0.5 * (Vt.max(1) + Vt.min(1)))[None], 1, 0)[0].copy()
    dv = (0.5 * (Vt.max(1) - Vt.min(1))).copy()
    for r in range(4):
        if r not in tile_r:
            v0[:, r] = V0[r]
            dv[:, r] = DV[r]
    base = V[P] @ U0 + v4[P]
    cand = d @ v0.T - np.abs(d) @ dv.T
    LB = base[None, :] + np.repeat(cand, 32, axis=1)
    nflag = (s[: NST * 32][:, P] > LB).reshape(NST, 32, nqt, 32).any(axis=(1, 3))
    slack = (Sstar[: NST * 32][:, P] - LB)
    print(f"  {label:58s} flagged steps {nflag.mean():.2e}  ({nflag.sum()} of {nflag.size})  slack mean {slack.mean():.1f} p99 {np.percentile(slack, 99):.1f} max {slack.max():.1f}")

print("both references per LIST (no per-tile scalars at all):")
additive_list_level(ident, "arrival order, all four rows list-level")
additive_list_level(ident, "rows 0 and 2 per tile", (0, 2))
additive_list_level(np.argsort(V[:, 2]), "rows 0 and 2 per tile, sorted by v'2", (0, 2))
additive_list_level(ident, "row 2 per tile", (2,))
additive_list_level(np.argsort(V[:, 2]), "row 2 per tile, sorted by v'2", (2,))
