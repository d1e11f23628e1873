"""Seeded synthetic inputs shared by the tests, smoke() and bench.py."""
import numpy as np


def random_orthogonal(dim: int, seed: int = 99) -> np.ndarray:
    """Q factor of a seeded N(0,1) matrix (what utils.rs:16-20 draws unseeded), f32, row-major."""
    rng = np.random.default_rng(seed)
    q, r = np.linalg.qr(rng.standard_normal((dim, dim)))
    q = q * np.sign(np.diag(r))  # fix the sign convention so the matrix is unique per seed
    return np.ascontiguousarray(q, dtype=np.float32)


def mixture(n: int, d: int, k: int, sigma: float = 0.25, seed: int = 42, centre_seed: int = 1234,
            centre_scale: float = 1.0):
    """k centres ~ N(0, centre_scale^2 I); points = centre[u] + sigma * N(0, I), u uniform."""
    crng = np.random.default_rng(centre_seed)
    centres = (crng.standard_normal((k, d)) * centre_scale).astype(np.float32)
    rng = np.random.default_rng(seed)
    u = rng.integers(0, k, size=n)
    x = centres[u] + sigma * rng.standard_normal((n, d)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), centres, u


def brute_force_topk(base: np.ndarray, queries: np.ndarray, topk: int) -> np.ndarray:
    d2 = ((queries[:, None, :].astype(np.float64) - base[None, :, :].astype(np.float64)) ** 2).sum(-1)
    return np.argsort(d2, axis=1, kind="stable")[:, :topk].astype(np.int32)


# ---- device-side generators (torch-ROCm; bench.py, scripts/, the full-size -m gpu tests) -------------------
def device_mixture_chunk(centres, i0: int, m: int, sigma: float, chunk_id: int, seed_base: int = 42, lo: int = 0,
                         k_local: int | None = None, weights=None):
    """Rows [i0, i0+m) of the synthetic base: centre[u] + sigma * N(0, I), seeded per chunk, so any chunk can be
    regenerated on its own (the two-pass streamed build of a beyond-HBM index feeds every chunk twice).
    `weights` (k_local probabilities) draws unbalanced lists; None = uniform."""
    import torch
    dev = centres.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed_base + chunk_id)
    k_local = centres.shape[0] if k_local is None else k_local
    if weights is None:
        u = torch.randint(0, k_local, (m,), generator=g, device=dev) + lo
    else:
        u = torch.multinomial(weights, m, replacement=True, generator=g) + lo
    return centres[u] + sigma * torch.randn(m, centres.shape[1], generator=g, device=dev, dtype=torch.float32), u


def device_centres(k: int, d: int, dev, scale: float = 1.0, seed: int = 1234):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randn(k, d, generator=g, device=dev, dtype=torch.float32) * scale


def device_mixture(n: int, d: int, k: int, sigma: float, dev, chunk: int = 4_000_000, centre_scale: float = 1.0,
                   seed_base: int = 42, lo: int = 0, k_local: int | None = None, weights=None):
    """The whole n x d base in HBM (SURVEY.md section 8d mixture), generated chunk by chunk."""
    import torch
    centres = device_centres(k, d, dev, centre_scale)
    x = torch.empty((n, d), device=dev, dtype=torch.float32)
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        x[i0:i0 + m], _ = device_mixture_chunk(centres, i0, m, sigma, ci, seed_base, lo, k_local, weights)
    return x, centres


def device_queries(centres, nq: int, sigma: float, dev, seed: int = 7, weights=None):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if weights is None:
        uq = torch.randint(0, centres.shape[0], (nq,), generator=g, device=dev)
    else:
        uq = torch.multinomial(weights, nq, replacement=True, generator=g)
    return (centres[uq] + sigma * torch.randn(nq, centres.shape[1], generator=g, device=dev, dtype=torch.float32)).contiguous()
