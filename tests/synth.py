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
