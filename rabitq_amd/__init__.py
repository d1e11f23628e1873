"""rabitq_amd -- MI355X (gfx950) engine for the RaBitQ build/query hot path of kemingy/rabitq.

Host-side mirror of the crate's public surface (`pub use rabitq::RaBitQ`, src/lib.rs:12) over the
C ABI in include/rabitq_hip.h.  All arithmetic runs in hand-written HIP kernels
(rabitq_amd/csrc); there is no CPU fallback.
"""
from .index import RaBitQ, metrics, metrics_reset, metrics_str, calculate_recall  # noqa: F401
from . import ops, vecs  # noqa: F401
from ._lib import RabitqError, build  # noqa: F401

__all__ = ["RaBitQ", "metrics", "metrics_reset", "metrics_str", "calculate_recall", "ops", "vecs", "RabitqError",
           "build"]
