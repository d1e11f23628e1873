"""fvecs / ivecs / u64vecs framing used by the reference's datasets and index directory
(src/utils.rs:280-364): records of [u32 LE count][count x element LE].  Host-side plumbing for
callers that keep their data in those files; the index directory itself is read and written by the
C++ side (rq_load_dir / rq_dump_dir)."""
from __future__ import annotations

import numpy as np


def read_vecs(path, dtype=np.float32) -> list:
    """utils.rs:280-303 (`read_vecs`) / :309-330 (`read_u64_vecs`): a list of 1-D arrays."""
    raw = np.fromfile(path, dtype=np.uint8)
    dtype = np.dtype(dtype)
    out, off = [], 0
    while off + 4 <= raw.size:
        cnt = int(raw[off:off + 4].view("<u4")[0])
        off += 4
        nbytes = cnt * dtype.itemsize
        out.append(raw[off:off + nbytes].view(dtype).copy())
        off += nbytes
    return out


def read_matrix(path, dtype=np.float32) -> np.ndarray:
    """utils.rs:44-49 (`matrix_from_fvecs`): all records have the same length."""
    recs = read_vecs(path, dtype)
    return np.stack(recs) if recs else np.zeros((0, 0), dtype=dtype)


def write_vecs(path, records) -> None:
    """utils.rs:350-364 (`write_vecs`) and :333-347 (`write_matrix`, one record per row)."""
    with open(path, "wb") as f:
        for r in records:
            r = np.ascontiguousarray(r)
            f.write(np.uint32(r.size).tobytes())
            f.write(r.tobytes())
