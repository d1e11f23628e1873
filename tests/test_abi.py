"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a
GPU, exports every symbol include/rabitq_hip.h declares, and refuses to compute without a device
(no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from rabitq_amd import _lib
    _lib.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    from rabitq_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rabitq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rq_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_reference():
    from rabitq_amd import _lib
    import ctypes as C
    assert C.sizeof(_lib.MetricsT) == 32            # 4 x u64, src/metrics.rs:7-18
    assert C.sizeof(_lib.Info) == 24
    # Factor is repr(C) 4 x f32 (src/rabitq.rs:21-32): (n, 4) f32 arrays are passed as rq_factor_t*
    assert np.dtype(np.float32).itemsize * 4 == 16


def test_no_cpu_fallback_without_device(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import rabitq_amd
    with pytest.raises(rabitq_amd.RabitqError) as e:
        rabitq_amd.RaBitQ.build(np.zeros((4, 64), np.float32), np.zeros((2, 64), np.float32))
    assert e.value.status == -5
    with pytest.raises(rabitq_amd.RabitqError):
        rabitq_amd.ops.rotate(np.zeros((4, 64), np.float32), np.eye(64, dtype=np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rabitq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "rabitq_oracle" not in text, f


def test_vecs_roundtrip(tmp_path):
    from rabitq_amd import vecs
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    vecs.write_vecs(tmp_path / "a.fvecs", a)
    assert np.array_equal(vecs.read_matrix(tmp_path / "a.fvecs"), a)
    raw = (tmp_path / "a.fvecs").read_bytes()
    assert len(raw) == 3 * (4 + 16) and int.from_bytes(raw[:4], "little") == 4
    vecs.write_vecs(tmp_path / "b.u64vecs", [np.array([1, 2**63], dtype=np.uint64)])
    assert vecs.read_vecs(tmp_path / "b.u64vecs", np.uint64)[0].tolist() == [1, 2**63]
