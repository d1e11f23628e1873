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
    assert C.sizeof(_lib.Info) == 40            # rq_info_t: struct_size, dim, k, max_list_len, n, n_hbm, split_rows, reserved0
    # Factor is repr(C) 4 x f32 (src/rabitq.rs:21-32): (n, 4) f32 arrays are passed as rq_factor_t*
    assert np.dtype(np.float32).itemsize * 4 == 16


def test_ctypes_mirror_matches_the_compiled_header(tmp_path):
    """sizeof / offsetof of the sized out-structs as gcc lays them out from include/rabitq_hip.h == the ctypes mirror."""
    import ctypes as C
    import subprocess
    from rabitq_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "rabitq_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu\\n", sizeof(rq_info_t), offsetof(rq_info_t, n), offsetof(rq_info_t, n_hbm), offsetof(rq_info_t, max_list_len));
    printf("%zu %zu %zu %zu\\n", sizeof(rq_build_stats_t), offsetof(rq_build_stats_t, rows_assigned), offsetof(rq_build_stats_t, rows_in_host_memory), offsetof(rq_build_stats_t, rows_exact_redo));
    printf("%zu %zu %zu %zu\\n", sizeof(rq_profile_t), offsetof(rq_profile_t, ms_rotate), offsetof(rq_profile_t, scan_bytes), offsetof(rq_profile_t, rerank_shadow_rejects));
    printf("%d\\n", RQ_ABI_VERSION);
    return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rows = [[int(v) for v in ln.split()] for ln in subprocess.check_output([str(exe)]).decode().splitlines()]
    I, B, P = _lib.Info, _lib.BuildStatsT, _lib.ProfileT
    assert rows[0] == [C.sizeof(I), I.n.offset, I.n_hbm.offset, I.max_list_len.offset]
    assert rows[1] == [C.sizeof(B), B.rows_assigned.offset, B.rows_in_host_memory.offset, B.rows_exact_redo.offset]
    assert rows[2] == [C.sizeof(P), P.ms_rotate.offset, P.scan_bytes.offset, P.rerank_shadow_rejects.offset]
    assert rows[3] == [_lib.ABI_VERSION]


def test_sized_out_structs_are_never_overrun(L):
    """A host compiled against an older, shorter struct passes its own sizeof: nothing is written beyond it; an unset
    struct_size is refused (rq_info_t / rq_profile_t grew without a version)."""
    import ctypes as C
    from rabitq_amd import _lib
    assert L.rq_abi_version() == _lib.ABI_VERSION and b"abi 4" in L.rq_version()
    buf = (C.c_uint8 * 256)(*([0xAB] * 256))
    C.cast(buf, C.POINTER(C.c_uint32))[0] = 24                      # an "old" rq_profile_t of 24 bytes
    assert L.rq_last_profile(C.cast(buf, C.POINTER(_lib.ProfileT))) == 0
    assert bytes(buf[24:]) == b"\xab" * (256 - 24)
    assert C.cast(buf, C.POINTER(C.c_uint32))[0] == 24
    C.cast(buf, C.POINTER(C.c_uint32))[0] = 0
    assert L.rq_last_profile(C.cast(buf, C.POINTER(_lib.ProfileT))) == -1
    big = (C.c_uint8 * 1024)(*([0xCD] * 1024))
    C.cast(big, C.POINTER(C.c_uint32))[0] = 1024                    # a "newer" host: only the known bytes are written
    assert L.rq_last_profile(C.cast(big, C.POINTER(_lib.ProfileT))) == 0
    assert C.cast(big, C.POINTER(C.c_uint32))[0] == C.sizeof(_lib.ProfileT)
    assert bytes(big[C.sizeof(_lib.ProfileT):]) == b"\xcd" * (1024 - C.sizeof(_lib.ProfileT))


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


def _write_dir(tmp, dim=64, k=3, n=5, **bad):
    """A well-formed five-file index directory (src/rabitq.rs:128-156), optionally broken in one place."""
    from rabitq_amd import vecs
    rng = np.random.default_rng(0)
    os.makedirs(tmp, exist_ok=True)
    ortho = [rng.standard_normal(dim).astype(np.float32) for _ in range(dim)]
    cent = [rng.standard_normal(k).astype(np.float32) for _ in range(dim)]
    base = [rng.standard_normal(dim).astype(np.float32) for _ in range(n)]
    offsets = np.array([0, 2, 2, n], np.uint32)
    ids = np.arange(n, dtype=np.uint32)
    if bad.get("short_ortho_row"):
        ortho[7] = ortho[7][:-1]
    if bad.get("ragged_centroids"):
        cent[3] = cent[3][:-1]
    if bad.get("short_base_row"):
        base[1] = base[1][:-4]
    if bad.get("bad_offsets"):
        offsets = np.array([0, 4, 2, n], np.uint32)
    if bad.get("offsets_end"):
        offsets = np.array([0, 2, 2, n - 1], np.uint32)
    vecs.write_vecs(os.path.join(tmp, "orthogonal.fvecs"), ortho)
    vecs.write_vecs(os.path.join(tmp, "centroids.fvecs"), cent)
    vecs.write_vecs(os.path.join(tmp, "base.fvecs"), base)
    recs = [offsets, ids] + ([ids] if bad.get("three_records") else [])
    vecs.write_vecs(os.path.join(tmp, "offsets_ids.ivecs"), recs)
    vecs.write_vecs(os.path.join(tmp, "factors.fvecs"), [rng.standard_normal(4 * n).astype(np.float32)])
    vecs.write_vecs(os.path.join(tmp, "x_binary_vec.u64vecs"), [np.arange(n * dim // 64, dtype=np.uint64)])


@pytest.mark.parametrize("bad", ["short_ortho_row", "ragged_centroids", "short_base_row", "bad_offsets", "offsets_end",
                                 "three_records"])
def test_load_dir_rejects_malformed_directories(L, tmp_path, bad):
    """rq_load_dir validates every record length before indexing by it (the reference's matrix_from_fvecs panics on
    ragged input): RQ_ERR_IO, no heap over-read, and no device needed to say so."""
    import ctypes as C
    _write_dir(str(tmp_path / "idx"), **{bad: True})
    h = C.c_void_p()
    st = L.rq_load_dir(os.fsencode(str(tmp_path / "idx")), C.byref(h))
    assert st == -3, (bad, st, L.rq_last_error())
    assert not h.value


@pytest.mark.parametrize("text", ["", "{", '{"dim": 64}', '{"dim":64,"base":{"nrows":64,"ncols":1,"data":[null]}}',
                                  '{"dim":63,"orthogonal":{"nrows":63,"ncols":63,"data":[]}}'])
def test_load_json_rejects_malformed_text(L, tmp_path, text):
    """rq_load_json: "deserialize error" (src/rabitq.rs:74) is RQ_ERR_IO, decided before any device work."""
    import ctypes as C
    f = tmp_path / "idx.json"
    f.write_text(text)
    h = C.c_void_p()
    assert L.rq_load_json(os.fsencode(str(f)), C.byref(h)) == -3, L.rq_last_error()
    assert not h.value
    assert L.rq_load_json(os.fsencode(str(tmp_path / "missing.json")), C.byref(h)) == -3


def _build_c_host(tmp_path):
    """gcc on tests/c_host/c_abi_host.c: a compiled host that sees only include/rabitq_hip.h and the shared library."""
    import subprocess
    from rabitq_amd import _lib
    exe = str(tmp_path / "c_abi_host")
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_host", "c_abi_host.c"), "-o", exe, "-L", libdir, "-lrabitq_hip",
                           f"-Wl,-rpath,{libdir}", "-lm"])
    return exe


def test_c_host_links_and_fails_loudly_without_device(L, tmp_path):
    """The header is plain C (compiles with gcc -Wall -Werror), the library links without torch / Python, and without a
    GPU the host gets RQ_ERR_NO_DEVICE -- not a CPU answer."""
    import subprocess
    import torch
    exe = _build_c_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_c_host_gpu.py runs the program")
    p = subprocess.run([exe, str(tmp_path / "idx")], capture_output=True, text=True)
    assert p.returncode == 2 and "no HIP device" in p.stderr, (p.returncode, p.stderr)


def test_cli_flags_are_the_reference_clis():
    """crates/cli/src/main.rs:11-38: -b -c -q -t -p -k -s and the SWITCH -h = heuristic_rank (argh keeps only --help for help),
    defaults probe 100 / topk 10."""
    from rabitq_amd import cli
    ap = cli.build_parser()
    base = ["-b", "b.fvecs", "-c", "c.fvecs", "-q", "q.fvecs", "-t", "t.ivecs", "-s", "saved"]
    a = ap.parse_args(base)
    assert (a.probe, a.topk, a.heuristic_rank) == (100, 10, False)
    a = ap.parse_args(base + ["-h", "-p", "32", "-k", "100"])
    assert (a.probe, a.topk, a.heuristic_rank) == (32, 100, True)
    assert ap.parse_args(base + ["--heuristic-rank"]).heuristic_rank
    long_form = ["--base", "b", "--centroids", "c", "--query", "q", "--truth", "t", "--saved", "s", "--probe", "7", "--topk", "3"]
    a = ap.parse_args(long_form)
    assert (a.base, a.centroids, a.query, a.truth, a.saved, a.probe, a.topk) == ("b", "c", "q", "t", "s", 7, 3)
    with pytest.raises(SystemExit) as e:     # --help is the help flag, as with argh
        ap.parse_args(["--help"])
    assert e.value.code == 0
