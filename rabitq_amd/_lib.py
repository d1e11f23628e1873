"""ctypes loader for librabitq_hip.so (the C ABI in include/rabitq_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable, every
operation raises.  Build it with `make -C rabitq_amd/csrc` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RABITQ_HIP_SO") or os.path.join(_HERE, "librabitq_hip.so")   # (override: kernel experiments)
CSRC = os.path.join(_HERE, "csrc")

RQ_OK = 0
RQ_ERR_EMPTY = -7
STATUS_NAMES = {0: "RQ_OK", -1: "RQ_ERR_INVALID", -2: "RQ_ERR_DIM_MISMATCH", -3: "RQ_ERR_IO", -4: "RQ_ERR_HIP",
                -5: "RQ_ERR_NO_DEVICE", -6: "RQ_ERR_UNSUPPORTED", -7: "RQ_ERR_EMPTY", -8: "RQ_ERR_OOM"}

# every symbol include/rabitq_hip.h declares (checked by tests/test_abi.py against the header)
EXPORTS = [
    "rq_version", "rq_abi_version", "rq_last_error", "rq_init", "rq_build", "rq_build_device", "rq_build_from_path", "rq_kmeans_device", "rq_builder_create", "rq_builder_assign_chunk", "rq_builder_order", "rq_builder_place_chunk", "rq_builder_finish", "rq_builder_free", "rq_builder_stats", "rq_load_dir",
    "rq_dump_dir", "rq_load_json", "rq_dump_json", "rq_free", "rq_from_arrays", "rq_info", "rq_get_array", "rq_get_device_ptr", "rq_query",
    "rq_query_batch", "rq_query_batch_device", "rq_query_batch_device_begin", "rq_query_batch_device_end", "rq_coarse_topk_device", "rq_merge_smallest_u64_device", "rq_query_batch_device_probed", "rq_query_batch_device_seeded", "rq_partition_lists", "rq_shard_index", "rq_query_batch_sharded_device", "rq_set_collectives", "rq_metrics", "rq_metrics_reset", "rq_rotate", "rq_rotate_device",
    "rq_quantize_pack",
    "rq_coarse_rank", "rq_query_prep", "rq_scan", "rq_rerank", "rq_set_profiling", "rq_set_option", "rq_last_profile",
]


class RabitqError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


ABI_VERSION = 4   # RQ_ABI_VERSION of include/rabitq_hip.h this mirror was written against


class _Sized(C.Structure):
    """Out-struct versioned by size: `struct_size` is set to the mirror's sizeof before every call."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))


class Info(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("dim", C.c_uint32), ("k", C.c_uint32), ("max_list_len", C.c_uint32),
                ("n", C.c_uint64), ("n_hbm", C.c_uint64), ("split_rows", C.c_uint32), ("reserved0", C.c_uint32)]


class MetricsT(C.Structure):
    _fields_ = [("rough", C.c_uint64), ("precise", C.c_uint64), ("query", C.c_uint64), ("miss", C.c_uint64)]


class BuildStatsT(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("ms_rotate", C.c_float), ("ms_assign", C.c_float), ("ms_quantize", C.c_float),
                ("rows_assigned", C.c_uint64), ("rows_in_hbm", C.c_uint64), ("rows_in_host_memory", C.c_uint64),
                ("rows_exact_redo", C.c_uint64)]


class ProfileT(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("coarse_fallback_rows", C.c_uint32)] + [(n, C.c_float) for n in ("ms_rotate", "ms_coarse", "ms_select", "ms_prep", "ms_group", "ms_scan",
                                         "ms_rerank", "ms_sort", "ms_replay", "ms_total")] + [
        ("scan_bytes", C.c_uint64), ("scan_candidates", C.c_uint64), ("rerank_candidates", C.c_uint64),
        ("scan_launches", C.c_uint32), ("retries", C.c_uint32),
        ("ms_scan_matrix", C.c_float), ("matrix_launches", C.c_uint32), ("matrix_pairs", C.c_uint64),
        ("matrix_subtile_steps", C.c_uint64), ("matrix_exact_steps", C.c_uint64),
        ("rerank_shadow_rejects", C.c_uint64), ("ms_early", C.c_float), ("small_batch_passes", C.c_uint32),
        ("survivor_workspace_bytes", C.c_uint64), ("segmented_passes", C.c_uint32), ("matrix_additive_launches", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RabitqError(-5, f"{SO_PATH} not built (run `make -C rabitq_amd/csrc`); there is no CPU fallback")
    # A process that also uses PyTorch must load torch's bundled ROCm runtime BEFORE this library pulls in the system
    # one: the other way round torch later reports "No HIP GPUs are available" (measured on this image).  The library
    # itself does not need torch; the import only fixes the load order when torch is installed.  Hosts that never use
    # torch set RABITQ_NO_TORCH_PRELOAD=1 and skip the (heavy) import ("Loading order").
    import importlib.util
    import sys
    if (not os.environ.get("RABITQ_NO_TORCH_PRELOAD") and "torch" not in sys.modules
            and importlib.util.find_spec("torch") is not None):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(SO_PATH)
    vp, f32p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses (host or device)
    u32, u64, i32, flt = C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    pp = C.POINTER(C.c_void_p)
    sig = {
        "rq_version": (C.c_char_p, []),
        "rq_abi_version": (C.c_uint32, []),
        "rq_last_error": (C.c_char_p, []),
        "rq_init": (i32, [C.c_int]),
        "rq_build": (i32, [f32p, u64, u32, f32p, u32, f32p, u64, pp]),
        "rq_build_device": (i32, [f32p, u64, u32, f32p, u32, f32p, u64, pp]),
        "rq_build_from_path": (i32, [C.c_char_p, C.c_char_p, f32p, u64, pp]),
        "rq_kmeans_device": (i32, [f32p, u64, u32, u32, u32, u32, u64, f32p]),
        "rq_builder_create": (i32, [u64, u32, f32p, u32, f32p, u64, u64, pp]),
        "rq_builder_assign_chunk": (i32, [vp, f32p, u64, u64]),
        "rq_builder_order": (i32, [vp]),
        "rq_builder_place_chunk": (i32, [vp, f32p, u64, u64]),
        "rq_builder_finish": (i32, [vp, pp]),
        "rq_builder_free": (None, [vp]),
        "rq_builder_stats": (i32, [vp, C.POINTER(BuildStatsT)]),
        "rq_load_dir": (i32, [C.c_char_p, pp]),
        "rq_dump_dir": (i32, [vp, C.c_char_p]),
        "rq_load_json": (i32, [C.c_char_p, pp]),
        "rq_dump_json": (i32, [vp, C.c_char_p]),
        "rq_free": (None, [vp]),
        "rq_from_arrays": (i32, [u32, u64, u32, f32p, f32p, f32p, u32p, u32p, u64p, vp, pp]),
        "rq_info": (i32, [vp, C.POINTER(Info)]),
        "rq_get_array": (i32, [vp, C.c_int, vp, u64]),
        "rq_get_device_ptr": (i32, [vp, C.c_int, pp, C.POINTER(u64)]),
        "rq_query": (i32, [vp, f32p, u32, u32, u32, C.c_int, f32p, u32p, u32p]),
        "rq_query_batch": (i32, [vp, f32p, u32, u32, u32, u32, C.c_int, f32p, u32p, u32p]),
        "rq_query_batch_device": (i32, [vp, f32p, u32, u32, u32, u32, C.c_int, f32p, u32p, u32p]),
        "rq_query_batch_device_begin": (i32, [vp, f32p, u32, u32, u32, u32, C.c_int, f32p, u32p, u32p, pp]),
        "rq_query_batch_device_end": (i32, [vp]),
        "rq_coarse_topk_device": (i32, [vp, f32p, u32, u32, u32, u32, u32, u32p, f32p]),
        "rq_merge_smallest_u64_device": (i32, [vp, u32, u32, u32, u32, vp]),
        "rq_query_batch_device_probed": (i32, [vp, f32p, u32, u32, u32p, f32p, u32, u32, C.c_int, f32p, u32p, u32p]),
        "rq_query_batch_device_seeded": (i32, [vp, f32p, u32, u32, u32p, f32p, u32, u32, C.c_int, f32p, f32p, u32p, u32p]),
        "rq_partition_lists": (i32, [vp, u32, u32p, u64p]),
        "rq_shard_index": (i32, [vp, u32p, u32, pp]),
        "rq_query_batch_sharded_device": (i32, [vp, vp, u32, u32, f32p, u32, u32, u32, u32, C.c_int, f32p, u32p, u32p]),
        "rq_set_collectives": (i32, [vp]),
        "rq_metrics": (i32, [C.POINTER(MetricsT)]),
        "rq_metrics_reset": (i32, []),
        "rq_rotate": (i32, [f32p, u64, u32, f32p, C.c_int, f32p]),
        "rq_rotate_device": (i32, [vp, f32p, u64, f32p, C.POINTER(C.c_float)]),
        "rq_quantize_pack": (i32, [f32p, u64, u32, f32p, u32, u32p, f32p, u64p, vp]),
        "rq_coarse_rank": (i32, [vp, f32p, u32, u32, u32, f32p, u32p, f32p]),
        "rq_query_prep": (i32, [vp, f32p, u32, u32p, f32p, f32p, u32p, u64p]),
        "rq_scan": (i32, [vp, u32, flt, u64p, flt, flt, flt, f32p]),
        "rq_rerank": (i32, [vp, f32p, u32p, u32, f32p]),
        "rq_set_profiling": (i32, [C.c_int]),
        "rq_set_option": (i32, [C.c_char_p, C.c_int]),
        "rq_last_profile": (i32, [C.POINTER(ProfileT)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.rq_abi_version() != ABI_VERSION:
        raise RabitqError(-6, f"{SO_PATH} has ABI revision {L.rq_abi_version()}, this mirror expects {ABI_VERSION}: rebuild "
                              "(make -C rabitq_amd/csrc)")
    _lib = L
    return L


def check(status: int):
    if status != RQ_OK:
        raise RabitqError(status, lib().rq_last_error().decode(errors="replace"))
