"""ctypes binding of ``libtwotower_hip.so`` (the C ABI declared in ``include/twotower_hip.h``).

The product has NO CPU fallback: if the shared library is missing or a symbol is
absent, importing an op raises.  ``build()`` (re)builds the library in-tree with hipcc
(``csrc/Makefile``); it cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess
import threading

_PKG = pathlib.Path(__file__).resolve().parent
# TT_LIB_PATH: load another build of the same C ABI (kernel A/B experiments); the product default is the in-tree library
LIB_PATH = pathlib.Path(os.environ["TT_LIB_PATH"]) if os.environ.get("TT_LIB_PATH") else _PKG / "libtwotower_hip.so"
ABI_VERSION = 9

TT_OK, TT_ERR_INVALID_ARG, TT_ERR_LAUNCH, TT_ERR_UNSUPPORTED, TT_ERR_WORKSPACE = range(5)
TT_OPT_SGD, TT_OPT_ADAGRAD = 0, 1
TT_IDS_UNIFORM, TT_IDS_POWERLAW = 0, 1
TT_MAX_DENSE_SEGS = 16
TT_MAX_TOWER_LAYERS = 8


class IdBuckets(C.Structure):
    """Mirror of ``tt_id_buckets`` (ABI v9): the row-range id lists the forward lookup fills for the optimizer launch."""
    _fields_ = [("counts", C.c_void_p), ("pairs", C.c_void_p), ("groups", C.c_int32), ("width", C.c_uint32),
                ("cap", C.c_int32), ("gen", C.c_uint32)]


class DenseLookup(C.Structure):
    """Mirror of ``tt_dense_lookup``: the embedding lookup fused into a tower's first Dense layer."""
    _fields_ = [("table", C.c_void_p), ("ids", C.c_void_p), ("table_rows", C.c_int64),
                ("table2", C.c_void_p), ("ids2", C.c_void_p), ("table2_rows", C.c_int64), ("oob_flag", C.c_void_p),
                ("buckets", IdBuckets)]


class DenseFwdArgs(C.Structure):
    """Mirror of ``tt_dense_fwd_args``."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("dropout_tensor_id", C.c_uint64),
                ("lookup", DenseLookup), ("relu_bits", C.c_void_p)]


class DenseBwdArgs(C.Structure):
    """Mirror of ``tt_dense_bwd_args``."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("dz", C.c_void_p), ("dx", C.c_void_p), ("dx_relu_src", C.c_void_p),
                ("dw_slabs", C.c_void_p), ("db_slabs", C.c_void_p), ("lookup", DenseLookup), ("dx_relu_bits", C.c_void_p)]


class RouteTable(C.Structure):
    """Mirror of ``tt_route_table`` (include/twotower_hip.h)."""
    _fields_ = [("ids", C.c_void_p), ("num_rows", C.c_int64), ("local_offset", C.c_int64), ("pos_flat", C.c_void_p)]


class SparsePlanArgs(C.Structure):
    """Mirror of ``tt_sparse_plan_args``."""
    _fields_ = [("ids", C.c_void_p), ("n_ids", C.c_int64), ("num_rows", C.c_int64), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("sorted_ids", C.c_void_p), ("order", C.c_void_p)]


class SparseTable(C.Structure):
    """Mirror of ``tt_sparse_table`` (one embedding table of the single-launch optimizer step)."""
    _fields_ = [("table", C.c_void_p), ("accum", C.c_void_p), ("rows", C.c_int64), ("grads", C.c_void_p),
                ("sorted_ids", C.c_void_p), ("order", C.c_void_p), ("apply_ws", C.c_void_p)]


class SparseTableIds(C.Structure):
    """Mirror of ``tt_sparse_table_ids`` (one embedding table of the optimizer step that starts from the raw ids)."""
    _fields_ = [("table", C.c_void_p), ("accum", C.c_void_p), ("rows", C.c_int64), ("grads", C.c_void_p),
                ("ids", C.c_void_p), ("apply_ws", C.c_void_p), ("buckets", IdBuckets)]


class DenseSeg(C.Structure):
    """Mirror of ``tt_dense_seg`` (include/twotower_hip.h)."""
    _fields_ = [
        ("param", C.c_void_p), ("accum", C.c_void_p), ("grad_slabs", C.c_void_p), ("grad_out", C.c_void_p),
        ("count", C.c_int64), ("slab_stride", C.c_int64), ("n_slabs", C.c_int32), ("l2", C.c_float),
    ]


class TrainStep(C.Structure):
    """Mirror of ``tt_train_step``: the whole train step behind one C call (``tt_train_step_f32``)."""
    _fields_ = [
        ("batch", C.c_int64), ("n_layers", C.c_int32), ("dims", C.c_int32 * (TT_MAX_TOWER_LAYERS + 1)),
        ("fwd", (DenseFwdArgs * 2) * TT_MAX_TOWER_LAYERS), ("bwd", (DenseBwdArgs * 2) * TT_MAX_TOWER_LAYERS),
        ("dropout_rate", C.c_float), ("dropout_seed", C.c_uint64), ("dropout_row0", C.c_uint64),
        ("scorer_precision", C.c_int32), ("inv_temperature", C.c_float),
        ("sample_weight", C.c_void_p), ("cand_prob", C.c_void_p), ("cand_ids", C.c_void_p),
        ("retrieval_ws", C.c_void_p), ("retrieval_ws_bytes", C.c_int64),
        ("lse", C.c_void_p), ("per_row", C.c_void_p), ("loss", C.c_void_p),
        ("opt", C.c_int32), ("n_tables", C.c_int32), ("tables", SparseTableIds * 3),
        ("n_segs", C.c_int32), ("segs", DenseSeg * TT_MAX_DENSE_SEGS), ("lr", C.c_float), ("eps", C.c_float),
        ("id_bucket_ws", C.c_void_p), ("id_bucket_ws_bytes", C.c_int64),
    ]


_p, _i64, _i32, _u64, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_float

# name -> (restype, argtypes); exactly the declarations of include/twotower_hip.h
SIGNATURES = {
    "tt_abi_version": (C.c_int, []),
    "tt_last_error": (C.c_char_p, []),
    "tt_abi_struct_bytes": (_i64, [_i32]),
    "tt_profile_enable": (C.c_int, [C.c_char_p, _i32]),
    "tt_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_float), _i32, C.POINTER(_i32)]),
    "tt_profile_set_stride": (C.c_int, [_i32]),
    "tt_fill_uniform_f32": (C.c_int, [_p, _i64, _u64, _u64, _i64, _f, _f, _p]),
    "tt_fill_uniform_rows_f32": (C.c_int, [_p, _i64, _i32, _i64, _i64, _u64, _u64, _f, _f, _p]),
    "tt_fill_ids_i64": (C.c_int, [_p, _i64, _u64, _u64, _i64, _i64, _i32, _p]),
    "tt_encode_ids_workspace_bytes": (_i64, [_i64]),
    "tt_encode_ids_u8": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "tt_embedding_gather_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "tt_embedding_gather2_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _i32, _i64, _p, _p]),
    "tt_embedding_gather_add_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "tt_hash_bucket_u8": (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    "tt_route_by_owner_i64": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _p, _p]),
    "tt_route_tables_by_owner_i64": (C.c_int, [_p, _i32, _i64, _i32, _i32, _p, _p, _p]),
    "tt_scatter_rows_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _p]),
    "tt_sparse_plan_workspace_bytes": (_i64, [_i64]),
    "tt_sparse_plan": (C.c_int, [_p, _i64, _i64, _p, _i64, _p, _p, _p]),
    "tt_sparse_plan_batched": (C.c_int, [_p, _i32, _p]),
    "tt_sparse_plan_max_lds_ids": (_i32, []),
    "tt_optimizer_ids_max_ids": (_i32, []),
    "tt_sparse_apply_workspace_bytes": (_i64, [_i64, _i32]),
    "tt_sparse_sgd_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _f, _p, _p]),
    "tt_sparse_adagrad_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _i64, _f, _f, _p, _p]),
    "tt_sparse_update2_f32": (C.c_int, [_i32, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i32, _i64, _f, _f, _p, _p, _p]),
    "tt_dense_fwd_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "tt_dense_fwd_dropout_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _f, _u64, _u64, _u64, _p]),
    "tt_dense_bwd_scaled_f32": (C.c_int, [_p, _p, _p, _p, _p, _f, _p, _p, _i64, _i32, _i32, _p]),
    "tt_dense_fwd_batched_f32": (C.c_int, [C.POINTER(DenseFwdArgs), _i32, _i64, _i32, _i32, _i32, _f, _u64, _u64, _p]),
    "tt_dense_bwd_batched_f32": (C.c_int, [C.POINTER(DenseBwdArgs), _i32, _f, _i64, _i32, _i32, _p]),
    "tt_dense_bwd_batched_update_f32": (C.c_int, [C.POINTER(DenseBwdArgs), _i32, _f, _i64, _i32, _i32, C.POINTER(DenseSeg), _i32, _i32,
                                                  _f, _f, _p]),
    "tt_tower_bwd2_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "tt_tower_bwd2_workspace_bytes": (_i64, [_i64]),
    "tt_tower_bwd2_batched_f32": (C.c_int, [C.POINTER(DenseBwdArgs), C.POINTER(DenseBwdArgs), _i32, _f, _f, _i64, _i32, _i32, _i32, _p, _p]),
    "tt_tower_fwd2_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "tt_tower_fwd2_batched_f32": (C.c_int, [C.POINTER(DenseFwdArgs), C.POINTER(DenseFwdArgs), _i32, _i64, _i32, _i32, _i32, _f, _u64, _u64, _p]),
    "tt_dense_bwd_num_slabs": (_i32, [_i64]),
    "tt_dense_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "tt_dense_update_f32": (C.c_int, [C.POINTER(DenseSeg), _i32, _i32, _i32, _f, _f, _p]),
    "tt_optimizer_step_f32": (C.c_int, [_i32, C.POINTER(SparseTable), _i32, _i32, _i64, C.POINTER(DenseSeg), _i32, _f, _f, _p]),
    "tt_optimizer_step_ids_f32": (C.c_int, [_i32, C.POINTER(SparseTableIds), _i32, _i32, _i64, C.POINTER(DenseSeg), _i32, _f, _f, _p]),
    "tt_optimizer_ids_geometry": (C.c_int, [C.POINTER(_i64), _i32, _i32, _i64, C.POINTER(DenseSeg), _i32, C.POINTER(_i32),
                                            C.POINTER(C.c_uint32), C.POINTER(_i32)]),
    "tt_id_buckets_workspace_bytes": (_i64, []),
    "tt_id_range_load": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(_i64), _i32, _i32, _i64, C.POINTER(DenseSeg), _i32, _p, _p]),
    "tt_train_step_f32": (C.c_int, [C.POINTER(TrainStep), _p]),
    "tt_retrieval_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "tt_retrieval_num_splits": (_i32, [_i64, _i64, _i32, _i32]),
    "tt_retrieval_fwd_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "tt_retrieval_rank_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "tt_retrieval_fwd_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "tt_retrieval_fwd_bf16x3_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "tt_retrieval_rank_bf16x3_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _f, _p, _p, _p, _i64, _p, _p]),
    "tt_retrieval_fwd_bwd_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "tt_retrieval_fwd_bwd_bf16x3_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "tt_retrieval_hard_negative_thresholds_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _i32, _p, _i64, _p, _i64, _p, _p]),
    "tt_retrieval_rank_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _f, _p, _p, _p, _i64, _p, _p]),
    "tt_retrieval_batch_rank_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _i64, _p, _p]),
    "tt_retrieval_bwd_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _i64, _f, _p, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _p]),
}

_lock = threading.Lock()
_lib = None


class TwoTowerHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> pathlib.Path:
    """Compile every HIP source for gfx950 into ``libtwotower_hip.so`` (in-tree)."""
    cmd = ["make", "-C", str(_PKG / "csrc"), f"-j{min(8, os.cpu_count() or 1)}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise TwoTowerHipError(f"building libtwotower_hip.so failed (exit {res.returncode})")
    return LIB_PATH


def load() -> C.CDLL:
    """Load the library and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # The library and torch must share ONE HIP runtime (tensors and streams come from torch's).
        # Both link `libamdhip64.so.7`; the first one loaded wins, so torch goes first.
        import torch  # noqa: F401
        if not LIB_PATH.exists():
            raise TwoTowerHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise TwoTowerHipError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
        got = lib.tt_abi_version()
        if got != ABI_VERSION:
            raise TwoTowerHipError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}")
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != TT_OK:
        msg = load().tt_last_error().decode("utf-8", "replace")
        exc = {TT_ERR_INVALID_ARG: ValueError, TT_ERR_UNSUPPORTED: NotImplementedError}.get(rc, TwoTowerHipError)
        raise exc(f"{what} failed (code {rc}): {msg}")


def profile_enable(tags: str = "", capacity: int = 4096) -> None:
    """Enable the built-in hipEvent kernel timing for the comma-separated tags ("" disables)."""
    check(load().tt_profile_enable(tags.encode(), capacity), "tt_profile_enable")


def profile_set_stride(stride: int) -> None:
    """Bracket only every stride-th launch of a tag with hipEvents (each record stalls the stream for 4-7 us)."""
    check(load().tt_profile_set_stride(stride), "tt_profile_set_stride")


def profile_read(tag: str, capacity: int = 4096) -> tuple[list[float], int]:
    """(durations in ms of the recorded launches, their number); synchronises and clears the tag."""
    buf = (C.c_float * capacity)()
    n = _i32(0)
    check(load().tt_profile_read(tag.encode(), buf, capacity, C.byref(n)), "tt_profile_read")
    return list(buf[:min(n.value, capacity)]), n.value
