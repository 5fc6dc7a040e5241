"""The C-ABI library loads (no GPU needed) and exports every symbol include/twotower_hip.h declares."""
import pathlib
import re

from two_tower_amazon_recommender_amd import _lib

ROOT = pathlib.Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "twotower_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_bound_and_exported():
    syms = declared_symbols()
    assert len(syms) >= 18
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES must mirror the header exactly"
    lib = _lib.load()
    for s in syms:
        assert hasattr(lib, s), s


def test_abi_version_and_host_only_queries():
    lib = _lib.load()
    assert lib.tt_abi_version() == _lib.ABI_VERSION
    assert lib.tt_retrieval_workspace_bytes(8192, 8192, 128) > 0
    assert lib.tt_retrieval_workspace_bytes(0, 10, 128) == 0
    assert lib.tt_sparse_plan_workspace_bytes(8192) >= 256
    assert lib.tt_dense_bwd_num_slabs(8192) == 32 and lib.tt_dense_bwd_num_slabs(1) == 1


def test_invalid_args_return_codes_without_gpu():
    """Argument validation happens before any launch, so it is checkable on CPU."""
    lib = _lib.load()
    rc = lib.tt_embedding_gather_f32(None, 10, 6, None, 4, None, None, None)   # dim % 4 != 0
    assert rc == _lib.TT_ERR_INVALID_ARG
    assert b"multiple of 4" in lib.tt_last_error()
    rc = lib.tt_retrieval_fwd_f32(None, None, 4, 4, 128, 0, 10.0, None, None, None, None, None, 0, None, None, None, None)
    assert rc == _lib.TT_ERR_INVALID_ARG
    rc = lib.tt_dense_fwd_f32(None, None, None, None, 8, 6, 8, 0, None)
    assert rc == _lib.TT_ERR_INVALID_ARG


def test_train_step_struct_is_validated_before_any_launch():
    """tt_train_step_f32 (the composite entry, ABI v8) refuses a malformed step description with TT_ERR_INVALID_ARG and a
    message - layer count, batch, dropout rate, precision code, table / segment counts - before it touches the GPU; the
    two query entries of the fused tower answer on the host."""
    import ctypes as C
    lib = _lib.load()
    st = _lib.TrainStep()
    assert lib.tt_train_step_f32(None, None) == _lib.TT_ERR_INVALID_ARG
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"layers" in lib.tt_last_error()
    st.n_layers, st.batch = 2, 0
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"batch" in lib.tt_last_error()
    st.batch, st.dropout_rate = 256, 1.5
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"dropout" in lib.tt_last_error()
    st.dropout_rate, st.scorer_precision = 0.0, 7
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"precision" in lib.tt_last_error()
    st.scorer_precision, st.n_tables = 1, 1
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"tables" in lib.tt_last_error()
    st.n_tables, st.n_segs = 2, 0
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG and b"segments" in lib.tt_last_error()
    st.n_layers = _lib.TT_MAX_TOWER_LAYERS + 1
    assert lib.tt_train_step_f32(C.byref(st), None) == _lib.TT_ERR_INVALID_ARG
    assert lib.tt_tower_fwd2_supported(8192, 128, 256, 128) == 1 and lib.tt_tower_fwd2_supported(8192, 128, 512, 256) == 0
    assert lib.tt_tower_fwd2_supported(8192, 36, 128, 128) == 0 and lib.tt_tower_fwd2_supported(8192, 1024, 128, 128) == 0
