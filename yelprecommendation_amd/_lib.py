"""ctypes binding of the C-ABI engine (include/yelprec_engine.h).

The library is built in-tree by ``__graft_entry__.build()`` (or
``make -C yelprecommendation_amd/csrc``) as
``yelprecommendation_amd/libyelprec_engine.so``.  There is NO fallback: if the
library is missing or an entry point is absent, loading raises, and every op in
:mod:`yelprecommendation_amd.engine` raises with it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YR_ENGINE_LIB: measurement hook — an instrumented build of the same sources (scratch/inst_build.sh writes it to
# a temp directory so that the product objects and library are never overwritten); unset in every product run.
LIB_PATH = os.environ.get("YR_ENGINE_LIB") or os.path.join(_HERE, "libyelprec_engine.so")
ENGINE_VERSION = 28

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_f = C.c_float
_d = C.c_double

# name -> argtypes, mirroring include/yelprec_engine.h declaration by declaration
SIGNATURES = {
    "yr_engine_version": [],
    "yr_engine_arch": [],
    "yr_mf_score": [_p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _p, _p],
    "yr_mf_score_backward": [_p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _p, _p, _p],
    "yr_bpr_mf_fwd_bwd": [_p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _f, _p, _p, _p, _p, _p],
    "yr_bpr_mf_scatter_step": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _f,
                               _d, _d, _d, _d, _d, _d, _d, _int, _p, _p, _p, _p, _p],
    "yr_adam_dense_dual": [_p, _p, _p, _p, _i64, _p, _p, _p, _p, _i64, _int, _p, _p,
                           _d, _d, _d, _d, _d, _d, _d, _int, _p, _f, _p, _p, _p],
    "yr_bpr_mf_pull_workspace_bytes": [_i64, _i64, _i64, _int],
    "yr_bpr_mf_pull_step": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _f,
                            _d, _d, _d, _d, _d, _d, _d, _int, _int, _p, _i64, _p, _p, _p, _p, _p],
    "yr_spmm_csr": [_p, _p, _p, _p, _p, _i64, _int, _int, _p, _i64, _int, _p],
    "yr_spmm_csr_sliced": [_p, _p, _p, _p, _p, _i64, _int, _int, _p, _p],
    "yr_ngcf_score_fwd": [_p, _int, _p, _p, _p, _i64, _int, _i64, _i64, _p, _p, _p, _p],
    "yr_ngcf_score_bwd": [_p, _p, _int, _p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _p],
    "yr_ngcf_dense_fwd": [_p, _p, _p, _p, _i64, _int, _p, _p],
    "yr_ngcf_dense_bwd_data": [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p],
    "yr_ngcf_dense_bwd_weight": [_p, _p, _p, _p, _i64, _int, _p, _p, _p],
    "yr_ngcf_frontier_mark": [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _int, _p],
    "yr_ngcf_frontier_expand": [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _int, _p],
    "yr_spmm_csr_subset": [_p, _p, _p, _p, _p, _i64, _int, _int, _p, _i64, _int, _p, _p, _p, _i64, _p],
    "yr_spmm_csr_push_rows": [_p, _p, _p, _p, _p, _i64, _int, _p, _p, _i64, _p],
    "yr_ngcf_dense_fwd_rows": [_p, _p, _p, _p, _i64, _int, _p, _p, _p, _i64, _p, _p],
    "yr_ngcf_dense_bwd_data_rows": [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _i64, _p],
    "yr_ngcf_dense_bwd_weight_rows": [_p, _p, _p, _p, _i64, _int, _p, _p, _p, _p, _i64, _p],
    "yr_ngcf_step_workspace_bytes": [_i64, _int, _int, _i64],
    "yr_ngcf_bpr_step": [_p, _p, _p, _i64, _i64, _p, _i64, _int, _i64, _p, _p, _p, _int, _int, _p, _p, _p, _i64,
                         _d, _d, _d, _d, _d, _d, _d, _int, _d, _p, _i64, _p, _p, _p, _p],
    "yr_gemm_f32": [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _int, _int, _int, _p],
    "yr_gemm_f32_ex": [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _int, _int, _int, _p, _p, _p],
    "yr_cdae_decode_loss_partials": [_i64, _i64],
    "yr_cdae_decode_loss": [_p, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _i64, _p, _p, _p, _p],
    "yr_cdae_hidden_bwd": [_p, _p, _int, _p, _i64, _int, _i64, _p, _p, _p, _p, _i64, _p, _p, _p, _int, _p],
    "yr_cdae_compact_pair": [_p, _p, _i64, _i64, C.c_uint64, _d, _p, _p, _p, _p, _p, _p, _p],
    "yr_cdae_train_lists": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _int, C.c_uint64, C.c_uint64, _d, _p, _p, _p, _p, _p,
                            _p, _p, _p],
    "yr_cdae_train_lists_batched": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _i64, _d, _p, _p, _p, _p, _p,
                                    _p, _p, _p],
    "yr_cdae_loss_finalize_batched": [_p, _int, _p, _i64, _i64, _p, _p, _p, _p],
    "yr_cdae_loss_finalize": [_p, _i64, _p, _p, _p, _p],
    "yr_cdae_sampled_decode_splits": [_i64],
    "yr_cdae_sampled_decode": [_p, _p, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p, _p, _p, _p, _p],
    "yr_cdae_hidden_init": [_p, _p, _p, _p, _i64, _int, _i64, _p, _p],
    "yr_dropout": [_p, _p, _d, _i64, _p, _p],
    "yr_sigmoid": [_p, _i64, _p],
    "yr_dropout_seeded": [_p, C.c_uint64, _d, _i64, _p, _p],
    "yr_sigmoid_bwd": [_p, _p, _p, _i64, _p],
    "yr_colsum": [_p, _i64, _i64, _p, _int, _p],
    "yr_row_scatter_add": [_p, _p, _i64, _int, _i64, _p, _p],
    "yr_nsbce_fwd": [_p, _p, _p, _i64, _p, _p, _p],
    "yr_nsbce_bwd": [_p, _p, _p, _p, _p, _i64, _p, _p],
    "yr_bpr_mf_pull_index": [_p, _p, _p, _i64, _int, _i64, _i64, _p, _i64, _p, _p],
    "yr_bpr_mf_pull_apply": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _f,
                             _d, _d, _d, _d, _d, _d, _d, _int, _int, _p, _i64, _p, _p, _p, _int, _i64, _i64, _p],
    "yr_bpr_mf_pull_apply_ordered": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _i64, _i64, _f,
                             _d, _d, _d, _d, _d, _d, _d, _int, _int, _p, _i64, _p, _p, _p, _int, _i64, _i64, _p, _p],
    "yr_bpr_mf_pull_item_buckets": [_i64, _int],
    "yr_loss_finalize": [_p, _f, _p, _p, _p],
    "yr_mf_scores_gemm": [_p, _p, _p, _i64, _int, _i64, _i64, _p, _i64, _p, _p],
    "yr_mf_eval_topk_planes_bytes": [_i64, _int],
    "yr_mf_eval_topk_workspace_bytes": [_i64, _i64, _int, _int, _int],
    "yr_mf_eval_topk": [_p, _p, _p, _i64, _int, _i64, _i64, _p, _p, _f, _int, _p, _p, _i64, _int, _p, _p, _p],
    "yr_mf_eval_topk_bias": [_p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _p, _f, _int, _p, _p, _i64, _int, _p, _p, _p],
    "yr_topk_masked": [_p, _i64, _i64, _i64, _p, _p, _p, _f, _int, _p, _p],
    "yr_rank_metrics_workspace_bytes": [_i64],
    "yr_rank_metrics": [_p, _i64, _int, _p, _p, _p, _p, _p, _p],
    "yr_bpr_loss_fwd": [_p, _p, _i64, _p, _p],
    "yr_bpr_loss_bwd": [_p, _p, _p, _f, _i64, _p, _p, _p],
    "yr_adam_dense": [_p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _d, _d, _int, _int, _p],
    "yr_adam_dense_multi": [_p, _p, _p, _p, _p, _int, _d, _d, _d, _d, _d, _d, _d, _int, _int, _p],
    "yr_adam_dense_flat": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _d, _d, _d, _d, _d, _d, _d, _int, _p],
    "yr_csr_rows_to_dense": [_p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _p],
    "yr_negative_mask": [_p, _i64, _i64, _int, C.c_uint64, _p, _p, _p],
    "yr_sgd_dense": [_p, _p, _i64, _d, _d, _int, _p],
    "yr_cdae_sparse_part_columns": [_i64],
    "yr_cdae_compact_rows": [_p, _i64, _i64, C.c_uint64, _d, _p, _p, _p, _p],
    "yr_cdae_sparse_encode": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _i64, _int, _p, _p, _p],
    "yr_cdae_sparse_dwh": [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, C.c_int32, _p, _p, _p],
    "yr_cdae_sparse_encode_t": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _i64, _int, _p, _p, _p],
    "yr_cdae_sparse_dwh_t": [_p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p],
    "yr_cdae_hidden_bwd_dwh_t": [_p, _p, _p, _p, _p, _int, _int, _p, _p, _i64, _i64, _int, _i64, _p, _p, _p, _p, _p, _p,
                                 _i64, _p, _p, _p],
    "yr_triplet_sample": [_p, _p, _i64, _p, _p, _i64, _i64, C.c_uint64, C.c_uint64, _int, _i64, _i64,
                          _p, _p, _p, _p, _p],
}

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises EngineError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  It must be the first HIP
    # runtime mapped into the process so that this library and torch share ONE runtime (and
    # one set of streams / device pointers); loading ours first leaves torch with no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"HIP engine library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C yelprecommendation_amd/csrc`. There is no CPU/PyTorch fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise EngineError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EngineError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = {"yr_engine_arch": C.c_char_p,
                      "yr_bpr_mf_pull_workspace_bytes": C.c_int64,
                      "yr_rank_metrics_workspace_bytes": C.c_int64,
                      "yr_cdae_sparse_part_columns": C.c_int64,
                      "yr_cdae_decode_loss_partials": C.c_int64,
                      "yr_mf_eval_topk_workspace_bytes": C.c_int64,
                      "yr_mf_eval_topk_planes_bytes": C.c_int64,
                      "yr_ngcf_step_workspace_bytes": C.c_int64}.get(name, C.c_int)
    v = lib.yr_engine_version()
    if v != ENGINE_VERSION:
        raise EngineError(f"engine ABI version {v} != expected {ENGINE_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str):
    if status == 0:
        return
    if status == -1:
        raise EngineError(f"{what}: unsupported configuration (embedding width must be 16/32/64/128)")
    if status == -2:
        raise EngineError(f"{what}: bad argument (null pointer, negative size or misaligned buffer)")
    raise EngineError(f"{what}: HIP error {status}")
