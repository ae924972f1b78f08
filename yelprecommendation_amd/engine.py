"""Thin host wrappers: torch tensors in, C-ABI calls out (include/yelprec_engine.h).

PyTorch is plumbing here — it owns device memory and the stream; every arithmetic
op of the hot path runs in the hand-written HIP kernels of ``csrc/``.  All tensors
must live on a ROCm device (``tensor.is_cuda``); there is no CPU or eager fallback.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import EngineError, check

LOSS_PARTIALS = 2048        # YR_LOSS_PARTIALS
FLAG_BAD_USER, FLAG_BAD_ITEM = 1, 2
OPT_ADAM, OPT_ADAMW = 0, 1
PULL_USER_PHASE, PULL_ITEM_PHASE = 1, 2
SUPPORTED_WIDTHS = (16, 32, 64, 128)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device (follows torch.cuda.stream()).
    The raw accessor costs 0.3 us per call against 2.7 us for current_stream().cuda_stream — it is on
    every launch."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise EngineError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise EngineError(f"{name}: tensor is on {t.device}; the engine runs on MI355X only (no CPU fallback)")
    if t.dtype != dtype:
        raise EngineError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise EngineError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _rows(t: torch.Tensor, dtype, name: str) -> int:
    """A 2-D row-major operand whose rows may be views into a wider buffer (stride(1) == 1)."""
    if isinstance(t, torch.Tensor) and t.dim() == 2 and t.is_cuda and t.dtype == dtype and t.stride(1) == 1 \
            and t.stride(0) >= t.shape[1]:
        return t.data_ptr()
    return _dev(t, dtype, name)


def _opt(t, dtype, name):
    return None if t is None else _dev(t, dtype, name)


def _table_dims(U: torch.Tensor, I: torch.Tensor):
    if U.dim() != 2 or I.dim() != 2 or U.shape[1] != I.shape[1]:
        raise EngineError(f"tables must be [rows, D] with equal D, got {tuple(U.shape)} / {tuple(I.shape)}")
    return U.shape[0], I.shape[0], U.shape[1]


def new_error_flag(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_on_flag(flag: torch.Tensor, what: str = "index"):
    """Host-side check of an err_flag word (one device sync)."""
    v = int(flag.item())
    if v:
        kinds = [k for b, k in ((FLAG_BAD_USER, "user"), (FLAG_BAD_ITEM, "item")) if v & b]
        flag.zero_()
        raise IndexError(f"{what}: out-of-range {' and '.join(kinds)} id(s) in a batch")


def mf_score(U, I, user, item, out=None, err_flag=None):
    """out[b] = U[user[b]] . I[item[b]]    (reference models/mf.py:20-23)."""
    lib = _lib.load()
    nu, ni, d = _table_dims(U, I)
    B = user.numel()
    if item.numel() != B:
        raise EngineError("user and item index tensors differ in length")
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=U.device)
    check(lib.yr_mf_score(_dev(U, torch.float32, "U"), _dev(I, torch.float32, "I"),
                          _dev(user, torch.int64, "user"), _dev(item, torch.int64, "item"),
                          B, d, nu, ni, _dev(out, torch.float32, "out"),
                          _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_mf_score")
    return out


def mf_score_backward(U, I, user, item, gout, gradU, gradI, err_flag=None):
    """gradU[user[b]] += gout[b] I[item[b]]; gradI[item[b]] += gout[b] U[user[b]]."""
    lib = _lib.load()
    nu, ni, d = _table_dims(U, I)
    B = user.numel()
    check(lib.yr_mf_score_backward(_dev(U, torch.float32, "U"), _dev(I, torch.float32, "I"),
                                   _dev(user, torch.int64, "user"), _dev(item, torch.int64, "item"),
                                   _dev(gout, torch.float32, "gout"), B, d, nu, ni,
                                   _dev(gradU, torch.float32, "gradU"), _dev(gradI, torch.float32, "gradI"),
                                   _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_mf_score_backward")


def bpr_mf_fwd_bwd(U, I, user, pos, neg, gradU, gradI, loss_partials, inv_batch=None, err_flag=None):
    """Fused BPR forward + loss (+ backward into dense gradU/gradI unless both None).

    reference trainers/mf_trainer.py:106-111.  ``loss_partials`` (float32[LOSS_PARTIALS])
    receives unscaled partial sums of softplus(-(s+ - s-)); see :func:`loss_finalize`.
    """
    lib = _lib.load()
    nu, ni, d = _table_dims(U, I)
    B = user.numel()
    if pos.numel() != B or neg.numel() != B:
        raise EngineError("user/pos/neg index tensors differ in length")
    if loss_partials.numel() != LOSS_PARTIALS:
        raise EngineError(f"loss_partials must hold {LOSS_PARTIALS} floats")
    if inv_batch is None:
        inv_batch = 1.0 / B if B else 0.0
    if gradU is not None and (gradU.shape != U.shape or gradI.shape != I.shape):
        raise EngineError("gradient buffers must match the table shapes")
    check(lib.yr_bpr_mf_fwd_bwd(_dev(U, torch.float32, "U"), _dev(I, torch.float32, "I"),
                                _dev(user, torch.int64, "user"), _dev(pos, torch.int64, "pos"),
                                _dev(neg, torch.int64, "neg"), B, d, nu, ni, float(inv_batch),
                                _opt(gradU, torch.float32, "gradU"), _opt(gradI, torch.float32, "gradI"),
                                _dev(loss_partials, torch.float32, "loss_partials"),
                                _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_bpr_mf_fwd_bwd")


PULL_MAX_BUCKETS = 16383      # csrc/bpr_pull.hip kMaxBuckets
PULL_MAX_USERS = (1 << 26) - 1


def pull_bucket_rows(d):
    """Rows per owner bucket of the pull step (1024 / D): the granularity of item-row chunks."""
    return 1024 // int(d)


def pull_supported(num_users, num_items, d):
    """Table shapes the pull step covers (include/yelprec_engine.h, limits of yr_bpr_mf_pull_step)."""
    if d not in SUPPORTED_WIDTHS:
        return False
    r = pull_bucket_rows(d)
    ru = 4 if (-(-num_users // r) < 768 and r > 4) else r       # few user rows: buckets of 4 rows (csrc/bpr_pull.hip)
    return (-(-num_users // ru) <= PULL_MAX_BUCKETS and -(-num_items // r) <= PULL_MAX_BUCKETS
            and num_users <= PULL_MAX_USERS)


def bpr_mf_pull_workspace(max_batch, num_users, num_items, d, device):
    """Scratch buffer for :func:`bpr_mf_pull_step` (uint8 tensor, 256-byte aligned by torch)."""
    lib = _lib.load()
    n = lib.yr_bpr_mf_pull_workspace_bytes(int(max_batch), int(num_users), int(num_items), int(d))
    if n < 0:
        check(int(n), "yr_bpr_mf_pull_workspace_bytes")
    return torch.empty(int(n), dtype=torch.uint8, device=device)


def bpr_mf_pull_step(U_old, U_new, I, mU, vU, mI, vI, user, pos, neg, step, lr, loss_partials, workspace,
                     beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False, inv_batch=None,
                     gradI_out=None, err_flag=None, loss_out=None, loss_accum=None, deterministic=False):
    """One whole BPR-MF step (forward, loss, both gradients, dense Adam) without float atomics.

    reference trainers/mf_trainer.py:106-112.  Reads ``U_old``, writes ``U_new`` (distinct
    buffers, ping-ponged by the caller), updates ``I`` and the Adam state in place; with
    ``gradI_out`` the item pass emits the dense item gradient instead of applying Adam.
    """
    lib = _lib.load()
    nu, ni, d = _table_dims(U_old, I)
    B = user.numel()
    if pos.numel() != B or neg.numel() != B:
        raise EngineError("user/pos/neg index tensors differ in length")
    if U_new.shape != U_old.shape or U_new.data_ptr() == U_old.data_ptr():
        raise EngineError("U_new must be a distinct buffer of U_old's shape")
    if loss_partials.numel() != LOSS_PARTIALS:
        raise EngineError(f"loss_partials must hold {LOSS_PARTIALS} floats")
    if inv_batch is None:
        inv_batch = 1.0 / B if B else 0.0
    step_size, bc2_sqrt = adam_scalars(step, lr, beta1, beta2)
    f32 = torch.float32
    check(lib.yr_bpr_mf_pull_step(
        _dev(U_old, f32, "U_old"), _dev(U_new, f32, "U_new"), _dev(I, f32, "I"),
        _dev(mU, f32, "mU"), _dev(vU, f32, "vU"), _opt(mI, f32, "mI"), _opt(vI, f32, "vI"),
        _opt(gradI_out, f32, "gradI_out"),
        _dev(user, torch.int64, "user"), _dev(pos, torch.int64, "pos"), _dev(neg, torch.int64, "neg"),
        B, d, nu, ni, float(inv_batch), float(lr), float(step_size), float(bc2_sqrt), float(beta1), float(beta2),
        float(eps), float(weight_decay), OPT_ADAMW if decoupled else OPT_ADAM, 1 if deterministic else 0,
        _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
        _dev(loss_partials, f32, "loss_partials"), _opt(loss_out, f32, "loss_out"),
        _opt(loss_accum, torch.float64, "loss_accum"), _opt(err_flag, torch.int32, "err_flag"), _stream()),
        "yr_bpr_mf_pull_step")


def triplet_sample(row_user, row_item, avoid_ptr, avoid_idx, num_users, num_items, seed, epoch, shuffle=True,
                   first=0, count=None, err_flag=None):
    """(user, pos, neg) for stream positions [first, first + count) of one epoch (csrc/triplets.hip;
    reference train.py:76-77 + data/datasets/mf_dataset.py:18-32)."""
    lib = _lib.load()
    n_rows = row_user.numel()
    count = n_rows - first if count is None else int(count)
    i64 = torch.int64
    out = [torch.empty(count, dtype=i64, device=row_user.device) for _ in range(3)]
    if avoid_ptr.numel() != num_users + 1:
        raise EngineError("avoid_ptr must hold num_users + 1 offsets")
    check(lib.yr_triplet_sample(_dev(row_user, i64, "row_user"), _dev(row_item, i64, "row_item"), n_rows,
                                _dev(avoid_ptr, i64, "avoid_ptr"),
                                _dev(avoid_idx, i64, "avoid_idx") if avoid_idx.numel() else _dev(avoid_ptr, i64, "avoid_ptr"),
                                int(num_users), int(num_items), int(seed) & (2**64 - 1), int(epoch) & (2**64 - 1),
                                1 if shuffle else 0, int(first), count,
                                *(_dev(t, i64, "out") if count else None for t in out),
                                _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_triplet_sample")
    return tuple(out)


# "rows": one wave per output row (default: the faster form on MI355X at Yelp2018 size, 76 us per launch);
# "sliced": feature slices pinned per XCD — half the fabric traffic (425 -> 217 MB, L2 hits 52 -> 76 %) but
# 80-89 us: the product is bound by the three dependent loads per short row, not by bytes
SPMM_FORM = "rows"


def spmm_csr(graph, X, out=None, accumulate=False, form=None):
    """Y = L X (or Y += L X) with L a :class:`yelprecommendation_amd.graph.LaplacianCSR`
    (reference models/ngcf.py:64,67: torch.sparse.mm(L, E))."""
    lib = _lib.load()
    n, d = X.shape
    if n != graph.n:
        raise EngineError(f"X has {n} rows, the graph {graph.n}")
    if out is None:
        if accumulate:
            raise EngineError("accumulate needs an output buffer")
        out = torch.empty_like(X)
    form = form or SPMM_FORM
    if form not in ("rows", "sliced"):
        raise EngineError("form must be 'rows' or 'sliced'")
    if form == "sliced":
        check(lib.yr_spmm_csr_sliced(_dev(graph.rowptr, torch.int32, "rowptr"), _dev(graph.col, torch.int32, "col"),
                                     _dev(graph.val, torch.float32, "val"), _dev(X, torch.float32, "X"),
                                     _dev(out, torch.float32, "Y"), n, d, 1 if accumulate else 0,
                                     _opt(graph.row_order, torch.int32, "row_order"), _stream()), "yr_spmm_csr_sliced")
        return out
    check(lib.yr_spmm_csr(_dev(graph.rowptr, torch.int32, "rowptr"), _dev(graph.col, torch.int32, "col"),
                          _dev(graph.val, torch.float32, "val"), _dev(X, torch.float32, "X"),
                          _dev(out, torch.float32, "Y"), n, d, 1 if accumulate else 0,
                          _opt(graph.heavy_rows, torch.int32, "heavy_rows"), graph.n_heavy, graph.heavy_threshold,
                          _stream()), "yr_spmm_csr")
    return out


NGCF_MAX_LAYERS = 8


def _ptr_array(tensors, what):
    import ctypes
    if not 0 < len(tensors) <= NGCF_MAX_LAYERS:
        raise EngineError(f"{what}: between 1 and {NGCF_MAX_LAYERS} layer buffers")
    return (ctypes.c_void_p * len(tensors))(*[_dev(t, torch.float32, what) for t in tensors])


class NGCFRowSet:
    """A set of graph nodes for the batch-aware propagation: ``flags`` int32[N] (1 = member), ``rows`` int32[N]
    of which the first ``count[0]`` entries list the members in no particular order (the count stays on the device),
    ``max_rows`` the host's upper bound of the count (sizes grids only); ``push``: few rows — the backward product
    scatters from them (spmm_csr_push_rows)."""
    __slots__ = ("flags", "rows", "count", "max_rows", "push")

    def __init__(self, n, device, max_rows, push=False):
        self.flags = torch.empty(n, dtype=torch.int32, device=device)
        self.rows = torch.empty(n, dtype=torch.int32, device=device)
        self.count = torch.empty(1, dtype=torch.int32, device=device)
        self.max_rows, self.push = int(min(max_rows, n)), bool(push)


def ngcf_frontier_mark(num_users, num_items, user_id, pos_ids, neg_ids=None, max_rows=None, push=False):
    """The rows a batch's scores read — user_id, num_users + pos_ids, num_users + neg_ids (reference
    models/ngcf.py:31-32,37-39) — as an NGCFRowSet."""
    lib = _lib.load()
    i64 = torch.int64
    B = user_id.numel()
    out = NGCFRowSet(num_users + num_items, user_id.device, (2 if neg_ids is None else 3) * B if max_rows is None else max_rows, push)
    check(lib.yr_ngcf_frontier_mark(_dev(user_id, i64, "user_id") if B else None, _dev(pos_ids, i64, "pos_ids") if B else None,
                                    _dev(neg_ids, i64, "neg_ids") if (neg_ids is not None and B) else None, B,
                                    num_users, num_items, out.flags.data_ptr(), out.rows.data_ptr(), out.count.data_ptr(),
                                    1, _stream()), "yr_ngcf_frontier_mark")
    return out


def ngcf_frontier_expand(graph, rows, max_rows=None, push=False):
    """The given NGCFRowSet plus all neighbours of its rows in the graph, as a new NGCFRowSet."""
    lib = _lib.load()
    out = NGCFRowSet(graph.n, rows.flags.device, graph.n if max_rows is None else max_rows, push)
    check(lib.yr_ngcf_frontier_expand(_dev(graph.rowptr, torch.int32, "rowptr"), _dev(graph.col, torch.int32, "col"),
                                      graph.n, rows.rows.data_ptr(), rows.count.data_ptr(), rows.max_rows,
                                      out.flags.data_ptr(), out.rows.data_ptr(), out.count.data_ptr(), 1, _stream()),
          "yr_ngcf_frontier_expand")
    return out


def spmm_csr_subset(graph, X, out, row_active=None, accumulate=False, rows=None):
    """:func:`spmm_csr` on the rows flagged in ``row_active`` (int32 flags; None: all); other rows of ``out`` are left
    as they are.  ``rows`` (NGCFRowSet, instead of ``row_active``): the same with the set's list driving the launch
    (cheap for small sets)."""
    if rows is not None:
        row_active = rows.flags
    lib = _lib.load()
    n, d = X.shape
    if n != graph.n or out.shape != X.shape:
        raise EngineError(f"X / out must be [{graph.n}, D]")
    check(lib.yr_spmm_csr_subset(_dev(graph.rowptr, torch.int32, "rowptr"), _dev(graph.col, torch.int32, "col"),
                                 _dev(graph.val, torch.float32, "val"), _dev(X, torch.float32, "X"),
                                 _dev(out, torch.float32, "Y"), n, d, 1 if accumulate else 0,
                                 _opt(graph.heavy_rows, torch.int32, "heavy_rows"), graph.n_heavy, graph.heavy_threshold,
                                 _opt(row_active, torch.int32, "row_active"),
                                 None if rows is None else rows.rows.data_ptr(),
                                 None if rows is None else rows.count.data_ptr(), 0 if rows is None else rows.max_rows,
                                 _stream()),
          "yr_spmm_csr_subset")
    return out


def spmm_csr_push_rows(graph, X, out, rows):
    """out[j] += sum over the rows r of ``rows`` (NGCFRowSet) of L[r, j] * X[r]: for the symmetric L the product
    out += L X restricted to those columns, as a scatter from the listed rows (yr_spmm_csr_push_rows)."""
    lib = _lib.load()
    n, d = X.shape
    if n != graph.n or out.shape != X.shape:
        raise EngineError(f"X / out must be [{graph.n}, D]")
    check(lib.yr_spmm_csr_push_rows(_dev(graph.rowptr, torch.int32, "rowptr"), _dev(graph.col, torch.int32, "col"),
                                    _dev(graph.val, torch.float32, "val"), _dev(X, torch.float32, "X"),
                                    _dev(out, torch.float32, "Y"), n, d, rows.rows.data_ptr(), rows.count.data_ptr(),
                                    rows.max_rows, _stream()), "yr_spmm_csr_push_rows")
    return out


def ngcf_score(layers, num_users, user_id, pos_ids, neg_ids=None, err_flag=None):
    """Scores of the concatenated layer embeddings (reference models/ngcf.py:44-58): returns
    ``pos`` or ``(pos, neg)``, each the sum over layers of per-layer dot products."""
    lib = _lib.load()
    n, d = layers[0].shape
    B = user_id.numel()
    i64 = torch.int64
    pos = torch.empty(B, dtype=torch.float32, device=layers[0].device)
    neg = torch.empty_like(pos) if neg_ids is not None else None
    check(lib.yr_ngcf_score_fwd(_ptr_array(layers, "layers"), len(layers), _dev(user_id, i64, "user_id"),
                                _dev(pos_ids, i64, "pos_ids"), _dev(neg_ids, i64, "neg_ids") if neg_ids is not None else None,
                                B, d, num_users, n - num_users, pos.data_ptr(),
                                neg.data_ptr() if neg is not None else None,
                                err_flag.data_ptr() if err_flag is not None else None, _stream()), "yr_ngcf_score_fwd")
    return pos if neg is None else (pos, neg)


def ngcf_score_backward(layers, dlayers, num_users, user_id, pos_ids, neg_ids, gpos, gneg, err_flag=None):
    """dlayers[k] += gradient of :func:`ngcf_score` w.r.t. layers[k] (float atomics)."""
    lib = _lib.load()
    n, d = layers[0].shape
    i64, f32 = torch.int64, torch.float32
    if len(layers) != len(dlayers):
        raise EngineError("layers / dlayers length mismatch")
    check(lib.yr_ngcf_score_bwd(_ptr_array(layers, "layers"), _ptr_array(dlayers, "dlayers"), len(layers),
                                _dev(user_id, i64, "user_id"), _dev(pos_ids, i64, "pos_ids"),
                                _dev(neg_ids, i64, "neg_ids") if neg_ids is not None else None,
                                _dev(gpos, f32, "gpos"), _dev(gneg, f32, "gneg") if neg_ids is not None else None,
                                user_id.numel(), d, num_users, n - num_users,
                                err_flag.data_ptr() if err_flag is not None else None, _stream()), "yr_ngcf_score_bwd")


def ngcf_dense_fwd(E, Z, W1, W2, out=None, rows=None):
    """leaky_relu((Z + E) W1^T + (E * Z) W2^T)   (reference models/ngcf.py:64-72).  ``rows`` (NGCFRowSet): only
    those rows are computed (and written)."""
    lib = _lib.load()
    n, d = E.shape
    if out is None:
        out = torch.empty_like(E)
    f32 = torch.float32
    if rows is not None:
        check(lib.yr_ngcf_dense_fwd_rows(_dev(E, f32, "E"), _dev(Z, f32, "Z"), _dev(W1, f32, "W1"), _dev(W2, f32, "W2"),
                                         n, d, _dev(out, f32, "Eout"), rows.rows.data_ptr(), rows.count.data_ptr(),
                                         rows.max_rows, None, _stream()), "yr_ngcf_dense_fwd_rows")
        return out
    check(lib.yr_ngcf_dense_fwd(_dev(E, f32, "E"), _dev(Z, f32, "Z"), _dev(W1, f32, "W1"), _dev(W2, f32, "W2"),
                                n, d, _dev(out, f32, "Eout"), _stream()), "yr_ngcf_dense_fwd")
    return out


def ngcf_dense_bwd(dEout, Eout, E, Z, W1, W2, dE, dW1, dW2, dZ=None, W1T=None, W2T=None, rows=None):
    """Backward of :func:`ngcf_dense_fwd`: dE += ..., dW1 += ..., dW2 += ..., returns dZ.
    ``W1T / W2T``: the transposed weights if the caller already has them (one batched transpose
    for all layers instead of two small launches per layer).  ``rows`` (NGCFRowSet): the rows outside it have
    dEout = 0 and are skipped (their dZ is not written)."""
    lib = _lib.load()
    n, d = E.shape
    f32 = torch.float32
    if dZ is None:
        dZ = torch.empty_like(E)
    if W1T is None:
        W1T, W2T = W1.t().contiguous(), W2.t().contiguous()  # [in, out] layout for the data-gradient GEMM
    if rows is not None:
        r, c, m = rows.rows.data_ptr(), rows.count.data_ptr(), rows.max_rows
        check(lib.yr_ngcf_dense_bwd_weight_rows(_dev(dEout, f32, "dEout"), _dev(Eout, f32, "Eout"), _dev(E, f32, "E"),
                                                _dev(Z, f32, "Z"), n, d, _dev(dW1, f32, "dW1"), _dev(dW2, f32, "dW2"),
                                                r, c, m, _stream()), "yr_ngcf_dense_bwd_weight_rows")
        check(lib.yr_ngcf_dense_bwd_data_rows(_dev(dEout, f32, "dEout"), _dev(Eout, f32, "Eout"), _dev(E, f32, "E"),
                                              _dev(Z, f32, "Z"), _dev(W1T, f32, "W1T"), _dev(W2T, f32, "W2T"), n, d,
                                              _dev(dZ, f32, "dZ"), _dev(dE, f32, "dE"), r, c, m, _stream()),
              "yr_ngcf_dense_bwd_data_rows")
        return dZ
    check(lib.yr_ngcf_dense_bwd_weight(_dev(dEout, f32, "dEout"), _dev(Eout, f32, "Eout"), _dev(E, f32, "E"),
                                       _dev(Z, f32, "Z"), n, d, _dev(dW1, f32, "dW1"), _dev(dW2, f32, "dW2"),
                                       _stream()), "yr_ngcf_dense_bwd_weight")
    check(lib.yr_ngcf_dense_bwd_data(_dev(dEout, f32, "dEout"), _dev(Eout, f32, "Eout"), _dev(E, f32, "E"),
                                     _dev(Z, f32, "Z"), _dev(W1T, f32, "W1T"), _dev(W2T, f32, "W2T"), n, d,
                                     _dev(dZ, f32, "dZ"), _dev(dE, f32, "dE"), _stream()), "yr_ngcf_dense_bwd_data")
    return dZ


ACT_IDENTITY, ACT_SIGMOID = 0, 1
COUNT_WORDS, COUNT_SLOTS = 2048, 64        # include/yelprec_engine.h YR_COUNT_WORDS / YR_COUNT_SLOTS


def gemm_f32(A, B, transA=False, transB=False, out=None, bias=None, act=ACT_IDENTITY, accumulate=False,
             split_k=1, alpha_count=None, rowsum=None):
    """out (+)= op(A) @ op(B) on the f32 matrix cores.  A, B 2-D row-major f32 GPU tensors;
    ``transA``: use A^T, ``transB``: use B^T (nn.Linear's ``x @ W^T`` is ``transB=True``).
    ``alpha_count`` (spread int32 count, COUNT_WORDS words): the product is scaled by 1 / count; ``rowsum`` ([M] f32): receives
    alpha * the row sums of op(A) (yr_gemm_f32_ex)."""
    lib = _lib.load()
    f32 = torch.float32
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if transB else (B.shape[0], B.shape[1])
    if K != K2:
        raise EngineError(f"inner dimensions differ: {K} vs {K2}")
    if out is None:
        if accumulate:
            raise EngineError("accumulate needs an output buffer")
        out = (torch.zeros if split_k > 1 else torch.empty)((M, N), dtype=f32, device=A.device)
    if out.shape != (M, N) or out.stride(1) != 1:
        raise EngineError("bad output buffer")
    if alpha_count is not None and alpha_count.numel() != COUNT_WORDS:
        raise EngineError(f"alpha_count: a spread count of {COUNT_WORDS} int32 words")
    if alpha_count is not None or rowsum is not None:
        if rowsum is not None and rowsum.numel() != M:
            raise EngineError("rowsum needs one element per row of op(A)")
        check(lib.yr_gemm_f32_ex(1 if transA else 0, 1 if transB else 0, M, N, K, _rows(A, f32, "A"), A.stride(0),
                                 _rows(B, f32, "B"), B.stride(0), out.data_ptr(), out.stride(0),
                                 _opt(bias, f32, "bias"), int(act), 1 if accumulate else 0, int(split_k),
                                 _opt(alpha_count, torch.int32, "alpha_count"), _opt(rowsum, f32, "rowsum"),
                                 _stream()), "yr_gemm_f32_ex")
        return out
    check(lib.yr_gemm_f32(1 if transA else 0, 1 if transB else 0, M, N, K, _rows(A, f32, "A"), A.stride(0),
                          _rows(B, f32, "B"), B.stride(0), out.data_ptr(), out.stride(0),
                          _opt(bias, f32, "bias"), int(act), 1 if accumulate else 0, int(split_k), _stream()),
          "yr_gemm_f32")
    return out


def spread_count(value, device):
    """A spread count holding ``value`` (tests; the kernels fill it with atomics)."""
    c = torch.zeros(COUNT_WORDS, dtype=torch.int32, device=device)
    c[0] = int(value)
    return c


def cdae_decode_loss_partials(B, I):
    return int(_lib.load().yr_cdae_decode_loss_partials(int(B), int(I)))


def cdae_decode_loss(z, Wo, bo, target, negative_mask, act, G, partial_loss, count, pred=None):
    """Decoder of the CDAE training step with the NS-BCE / BCE loss in its epilogue (yr_cdae_decode_loss):
    G = d loss / d pre-activation without the 1 / count factor, per-workgroup loss partials, count of the
    selected positions (``count`` must be zero on entry)."""
    lib = _lib.load()
    f32 = torch.float32
    B, H = z.shape
    I = Wo.shape[0]
    if partial_loss.numel() < cdae_decode_loss_partials(B, I) or G.shape != (B, I) or target.shape != (B, I):
        raise EngineError("bad buffers for cdae_decode_loss")
    if count.numel() != COUNT_WORDS:
        raise EngineError(f"count: a spread count of {COUNT_WORDS} int32 words")
    if pred is not None and (pred.shape != G.shape or pred.stride() != G.stride()):
        raise EngineError("pred must have the layout of G")
    check(lib.yr_cdae_decode_loss(_dev(z, f32, "z"), _dev(Wo, f32, "Wo"), _opt(bo, f32, "bo"),
                                  _dev(target, f32, "target"), _opt(negative_mask, f32, "negative_mask"), B, I, H,
                                  int(act), _rows(G, f32, "G"), G.stride(0),
                                  None if pred is None else _rows(pred, f32, "pred"),
                                  _dev(partial_loss, f32, "partial_loss"), _dev(count, torch.int32, "count"),
                                  _stream()), "yr_cdae_decode_loss")
    return G


class TrainLists:
    """A CDAE training batch as lists (yr_cdae_train_lists): ``rows`` — the encoder's input, a SparseRows of
    dropout_p(train items) — and ``loss`` — (columns, targets, counts) of the NS-BCE positions (positives +
    sampled negatives).  The buffers are sized for the worst case once per (device, B, I) and reused by the
    next batch made from the same ``pool`` — ONE batch of a pool is alive at a time: the next one overwrites the
    storage, and a consumer handed a batch that is no longer the pool's latest gets an EngineError from
    :meth:`alive` (CDAEStep.step_lists, CDAETrainer._scored_by_lists and loss_dense call it) instead of silently
    reading another batch's lists.  Every CDAEBatchLoader owns its pool, so the train / valid / test loaders do
    not alias each other; without ``pool`` a class-level one is used."""
    _pool = {}

    def __init__(self, ptr, idx, users, num_users, num_items, neg_times, neg_seed, drop_seed, p, err_flag=None,
                 extra=None, pool=None, batch_rows=None):
        """``extra``: (ptr, idx) of a second per-user CSR whose items are positives of the loss list too (the
        held-out items in validation) without entering the encoder list.
        ``batch_rows``: ``users`` holds SEVERAL batches of that many rows (the last may be short) and ``neg_seed`` /
        ``drop_seed`` are int64 device tensors with one seed per batch — one launch makes exactly the lists one
        TrainLists per batch would (yr_cdae_train_lists_batched)."""
        lib = _lib.load()
        B, I = users.numel(), int(num_items)
        dev = users.device
        key = (dev, B, I)
        pool = TrainLists._pool if pool is None else pool
        slot = pool.get(key)
        if slot is None:
            n, parts = B * SPARSE_PARTS * sparse_part_columns(I), B * SPARSE_PARTS
            mk = lambda m, dt: torch.empty(m, dtype=dt, device=dev)
            slot = [(mk(n, torch.int32), mk(n, torch.float32), mk(parts, torch.int32),
                     mk(n, torch.int32), mk(n, torch.float32), mk(parts, torch.int32)), 0]
            pool.clear()                                # a batch of the old shape keeps its own references
            pool[key] = slot
        slot[1] += 1
        self._slot, self._generation = slot, slot[1]
        buf = slot[0]
        i64 = torch.int64
        some = lambda t: t if t.numel() else torch.zeros(1, dtype=i64, device=dev)   # an empty index: never read, needs an address
        if batch_rows is not None:
            nb = -(-B // int(batch_rows)) if B else 0
            if not (torch.is_tensor(neg_seed) and torch.is_tensor(drop_seed) and neg_seed.numel() >= nb
                    and drop_seed.numel() >= nb):
                raise EngineError("batched TrainLists: one neg / drop seed per batch (int64 device tensors)")
            check(lib.yr_cdae_train_lists_batched(
                _dev(ptr, i64, "ptr"), _dev(some(idx), i64, "idx"), None if extra is None else _dev(extra[0], i64, "ptr2"),
                None if extra is None else _dev(some(extra[1]), i64, "idx2"), _dev(users, i64, "users"), B, int(num_users),
                I, int(neg_times), _dev(neg_seed, i64, "neg_seeds"), _dev(drop_seed, i64, "drop_seeds"), int(batch_rows),
                float(p), *(t.data_ptr() for t in buf), _opt(err_flag, torch.int32, "err_flag"), _stream()),
                "yr_cdae_train_lists_batched")
        else:
            check(lib.yr_cdae_train_lists(_dev(ptr, i64, "ptr"), _dev(some(idx), i64, "idx"),
                                          None if extra is None else _dev(extra[0], i64, "ptr2"),
                                          None if extra is None else _dev(some(extra[1]), i64, "idx2"), _dev(users, i64, "users"), B,
                                          int(num_users), I, int(neg_times), int(neg_seed) & (2**64 - 1),
                                          int(drop_seed) & (2**64 - 1), float(p), *(t.data_ptr() for t in buf),
                                          _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_cdae_train_lists")
        self.rows = SparseRows.from_buffers(buf[0], buf[1], buf[2], B, I)
        self.loss = (buf[3], buf[4], buf[5])
        self.B, self.I = B, I

    def alive(self):
        """Raise when a later batch of the same pool has overwritten this batch's storage."""
        if self._slot[1] != self._generation:
            raise EngineError("TrainLists: this batch's buffers were reused by a later batch of the same loader "
                              "(one batch of a loader is alive at a time; consume it before fetching the next)")
        return self

    def loss_dense(self):
        """(target, negative_mask) [B, I] the loss lists stand for (tests)."""
        self.alive()
        t = SparseRows.from_buffers(self.loss[0], self.loss[1] + 1.0, self.loss[2], self.B, self.I).to_dense()
        return (t == 2.0).float(), (t == 1.0).float()


def cdae_sampled_decode_splits(B):
    return int(_lib.load().yr_cdae_sampled_decode_splits(int(B)))


def cdae_sampled_decode(loss_lists, z, Wo, bo, act, dz, dWo, dbo, partial_loss, count):
    """Decoder forward + loss + all three decoder gradients on the loss positions only (yr_cdae_sampled_decode);
    everything it adds to comes out WITHOUT the 1 / count of the mean loss.  ``dz = dWo = dbo = None``: the loss
    partials and the count only (validation)."""
    lib = _lib.load()
    f32 = torch.float32
    lc, lv, ln = loss_lists
    B, H = z.shape
    I = Wo.shape[0]
    if partial_loss.numel() < B * cdae_sampled_decode_splits(B) or count.numel() != COUNT_WORDS:
        raise EngineError("bad buffers for cdae_sampled_decode")
    check(lib.yr_cdae_sampled_decode(_dev(lc, torch.int32, "loss_cols"), _dev(lv, f32, "loss_targets"),
                                     _dev(ln, torch.int32, "loss_count"), _dev(z, f32, "z"), _dev(Wo, f32, "Wo"),
                                     _opt(bo, f32, "bo"), B, I, H, int(act), _opt(dz, f32, "dz"), _opt(dWo, f32, "dWo"),
                                     _opt(dbo, f32, "dbo"), _dev(partial_loss, f32, "partial_loss"), _dev(count, torch.int32, "count"),
                                     _stream()), "yr_cdae_sampled_decode")


def cdae_loss_finalize_batched(partial_loss, splits, loss_count, rows, batch_rows, means, arrive, loss_accum):
    """loss_accum += sum over the batches of ``batch_rows`` rows of (the batch's loss partials / its positions):
    the validation loss of several batches scored by one cdae_sampled_decode launch (yr_cdae_loss_finalize_batched)."""
    lib = _lib.load()
    check(lib.yr_cdae_loss_finalize_batched(_dev(partial_loss, torch.float32, "partial_loss"), int(splits),
                                            _dev(loss_count, torch.int32, "loss_count"), int(rows), int(batch_rows),
                                            _dev(means, torch.float32, "means"), _dev(arrive, torch.int32, "arrive"),
                                            _opt(loss_accum, torch.float64, "loss_accum"), _stream()),
          "yr_cdae_loss_finalize_batched")


def cdae_loss_finalize(partial_loss, n_partials, count, stats, loss_accum=None):
    """stats[0] = sum(partials) / count, stats[1] = count; ``loss_accum`` (float64 device scalar) += stats[0]."""
    lib = _lib.load()
    check(lib.yr_cdae_loss_finalize(_dev(partial_loss, torch.float32, "partial_loss"), int(n_partials),
                                    _dev(count, torch.int32, "count"), _dev(stats, torch.float32, "stats"),
                                    _opt(loss_accum, torch.float64, "loss_accum"), _stream()), "yr_cdae_loss_finalize")


def cdae_hidden_bwd(dz, z, act, user, dV, touched_users, dbh, partial_loss=None, n_partials=0, count=None,
                    stats=None, loss_accum=None, scale_dz=False):
    """dz <- dz * act'(z) (``scale_dz``: dz / count first); dbh = column sums; dV[user] += dz (rows marked in
    ``touched_users``); with ``n_partials`` > 0 also the loss of the step into ``stats`` / ``loss_accum``."""
    lib = _lib.load()
    f32 = torch.float32
    B, H = dz.shape
    check(lib.yr_cdae_hidden_bwd(_dev(dz, f32, "dz"), _dev(z, f32, "z"), int(act), _dev(user, torch.int64, "user"),
                                 B, H, dV.shape[0], _dev(dV, f32, "dV"), _opt(touched_users, torch.uint8, "touched"),
                                 _dev(dbh, f32, "dbh"), _opt(partial_loss, f32, "partial_loss"), int(n_partials),
                                 _opt(count, torch.int32, "count"), _opt(stats, f32, "stats"),
                                 _opt(loss_accum, torch.float64, "loss_accum"), 1 if scale_dz else 0, _stream()),
          "yr_cdae_hidden_bwd")


def cdae_hidden_init(bias, V, user, err_flag=None):
    """zpre[b] = bias + V[user[b]]   (reference models/cdae.py:49)."""
    lib = _lib.load()
    B, H = user.numel(), V.shape[1]
    out = torch.empty((B, H), dtype=torch.float32, device=V.device)
    check(lib.yr_cdae_hidden_init(out.data_ptr(), _dev(bias, torch.float32, "bias"), _dev(V, torch.float32, "V"),
                                  _dev(user, torch.int64, "user"), B, H, V.shape[0],
                                  _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_cdae_hidden_init")
    return out


def dropout(x, rnd, p):
    lib = _lib.load()
    out = torch.empty_like(x)
    check(lib.yr_dropout(_dev(x, torch.float32, "x"), _dev(rnd, torch.float32, "rnd"), float(p), x.numel(),
                         out.data_ptr(), _stream()), "yr_dropout")
    return out


SPARSE_PARTS = 32       # csrc/cdae_sparse.hip kParts


def sparse_part_columns(I):
    """Columns per sub-list of a SparseRows row (list buffers hold B * SPARSE_PARTS * this many entries)."""
    return int(_lib.load().yr_cdae_sparse_part_columns(int(I)))


class SparseRows:
    """(column, value) lists of the non-zeros of dropout_p(x), 32 sub-lists per row (csrc/cdae_sparse.hip);
    the list buffers are sized for the worst case once per batch shape and reused by the next call."""
    _pool = {}

    def __init__(self, x, seed=0, p=0.0, count=None, negative_mask=None, loss_lists=None):
        """``negative_mask`` + ``loss_lists`` (cols int32, targets f32, count int32 buffers like this object's):
        the same pass also writes the loss positions of every row — target + negative_mask != 0 — as
        (column, target) lists (yr_cdae_compact_pair)."""
        lib = _lib.load()
        B, I = x.shape
        self.cpp = int(lib.yr_cdae_sparse_part_columns(I))
        key = (x.device, B, I)
        buf = SparseRows._pool.get(key)
        if buf is None:
            n = B * SPARSE_PARTS * self.cpp
            buf = (torch.empty(n, dtype=torch.int32, device=x.device), torch.empty(n, dtype=torch.float32, device=x.device))
            SparseRows._pool = {key: buf}                  # one batch shape at a time
        self.cols, self.vals = buf
        self.count = torch.empty(B * SPARSE_PARTS, dtype=torch.int32, device=x.device) if count is None else count
        self.B, self.I = B, I
        if negative_mask is not None:
            lc, lv, ln = loss_lists
            if lc.numel() < self.cols.numel() or lv.numel() < self.cols.numel() or ln.numel() < self.count.numel():
                raise EngineError("loss list buffers too small")
            check(lib.yr_cdae_compact_pair(_dev(x, torch.float32, "x"), _dev(negative_mask, torch.float32, "negative_mask"),
                                           B, I, int(seed) & (2**64 - 1), float(p), self.cols.data_ptr(),
                                           self.vals.data_ptr(), self.count.data_ptr(), _dev(lc, torch.int32, "loss_cols"),
                                           _dev(lv, torch.float32, "loss_targets"), _dev(ln, torch.int32, "loss_count"),
                                           _stream()), "yr_cdae_compact_pair")
            return
        check(lib.yr_cdae_compact_rows(_dev(x, torch.float32, "x"), B, I, int(seed) & (2**64 - 1), float(p),
                                       self.cols.data_ptr(), self.vals.data_ptr(), self.count.data_ptr(), _stream()),
              "yr_cdae_compact_rows")

    @classmethod
    def from_buffers(cls, cols, vals, count, B, I):
        """Lists that something else filled (yr_cdae_train_lists)."""
        self = cls.__new__(cls)
        self.cols, self.vals, self.count, self.B, self.I = cols, vals, count, int(B), int(I)
        self.cpp = sparse_part_columns(I)
        return self

    def to_dense(self):
        """The dense matrix the lists stand for (tests)."""
        B, P, cpp = self.B, SPARSE_PARTS, self.cpp
        out = torch.zeros(B, self.I, dtype=torch.float32, device=self.cols.device)
        m = torch.arange(cpp, device=out.device)[None, :] < self.count[:, None]          # [B*P, cpp]
        rows = torch.arange(B, device=out.device).repeat_interleave(P)[:, None].expand(-1, cpp)[m]
        out[rows, self.cols.view(B * P, cpp)[m].long()] = self.vals.view(B * P, cpp)[m]
        return out

    def row_columns(self, r):
        """Columns of row r in list order (tests)."""
        c = self.cols.view(self.B, SPARSE_PARTS, self.cpp)[r]
        n = self.count.view(self.B, SPARSE_PARTS)[r]
        return torch.cat([c[q, :int(n[q])] for q in range(SPARSE_PARTS)])


def cdae_sparse_encode(rows: "SparseRows", Wh, bh, V, user, act, err_flag=None, out=None, transposed=False):
    """z = act(Wh . rows + bh + V[user])   (reference models/cdae.py:49, sparse input).  ``transposed``: ``Wh`` is the
    [I, H] working copy (yr_cdae_sparse_encode_t)."""
    lib = _lib.load()
    H = Wh.shape[1] if transposed else Wh.shape[0]
    z = torch.empty(rows.B, H, dtype=torch.float32, device=Wh.device) if out is None else out
    fn = lib.yr_cdae_sparse_encode_t if transposed else lib.yr_cdae_sparse_encode
    check(fn(rows.cols.data_ptr(), rows.vals.data_ptr(), rows.count.data_ptr(),
                                    _dev(Wh, torch.float32, "Wh"), _dev(bh, torch.float32, "bh"),
                                    _dev(V, torch.float32, "V"), _dev(user, torch.int64, "user"), rows.B, rows.I, H,
                                    V.shape[0], int(act), z.data_ptr(), _opt(err_flag, torch.int32, "err_flag"),
                                    _stream()), "yr_cdae_sparse_encode")
    return z


def cdae_sparse_dwh_t(rows: "SparseRows", dz, dWhT, touched_items):
    """dWhT [I, H] (zero where unmarked) += rows^T . dz, rows marked in ``touched_items`` (yr_cdae_sparse_dwh_t)."""
    lib = _lib.load()
    check(lib.yr_cdae_sparse_dwh_t(rows.cols.data_ptr(), rows.vals.data_ptr(), rows.count.data_ptr(),
                                   _dev(dz, torch.float32, "dz"), rows.B, rows.I, dz.shape[1],
                                   _dev(dWhT, torch.float32, "dWhT"), _dev(touched_items, torch.uint8, "touched_items"),
                                   _stream()), "yr_cdae_sparse_dwh_t")


def cdae_hidden_bwd_dwh_t(rows: "SparseRows", dz, z, act, user, count, dV, touched_users, dbh, dWhT, touched_items,
                          partial_loss, n_partials, stats, loss_accum=None, scale_dz=False):
    """cdae_hidden_bwd + cdae_sparse_dwh_t in one launch (yr_cdae_hidden_bwd_dwh_t); ``dbh`` must be zero on entry."""
    lib = _lib.load()
    f32 = torch.float32
    check(lib.yr_cdae_hidden_bwd_dwh_t(rows.cols.data_ptr(), rows.vals.data_ptr(), rows.count.data_ptr(),
                                       _dev(dz, f32, "dz"), _dev(z, f32, "z"), int(act), 1 if scale_dz else 0,
                                       _dev(count, torch.int32, "count"), _dev(user, torch.int64, "user"), rows.B,
                                       rows.I, dz.shape[1], dV.shape[0], _dev(dV, f32, "dV"),
                                       _opt(touched_users, torch.uint8, "touched_users"), _dev(dbh, f32, "dbh"),
                                       _dev(dWhT, f32, "dWhT"), _dev(touched_items, torch.uint8, "touched_items"),
                                       _opt(partial_loss, f32, "partial_loss"), int(n_partials),
                                       _opt(stats, f32, "stats"), _opt(loss_accum, torch.float64, "loss_accum"),
                                       _stream()), "yr_cdae_hidden_bwd_dwh_t")


_dwh_scratch = {}


def cdae_sparse_dwh(rows: "SparseRows", dz, dWh):
    """dWh (zero on entry) = dz^T . rows (dWh [H, I]); the transposed scratch and the column-claim words live per (device, I, H)."""
    lib = _lib.load()
    H = dz.shape[1]
    key = (dz.device, rows.I, H)
    sc = _dwh_scratch.get(key)
    if sc is None:
        dev = dz.device
        sc = _dwh_scratch[key] = [torch.zeros(rows.I, H, dtype=torch.float32, device=dev),
                                  torch.zeros(rows.I, dtype=torch.int32, device=dev),
                                  torch.empty(rows.I, dtype=torch.int32, device=dev),
                                  torch.zeros(1, dtype=torch.int32, device=dev), 0]
    sc[4] = sc[4] % 0x7ffffff0 + 1                       # a fresh non-zero epoch per call
    check(lib.yr_cdae_sparse_dwh(rows.cols.data_ptr(), rows.vals.data_ptr(), rows.count.data_ptr(),
                                 _dev(dz, torch.float32, "dz"), rows.B, rows.I, H, _dev(dWh, torch.float32, "dWh"),
                                 sc[0].data_ptr(), sc[1].data_ptr(), sc[4], sc[2].data_ptr(), sc[3].data_ptr(),
                                 _stream()), "yr_cdae_sparse_dwh")
    return dWh


def sigmoid_(x):
    lib = _lib.load()
    check(lib.yr_sigmoid(_dev(x, torch.float32, "x"), x.numel(), _stream()), "yr_sigmoid")
    return x


def sigmoid_bwd(dy, y, out=None):
    """dy * y * (1 - y); ``out`` may be ``dy`` itself (in place) or omitted (new tensor)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(dy)
    check(lib.yr_sigmoid_bwd(_dev(dy, torch.float32, "dy"), _dev(y, torch.float32, "y"), _dev(out, torch.float32, "g"),
                             dy.numel(), _stream()), "yr_sigmoid_bwd")
    return out


def sigmoid_bwd_(g, y):
    return sigmoid_bwd(g, y, out=g)


def dropout_seeded(x, seed, p):
    """nn.Dropout(p) in training mode with the mask drawn inside the kernel from (seed, position)."""
    lib = _lib.load()
    out = torch.empty_like(x)
    check(lib.yr_dropout_seeded(_dev(x, torch.float32, "x"), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p), x.numel(),
                                out.data_ptr(), _stream()), "yr_dropout_seeded")
    return out


def colsum(X, out=None, accumulate=False):
    lib = _lib.load()
    rows, cols = X.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=X.device)
    check(lib.yr_colsum(_dev(X, torch.float32, "X"), rows, cols, _dev(out, torch.float32, "out"),
                        1 if accumulate else 0, _stream()), "yr_colsum")
    return out


def row_scatter_add(G, user, dV):
    lib = _lib.load()
    check(lib.yr_row_scatter_add(_dev(G, torch.float32, "G"), _dev(user, torch.int64, "user"), G.shape[0],
                                 G.shape[1], dV.shape[0], _dev(dV, torch.float32, "dV"), _stream()),
          "yr_row_scatter_add")


def nsbce_fwd(pred, target, negative_mask):
    """(stats, workspace): stats[0] = mean BCE over nonzero(target + negative_mask), stats[1] = count."""
    lib = _lib.load()
    f32 = torch.float32
    ws = torch.empty(2 * LOSS_PARTIALS, dtype=f32, device=pred.device)
    stats = torch.empty(2, dtype=f32, device=pred.device)
    check(lib.yr_nsbce_fwd(_dev(pred, f32, "pred"), _dev(target, f32, "target"), _opt(negative_mask, f32, "neg"),
                           pred.numel(), ws.data_ptr(), stats.data_ptr(), _stream()), "yr_nsbce_fwd")
    return stats


def nsbce_bwd(pred, target, negative_mask, stats, gout):
    lib = _lib.load()
    f32 = torch.float32
    dpred = torch.empty_like(pred)
    check(lib.yr_nsbce_bwd(_dev(pred, f32, "pred"), _dev(target, f32, "target"), _opt(negative_mask, f32, "neg"),
                           _dev(stats, f32, "stats"), _dev(gout, f32, "gout"), pred.numel(), dpred.data_ptr(),
                           _stream()), "yr_nsbce_bwd")
    return dpred


def loss_finalize(loss_partials, scale, loss_out=None, loss_accum=None):
    """loss_out[0] = scale * sum(partials); loss_accum[0] (float64) += the same."""
    lib = _lib.load()
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=loss_partials.device)
    check(lib.yr_loss_finalize(_dev(loss_partials, torch.float32, "loss_partials"), float(scale),
                               _dev(loss_out, torch.float32, "loss_out"),
                               _opt(loss_accum, torch.float64, "loss_accum"), _stream()), "yr_loss_finalize")
    return loss_out


MASK_VALUE = -3.40282e+38     # reference trainers/mf_trainer.py:167


def topk_masked(scores, mask_ptr, mask_idx, k, mask_value=MASK_VALUE, out=None, mask_rows=None):
    """Row-wise top-k of ``scores[R, N]`` with per-row masked columns (CSR) forced to
    ``mask_value`` first (reference trainers/mf_trainer.py:163-178).  -> int64 [R, k].
    ``mask_rows`` [R]: row r's mask list is CSR row ``mask_rows[r]`` (a batch of users pointing into one
    per-user CSR) instead of row r."""
    lib = _lib.load()
    if scores.dim() != 2 or scores.stride(1) != 1:
        raise EngineError("scores must be [rows, cols] with unit column stride")
    if not scores.is_cuda or scores.dtype != torch.float32:
        raise EngineError("scores must be a float32 GPU tensor (no CPU fallback)")
    R, N = scores.shape
    if out is None:
        out = torch.empty((R, k), dtype=torch.int64, device=scores.device)
    if mask_ptr is not None and mask_rows is None and mask_ptr.numel() != R + 1:
        raise EngineError("mask_ptr must have rows + 1 entries")
    if mask_idx is not None and mask_idx.numel() == 0:
        mask_ptr = mask_idx = mask_rows = None             # every list empty: an empty tensor has no address to pass
    if mask_rows is not None and mask_rows.numel() != R:
        raise EngineError("mask_rows must have one entry per row")
    check(lib.yr_topk_masked(scores.data_ptr(), R, N, scores.stride(0) if R > 1 else N,
                             _opt(mask_ptr, torch.int64, "mask_ptr"), _opt(mask_idx, torch.int64, "mask_idx"),
                             _opt(mask_rows, torch.int64, "mask_rows"),
                             float(mask_value), int(k), _dev(out, torch.int64, "out"), _stream()),
          "yr_topk_masked")
    return out


def mf_scores_gemm(U, I, users, out=None, err_flag=None):
    """scores[r, j] = U[users[r]] . I[j] for every item j — the reference's per-user
    ``model([u]*I, arange(I))`` (trainers/mf_trainer.py:138-140) as one f32 MFMA GEMM."""
    lib = _lib.load()
    nu, ni, d = _table_dims(U, I)
    n = users.numel()
    if out is None:
        out = torch.empty((n, ni), dtype=torch.float32, device=U.device)
    if out.dim() != 2 or out.shape[0] < n or out.shape[1] < ni or out.stride(1) != 1:
        raise EngineError("scores buffer must be [>= rows, >= num_items] with unit column stride")
    check(lib.yr_mf_scores_gemm(_dev(U, torch.float32, "U"), _dev(I, torch.float32, "I"),
                                _dev(users, torch.int64, "users"), n, d, nu, ni, out.data_ptr(), out.stride(0),
                                _opt(err_flag, torch.int32, "err_flag"), _stream()), "yr_mf_scores_gemm")
    return out[:n]


def sort_mask_rows(mask_ptr, mask_idx):
    """Mask lists with the item ids ascending inside each row (index bookkeeping, done once per
    eval set): what :func:`mf_eval_topk` requires."""
    n = mask_ptr.numel() - 1
    if mask_idx.numel() == 0:
        return mask_idx
    rows = torch.repeat_interleave(torch.arange(n, device=mask_idx.device), mask_ptr[1:] - mask_ptr[:-1])
    span = int(mask_idx.max().item()) + 1
    order = torch.argsort(rows * span + mask_idx)
    return mask_idx[order].contiguous()


def fused_eval_supports(k, d):
    """The fused evaluation kernel keeps top-k lists of 4 / 10 / 16 / 32 entries in registers (32: up to D = 64)."""
    return k <= 16 or (k <= 32 and d <= 64)


EVAL_MODES = {"f32": 0, "bf16x3": 1}          # YR_EVAL_F32 / YR_EVAL_BF16X3 (include/yelprec_engine.h)


def mf_eval_topk(U, I, users, mask_ptr, mask_idx_sorted, k, mask_value=MASK_VALUE, out=None, sliced=True,
                 item_bias=None, precision="bf16x3", prescan=None, hint=None, form=None):
    """Fused full-catalogue scoring + mask + top-k (no score matrix).  ``mask_idx_sorted``: CSR mask
    lists with ascending ids inside each row (see :func:`sort_mask_rows`).  ``sliced=False`` withholds
    the room for the partial lists, i.e. forces the one-slice form of the kernel (tests).  ``item_bias``: scores
    U[u] . I[j] + item_bias[j] (the CDAE decoder before its sigmoid).  ``precision``: "bf16x3" — f32 scores from
    three-term bfloat16 splits of both operands on the bf16 matrix instructions (six partial products, error below
    the f32 rounding of a product) — or "f32", the f32 matrix instruction itself.  ``prescan``: None = the
    library's rule (catalogues of 16,384 items and more), False / True = never / always run the sampling launch that
    gives the lists their starting thresholds (same result either way; ``sliced=False`` implies False).
    ``hint``: int64 [n, k] item ids, e.g. the result of the previous evaluation of the same rows (may be ``out``
    itself): the lists start from the smallest score among a row's k hint items, which the row's k-th best score
    cannot be below — the result does not depend on it, a good hint saves most candidate insertions and the
    prescan launch.  ``form``: None = the library's rule (from 2,048 rows the two-role sweep at D = 128, and at D = 64
    for k > 10 when a hint gives the thresholds; the four-wave sweep otherwise), "two_roles" / "four_waves" = that form where both exist (YR_EVAL_TWO_ROLES / YR_EVAL_FOUR_WAVES:
    same lists; tests and comparisons)."""
    lib = _lib.load()
    nu, ni, d = _table_dims(U, I)
    n = users.numel()
    if mask_idx_sorted is not None and mask_idx_sorted.numel() == 0:
        mask_ptr = mask_idx_sorted = None              # every list empty: an empty tensor has no address to pass
    mode = EVAL_MODES[precision]
    planes_only = lib.yr_mf_eval_topk_planes_bytes(ni, d) if mode else 0
    if prescan is False or not sliced:
        mode |= 2                                   # YR_EVAL_NO_PRESCAN
    elif prescan:
        mode |= 4                                   # YR_EVAL_FORCE_PRESCAN
    mode |= {None: 0, "two_roles": 8, "four_waves": 16}[form]   # YR_EVAL_TWO_ROLES / YR_EVAL_FOUR_WAVES
    if out is None:
        out = torch.empty((n, k), dtype=torch.int64, device=U.device)
    if hint is not None and tuple(hint.shape) != (n, k):
        raise EngineError(f"hint must be [{n}, {k}] item ids, got {tuple(hint.shape)}")
    flag = new_error_flag(U.device)
    if sliced:
        ws_bytes = lib.yr_mf_eval_topk_workspace_bytes(n, ni, d, int(k), mode)
    else:
        ws_bytes = planes_only + (-(-n * 4 // 256) * 256 if hint is not None else 0)
    if ws_bytes < 0:
        check(int(ws_bytes), "yr_mf_eval_topk_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=U.device) if ws_bytes else None
    check(lib.yr_mf_eval_topk_bias(_dev(U, torch.float32, "U"), _dev(I, torch.float32, "I"),
                              _opt(item_bias, torch.float32, "item_bias"),
                              _dev(users, torch.int64, "users"), n, d, nu, ni,
                              _opt(mask_ptr, torch.int64, "mask_ptr"), _opt(mask_idx_sorted, torch.int64, "mask_idx"),
                              float(mask_value), int(k), _dev(out, torch.int64, "out"),
                              ws.data_ptr() if ws is not None else None, ws_bytes, mode,
                              _opt(hint, torch.int64, "hint"), flag.data_ptr(), _stream()),
          "yr_mf_eval_topk")
    raise_on_flag(flag, "mf_eval_topk")
    return out


def mf_recommend(U, I, users, mask_ptr, mask_idx, k, chunk_users=4096, fused=None):
    """Top-k unmasked items for each user id in ``users`` (reference
    trainers/mf_trainer.py:134-144 + :163-178, batched), all on the device.
    ``fused`` (default when k <= 16, or k <= 32 with D <= 64): one kernel, scores on the matrix cores with mask + top-k in the
    epilogue.  Otherwise: score GEMM into a chunked buffer + the row-wise masked top-k kernel."""
    nu_tab, ni, d = _table_dims(U, I)
    n = users.numel()
    if fused is None:
        fused = fused_eval_supports(k, d)
    if fused:
        return mf_eval_topk(U, I, users.contiguous(), mask_ptr, sort_mask_rows(mask_ptr, mask_idx), k)
    out = torch.empty((n, k), dtype=torch.int64, device=U.device)
    flag = new_error_flag(U.device)
    chunk_users = max(1, min(chunk_users, n))
    scores = torch.empty((chunk_users, ni), dtype=torch.float32, device=U.device)
    ptr_host = mask_ptr.cpu()
    for lo in range(0, n, chunk_users):
        hi = min(lo + chunk_users, n)
        mf_scores_gemm(U, I, users[lo:hi].contiguous(), out=scores, err_flag=flag)
        ptr = (mask_ptr[lo:hi + 1] - mask_ptr[lo]).contiguous()
        topk_masked(scores[:hi - lo], ptr, mask_idx[int(ptr_host[lo]):], k, out=out[lo:hi])
    raise_on_flag(flag, "mf_recommend")
    return out


def csr_rows_to_dense(ptr, idx, users, num_items, out=None, accumulate=False, err_flag=None):
    """[len(users), num_items] float32 0/1 rows from a per-user item CSR (device-side counterpart of
    the dense masks of reference cdae_data_pipeline.py:33-37)."""
    lib = _lib.load()
    B = users.numel()
    if out is None:
        if accumulate:
            raise EngineError("accumulate needs an output buffer")
        out = torch.empty((B, num_items), dtype=torch.float32, device=users.device)
    i64 = torch.int64
    ii = idx if idx.numel() else torch.zeros(1, dtype=i64, device=users.device)
    check(lib.yr_csr_rows_to_dense(_dev(ptr, i64, "ptr"), _dev(ii, i64, "idx"), _dev(users, i64, "users"), B,
                                   ptr.numel() - 1, int(num_items), 1 if accumulate else 0,
                                   _dev(out, torch.float32, "out"),
                                   err_flag.data_ptr() if err_flag is not None else None, _stream()),
          "yr_csr_rows_to_dense")
    return out


def negative_mask(positives, neg_times, seed, err_flag=None):
    """Exactly ``neg_times * positives`` distinct non-positive items per row, uniformly at random
    (reference cdae_dataset.py:20-34); ``err_flag`` gets FLAG_BAD_ITEM if a row has too few
    non-positives (where np.random.choice raises)."""
    lib = _lib.load()
    B, I = positives.shape
    out = torch.empty_like(positives)
    check(lib.yr_negative_mask(_dev(positives, torch.float32, "positives"), B, I, int(neg_times),
                               int(seed) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(),
                               err_flag.data_ptr() if err_flag is not None else None, _stream()), "yr_negative_mask")
    return out


def rank_metrics(topk, pos_ptr, pos_idx, pos_rows=None):
    """(precision, recall, map, ndcg)@k of reference metric.py computed on the device from the top-k
    lists ``topk [n, k]`` and the held-out items (CSR, original order).  Returns a float64[10] device
    tensor: the four metrics, the number of users with a non-empty list, then the four un-normalised
    sums and n (for a cross-rank reduction); one host sync to read it."""
    lib = _lib.load()
    n, k = topk.shape
    if pos_rows is None and pos_ptr.numel() != n + 1:
        raise EngineError("pos_ptr must have rows + 1 entries")
    if pos_rows is not None and pos_rows.numel() != n:
        raise EngineError("pos_rows must have one entry per row")
    ws = torch.empty(lib.yr_rank_metrics_workspace_bytes(n) // 8, dtype=torch.float64, device=topk.device)
    out = torch.empty(10, dtype=torch.float64, device=topk.device)
    idx = pos_idx if pos_idx.numel() else torch.zeros(1, dtype=torch.int64, device=topk.device)
    check(lib.yr_rank_metrics(_dev(topk, torch.int64, "topk"), n, k, _dev(pos_ptr, torch.int64, "pos_ptr"),
                              _dev(idx, torch.int64, "pos_idx"), _opt(pos_rows, torch.int64, "pos_rows"),
                              ws.data_ptr(), out.data_ptr(), _stream()),
          "yr_rank_metrics")
    return out


def bpr_loss_fwd(pos, neg, loss_partials):
    """Partial sums of softplus(-(pos - neg))  (reference loss.py:25-27)."""
    lib = _lib.load()
    B = pos.numel()
    if neg.numel() != B:
        raise EngineError("pos and neg differ in length")
    check(lib.yr_bpr_loss_fwd(_dev(pos, torch.float32, "pos"), _dev(neg, torch.float32, "neg"), B,
                              _dev(loss_partials, torch.float32, "loss_partials"), _stream()), "yr_bpr_loss_fwd")


def bpr_loss_bwd(pos, neg, gout, gpos, gneg):
    """d mean(-logsigmoid(pos-neg)) / d pos, d neg, scaled by the scalar gout."""
    lib = _lib.load()
    B = pos.numel()
    check(lib.yr_bpr_loss_bwd(_dev(pos, torch.float32, "pos"), _dev(neg, torch.float32, "neg"),
                              _dev(gout, torch.float32, "gout"), 1.0 / B if B else 0.0, B,
                              _dev(gpos, torch.float32, "gpos"), _dev(gneg, torch.float32, "gneg"), _stream()),
          "yr_bpr_loss_bwd")


def adam_scalars(step: int, lr: float, beta1: float, beta2: float):
    """Host doubles of torch's Adam step: (lr / (1 - b1^t), sqrt(1 - b2^t))."""
    return lr / (1.0 - beta1 ** step), (1.0 - beta2 ** step) ** 0.5


def adam_dense(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
               decoupled=False, zero_grad=False):
    """Dense Adam/AdamW step in place (reference trainers/base_trainer.py:34-38)."""
    lib = _lib.load()
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n:
        raise EngineError("p/g/m/v sizes differ")
    step_size, bc2_sqrt = adam_scalars(step, lr, beta1, beta2)
    check(lib.yr_adam_dense(_dev(p, torch.float32, "p"), _dev(g, torch.float32, "g"),
                            _dev(m, torch.float32, "m"), _dev(v, torch.float32, "v"), n,
                            float(lr), float(step_size), float(bc2_sqrt), float(beta1), float(beta2),
                            float(eps), float(weight_decay), OPT_ADAMW if decoupled else OPT_ADAM,
                            1 if zero_grad else 0, _stream()), "yr_adam_dense")


ADAM_MULTI_MAX = 16


def adam_dense_multi(tensors, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                     decoupled=False, zero_grad=False):
    """The same Adam/AdamW step for several (p, g, m, v) tuples at the same step count in ONE launch
    per ADAM_MULTI_MAX tensors — for the small weight matrices and biases of a model."""
    import ctypes
    lib = _lib.load()
    step_size, bc2_sqrt = adam_scalars(step, lr, beta1, beta2)
    f32 = torch.float32
    for s0 in range(0, len(tensors), ADAM_MULTI_MAX):
        group = tensors[s0:s0 + ADAM_MULTI_MAX]
        cols = []
        for k, name in enumerate("pgmv"):
            cols.append((ctypes.c_void_p * len(group))(*[_dev(t[k], f32, name) for t in group]))
        for p, g, m, v in group:
            if not (p.numel() == g.numel() == m.numel() == v.numel()):
                raise EngineError("p/g/m/v sizes differ")
        counts = (ctypes.c_int64 * len(group))(*[t[0].numel() for t in group])
        check(lib.yr_adam_dense_multi(cols[0], cols[1], cols[2], cols[3], counts, len(group), float(lr),
                                      float(step_size), float(bc2_sqrt), float(beta1), float(beta2), float(eps),
                                      float(weight_decay), OPT_ADAMW if decoupled else OPT_ADAM,
                                      1 if zero_grad else 0, _stream()), "yr_adam_dense_multi")


def adam_dense_flat(tensors, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False,
                    grad_count=None):
    """One launch, 16 bytes per lane, for up to ADAM_MULTI_MAX tensors of any size (yr_adam_dense_flat).
    ``tensors``: (p, g, m, v, touched, clear[, scaled]) — ``touched`` a uint8 mark per row of p (or None),
    ``clear`` 1 / True: the gradient is zeroed after it is read, 2: only where it is non-zero, ``scaled`` whether it is first multiplied by
    1 / ``grad_count`` (a spread int32 count on the device)."""
    import ctypes
    lib = _lib.load()
    if len(tensors) > ADAM_MULTI_MAX:
        raise EngineError(f"at most {ADAM_MULTI_MAX} tensors per launch")
    step_size, bc2_sqrt = adam_scalars(step, lr, beta1, beta2)
    f32 = torch.float32
    n = len(tensors)
    cols = [(ctypes.c_void_p * n)(*[_dev(t[k], f32, name) for t in tensors]) for k, name in enumerate("pgmv")]
    for t in tensors:
        if not (t[0].numel() == t[1].numel() == t[2].numel() == t[3].numel()):
            raise EngineError("p/g/m/v sizes differ")
    counts = (ctypes.c_int64 * n)(*[t[0].numel() for t in tensors])
    marks = (ctypes.c_void_p * n)(*[_opt(t[4], torch.uint8, "touched") for t in tensors])
    widths = (ctypes.c_int * n)(*[int(t[0].shape[-1]) if t[4] is not None else 0 for t in tensors])
    clear = (ctypes.c_int * n)(*[int(t[5]) for t in tensors])
    scaled = (ctypes.c_int * n)(*[1 if len(t) > 6 and t[6] else 0 for t in tensors])
    if grad_count is not None and grad_count.numel() != COUNT_WORDS:
        raise EngineError(f"grad_count: a spread count of {COUNT_WORDS} int32 words")
    check(lib.yr_adam_dense_flat(cols[0], cols[1], cols[2], cols[3], counts, marks, widths, clear, scaled,
                                 _opt(grad_count, torch.int32, "grad_count"), n, float(lr),
                                 float(step_size), float(bc2_sqrt), float(beta1), float(beta2), float(eps),
                                 float(weight_decay), OPT_ADAMW if decoupled else OPT_ADAM, _stream()),
          "yr_adam_dense_flat")


def sgd_dense(p, g, lr, weight_decay=0.0, zero_grad=False):
    """Dense SGD step in place (reference trainers/base_trainer.py:39-40)."""
    lib = _lib.load()
    n = p.numel()
    check(lib.yr_sgd_dense(_dev(p, torch.float32, "p"), _dev(g, torch.float32, "g"), n,
                           float(lr), float(weight_decay), 1 if zero_grad else 0, _stream()), "yr_sgd_dense")
