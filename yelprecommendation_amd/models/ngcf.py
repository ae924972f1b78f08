"""NGCF — drop-in for reference models/ngcf.py:7-72.

Same constructor ``NGCF(cfg, num_users, num_items)``, same parameters (``embedding.weight``
[users + items, D] with items offset by ``num_users``; ``W1.{k}.weight`` / ``W2.{k}.weight``
``Linear(D, D, bias=False)`` x ``cfg.num_orders``), same default initialisation (the reference
defines ``_init_weights`` but never calls it, models/ngcf.py:25-28, so the embedding keeps
N(0, 1) and the linears kaiming-uniform), same methods ``bpr_forward / forward /
embedding_propagation``.

``laplacian_matrix`` may be the torch sparse COO tensor the reference's pipeline produces or a
:class:`yelprecommendation_amd.graph.LaplacianCSR`; it is converted to CSR once and cached.

Underneath: one CSR SpMM per layer feeds both terms ((L + I)E = LE + E — the reference builds
eye(N, N).to_sparse() per layer per batch), the dense part runs on the f32 matrix cores, the
scores are the gather+dot kernel over the K+1 layer buffers, and the whole forward/backward is
ONE autograd node with hand-written gradients accumulated in place (no ATen arithmetic).
"""
import torch
import torch.nn as nn

from .. import engine
from ..graph import LaplacianCSR
from .base_model import BaseModel


def _zero_like(t):
    return torch.zeros_like(t, memory_format=torch.contiguous_format)


class _NGCFScores(torch.autograd.Function):
    """(pos, neg) of bpr_forward — or pos only when ``neg_ids`` is None (forward) — with the
    gradients of the embedding table and every W1/W2."""

    @staticmethod
    def forward(ctx, graph, num_users, subset_fraction, user_id, pos_ids, neg_ids, err_flag, E0, *weights):
        K = len(weights) // 2
        W1s, W2s = weights[:K], weights[K:]
        E0d = E0.detach()
        user_id = user_id.contiguous()
        pos_ids = pos_ids.contiguous()
        neg_ids = None if neg_ids is None else neg_ids.contiguous()
        # Batch-aware propagation: the scores read layer K at the batch's rows only (models/ngcf.py:37-39), layer
        # k-1 is needed at those rows and their neighbours, ...: sets[k] is the NGCFRowSet layer k + 1 is computed
        # on, or None for the whole graph.  Which layers are restricted is decided on the host from an ESTIMATE of
        # the set sizes (rows x (1 + average degree) per hop, against subset_fraction x N) so that every rank /
        # run takes the same launches without reading a size back; the kernels themselves take the exact sets.
        sets = _subset_plan(graph, num_users, K, user_id, pos_ids, neg_ids, subset_fraction)
        layers, Zs = [E0d], []
        for k in range(K):
            if sets[k] is None:
                Z = engine.spmm_csr(graph, layers[-1])
                out = engine.ngcf_dense_fwd(layers[-1], Z, W1s[k].detach(), W2s[k].detach())
            else:
                # rows outside the set stay unwritten: nothing downstream reads them (the next layer's set and
                # its neighbours lie inside this one)
                Z = engine.spmm_csr_subset(graph, layers[-1], torch.empty_like(E0d), rows=sets[k])
                out = engine.ngcf_dense_fwd(layers[-1], Z, W1s[k].detach(), W2s[k].detach(), rows=sets[k])
            Zs.append(Z)
            layers.append(out)
        ctx.sets = sets
        # users are rows [0, num_users), items rows [num_users, N) of every layer buffer
        res = engine.ngcf_score(layers, num_users, user_id, pos_ids, neg_ids, err_flag=err_flag)
        pos, neg = res if neg_ids is not None else (res, None)
        ctx.graph, ctx.num_users, ctx.K, ctx.err_flag = graph, num_users, K, err_flag
        ctx.has_neg = neg_ids is not None
        ctx.save_for_backward(user_id, pos_ids, neg_ids if neg_ids is not None else pos_ids,
                              *layers, *Zs, *[w.detach() for w in weights])
        if neg_ids is None:
            return pos
        return pos, neg

    @staticmethod
    def backward(ctx, gpos, gneg=None):
        K, nu, graph = ctx.K, ctx.num_users, ctx.graph
        saved = ctx.saved_tensors
        user_id, pos_ids, neg_ids = saved[:3]
        layers = saved[3:3 + K + 1]
        Zs = saved[4 + K:4 + 2 * K]
        weights = saved[4 + 2 * K:]
        W1s, W2s = weights[:K], weights[K:]
        gpos = gpos.contiguous()
        # gradient of every layer buffer from the scores (dense scatter-add, like index_add_)
        # one zero-fill for all layer gradients, one for all weight gradients, one batched transpose
        # for all weights: these were ~25 launch-bound ATen kernels per step
        dstack = torch.zeros((K + 1,) + tuple(layers[0].shape), dtype=layers[0].dtype, device=layers[0].device)
        dlayers = list(dstack.unbind(0))
        engine.ngcf_score_backward(layers, dlayers, nu, user_id, pos_ids, neg_ids if ctx.has_neg else None,
                                   gpos, gneg.contiguous() if ctx.has_neg else None, err_flag=ctx.err_flag)
        dWstack = torch.zeros((2 * K,) + tuple(W1s[0].shape), dtype=W1s[0].dtype, device=W1s[0].device) if K else None
        WT = torch.stack(list(weights)).transpose(1, 2).contiguous() if K else None
        dW1s, dW2s = [None] * K, [None] * K
        for k in range(K - 1, -1, -1):
            dW1s[k], dW2s[k] = dWstack[k], dWstack[K + k]
            # dlayers[k] += dA + dH * Z ; dZ = dA + dH * E ; then dlayers[k] += L^T dZ (L symmetric)
            cur = ctx.sets[k]
            dZ = None if cur is None or cur.push else torch.zeros_like(layers[k])
            dZ = engine.ngcf_dense_bwd(dlayers[k + 1], layers[k + 1], layers[k], Zs[k], W1s[k], W2s[k],
                                       dlayers[k], dW1s[k], dW2s[k], dZ=dZ, W1T=WT[k], W2T=WT[K + k], rows=cur)
            if cur is None:
                engine.spmm_csr(graph, dZ, out=dlayers[k], accumulate=True)
            elif cur.push:
                # few rows: they scatter into their neighbours (cost = their non-zeros, not the graph's)
                engine.spmm_csr_push_rows(graph, dZ, dlayers[k], cur)
            else:
                # dZ is non-zero on this layer's set only (zero-filled elsewhere): the pull product, into the rows of
                # the set of the layer below alone (it holds every neighbour of this one)
                below = ctx.sets[k - 1] if k > 0 else None
                engine.spmm_csr_subset(graph, dZ, dlayers[k], rows=below, accumulate=True)
        return (None, None, None, None, None, None, None, dlayers[0], *dW1s, *dW2s)


PUSH_MAX_PAIRS = 50000       # as kPushMaxPairs of csrc/ngcf_step.hip: scatter form of the backward product below this


def _subset_plan(graph, num_users, K, user_id, pos_ids, neg_ids, fraction):
    """[set of layer 1, ..., set of layer K] (engine.NGCFRowSet or None = all rows), see _NGCFScores.forward."""
    sets = [None] * K
    if K == 0 or not fraction or fraction <= 0:
        return sets
    n = graph.n
    est = min(n, user_id.numel() * (2 if neg_ids is None else 3))
    hop = 1.0 + graph.nnz / max(n, 1)
    cur = None
    for k in range(K - 1, -1, -1):
        if est > fraction * n:
            break
        # the last layer's set has at most `est` rows (exact bound); a deeper set's size is not known on the host
        push = est * hop <= PUSH_MAX_PAIRS
        cur = (engine.ngcf_frontier_mark(num_users, n - num_users, user_id, pos_ids, neg_ids, max_rows=est, push=push)
               if cur is None else engine.ngcf_frontier_expand(graph, cur, push=push))
        sets[k] = cur
        est = min(n, int(est * hop))
    return sets


def _iadd(acc, x):
    """acc += x on the device without ATen arithmetic: the dense SGD kernel with lr = -1
    (p <- p - (-1) * g)."""
    engine.sgd_dense(acc, x, -1.0)
    return acc


class _NGCFLayer(torch.autograd.Function):
    """embedding_propagation as a stand-alone differentiable op (for callers that compose
    layers themselves; NGCF.bpr_forward / forward use the single fused node above)."""

    @staticmethod
    def forward(ctx, graph, E, W1, W2):
        Ed = E.detach().contiguous()
        Z = engine.spmm_csr(graph, Ed)
        out = engine.ngcf_dense_fwd(Ed, Z, W1.detach(), W2.detach())
        ctx.graph = graph
        ctx.save_for_backward(Ed, Z, out, W1.detach(), W2.detach())
        return out

    @staticmethod
    def backward(ctx, dout):
        E, Z, out, W1, W2 = ctx.saved_tensors
        dE, dW1, dW2 = _zero_like(E), _zero_like(W1), _zero_like(W2)
        dZ = engine.ngcf_dense_bwd(dout.contiguous(), out, E, Z, W1, W2, dE, dW1, dW2)
        engine.spmm_csr(ctx.graph, dZ, out=dE, accumulate=True)
        return None, dE, dW1, dW2


class NGCF(BaseModel):

    def __init__(self, cfg, num_users, num_items):
        super().__init__()
        self.cfg = cfg
        self.num_users = num_users
        self.num_items = num_items
        self.embedding = nn.Embedding(num_users + num_items, cfg.embed_size, dtype=torch.float32)
        self.W1 = nn.ModuleList([nn.Linear(cfg.embed_size, cfg.embed_size, bias=False)
                                 for _ in range(cfg.num_orders)])
        self.W2 = nn.ModuleList([nn.Linear(cfg.embed_size, cfg.embed_size, bias=False)
                                 for _ in range(cfg.num_orders)])
        self._graph_cache = (None, None)
        self._err_flag = None

    def _init_weights(self):
        # reference models/ngcf.py:25-28 — defined, never invoked
        for child in self.children():
            if isinstance(child, nn.Embedding):
                nn.init.xavier_uniform_(child.weight)

    def graph(self, laplacian_matrix) -> LaplacianCSR:
        if isinstance(laplacian_matrix, LaplacianCSR):
            return laplacian_matrix
        key, g = self._graph_cache
        if key is not laplacian_matrix:
            g = LaplacianCSR.from_torch_sparse(laplacian_matrix, device=self.embedding.weight.device)
            if g.symmetric is False:
                raise ValueError("the propagation matrix must be symmetric (D^-1/2 A D^-1/2)")
            self._graph_cache = (laplacian_matrix, g)
        return g

    def _flag(self):
        dev = self.embedding.weight.device
        if self._err_flag is None or self._err_flag.device != dev:
            self._err_flag = engine.new_error_flag(dev)
        return self._err_flag

    def check_indices(self):
        if self._err_flag is not None:
            engine.raise_on_flag(self._err_flag, "NGCF")

    def _weights(self):
        return [w.weight for w in self.W1] + [w.weight for w in self.W2]

    def _subset_fraction(self):
        """cfg.ngcf_subset_fraction (default 0.5): a layer is propagated on the rows the batch needs when their
        estimated number is below this fraction of the graph; 0 = always the whole graph (the reference's shape)."""
        get = getattr(self.cfg, "get", None)
        return float(get("ngcf_subset_fraction", 0.5)) if get else 0.5

    def bpr_forward(self, user_id, pos_item_ids, neg_item_ids, laplacian_matrix):
        # reference models/ngcf.py:30-45
        return _NGCFScores.apply(self.graph(laplacian_matrix), self.num_users, self._subset_fraction(), user_id,
                                 pos_item_ids, neg_item_ids, self._flag(), self.embedding.weight, *self._weights())

    def forward(self, user_id, item_id, laplacian_matrix):
        # reference models/ngcf.py:47-58
        return _NGCFScores.apply(self.graph(laplacian_matrix), self.num_users, self._subset_fraction(), user_id,
                                 item_id, None, self._flag(), self.embedding.weight, *self._weights())

    def embedding_propagation(self, last_embed, w1, w2, laplacian_matrix):
        # reference models/ngcf.py:60-72
        return _NGCFLayer.apply(self.graph(laplacian_matrix), last_embed, w1.weight, w2.weight)

    @torch.no_grad()
    def propagate(self, laplacian_matrix):
        """[E_0, ..., E_K] for the current parameters (evaluation: propagate once, score many)."""
        g = self.graph(laplacian_matrix)
        layers = [self.embedding.weight.detach()]
        for w1, w2 in zip(self.W1, self.W2):
            Z = engine.spmm_csr(g, layers[-1])
            layers.append(engine.ngcf_dense_fwd(layers[-1], Z, w1.weight.detach(), w2.weight.detach()))
        return layers
