"""CDAE — drop-in for reference models/cdae.py:7-52.

Same constructor ``CDAE(cfg, num_items, num_users)`` (note the argument order), same parameters
(``hidden_layer.{weight,bias}`` Linear(I -> H), ``user_nodes.weight`` Embedding(U, H),
``output_layer.{weight,bias}`` Linear(H -> I)), same initialisation (xavier-uniform weights,
U(0,1) biases and user nodes, models/cdae.py:35-41) in the same order, same
``forward(user_id, x)`` / ``add_noise(x)``.

Underneath: both Linear layers and their three gradient products are float32 MFMA GEMMs
(``yr_gemm_f32``); bias + user-node add, dropout, sigmoid and their backward are small HIP
kernels; the whole forward is one autograd node with a hand-written backward.
"""
import torch
import torch.nn as nn

from .. import engine
from .base_model import BaseModel

_ACT = {"sigmoid": engine.ACT_SIGMOID, "identity": engine.ACT_IDENTITY}


def _split_k(K):
    """K-splits for the skinny products whose reduction runs over the catalogue."""
    return max(1, min(256, K // 256))


class _CDAEForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user_id, x_in, hidden_act, output_act, err_flag, Wh, bh, V, Wo, bo):
        user_id = user_id.contiguous()
        x_in = x_in.contiguous()
        Whd, Wod = Wh.detach(), Wo.detach()
        # z = act_h(x_in @ Wh^T + bh + V[user])            (models/cdae.py:49)
        z = engine.cdae_hidden_init(bh.detach(), V.detach(), user_id, err_flag=err_flag)
        engine.gemm_f32(x_in, Whd, transB=True, out=z, accumulate=True, split_k=_split_k(x_in.shape[1]))
        if hidden_act == engine.ACT_SIGMOID:
            engine.sigmoid_(z)
        # y = act_o(z @ Wo^T + bo)                           (models/cdae.py:52)
        y = engine.gemm_f32(z, Wod, transB=True, bias=bo.detach(), act=output_act)
        ctx.hidden_act, ctx.output_act = hidden_act, output_act
        ctx.save_for_backward(user_id, x_in, z, y, Whd, Wod, V.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        user_id, x_in, z, y, Wh, Wo, V = ctx.saved_tensors
        dy = dy.contiguous()
        # d pre-activation of the output layer (a fresh buffer: dy belongs to autograd)
        g = engine.sigmoid_bwd(dy, y) if ctx.output_act == engine.ACT_SIGMOID else dy
        dWo = engine.gemm_f32(g, z, transA=True)             # [I, H] = g^T z
        dbo = engine.colsum(g)
        dz = engine.gemm_f32(g, Wo, split_k=_split_k(g.shape[1]))       # [B, H] = g Wo
        if ctx.hidden_act == engine.ACT_SIGMOID:
            engine.sigmoid_bwd_(dz, z)
        dWh = engine.gemm_f32(dz, x_in, transA=True)         # [H, I] = dz^T x_in
        dbh = engine.colsum(dz)
        dV = torch.zeros_like(V)
        engine.row_scatter_add(dz, user_id, dV)
        return None, None, None, None, None, dWh, dbh, dV, dWo, dbo


class _CDAEForwardSparse(torch.autograd.Function):
    """The same node with the encoder over the NON-ZEROS of the (corrupted) input only
    (csrc/cdae_sparse.hip): row compaction with the dropout mask applied on the fly, gather-form encoder
    with bias / user-node add / activation fused, scatter-form dW_h.  Mathematically the dense node
    (the skipped terms are exact zeros); sums run in another order."""

    @staticmethod
    def forward(ctx, user_id, x, seed, p, hidden_act, output_act, err_flag, Wh, bh, V, Wo, bo):
        user_id = user_id.contiguous()
        rows = engine.SparseRows(x.contiguous(), seed, p)
        Whd, Wod = Wh.detach(), Wo.detach()
        z = engine.cdae_sparse_encode(rows, Whd, bh.detach(), V.detach(), user_id, hidden_act, err_flag=err_flag)
        y = engine.gemm_f32(z, Wod, transB=True, bias=bo.detach(), act=output_act)
        ctx.hidden_act, ctx.output_act, ctx.rows = hidden_act, output_act, rows
        ctx.save_for_backward(user_id, z, y, Whd, Wod, V.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        user_id, z, y, Wh, Wo, V = ctx.saved_tensors
        dy = dy.contiguous()
        g = engine.sigmoid_bwd(dy, y) if ctx.output_act == engine.ACT_SIGMOID else dy
        dWo = engine.gemm_f32(g, z, transA=True)             # [I, H] = g^T z
        dbo = engine.colsum(g)
        dz = engine.gemm_f32(g, Wo, split_k=_split_k(g.shape[1]))       # [B, H] = g Wo
        if ctx.hidden_act == engine.ACT_SIGMOID:
            engine.sigmoid_bwd_(dz, z)
        dWh = engine.cdae_sparse_dwh(ctx.rows, dz, torch.zeros_like(Wh))
        dbh = engine.colsum(dz)
        dV = torch.zeros_like(V)
        engine.row_scatter_add(dz, user_id, dV)
        return None, None, None, None, None, None, None, dWh, dbh, dV, dWo, dbo


class CDAE(BaseModel):

    def __init__(self, cfg, num_items, num_users):
        super().__init__()
        self.num_items = num_items
        self.num_users = num_users
        self.hidden_size = cfg.hidden_size
        self.device = cfg.device
        self.corruption_level = cfg.corruption_level

        self.dropout_layer = nn.Dropout(p=self.corruption_level)
        self.hidden_layer = nn.Linear(in_features=self.num_items, out_features=self.hidden_size,
                                      bias=True, device=self.device, dtype=torch.float32)
        self.user_nodes = nn.Embedding(num_embeddings=self.num_users, embedding_dim=self.hidden_size,
                                       device=self.device, dtype=torch.float32)
        self.output_layer = nn.Linear(in_features=self.hidden_size, out_features=self.num_items,
                                      bias=True, device=self.device, dtype=torch.float32)
        self.hidden_activation = self._activation_module(cfg.hidden_activation)
        self.output_activation = self._activation_module(cfg.output_activation)
        self._hidden_act = _ACT[cfg.hidden_activation]
        self._output_act = _ACT[cfg.output_activation]
        # encoder over the non-zeros of the input (default) or as a dense GEMM over the whole catalogue
        self.sparse_encoder = bool(cfg.get("sparse_encoder", True)) if hasattr(cfg, "get") else True
        self._err_flag = None
        self._init_weights()

    def _init_weights(self):
        # reference models/cdae.py:35-41
        for child in self.children():
            if isinstance(child, nn.Linear):
                nn.init.xavier_uniform_(child.weight)
                nn.init.uniform_(child.bias)
            elif isinstance(child, nn.Embedding):
                nn.init.uniform_(child.weight)

    def _flag(self):
        dev = self.user_nodes.weight.device
        if self._err_flag is None or self._err_flag.device != dev:
            self._err_flag = engine.new_error_flag(dev)
        return self._err_flag

    def check_indices(self):
        if self._err_flag is not None:
            engine.raise_on_flag(self._err_flag, "CDAE")

    def add_noise(self, x):
        """reference models/cdae.py:43-44: nn.Dropout(p) — inverted dropout in training mode,
        identity in eval mode.  One 62-bit seed per call comes from torch's (CPU) generator, so
        ``set_seed`` fixes the masks; the uniforms themselves are Philox draws inside the HIP
        kernel (no [B, I] random tensor).  ``engine.dropout(x, rnd, p)`` takes explicit uniforms."""
        if not self.training or self.corruption_level == 0:
            return x
        seed = int(torch.randint(0, 1 << 62, (1,)).item())
        return engine.dropout_seeded(x.contiguous(), seed, self.corruption_level)

    def _params(self):
        return (self.hidden_layer.weight, self.hidden_layer.bias, self.user_nodes.weight,
                self.output_layer.weight, self.output_layer.bias)

    def encode_decode(self, user_id, x_in):
        """forward() on an already-corrupted input (used by tests that replay recorded masks)."""
        if self.sparse_encoder:
            return _CDAEForwardSparse.apply(user_id, x_in, 0, 0.0, self._hidden_act, self._output_act, self._flag(),
                                            *self._params())
        return _CDAEForward.apply(user_id, x_in, self._hidden_act, self._output_act, self._flag(), *self._params())

    def forward(self, user_id, x):
        # reference models/cdae.py:46-52 (`if self.train:` is always true there; nn.Dropout itself
        # follows train()/eval(), which add_noise reproduces)
        if self.sparse_encoder and "add_noise" not in self.__dict__ and type(self).add_noise is CDAE.add_noise:
            # the dropout mask of add_noise() is applied while the rows are compacted: same seed draw from
            # torch's generator, same Philox words per position, no dense corrupted copy of x (an overridden
            # add_noise — the tests replay recorded dropout outcomes through it — is honoured below)
            p = self.corruption_level if self.training else 0.0
            seed = int(torch.randint(0, 1 << 62, (1,)).item()) if p > 0 else 0
            return _CDAEForwardSparse.apply(user_id, x, seed, p, self._hidden_act, self._output_act, self._flag(),
                                            *self._params())
        return self.encode_decode(user_id, self.add_noise(x))
