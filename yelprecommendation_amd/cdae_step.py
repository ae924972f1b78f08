"""One CDAE training step as a fixed sequence of HIP launches (the CDAE counterpart of bpr_step.py).

What the reference does per batch (trainers/cdae_trainer.py:36-54) —
``pred = model(user_id, input_mask); optimizer.zero_grad(); loss(pred, input_mask, negative_mask).backward();
optimizer.step()`` — with models/cdae.py:46-52 and loss.py:12-16 underneath, computed without autograd and
without materialising ``pred`` or the loss gradient w.r.t. it:

    1  non-zeros of dropout_p(x) as per-row lists                                yr_cdae_compact_rows
    2  z = act_h(W_h . rows + b_h + V[u])                                        yr_cdae_sparse_encode
    3  (clear dz and the position counter: one memset)
    4  y = act_o(z W_o^T + b_o) on the matrix cores; epilogue: BCE terms of the selected positions ->
       per-workgroup loss partials + their count, G = d loss / d pre-activation without 1 / count
                                                                                 yr_cdae_decode_loss
    5  dW_o = (G^T z) / count, db_o = its row sums of G^T / count                yr_gemm_f32_ex (rowsum)
    6  dz = (G W_o) / count  (K split over the catalogue, atomics)               yr_gemm_f32_ex
    7  dz *= act_h'(z); db_h = column sums; dV[u] += dz; loss of the step        yr_cdae_hidden_bwd
    8  dW_h = dz^T . rows                                                        yr_cdae_sparse_dwh
    9  Adam / AdamW on all five parameters                                       yr_adam_dense_flat

Eleven launches (8 is a memset and two kernels) against ~26 through autograd, and 5 passes over [B, I]
data instead of 11.  With ``transposed_wh`` (what CDAETrainer uses) steps 2 and 8 work on an [I, H] working copy of
W_h: 2 reads one 512-byte row per input item, 7 and 8 become ONE launch (yr_cdae_hidden_bwd_dwh_t).

With NS-BCE the loss reads the prediction on the positions where target + negative_mask != 0 only (loss.py:14-16)
— (1 + neg_times) x the positives of a row, a fraction of a percent of the catalogue — and its gradient w.r.t.
every other position is exactly zero.  ``decoder="sampled"`` (the default then) replaces 4-6 by one pass over
those positions: step 1 also lists them (yr_cdae_compact_pair), and yr_cdae_sampled_decode computes, per
position, the prediction, its BCE term, and its contribution to dz, dW_o and db_o (the 1 / count of the mean is
applied by the consumers: hidden_bwd to dz, the Adam launch to dW_o / db_o).  Nine launches; the three 2.5 GFLOP
decoder products are gone.  Validation / evaluation still decode the whole catalogue (models/cdae.py).  The gradient buffers belong to the step: dW_h and dV stay all-zero between steps (the
Adam launch clears what it read — for dV only the rows the batch touched), dW_o / db_o / db_h are
overwritten whole.  Results equal the autograd route (models/cdae.py + loss.py + optim.py) up to float
rounding: the 1 / count factor is applied to the products instead of to G.
"""
import torch

from . import engine


class CDAEStep:
    def __init__(self, model, optimizer, negative_sampling=True, decoder="auto", transposed_wh=False):
        """``decoder``: "sampled" — forward, loss and the three decoder gradients on the loss positions only
        (NS-BCE reads nothing else; needs a negative mask), "dense" — the full-catalogue products on the matrix
        cores, "auto" — sampled when the loss is NS-BCE and the hidden size allows it.
        ``transposed_wh``: between ``acquire()`` (implicit in the first step) and ``release()`` the step trains a
        TRANSPOSED working copy of W_h and of its two Adam moments ([I, H]): an input item is then a 512-byte row
        for the encoder and for dW_h (contiguous float atomics straight into the gradient, no scratch, one launch),
        and the Adam launch reads / clears the gradient rows of the batch's items only.  Adam is element-wise, so
        the layout changes nothing in the arithmetic; ``release()`` writes the three tensors back — the caller must
        do that before anything else reads ``hidden_layer.weight`` or the optimizer state (CDAETrainer.train does,
        at the end of the epoch)."""
        from . import optim
        if not isinstance(optimizer, optim.Adam):
            raise NotImplementedError("CDAEStep: optimizer adam or adamw")
        self.model, self.optimizer, self.negative_sampling = model, optimizer, bool(negative_sampling)
        self.params = [model.hidden_layer.weight, model.hidden_layer.bias, model.user_nodes.weight,
                       model.output_layer.weight, model.output_layer.bias]
        H = model.hidden_size
        if H % 4:
            raise NotImplementedError("CDAEStep: hidden size must be a multiple of 4")
        # row marks (gradient rows read / cleared only where the batch touched them) need H / 4 lanes per row to
        # tile a wave
        self.row_marks = (H // 4) & (H // 4 - 1) == 0 and H // 4 <= 64
        can_sample = self.negative_sampling and self.row_marks and H <= 256
        if decoder == "auto":
            decoder = "sampled" if can_sample else "dense"
        if decoder not in ("sampled", "dense") or (decoder == "sampled" and not can_sample):
            raise NotImplementedError(f"CDAEStep: decoder {decoder!r} with negative_sampling={negative_sampling}, H={H}")
        self.decoder = decoder
        self.transposed_wh = bool(transposed_wh) and self.row_marks
        self._wht = None                               # (WhT, mT, vT, dWhT, item marks) while acquired
        dev = self.params[0].device
        Wh, bh, V, Wo, bo = (p.data for p in self.params)
        f32 = torch.float32
        self.dV = torch.zeros_like(V)                                          # all-zero between steps, like
        self.dWh = None if self.transposed_wh else torch.zeros_like(Wh)        # dW_h (or its transposed form)
        self.dbh = torch.zeros_like(bh)                 # accumulated into by the fused hidden backward: cleared by Adam
        self.hidden_size = H
        if decoder == "sampled":                                                # accumulated into: zero between steps
            self.dWo, self.dbo = torch.zeros_like(Wo), torch.zeros_like(bo)
        else:                                                                   # overwritten whole
            self.dWo, self.dbo = torch.empty_like(Wo), torch.empty_like(bo)
        self.touched_users = torch.zeros(V.shape[0], dtype=torch.uint8, device=dev) if self.row_marks else None
        self.stats = torch.zeros(2, dtype=f32, device=dev)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.flag = engine.new_error_flag(dev)
        self._batch = None
        for p in self.params:                                                   # the optimizer's own Adam state
            st = optimizer.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)

    def bound_to(self, model, optimizer):
        """Still bound to these parameter / state tensors (they are replaced by load_state_dict)?"""
        if model is not self.model or optimizer is not self.optimizer:
            return False
        ps = [model.hidden_layer.weight, model.hidden_layer.bias, model.user_nodes.weight,
              model.output_layer.weight, model.output_layer.bias]
        return all(a is b for a, b in zip(ps, self.params)) and all("exp_avg" in optimizer.state[p] for p in ps)

    def acquire(self):
        """Take W_h and its Adam moments into the transposed working layout (no-op when already there)."""
        if not self.transposed_wh or self._wht is not None:
            return
        Wh = self.params[0]
        st = self.optimizer.state[Wh]
        dev = Wh.device
        self._wht = (Wh.data.t().contiguous(), st["exp_avg"].t().contiguous(), st["exp_avg_sq"].t().contiguous(),
                     torch.zeros(Wh.shape[1], Wh.shape[0], dtype=torch.float32, device=dev),
                     torch.zeros(Wh.shape[1], dtype=torch.uint8, device=dev))

    def release(self):
        """Write the working copy back into ``hidden_layer.weight`` and the optimizer's moments."""
        if self._wht is None:
            return
        Wh = self.params[0]
        st = self.optimizer.state[Wh]
        WhT, mT, vT, _, _ = self._wht
        Wh.data.copy_(WhT.t())
        st["exp_avg"].copy_(mT.t())
        st["exp_avg_sq"].copy_(vT.t())
        self._wht = None

    def _buffers(self, B):
        if self._batch != B:
            Wh = self.params[0]
            H, I = Wh.shape
            dev, f32 = Wh.device, torch.float32
            self.z = torch.empty(B, H, dtype=f32, device=dev)
            blob = torch.zeros(B * H + engine.COUNT_WORDS, dtype=f32, device=dev)   # dz and the (spread) position
            self._blob = blob                                                   # counter: one memset clears both
            self.dz = blob[:B * H].view(B, H)
            self.count = blob[B * H:].view(torch.int32)
            self.row_count = torch.empty(B * engine.SPARSE_PARTS, dtype=torch.int32, device=dev)
            self.loss_lists = None                        # made on the first dense batch (step_lists brings its own)
            if self.decoder == "sampled":
                self.n_partials = B * engine.cdae_sampled_decode_splits(B)
            else:
                ldg = (I + 3) // 4 * 4                                            # 16-byte rows: the gradient products
                self.G = torch.empty(B, ldg, dtype=f32, device=dev)[:, :I]       # that read G take the tiled kernel
                self.n_partials = engine.cdae_decode_loss_partials(B, I)
            self.partials = torch.empty(self.n_partials, dtype=f32, device=dev)
            self._batch = B

    @torch.no_grad()
    def step(self, user_id, x, negative_mask=None, seed=0, p=0.0, x_in=None):
        """One training step on the batch.  ``x``: the uncorrupted [B, I] input (the loss target);
        ``negative_mask``: [B, I] or None (plain BCE over every position).  The encoder input is
        dropout_p(x) with the Philox mask of ``seed`` — or ``x_in`` when given (an already corrupted input)."""
        model = self.model
        Wh, bh, V, Wo, bo = (q.data for q in self.params)
        B = x.shape[0]
        if B == 0:
            return
        self._buffers(B)
        user_id = user_id.contiguous()
        x = x.contiguous()
        neg = None if negative_mask is None else negative_mask.contiguous()
        sampled = self.decoder == "sampled"
        if sampled and neg is None:
            raise engine.EngineError("the sampled decoder needs the negative mask")
        if sampled and self.loss_lists is None:
            n, dev = B * engine.SPARSE_PARTS * engine.sparse_part_columns(x.shape[1]), x.device
            self.loss_lists = (torch.empty(n, dtype=torch.int32, device=dev),
                               torch.empty(n, dtype=torch.float32, device=dev),
                               torch.empty(B * engine.SPARSE_PARTS, dtype=torch.int32, device=dev))
        if sampled and x_in is None:                      # both lists of every row from one pass over x and the mask
            rows = engine.SparseRows(x, seed, p, count=self.row_count, negative_mask=neg, loss_lists=self.loss_lists)
        else:
            if sampled:                                   # loss positions of (x, mask); the encoder lists follow
                engine.SparseRows(x, 0, 0.0, count=self.row_count, negative_mask=neg, loss_lists=self.loss_lists)
            rows = engine.SparseRows(x if x_in is None else x_in.contiguous(), seed if x_in is None else 0,
                                     p if x_in is None else 0.0, count=self.row_count)
        self._run(user_id, rows, self.loss_lists if sampled else None, x, neg)

    @torch.no_grad()
    def step_lists(self, user_id, lists):
        """One training step on a batch that arrives as lists (engine.TrainLists from data/cdae_batches.py:
        encoder input and loss positions straight from the per-user CSR — no dense row, no dense mask, no
        compaction pass).  Sampled decoder only."""
        if self.decoder != "sampled":
            raise NotImplementedError("step_lists needs the sampled decoder (NS-BCE)")
        if lists.B == 0:
            return
        lists.alive()
        self._buffers(lists.B)
        self._run(user_id.contiguous(), lists.rows, lists.loss, None, None)

    def _run(self, user_id, rows, loss_lists, x, neg):
        model = self.model
        Wh, bh, V, Wo, bo = (q.data for q in self.params)
        sampled = loss_lists is not None
        self.acquire()
        wt = self._wht
        if wt is not None:
            engine.cdae_sparse_encode(rows, wt[0], bh, V, user_id, model._hidden_act, err_flag=self.flag, out=self.z,
                                      transposed=True)
        else:
            engine.cdae_sparse_encode(rows, Wh, bh, V, user_id, model._hidden_act, err_flag=self.flag, out=self.z)
        self._blob.zero_()
        if sampled:
            engine.cdae_sampled_decode(loss_lists, self.z, Wo, bo, model._output_act, self.dz, self.dWo,
                                       self.dbo, self.partials, self.count)
        else:
            engine.cdae_decode_loss(self.z, Wo, bo, x, neg, model._output_act, self.G, self.partials, self.count)
            engine.gemm_f32(self.G, self.z, transA=True, out=self.dWo, alpha_count=self.count, rowsum=self.dbo)
            engine.gemm_f32(self.G, Wo, out=self.dz, accumulate=True, split_k=max(1, min(256, Wo.shape[0] // 256)),
                            alpha_count=self.count)
        if wt is not None and self.hidden_size <= 512:
            # hidden layer's backward, db_h, dV, dW_h^T and the step's loss in one launch (db_h accumulates: the
            # Adam launch clears it)
            engine.cdae_hidden_bwd_dwh_t(rows, self.dz, self.z, model._hidden_act, user_id, self.count, self.dV,
                                         self.touched_users, self.dbh, wt[3], wt[4], self.partials, self.n_partials,
                                         self.stats, self.loss_accum, scale_dz=sampled)
        else:
            engine.cdae_hidden_bwd(self.dz, self.z, model._hidden_act, user_id, self.dV, self.touched_users, self.dbh,
                                   self.partials, self.n_partials, self.count, self.stats, self.loss_accum,
                                   scale_dz=sampled)
            if wt is not None:
                engine.cdae_sparse_dwh_t(rows, self.dz, wt[3], wt[4])
            else:
                engine.cdae_sparse_dwh(rows, self.dz, self.dWh)
        group = self.optimizer.param_groups[0]
        st = [self.optimizer.state[q] for q in self.params]
        t = int(st[0]["step"]) + 1
        grads = (self.dWh, self.dbh, self.dV, self.dWo, self.dbo)
        marks = (None, None, self.touched_users, None, None)
        clear = (1, 1, int(self.touched_users is None), 2 if sampled else 0, int(sampled))
        scaled = (False, False, False, sampled, sampled)
        tensors = [(q.data, g, s["exp_avg"], s["exp_avg_sq"], m, c, sc)
                   for q, g, s, m, c, sc in list(zip(self.params, grads, st, marks, clear, scaled))[1 if wt is not None else 0:]]
        if wt is not None:                             # W_h: the working copy, gradient rows by item mark
            tensors.insert(0, (wt[0], wt[3], wt[1], wt[2], wt[4], 0, False))
        engine.adam_dense_flat(tensors, t, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                               group["weight_decay"], decoupled=self.optimizer._decoupled,
                               grad_count=self.count if sampled else None)
        for s in st:
            s["step"] = t

    def last_loss(self):
        """Mean loss of the last step (device scalar)."""
        return self.stats[0]

    def epoch_loss(self):
        """Sum of the per-step losses since the last call (what CDAETrainer.train returns); one read-back."""
        v = float(self.loss_accum.item())
        self.loss_accum.zero_()
        return v

    def check(self):
        engine.raise_on_flag(self.flag, "CDAE")
