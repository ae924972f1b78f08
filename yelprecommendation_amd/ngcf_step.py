"""One NGCF training step as ONE engine call (the NGCF counterpart of bpr_step.py / cdae_step.py).

What the reference does per batch (trainers/ngcf_trainer.py:104-116) —
``pos, neg = model.bpr_forward(u, p, n, L); optimizer.zero_grad(); loss = BPRLoss()(pos, neg); loss.backward();
optimizer.step(); train_loss += loss.item()`` — is ~35 kernel launches.  Through Python (one autograd node with
hand-written gradients, models/ngcf.py) those launches cost the host 0.55-0.6 ms per step, a floor under the step
whatever the GPU does; ``yr_ngcf_bpr_step`` (csrc/ngcf_step.hip) issues the SAME launches from C in one call.  It
trains the module's own parameters and the optimizer's own Adam state in place (``optimizer.state_dict()`` and
checkpoints are unchanged), keeps the running loss on the device, and propagates layer k on the rows the batch's
scores need when that set is small (``cfg.ngcf_subset_fraction``, see models/ngcf.py).  Results equal the autograd
route's: the same kernels on the same data (tests/test_gpu_ngcf.py).
"""
import ctypes

import torch

from . import _lib, engine


class NGCFStep:
    def __init__(self, model, optimizer, graph, subset_fraction=0.5):
        from . import optim
        if not isinstance(optimizer, optim.Adam):
            raise NotImplementedError("NGCFStep: optimizer adam or adamw")
        self.model, self.optimizer, self.graph = model, optimizer, graph
        self.subset_fraction = float(subset_fraction)
        self.params = [model.embedding.weight] + [w.weight for w in model.W1] + [w.weight for w in model.W2]
        self.K = len(model.W1)
        n, d = self.params[0].shape
        if n != graph.n:
            raise engine.EngineError(f"embedding table has {n} rows, the graph {graph.n}")
        self.n, self.D = n, d
        dev = self.params[0].device
        f32 = torch.float32
        for p in self.params:                                                   # the optimizer's own Adam state
            engine._dev(p.data, f32, "parameter")
            st = optimizer.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        arr = ctypes.c_void_p * len(self.params)
        self._p = arr(*[p.data.data_ptr() for p in self.params])
        self._m = arr(*[engine._dev(optimizer.state[p]["exp_avg"], f32, "exp_avg") for p in self.params])
        self._v = arr(*[engine._dev(optimizer.state[p]["exp_avg_sq"], f32, "exp_avg_sq") for p in self.params])
        self._bound = [p.data.data_ptr() for p in self.params]
        self.loss = torch.zeros(1, dtype=f32, device=dev)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.flag = model._flag()
        self._ws, self._ws_batch = None, -1
        self._lib = _lib.load()
        g = graph
        self._graph_args = (engine._dev(g.rowptr, torch.int32, "rowptr"), engine._dev(g.col, torch.int32, "col"),
                            engine._dev(g.val, f32, "val"), g.n, g.nnz,
                            engine._opt(g.heavy_rows, torch.int32, "heavy_rows"), g.n_heavy, g.heavy_threshold)

    def bound_to(self, model, optimizer, graph):
        """Still bound to these parameter / state tensors (load_state_dict replaces them)?"""
        if model is not self.model or optimizer is not self.optimizer or graph is not self.graph:
            return False
        ps = [model.embedding.weight] + [w.weight for w in model.W1] + [w.weight for w in model.W2]
        return (len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params))
                and all(p.data.data_ptr() == q for p, q in zip(ps, self._bound))
                and all(optimizer.state[p].get("exp_avg") is not None
                        and optimizer.state[p]["exp_avg"].data_ptr() == self._m[k] for k, p in enumerate(ps)))

    def _workspace(self, B):
        if self._ws is None or B > self._ws_batch:
            nbytes = int(self._lib.yr_ngcf_step_workspace_bytes(self.n, self.D, self.K, B))
            if nbytes < 0:
                engine.check(nbytes, "yr_ngcf_step_workspace_bytes")
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.params[0].device)
            self._ws_batch = B
        return self._ws

    @torch.no_grad()
    def step(self, user_id, pos_item, neg_item):
        B = user_id.numel()
        i64 = torch.int64
        u, p, n = (engine._dev(t.contiguous(), i64, name) if B else None
                   for t, name in ((user_id, "user_id"), (pos_item, "pos_item"), (neg_item, "neg_item")))
        if B and not (pos_item.numel() == B and neg_item.numel() == B):
            raise engine.EngineError("user_id / pos_item / neg_item lengths differ")
        ws = self._workspace(B)
        group = self.optimizer.param_groups[0]
        st = [self.optimizer.state[q] for q in self.params]
        t = int(st[0]["step"]) + 1
        b1, b2 = group["betas"]
        step_size, bc2_sqrt = engine.adam_scalars(t, group["lr"], b1, b2)
        rc = self._lib.yr_ngcf_bpr_step(*self._graph_args, self.model.num_users, self._p, self._m, self._v, self.K,
                                        self.D, u, p, n, B, group["lr"], step_size, bc2_sqrt, b1, b2, group["eps"],
                                        group["weight_decay"],
                                        engine.OPT_ADAMW if self.optimizer._decoupled else engine.OPT_ADAM,
                                        self.subset_fraction, ws.data_ptr(), ws.numel(), self.loss.data_ptr(),
                                        self.loss_accum.data_ptr(), self.flag.data_ptr(), engine._stream())
        if rc:
            engine.check(rc, "yr_ngcf_bpr_step")
        for s in st:
            s["step"] = t

    def last_loss(self):
        """Mean loss of the last step (device scalar)."""
        return self.loss[0]

    def epoch_loss(self):
        """Sum of the per-step mean losses since the last call (what NGCFTrainer.train returns); one read-back."""
        v = float(self.loss_accum.item())
        self.loss_accum.zero_()
        return v

    def check(self):
        engine.raise_on_flag(self.flag, "NGCF")
