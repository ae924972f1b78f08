"""BPR-MF trainer — drop-in for reference trainers/mf_trainer.py:22-178.

Same constructor ``MFTrainer(cfg, num_items, num_users)`` and the same
``run / train / validate / evaluate / _generate_top_k_recommendation`` contract:
``train`` and ``validate`` return the SUM of per-batch mean losses, ``evaluate``
returns ``(precision, recall, map, ndcg)@top_n``.

Differences underneath (results equal to float rounding):
* with Adam/AdamW the whole batch step — 2 x forward, loss, backward, optimizer.step()
  (mf_trainer.py:106-112) — is ONE engine call (BPRMFStep, csrc/bpr_pull.hip) that shares the
  optimizer's state tensors; with SGD it is the fused gather/score/scatter-add kernel plus a
  dense SGD launch;
* the running loss stays on the device — one host sync per epoch, not per step
  (mf_trainer.py:114 syncs every batch);
* ``evaluate`` scores all eval users against the whole catalogue on the device with
  the train-item mask and top-k fused in, instead of a Python loop of per-user
  forward calls (mf_trainer.py:139-144).

Multi-GPU (new; the reference is single-process): when ``torch.distributed`` is initialised with
more than one rank, the interaction matrix is sharded by user.  Every rank runs the
same script on the same seeded loaders; of each global batch it trains the triplets of the users
it owns (the epoch is cut by owner once, on the device), the item gradient is all-reduced inside ``BPRMFStep`` (RCCL), the
user rows are exchanged once per epoch, ``evaluate`` scores only the rank's users and sums the
per-user metric terms with one 6-double all-reduce, and only rank 0 writes ``best_model.pt``.
"""
import numpy as np
import torch

from .. import engine
from ..bpr_step import BPRMFStep
from ..loss import BPRLoss
from ..metric import ranking_metrics
from ..models.mf import MatrixFactorization
from ..user_shard import UserShard
from ..utils import logger
from .base_trainer import BaseTrainer


def _dist():
    """torch.distributed when a multi-rank process group is up, else None."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _lists_to_csr(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum([len(l) for l in lists], out=ptr[1:])
    idx = np.fromiter((x for l in lists for x in l), dtype=np.int64, count=int(ptr[-1]))
    return ptr, idx


class MFTrainer(BaseTrainer):
    def __init__(self, cfg, num_items: int, num_users: int) -> None:
        super().__init__(cfg)
        self.num_items = num_items
        self.num_users = num_users
        self.model = MatrixFactorization(self.cfg, num_users, num_items).to(self.device)
        self.optimizer = self._optimizer(self.cfg.optimizer, self.model, self.cfg.lr, self.cfg.weight_decay)
        self.loss = self._loss()
        self._loss_accum = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._eval_cache = {}
        self._eval_hints = {}                              # eval set -> its last top-n lists (hints of the next evaluation)
        dist = _dist()
        self.world_size = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        self.shard = UserShard(num_users, self.world_size, self.rank)
        self._step = None                                  # the BPRMFStep, kept across epochs

    def _loss(self):
        return BPRLoss()

    def run(self, train_dataloader, valid_dataloader, valid_eval_data):
        # reference mf_trainer.py:34-97
        logger.info("[Trainer] run...")
        best = (1e+6, .0, .0, .0, .0)
        endurance = 0
        for epoch in range(self.cfg.epochs):
            train_loss = self.train(train_dataloader)
            valid_loss = self.validate(valid_dataloader)
            current = (valid_loss,) + tuple(self.evaluate(valid_eval_data, 'valid'))
            self._log_epoch(epoch, train_loss, *current)
            if self._is_surpass_best_metric(current=current, best=best):
                logger.info("[Trainer] update best model...")
                best = current
                endurance = 0
                if self.rank == 0:
                    torch.save(self.model.state_dict(), f'{self.cfg.model_dir}/best_model.pt')
                if self.world_size > 1:
                    _dist().barrier()                      # the file is complete before anyone loads it
            else:
                endurance += 1
                if endurance > self.cfg.patience:
                    logger.info("[Trainer] ealry stopping...")
                    break

    def _batch(self, data):
        dev = self.device
        return (data['user_id'].to(dev, non_blocking=True), data['pos_item'].to(dev, non_blocking=True),
                data['neg_item'].to(dev, non_blocking=True))

    def _fused_step(self):
        """A BPRMFStep over the model's tables and the optimizer's own Adam state (created on
        first use exactly as optimizer.step() would), or None when the optimizer is not Adam/AdamW."""
        from .. import optim
        if not isinstance(self.optimizer, optim.Adam):
            return None
        U, I = self.model.user_embedding.weight, self.model.item_embedding.weight
        group = self.optimizer.param_groups[0]
        for p in (U, I):
            st = self.optimizer.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        sU, sI = self.optimizer.state[U], self.optimizer.state[I]
        lo, hi = self.shard.lo, self.shard.hi              # the whole table on one GPU
        # one step object for the whole run (its second user buffer, workspaces and partials are
        # allocated once); rebuilt only when the tensors it is bound to were replaced, e.g. by
        # optimizer.load_state_dict()
        bound = (U.data[lo:hi].data_ptr(), I.data.data_ptr(), sU["exp_avg"].data_ptr(), sU["exp_avg_sq"].data_ptr(),
                 sI["exp_avg"].data_ptr(), sI["exp_avg_sq"].data_ptr(), group["lr"], group["betas"], group["eps"],
                 group["weight_decay"])
        st = self._step
        if st is not None and self._step_bound[1:] == bound[1:] and st.U.data_ptr() == bound[0]:
            st.t = int(sU["step"])
            return st
        self._step = BPRMFStep(U.data[lo:hi], I.data, lr=group["lr"], betas=group["betas"], eps=group["eps"],
                               weight_decay=group["weight_decay"],
                               optimizer="adamw" if self.optimizer._decoupled else "adam",
                               world_size=self.world_size, rank=self.rank,
                               deterministic=bool(self.cfg.get("deterministic", False)),
                               item_exchange=self.cfg.get("item_exchange", "all_reduce"),
                               state=dict(mU=sU["exp_avg"][lo:hi], vU=sU["exp_avg_sq"][lo:hi], mI=sI["exp_avg"],
                                          vI=sI["exp_avg_sq"], t=sU["step"]))
        self._step_bound = bound
        return self._step

    def _exchange_user_rows(self):
        """Every rank receives the rows the other ranks trained (once per epoch, 8 MB in total)."""
        dist = _dist()
        U = self.model.user_embedding.weight.data
        for r in range(self.world_size):
            lo, hi = self.shard.bounds(r)
            if hi > lo:
                dist.broadcast(U[lo:hi], src=r)

    def train(self, train_dataloader) -> float:
        # reference mf_trainer.py:100-116
        self.model.train()
        step = self._fused_step()
        if step is None and self.world_size > 1:
            return self._train_sharded_sgd(train_dataloader)
        if step is None:                                   # SGD: fused fwd/bwd kernel + dense update
            self._loss_accum.zero_()
            for data in train_dataloader:
                user_id, pos_item, neg_item = self._batch(data)
                self.model.bpr_loss_backward(user_id, pos_item, neg_item, loss_accum=self._loss_accum)
                self.optimizer.step(zero_grad=True)
            self.model.check_indices()
            return float(self._loss_accum.item())
        U, I = self.model.user_embedding.weight, self.model.item_embedding.weight
        if self.world_size > 1:
            # The loaders do not depend on the model, so the epoch's batches are drawn first and cut by
            # owner ONCE on the device: one host read-back per epoch (the per-batch counts) instead of a
            # boolean-mask selection — and its synchronisation — in front of every step.
            batches = [self._batch(data) for data in train_dataloader]
            sizes = [b[0].numel() for b in batches]
            if batches:
                eu, ep, en = (torch.cat([b[k] for b in batches]) for k in range(3))
                mine = self.shard.mine(eu)
                batch_of = torch.repeat_interleave(torch.arange(len(sizes), device=eu.device),
                                                   torch.tensor(sizes, device=eu.device))
                counts = torch.bincount(batch_of[mine], minlength=len(sizes)).tolist()
                lu, lp, ln = self.shard.localize(eu[mine]).contiguous(), ep[mine].contiguous(), en[mine].contiguous()
                at = 0
                for size, c in zip(sizes, counts):
                    step.step(lu[at:at + c], lp[at:at + c], ln[at:at + c], global_batch=size)
                    at += c
            own = U.data[self.shard.lo:self.shard.hi]
            if step.U.data_ptr() != own.data_ptr():
                own.copy_(step.U)
                step.U, step._U_alt = own, step.U          # keep the step bound to the module's rows
            self._exchange_user_rows()
        else:
            for data in train_dataloader:
                step.step(*self._batch(data))
            # hand the (double-buffered) user table back to the module
            U.data = step.U
        self.optimizer.state[U]["step"] = self.optimizer.state[I]["step"] = step.t
        total = step.epoch_loss()
        if self.world_size > 1:
            # every rank learns of a bad index on any rank and raises with it (a rank raising alone
            # would leave its peers blocked in the next collective)
            _dist().all_reduce(step.flag, op=_dist().ReduceOp.MAX)
        step.check()
        return total

    def _train_sharded_sgd(self, train_dataloader) -> float:
        """SGD (trainers/base_trainer.py:39-40) under user sharding: every rank scatters the gradients of the
        triplets whose users it owns with 1 / B_global (the mean of loss.py:27 over the whole batch), the dense
        item gradient is all-reduced (RCCL), and the dense update — p -= lr (g + wd p) — is applied to the item table
        (identical on every rank) and to the rank's OWN user rows; the other user rows arrive with the per-epoch
        exchange.  Equal to the single-process SGD epoch up to summation order."""
        dist = _dist()
        from .. import optim
        if not isinstance(self.optimizer, optim.SGD):
            raise NotImplementedError(f"user-sharded training: optimizer {type(self.optimizer).__name__}")
        U, I = self.model.user_embedding.weight, self.model.item_embedding.weight
        group = self.optimizer.param_groups[0]
        lo, hi = self.shard.lo, self.shard.hi
        self._loss_accum.zero_()
        for data in train_dataloader:
            user_id, pos_item, neg_item = self._batch(data)
            B = user_id.numel()
            mine = self.shard.mine(user_id)
            self.model.bpr_loss_backward(user_id[mine].contiguous(), pos_item[mine].contiguous(),
                                         neg_item[mine].contiguous(), loss_accum=self._loss_accum,
                                         inv_batch=1.0 / B if B else 0.0)
            dist.all_reduce(I.grad, op=dist.ReduceOp.SUM)
            engine.sgd_dense(I.data, I.grad, group["lr"], group["weight_decay"], zero_grad=True)
            if hi > lo:
                engine.sgd_dense(U.data[lo:hi], U.grad[lo:hi], group["lr"], group["weight_decay"], zero_grad=True)
        self._exchange_user_rows()
        dist.all_reduce(self._loss_accum, op=dist.ReduceOp.SUM)      # per-rank partial means (already / B_global)
        flag = self.model._flag()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)                  # a bad index raises on every rank together
        self.model.check_indices()
        return float(self._loss_accum.item())

    def save_checkpoint(self, path, **extra):
        """As BaseTrainer.save_checkpoint; when the run is user-sharded, rank r holds the current Adam
        moments of ITS user rows only, so the slices are exchanged first and rank 0 writes the file."""
        if self.world_size > 1:
            U = self.model.user_embedding.weight
            st = self.optimizer.state.get(U, {})
            for name in ("exp_avg", "exp_avg_sq"):
                if name in st:
                    for r in range(self.world_size):
                        lo, hi = self.shard.bounds(r)
                        if hi > lo:
                            _dist().broadcast(st[name][lo:hi], src=r)
            if self.cfg.get("item_exchange", "all_reduce") == "reduce_scatter":
                # the item moments of a rank are current on its slice of the item rows only
                from ..user_shard import ItemSlices
                I = self.model.item_embedding.weight
                sti = self.optimizer.state.get(I, {})
                for name in ("exp_avg", "exp_avg_sq"):
                    if name in sti:
                        for r in range(self.world_size):
                            sl = ItemSlices(self.num_items, self.world_size, r)
                            if sl.hi > sl.lo:
                                _dist().broadcast(sti[name][sl.lo:sl.hi], src=r)
            if self.rank == 0:
                super().save_checkpoint(path, **extra)
            _dist().barrier()
            return
        super().save_checkpoint(path, **extra)

    def validate(self, valid_dataloader) -> float:
        # reference mf_trainer.py:118-132
        self.model.eval()
        self._loss_accum.zero_()
        from ..data.triplets import EpochLoader
        if isinstance(valid_dataloader, EpochLoader) and self.cfg.get("whole_epoch_validate", True):
            # the device-side loader holds the whole epoch: the sum over batches of the batch-MEAN loss is two
            # launches — all full batches with weight 1 / batch_size, the short last one with 1 / its length —
            # instead of one launch + one finalize per batch (9,600 batches at the reference's batch size 32:
            # 194 ms of host time for 3 ms of kernels)
            u, p, n = valid_dataloader.sampler.epoch(valid_dataloader.shuffle)
            bs, total = valid_dataloader.batch_size, u.numel()
            full = total // bs * bs
            for lo, hi in ((0, full), (full, total)):
                if hi > lo:
                    self.model.bpr_loss_backward(u[lo:hi], p[lo:hi], n[lo:hi], loss_accum=self._loss_accum,
                                                 backward=False, inv_batch=1.0 / min(bs, hi - lo))
            self.model.check_indices()
            return float(self._loss_accum.item())
        for data in valid_dataloader:
            user_id, pos_item, neg_item = self._batch(data)
            self.model.bpr_loss_backward(user_id, pos_item, neg_item, loss_accum=self._loss_accum, backward=False)
        self.model.check_indices()
        return float(self._loss_accum.item())

    # -- evaluation -------------------------------------------------------------------------
    def _eval_arrays(self, eval_data):
        """eval_data: DataFrame indexed by user_id with list columns 'pos_items' and
        'mask_items' (reference mf_data_pipeline.py:49-50).  Cached CSR + device copies."""
        key = id(eval_data)
        if key not in self._eval_cache:
            users = np.asarray(eval_data.index.values, dtype=np.int64)
            pos = [list(x) for x in eval_data['pos_items']]
            # ascending ids inside every mask list: what the fused evaluation kernel walks with a cursor
            # (sorted once here instead of on the device at every evaluate())
            masks = [sorted(x) for x in eval_data['mask_items']]
            if self.world_size > 1:                            # this rank scores the users it owns
                keep = np.flatnonzero(self.shard.mine(users))
                users, pos, masks = users[keep], [pos[k] for k in keep], [masks[k] for k in keep]
            mask_ptr, mask_idx = _lists_to_csr(masks)
            pos_ptr, pos_idx = _lists_to_csr(pos)
            dev = self.device
            self._eval_cache[key] = (eval_data, pos, torch.from_numpy(users).to(dev),
                                     torch.from_numpy(mask_ptr).to(dev), torch.from_numpy(mask_idx).to(dev),
                                     torch.from_numpy(pos_ptr).to(dev), torch.from_numpy(pos_idx).to(dev))
        return self._eval_cache[key][1:5]

    def recommend(self, users, mask_ptr, mask_idx, hint_key=None):
        """Top-``top_n`` item ids per user, masked items excluded ([n_users, top_n] int64, device).
        ``hint_key``: evaluations under the same key hand their result to the next one as hint lists
        (engine.mf_eval_topk: the lists start from the smallest score among a user's previous top-n under the
        CURRENT model — a bound the result cannot depend on; cfg.eval_hints=False turns it off)."""
        U, I = self.model.user_embedding.weight.detach(), self.model.item_embedding.weight.detach()
        if engine.fused_eval_supports(self.cfg.top_n, U.shape[1]):   # fused scores + mask + top-k; masks are pre-sorted
            # cfg.eval_precision: "bf16x3" (default; f32 scores from three-term bf16 splits) or "f32"
            hinted = hint_key is not None and self.cfg.get("eval_hints", True)
            hint = self._eval_hints.get(hint_key) if hinted else None
            if hint is not None and tuple(hint.shape) != (users.numel(), self.cfg.top_n):
                hint = None
            top = engine.mf_eval_topk(U, I, users.contiguous(), mask_ptr, mask_idx, self.cfg.top_n,
                                      precision=self.cfg.get("eval_precision", "bf16x3"), hint=hint)
            if hinted:
                self._eval_hints[hint_key] = top
            return top
        return engine.mf_recommend(U, I, users, mask_ptr, mask_idx, self.cfg.top_n, fused=False)

    def evaluate(self, eval_data, mode='valid') -> tuple:
        # reference mf_trainer.py:134-161
        self.model.eval()
        actual, users, mask_ptr, mask_idx = self._eval_arrays(eval_data)
        if self.world_size > 1:
            p, r, m, n = self._evaluate_sharded(eval_data, users, mask_ptr, mask_idx)
        elif self.cfg.get("host_metrics", False):
            predicted = self.recommend(users, mask_ptr, mask_idx, hint_key=id(eval_data))
            # the checked definition (Python loops of reference metric.py); ~100x the kernel time
            p, r, m, n = ranking_metrics(actual, predicted.cpu().numpy().tolist(), self.cfg.top_n)
        else:
            predicted = self.recommend(users, mask_ptr, mask_idx, hint_key=id(eval_data))
            pos_ptr, pos_idx = self._eval_cache[id(eval_data)][5:7]
            p, r, m, n = engine.rank_metrics(predicted, pos_ptr, pos_idx)[:4].tolist()
        if mode == 'test':
            logger.info(f"[Trainer] Test > precision@{self.cfg.top_n} : {p:.4f} / Recall@{self.cfg.top_n}: {r:.4f} / "
                        f"MAP@{self.cfg.top_n}: {m:.4f} / NDCG@{self.cfg.top_n}: {n:.4f}")
        return (p, r, m, n)

    def _evaluate_sharded(self, eval_data, users, mask_ptr, mask_idx):
        """Metrics over ALL eval users from per-rank sums: [non-empty users, 4 sums, users]."""
        sums = torch.zeros(6, dtype=torch.float64, device=self.device)
        if users.numel():
            pos_ptr, pos_idx = self._eval_cache[id(eval_data)][5:7]
            sums = engine.rank_metrics(self.recommend(users, mask_ptr, mask_idx, hint_key=id(eval_data)), pos_ptr,
                                       pos_idx)[4:10].clone()
        _dist().all_reduce(sums)
        cnt, ps, rs, ms, ns, total = sums.tolist()
        return (ps / total, rs / cnt, ms / cnt, ns / cnt)

    def _generate_top_k_recommendation(self, pred, mask_items):
        """reference mf_trainer.py:163-178 for ONE user's score vector (kept for callers
        that score users one at a time); the batched path is :meth:`recommend`."""
        dev = pred.device
        mask = torch.as_tensor(np.asarray(mask_items, dtype=np.int64), device=dev)
        ptr = torch.tensor([0, mask.numel()], dtype=torch.int64, device=dev)
        top = engine.topk_masked(pred.detach().reshape(1, -1).contiguous(), ptr, mask, self.cfg.top_n)
        return top[0].cpu().numpy()
