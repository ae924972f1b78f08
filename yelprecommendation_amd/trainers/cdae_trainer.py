"""CDAE trainer — drop-in for reference trainers/cdae_trainer.py:20-144.

Same constructor ``CDAETrainer(cfg, num_items, num_users)`` and the same contract:
``train`` returns the sum of per-batch losses, ``validate`` returns
``(loss, precision, recall, map, ndcg)``, ``evaluate`` returns the four metrics.  Seen items are
masked the reference's way — scores multiplied by ``logical_not(input_mask)`` (-> 0, valid
because sigmoid > 0; trainers/cdae_trainer.py:132) — through the masked top-k kernel with
mask value 0.  The four metrics are computed on the device per batch (``yr_rank_metrics`` sums,
one read-back per validate / evaluate) instead of copying every [B, I] mask to the host for
``np.nonzero`` and the Python loops of metric.py; ``cfg.host_metrics=True`` selects the reference's
host route (``_generate_target_and_top_k_recommendation`` + ``metric.ranking_metrics``).
"""
import numpy as np
import torch

from .. import engine
from ..loss import BCELoss, NSBCELoss
from ..metric import ranking_metrics
from ..cdae_step import CDAEStep
from ..models.cdae import CDAE
from ..utils import log_metric, logger
from .base_trainer import BaseTrainer


class CDAETrainer(BaseTrainer):
    def __init__(self, cfg, num_items: int, num_users: int) -> None:
        super().__init__(cfg)
        self.model = CDAE(self.cfg, num_items, num_users)
        self.optimizer = self._optimizer(self.cfg.optimizer, self.model, self.cfg.lr)
        self.loss = self._loss()
        self._loss_accum = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._partials = torch.zeros(engine.LOSS_PARTIALS, dtype=torch.float32, device=self.device)
        self._step = None                                  # the CDAEStep, kept across epochs

    def _loss(self):
        # reference cdae_trainer.py:26-33
        if self.cfg.loss_name.lower() == 'bce' and self.cfg.negative_sampling:
            return NSBCELoss()
        elif self.cfg.loss_name.lower() == 'bce' and not self.cfg.negative_sampling:
            return BCELoss()
        else:
            logger.error(f"Loss Not Exists: {self.cfg.loss_name} when negative_sampling == {self.cfg.negative_sampling}")
            raise NotImplementedError(f"Loss Not Exists: {self.cfg.loss_name}")

    def _accumulate(self, loss):
        self._partials[:1].copy_(loss.detach().reshape(1))
        engine.loss_finalize(self._partials, 1.0, None, self._loss_accum)

    def _fused_step(self):
        """The CDAEStep bound to the model's parameters and the optimizer's Adam state (cdae_step.py), or None
        when the optimizer is not Adam / AdamW or ``cfg.fused_step`` is off — training then goes through the
        model's autograd node, the loss module and optimizer.step(), launch by launch."""
        from .. import optim
        if not self.cfg.get("fused_step", True) or not isinstance(self.optimizer, optim.Adam):
            return None
        if self.model.hidden_size % 4:                     # 16-byte rows for the matrix-core operands
            return None
        if self._step is None or not self._step.bound_to(self.model, self.optimizer):
            self._step = CDAEStep(self.model, self.optimizer, self.cfg.negative_sampling,
                                  decoder=self.cfg.get("train_decoder", "auto"),
                                  transposed_wh=self.cfg.get("transposed_wh", True))
        return self._step

    def train(self, train_dataloader) -> float:
        # reference cdae_trainer.py:36-54
        self.model.train()
        step = self._fused_step()
        if step is not None:
            step.loss_accum.zero_()
            try:
                self._train_fused(step, train_dataloader)
            finally:
                step.release()                             # W_h and its moments back into the module / optimizer
            step.check()
            return step.epoch_loss()
        self._loss_accum.zero_()
        for data in train_dataloader:
            user_id, input_mask = data['user_id'].to(self.device), data['input_mask'].to(self.device)
            pred = self.model(user_id, input_mask)
            self.optimizer.zero_grad()
            if self.cfg.negative_sampling:
                negative_mask = data['negative_mask'].to(self.device)
                loss = self.loss(pred, input_mask, negative_mask)
            else:
                loss = self.loss(pred, input_mask)
            loss.backward()
            self.optimizer.step()
            self._accumulate(loss)
        self.model.check_indices()
        return float(self._loss_accum.item())

    def _train_fused(self, step, train_dataloader):
        model = self.model
        own_noise = "add_noise" not in model.__dict__ and type(model).add_noise is CDAE.add_noise
        for data in train_dataloader:
            if 'lists' in data:                            # data/cdae_batches.py CDAEBatchLoader(lists=True)
                step.step_lists(data['user_id'].to(self.device), data['lists'])
                continue
            user_id, input_mask = data['user_id'].to(self.device), data['input_mask'].to(self.device)
            negative_mask = data['negative_mask'].to(self.device) if self.cfg.negative_sampling else None
            if own_noise:
                # the seed draw of CDAE.forward (same position in torch's generator stream)
                p = model.corruption_level
                seed = int(torch.randint(0, 1 << 62, (1,)).item()) if p > 0 else 0
                step.step(user_id, input_mask, negative_mask, seed=seed, p=p)
            else:                                           # an overridden add_noise (tests replay recorded masks)
                step.step(user_id, input_mask, negative_mask, x_in=model.add_noise(input_mask))

    def _scored_by_lists(self, dataloader, with_loss):
        """validate / evaluate over list batches (CDAEBatchLoader(lists=True)): per batch only the encoder runs
        (and, in validation, the NS-BCE terms on the loss positions: yr_cdae_sampled_decode without gradients);
        the hidden rows of ALL users are then scored against the whole catalogue in one launch of the fused
        evaluation kernel — z . W_o[i] + b_o[i] on the matrix cores with the seen-item mask and the top-k in the
        epilogue (yr_mf_eval_topk_bias), no [B, I] prediction — and the metrics in one more.  The output
        activation is monotone, so the top-k of the pre-activations is the reference's top-k of ``pred``; its
        multiply-mask (seen items -> 0, cdae_trainer.py:132) is below every sigmoid output, i.e. -FLT_MAX before
        the sigmoid (with the identity activation: the value 0 itself).
        Returns the six metric sums of _metric_sums over the whole user set."""
        model, dev = self.model, self.device
        if with_loss and not self.cfg.negative_sampling:
            raise NotImplementedError("list batches carry the NS-BCE positions; plain BCE needs dense batches")
        Wh, bh, V, Wo, bo = (q.data for q in model._params())
        # the encoder gathers one column of W_h per input item: from a transposed copy ([I, H], made once per pass:
        # one 19.5 MB copy) that is one contiguous 4 H-byte row instead of H scattered cache lines — the encoder was
        # the largest kernel of a validation pass after the scoring launch (1.7 ms of 5.4)
        WhT = Wh.t().contiguous() if model.hidden_size % 4 == 0 else None
        Z = torch.zeros(model.num_users, model.hidden_size, dtype=torch.float32, device=dev)   # rows by user id
        covered, item_lists = 0, None
        stats = torch.zeros(2, dtype=torch.float32, device=dev)
        count = torch.zeros(engine.COUNT_WORDS, dtype=torch.int32, device=dev)
        partials = None
        # several batches per launch when the loader can make them (CDAEBatchLoader.super_batches: the same lists, the
        # same loss per batch; ~10 engine calls per GROUP of batches instead of per batch — the per-batch loop was
        # bound by its host calls: 72 us per 256-row batch for 42 us of kernels)
        # (32 batches of 256 rows per launch: 3.2 ms per validation pass at Yelp2018 size against 3.4 at 16 and 3.0 with
        # the whole pass in one group; the list buffers are sized for full rows: 0.6 MB per row, 5 GB at 32 x 256)
        group = int(self.cfg.get("eval_batch_group", 32))
        grouped = group > 1 and hasattr(dataloader, "super_batches")
        means = arrive = None
        for data in (dataloader.super_batches(group) if grouped else dataloader):
            users, lists = data['user_id'].to(dev).contiguous(), data['lists'].alive()
            z = (engine.cdae_sparse_encode(lists.rows, WhT, bh, V, users, model._hidden_act, err_flag=model._flag(),
                                           transposed=True) if WhT is not None else
                 engine.cdae_sparse_encode(lists.rows, Wh, bh, V, users, model._hidden_act, err_flag=model._flag()))
            Z.index_copy_(0, users, z)
            covered += users.numel()
            item_lists = data['item_lists']
            if with_loss and grouped:
                splits = engine.cdae_sampled_decode_splits(users.numel())
                n = users.numel() * splits
                if partials is None or partials.numel() < n:
                    partials = torch.empty(n, dtype=torch.float32, device=dev)
                if means is None:
                    means = torch.zeros(max(group, 1), dtype=torch.float32, device=dev)
                    arrive = torch.zeros(1, dtype=torch.int32, device=dev)
                count.zero_()
                engine.cdae_sampled_decode(lists.loss, z, Wo, bo, model._output_act, None, None, None, partials, count)
                engine.cdae_loss_finalize_batched(partials, splits, lists.loss[2], users.numel(), data['batch_rows'],
                                                  means, arrive, self._loss_accum)
            elif with_loss:
                n = users.numel() * engine.cdae_sampled_decode_splits(users.numel())
                if partials is None or partials.numel() < n:
                    partials = torch.empty(n, dtype=torch.float32, device=dev)
                count.zero_()
                engine.cdae_sampled_decode(lists.loss, z, Wo, bo, model._output_act, None, None, None, partials, count)
                engine.cdae_loss_finalize(partials, n, count, stats, self._loss_accum)
        if covered != model.num_users or item_lists is None:
            raise NotImplementedError("list batches must cover every user exactly once (CDAEBatchLoader does)")
        (sp, si), (ap, ai) = item_lists["seen"], item_lists["actual"]
        mask_value = engine.MASK_VALUE if model._output_act == engine.ACT_SIGMOID else 0.0
        # the previous lists of the same pass (validation / test) are the hint of this one (engine.mf_eval_topk)
        hints = self.__dict__.setdefault("_eval_hints", {}) if self.cfg.get("eval_hints", True) else None
        hint = hints.get(with_loss) if hints is not None else None
        if hint is not None and tuple(hint.shape) != (model.num_users, self.cfg.top_n):
            hint = None
        top = engine.mf_eval_topk(Z, Wo, torch.arange(model.num_users, device=dev), sp, si, self.cfg.top_n,
                                  mask_value=mask_value, item_bias=bo,
                                  precision=self.cfg.get("eval_precision", "bf16x3"), hint=hint)
        if hints is not None:
            hints[with_loss] = top
        model.check_indices()
        return engine.rank_metrics(top, ap, ai)[4:10]

    def validate(self, valid_dataloader):
        # reference cdae_trainer.py:56-88
        self.model.eval()
        self._loss_accum.zero_()
        if getattr(valid_dataloader, "lists", False):
            with torch.no_grad():
                sums = self._scored_by_lists(valid_dataloader, with_loss=True)
            p, r, m, n = self._metrics(False, None, None, sums)
            return (float(self._loss_accum.item()), p, r, m, n)
        actual, predicted = [], []
        host = self.cfg.get("host_metrics", False)
        sums = torch.zeros(6, dtype=torch.float64, device=self.device)
        with torch.no_grad():
            for data in valid_dataloader:
                user_id, input_mask = data['user_id'].to(self.device), data['input_mask'].to(self.device)
                valid_mask = data['valid_mask'].to(self.device)
                pred = self.model(user_id, input_mask)
                target = input_mask.clone()                       # input_mask.add(valid_mask), cdae_trainer.py:67
                engine.sgd_dense(target, valid_mask.contiguous(), -1.0)
                if self.cfg.negative_sampling:
                    loss = self.loss(pred, target, data['negative_mask'].to(self.device))
                else:
                    loss = self.loss(pred, target)
                self._accumulate(loss)
                if host:
                    batch_actual, batch_predicted = self._generate_target_and_top_k_recommendation(pred, valid_mask, input_mask)
                    actual.extend(batch_actual)
                    predicted.extend(batch_predicted)
                else:
                    sums += self._metric_sums(pred, valid_mask, input_mask, data.get('item_lists'))
        p, r, m, n = self._metrics(host, actual, predicted, sums)
        return (float(self._loss_accum.item()), p, r, m, n)

    @log_metric
    def evaluate(self, test_dataloader):
        # reference cdae_trainer.py:90-121
        self.model.eval()
        if getattr(test_dataloader, "lists", False):
            with torch.no_grad():
                sums = self._scored_by_lists(test_dataloader, with_loss=False)
            p, r, m, n = self._metrics(False, None, None, sums)
            logger.info(f"[Trainer] Test > precision@{self.cfg.top_n} : {p:.4f} / Recall@{self.cfg.top_n}: {r:.4f} / "
                        f"MAP@{self.cfg.top_n}: {m:.4f} / NDCG@{self.cfg.top_n}: {n:.4f}")
            return (p, r, m, n)
        actual, predicted = [], []
        host = self.cfg.get("host_metrics", False)
        sums = torch.zeros(6, dtype=torch.float64, device=self.device)
        with torch.no_grad():
            for data in test_dataloader:
                input_mask, user_id, test_mask = data['input_mask'].to(self.device), \
                    data['user_id'].to(self.device), data['test_mask'].to(self.device)
                pred = self.model(user_id, input_mask)
                if host:
                    batch_actual, batch_predicted = self._generate_target_and_top_k_recommendation(pred, test_mask, input_mask)
                    actual.extend(batch_actual)
                    predicted.extend(batch_predicted)
                else:
                    sums += self._metric_sums(pred, test_mask, input_mask, data.get('item_lists'))
        p, r, m, n = self._metrics(host, actual, predicted, sums)
        logger.info(f"[Trainer] Test > precision@{self.cfg.top_n} : {p:.4f} / Recall@{self.cfg.top_n}: {r:.4f} / "
                    f"MAP@{self.cfg.top_n}: {m:.4f} / NDCG@{self.cfg.top_n}: {n:.4f}")
        return (p, r, m, n)

    def _metrics(self, host, actual, predicted, sums):
        if host:
            return ranking_metrics(actual, np.concatenate(predicted, axis=0).tolist(), self.cfg.top_n)
        cnt, ps, rs, ms, ns, total = sums.tolist()                 # the one read-back
        return (ps / total, rs / cnt, ms / cnt, ns / cnt)

    @staticmethod
    def _rows_to_csr(mask):
        """Non-zero column ids of every row of a [B, I] mask as CSR (ascending inside a row, the
        order np.nonzero gives the reference at cdae_trainer.py:125)."""
        nz = mask.nonzero()
        ptr = torch.zeros(mask.shape[0] + 1, dtype=torch.int64, device=mask.device)
        ptr[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=mask.shape[0]), 0)
        return ptr, nz[:, 1].contiguous()

    def _metric_sums(self, pred, actual_mask, pred_mask, item_lists=None):
        """[users with held-out items, precision / recall / AP / NDCG sums, users] of one batch.
        ``item_lists`` (from data/cdae_batches.py): the batch's users plus the per-user CSR of their
        seen and held-out items already on the device — the kernels index it by user, nothing is
        derived from the dense masks."""
        if item_lists is not None:
            users = item_lists["users"].contiguous()
            (sp, si), (ap, ai) = item_lists["seen"], item_lists["actual"]
            top = engine.topk_masked(pred.detach().contiguous(), sp, si, self.cfg.top_n, mask_value=0.0, mask_rows=users)
            return engine.rank_metrics(top, ap, ai, pos_rows=users)[4:10]
        mask_ptr, mask_idx = self._rows_to_csr(pred_mask)
        top = engine.topk_masked(pred.detach().contiguous(), mask_ptr, mask_idx, self.cfg.top_n, mask_value=0.0)
        pos_ptr, pos_idx = self._rows_to_csr(actual_mask)
        return engine.rank_metrics(top, pos_ptr, pos_idx)[4:10]

    def _generate_target_and_top_k_recommendation(self, pred, actual_mask, pred_mask):
        # reference cdae_trainer.py:123-144
        actual = [np.nonzero(row)[0] for row in actual_mask.cpu().numpy()]
        # CSR of the seen items per row (index bookkeeping only), masked scores become 0 in the kernel
        nz = pred_mask.nonzero()
        counts = torch.bincount(nz[:, 0], minlength=pred_mask.shape[0])
        ptr = torch.zeros(pred_mask.shape[0] + 1, dtype=torch.int64, device=pred.device)
        ptr[1:] = torch.cumsum(counts, 0)
        top = engine.topk_masked(pred.detach().contiguous(), ptr, nz[:, 1].contiguous(), self.cfg.top_n, mask_value=0.0)
        return actual, [top.cpu().numpy()]
