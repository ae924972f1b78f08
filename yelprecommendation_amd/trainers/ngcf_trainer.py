"""NGCF trainer — drop-in for reference trainers/ngcf_trainer.py:22-182.

Same constructor ``NGCFTrainer(cfg, num_items, num_users, laplacian_matrix)`` and the same
``run / train / validate / evaluate`` contract.  ``evaluate`` scores
``np.random.randint(len(eval_data), size=100)`` eval rows — WITH replacement, drawn from the global
NumPy RNG exactly like the reference (ngcf_trainer.py:140) — but propagates the graph once per
call instead of once per sampled user.  With Adam / AdamW ``train`` runs every batch as ONE engine call
(ngcf_step.NGCFStep: the launches of the autograd route, issued from C; cfg.fused_step=False keeps the reference's
loop shape — bpr_forward, zero_grad, loss, backward, step — on the HIP autograd ops).
"""
import numpy as np
import torch

from .. import engine
from ..loss import BPRLoss
from ..metric import ranking_metrics
from ..models.ngcf import NGCF, _iadd
from ..utils import logger
from .base_trainer import BaseTrainer
from .mf_trainer import _lists_to_csr


class NGCFTrainer(BaseTrainer):
    def __init__(self, cfg, num_items: int, num_users: int, laplacian_matrix) -> None:
        super().__init__(cfg)
        logger.info(f'[DEVICE] device = {self.device}')
        self.num_items = num_items
        self.num_users = num_users
        self.model = NGCF(self.cfg, num_users, num_items).to(self.device)
        self.optimizer = self._optimizer(self.cfg.optimizer, self.model, self.cfg.lr, self.cfg.weight_decay)
        self.loss = self._loss()
        self.laplacian_matrix = laplacian_matrix
        self._loss_accum = torch.zeros(1, dtype=torch.float64, device=self.device)

    def _loss(self):
        return BPRLoss()

    def run(self, train_dataloader, valid_dataloader, valid_eval_data):
        # reference ngcf_trainer.py:36-99
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
                torch.save(self.model.state_dict(), f'{self.cfg.model_dir}/best_model.pt')
            else:
                endurance += 1
                if endurance > self.cfg.patience:
                    logger.info("[Trainer] ealry stopping...")
                    break

    def _batch(self, data):
        dev = self.device
        return (data['user_id'].to(dev, non_blocking=True), data['pos_item'].to(dev, non_blocking=True),
                data['neg_item'].to(dev, non_blocking=True))

    def _accumulate(self, loss):
        # train_loss += loss.item() of ngcf_trainer.py:116 without the per-step host sync
        engine.loss_finalize(self._one_partial(loss), 1.0, None, self._loss_accum)

    def _one_partial(self, loss):
        p = getattr(self, "_partials", None)
        if p is None:
            p = self._partials = torch.zeros(engine.LOSS_PARTIALS, dtype=torch.float32, device=self.device)
        p[:1].copy_(loss.detach().reshape(1))
        return p

    def _fused_step(self):
        """The whole batch step as one engine call (ngcf_step.NGCFStep) for Adam / AdamW unless
        cfg.fused_step is False; None -> the launch-by-launch autograd route below (SGD, or on request)."""
        from .. import optim
        from ..ngcf_step import NGCFStep
        if not isinstance(self.optimizer, optim.Adam) or not self.cfg.get("fused_step", True):
            return None
        graph = self.model.graph(self.laplacian_matrix)
        st = getattr(self, "_step", None)
        if st is None or not st.bound_to(self.model, self.optimizer, graph):
            st = self._step = NGCFStep(self.model, self.optimizer, graph, self.model._subset_fraction())
        return st

    def train(self, train_dataloader) -> float:
        # reference ngcf_trainer.py:102-117
        self.model.train()
        step = self._fused_step()
        if step is not None:
            step.loss_accum.zero_()
            for data in train_dataloader:
                step.step(*self._batch(data))
            step.check()
            return step.epoch_loss()
        self._loss_accum.zero_()
        for data in train_dataloader:
            user_id, pos_item, neg_item = self._batch(data)
            pos_pred, neg_pred = self.model.bpr_forward(user_id, pos_item, neg_item, self.laplacian_matrix)
            self.optimizer.zero_grad()
            loss = self.loss(pos_pred, neg_pred)
            loss.backward()
            self.optimizer.step()
            self._accumulate(loss)
        self.model.check_indices()
        return float(self._loss_accum.item())

    def validate(self, valid_dataloader) -> float:
        # reference ngcf_trainer.py:119-132 (the reference keeps autograd on here; the values are the same)
        self.model.eval()
        self._loss_accum.zero_()
        with torch.no_grad():
            # the parameters do not change during validation, so the K propagation layers are the same for every
            # batch: computed once (the reference re-propagates the whole graph per batch, ngcf_trainer.py:124), then
            # every batch is one scoring launch on them — the same kernels on the same data, bit-identical values
            once = self.cfg.get("propagate_once", True)
            layers = self.model.propagate(self.laplacian_matrix) if once else None
            for data in valid_dataloader:
                user_id, pos_item, neg_item = self._batch(data)
                if once:
                    pos_pred, neg_pred = engine.ngcf_score(layers, self.num_users, user_id.contiguous(),
                                                           pos_item.contiguous(), neg_item.contiguous(),
                                                           err_flag=self.model._flag())
                else:
                    pos_pred, neg_pred = self.model.bpr_forward(user_id, pos_item, neg_item, self.laplacian_matrix)
                self._accumulate(self.loss(pos_pred, neg_pred))
        self.model.check_indices()
        return float(self._loss_accum.item())

    def recommend(self, users, mask_ptr, mask_idx):
        """Top-``top_n`` unmasked items for the given user ids ([n, top_n] int64 on the device):
        one propagation, then S = sum_k E_k[users] E_k[items]^T on the matrix cores, mask, top-k."""
        layers = self.model.propagate(self.laplacian_matrix)
        nu = self.num_users
        scores = None
        for E in layers:
            s = engine.mf_scores_gemm(E[:nu], E[nu:], users)
            scores = s if scores is None else _iadd(scores, s)
        return engine.topk_masked(scores, mask_ptr, mask_idx, self.cfg.top_n)

    def evaluate(self, eval_data, mode='valid') -> tuple:
        # reference ngcf_trainer.py:134-165
        self.model.eval()
        positions = np.random.randint(eval_data.shape[0], size=100)            # :140, global NumPy RNG
        users = np.asarray(eval_data.index.values, dtype=np.int64)[positions]
        rows = eval_data.iloc[positions]
        actual = [list(x) for x in rows['pos_items']]
        mask_ptr, mask_idx = _lists_to_csr([list(x) for x in rows['mask_items']])
        dev = self.device
        predicted = self.recommend(torch.from_numpy(users).to(dev), torch.from_numpy(mask_ptr).to(dev),
                                   torch.from_numpy(mask_idx).to(dev)).cpu().numpy()
        p, r, m, n = ranking_metrics(actual, predicted.tolist(), self.cfg.top_n)
        if mode == 'test':
            logger.info(f"[Trainer] Test > precision@{self.cfg.top_n} : {p:.4f} / Recall@{self.cfg.top_n}: {r:.4f} / "
                        f"MAP@{self.cfg.top_n}: {m:.4f} / NDCG@{self.cfg.top_n}: {n:.4f}")
        return (p, r, m, n)

    def _generate_top_k_recommendation(self, pred, mask_items):
        # reference ngcf_trainer.py:167-182 for one user's score vector
        dev = pred.device
        mask = torch.as_tensor(np.asarray(mask_items, dtype=np.int64), device=dev)
        ptr = torch.tensor([0, mask.numel()], dtype=torch.int64, device=dev)
        return engine.topk_masked(pred.detach().reshape(1, -1).contiguous(), ptr, mask, self.cfg.top_n)[0].cpu().numpy()
