"""One BPR-MF optimisation step as a short, fixed sequence of HIP launches.

This is the engine object behind ``MFTrainer.train`` and ``bench.py``: it owns the dense
gradient buffers and the Adam state next to the two tables and enqueues, per batch,
  fused gather + BPR score + loss + scatter-add gradient      (csrc/bpr_mf.hip)
  [N > 1 GPUs: RCCL all-reduce(SUM) of the item gradient, overlapped with the user update]
  dense Adam on the user rows, dense Adam on the item rows (clearing the gradients)
equal — to float rounding — to the reference's per-batch sequence
(trainers/mf_trainer.py:104-112 with torch.optim.Adam, base_trainer.py:34-36).
The running loss stays on the device.

Two implementations of the same step:
  impl="pull"   (default) csrc/bpr_pull.hip, three launches: tile partition of the batch by
                destination bucket, then one owner pass per table in which each row pulls its
                contributions into registers and applies Adam — no float atomics, no global integer
                atomics, no gradient buffers; the loss is reduced by the last launch.  The user
                table is double-buffered; ``self.U`` is always the current one.
  impl="atomic" csrc/bpr_mf.hip + csrc/optim.hip, two launches: scatter-add with float atomics into
                dense gradient buffers (marking the rows it touches), then ONE dense Adam pass over both
                tables that also reduces the loss.
"""
import torch

from . import engine
from .user_shard import ItemSlices, reduce_scatter_item_exchange, sharded_item_exchange


# impl="auto": the pull form from this many triplets per rank and step upwards, the two-launch atomic
# form below (measured on MI355X at Yelp2018 shape, atomic vs pull: 24 vs 37 us at 4,096, 35 vs 38 us at
# 16,384, 40.9 vs 40.1 us at 24,576, 45 vs 42 us at 32,768, 68 vs 47 us at 65,536); the choice is made from
# the GLOBAL batch so that every rank of a sharded run takes the same form (the two forms issue different
# collectives)
AUTO_PULL_MIN_BATCH = 24576

IMPL_NAMES = {
    "auto": "auto: atomic scatter + one Adam launch below %d triplets per rank and step, pull above" % AUTO_PULL_MIN_BATCH,
    "pull": "pull: tile partition + owner pass per table (gather/score/loss/grad/Adam fused, no atomics)",
    "atomic": "atomic: fused gather/score/loss + float-atomic scatter-add, dense Adam",
}


class BPRMFStep:
    def __init__(self, U, I, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, optimizer="adam",
                 world_size=1, process_group=None, time_kernels=False, impl="auto", max_batch=0,
                 state=None, split_item_update=False, item_chunks=1, item_exchange="all_reduce", rank=0,
                 deterministic=False):
        """``state``: optional dict with pre-existing Adam tensors ``mU, vU, mI, vI`` (shared, updated
        in place) and the step count ``t`` — lets a trainer keep its torch-style optimizer state
        in sync with the fused step (see MFTrainer).
        ``split_item_update``: take the multi-GPU shape of the step (item pass emits the dense item
        gradient, then a separate dense Adam launch) even with one rank — used by tests."""
        if item_exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError("item_exchange must be 'all_reduce' or 'reduce_scatter'")
        # N > 1 only.  "all_reduce": the item gradient is summed everywhere and every rank applies the same
        # dense Adam to its replica (item Adam state replicated).  "reduce_scatter": rank r receives the
        # summed gradient of ITS slice of the item rows, applies Adam to that slice alone (1/N of the
        # update, item Adam state sharded: only the slice's moments are current on a rank) and the
        # updated slices are all-gathered.
        # deterministic: two runs on the same batches give bit-identical tables (always the pull form: the
        # atomic form's float atomics add in arrival order)
        self.deterministic = bool(deterministic)
        self.skip_collective = False       # measurement only (bench.py: exposed collective time): N > 1 steps without the exchange
        self.item_exchange = item_exchange
        self.rank = int(rank)
        self.split_item_update = split_item_update
        # multi-GPU: the item pass / all-reduce / item Adam run in this many chunks of item rows so
        # that chunk c is on the wire while chunk c+1 is computed
        self.item_chunks = max(1, int(item_chunks))
        if optimizer.lower() not in ("adam", "adamw"):
            raise NotImplementedError(f"BPRMFStep: optimizer {optimizer}")
        if impl not in IMPL_NAMES:
            raise ValueError(f"impl must be one of {list(IMPL_NAMES)}")
        self.impl_key, self.impl = impl, IMPL_NAMES[impl]
        self.launches = ""
        self.U, self.I = U, I
        self._U_alt = None
        self._ws_slots, self._ws_batch, self._indexed = [None, None], [0, 0], None
        if impl == "pull" and max_batch:
            self._workspace(max_batch, 0)
        self._gI_dirty = False
        self._touched = None
        self._item_order, self.auto_item_order = None, True
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.decoupled = optimizer.lower() == "adamw"
        self.world_size, self.pg = world_size, process_group
        dev = U.device
        # buffers are created on first use by the implementation that needs them
        self.gU = None
        self.gI = torch.zeros_like(I) if (world_size > 1 or split_item_update) else None
        self._slices = self._I_pad = None
        if item_exchange == "reduce_scatter" and (world_size > 1 or split_item_update):
            self._slices = ItemSlices(I.shape[0], world_size, self.rank)
            self._gI_pad = torch.zeros(self._slices.padded, I.shape[1], dtype=I.dtype, device=I.device)
            self.gI = self._gI_pad[:I.shape[0]]                # the kernels write the real rows
            self._I_pad = torch.zeros(self._slices.padded, I.shape[1], dtype=I.dtype, device=I.device)
        if state is not None:
            self.mU, self.vU, self.mI, self.vI = state["mU"], state["vU"], state["mI"], state["vI"]
            self.t = int(state.get("t", 0))
        else:
            self.mU, self.vU = torch.zeros_like(U), torch.zeros_like(U)
            self.mI, self.vI = torch.zeros_like(I), torch.zeros_like(I)
            self.t = 0
        self.partials = torch.zeros(engine.LOSS_PARTIALS, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.flag = engine.new_error_flag(dev)
        self.time_kernels = time_kernels
        self._ev = {}
        self._bytes = {}
        # validate the long-lived buffers once and cache their addresses for the lean step path
        from . import _lib
        self._lib = _lib.load()
        f32 = torch.float32
        for name, t in (("U", U), ("I", I), ("mU", self.mU), ("vU", self.vU), ("mI", self.mI), ("vI", self.vI)):
            engine._dev(t, f32, name)
        if U.dim() != 2 or I.dim() != 2 or U.shape[1] != I.shape[1]:
            raise engine.EngineError("tables must be [rows, D] with equal D")
        if self.mU.shape != U.shape or self.vU.shape != U.shape or self.mI.shape != I.shape or self.vI.shape != I.shape:
            raise engine.EngineError("Adam state must match the table shapes")
        self._pI, self._pmU, self._pvU = I.data_ptr(), self.mU.data_ptr(), self.vU.data_ptr()
        self._pmI, self._pvI = self.mI.data_ptr(), self.vI.data_ptr()
        self._ppartials, self._pflag = self.partials.data_ptr(), self.flag.data_ptr()
        # table shapes the pull form does not cover (very many rows) fall back to the atomic form under "auto"
        self._pull_ok = self._lib.yr_bpr_mf_pull_workspace_bytes(1, U.shape[0], I.shape[0], U.shape[1]) > 0 and \
            engine.pull_supported(U.shape[0], I.shape[0], U.shape[1])

    # -- timing of individual kernels with events on the launch stream ------------------------
    def reset_timers(self):
        self._ev = {}

    def _timed(self, name, alg_bytes, record, fn):
        if not (record and self.time_kernels):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self._ev.setdefault(name, []).append((a, b))
        self._bytes[name] = alg_bytes

    def kernel_times(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) * 1e3 / len(v), len(v), self._bytes[k])
                for k, v in self._ev.items()}

    def _workspace(self, batch, slot=0):
        """Scratch for the batch index; two slots so that the index of batch k+1 can be built while
        batch k's all-reduce is in flight."""
        ws = self._ws_slots[slot]
        if ws is None or batch > self._ws_batch[slot]:
            ws = engine.bpr_mf_pull_workspace(batch, self.U.shape[0], self.I.shape[0], self.U.shape[1],
                                              self.U.device)
            self._ws_slots[slot], self._ws_batch[slot] = ws, batch
        return ws

    # -- the step -----------------------------------------------------------------------------
    def step(self, u, p, n, record=False, global_batch=None, next_batch=None):
        """One optimisation step on this rank's triplets.  ``global_batch``: size of the batch
        over ALL ranks (the mean of loss.py:27 is over it); default = local size x world_size,
        i.e. equal slices.  ``next_batch``: optional (u, p, n) of the following step — its index
        (which does not depend on the tables) is then built while this step's all-reduce is in
        flight, and the following ``step`` call finds it ready."""
        if global_batch is None:
            global_batch = u.numel() * self.world_size
        key = self.impl_key
        if key == "auto":
            # decided from quantities that are equal on every rank (the two forms issue different
            # collectives): global batch and the table shapes
            key = "pull" if (global_batch >= AUTO_PULL_MIN_BATCH * self.world_size and self._pull_ok) else "atomic"
            if self._slices is not None or self.deterministic:
                key = "pull"                        # sharded item Adam state / fixed summation order: one form
        if key == "pull":
            if self._U_alt is None:
                self._U_alt = torch.empty_like(self.U)
            self.impl = IMPL_NAMES["pull"]
            self.launches = "tile_partition, owner_pass<user>, owner_pass<item>"
            return self._step_pull(u, p, n, record, global_batch, next_batch)
        if self.gU is None:
            self.gU = torch.zeros_like(self.U)
        if self.gI is None:
            self.gI = torch.zeros_like(self.I)
        self.impl = IMPL_NAMES["atomic"]
        self.launches = "bpr_fwd_bwd, adam_dual"
        return self._step_atomic(u, p, n, record, global_batch)

    def _check_triplets(self, u, p, n):
        B = u.numel()
        for name, t in (("user", u), ("pos", p), ("neg", n)):
            if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == B):
                raise engine.EngineError(f"{name}: need a contiguous int64 GPU tensor of length {B}")
        return B

    def _build_index(self, u, p, n, slot):
        B = self._check_triplets(u, p, n)
        ws = self._workspace(B, slot)
        rc = self._lib.yr_bpr_mf_pull_index(u.data_ptr(), p.data_ptr(), n.data_ptr(), B, self.U.shape[1],
                                            self.U.shape[0], self.I.shape[0], ws.data_ptr(), ws.numel(),
                                            self._pflag, engine._stream())
        if rc:
            engine.check(rc, "yr_bpr_mf_pull_index")
        # keep the tensors alive (and identify the batch) until the index is consumed
        self._indexed = (slot, (u, p, n))

    def set_item_order(self, item_weight=None):
        """Start order of the item pass's workgroups: buckets of 1024 / D item rows sorted by how many
        occurrences they receive per step, heaviest first (``item_weight``: anything proportional to the items'
        expected occurrences, e.g. the train set's item degrees + the uniform negatives' share; None: back to
        index order).  There are more item buckets than resident workgroups, so the late starters should be the
        light ones; results do not depend on the order."""
        if item_weight is None:
            self._item_order = None
            return
        nb = int(self._lib.yr_bpr_mf_pull_item_buckets(self.I.shape[0], self.I.shape[1]))
        rows = -(-self.I.shape[0] // nb) if nb else 1
        per = 1024 // self.I.shape[1]
        w = torch.zeros(nb * per, dtype=torch.float64, device=self.I.device)
        w[:self.I.shape[0]] = item_weight.to(self.I.device, torch.float64)
        load = w.view(nb, per).sum(1)
        self._item_order = torch.argsort(load, descending=True, stable=True).to(torch.int32).contiguous()

    def _auto_item_order(self, p, n):
        # first pull step: the batch's own occurrence counts stand for the data set's item popularity
        # (ids outside the table are dropped here: the kernels' own range check reports them through the error
        # flag, on every rank together — a torch error raised from bincount would be rank-local)
        rows = self.I.shape[0]
        ids = torch.cat([p, n])
        ids = ids[(ids >= 0) & (ids < rows)]
        self.set_item_order(torch.bincount(ids, minlength=rows)[:rows])

    def _step_pull(self, u, p, n, record, global_batch, next_batch=None):
        B = u.numel()
        D = self.U.shape[1]
        if self._item_order is None and self.auto_item_order and B >= 4 * self.I.shape[0]:
            self._auto_item_order(p, n)
        inv = 1.0 / global_batch if global_batch else 0.0
        self.t += 1
        multi = self.world_size > 1 or self.split_item_update
        nU, nI = self.U.numel(), self.I.numel()
        # algorithmic bytes of the launch group: the per-triplet figure of SURVEY §8d plus the
        # dense Adam pass it absorbs (read p,m,v + write p,m,v on every row of both tables)
        alg = B * (24 + 24 * D) + 6 * 4 * (nU + (0 if multi else nI))
        ready = self._indexed is not None and all(a is b for a, b in zip(self._indexed[1], (u, p, n)))
        slot = self._indexed[0] if ready else 0

        nchunks = self.item_chunks if multi else 1
        rows = self.I.shape[0]
        gran = engine.pull_bucket_rows(D)           # item chunks are whole owner buckets
        bounds = [min(rows, (rows * c // nchunks + gran - 1) // gran * gran) for c in range(nchunks)] + [rows]
        both = engine.PULL_USER_PHASE | engine.PULL_ITEM_PHASE
        ploss, paccum = self.loss.data_ptr(), self.loss_accum.data_ptr()

        def apply(phases, r0, r1, with_loss=None):
            ws = self._ws_slots[slot]
            if with_loss is None:
                with_loss = bool(phases & engine.PULL_USER_PHASE)   # that call's last launch reduces the loss
            step_size, bc2_sqrt = engine.adam_scalars(self.t, self.lr, self.betas[0], self.betas[1])
            order = self._item_order
            rc = self._lib.yr_bpr_mf_pull_apply_ordered(
                self.U.data_ptr(), self._U_alt.data_ptr(), self._pI, self._pmU, self._pvU, self._pmI, self._pvI,
                self.gI.data_ptr() if multi else None, B, D, self.U.shape[0], rows, inv, self.lr,
                step_size, bc2_sqrt, self.betas[0], self.betas[1], self.eps, self.wd,
                engine.OPT_ADAMW if self.decoupled else engine.OPT_ADAM, 1 if self.deterministic else 0,
                ws.data_ptr(), ws.numel(), self._ppartials, ploss if with_loss else None,
                paccum if with_loss else None, phases, r0, r1, order.data_ptr() if order is not None else None,
                engine._stream())
            if rc:
                engine.check(rc, "yr_bpr_mf_pull_apply_ordered")

        def first_chunk():
            # lean host path (these calls are the whole step): index tensors are checked here, the
            # long-lived buffers were checked when they were created
            if not ready:
                self._build_index(u, p, n, slot)
            apply(both, bounds[0], bounds[1])
            if nchunks == 1:
                self._indexed = None

        def local_first():
            if record and self.time_kernels and not multi and self.t % 2 == 0:
                # (recorded steps alternate: odd ones bracket the whole launch group with ONE pair of events — its
                # duration carries two launch gaps instead of three event pairs — even ones, here, each launch)
                # the same three launches, each between its own pair of events; algorithmic bytes per
                # SURVEY 8d split by what each launch must move at least once (ids, partner rows,
                # gradient rows folded into the Adam pass; the dense Adam bytes listed apart)
                self._timed("tile_partition", B * 24, record,
                            lambda: None if ready else self._build_index(u, p, n, slot))
                self._timed("owner_pass_user", B * (16 * D), record,   # 3 rows read + the user gradient row
                            lambda: apply(engine.PULL_USER_PHASE, 0, 0, with_loss=False))
                self._timed("owner_pass_item", B * (8 * D), record,    # the two item gradient rows
                            lambda: apply(engine.PULL_ITEM_PHASE, bounds[0], bounds[1], with_loss=True))
                self._indexed = None
                return
            self._timed("bpr_pull_step", alg, record, first_chunk)

        def make_chunk(c):
            def run():
                apply(engine.PULL_ITEM_PHASE, bounds[c], bounds[c + 1])
                if c == nchunks - 1:
                    self._indexed = None
            return run

        def overlap():
            if next_batch is not None:
                self._build_index(*next_batch, 1 - slot)

        def make_update(c):
            def run():
                if multi:
                    r0, r1 = bounds[c], bounds[c + 1]
                    self._timed("adam_dense_item", 7 * 4 * (r1 - r0) * D, record, lambda: engine.adam_dense(
                        self.I[r0:r1], self.gI[r0:r1], self.mI[r0:r1], self.vI[r0:r1], self.t, self.lr,
                        self.betas[0], self.betas[1], self.eps, self.wd, decoupled=self.decoupled, zero_grad=False))
            return run

        if multi and self._slices is not None:
            sl = self._slices

            def whole_item_pass():
                local_first()
                for c in range(1, nchunks):
                    make_chunk(c)()

            def slice_update():
                lo, hi, at = sl.lo, sl.hi, sl.rank * sl.per
                if hi > lo:
                    self._timed("adam_dense_item_slice", 7 * 4 * (hi - lo) * D, record, lambda: engine.adam_dense(
                        self.I[lo:hi], self._gI_pad[at:at + hi - lo], self.mI[lo:hi], self.vI[lo:hi], self.t, self.lr,
                        self.betas[0], self.betas[1], self.eps, self.wd, decoupled=self.decoupled, zero_grad=False))
                    self._I_pad[at:at + hi - lo].copy_(self.I[lo:hi])

            reduce_scatter_item_exchange(whole_item_pass, slice_update, self._gI_pad, self._I_pad, sl, self.pg, overlap,
                                         collective=not self.skip_collective)
            if self.world_size > 1 and not self.skip_collective:
                self.I.copy_(self._I_pad[:rows])
            self._gI_dirty = True
            self.U, self._U_alt = self._U_alt, self.U
            return
        steps = [local_first] + [make_chunk(c) for c in range(1, nchunks)]
        grads = [self.gI[bounds[c]:bounds[c + 1]] for c in range(nchunks)] if multi else [None]
        sharded_item_exchange(steps, [make_update(c) for c in range(nchunks)], grads, self.pg, self.world_size,
                              overlap, collective=not self.skip_collective)
        self._gI_dirty = multi
        self.U, self._U_alt = self._U_alt, self.U

    def _step_atomic(self, u, p, n, record, global_batch):
        B = self._check_triplets(u, p, n)
        D = self.U.shape[1]
        inv = 1.0 / global_batch if global_batch else 0.0
        lib, stream = self._lib, engine._stream()
        if self._gI_dirty:                      # a pull-form multi-GPU step left the reduced gradient there
            self.gI.zero_()
            self._gI_dirty = False
        nU, nI = self.U.shape[0], self.I.shape[0]
        pU, pgU, pgI = self.U.data_ptr(), self.gU.data_ptr(), self.gI.data_ptr()
        self.t += 1
        step_size, bc2_sqrt = engine.adam_scalars(self.t, self.lr, self.betas[0], self.betas[1])
        mode = engine.OPT_ADAMW if self.decoupled else engine.OPT_ADAM
        ploss, paccum = self.loss.data_ptr(), self.loss_accum.data_ptr()
        if self.world_size == 1:
            # two launches: scatter (marks the rows it touches) + Adam over both tables with the loss reduction
            if self._touched is None:
                self._touched = torch.zeros(nU + nI, dtype=torch.uint8, device=self.U.device)

            def whole():
                rc = lib.yr_bpr_mf_scatter_step(pU, self._pI, pgU, pgI, self._pmU, self._pvU, self._pmI, self._pvI,
                                                self._touched.data_ptr(), u.data_ptr(), p.data_ptr(), n.data_ptr(),
                                                B, D, nU, nI, inv, self.lr, step_size, bc2_sqrt, self.betas[0],
                                                self.betas[1], self.eps, self.wd, mode, self._ppartials, ploss,
                                                paccum, self._pflag, stream)
                if rc:
                    engine.check(rc, "yr_bpr_mf_scatter_step")

            self._timed("bpr_scatter_step", B * (24 + 24 * D) + 6 * 4 * (nU + nI) * D, record, whole)
            return

        def fwd_bwd():
            rc = lib.yr_bpr_mf_fwd_bwd(pU, self._pI, u.data_ptr(), p.data_ptr(), n.data_ptr(), B, D, nU, nI, inv,
                                       pgU, pgI, self._ppartials, self._pflag, stream)
            if rc:
                engine.check(rc, "yr_bpr_mf_fwd_bwd")

        self._timed("bpr_fwd_bwd", B * (24 + 24 * D), record, fwd_bwd)
        if not self.skip_collective:
            import torch.distributed as dist
            dist.all_reduce(self.gI, op=dist.ReduceOp.SUM, group=self.pg)

        def adam():
            rc = lib.yr_adam_dense_dual(pU, pgU, self._pmU, self._pvU, nU * D, self._pI, pgI, self._pmI, self._pvI,
                                        nI * D, D, None, None, self.lr, step_size, bc2_sqrt, self.betas[0],
                                        self.betas[1], self.eps, self.wd, mode, self._ppartials, inv, ploss, paccum,
                                        stream)
            if rc:
                engine.check(rc, "yr_adam_dense_dual")

        self._timed("adam_dense_dual", 8 * 4 * (nU + nI) * D, record, adam)

    def epoch_loss(self, reset=True):
        """Sum of per-batch mean losses since the last reset (one host sync).  At N > 1 the
        per-rank partial means (already scaled by 1/B_global) are summed across ranks."""
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.loss_accum, op=dist.ReduceOp.SUM, group=self.pg)
        v = float(self.loss_accum.item())
        if reset:
            self.loss_accum.zero_()
        return v

    def check(self):
        engine.raise_on_flag(self.flag, "BPRMFStep")
