#!/usr/bin/env python3
"""Benchmark of the BPR-MF hot path on MI355X (BASELINE.json metric:
"BPR triplets/sec @ dim64 (1/2/4/8 GPU)").

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic Yelp2018-shaped triplets:
gather + BPR scoring + loss + gradient of both embedding tables + dense Adam on every row
(reference trainers/mf_trainer.py:104-112), inputs already resident in HBM.  At N > 1 the
interaction matrix is sharded by user (one rank per GPU, its users' rows + Adam state local,
the item table replicated) with one RCCL all-reduce on the item-embedding gradient per step;
per-GPU batch is fixed ("weak" scaling, default; `--scaling strong` fixes the global batch) and
`value` is the whole-job triplets/s.  The batch never exceeds one epoch of train rows.

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     the step's algorithmic bytes (1,560 B per triplet) over the summed average
                  durations of its launches vs the 8 TB/s HBM peak, every launch between its own pair of
                  HIP events inside the timed region; "kernels" / "dominant_kernel" = the per-launch split,
                  "algorithmic_GBps_with_adam_bytes" = the rate with the dense-Adam bytes the step also moves (a rate, not a
                  fraction: the working set is cache resident, so an algorithmic rate can exceed the HBM peak);
                  "batch_sweep" = the same at 32 / 4,096 / 65,536 / 262,144 / one epoch;
  "host_resident_batches": the PCIe-inclusive rate (batches start in pinned host memory; never `value`);
  "cpu_baseline": the reference's CPU op sequence (oracle/mf_torch_cpu.py) timed on this
                  host on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DIM = 64
DEFAULT_BATCH = 1 << 19        # triplets per GPU per step: two steps per epoch of the ~0.92 M train rows


def algorithmic_bytes_per_triplet(dim):
    """3 int64 ids + 3 gathered rows + 3 gradient rows, counted once."""
    return 24 + 24 * dim


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="triplets per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-baseline work")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep (N=1 only; < 1 s of GPU time)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --batch triplets per GPU; strong = --batch triplets in all (each rank keeps "
                         "the triplets of the users it owns)")
    ap.add_argument("--workload", default="bpr", choices=["bpr", "eval", "ngcf", "cdae"],
                    help="bpr (default, the BASELINE metric) or one of the other full-size paths (N=1 only): "
                         "eval = fused scoring + mask + top-10 + metrics, ngcf = configs[3] step, cdae = configs[4] step")
    return ap.parse_args()


def other_workload(args):
    """Full-size timing of the paths beside the BPR step (BASELINE configs[3], configs[4] and the
    evaluation of configs[1]) with the same harness: one JSON line, not the driver's default."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
    u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    out = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": False, "unit": "ms",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if args.workload == "eval":
        g = torch.Generator(device=dev).manual_seed(7)
        U = torch.randn(NU, DIM, device=dev, generator=g) * 0.1
        I = torch.randn(NI, DIM, device=dev, generator=g) * 0.1
        users = torch.arange(NU, device=dev)
        order = torch.argsort(u * NI + i)
        mask_idx = i[order].contiguous()
        cnt = torch.bincount(u, minlength=NU)
        mask_ptr = torch.zeros(NU + 1, dtype=torch.int64, device=dev); mask_ptr[1:] = torch.cumsum(cnt, 0)
        pos_ptr, pos_idx = mask_ptr, mask_idx                       # any lists do for timing the metric kernels
        dt = timed(lambda: engine.rank_metrics(engine.mf_eval_topk(U, I, users, mask_ptr, mask_idx, 10), pos_ptr, pos_idx))
        dt32 = timed(lambda: engine.rank_metrics(engine.mf_eval_topk(U, I, users, mask_ptr, mask_idx, 10, precision="f32"),
                                                 pos_ptr, pos_idx))
        # every later evaluation of a training run: the previous lists as hints (here: of tables that have moved by
        # 3 % of their spread since — 93 % of the top-10 kept)
        U_old, I_old = U + 0.003 * torch.randn(U.shape, device=dev, generator=g), I + 0.003 * torch.randn(I.shape, device=dev, generator=g)
        hint = engine.mf_eval_topk(U_old, I_old, users, mask_ptr, mask_idx, 10)
        dth = timed(lambda: engine.rank_metrics(engine.mf_eval_topk(U, I, users, mask_ptr, mask_idx, 10, hint=hint), pos_ptr, pos_idx))
        flops = 2.0 * NU * NI * DIM
        # The f32 scores come from six bf16 partial products per product (three-term splits, error below the f32
        # rounding of a product), so the kernel EXECUTES 6 x 2 U I D bf16 flops on v_mfma_f32_32x32x16_bf16:
        # `achieved` / `frac` price those against the dense bf16 peak (a roofline fraction compares what an
        # instruction stream does with the peak of that instruction).  The algorithmic 2 U I D f32 flops over the f32
        # matrix-core peak — what the same scores cost on v_mfma_f32_32x32x2_f32, measured as `f32_instruction` — is
        # a speed-up figure, not a roofline fraction (it exceeds 1 with hint lists), and is kept apart under
        # `algorithmic_vs_f32_peak`.
        BF16_PEAK, F32_PEAK = 2500.0, 157.3

        def ex(t):
            return {"ms": round(t * 1e3, 4), "achieved": round(6 * flops / t / 1e12, 1), "frac": round(6 * flops / t / 1e12 / BF16_PEAK, 4)}
        out.update(metric="full-catalogue evaluation (scores + mask + top-10 + metrics) @ dim64", value=round(dt * 1e3, 4),
                   ms_per_step=round(dt * 1e3, 4), dtype="f32 (bf16x3 split operands, f32 accumulate)",
                   config={"workload": "all 31,668 users x 38,048 items, train-item masks, top-10, 4 metrics"},
                   roofline={"bound": "mfma", "kernel": "prescan + mf_eval_topk_kernel<SPLIT> (+ split_rows + merge + rank_metrics)",
                             "achieved": ex(dt)["achieved"], "peak": BF16_PEAK, "unit": "TFLOP/s (executed bf16, dense peak)",
                             "frac": ex(dt)["frac"], "traffic": None,
                             "with_hint_lists": ex(dth),
                             "f32_instruction": {"ms": round(dt32 * 1e3, 4), "achieved": round(flops / dt32 / 1e12, 1),
                                                 "peak": F32_PEAK, "unit": "TFLOP/s (executed f32)",
                                                 "frac": round(flops / dt32 / 1e12 / F32_PEAK, 4)},
                             "algorithmic_vs_f32_peak": {"note": "2 U I D f32 flops / time / f32 MFMA peak: what the bf16x3 form "
                                                                 "gains over the f32 instruction, NOT a roofline fraction",
                                                         "cold": round(flops / dt / 1e12 / F32_PEAK, 4),
                                                         "with_hint_lists": round(flops / dth / 1e12 / F32_PEAK, 4)}})
    elif args.workload == "ngcf":
        from yelprecommendation_amd.graph import LaplacianCSR
        from yelprecommendation_amd.loss import BPRLoss
        from yelprecommendation_amd.models.ngcf import NGCF
        from yelprecommendation_amd.optim import Adam
        from yelprecommendation_amd.utils import make_config
        r = torch.randint(1, 6, u.shape, device=dev)
        graph = LaplacianCSR.from_interactions(u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), NU, NI, dev)
        cfg = make_config("NGCF", embed_size=DIM, num_orders=3, device="cuda", model_dir="/tmp/yr_bench")
        model = NGCF(cfg, NU, NI).to(dev)
        opt, lossf = Adam(model.parameters(), lr=1e-4), BPRLoss()
        B = 4096
        # batches as a loader over the train rows yields them: (user, one of its items, uniform negative)
        pick = torch.randint(0, u.numel(), (B,), device=dev)
        bu, bp, bn = u[pick].contiguous(), i[pick].contiguous(), torch.randint(0, NI, (B,), device=dev)

        def make_step(b):
            def step():
                pos, neg = model.bpr_forward(bu[:b], bp[:b], bn[:b], graph)
                opt.zero_grad(); lossf(pos, neg).backward(); opt.step()
            return step
        dt = timed(make_step(B))
        # batch-aware propagation (layer k on the rows the batch's scores need) vs the reference's shape (the whole
        # graph for every batch, cfg.ngcf_subset_fraction = 0), at the reference's default batch and at 4,096
        step_ms = {"4096": round(dt * 1e3, 4), "32": round(timed(make_step(32)) * 1e3, 4)}
        cfg.ngcf_subset_fraction = 0.0
        step_ms["4096_whole_graph"] = round(timed(make_step(B)) * 1e3, 4)
        step_ms["32_whole_graph"] = round(timed(make_step(32)) * 1e3, 4)
        cfg.ngcf_subset_fraction = 0.5
        X = model.embedding.weight.detach()
        t_spmm = timed(lambda: engine.spmm_csr(graph, X))
        alg = graph.nnz * 8 + (graph.n + 1) * 4 + 2 * graph.n * DIM * 4      # CSR + read E + write Z
        out.update(metric="NGCF 3-layer full-graph train step @ dim64", value=round(dt * 1e3, 4), ms_per_step=round(dt * 1e3, 4),
                   config={"workload": "NGCF K=3, 69,716 nodes, %d non-zeros, batch 4096" % graph.nnz},
                   step_ms=step_ms,
                   roofline={"bound": "hbm", "kernel": "spmm_csr_kernel (6 launches per step)",
                             "achieved": round(alg / t_spmm / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg / t_spmm / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                             "avg_kernel_us": round(t_spmm * 1e6, 2),
                             "gathered_GBps": round(graph.nnz * DIM * 4 / t_spmm / 1e9, 1)})
    else:
        from yelprecommendation_amd.cdae_step import CDAEStep
        from yelprecommendation_amd.loss import NSBCELoss
        from yelprecommendation_amd.models.cdae import CDAE
        from yelprecommendation_amd.optim import Adam
        from yelprecommendation_amd.utils import make_config
        B, H = 256, 128
        cfg = make_config("CDAE", hidden_size=H, device="cuda", model_dir="/tmp/yr_bench", lr=1e-4)
        users = torch.randperm(NU, device=dev)[:B]
        x = (torch.rand(B, NI, device=dev) < 0.0008).float()                 # ~30 positives per user
        neg = (torch.rand(B, NI, device=dev) < 0.004).float() * (1 - x)      # neg_times = 5 (train_config.yaml:33)
        forms = {}
        for form in ("autograd", "dense", "sampled"):
            # the trainer's step (cdae_step.py: NS-BCE -> decoder on the loss positions), the same with the
            # full-catalogue decoder on the matrix cores, and the launch-by-launch autograd route
            model = CDAE(cfg, NI, NU)
            model.train()
            opt = Adam(model.parameters(), lr=1e-4)
            if form == "autograd":
                lossf = NSBCELoss()

                def step():
                    pred = model(users, x)
                    opt.zero_grad(); lossf(pred, x, neg).backward(); opt.step()
            else:
                fused, k = CDAEStep(model, opt, True, decoder=form, transposed_wh=True), [0]    # as CDAETrainer builds it

                def step():
                    k[0] += 1
                    fused.step(users, x, neg, seed=k[0], p=model.corruption_level)
            forms[form] = timed(step)
            del model, opt
        dt = forms["sampled"]
        # dominant kernel of the step: the dense Adam pass over all five parameters (7 x 4 bytes per element)
        model = CDAE(cfg, NI, NU)
        tensors = [(p.data, torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p), None, 0)
                   for p in model.parameters()]
        t_adam = timed(lambda: engine.adam_dense_flat(tensors, 1, 1e-4))
        adam_bytes = 28 * sum(p.numel() for p in model.parameters())
        z = torch.rand(B, H, device=dev)
        Wo, bo = model.output_layer.weight.detach(), model.output_layer.bias.detach()
        t_dec = timed(lambda: engine.gemm_f32(z, Wo, transB=True, bias=bo, act=1))
        flops = 2.0 * B * NI * H
        out.update(metric="CDAE full-catalogue train step @ hidden128 batch256", value=round(dt * 1e3, 4),
                   ms_per_step=round(dt * 1e3, 4),
                   config={"workload": "CDAE H=128, 38,048 items, batch 256, NS-BCE (neg_times 5), Adam",
                           "step": "cdae_step.CDAEStep, decoder on the loss positions"},
                   step_forms_ms={k: round(v * 1e3, 4) for k, v in forms.items()},
                   roofline={"bound": "hbm", "kernel": "adam_flat_kernel (all five parameters, one launch)",
                             "achieved": round(adam_bytes / t_adam / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(adam_bytes / t_adam / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                             "avg_kernel_us": round(t_adam * 1e6, 2),
                             "eval_decoder": {"bound": "mfma", "kernel": "gemm_f32_tiled_kernel (z W_o^T + b, sigmoid)",
                                              "achieved": round(flops / t_dec / 1e12, 1), "peak": 157.3,
                                              "unit": "TFLOP/s", "frac": round(flops / t_dec / 1e12 / 157.3, 4),
                                              "avg_kernel_us": round(t_dec * 1e6, 2)}})
    print(json.dumps(out), flush=True)


EVENT_EVERY = 5                # launches are bracketed by HIP events on every 5th timed step (odd / even steps
                               # alternate between one pair around the group and one pair per launch): an event pair
                               # per launch on every step would itself cost ~20 % of a 100 us step


def time_steps_gpu(step, batch, steps, warmup):
    """(seconds per step, kernel-time dict) of `steps` steps on one resident batch; the kernel
    times come from a few further steps with an event pair around every launch."""
    u, p, n = batch
    for _ in range(warmup):
        step.step(u, p, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step(u, p, n)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    step.reset_timers()
    for _ in range(10):
        step.step(u, p, n, record=True)
    return sec, step.kernel_times()


def main():
    args = parse()
    if args.workload != "bpr":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        return other_workload(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node "
                             f"{args.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # rehearsal hook for a one-GPU box: every rank on cuda:0 with gloo collectives (RCCL needs one
    # device per rank); the driver's multi-GPU runs never set it
    rehearsal = os.environ.get("YR_BENCH_REHEARSAL_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI

    from yelprecommendation_amd.bpr_step import BPRMFStep
    from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS, YELP2018_USERS, make_interactions_torch
    from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
    from yelprecommendation_amd.user_shard import UserShard

    # ---- synthetic Yelp2018-shaped input (identical on every rank: same seed) ------------------
    t0 = time.time()
    gen = torch.Generator(device=dev).manual_seed(4321)
    iu, ii = make_interactions_torch(YELP2018_USERS, YELP2018_ITEMS, 47.0, seed=1234, device=dev)
    num_users, num_items, nnz = YELP2018_USERS, YELP2018_ITEMS, iu.numel()
    label = split_train_rows(iu, ii, generator=gen)
    tr = label == 0
    n_train_global = int(tr.sum())
    shard = UserShard(num_users, world, rank)
    strong = args.scaling == "strong"
    # the batch never exceeds one epoch of train rows (nothing a DataLoader over the train set could not produce)
    B = min(args.batch, n_train_global)
    n_pool = max(1, min(4, args.steps + args.warmup))

    def strong_pool(gb, n):
        """fixed GLOBAL batch `gb`: every rank draws the same global stream and keeps the triplets of its users"""
        gs = TripletSampler(iu[tr], ii[tr], num_users, num_items, seed=99)
        su, sp, sn = gs.stream(gb * n)
        out = []
        for k in range(n):
            gu, gp, gn = (t[k * gb:(k + 1) * gb] for t in (su, sp, sn))
            m = (gu >= shard.lo) & (gu < shard.hi)
            out.append(((gu[m] - shard.lo).contiguous(), gp[m].contiguous(), gn[m].contiguous()))
        return out, (su[:gb].contiguous(), sp[:gb].contiguous(), sn[:gb].contiguous())

    def weak_pool(b, n):
        mine = tr & (iu >= shard.lo) & (iu < shard.hi)
        sampler = TripletSampler(iu[mine] - shard.lo, ii[mine], shard.size, num_items, seed=99 + rank)
        su, sp, sn = sampler.stream(b * n)
        return [(su[k * b:(k + 1) * b].contiguous(), sp[k * b:(k + 1) * b].contiguous(),
                 sn[k * b:(k + 1) * b].contiguous()) for k in range(n)]

    # the primary measurement (`value`) is --scaling's; at N > 1 the OTHER one is measured as well (fewer steps) so
    # that one line carries a throughput figure (weak: fixed per-GPU batch) and a training speed-up figure
    # (strong: ONE EPOCH of this data set as the global batch, what a training run can actually use)
    pools = {}
    single_ref_batch = None
    if strong:
        sp_, single_ref_batch = strong_pool(B, n_pool)
        pools["strong"] = sp_, B
        if world > 1:
            pools["weak"] = weak_pool(B, 2), B * world
    else:
        pools["weak"] = weak_pool(B, n_pool), B * world
        if world > 1:
            sp_, single_ref_batch = strong_pool(n_train_global, 2)   # ... and the same batch unsharded (single-GPU reference)
            pools["strong"] = sp_, n_train_global
    primary = "strong" if strong else "weak"
    pool, global_batch = pools[primary]
    del iu, ii, label, tr
    data_s = time.time() - t0

    # ---- tables: xavier-uniform like models/mf.py:15-18; user rows sharded, items replicated ---
    g2 = torch.Generator(device=dev).manual_seed(7)
    bu = (6.0 / (num_users + DIM)) ** 0.5
    bi = (6.0 / (num_items + DIM)) ** 0.5
    U_full = (torch.rand(num_users, DIM, generator=g2, device=dev) * 2 - 1) * bu
    U = U_full[shard.lo:shard.hi].contiguous()
    I = (torch.rand(num_items, DIM, generator=g2, device=dev) * 2 - 1) * bi
    step = BPRMFStep(U, I, lr=1e-4, optimizer="adam", world_size=world,
                     process_group=(dist.group.WORLD if dist else None), time_kernels=True)

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run(st, pl, gb, steps, timed):
        for k in range(steps):
            u, p, n = pl[k % len(pl)]
            # the following batch is known: at N > 1 its index is built under this step's all-reduce
            nxt = pl[(k + 1) % len(pl)] if (world > 1 and k + 1 < steps) else None
            st.step(u, p, n, record=timed and k % EVENT_EVERY == 0, global_batch=gb, next_batch=nxt)

    def measure(st, pl, gb, steps, warmup, timed=False):
        """seconds for exactly `steps` steps between barriers, max over ranks"""
        run(st, pl, gb, warmup, False)
        barrier()
        if timed:
            st.reset_timers()
        t0 = time.perf_counter()
        run(st, pl, gb, steps, timed)
        barrier()
        el = time.perf_counter() - t0
        if dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    elapsed = measure(step, pool, global_batch, args.steps, args.warmup, timed=True)
    kt_primary = step.kernel_times()                 # {name: (avg_us, launches, algorithmic bytes per launch)}
    impl_primary, launches_primary = step.impl, step.launches

    # ---- N > 1: the other scaling mode, the exposed collective time, a single-GPU reference of the strong batch ----
    extra = {}
    if world > 1:
        import torch.distributed as tdist
        k2, w2 = max(5, min(args.steps, 20)), max(2, min(args.warmup, 5))
        secs = {primary: elapsed / args.steps}
        other = "weak" if strong else "strong"
        secs[other] = measure(step, pools[other][0], pools[other][1], k2, w2) / k2
        # the same steps with the collective skipped (measurement only: the tables of the ranks diverge from here
        # on, which is why this comes last): exposed = with - without
        step.skip_collective = True
        nocoll = {m: measure(step, pools[m][0], pools[m][1], k2, w2) / k2 for m in (primary, other)}
        step.skip_collective = False
        item_bytes = I.numel() * 4
        form = step.item_exchange
        extra["collective"] = {
            "backend": tdist.get_backend(), "library": "RCCL over xGMI" if tdist.get_backend() == "nccl" else "gloo (one-GPU rehearsal)",
            "ranks": tdist.get_world_size(), "item_exchange": form,
            "what": "all_reduce(SUM, f32) of the dense item-embedding gradient [I, D], one per step" if form == "all_reduce"
                    else "reduce_scatter of the item gradient + all_gather of the updated item rows, one pair per step",
            "bytes_per_step": item_bytes, "wire_bytes_per_rank": int(2 * (world - 1) / world * item_bytes),
            "exposed_us": round((secs[primary] - nocoll[primary]) * 1e6, 1),
            "step_us_without_collective": round(nocoll[primary] * 1e6, 1), "measured_on": primary}
        for m in ("weak", "strong"):
            gb = pools[m][1]
            extra[m] = {"global_batch": gb, "global_batch_in_epochs": round(gb / n_train_global, 2),
                        "us_per_step": round(secs[m] * 1e6, 1), "triplets_per_s": round(gb / secs[m], 1),
                        "us_per_step_without_collective": round(nocoll[m] * 1e6, 1),
                        "exposed_collective_us": round((secs[m] - nocoll[m]) * 1e6, 1), "steps": args.steps if m == primary else k2}
        # what ONE GPU needs for the strong line's global batch (full tables, no sharding, no collective), measured by
        # every rank on its own GPU in this same run: the strong line's speed-up is a training speed-up
        single = BPRMFStep(U_full.clone(), I.clone(), lr=1e-4, optimizer="adam")
        t1 = measure(single, [single_ref_batch], pools["strong"][1], k2, w2) / k2
        extra["strong"]["single_gpu_us_per_step"] = round(t1 * 1e6, 1)
        extra["strong"]["speedup_vs_single_gpu"] = round(t1 / secs["strong"], 3)
        del single
    step.check()
    value = global_batch * args.steps / elapsed
    per_triplet = algorithmic_bytes_per_triplet(DIM)
    adam_bytes = 6 * 4 * (num_users + num_items) * DIM          # read p,m,v + write p,m,v on every row (fused)

    # ---- roofline (HIP events inside the timed region, one pair per launch) ----------------------
    kt = kt_primary
    # single-GPU pull form: recorded steps alternate between one event pair around the whole launch group
    # ("bpr_pull_step": the step's GPU time with its two launch gaps) and one pair per launch (the split)
    group = kt.pop("bpr_pull_step", None) if "owner_pass_item" in kt else None
    dom = max(kt, key=lambda k: kt[k][0] * kt[k][1])
    avg_us, launches, alg_bytes = kt[dom]
    achieved = alg_bytes / (avg_us * 1e-6) / 1e9
    # the step's kernel time = the SUM of its launches' average durations, each between its own event pair (what the
    # rocprofv3 kernel stats of this command list); the pair around the whole group also spans the two launch gaps and
    # the event records themselves and is reported apart ("group_span_us")
    group_us = sum(v[0] for v in kt.values())
    group_span_us = group[0] if group else None
    local_B = sum(t[0].numel() for t in pool) / len(pool)
    # HBM-side traffic per step from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes, + WRITE_SIZE), committed under profiles/ by
    # scripts/pmc_traffic.py; null when no such measurement is in the tree.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch_per_gpu") == B and tj.get("step_impl") == impl_primary:
                traffic = tj["hbm_bytes_per_step"]
                traffic_source = tj.get("source", "profiles/traffic.json (rocprofv3 --pmc passes of this command, earlier run)")
        except (ValueError, KeyError):
            traffic = None
    # SURVEY 8d's figure (1,560 B per triplet at dim 64) is defined for the whole step; the step is a group of
    # launches, so the headline fraction is the step's bytes over the SUM of its kernels' average durations (the
    # rocprofv3 kernel stats of this command list the same averages), without the dense-Adam bytes the owner
    # passes also move ("algorithmic_GBps_with_adam_bytes" counts them, as a rate).  "kernels" splits the 1,560 B by what each launch must
    # move at least once (ids; three rows read + the user gradient row; the two item gradient rows);
    # "dominant_kernel" is the longest launch under that split.
    step_alg = int(local_B * per_triplet)
    step_rate = step_alg / (group_us * 1e-6) / 1e9
    roofline = {"bound": "hbm", "kernel": f"{impl_primary.split(':')[0]} step = launch group ({launches_primary})",
                "achieved": round(step_rate, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(step_rate / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                # `achieved` is SURVEY 8d's ALGORITHMIC bytes over the kernel time (the contract's definition), not a
                # measured HBM rate: the tables + optimizer state (71 MB) are Infinity-Cache / L2 resident and gathered
                # rows are reused there, so the memory side moves fewer bytes than the algorithm names.  The measured
                # figure is `fabric_GBps` = PMC traffic / kernel time; the kernel group is bound by gather latency
                # (DESIGN 4.2), not by HBM bandwidth.
                "achieved_is": "algorithmic bytes / kernel time (SURVEY 8d), not a measured HBM rate - see fabric_GBps",
                "fabric_GBps": round(traffic / (group_us * 1e-6) / 1e9, 1) if traffic else None,
                "fabric_frac": round(traffic / (group_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "working_set": "tables + Adam state 71 MB: Infinity-Cache resident (256 MiB); bound = gather latency, not HBM bandwidth",
                "avg_kernel_us": round(group_us, 2), "launches": launches,
                "group_span_us": round(group_span_us, 2) if group_span_us else None,
                "algorithmic_bytes_per_launch": step_alg,
                "algorithmic_GBps_with_adam_bytes": round((step_alg + adam_bytes) / (group_us * 1e-6) / 1e9, 1),
                "dominant_kernel": {"kernel": dom, "avg_us": round(avg_us, 2), "algorithmic_bytes": alg_bytes,
                                    "GBps": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4)},
                "kernels": {k: {"avg_us": round(v[0], 2), "algorithmic_bytes": v[2],
                                "GBps": round(v[2] / (v[0] * 1e-6) / 1e9, 1)} for k, v in kt.items()},
                # the whole step (all launches of one batch) against SURVEY 8d's 1,560 B per triplet
                "step": {"launches": launches_primary, "sum_kernel_us": round(group_us, 2),
                         "algorithmic_bytes": int(local_B * per_triplet),
                         "frac": round(local_B * per_triplet / (group_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_GBps_with_adam_bytes": round((local_B * per_triplet + adam_bytes) / (group_us * 1e-6) / 1e9, 1),
                         "frac_wall": round(local_B * per_triplet / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}}

    out = {
        "metric": "BPR triplets/sec @ dim64", "value": round(value, 1), "unit": "triplets/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BPR-MF dim=64 train step (gather+score+loss+grad+dense Adam), "
                               "synthetic Yelp2018 shape, user-sharded" if world > 1 else
                               "BPR-MF dim=64 train step (gather+score+loss+grad+dense Adam), synthetic Yelp2018 shape",
                   "users": num_users, "items": num_items, "interactions": nnz,
                   "train_triplets_per_epoch": n_train_global, "batch_per_gpu": int(local_B) if strong else B,
                   "global_batch": global_batch,
                   # weak scaling keeps the per-GPU batch fixed, so at N > 1 a step spans more than one epoch of
                   # this (fixed-size) data set: every rank's stream wraps over its shard's rows
                   "global_batch_in_epochs": round(global_batch / n_train_global, 2),
                   "optimizer": "adam(dense)", "parallelism": f"user-shard x{world}" if world > 1 else "single",
                   "step_impl": impl_primary, "data_gen_s": round(data_s, 1)},
        "roofline": roofline,
    }
    out.update(extra)

    if rank == 0 and world == 1 and not args.no_sweep:
        # the step at the batch sizes training runs use (reference default 32 ... one epoch per step)
        sweep = {}
        for b in (32, 4096, 65536, 262144, n_train_global):
            if b > pool[0][0].numel() * n_pool:
                continue
            src = torch.cat([t[0] for t in pool])[:b], torch.cat([t[1] for t in pool])[:b], torch.cat([t[2] for t in pool])[:b]
            sec, k2 = time_steps_gpu(step, tuple(t.contiguous() for t in src), 50, 10)
            ksum = sum(v[0] for k, v in k2.items() if k != "bpr_pull_step") * 1e-6
            sweep[str(b)] = {"us_per_step": round(sec * 1e6, 1), "triplets_per_s": round(b / sec, 1),
                             "impl": step.impl.split(":")[0], "sum_kernel_us": round(ksum * 1e6, 1),
                             # rates of SURVEY 8d's algorithmic bytes (not fractions: tables + optimizer state are
                             # cache resident, at one epoch per step the rate passes the 8 TB/s HBM peak)
                             "algorithmic_GBps": round(b * per_triplet / sec / 1e9, 1),
                             "algorithmic_GBps_with_adam_bytes": round((b * per_triplet + adam_bytes) / sec / 1e9, 1)}
        out["batch_sweep"] = sweep
        # The boundary takes device pointers.  A caller whose batches start on the HOST (the reference's DataLoader
        # hands over CPU int64 tensors: 24 B per triplet) pays the PCIe copy first: the same step fed from pinned host
        # batches, copy then step on one stream, and with the copy of batch k + 1 on a second stream under step k.
        # Reported apart: it is never `value`.
        hb = [tuple(t.cpu().pin_memory() for t in pool[k]) for k in range(min(2, n_pool))]
        bufs = [tuple(torch.empty_like(t, device=dev) for t in hb[0]) for _ in range(2)]
        def host_fed(overlap, n):
            side = torch.cuda.Stream(device=dev) if overlap else torch.cuda.current_stream()
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            def upload(k):
                with torch.cuda.stream(side):
                    side.wait_event(done[k & 1])                      # the step that read this buffer last is over
                    for d, h in zip(bufs[k & 1], hb[k % len(hb)]):
                        d.copy_(h, non_blocking=True)
                    ready[k & 1].record(side)
            for e in done:
                e.record()
            upload(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                if overlap and k + 1 < n:
                    upload(k + 1)
                torch.cuda.current_stream().wait_event(ready[k & 1])
                step.step(*bufs[k & 1])
                done[k & 1].record()
                if not overlap and k + 1 < n:
                    upload(k + 1)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        host_fed(True, 10)
        serial, overlapped = host_fed(False, 40), host_fed(True, 40)
        out["host_resident_batches"] = {
            "note": "batches start in pinned host memory (3 x int64 per triplet over PCIe); not `value`",
            "bytes_per_step": int(3 * 8 * B),
            "copy_then_step": {"us_per_step": round(serial * 1e6, 1), "triplets_per_s": round(B / serial, 1)},
            "copy_under_previous_step": {"us_per_step": round(overlapped * 1e6, 1),
                                         "triplets_per_s": round(B / overlapped, 1)}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.mf_torch_cpu import time_steps      # CPU baseline leg only (never the product path)
        cpu_b = B
        cpu_batches = [tuple(t[:cpu_b].cpu() for t in pool[k]) for k in range(min(2, n_pool))]
        tps, csteps, csec = time_steps(num_users, num_items, DIM, cpu_batches, budget_s=args.cpu_budget)
        out["cpu_baseline"] = {"value": round(tps, 1), "unit": "triplets/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"{csteps} steps of batch {cpu_b} (same tables/shape/stream), {csec:.1f} s, "
                                         "torch-CPU op sequence of mf_trainer.py:104-114 with dense Adam"}
        if not args.no_sweep:
            # the CPU op sequence at the reference's default batch and two larger ones
            cs = {}
            for b in (32, 4096, 65536):
                cb = [tuple(t[:b].cpu() for t in pool[k]) for k in range(min(2, n_pool))]
                tps_b, st_b, sec_b = time_steps(num_users, num_items, DIM, cb, budget_s=3.0)
                cs[str(b)] = {"triplets_per_s": round(tps_b, 1), "steps": st_b, "seconds": round(sec_b, 1)}
            out["cpu_baseline"]["batch_sweep"] = cs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
