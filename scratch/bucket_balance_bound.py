"""Upper bound of popularity-aware item buckets (timing experiment): the item pass needs two rounds of workgroups
(2,378 sixteen-row buckets > 2,048 resident slots).  If the items of a bucket could be CHOSEN (an indirection instead of
16 consecutive ids), 1,718 'normal' buckets and 660 'light' ones of half the load would let every resident workgroup end
at about the same time.  Emulated here by RELABELLING the batch's item ids so that consecutive ids form such buckets
(the tables are random either way), then timing the unchanged step.  python scratch/bucket_balance_bound.py [B]"""
import sys, heapq, time, numpy as np, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
d, R = 64, 16
gen = torch.Generator(device=dev).manual_seed(4321)
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii, generator=gen) == 0
u, p, n = (t.contiguous() for t in TripletSampler(iu[tr], ii[tr], NU, NI, seed=99).stream(B))
cnt = (torch.bincount(p, minlength=NI) + torch.bincount(n, minlength=NI)).cpu().numpy()
nb = (NI + R - 1) // R
slots = 2048 - 40                                   # resident workgroups left to owners (helpers / sizing aside)
n_light = max(0, 2 * (nb - slots)); n_norm = nb - n_light

def relabel(light_share):
    """items -> buckets: greedy, heaviest item first, into the bucket furthest below its target with a free row"""
    target = np.concatenate([np.full(n_norm, 1.0), np.full(n_light, light_share)])
    cap = np.full(nb, R); cap[-1] = NI - R * (nb - 1)
    order = np.argsort(-cnt, kind="stable")
    load = np.zeros(nb); used = np.zeros(nb, int)
    heap = [(0.0, b) for b in range(nb)]
    heapq.heapify(heap)
    new_id = np.empty(NI, np.int64)
    for it in order:
        while True:
            f, b = heapq.heappop(heap)
            if used[b] < cap[b]:
                break
        new_id[it] = b * R + used[b]
        used[b] += 1; load[b] += cnt[it]
        if used[b] < cap[b]:
            heapq.heappush(heap, (load[b] / target[b], b))
    return new_id, load

def run(pp, nn, name):
    step = BPRMFStep(torch.randn(NU, d, device=dev) * 0.05, torch.randn(NI, d, device=dev) * 0.05, lr=1e-4, impl="pull", time_kernels=True)
    for _ in range(10): step.step(u, pp, nn)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): step.step(u, pp, nn)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    step.reset_timers()
    for _ in range(10): step.step(u, pp, nn, record=True)
    kt = step.kernel_times()
    print(f"{name}: {dt * 1e6:.1f} us per step; " + ", ".join(f"{k} {v[0]:.1f}" for k, v in kt.items() if k != "bpr_pull_step"), flush=True)

print(f"B = {B}: {nb} item buckets, {n_norm} normal + {n_light} light; records per bucket now: mean {cnt.sum() / nb:.0f}")
run(p, n, "ids as generated")
for share in (1.0, 0.5, 0.35, 0.25):
    new_id, load = relabel(share)
    t = torch.from_numpy(new_id).to(dev)
    print(f"  light share {share}: normal buckets {load[:n_norm].mean():.0f} records (max {load[:n_norm].max():.0f}), light {load[n_norm:].mean() if n_light else 0:.0f}")
    run(t[p], t[n], f"relabelled, light buckets at {share} of a normal one")
