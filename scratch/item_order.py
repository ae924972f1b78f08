"""Item pass with and without the heaviest-first start order (scratch): us per step at several batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(4321)
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii, generator=gen) == 0
sampler = TripletSampler(iu[tr], ii[tr], NU, NI, seed=99)
for B in (65536, 131072, 262144, 524288, 920629):
    u, p, n = (t.contiguous() for t in sampler.stream(min(B, int(tr.sum()))))
    for mode in ("index order", "heaviest first", "train-set degrees"):
        step = BPRMFStep(torch.randn(NU, 64, device=dev) * 0.05, torch.randn(NI, 64, device=dev) * 0.05, lr=1e-4, impl="pull")
        step.auto_item_order = mode == "heaviest first"
        if mode == "train-set degrees":
            step.set_item_order(torch.bincount(ii[tr], minlength=NI).double() + tr.sum().double() / NI)
        for _ in range(10): step.step(u, p, n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): step.step(u, p, n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 200
        print(f"B={u.numel():7d} {mode:18s} {dt * 1e6:7.1f} us/step", flush=True)
