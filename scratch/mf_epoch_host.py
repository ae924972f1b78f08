"""MF training epochs through MFTrainer.train + EpochLoader at several batch sizes: wall time per step against the GPU
time of the step (bench.py's batch sweep) — is the Python loop or the GPU the limit?  python scratch/mf_epoch_host.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import EpochLoader, TripletSampler
from yelprecommendation_amd.trainers import MFTrainer
from yelprecommendation_amd.utils import make_config
dev = torch.device("cuda")
u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
keep = torch.rand(u.shape, device=dev) < 0.6                      # ~ the 60 % train split
u, i = u[keep], i[keep]
for bs in (32, 256, 4096, 65536):
    t = MFTrainer(make_config("MF", device="cuda", model_dir="/tmp/yr_mf_host", embed_size=64, batch_size=bs), NI, NU)
    loader = EpochLoader(TripletSampler(u, i, NU, NI, seed=5), bs, True)
    steps = len(loader)
    if bs == 32:                                                   # a slice of the 29 k steps is enough
        import itertools
        class Head:
            def __init__(s, l, n): s.l, s.n = l, n
            def __iter__(s): return itertools.islice(iter(s.l), s.n)
            def __len__(s): return s.n
        loader, steps = Head(loader, 3000), 3000
    t.train(loader); torch.cuda.synchronize()
    t0 = time.perf_counter(); t.train(loader); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {bs}: {steps} steps in {dt * 1e3:.1f} ms = {dt / steps * 1e6:.1f} us per step", flush=True)
