"""A fixed number of BPR steps at one batch size on Yelp2018-shaped synthetic triplets (for rocprofv3):
python scratch/step_prof.py B [impl] [synth|uniform|skew] [steps]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
dev = torch.device('cuda:0')
B = int(sys.argv[1]); impl = sys.argv[2] if len(sys.argv) > 2 else "auto"
dist = sys.argv[3] if len(sys.argv) > 3 else "synth"; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
d = 64
import os
if os.environ.get("YR_NI"):            # experiments: another catalogue size (uniform / skew ids only)
    NI = int(os.environ["YR_NI"])
if dist == "synth":
    gen = torch.Generator(device=dev).manual_seed(4321)
    iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
    tr = split_train_rows(iu, ii, generator=gen) == 0
    u, p, n = TripletSampler(iu[tr], ii[tr], NU, NI, seed=99).stream(B)
    u, p, n = u.contiguous(), p.contiguous(), n.contiguous()
else:
    u = torch.randint(0, NU, (B,), device=dev); n = torch.randint(0, NI, (B,), device=dev)
    p = (torch.rand(B, device=dev).pow(3) * NI).long().clamp_(max=NI - 1) if dist == "skew" else torch.randint(0, NI, (B,), device=dev)
step = BPRMFStep(torch.randn(NU, d, device=dev) * 0.05, torch.randn(NI, d, device=dev) * 0.05, lr=1e-4, impl=impl)
for _ in range(10): step.step(u, p, n)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step.step(u, p, n)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
step.check()
print(f"{impl} {dist} B={B}: {dt*1e6:.1f} us/step  {B/dt/1e6:.2f} M triplets/s", flush=True)
