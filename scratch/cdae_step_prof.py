"""CDAE full-size step: fused (cdae_step.py) vs autograd route, ms per step; optional per-launch profile via rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.cdae_step import CDAEStep
from yelprecommendation_amd.loss import NSBCELoss
from yelprecommendation_amd.models.cdae import CDAE
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
NU, NI, B, H = 31668, 38048, int(os.environ.get('YR_B', '256')), 128
dev = torch.device("cuda")
model = CDAE(make_config("CDAE", hidden_size=H, device="cuda", model_dir="/tmp/yr_bench", lr=1e-4), NI, NU)
opt, lossf = Adam(model.parameters(), lr=1e-4), NSBCELoss()
users = torch.randperm(NU, device=dev)[:B]
x = (torch.rand(B, NI, device=dev) < 0.0008).float()
neg = (torch.rand(B, NI, device=dev) < 0.004).float() * (1 - x)
model.train()


def timed(fn, n=60, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def auto():
    pred = model(users, x)
    opt.zero_grad(); lossf(pred, x, neg).backward(); opt.step()


if mode in ("both", "auto"):
    print("autograd route  %.4f ms" % timed(auto), flush=True)
if mode in ("both", "fused", "dense", "sampled"):
    step = CDAEStep(model, opt, decoder="auto" if mode in ("both", "fused") else mode, transposed_wh=os.environ.get("YR_WHT", "1") == "1")
    print("decoder", step.decoder)
    k = [0]

    def fused():
        k[0] += 1
        step.step(users, x, neg, seed=k[0], p=model.corruption_level)
    print("fused step      %.4f ms" % timed(fused), flush=True)
    print("loss", float(step.last_loss()))
