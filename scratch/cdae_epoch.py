"""A whole CDAE training epoch at Yelp2018 size with the device-side batch loader: ms per step including the
batch construction (dense rows + negative masks), and the loader alone."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.trainers import CDAETrainer
from yelprecommendation_amd.utils import make_config

dev = torch.device("cuda")
u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
data = CDAEInteractions.from_interactions(u, i, NU, NI, seed=1, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/yr_cdae_epoch", lr=1e-4, batch_size=B,
                  negative_sampling=True, neg_times=5, loss_name="bce")
trainer = CDAETrainer(cfg, NI, NU)
form = sys.argv[2] if len(sys.argv) > 2 else "lists"
loader = CDAEBatchLoader(data, "train", batch_size=B, neg_times=5, shuffle=True, seed=3, lists=form == "lists",
                         dropout=trainer.model.corruption_level)
steps = len(loader)
print("batches as", form)
for name, fn in (("loader alone", lambda: [None for _ in loader]), ("epoch (loader + step)", lambda: trainer.train(loader))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.1f} ms / {steps} steps = {dt / steps * 1e3:.3f} ms per step", flush=True)
print("train loss", out)
