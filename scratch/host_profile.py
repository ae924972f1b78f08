"""cProfile of the host side of a CDAE step at batch 32 (launch-bound regime)."""
import sys, cProfile, pstats, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.data.synthetic import YELP2018_USERS as NU, YELP2018_ITEMS as NI
from yelprecommendation_amd.loss import NSBCELoss
from yelprecommendation_amd.models.cdae import CDAE
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device('cuda:0'); B = 32
cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/m", lr=1e-4)
model = CDAE(cfg, NI, NU); opt = Adam(model.parameters(), lr=1e-4); lossf = NSBCELoss()
users = torch.randperm(NU, device=dev)[:B]
x = (torch.rand(B, NI, device=dev) < 0.0008).float(); neg = ((torch.rand(B, NI, device=dev) < 0.004).float() * (1 - x))
model.train()
def step():
    pred = model(users, x); opt.zero_grad(); l = lossf(pred, x, neg); l.backward(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
