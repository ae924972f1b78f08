"""Run N full-size steps of one model (for rocprofv3): python scratch/step_models.py ngcf|cdae [batch] [steps]"""
import sys, time
import torch
sys.path.insert(0, '.')
from yelprecommendation_amd.data.synthetic import make_interactions_torch, YELP2018_USERS as NU, YELP2018_ITEMS as NI
from yelprecommendation_amd.graph import LaplacianCSR
from yelprecommendation_amd.loss import BPRLoss, NSBCELoss
from yelprecommendation_amd.models.ngcf import NGCF
from yelprecommendation_amd.models.cdae import CDAE
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device('cuda:0')
which = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if which == 'ngcf':
    u, i = make_interactions_torch(NU, NI, 47.0, device=dev)
    r = torch.randint(1, 6, u.shape, device=dev)
    graph = LaplacianCSR.from_interactions(u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), NU, NI, dev)
    cfg = make_config("NGCF", embed_size=64, num_orders=3, device="cuda", model_dir="/tmp/m")
    model = NGCF(cfg, NU, NI).to(dev)
    opt = Adam(model.parameters(), lr=1e-4); lossf = BPRLoss()
    bu = torch.randint(0, NU, (B,), device=dev); bp = torch.randint(0, NI, (B,), device=dev); bn = torch.randint(0, NI, (B,), device=dev)
    def step():
        pos, neg = model.bpr_forward(bu, bp, bn, graph)
        opt.zero_grad(); l = lossf(pos, neg); l.backward(); opt.step()
else:
    cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/m", lr=1e-4)
    model = CDAE(cfg, NI, NU)
    opt = Adam(model.parameters(), lr=1e-4); lossf = NSBCELoss()
    users = torch.randperm(NU, device=dev)[:B]
    x = (torch.rand(B, NI, device=dev) < 0.0008).float()
    neg = ((torch.rand(B, NI, device=dev) < 0.004).float() * (1 - x))
    model.train()
    def step():
        pred = model(users, x)
        opt.zero_grad(); l = lossf(pred, x, neg); l.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
print(f"{which} B={B}: {dt*1e3:.3f} ms/step")
