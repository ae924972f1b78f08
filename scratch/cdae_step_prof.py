"""Full-size CDAE step (I = 38,048, H = 128, batch 256): wall time per step; run under rocprofv3 for the kernel split."""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.loss import NSBCELoss
from yelprecommendation_amd.models.cdae import CDAE
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device('cuda:0'); NU, NI = 31668, 38048
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
model = CDAE(make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/yr_bench", lr=1e-4), NI, NU)
opt, lossf = Adam(model.parameters(), lr=1e-4), NSBCELoss()
users = torch.randperm(NU, device=dev)[:B]
x = (torch.rand(B, NI, device=dev) < 0.0013).float()
neg = (torch.rand(B, NI, device=dev) < 0.0065).float() * (1 - x)
model.train()
def step():
    pred = model(users, x)
    opt.zero_grad(); lossf(pred, x, neg).backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
print(f"CDAE step B={B}: {dt*1e3:.3f} ms", flush=True)
