"""The reference's own loop shape (model(u,p), model(u,n), zero_grad, loss, backward, step: mf_trainer.py:106-112)
on the drop-in classes, vs the fused BPRMFStep the trainer uses."""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.models.mf import MatrixFactorization
from yelprecommendation_amd.loss import BPRLoss
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device('cuda')
nu, ni = 31668, 38048
cfg = make_config("MF", embed_size=64, device="cuda", model_dir="/tmp/m")
model = MatrixFactorization(cfg, nu, ni).to(dev)
opt = Adam(model.parameters(), lr=1e-4); lossf = BPRLoss()
for B in (32, 4096, 65536):
    u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
    def step():
        pos = model(u, p); neg = model(u, n)
        opt.zero_grad(); loss = lossf(pos, neg); loss.backward(); opt.step()
        return loss
    for _ in range(10): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 100
    print(f"reference-style loop B={B}: {dt*1e6:.0f} us/step")
