import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
dev = torch.device('cuda:0')
nu, ni, d = 31668, 38048, 64
U = torch.randn(nu, d, device=dev) * 0.05; I = torch.randn(ni, d, device=dev) * 0.05
step = BPRMFStep(U, I, lr=1e-4)
for B in (32, 256, 4096, 32768, 65536):
    u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
    for _ in range(20): step.step(u, p, n)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): step.step(u, p, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 200
    print(f"B={B}: {dt*1e6:.1f} us/step  {B/dt/1e6:.2f} M triplets/s")
