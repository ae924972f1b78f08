import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
dev = torch.device('cuda:0')
nu, ni, d = 31668, 38048, 64
for B in (16384, 32768, 49152, 65536, 98304, 131072):
    u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
    out = []
    for impl in ("atomic", "pull"):
        U = torch.randn(nu, d, device=dev) * 0.05; I = torch.randn(ni, d, device=dev) * 0.05
        step = BPRMFStep(U, I, lr=1e-4, impl=impl)
        for _ in range(10): step.step(u, p, n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): step.step(u, p, n)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 100 * 1e6)
    print(f"B={B}: atomic {out[0]:.1f} us  pull {out[1]:.1f} us")
