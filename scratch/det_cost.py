"""Cost of the deterministic mode: python scratch/det_cost.py [sizes...]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
dev = torch.device('cuda:0'); nu, ni, d = 31668, 38048, 64
for B in [int(x) for x in sys.argv[1:]] or [65536, 262144, 524288, 920629]:
    u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
    out = []
    for det in (False, True):
        st = BPRMFStep(torch.randn(nu, d, device=dev) * 0.05, torch.randn(ni, d, device=dev) * 0.05, lr=1e-4, impl="pull", deterministic=det)
        for _ in range(20): st.step(u, p, n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): st.step(u, p, n)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 100 * 1e6)
    print(f"B={B}: default {out[0]:.1f} us/step, deterministic {out[1]:.1f} us/step (+{(out[1]/out[0]-1)*100:.0f} %)", flush=True)
