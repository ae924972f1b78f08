"""Step time across batch sizes at Yelp2018 shape (uniform ids): python scratch/small_batch.py [impl] [sizes...]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
dev = torch.device('cuda:0')
nu, ni, d = 31668, 38048, 64
impl = sys.argv[1] if len(sys.argv) > 1 else "auto"
sizes = [int(x) for x in sys.argv[2:]] or [32, 256, 4096, 16384, 65536, 131072, 262144, 524288, 920629, 1048576]
U = torch.randn(nu, d, device=dev) * 0.05; I = torch.randn(ni, d, device=dev) * 0.05
step = BPRMFStep(U, I, lr=1e-4, impl=impl)
for B in sizes:
    u = torch.randint(0, nu, (B,), device=dev)
    p = torch.randint(0, ni, (B,), device=dev)
    n = torch.randint(0, ni, (B,), device=dev)
    for _ in range(20): step.step(u, p, n)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): step.step(u, p, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 200
    print(f"{impl} B={B}: {dt*1e6:.1f} us/step  {B/dt/1e6:.2f} M triplets/s", flush=True)
step.check()
