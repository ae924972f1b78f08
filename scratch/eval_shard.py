# usage (GPU box): python scratch/eval_shard.py — fused evaluation of a user shard (1/8, 1/4, 1/2 of the eval users, as a rank
# of the user-sharded run has them): few row blocks, many catalogue slices
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine
dev = torch.device('cuda:0')
nu, ni, d = 31668, 38048, 64
g = torch.Generator(device=dev).manual_seed(0)
U = torch.randn(nu, d, device=dev, generator=g) * 0.1; I = torch.randn(ni, d, device=dev, generator=g) * 0.1
for frac in (8, 4, 2, 1):
    n = nu // frac
    users = torch.arange(n, device=dev)
    cnt = torch.randint(10, 60, (n,), device=dev, generator=g)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(cnt, 0)
    idx = engine.sort_mask_rows(ptr, torch.randint(0, ni, (int(ptr[-1]),), device=dev, generator=g))
    top = engine.mf_eval_topk(U, I, users, ptr, idx, 10)
    for name, kw in (("cold", {}), ("hint", dict(hint=top))):
        engine.mf_eval_topk(U, I, users, ptr, idx, 10, **kw); torch.cuda.synchronize(); t = time.time()
        for _ in range(10): engine.mf_eval_topk(U, I, users, ptr, idx, 10, **kw)
        torch.cuda.synchronize(); ms = (time.time() - t) / 10 * 1e3
        print(f"{n} users {name}: {ms:.3f} ms  ({2 * n * ni * d / ms / 1e9:.0f} TFLOP/s algorithmic)", flush=True)
