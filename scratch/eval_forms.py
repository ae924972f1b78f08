"""Both forms of the fused evaluation sweep at Yelp2018 size: identical lists, time each (cold / hinted).
python scratch/eval_forms.py [D] [k]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine
dev = torch.device('cuda:0')
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nu, ni = 31668, 38048
g = torch.Generator(device=dev).manual_seed(0)
U = torch.randn(nu, d, device=dev, generator=g) * 0.1; I = torch.randn(ni, d, device=dev, generator=g) * 0.1
users = torch.arange(nu, device=dev)
cnt = torch.randint(10, 60, (nu,), device=dev, generator=g)
ptr = torch.zeros(nu + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(cnt, 0)
idx = torch.randint(0, ni, (int(ptr[-1]),), device=dev, generator=g)
sidx = engine.sort_mask_rows(ptr, idx)
def tk(f, n=20):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
res = {}
for form in ("four_waves", "two_roles"):
    res[form] = engine.mf_eval_topk(U, I, users, ptr, sidx, k, form=form)
    cold = tk(lambda: engine.mf_eval_topk(U, I, users, ptr, sidx, k, form=form))
    hinted = tk(lambda: engine.mf_eval_topk(U, I, users, ptr, sidx, k, form=form, hint=res[form]))
    nopre = tk(lambda: engine.mf_eval_topk(U, I, users, ptr, sidx, k, form=form, prescan=False))
    print(f"D={d} k={k} {form}: cold {cold:.3f} ms, hinted {hinted:.3f} ms, no prescan {nopre:.3f} ms", flush=True)
print("identical lists:", bool(torch.equal(res["four_waves"], res["two_roles"])))
