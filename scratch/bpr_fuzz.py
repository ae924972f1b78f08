"""Randomised consistency check of the BPR-MF step: for random table sizes, D, batch sizes (around the form thresholds, tiny,
ragged), id skew and optimiser settings, the pull form, the atomic form and the deterministic pull form must agree after
three steps (tables, all Adam moments, loss) within f32 summation-order tolerance, and the deterministic form must repeat
bit for bit.  python scratch/bpr_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yelprecommendation_amd.bpr_step import BPRMFStep
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for c in range(cases):
    d = int(rs.choice([16, 32, 64, 128]))
    nu, ni = int(rs.randint(3, 5000)), int(rs.randint(3, 5000))
    B = int(rs.choice([1, 2, 63, 64, 65, 257, 1000, 4096, 24575, 24576, 24577, 70000, rs.randint(1, 200000)]))
    opt = str(rs.choice(["adam", "adamw"])); wd = float(rs.choice([0.0, 1e-2]))
    U0 = torch.from_numpy((rs.standard_normal((nu, d)) * 0.1).astype(np.float32)).to(dev)
    I0 = torch.from_numpy((rs.standard_normal((ni, d)) * 0.1).astype(np.float32)).to(dev)
    batches, any_skew = [], False
    for _ in range(3):
        u = rs.randint(0, nu, B); skew = rs.rand() < 0.5
        any_skew |= skew
        p = np.minimum((rs.pareto(1.2, B) * 3).astype(np.int64), ni - 1) if skew else rs.randint(0, ni, B)
        n = rs.randint(0, ni, B)
        batches.append(tuple(torch.from_numpy(a.astype(np.int64)).to(dev) for a in (u, p, n)))
    out = {}
    for name, kw in (("pull", dict(impl="pull")), ("atomic", dict(impl="atomic")), ("det", dict(deterministic=True)), ("det2", dict(deterministic=True))):
        s = BPRMFStep(U0.clone(), I0.clone(), lr=1e-2, weight_decay=wd, optimizer=opt, **kw)
        for b in batches: s.step(*b)
        s.check()
        out[name] = [t.clone() for t in (s.U, s.I, s.mU, s.vU, s.mI, s.vI)] + [s.loss_accum.clone().float().reshape(-1)[:1]]
    # tables: where |g| cancels to the order of Adam's eps, lr * m / (sqrt(v) + eps) turns summation-order noise into a
    # fraction of lr (here 1e-2 per step): 1 % of the largest possible movement is allowed on top; the moments are strict
    for name in ("atomic", "det"):
        for a, b, tol in zip(out["pull"], out[name], (2e-5 + 3e-4, 2e-5 + 3e-4, 2e-6, 1e-8, 2e-6, 1e-8, 1e-3)):
            assert torch.allclose(a.double(), b.double(), rtol=2e-3, atol=tol), (c, name, d, nu, ni, B, opt, wd, float((a - b).abs().max()))
    for a, b in zip(out["det"], out["det2"]):
        assert torch.equal(a, b), (c, "deterministic form repeats", d, nu, ni, B)
    print(f"case {c}: D={d} users={nu} items={ni} B={B} {opt} wd={wd}: ok", flush=True)
print("all", cases, "cases agree")
