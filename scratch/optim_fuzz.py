"""Randomised check of the dense optimizers against oracle/adam.py: random tensor sizes (1 … 3 M elements, tails that are
not multiples of 4), step counts, learning rates, betas, eps, weight decay, Adam / AdamW, zero_grad; single-tensor and
multi-tensor launches (up to 40 tensors incl. empty ones); SGD.  python scratch/optim_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import adam as oadam
from yelprecommendation_amd import engine
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(a.copy()).to(dev)
for c in range(cases):
    step = int(rs.choice([1, 2, 10, 1000, 100000])); lr = float(rs.choice([1e-4, 1e-2, 1.0]))
    b1, b2 = float(rs.choice([0.9, 0.5, 0.0])), float(rs.choice([0.999, 0.9]))
    eps = float(rs.choice([1e-8, 1e-3])); wd = float(rs.choice([0.0, 1e-2, 0.3])); dec = bool(rs.rand() < 0.5)
    sizes = [int(rs.choice([0, 1, 3, 4, 5, 255, 1023, 4097, rs.randint(1, 3000000)])) for _ in range(int(rs.randint(1, 41)))]
    host = [tuple((rs.standard_normal(n) * s).astype(np.float32) for s in (1.0, 1.0, 0.1)) + (np.abs(rs.standard_normal(n) * 0.01).astype(np.float32),) for n in sizes]
    devs = [tuple(t(a) for a in tup) for tup in host]
    multi = bool(rs.rand() < 0.5)
    kw = dict(beta1=b1, beta2=b2, eps=eps, weight_decay=wd, decoupled=dec)
    if multi:
        engine.adam_dense_multi(devs, step, lr, zero_grad=True, **kw)
    else:
        for tup in devs:
            if tup[0].numel(): engine.adam_dense(*tup, step, lr, zero_grad=True, **kw)
    for (p, g, m, v), (dp, dg, dm, dv) in zip(host, devs):
        oadam.adam_update(p, g, m, v, step, lr, b1, b2, eps, wd, dec)
        np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=2e-5, atol=1e-7, err_msg=f"case {c} p")
        np.testing.assert_allclose(dm.cpu().numpy(), m, rtol=2e-6, atol=1e-9, err_msg=f"case {c} m")
        np.testing.assert_allclose(dv.cpu().numpy(), v, rtol=2e-6, atol=1e-12, err_msg=f"case {c} v")
        assert dg.numel() == 0 or float(dg.abs().max()) == 0.0
    n = int(rs.randint(1, 100000)); p, g = rs.standard_normal(n).astype(np.float32), rs.standard_normal(n).astype(np.float32)
    dp, dg = t(p), t(g)
    oadam.sgd_update(p, g, lr, wd); engine.sgd_dense(dp, dg, lr, wd)
    np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=1e-6, atol=1e-7, err_msg=f"case {c} sgd")
    print(f"case {c}: {len(sizes)} tensors (max {max(sizes)}) step={step} lr={lr} betas=({b1},{b2}) eps={eps} wd={wd} {'adamw' if dec else 'adam'} {'multi' if multi else 'single'}: ok", flush=True)
print("all", cases, "cases agree")
