"""Randomised check of the NGCF step against the NumPy oracle: random bipartite graphs (sizes not multiples of 32, isolated
nodes, a few very popular items => heavy rows), D in {16, 32, 64, 128}, K in 1..3, random batch sizes, random route (the
autograd ops or the C-side step yr_ngcf_bpr_step) and random subset fraction (whole graph / default / every layer
restricted); two Adam steps: loss and every parameter against oracle/ngcf.py.  python scratch/ngcf_fuzz.py [cases] [seed]"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import ngcf as ongcf
from yelprecommendation_amd.graph import LaplacianCSR, laplacian_scipy
from yelprecommendation_amd.loss import BPRLoss
from yelprecommendation_amd.models.ngcf import NGCF
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tmp = tempfile.mkdtemp()
for c in range(cases):
    d, K = int(rs.choice([16, 32, 64, 128])), int(rs.randint(1, 4))
    nu, ni = int(rs.randint(2, 1200)), int(rs.randint(2, 700))
    nnz = int(rs.randint(1, 12 * (nu + ni)))
    u, i = rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)
    if rs.rand() < 0.5:                                   # a few very popular items
        hot = rs.rand(nnz) < 0.4
        i[hot] = rs.randint(0, min(ni, 3), int(hot.sum()))
    key = np.unique(u.astype(np.int64) * ni + i)          # a set of interactions
    u, i = key // ni, key % ni
    r = rs.randint(1, 6, u.shape[0])
    B = int(rs.choice([1, 7, 64, 513, 4096]))
    L = laplacian_scipy(u, i, r, nu, ni)
    graph = LaplacianCSR.from_scipy(L, dev, heavy_threshold=int(rs.choice([16, 128, 1024])))
    frac = float(rs.choice([0.0, 0.5, 1e9]))
    fused = bool(rs.rand() < 0.5)
    cfg = make_config("NGCF", embed_size=d, num_orders=K, device="cuda", model_dir=tmp, ngcf_subset_fraction=frac)
    torch.manual_seed(c)
    model = NGCF(cfg, nu, ni)
    with torch.no_grad():
        model.embedding.weight.mul_(0.1)
    E0 = model.embedding.weight.detach().numpy().copy()
    W1 = [w.weight.detach().numpy().copy() for w in model.W1]
    W2 = [w.weight.detach().numpy().copy() for w in model.W2]
    model = model.to(dev)
    ref = ongcf.NGCFState(E0, W1, W2, L, nu, lr=1e-3)
    opt = Adam(model.parameters(), lr=1e-3)
    from yelprecommendation_amd.ngcf_step import NGCFStep
    fstep = NGCFStep(model, opt, graph, frac) if fused else None
    for step in range(2):
        bu, bp, bn = rs.randint(0, nu, B), rs.randint(0, ni, B), rs.randint(0, ni, B)
        want = float(ref.train_step(bu, bp, bn))
        ids = [torch.from_numpy(a.astype(np.int64)).to(dev) for a in (bu, bp, bn)]
        if fused:
            fstep.step(*ids)
            got = float(fstep.last_loss().item())
        else:
            pos, neg = model.bpr_forward(*ids, graph)
            opt.zero_grad()
            loss = BPRLoss()(pos, neg)
            loss.backward()
            opt.step()
            got = loss.item()
        np.testing.assert_allclose(got, want, rtol=5e-4, err_msg=f"case {c} loss step {step}")
    if fused:
        fstep.check()
    def close(got, want, what):
        # Adam moves an element by up to lr per step whatever its gradient: where the gradient is at the rounding
        # noise of its summed terms the step may go the other way (+-lr at t = 1, 2) — allowed on a few elements
        err = np.abs(got - want)
        ok = err <= 3e-3 * np.abs(want) + 3e-4
        assert ok.mean() >= 0.998, f"case {c} {what}: {ok.mean():.5f} inside the bar"
        assert err.max() <= 2.1 * 1e-3 * 2, f"case {c} {what}: max error {err.max():.3g}"
    close(model.embedding.weight.detach().cpu().numpy(), ref.E, "E")
    for k in range(K):
        close(model.W1[k].weight.detach().cpu().numpy(), ref.W1[k], f"W1[{k}]")
        close(model.W2[k].weight.detach().cpu().numpy(), ref.W2[k], f"W2[{k}]")
    print(f"case {c}: D={d} K={K} users={nu} items={ni} nnz={len(u)} heavy rows {graph.n_heavy} B={B} {'C step' if fused else 'autograd'} fraction {frac:g}: ok", flush=True)
print("all", cases, "cases agree")
