"""Randomised check of yr_topk_masked (row-wise masked top-k, k <= 64) against NumPy: random row counts / widths (ragged, up
to 60 k columns), strided score buffers, k, mask densities, mask values (-FLT_MAX, 0), exact score ties (order: score
descending, id ascending), rows shorter than k (-1 padding), the mask_rows indirection.  python scratch/topk_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yelprecommendation_amd import engine
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for c in range(cases):
    R = int(rs.randint(1, 200)); N = int(rs.choice([rs.randint(1, 70), rs.randint(70, 3000), rs.randint(3000, 60000)]))
    k = int(rs.choice([1, 2, 10, 16, 33, 64]))
    S = rs.standard_normal((R, N)).astype(np.float32)
    if rs.rand() < 0.4: S = np.round(S * 4) / 4                            # many exact ties
    mv = float(rs.choice([-3.40282e+38, 0.0]))
    nrows = R if rs.rand() < 0.6 else int(rs.randint(1, 30))
    dens = rs.choice([0.0, 0.01, 0.3, 0.98])
    lists = [np.sort(rs.choice(N, size=min(N, rs.binomial(N, dens)), replace=False)) for _ in range(nrows)]
    ptr = np.zeros(nrows + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists + [np.zeros(0)]).astype(np.int64)
    rows = np.arange(R) if nrows == R else rs.randint(0, nrows, R)
    buf = torch.zeros(R, N + int(rs.randint(0, 9)), device=dev)           # row pitch >= N
    buf[:, :N] = t(S)
    got = engine.topk_masked(buf[:, :N], t(ptr), t(idx), k, mask_value=mv,
                             mask_rows=None if nrows == R else t(rows.astype(np.int64))).cpu().numpy()
    for r in range(R):
        s = S[r].copy(); s[lists[rows[r]]] = np.float32(mv)
        order = np.lexsort((np.arange(N), -s.astype(np.float64)))[:k]      # score descending, id ascending
        want = np.r_[order, -np.ones(max(0, k - N), np.int64)]
        assert (got[r] == want).all(), (c, r, R, N, k, mv, got[r][:8], want[:8])
    print(f"case {c}: rows={R} cols={N} k={k} mask density {dens} mask value {mv:g} csr rows={nrows}: ok", flush=True)
print("all", cases, "cases agree")
