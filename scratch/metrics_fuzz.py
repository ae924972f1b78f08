"""Randomised check of the device metrics (yr_rank_metrics) against the Python definitions of metric.py: random top-k lists
(with repeated ids, -1 padding), random held-out lists (empty, longer than k, longer than 64, with repeated ids), k in 1..32,
with and without the row indirection.  python scratch/metrics_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yelprecommendation_amd import engine
from yelprecommendation_amd.metric import ranking_metrics
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for c in range(cases):
    n, k, ni = int(rs.randint(1, 400)), int(rs.choice([1, 2, 5, 10, 16, 20, 32, 64, 70])), int(rs.choice([5, 50, 5000]))
    top = rs.randint(0, ni, (n, k)).astype(np.int64)
    if rs.rand() < 0.3: top[rs.rand(n, k) < 0.1] = -1
    nrows = n if rs.rand() < 0.5 else int(rs.randint(1, 50))          # rows of the CSR (indirection: several users share one)
    lens = rs.choice([0, 1, 3, 12, 70, 150], nrows, p=[.15, .2, .3, .25, .07, .03])
    lists = [rs.randint(0, ni, l).astype(np.int64) for l in lens]
    ptr = np.zeros(nrows + 1, np.int64); ptr[1:] = np.cumsum(lens); idx = np.concatenate(lists + [np.zeros(0, np.int64)])
    rows = np.arange(n) if nrows == n else rs.randint(0, nrows, n)
    actual = [lists[r].tolist() for r in rows]
    try:
        want = ranking_metrics(actual, top.tolist(), k)
    except ZeroDivisionError:                            # no user with a held-out item: the definitions divide by zero
        want = None
    got = engine.rank_metrics(torch.from_numpy(top).to(dev), torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev),
                              None if nrows == n else torch.from_numpy(rows.astype(np.int64)).to(dev))[:4].tolist()
    if want is None:
        assert not any(np.isfinite(g) for g in got[1:]), (c, got)
        print(f"case {c}: no held-out items at all: non-finite on the device too", flush=True)
        continue
    ok = all((np.isnan(w) and np.isnan(g)) or abs(w - g) <= 1e-12 * max(1.0, abs(w)) for w, g in zip(want, got))
    assert ok, (c, n, k, ni, nrows, want, got)
    print(f"case {c}: n={n} k={k} items={ni} csr rows={nrows}: ok", flush=True)
print("all", cases, "cases agree")
