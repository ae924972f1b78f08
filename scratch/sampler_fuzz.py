"""Randomised check of the device triplet sampler: for random interaction sets (tiny to mid, users without rows, users who
own almost the whole catalogue), every epoch must (a) visit each (user, item) row exactly once, in another order than the
previous epoch when shuffled and in row order when not, (b) draw a negative inside the catalogue that is not a positive of
the user, (c) cut at [first, first + count) consistently with the whole epoch, (d) repeat for the same (seed, epoch) and
change with the epoch.  python scratch/sampler_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yelprecommendation_amd.data.triplets import TripletSampler
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for c in range(cases):
    nu, ni = int(rs.randint(1, 600)), int(rs.randint(2, 900))
    nnz = int(rs.randint(1, 20 * nu + 2))
    u, i = rs.randint(0, nu, nnz), rs.randint(0, ni, nnz)
    if rs.rand() < 0.3:                                  # one user owns all but two items
        u = np.r_[u, np.zeros(ni - 2, np.int64)]; i = np.r_[i, np.arange(ni - 2)]
    key = np.unique(u.astype(np.int64) * ni + i); u, i = key // ni, key % ni
    deg = np.bincount(u, minlength=nu)
    keep = deg[u] < ni                                   # a user with every item has no negative (the reference loops forever)
    u, i = u[keep], i[keep]
    if len(u) == 0: continue
    s = TripletSampler(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), nu, ni, seed=int(rs.randint(1 << 30)))
    pos = set(map(int, u.astype(np.int64) * ni + i))
    prev = None
    for ep, shuffle in ((0, True), (1, True), (2, False)):
        a, b, n = (t.cpu().numpy() for t in s.draw(ep, shuffle=shuffle))
        a2, b2, n2 = (t.cpu().numpy() for t in s.draw(ep, shuffle=shuffle))
        assert (a == a2).all() and (b == b2).all() and (n == n2).all(), (c, "repeatable")
        rows = a.astype(np.int64) * ni + b
        assert sorted(rows.tolist()) == sorted(pos), (c, "each row once")
        if not shuffle: assert (rows == np.sort(rows)).all() or (rows == u.astype(np.int64) * ni + i).all(), (c, "row order")
        assert ((n >= 0) & (n < ni)).all() and not (set(map(int, a.astype(np.int64) * ni + n)) & pos), (c, "negatives")
        if prev is not None and shuffle and len(rows) > 20: assert (rows != prev).any(), (c, "epochs differ")
        prev = rows
        first = int(rs.randint(0, len(rows))); count = int(rs.randint(0, len(rows) - first + 1))
        if count:
            ca, cb, cn = (t.cpu().numpy() for t in s.draw(ep, first, count, shuffle))
            assert (ca == a[first:first + count]).all() and (cb == b[first:first + count]).all() and (cn == n[first:first + count]).all(), (c, "window")
    s.check()
    print(f"case {c}: users={nu} items={ni} rows={len(u)} max degree {deg.max()}: ok", flush=True)
print("all", cases, "cases agree")
