"""Randomised check of yr_cdae_train_lists: for random catalogue widths (ragged, up to 30 k, and a few past 65,536 where a
part is wider than one 2,048-column wave step of the emission), per-user item counts (empty
rows, rows that want more than half of the non-positives), neg_times, dropout levels and an optional second CSR, the encoder
list must equal what the dense route compacts from dropout_p(dense row) (same seed) and the loss list must hold every
positive with target 1 and exactly neg_times x as many distinct non-positives with target 0.
python scratch/lists_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yelprecommendation_amd import engine
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(a).to(dev)
def csr(nu, ni, counts):
    ptr = np.zeros(nu + 1, np.int64); ptr[1:] = np.cumsum(counts)
    idx = np.concatenate([np.sort(rs.choice(ni, c, replace=False)) for c in counts] + [np.zeros(0)]).astype(np.int64)
    return ptr, idx
for c in range(cases):
    ni = int(rs.choice([rs.randint(8, 200), rs.randint(200, 3000), rs.randint(3000, 30000), rs.randint(66000, 160000)], p=[.3, .3, .3, .1]))
    nu, B = int(rs.randint(1, 80)), int(rs.randint(1, 120))
    if ni > 30000: B = min(B, 24)
    neg_times = int(rs.choice([0, 1, 5, 9])); p = float(rs.choice([0.0, 0.3, 0.6, 0.9]))
    cap = ni // (neg_times + 1)                                              # count x (neg_times + 1) <= I: enough non-positives
    counts = np.minimum(rs.choice([0, 1, 5, 40, 400], nu, p=[.1, .2, .4, .25, .05]), cap)
    if rs.rand() < 0.4: counts[rs.randint(nu)] = cap                         # wants (almost) all of the rest: inverted draw
    counts = np.minimum(counts, cap)
    ptr, idx = csr(nu, ni, counts)
    extra = None
    if rs.rand() < 0.4 and neg_times:                                        # held-out items: loss positives, not input
        c2 = np.array([min(int(rs.randint(0, 6)), max(0, (ni - (neg_times + 1) * k) // (neg_times + 1))) for k in counts])
        p2 = np.zeros(nu + 1, np.int64); p2[1:] = np.cumsum(c2)
        i2 = np.concatenate([rs.choice(np.setdiff1d(np.arange(ni), idx[ptr[u]:ptr[u + 1]]), c2[u], replace=False) for u in range(nu)] + [np.zeros(0)]).astype(np.int64)
        extra = (p2, i2)
    users = rs.randint(0, nu, B).astype(np.int64)
    flag = engine.new_error_flag(dev)
    nseed, dseed = int(rs.randint(1 << 40)), int(rs.randint(1 << 40))
    L = engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, nseed, dseed, p, err_flag=flag,
                          extra=None if extra is None else (t(extra[0]), t(extra[1])))
    x = np.zeros((B, ni), np.float32); pos = np.zeros((B, ni), np.float32)
    for b, u in enumerate(users):
        x[b, idx[ptr[u]:ptr[u + 1]]] = 1.0
        pos[b] = x[b]
        if extra is not None: pos[b, extra[1][extra[0][u]:extra[0][u + 1]]] = 1.0
    want = engine.SparseRows(t(x), dseed, p)
    assert torch.equal(L.rows.count, want.count) and torch.equal(L.rows.to_dense(), want.to_dense()), (c, "encoder list")
    target, neg = (a.cpu().numpy() for a in L.loss_dense())
    assert (target == pos).all(), (c, "positives")
    assert float((neg * pos).sum()) == 0.0 and (neg.sum(1) == neg_times * pos.sum(1)).all(), (c, "negatives", ni, neg_times)
    assert int(flag.item()) == 0, (c, "flag", int(flag.item()))
    print(f"case {c}: I={ni} users={nu} B={B} neg_times={neg_times} p={p} max count {counts.max()} second CSR {extra is not None}: ok", flush=True)
print("all", cases, "cases agree")
