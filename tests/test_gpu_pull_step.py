"""GPU parity of the pull-based BPR-MF step (csrc/bpr_pull.hip, through the C ABI) against the
NumPy oracle: loss, both tables and the Adam state after several steps, for every supported
width, uniform and popularity-skewed batches (rows far longer than one chunk of the owner pass),
empty and ragged batches, one and several partition tiles."""
import os
import numpy as np
import pytest
import torch

from oracle import bpr_mf as obpr

pytestmark = pytest.mark.gpu


def _setup(rs, nu, ni, d):
    U = (rs.standard_normal((nu, d)) * 0.2).astype(np.float32)
    I = (rs.standard_normal((ni, d)) * 0.2).astype(np.float32)
    return U, I


def _batch(rs, nu, ni, B, skew):
    u = rs.randint(0, nu, size=B)
    if skew:   # popularity-skewed positives => a few very heavy item rows
        p = np.minimum((rs.pareto(1.2, size=B) * 3).astype(np.int64), ni - 1)
    else:
        p = rs.randint(0, ni, size=B)
    n = rs.randint(0, ni, size=B)
    return u.astype(np.int64), p.astype(np.int64), n.astype(np.int64)


@pytest.mark.parametrize("d", [16, 32, 64, 128])
@pytest.mark.parametrize("B,skew", [(1, False), (257, False), (3000, True), (3000, False), (9000, True),
                                    (20000, False), (40000, True)])
def test_pull_step_matches_oracle(device, d, B, skew):
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(7 * d + B + int(skew))
    nu, ni = 211, 307
    U, I = _setup(rs, nu, ni, d)
    ref = obpr.MFState(U, I, "adam", lr=5e-3)
    step = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3,
                     impl="pull")
    total = 0.0
    for k in range(4):
        u, p, n = _batch(rs, nu, ni, B if k != 2 else max(1, B // 3), skew)   # ragged: step 2 is shorter
        total += float(ref.train_step(u, p, n))
        step.step(*(torch.from_numpy(a).to(device) for a in (u, p, n)))
    got = step.epoch_loss()
    step.check()
    np.testing.assert_allclose(got, total, rtol=2e-5)
    # Adam's m/sqrt(v) amplifies gradient rounding where a gradient nearly cancels: an element
    # may differ by a small fraction of one step (lr = 5e-3), hence atol = 2e-3 * lr
    np.testing.assert_allclose(step.U.cpu().numpy(), ref.U, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(step.I.cpu().numpy(), ref.I, rtol=1e-3, atol=1e-5)
    # all four Adam moments
    np.testing.assert_allclose(step.mU.cpu().numpy(), ref.opt.m[0], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(step.mI.cpu().numpy(), ref.opt.m[1], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(step.vU.cpu().numpy(), ref.opt.v[0], rtol=1e-3, atol=1e-11)
    np.testing.assert_allclose(step.vI.cpu().numpy(), ref.opt.v[1], rtol=1e-3, atol=1e-11)


@pytest.mark.parametrize("shape", ["small", "yelp"])
def test_deterministic_mode_is_bitwise_reproducible(device, shape):
    """The reference is bit-reproducible under a seed.  deterministic=True: two runs of 50 steps
    on the same batches give bit-identical tables, Adam moments and loss (rows summed in triplet order,
    chunks cut at tile boundaries) — and still equal the default mode to rounding."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    g = torch.Generator(device=device).manual_seed(3)
    nu, ni, d, B, steps = (211, 307, 32, 5000, 50) if shape == "small" else (31668, 38048, 64, 1 << 17, 50)
    U = (torch.rand(nu, d, generator=g, device=device) - 0.5) * 0.2
    I = (torch.rand(ni, d, generator=g, device=device) - 0.5) * 0.2
    batches = []
    for _ in range(5):
        u = torch.randint(0, nu, (B,), generator=g, device=device)
        p = (torch.rand(B, generator=g, device=device).pow(2) * ni).long().clamp_(max=ni - 1)   # some long rows
        batches.append((u, p, torch.randint(0, ni, (B,), generator=g, device=device)))
    runs = []
    for run, det in enumerate((True, True, False)):
        st = BPRMFStep(U.clone(), I.clone(), lr=1e-3, deterministic=det, impl="pull")
        # the start order of the item pass's workgroups (heaviest buckets first by default at large batches) is a
        # scheduling matter only: run 0 keeps index order, run 1 takes a random permutation — still bit-identical
        st.auto_item_order = False
        if run == 1:
            st.set_item_order(torch.rand(ni, generator=g, device=device))
            assert st._item_order is not None and sorted(st._item_order.tolist()) == list(range(st._item_order.numel()))
        for k in range(steps):
            st.step(*batches[k % 5])
        st.check()
        runs.append((st.U.clone(), st.I.clone(), st.mU.clone(), st.vU.clone(), st.mI.clone(), st.vI.clone(),
                     torch.tensor(st.epoch_loss(), dtype=torch.float64)))
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    torch.testing.assert_close(runs[0][0], runs[2][0], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(runs[0][1], runs[2][1], rtol=1e-3, atol=1e-5)
    assert abs(runs[0][6].item() - runs[2][6].item()) <= 1e-5 * abs(runs[2][6].item())


@pytest.mark.parametrize("d,nu,ni,B", [(16, 300, 200, 24576), (32, 300, 200, 100000), (64, 300, 200, 24576),
                                       (16, 2516, 3568, 100000), (128, 90, 70, 9000)])
def test_deterministic_mode_with_oversize_tile_segments(device, d, nu, ni, B):
    """Few buckets (tables of a few hundred rows) and a handful of items taking most of the positives: a bucket then
    receives more than one chunk of records from a single tile of the batch.  The owner pass takes such a segment in
    windows of triplet ids (not in the arrival order the partition's LDS atomics left): three runs of three steps
    are bit-identical, and equal the atomic form to summation-order tolerance."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(d + B)
    U0 = torch.from_numpy((rs.standard_normal((nu, d)) * 0.1).astype(np.float32)).to(device)
    I0 = torch.from_numpy((rs.standard_normal((ni, d)) * 0.1).astype(np.float32)).to(device)
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(device)
    batches = [(t(rs.randint(0, nu, B)), t(np.minimum((rs.pareto(1.2, B) * 3).astype(np.int64), ni - 1)), t(rs.randint(0, ni, B)))
               for _ in range(3)]
    runs = []
    for kw in (dict(deterministic=True), dict(deterministic=True), dict(deterministic=True), dict(impl="atomic")):
        st = BPRMFStep(U0.clone(), I0.clone(), lr=1e-2, **kw)
        for b in batches:
            st.step(*b)
        st.check()
        runs.append([x.clone() for x in (st.U, st.I, st.mU, st.vU, st.mI, st.vI)])
    for other in runs[1:3]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    for a, b, tol in zip(runs[0], runs[3], (2e-5, 2e-5, 2e-6, 1e-8, 2e-6, 1e-8)):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=tol)


@pytest.mark.parametrize("B", [1 << 16, 1 << 19])
def test_oversize_item_buckets_are_shared_between_workgroups(device, B):
    """A few items taking several per cent of a batch (here: 8 % of the positives on ONE bucket of 16 items, 3 % on a
    second, one item alone 2 %) at Yelp2018 shape.  The item pass shares such a bucket's tiles between its owner and
    helper workgroups — each leaves its partial sums in a scratch slot, the last to arrive adds them in part order
    and applies Adam (round 2 walked the bucket with ONE workgroup: 1.3 ms at 2^20 triplets, now 0.26 ms).  The
    result equals the atomic form's to summation order, two deterministic runs are bit-identical, and the loss is
    the same in all forms."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    g = torch.Generator(device=device).manual_seed(B)
    nu, ni, d = 31668, 38048, 64
    U = (torch.rand(nu, d, generator=g, device=device) - 0.5) * 0.2
    I = (torch.rand(ni, d, generator=g, device=device) - 0.5) * 0.2
    batches = []
    for _ in range(2):
        u = torch.randint(0, nu, (B,), generator=g, device=device)
        p = torch.randint(0, ni, (B,), generator=g, device=device)
        r = torch.rand(B, generator=g, device=device)
        p = torch.where(r < 0.08, torch.randint(4000, 4016, (B,), generator=g, device=device), p)          # one bucket
        p = torch.where((r >= 0.08) & (r < 0.11), torch.randint(20000, 20016, (B,), generator=g, device=device), p)
        p = torch.where((r >= 0.11) & (r < 0.13), torch.full_like(p, 77), p)                               # one item
        batches.append((u, p.contiguous(), torch.randint(0, ni, (B,), generator=g, device=device)))
    out = {}
    for name, kw in (("pull", dict(impl="pull")), ("det", dict(impl="pull", deterministic=True)),
                     ("det2", dict(impl="pull", deterministic=True)), ("atomic", dict(impl="atomic"))):
        st = BPRMFStep(U.clone(), I.clone(), lr=1e-3, **kw)
        for k in range(4):
            st.step(*batches[k % 2])
        st.check()
        out[name] = [x.clone() for x in (st.U, st.I, st.mI, st.vI)] + [st.epoch_loss()]
    for a, b in zip(out["det"], out["det2"]):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    for other in ("pull", "det"):
        for a, b, tol in zip(out[other][:4], out["atomic"][:4], (2e-5, 2e-5, 2e-6, 1e-8)):
            torch.testing.assert_close(a, b, rtol=2e-3, atol=tol)
        assert abs(out[other][4] - out["atomic"][4]) <= 1e-5 * abs(out["atomic"][4])


def test_pull_step_empty_batch_still_decays_state(device):
    """Dense-Adam semantics: a step with no triplets still moves every row (m/v decay)."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(3)
    U, I = _setup(rs, 50, 60, 64)
    ref = obpr.MFState(U, I, "adam", lr=1e-2)
    step = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=1e-2, impl="pull")
    u, p, n = _batch(rs, 50, 60, 500, False)
    ref.train_step(u, p, n)
    step.step(*(torch.from_numpy(a).to(device) for a in (u, p, n)))
    e = np.zeros(0, np.int64)
    gU, gI = np.zeros_like(ref.U), np.zeros_like(ref.I)
    ref.opt.step([gU, gI])                                   # zero gradient, state still advances
    step.step(*(torch.from_numpy(e).to(device) for _ in range(3)))
    np.testing.assert_allclose(step.U.cpu().numpy(), ref.U, rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(step.I.cpu().numpy(), ref.I, rtol=1e-3, atol=2e-6)


def test_pull_step_flags_bad_indices(device):
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(4)
    U, I = _setup(rs, 20, 30, 32)
    step = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), impl="pull")
    u = torch.tensor([0, 1, 20, 3], dtype=torch.int64, device=device)
    p = torch.tensor([0, 30, 2, 3], dtype=torch.int64, device=device)
    n = torch.tensor([5, 6, 7, -2], dtype=torch.int64, device=device)
    step.step(u, p, n)
    with pytest.raises(IndexError):
        step.check()
    # the single valid triplet (0, 0, 5) was applied with inv_batch = 1/4
    ref = obpr.MFState(U, I, "adam", lr=1e-4)
    loss, gU, gI = obpr.loss_and_grads(ref.U, ref.I, np.array([0]), np.array([0]), np.array([5]))
    ref.opt.step([gU * 0.25, gI * 0.25])
    np.testing.assert_allclose(step.I.cpu().numpy(), ref.I, rtol=1e-3, atol=2e-6)


def test_pull_equals_atomic_at_yelp_shape(device):
    """Size-independent property at BASELINE's full table size: the two implementations of the
    step agree (same loss to 1e-5, same tables to rounding) on a popularity-skewed batch."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    g = torch.Generator(device=device).manual_seed(1)
    nu, ni, d, B = 31668, 38048, 64, 1 << 18
    U = (torch.rand(nu, d, generator=g, device=device) - 0.5) * 0.05
    I = (torch.rand(ni, d, generator=g, device=device) - 0.5) * 0.05
    a = BPRMFStep(U.clone(), I.clone(), lr=1e-3, impl="pull")
    b = BPRMFStep(U.clone(), I.clone(), lr=1e-3, impl="atomic")
    for k in range(3):
        u = torch.randint(0, nu, (B,), generator=g, device=device)
        p = (torch.rand(B, generator=g, device=device).pow(3) * ni).long().clamp_(max=ni - 1)
        n = torch.randint(0, ni, (B,), generator=g, device=device)
        a.step(u, p, n)
        b.step(u, p, n)
    la, lb = a.epoch_loss(), b.epoch_loss()
    a.check(); b.check()
    assert abs(la - lb) <= 1e-5 * abs(lb)
    torch.testing.assert_close(a.U, b.U, rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(a.I, b.I, rtol=1e-3, atol=1e-6)


def test_auto_impl_switches_by_batch_and_stays_consistent(device):
    """impl='auto': small batches take the atomic form, large ones the pull form, on the same state
    (tables, Adam moments, step count); the mixed sequence must equal the all-pull sequence."""
    from yelprecommendation_amd import bpr_step
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(123)
    nu, ni, d = 500, 700, 64
    U, I = _setup(rs, nu, ni, d)
    old = bpr_step.AUTO_PULL_MIN_BATCH
    bpr_step.AUTO_PULL_MIN_BATCH = 2000
    try:
        a = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3, impl="auto")
        b = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3, impl="pull")
        used = []
        for B in (300, 5000, 1999, 2000, 64, 9000):
            t = [torch.from_numpy(x).to(device) for x in _batch(rs, nu, ni, B, True)]
            a.step(*t)
            used.append(a.impl.split(":")[0])
            b.step(*t)
        assert used == ["atomic", "pull", "atomic", "pull", "atomic", "pull"]
        assert abs(a.epoch_loss() - b.epoch_loss()) < 1e-5
        torch.testing.assert_close(a.U, b.U, rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(a.I, b.I, rtol=1e-3, atol=1e-5)
        a.check(); b.check()
    finally:
        bpr_step.AUTO_PULL_MIN_BATCH = old


def test_item_gradient_columns_sum_to_zero_at_full_size(device):
    """Size-independent property at BASELINE's full table size and batch: every triplet adds +g*u to
    its positive item row and -g*u to its negative one, so the columns of the dense item gradient sum
    to zero, and the user/item gradients are tied by  sum_rows(gradI[p]-side) = -sum_rows(...)."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    g = torch.Generator(device=device).manual_seed(5)
    nu, ni, d, B = 31668, 38048, 64, 1 << 20
    U = (torch.rand(nu, d, generator=g, device=device) - 0.5) * 0.1
    I = (torch.rand(ni, d, generator=g, device=device) - 0.5) * 0.1
    st = BPRMFStep(U, I, lr=1e-3, impl="pull", split_item_update=True, item_chunks=3)
    u = torch.randint(0, nu, (B,), generator=g, device=device)
    p = (torch.rand(B, generator=g, device=device).pow(3) * ni).long().clamp_(max=ni - 1)
    n = torch.randint(0, ni, (B,), generator=g, device=device)
    st.step(u, p, n)
    st.check()
    gI = st.gI.double()
    col = gI.sum(0).abs().max().item()
    scale = gI.abs().sum(0).max().item()
    assert col <= 1e-5 * scale, (col, scale)
    loss = st.epoch_loss()
    assert 0.6 < loss < 0.75                       # near-zero scores => loss ~ ln 2


def test_integration_md_binding_stub_runs(device):
    """The ctypes stub printed in INTEGRATION.md (what a maintainer of the reference would paste) is
    executed as written against the built library and must reproduce BPRMFStep."""
    import re
    from yelprecommendation_amd.bpr_step import BPRMFStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# engine_binding\.py.*?)```", text, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                    # the stub loads the library by its repo-relative path
    try:
        exec(compile(block, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    rs = np.random.RandomState(8)
    nu, ni, d, B = 200, 300, 64, 5000
    U = (rs.standard_normal((nu, d)) * 0.1).astype(np.float32)
    I = (rs.standard_normal((ni, d)) * 0.1).astype(np.float32)

    class _Emb:
        def __init__(self, w): self.weight = torch.nn.Parameter(torch.from_numpy(w.copy()).to(device))

    class _Model:
        def __init__(self): self.user_embedding, self.item_embedding = _Emb(U), _Emb(I)

    stub = ns["FusedBPRStep"](_Model(), lr=1e-3)
    ref = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=1e-3, impl="pull")
    for _ in range(3):
        t = [torch.from_numpy(rs.randint(0, hi, B).astype(np.int64)).to(device) for hi in (nu, ni, ni)]
        stub(*t)
        ref.step(*t)
    torch.testing.assert_close(stub.U, ref.U, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(stub.I, ref.I, rtol=1e-4, atol=1e-6)
    assert abs(float(stub.loss_sum.item()) - ref.epoch_loss()) < 1e-5
    assert int(stub.flag.item()) == 0
    # the evaluation stub of the same document (scores + mask + top-k + metrics through the C ABI)
    from yelprecommendation_amd import engine
    block2 = re.search(r"```python\n(_lib\.yr_mf_eval_topk_workspace_bytes.*?)```", text, re.S).group(1)
    exec(compile(block2, "INTEGRATION.md#evaluate", "exec"), ns)
    model = _Model()
    model.user_embedding.weight.data, model.item_embedding.weight.data = stub.U, stub.I
    users = torch.arange(nu, device=device)
    lists = [np.sort(rs.choice(ni, size=rs.randint(0, 30), replace=False)) for _ in range(nu)]
    mptr = np.zeros(nu + 1, np.int64); mptr[1:] = np.cumsum([len(l) for l in lists])
    midx = np.concatenate(lists).astype(np.int64)
    pos = [rs.choice(ni, size=rs.randint(0, 12), replace=False) for _ in range(nu)]
    pptr = np.zeros(nu + 1, np.int64); pptr[1:] = np.cumsum([len(l) for l in pos])
    pidx = np.concatenate(pos).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    got = ns["evaluate"](model, users, t(mptr), t(midx), t(pptr), t(pidx), 10)
    top = engine.mf_eval_topk(stub.U, stub.I, users, t(mptr), t(midx), 10)
    want = engine.rank_metrics(top, t(pptr), t(pidx))[:4].tolist()
    np.testing.assert_allclose(got, want, rtol=1e-12)
