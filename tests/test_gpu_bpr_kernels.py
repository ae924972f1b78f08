"""GPU parity: HIP kernels (through the C ABI) vs the NumPy oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import adam as oadam
from oracle import bpr_mf as obpr

from replay import assert_topk_equal_up_to_near_ties

pytestmark = pytest.mark.gpu

# float32 parity: the kernels reassociate sums (wave shuffles, float atomics), so
# results agree to rounding, not bitwise.
RTOL, ATOL = 2e-5, 2e-6


def _tables(rs, nu, ni, d):
    U = (rs.standard_normal((nu, d)) * 0.3).astype(np.float32)
    I = (rs.standard_normal((ni, d)) * 0.3).astype(np.float32)
    return U, I


@pytest.mark.parametrize("d", [16, 32, 64, 128])
@pytest.mark.parametrize("B", [1, 63, 256, 1000, 5000])
def test_mf_score(device, d, B):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(B + d)
    nu, ni = 301, 517
    U, I = _tables(rs, nu, ni, d)
    u = rs.randint(0, nu, size=B).astype(np.int64)
    i = rs.randint(0, ni, size=B).astype(np.int64)
    out = engine.mf_score(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device),
                          torch.from_numpy(u).to(device), torch.from_numpy(i).to(device))
    np.testing.assert_allclose(out.cpu().numpy(), obpr.forward(U, I, u, i), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d", [16, 32, 64, 128])
@pytest.mark.parametrize("B", [1, 7, 256, 777, 4096])
def test_bpr_fwd_bwd(device, d, B):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(1000 + B + d)
    nu, ni = 97, 131            # small tables => many duplicate rows inside the batch
    U, I = _tables(rs, nu, ni, d)
    u = rs.randint(0, nu, size=B).astype(np.int64)
    p = rs.randint(0, ni, size=B).astype(np.int64)
    n = rs.randint(0, ni, size=B).astype(np.int64)
    loss, gU, gI = obpr.loss_and_grads(U, I, u, p, n)

    dU, dI = torch.from_numpy(U).to(device), torch.from_numpy(I).to(device)
    gradU, gradI = torch.zeros_like(dU), torch.zeros_like(dI)
    partials = torch.full((engine.LOSS_PARTIALS,), 7.0, dtype=torch.float32, device=device)
    flag = engine.new_error_flag(device)
    engine.bpr_mf_fwd_bwd(dU, dI, *(torch.from_numpy(a).to(device) for a in (u, p, n)),
                          gradU, gradI, partials, err_flag=flag)
    out = engine.loss_finalize(partials, 1.0 / B)
    assert int(flag.item()) == 0
    np.testing.assert_allclose(out.item(), loss, rtol=1e-5)
    np.testing.assert_allclose(gradU.cpu().numpy(), gU, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gradI.cpu().numpy(), gI, rtol=1e-4, atol=1e-6)

    # forward-only mode (validate): same loss, no gradient buffers
    partials.fill_(3.0)
    engine.bpr_mf_fwd_bwd(dU, dI, *(torch.from_numpy(a).to(device) for a in (u, p, n)), None, None, partials)
    np.testing.assert_allclose(engine.loss_finalize(partials, 1.0 / B).item(), loss, rtol=1e-5)


def test_bpr_empty_and_bad_index(device):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(5)
    U, I = _tables(rs, 10, 12, 64)
    dU, dI = torch.from_numpy(U).to(device), torch.from_numpy(I).to(device)
    gradU, gradI = torch.zeros_like(dU), torch.zeros_like(dI)
    partials = torch.ones(engine.LOSS_PARTIALS, dtype=torch.float32, device=device)
    e = torch.zeros(0, dtype=torch.int64, device=device)
    engine.bpr_mf_fwd_bwd(dU, dI, e, e, e, gradU, gradI, partials)        # empty batch
    assert float(partials.sum().item()) == 0.0 and float(gradU.abs().sum().item()) == 0.0
    u = torch.tensor([0, 3, 10, 2], dtype=torch.int64, device=device)      # 10 is out of range
    p = torch.tensor([1, 2, 3, 12], dtype=torch.int64, device=device)      # 12 is out of range
    n = torch.tensor([4, 5, 6, -1], dtype=torch.int64, device=device)      # -1 is out of range
    flag = engine.new_error_flag(device)
    engine.bpr_mf_fwd_bwd(dU, dI, u, p, n, gradU, gradI, partials, err_flag=flag)
    assert int(flag.item()) == (engine.FLAG_BAD_USER | engine.FLAG_BAD_ITEM)
    # the two valid triplets still contributed, scaled by inv_batch = 1/4 (not 1/2)
    _, gU, gI = obpr.loss_and_grads(U, I, np.array([0, 3]), np.array([1, 2]), np.array([4, 5]))
    np.testing.assert_allclose(gradU.cpu().numpy(), gU * 0.5, rtol=1e-4, atol=1e-7)
    with pytest.raises(IndexError):
        engine.raise_on_flag(flag)


@pytest.mark.parametrize("mode", ["adam", "adamw", "adam_wd"])
def test_adam_dense(device, mode):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(11)
    n = 4 * 12345
    p = rs.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    dp, dm, dv = (torch.from_numpy(a.copy()).to(device) for a in (p, m, v))
    wd = 0.0 if mode == "adam" else 1e-2
    for step in range(1, 6):
        g = (rs.standard_normal(n) * (rs.rand(n) < 0.3)).astype(np.float32)   # mostly-zero dense grad
        dg = torch.from_numpy(g.copy()).to(device)
        oadam.adam_update(p, g, m, v, step, 1e-3, weight_decay=wd, decoupled=(mode == "adamw"))
        engine.adam_dense(dp, dg, dm, dv, step, 1e-3, weight_decay=wd, decoupled=(mode == "adamw"),
                          zero_grad=True)
        assert float(dg.abs().sum().item()) == 0.0
    np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dm.cpu().numpy(), m, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(dv.cpu().numpy(), v, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("mode", ["adam", "adamw"])
def test_adam_dense_multi_equals_single_launches(device, mode):
    """yr_adam_dense_multi (one launch for the small tensors of a model) is bit-identical to one
    yr_adam_dense per tensor: odd sizes, an empty tensor, more tensors than one launch takes."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(3)
    sizes = [1, 7, 64 * 64, 4099, 0, 128] + [33] * 14                       # 20 tensors > ADAM_MULTI_MAX
    mk = lambda n: torch.from_numpy(rs.standard_normal(n).astype(np.float32)).to(device)
    A = [(mk(n), mk(n), mk(n).abs() * 0, mk(n).abs() * 0) for n in sizes]
    B = [tuple(t.clone() for t in tup) for tup in A]
    kw = dict(weight_decay=1e-2, decoupled=(mode == "adamw"))
    for step in range(1, 4):
        for (p, g, m, v), (p2, g2, m2, v2) in zip(A, B):
            new_g = mk(p.numel())
            g.copy_(new_g); g2.copy_(new_g)
            if p.numel():
                engine.adam_dense(p, g, m, v, step, 1e-3, zero_grad=True, **kw)
        engine.adam_dense_multi(B, step, 1e-3, zero_grad=True, **kw)
    for ta, tb in zip(A, B):
        for x, y in zip(ta, tb):
            assert torch.equal(x, y)


def test_sgd_dense(device):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(12)
    n = 4096
    p = rs.standard_normal(n).astype(np.float32)
    g = rs.standard_normal(n).astype(np.float32)
    dp, dg = torch.from_numpy(p.copy()).to(device), torch.from_numpy(g.copy()).to(device)
    oadam.sgd_update(p, g, 0.05, 1e-3)
    engine.sgd_dense(dp, dg, 0.05, 1e-3)
    np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=1e-6, atol=1e-7)


def test_cpu_tensor_rejected(device):
    from yelprecommendation_amd import engine
    from yelprecommendation_amd._lib import EngineError
    with pytest.raises(EngineError):
        engine.mf_score(torch.zeros(4, 64), torch.zeros(4, 64), torch.zeros(2, dtype=torch.int64),
                        torch.zeros(2, dtype=torch.int64))


@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_mf_scores_gemm_matches_oracle(device, d):
    """f32 MFMA GEMM for evaluation vs the per-user oracle scores (asymmetric operands, ragged
    edges: rows and items not multiples of the 128x128 block)."""
    from oracle import mf_eval
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(40 + d)
    nu, ni, n = 333, 517, 201
    U, I = _tables(rs, nu, ni, d)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    S = engine.mf_scores_gemm(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device),
                              torch.from_numpy(users).to(device)).cpu().numpy()
    ref = np.stack([mf_eval.scores_for_user(U, I, int(u)) for u in users])
    np.testing.assert_allclose(S, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("fused", [True, False])
def test_mf_recommend_matches_oracle(device, fused):
    from oracle import mf_eval
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(77)
    nu, ni, d, k = 150, 700, 64, 10
    U, I = _tables(rs, nu, ni, d)
    users = np.arange(nu, dtype=np.int64)
    lists = [rs.choice(ni, size=rs.randint(0, 40), replace=False) for _ in range(nu)]
    ptr = np.zeros(nu + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    got = engine.mf_recommend(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device),
                              torch.from_numpy(users).to(device), torch.from_numpy(ptr).to(device),
                              torch.from_numpy(idx).to(device), k, chunk_users=64, fused=fused).cpu().numpy()
    want = mf_eval.recommend(U, I, users, ptr, idx, k)
    assert_topk_equal_up_to_near_ties(got, want, U, I, users, lists)     # every differing row is a near-tie
    for r in range(nu):
        assert not set(got[r].tolist()) & set(lists[r].tolist())


@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_fused_eval_topk_edge_cases(device, d):
    """Fused kernel vs the unfused GEMM + top-k path: ragged user count (not a multiple of 64), item
    count not a multiple of 128 or 32, empty mask rows, rows whose mask covers most of the catalogue
    (masked items must then fill the tail with the mask value's order), k = 1 and k = 16."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(d)
    nu, ni, n = 97, 333, 77
    U, I = _tables(rs, nu, ni, d)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    lists = []
    for r in range(n):
        m = 0 if r % 5 == 0 else (ni - 5 if r % 7 == 0 else rs.randint(1, 60))
        lists.append(np.sort(rs.choice(ni, size=m, replace=False)))
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    for k in (1, 7, 10, 12, 16) + ((20, 32) if d <= 64 else ()):              # 32-entry lists up to D = 64; 7, 12, 20:
        # k below the list length (10, 16, 32) — phantom entries keep the threshold at the k-th best
        b = engine.mf_recommend(t(U), t(I), t(users), t(ptr), t(idx), k, fused=False).cpu().numpy()
        for precision in ("bf16x3", "f32"):
            a = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, precision=precision).cpu().numpy()
            assert_topk_equal_up_to_near_ties(a, b, U, I, users, lists)
            # with the sampling launch (here it scores the whole small catalogue): identical lists, masked tails included
            p = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, precision=precision, prescan=True)
            np.testing.assert_array_equal(p.cpu().numpy(), a)


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("ni", [20, 33, 64, 70, 97, 130, 333, 4100, 16411])
def test_fused_eval_two_role_form_gives_the_same_lists(device, ni, d):
    """YR_EVAL_TWO_ROLES (eight-wave workgroups whose halves alternate between the matrix instructions and the rest of a
    tile) against the default sweep: catalogues of one, two, three tiles and of several slices, ragged row counts
    (less than one 256-row workgroup, not a multiple of it), empty / nearly full mask rows, a mask value inside the
    score range, item bias, k below and at the list lengths, prescan and hint lists: identical lists every time."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(ni + d)
    nu = 700
    U, I = _tables(rs, nu, ni, d)
    t = lambda a: torch.from_numpy(a).to(device)
    bias = t((rs.standard_normal(ni) * 0.1).astype(np.float32))
    for n in (5, 256, 300, 1100):
        users = rs.randint(0, nu, size=n).astype(np.int64)
        lists = []
        for r in range(n):
            m = 0 if r % 5 == 0 else (max(0, ni - 5) if r % 7 == 0 else rs.randint(0, min(ni, 60)))
            lists.append(np.sort(rs.choice(ni, size=m, replace=False)))
        ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
        idx = np.concatenate(lists).astype(np.int64)
        args = (t(U), t(I), t(users), t(ptr), t(idx))
        for k in (1, 4, 7, 10) + ((16,) if d == 64 else ()):                   # (D = 128: two roles up to 10 entries)
            if k >= ni:
                continue
            for kw in (dict(), dict(item_bias=bias), dict(mask_value=0.0), dict(mask_value=-1.0e30, item_bias=bias)):
                want = engine.mf_eval_topk(*args, k, prescan=False, form="four_waves", **kw)
                for extra in (dict(prescan=False), dict(prescan=True), dict(hint=want), dict(sliced=False),
                              dict(hint=t(rs.randint(-1, ni + 1, size=(n, k)).astype(np.int64)))):
                    got = engine.mf_eval_topk(*args, k, form="two_roles", **kw, **extra)
                    assert torch.equal(got, want), (ni, n, k, kw.keys(), extra.keys())


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("ni,expect_slices", [(4100, 2), (16411, 8)])
def test_fused_eval_catalogue_slices_equal_one_slice(device, ni, expect_slices, precision):
    """The fused kernel cuts the catalogue into slices (one workgroup per 128 users x slice, partial
    top-k lists merged by a second kernel) when there are few user rows: the result must be
    IDENTICAL to the one-slice form (same scores, same tie order), masks included — also when a
    user's masked items straddle slice boundaries and when a slice holds fewer than k unmasked items."""
    from yelprecommendation_amd import engine, _lib
    rs = np.random.RandomState(ni)
    nu, n, d = 300, 201, 64
    U, I = _tables(rs, nu, ni, d)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    per = -(-ni // expect_slices)
    lists = []
    for r in range(n):
        if r % 6 == 0:
            m = np.arange(per - 20, per + 20)                              # straddles the first boundary
        elif r % 6 == 1:
            m = np.setdiff1d(np.arange(per), rs.choice(per, 3, replace=False))   # slice 0 keeps 3 items
        else:
            m = np.sort(rs.choice(ni, size=rs.randint(0, 80), replace=False))
        lists.append(m.astype(np.int64))
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    lib = _lib.load()
    for k in (1, 10, 16, 20):
        planes = lib.yr_mf_eval_topk_planes_bytes(ni, d)
        assert planes == -(-ni * 6 * d // 256) * 256
        taus = -(-n * 4 // 256) * 256                                        # row thresholds of hint lists
        assert lib.yr_mf_eval_topk_workspace_bytes(n, ni, d, k, 2) == taus + n * expect_slices * k * 8
        assert lib.yr_mf_eval_topk_workspace_bytes(n, ni, d, k, 3) == planes + taus + n * expect_slices * k * 8
        gmax = -(-n * expect_slices * 128 // 256) * 256                      # prescan: 32 floats per row and part
        assert lib.yr_mf_eval_topk_workspace_bytes(n, ni, d, k, 5) == planes + taus + gmax + n * expect_slices * k * 8
        a = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, sliced=True, precision=precision).cpu().numpy()
        b = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, sliced=False, precision=precision).cpu().numpy()
        np.testing.assert_array_equal(a, b)
        for prescan in (True, False):                                          # thresholds from the sampling launch
            p = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, precision=precision, prescan=prescan)
            np.testing.assert_array_equal(p.cpu().numpy(), a)
        c = engine.mf_recommend(t(U), t(I), t(users), t(ptr), t(idx), k, fused=False).cpu().numpy()
        assert_topk_equal_up_to_near_ties(a, c, U, I, users, lists)
        for r in range(n):
            assert not set(a[r].tolist()) & set(lists[r].tolist())


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_fused_eval_with_item_bias(device, d, precision):
    """yr_mf_eval_topk_bias: scores U[u] . I[j] + bias[j] (the CDAE decoder before its sigmoid).  The bias is the
    MFMA accumulator's initial value, so the kernel must equal the bias-free kernel on tables augmented by one
    column ([u, 1] . [i, b]) — checked through the near-tie comparison against float64 scores — sliced and
    unsliced forms identical, ragged sizes, masks, a bias large enough to reorder everything."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(100 + d)
    nu, ni, n = 150, 4131, 131
    U, I = _tables(rs, nu, ni, d)
    bias = (rs.standard_normal(ni) * 0.5).astype(np.float32)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    lists = [np.sort(rs.choice(ni, size=(0 if r % 5 == 0 else rs.randint(1, 70)), replace=False)) for r in range(n)]
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    Ua, Ia = np.c_[U, np.ones(nu, np.float32)], np.c_[I, bias]              # the same scores as plain dot products
    exact = Ua[users].astype(np.float64) @ Ia.astype(np.float64).T
    for r in range(n):
        exact[r, lists[r]] = -np.inf
    for k in (4, 10, 16):
        want = np.argsort(-exact, axis=1, kind="stable")[:, :k]
        a = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, item_bias=t(bias), precision=precision).cpu().numpy()
        b = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, item_bias=t(bias), sliced=False,
                                precision=precision).cpu().numpy()
        np.testing.assert_array_equal(a, b)
        assert_topk_equal_up_to_near_ties(a, want, Ua, Ia, users, lists)
        p = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, item_bias=t(bias), precision=precision, prescan=True)
        np.testing.assert_array_equal(p.cpu().numpy(), a)
        plain = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, precision=precision).cpu().numpy()
        assert (plain != a).any()                                            # the bias matters


def test_fused_eval_mask_value_paths_agree(device):
    """-FLT_MAX (the reference's value) takes the lazy masking path of the fused kernel, any other
    value rewrites the scores before selection; with a value below every real score both must give
    the same lists, including rows where masked items have to fill the tail."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(5)
    nu, ni, n, d = 64, 2100, 50, 32
    U, I = _tables(rs, nu, ni, d)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    lists = [np.sort(rs.choice(ni, size=(ni - 4 if r % 9 == 0 else rs.randint(0, 50)), replace=False)) for r in range(n)]
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    a = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), 10).cpu().numpy()
    b = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), 10, mask_value=-1.0e30).cpu().numpy()
    for mv in (-3.40282e+38, -1.0e30):
        p = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), 10, mask_value=mv, prescan=True).cpu().numpy()
        np.testing.assert_array_equal(p, a)
    # the 4 unmasked items lead; the masked tail is ordered by item id in both (equal scores)
    np.testing.assert_array_equal(a, b)
    for r in range(0, n, 9):
        assert set(a[r, :4].tolist()) == set(range(ni)) - set(lists[r].tolist())


def test_full_catalogue_eval_properties(device):
    """BASELINE configs[1] catalogue size (38,048 items, D = 64): the fused kernel (sliced: few user
    rows) against the unfused GEMM + top-k path on 2,000 users — same lists except at float near-ties,
    masked items never recommended, lists sorted by score — and the device metrics of those lists
    against the host definition."""
    from yelprecommendation_amd import engine, metric
    g = torch.Generator(device=device).manual_seed(0)
    nu, ni, d, n, k = 31668, 38048, 64, 2000, 10
    U = torch.randn(nu, d, device=device, generator=g) * 0.1
    I = torch.randn(ni, d, device=device, generator=g) * 0.1
    users = torch.randperm(nu, device=device, generator=g)[:n]
    cnt = torch.randint(0, 60, (n,), device=device, generator=g)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=device); ptr[1:] = torch.cumsum(cnt, 0)
    idx = torch.randint(0, ni, (int(ptr[-1]),), device=device, generator=g)
    fused = engine.mf_recommend(U, I, users, ptr, idx, k, fused=True)
    plain = engine.mf_recommend(U, I, users, ptr, idx, k, fused=False)
    lists = [idx[int(ptr[r]):int(ptr[r + 1])].cpu().numpy() for r in range(n)]
    ndiff = assert_topk_equal_up_to_near_ties(fused.cpu().numpy(), plain.cpu().numpy(), U.cpu().numpy(), I.cpu().numpy(),
                                              users.cpu().numpy(), lists)
    assert ndiff <= n // 100                                                  # and near-ties are rare
    scores = (U[users] @ I.t())
    picked = scores.gather(1, fused)
    assert bool((picked[:, :-1] >= picked[:, 1:]).all())                      # descending
    rows = torch.repeat_interleave(torch.arange(n, device=device), cnt)
    hit = torch.zeros(n, ni, dtype=torch.bool, device=device); hit[rows, idx] = True
    assert not bool(hit.gather(1, fused).any())                                # never a masked item
    # every recommended score beats every unmasked, unrecommended score of its row
    rest = scores.masked_fill(hit, float("-inf")).scatter(1, fused, float("-inf"))
    assert bool((picked[:, -1] >= rest.max(1).values - 1e-6).all())
    # metrics of these lists: device vs host definition
    pos_cnt = torch.randint(0, 30, (n,), device=device, generator=g)
    pptr = torch.zeros(n + 1, dtype=torch.int64, device=device); pptr[1:] = torch.cumsum(pos_cnt, 0)
    pidx = torch.randint(0, ni, (int(pptr[-1]),), device=device, generator=g)
    got = engine.rank_metrics(fused, pptr, pidx)[:4].cpu().numpy()
    pp, pi = pptr.cpu().numpy(), pidx.cpu().numpy()
    actual = [pi[pp[r]:pp[r + 1]].tolist() for r in range(n)]
    np.testing.assert_allclose(got, metric.ranking_metrics(actual, fused.cpu().numpy().tolist(), k), rtol=1e-12)


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_fused_eval_prescan_with_a_mask_value_inside_the_score_range(device, precision):
    """Mask value 0 (CDAE's multiply-mask on outputs that are not sigmoids): masked items keep a score that can
    belong to the top k.  The prescan counts them with that value, like the sweep: lists identical with and without
    it, and equal to float64 scores with the masked entries set to 0 (up to near-ties).  Catalogue large enough for
    several slices and for the prescan to sample (stride > 1), users with most of the catalogue masked included."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(31)
    nu, ni, n, d, k = 120, 20011, 101, 64, 10
    U, I = _tables(rs, nu, ni, d)
    users = rs.randint(0, nu, size=n).astype(np.int64)
    lists = [np.sort(rs.choice(ni, size=(ni - 7 if r % 11 == 0 else rs.randint(0, 4000)), replace=False)) for r in range(n)]
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    a = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, mask_value=0.0, precision=precision,
                            prescan=True).cpu().numpy()
    b = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, mask_value=0.0, precision=precision,
                            prescan=False).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    c = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, mask_value=0.0, precision=precision).cpu().numpy()
    np.testing.assert_array_equal(c, a)                                       # the library's own rule (>= 16,384 items)
    exact = U[users].astype(np.float64) @ I.astype(np.float64).T
    for r in range(n):
        exact[r, lists[r]] = 0.0
    want = np.argsort(-exact, axis=1, kind="stable")[:, :k]
    differing = np.flatnonzero((a != want).any(axis=1))
    for r in differing:                                                       # near-ties only (f32 vs float64 scores)
        tol = 2e-5 * float(np.abs(I[np.r_[a[r], want[r]]].astype(np.float64)) @ np.abs(U[users[r]].astype(np.float64))).max() + 1e-30
        assert np.all(np.abs(np.sort(exact[r, a[r]]) - np.sort(exact[r, want[r]])) <= tol), r
    assert len(differing) <= n // 10


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("d", [16, 64, 128])
def test_fused_eval_hint_lists_never_change_the_result(device, d, precision):
    """hint = k item ids per row whose smallest score becomes the lists' starting threshold (yr_mf_eval_topk's
    `hint`).  Whatever the hint holds, the result must be the one without it: the previous result itself, the result
    of a slightly different model, random ids, rows with repeated / out-of-range / -1 ids, hints made of masked
    items (mask value -FLT_MAX and a mask value inside the score range), with an item bias, sliced and unsliced,
    and with EXACT score ties at the k-th place (duplicated item rows: the tie must still go to the smaller id)."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(900 + d)
    nu, ni, n, k = 90, 5003, 77, (20 if d == 16 else 10)          # 20: a lane of the hint kernel owns two hints
    U, I = _tables(rs, nu, ni, d)
    I[1000:1500] = I[2000:2500]                                   # 500 pairs of items with identical scores for every user
    bias = (rs.standard_normal(ni) * 0.05).astype(np.float32)
    bias[1000:1500] = bias[2000:2500]
    users = rs.randint(0, nu, size=n).astype(np.int64)
    lists = [np.sort(rs.choice(ni, size=(0 if r % 5 == 0 else rs.randint(1, 300)), replace=False)) for r in range(n)]
    ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate(lists).astype(np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    for mask_value, b in ((-3.40282e+38, None), (0.0, None), (-3.40282e+38, bias)):
        kw = dict(mask_value=mask_value, precision=precision, item_bias=None if b is None else t(b))
        want = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, **kw)
        old = engine.mf_eval_topk(t(U + 0.01 * rs.standard_normal(U.shape).astype(np.float32)), t(I), t(users), t(ptr), t(idx), k, **kw)
        junk = rs.randint(-1, ni + 2, size=(n, k)).astype(np.int64)           # -1, ni, ni + 1 and repeats occur
        junk[::3, 1] = junk[::3, 0]
        masked = np.stack([np.resize(l, k) if len(l) else np.arange(k) for l in lists]).astype(np.int64)
        for name, hint in (("own result", want), ("older model", old), ("junk", t(junk)), ("masked items", t(masked))):
            for sliced in (True, False):
                got = engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, hint=hint, sliced=sliced, **kw)
                assert torch.equal(got, want), (name, sliced, mask_value, b is not None)
        out = want.clone()                                                      # hint and out in the same buffer
        engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, hint=out, out=out, **kw)
        assert torch.equal(out, want)
    # an item row of NaNs: its scores are never candidates; as a hint it must give no bound (not a bound from k - 1 items)
    In = I.copy(); In[777] = np.nan
    want = engine.mf_eval_topk(t(U), t(In), t(users), t(ptr), t(idx), k, precision=precision)
    hint = want.clone(); hint[:, k - 1] = 777
    assert torch.equal(engine.mf_eval_topk(t(U), t(In), t(users), t(ptr), t(idx), k, precision=precision, hint=hint), want)
    with pytest.raises(Exception):
        engine.mf_eval_topk(t(U), t(I), t(users), t(ptr), t(idx), k, hint=want[:, :3].contiguous())


def test_fused_eval_accepts_an_empty_mask_index(device):
    """All mask lists empty: the index tensor has no elements (and no address); same as no masks at all."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(8)
    U, I = _tables(rs, 40, 900, 32)
    users = np.arange(40, dtype=np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    a = engine.mf_eval_topk(t(U), t(I), t(users), t(np.zeros(41, np.int64)), t(np.zeros(0, np.int64)), 10)
    b = engine.mf_eval_topk(t(U), t(I), t(users), None, None, 10)
    assert torch.equal(a, b)
    for fused in (True, False):                                                # the same through mf_recommend / topk_masked
        c = engine.mf_recommend(t(U), t(I), t(users), t(np.zeros(41, np.int64)), t(np.zeros(0, np.int64)), 10, fused=fused)
        assert_topk_equal_up_to_near_ties(c.cpu().numpy(), a.cpu().numpy(), U, I, users)
