"""GPU parity of the HIP triplet sampler (csrc/triplets.hip through the C ABI) with its CPU statement
oracle/triplet_sampler.py — integer work: bit-exact — plus the size-independent properties of the law of
reference train.py:76-77 / data/datasets/mf_dataset.py:18-32 at BASELINE's full size: an exact permutation
of the rows per epoch, never a positive, uniform over the non-positives, reproducible per seed."""
import numpy as np
import pytest
import torch

from oracle import triplet_sampler as ts

pytestmark = pytest.mark.gpu


def _toy(rs, nu, ni, lo, hi):
    deg = rs.randint(lo, hi + 1, size=nu)
    rows_u = np.repeat(np.arange(nu), deg)
    rows_i = np.concatenate([np.sort(rs.choice(ni, d, replace=False)) for d in deg])
    return rows_u, rows_i, np.r_[0, np.cumsum(deg)]


def _dev(device, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).to(device) for a in arrs]


@pytest.mark.parametrize("nu,ni,lo,hi", [(1, 7, 1, 1), (3, 5, 1, 4), (200, 300, 5, 40), (2000, 1500, 1, 120)])
@pytest.mark.parametrize("shuffle", [True, False])
def test_kernel_equals_cpu_statement_bit_for_bit(device, nu, ni, lo, hi, shuffle):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(nu + ni)
    rows_u, rows_i, ptr = _toy(rs, nu, ni, lo, hi)
    d = _dev(device, rows_u, rows_i, ptr, rows_i)
    flag = engine.new_error_flag(device)
    for seed, epoch in ((0, 0), (12345, 3), (2**63 + 11, 2**33 + 5)):
        want = ts.sample(rows_u, rows_i, ptr, rows_i, ni, seed, epoch, shuffle=shuffle)
        got = engine.triplet_sample(*d, nu, ni, seed, epoch, shuffle, err_flag=flag)
        for w, g in zip(want, got):
            assert np.array_equal(w, g.cpu().numpy())
        n = len(rows_u)                              # any slice on its own (a batch, a rank's share)
        first, count = n // 3, max(1, n // 2)
        if first + count <= n:
            part = engine.triplet_sample(*d, nu, ni, seed, epoch, shuffle, first, count)
            for w, g in zip(want, part):
                assert np.array_equal(w[first:first + count], g.cpu().numpy())
    assert int(flag.item()) == 0


def test_valid_set_avoid_lists_and_edge_cases(device):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(5)
    nu, ni = 60, 90
    rows_u, rows_i, ptr = _toy(rs, nu, ni, 2, 10)
    # avoid lists = rows + extra items (train + valid positives for the valid rows: mf_data_pipeline.py:47-48)
    lists = [np.union1d(rows_i[ptr[k]:ptr[k + 1]], rs.choice(ni, 30, replace=False)) for k in range(nu)]
    aptr, aidx = np.r_[0, np.cumsum([len(l) for l in lists])], np.concatenate(lists)
    want = ts.sample(rows_u, rows_i, aptr, aidx, ni, 77, 1)
    got = engine.triplet_sample(*_dev(device, rows_u, rows_i, aptr, aidx), nu, ni, 77, 1)
    for w, g in zip(want, got):
        assert np.array_equal(w, g.cpu().numpy())
    assert all(nn not in set(lists[uu].tolist()) for uu, nn in zip(want[0].tolist(), got[2].cpu().tolist()))
    # empty request
    e = engine.triplet_sample(*_dev(device, rows_u, rows_i, aptr, aidx), nu, ni, 1, 0, True, 5, 0)
    assert all(t.numel() == 0 for t in e)
    # a user whose list leaves one item: found by the fallback walk; a user whose list is the whole
    # catalogue (the reference would spin forever): flagged, item 0 emitted
    full_u, full_i = np.zeros(ni, np.int64), np.arange(ni)
    flag = engine.new_error_flag(device)
    u, p, n = engine.triplet_sample(*_dev(device, full_u[:-1], full_i[:-1], [0, ni - 1], full_i[:-1]), 1, ni, 3, 0,
                                    err_flag=flag)
    assert (n == ni - 1).all() and int(flag.item()) == 0
    u, p, n = engine.triplet_sample(*_dev(device, full_u, full_i, [0, ni], full_i), 1, ni, 3, 0, err_flag=flag)
    assert int(flag.item()) == engine.FLAG_BAD_ITEM and (n == 0).all()
    # bad row ids are flagged, not dereferenced
    flag.zero_()
    engine.triplet_sample(*_dev(device, [0, 9], [1, 2], [0, 1, 2], [1, 2]), 2, 5, 0, 0, err_flag=flag)
    assert int(flag.item()) & engine.FLAG_BAD_USER


def test_uniform_law_over_the_non_positives(device):
    """One user with 6 positives out of 50 items and 220,000 rows: every non-positive item must be drawn
    equally often (chi-square, 43 degrees of freedom)."""
    from yelprecommendation_amd import engine
    ni, n = 50, 220_000
    pos = np.array([0, 7, 8, 21, 30, 49])
    rows_i = pos[np.arange(n) % 6]
    u, p, neg = engine.triplet_sample(*_dev(device, np.zeros(n), rows_i, [0, 6], pos), 1, ni, 2024, 0)
    cnt = torch.bincount(neg, minlength=ni).cpu().numpy().astype(np.float64)
    assert cnt[pos].sum() == 0
    free = np.setdiff1d(np.arange(ni), pos)
    exp = n / len(free)
    chi2 = ((cnt[free] - exp) ** 2 / exp).sum()
    assert chi2 < 43 + 5 * np.sqrt(2 * 43), chi2
    # the positive read at position t is the row P(t): all six positives equally often, whatever the order
    assert torch.bincount(p, minlength=ni).cpu().numpy()[pos].tolist() == np.bincount(rows_i, minlength=ni)[pos].tolist()


def test_full_size_epoch_properties(device):
    """BASELINE configs[1] size (31,668 x 38,048, ~0.92 M train rows): exact row multiset per epoch, never
    a positive, reproducible per (seed, epoch), different across epochs, EpochLoader covers the epoch."""
    from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
    from yelprecommendation_amd.data.triplets import EpochLoader, TripletSampler, split_train_rows
    g = torch.Generator(device=device).manual_seed(1)
    iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=device)
    tr = split_train_rows(iu, ii, generator=g) == 0
    s = TripletSampler(iu[tr], ii[tr], NU, NI, seed=5)
    u0, p0, n0 = s.draw(0)
    assert u0.numel() == int(tr.sum()) > 900_000
    rows = torch.sort(iu[tr] * NI + ii[tr]).values
    assert torch.equal(torch.sort(u0 * NI + p0).values, rows)                    # a permutation of the rows
    k = u0 * NI + n0
    at = torch.searchsorted(rows, k).clamp_(max=rows.numel() - 1)
    assert not bool((rows[at] == k).any()) and int(n0.min()) >= 0 and int(n0.max()) < NI   # never a positive
    u0b, p0b, n0b = s.draw(0)
    assert torch.equal(u0, u0b) and torch.equal(p0, p0b) and torch.equal(n0, n0b)           # reproducible
    u1, p1, n1 = s.draw(1)
    assert float(((u1 * NI + p1) != (u0 * NI + p0)).float().mean()) > 0.99                  # another order
    s2 = TripletSampler(iu[tr], ii[tr], NU, NI, seed=6)
    assert float((s2.draw(0)[2] != n0).float().mean()) > 0.9
    # negatives are spread over the whole catalogue
    cnt = torch.bincount(n0, minlength=NI).float()
    assert float(cnt.min()) > 0 and float(cnt.std() / cnt.mean()) < 0.3
    batches = list(EpochLoader(s, 65536))
    assert len(batches) == len(EpochLoader(s, 65536)) and sum(b["user_id"].numel() for b in batches) == len(s)
    s.check()
    su, sp, sn = s.stream(2 * len(s) + 5)
    assert su.numel() == 2 * len(s) + 5


def test_sampler_from_dataset_avoids_train_and_valid_positives(device, golden_dir):
    """MFDataset.to_sampler on the VALID rows of the golden frame: negatives avoid the rows' pos_items lists
    = train + valid positives (reference mf_data_pipeline.py:47-48)."""
    import os
    import pandas as pd
    from yelprecommendation_amd.data.datasets.mf_data_pipeline import MFDataPipeline
    from yelprecommendation_amd.data.datasets.mf_dataset import MFDataset
    from yelprecommendation_amd.utils import make_config
    g = np.load(os.path.join(golden_dir, "mf_small.npz"))
    df = pd.DataFrame({"user_id": g["tsv_user"].astype(np.int64), "business_id": g["tsv_item"].astype(np.int64),
                       "rating": g["tsv_rating"].astype(np.int64)})
    pipe = MFDataPipeline(make_config("MF", seed=int(g["seed"]) if "seed" in g.files else 42))
    pipe._set_num_items_and_num_users(df)
    train, valid, _, _ = pipe.split(df)
    s = MFDataset(valid, num_items=pipe.num_items).to_sampler(device, pipe.num_users)
    u, p, n = s.epoch()
    assert u.numel() == len(valid)
    keys = set((int(a) * pipe.num_items + int(b)) for a, b in
               zip(np.r_[train.user_id.values, valid.user_id.values], np.r_[train.business_id.values, valid.business_id.values]))
    assert not any((int(a) * pipe.num_items + int(b)) in keys for a, b in zip(u.cpu().tolist(), n.cpu().tolist()))
    assert sorted((u * pipe.num_items + p).cpu().tolist()) == sorted((valid.user_id.values * pipe.num_items + valid.business_id.values).tolist())
