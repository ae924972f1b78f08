"""CPU: the law of the triplet stream as stated by oracle/triplet_sampler.py (the kernel is compared with
this statement word for word in tests/test_gpu_triplets.py): reference train.py:76-77 (one permutation of
the rows per epoch) and data/datasets/mf_dataset.py:18-22 (uniform negative, redrawn while it is one of the
user's positives).  Parity unpinned against the reference's own RNG stream (NumPy / torch generators are
not reproduced); the recorded-stream replay used by the parity runs is checked here too."""
import numpy as np
import torch

from oracle import triplet_sampler as ts


def _toy(rs, nu, ni, per_user):
    rows_u = np.repeat(np.arange(nu), per_user)
    rows_i = np.concatenate([np.sort(rs.choice(ni, per_user, replace=False)) for _ in range(nu)])
    ptr = np.arange(0, nu * per_user + 1, per_user)
    return rows_u, rows_i, ptr


def test_permutation_is_a_bijection_and_changes_with_epoch_and_seed():
    for n in (1, 2, 3, 5, 16, 17, 1000, 65536, 65537, 920629):
        p = ts.permutation(n, 42, 3)
        assert np.array_equal(np.sort(p), np.arange(n)), n
    a, b, c = ts.permutation(5000, 1, 0), ts.permutation(5000, 1, 1), ts.permutation(5000, 2, 0)
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99
    # no visible structure: displacement |P(t) - t| of a random permutation has mean n/3
    assert abs(np.abs(a - np.arange(5000)).mean() / 5000 - 1 / 3) < 0.02
    # any slice on its own equals the slice of the whole
    assert np.array_equal(ts.permutation(5000, 1, 0, first=1234, count=77), a[1234:1311])


def test_negatives_avoid_the_lists_and_are_uniform_over_the_rest():
    rs = np.random.RandomState(0)
    nu, ni = 40, 50
    rows_u, rows_i, ptr = _toy(rs, nu, ni, 8)
    key = set((rows_u * ni + rows_i).tolist())
    counts = np.zeros((nu, ni))
    for epoch in range(60):
        u, p, n = ts.sample(rows_u, rows_i, ptr, rows_i, ni, seed=9, epoch=epoch)
        assert sorted((u * ni + p).tolist()) == sorted((rows_u * ni + rows_i).tolist())   # exact row multiset
        assert not any((uu * ni + nn) in key for uu, nn in zip(u.tolist(), n.tolist()))
        assert n.min() >= 0 and n.max() < ni
        np.add.at(counts, (u, n), 1)
    # per user: 480 draws over the 42 non-positives; chi-square against the uniform law, all users pooled
    free = np.ones((nu, ni), bool); free[rows_u, rows_i] = False
    exp = 60 * 8 / free.sum(1, keepdims=True)
    chi2 = (((counts - exp) ** 2 / exp) * free).sum()
    dof = free.sum() - nu
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof), (chi2, dof)
    assert counts[~free].sum() == 0


def test_slices_and_unshuffled_order_and_separate_avoid_lists():
    rs = np.random.RandomState(1)
    nu, ni = 30, 64
    rows_u, rows_i, ptr = _toy(rs, nu, ni, 5)
    full = ts.sample(rows_u, rows_i, ptr, rows_i, ni, 3, 7)
    part = ts.sample(rows_u, rows_i, ptr, rows_i, ni, 3, 7, first=40, count=33)
    for a, b in zip(full, part):
        assert np.array_equal(a[40:73], b)
    u, p, n = ts.sample(rows_u, rows_i, ptr, rows_i, ni, 3, 7, shuffle=False)
    assert np.array_equal(u, rows_u) and np.array_equal(p, rows_i)
    # valid-set form (reference mf_data_pipeline.py:47-48): the avoid lists are a superset of the rows
    extra = [np.sort(rs.choice(ni, 20, replace=False)) for _ in range(nu)]
    lists = [np.union1d(extra[k], rows_i[5 * k:5 * k + 5]) for k in range(nu)]
    aptr = np.r_[0, np.cumsum([len(l) for l in lists])]
    aidx = np.concatenate(lists)
    u, p, n = ts.sample(rows_u, rows_i, aptr, aidx, ni, 3, 7)
    assert all(nn not in set(lists[uu].tolist()) for uu, nn in zip(u.tolist(), n.tolist()))


def test_a_user_with_a_nearly_full_list_still_terminates():
    ni = 20
    rows_u, rows_i = np.zeros(19, np.int64), np.arange(19)
    u, p, n = ts.sample(rows_u, rows_i, np.array([0, 19]), rows_i, ni, 1, 0, max_draws=8)
    assert (n == 19).all()                           # the only non-positive item


def test_recorded_stream_replays_the_batches():
    from yelprecommendation_amd.data.triplets import RecordedStream
    u, p, n = np.arange(10), np.arange(10) + 100, np.arange(10) + 200
    got = list(RecordedStream(u, p, n, [4, 4, 2]))
    assert len(got) == 3 and [b["user_id"].numel() for b in got] == [4, 4, 2]
    assert torch.equal(torch.cat([b["neg_item"] for b in got]), torch.arange(200, 210))
    assert got[0]["pos_item"].dtype == torch.int64
