"""Helpers shared by the parity tests: replay a recorded batch stream; compare top-k lists up to float near-ties."""
import numpy as np
import torch


from yelprecommendation_amd.data.triplets import RecordedStream as ReplayLoader  # noqa: E402,F401  (the product's replay mode)


def epoch_slices(steps, batch_sizes):
    """[(first_batch, last_batch_excl, first_row, last_row_excl)] per epoch."""
    out, b0, r0 = [], 0, 0
    for s in steps:
        rows = int(np.sum(batch_sizes[b0:b0 + s]))
        out.append((b0, b0 + int(s), r0, r0 + rows))
        b0 += int(s)
        r0 += rows
    return out


def assert_topk_equal_up_to_near_ties(got, want, U, I, users, masked=None, rel=2e-5):
    """Two [n, k] top-k lists of the same masked score rows must be IDENTICAL, except that a row may
    differ between items whose exact (float64) scores are closer than the float32 rounding of a D-term dot
    product (rel x sum_d |u_d i_d|): then the sorted score values of both lists still agree to that
    tolerance, and each list is still in descending order to that tolerance.  No agreement quota: every
    differing row is examined.  ``masked``: optional per-row collections of excluded item ids (their
    score is the mask value: they order by id among themselves and after every real score).
    Returns the number of rows that differed."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    U, I, users = np.asarray(U, np.float64), np.asarray(I, np.float64), np.asarray(users)
    diff = np.flatnonzero((got != want).any(axis=1))
    for r in diff:
        u = U[users[r]]
        ids = np.r_[got[r], want[r]]
        tol = rel * float((np.abs(I[ids]) @ np.abs(u)).max() + 1e-30)
        m = set(np.asarray(masked[r]).tolist()) if masked is not None else set()

        def scores(row):
            s = I[row] @ u
            return np.array([-np.inf if int(i) in m else v for i, v in zip(row, s)])
        sg, sw = scores(got[r]), scores(want[r])
        fin = np.isfinite(sg)
        assert np.array_equal(fin, np.isfinite(sw)), f"row {r}: masked items in different positions"
        assert np.array_equal(got[r][~fin], want[r][~fin]), f"row {r}: masked tail differs"
        assert np.all(np.abs(np.sort(sg[fin]) - np.sort(sw[fin])) <= tol), f"row {r}: lists differ beyond a near-tie"
        assert np.all(sg[fin][:-1] >= sg[fin][1:] - tol), f"row {r}: not in descending order"
    return len(diff)
