"""Helpers shared by the parity tests: replay a recorded batch stream."""
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
