"""Helpers shared by the parity tests: replay a recorded batch stream."""
import numpy as np
import torch


class ReplayLoader:
    """Yields the recorded batches of ONE epoch, as the reference's DataLoader would
    (dict of int64 tensors), so trainers can be driven by a golden triplet stream."""

    def __init__(self, u, p, n, batch_sizes):
        self.u, self.p, self.n = (torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)) for a in (u, p, n))
        self.sizes = [int(b) for b in batch_sizes]

    def __iter__(self):
        pos = 0
        for b in self.sizes:
            s = slice(pos, pos + b)
            yield {"user_id": self.u[s], "pos_item": self.p[s], "neg_item": self.n[s]}
            pos += b

    def __len__(self):
        return len(self.sizes)


def epoch_slices(steps, batch_sizes):
    """[(first_batch, last_batch_excl, first_row, last_row_excl)] per epoch."""
    out, b0, r0 = [], 0, 0
    for s in steps:
        rows = int(np.sum(batch_sizes[b0:b0 + s]))
        out.append((b0, b0 + int(s), r0, r0 + rows))
        b0 += int(s)
        r0 += rows
    return out
