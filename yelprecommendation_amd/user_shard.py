"""User sharding of the interaction matrix across the GPUs of one node.

The reference is single-process (no torch.distributed anywhere); this is the MI355X-native
addition north_star asks for.  Users are cut into ``world`` contiguous ranges; rank r owns
the user-table rows of its range, their Adam state, and every triplet whose user falls in
the range — a triplet touches exactly one user row, so the user-side gather, gradient and
update are local.  The item table and its Adam state are replicated; the only exchange per
step is one all-reduce (SUM, f32) of the dense item-embedding gradient [I, D].
"""
import numpy as np


class UserShard:
    def __init__(self, num_users: int, world_size: int = 1, rank: int = 0):
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.num_users, self.world_size, self.rank = int(num_users), int(world_size), int(rank)
        self.lo, self.hi = self.bounds(rank)
        self.size = self.hi - self.lo

    def bounds(self, rank: int):
        """[lo, hi) of ``rank``: sizes differ by at most one row, earlier ranks larger."""
        q, r = divmod(self.num_users, self.world_size)
        lo = rank * q + min(rank, r)
        return lo, lo + q + (1 if rank < r else 0)

    def owner(self, user_id):
        """Rank that owns each (global) user id; works on numpy arrays and torch tensors."""
        q, r = divmod(self.num_users, self.world_size)
        big = (q + 1) * r                      # ids below `big` live in the r larger shards
        if hasattr(user_id, "where"):          # torch
            import torch
            return torch.where(user_id < big, torch.div(user_id, q + 1, rounding_mode="floor"),
                               r + torch.div(user_id - big, max(q, 1), rounding_mode="floor"))
        user_id = np.asarray(user_id)
        return np.where(user_id < big, user_id // (q + 1), r + (user_id - big) // max(q, 1))

    def mine(self, user_id):
        return (user_id >= self.lo) & (user_id < self.hi)

    def localize(self, user_id):
        """Global user ids (all owned by this rank) -> row numbers in the local table."""
        return user_id - self.lo

    def select(self, user_id, *others):
        """Keep the triplets this rank owns; user ids come back localized."""
        m = self.mine(user_id)
        return (self.localize(user_id[m]),) + tuple(o[m] for o in others)


class ItemSlices:
    """Equal slices of the (replicated) item rows for the reduce-scatter form of the exchange: rank r
    owns rows [r * per, min(rows, (r + 1) * per)), per = ceil(rows / world) rounded up to ``granule`` rows;
    buffers that take part in the collectives are padded to world * per rows."""

    def __init__(self, rows: int, world_size: int, rank: int, granule: int = 64):
        per = -(-int(rows) // int(world_size))
        self.per = -(-per // granule) * granule
        self.rows, self.world_size, self.rank = int(rows), int(world_size), int(rank)
        self.padded = self.per * self.world_size
        self.lo = min(self.rows, self.rank * self.per)
        self.hi = min(self.rows, self.lo + self.per)


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def reduce_scatter_item_exchange(local_step, slice_update, grad_padded, param_padded, slices, group=None, overlap=None,
                                 collective=True):
    """The other form of the exchange (one chunk): every rank keeps the Adam state of ONE slice of the item
    rows only.

        local_step()                        as in sharded_item_exchange; leaves the dense local item gradient
                                            in ``grad_padded[:rows]`` (padding rows zero)
        reduce_scatter(SUM)                 rank r receives the summed gradient of ITS slice (into its
                                            own slice of ``grad_padded``)
        overlap()                           optional independent work
        slice_update()                      Adam on the rank's slice of the item rows (1/world of the
                                            dense item update and of its state traffic); must leave the
                                            updated rows in ``param_padded[lo_pad : lo_pad + per]``
        all_gather                          every rank receives every slice -> ``param_padded``

    Same wire volume as the ring all-reduce (2 (N-1)/N of the table per rank).  RCCL:
    reduce_scatter_tensor / all_gather_into_tensor; backends without reduce-scatter (gloo in the CPU
    tests) take all_reduce + the own slice, and all_gather over the slice list.
    ``collective=False`` (measurement only, see sharded_item_exchange): both collectives are skipped."""
    import torch.distributed as dist
    local_step()
    w, per, r = slices.world_size if collective else 1, slices.per, slices.rank
    mine = grad_padded[r * per:(r + 1) * per]
    if w > 1:
        if dist.get_backend(group) == "nccl":
            dist.reduce_scatter_tensor(mine, grad_padded, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(grad_padded, op=dist.ReduceOp.SUM, group=group)
    if overlap is not None:
        overlap()
    slice_update()
    if w > 1:
        own = param_padded[r * per:(r + 1) * per]
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(param_padded, own, group=group)
        else:
            parts = [param_padded[k * per:(k + 1) * per] for k in range(w)]
            dist.all_gather(parts, own.clone(), group=group)


def sharded_item_exchange(local_step, item_update, grad_item, group=None, world_size=1, overlap=None, collective=True):
    """The exchange step of the user-sharded BPR-MF step, backend-agnostic.
    Each of ``local_step``, ``item_update``, ``grad_item`` is one object or a list of C chunks
    (chunk c covers a contiguous range of item rows):

        local_step[c]()          every rank: work on ITS triplets with inv_batch = 1 / B_global; the
                                 first chunk also updates the user rows and their Adam state
                                 locally; chunk c leaves its rows of the dense local item gradient
                                 in ``grad_item[c]``
        all_reduce(grad_item[c]) SUM over ranks (RCCL over xGMI on MI355X; gloo in the CPU tests),
                                 launched asynchronously right after local_step[c] — so chunk c is
                                 on the wire while chunk c+1 is being computed
        overlap()                optional: work that does not depend on the reduced gradient, enqueued
                                 while the collectives are in flight (the next batch's index build)
        item_update[c]()         identical dense Adam on the replicated item rows of chunk c, on
                                 every rank, after chunk c's collective

    ``bpr_step.BPRMFStep`` passes HIP-kernel closures; the CPU tests pass oracle closures to check
    that the sharded protocol reproduces the single-process step.
    ``collective=False`` — MEASUREMENT ONLY (bench.py's ``collective.exposed_us``): the same launches without the
    all-reduce, so that the step's time with and without the collective can be compared; the result is then the
    rank-local update, not the step.
    """
    steps, updates, grads = _as_list(local_step), _as_list(item_update), _as_list(grad_item)
    if not (len(steps) == len(updates) == len(grads)):
        raise ValueError("local_step / item_update / grad_item must have the same number of chunks")
    works = []
    for fn, g in zip(steps, grads):
        fn()
        if world_size > 1 and collective:
            import torch.distributed as dist
            works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
    if overlap is not None:
        overlap()
    for k, upd in enumerate(updates):
        if works:
            works[k].wait()
        upd()
