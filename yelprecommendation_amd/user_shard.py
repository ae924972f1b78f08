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


def sharded_item_exchange(local_step, item_update, grad_item, group=None, world_size=1, overlap=None):
    """The one exchange step of the user-sharded BPR-MF step, backend-agnostic:

        local_step()            every rank: forward/backward on ITS triplets with
                                inv_batch = 1 / B_global; user rows + their Adam state are updated
                                locally; the dense local item gradient lands in ``grad_item``
        all_reduce(grad_item)   SUM over ranks (RCCL over xGMI on MI355X; gloo in the CPU tests),
                                launched asynchronously
        overlap()               optional: work that does not depend on the reduced gradient, enqueued
                                while the collective is in flight (the next batch's index build)
        item_update()           identical dense Adam on the replicated item table on every rank

    ``bpr_step.BPRMFStep`` passes HIP-kernel closures; the CPU tests pass oracle closures to check
    that the sharded protocol reproduces the single-process step.
    """
    local_step()
    work = None
    if world_size > 1:
        import torch.distributed as dist
        work = dist.all_reduce(grad_item, op=dist.ReduceOp.SUM, group=group, async_op=True)
    if overlap is not None:
        overlap()
    if work is not None:
        work.wait()
    item_update()
