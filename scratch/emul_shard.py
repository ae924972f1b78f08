"""Per-rank compute time of the user-sharded step, emulated on one GPU (no collective):
rank 0 of `world` ranks, B = 2^20 local triplets."""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
from yelprecommendation_amd.user_shard import UserShard
dev = torch.device('cuda:0')
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii) == 0
B = 1 << 20
import itertools
for world, chunks in ((8, 1), (8, 2), (8, 3), (8, 4)):
    sh = UserShard(NU, world, 0)
    mine = tr & (iu >= sh.lo) & (iu < sh.hi)
    s = TripletSampler(iu[mine] - sh.lo, ii[mine], sh.size, NI, seed=1)
    su, sp, sn = s.stream(2 * B)
    pool = [(su[:B].contiguous(), sp[:B].contiguous(), sn[:B].contiguous()), (su[B:].contiguous(), sp[B:].contiguous(), sn[B:].contiguous())]
    U = torch.randn(sh.size, 64, device=dev) * 0.05; I = torch.randn(NI, 64, device=dev) * 0.05
    st = BPRMFStep(U, I, split_item_update=(world > 1), item_chunks=chunks)
    for k in range(5): st.step(*pool[k % 2], global_batch=B * world)
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(30): st.step(*pool[k % 2], global_batch=B * world, next_batch=pool[(k + 1) % 2] if world > 1 else None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
    print(f"world {world} chunks {chunks}: users/rank {sh.size}, local train rows {len(s)}, per-rank step {dt*1e6:.1f} us -> ideal aggregate {world*B/dt/1e9:.2f} G/s")
