"""Per-rank compute time of the user-sharded step, emulated on one GPU (no collective): rank 0 of `world`
ranks; weak scaling (B local triplets per rank) and strong scaling (one epoch split over the ranks), both
exchange forms.  python scratch/emul_shard.py [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
from yelprecommendation_amd.user_shard import UserShard
dev = torch.device('cuda:0')
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii) == 0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
epoch = int(tr.sum())
for world, chunks, form, b in ((1, 1, "all_reduce", B), (8, 1, "all_reduce", B), (8, 2, "all_reduce", B), (8, 1, "reduce_scatter", B),
                               (2, 2, "all_reduce", B), (4, 2, "all_reduce", B),
                               (8, 2, "all_reduce", epoch // 8), (8, 1, "reduce_scatter", epoch // 8)):
    sh = UserShard(NU, world, 0)
    mine = tr & (iu >= sh.lo) & (iu < sh.hi)
    s = TripletSampler(iu[mine] - sh.lo, ii[mine], sh.size, NI, seed=1)
    su, sp, sn = s.stream(2 * b)
    pool = [tuple(t[:b].contiguous() for t in (su, sp, sn)), tuple(t[b:].contiguous() for t in (su, sp, sn))]
    U = torch.randn(sh.size, 64, device=dev) * 0.05; I = torch.randn(NI, 64, device=dev) * 0.05
    st = BPRMFStep(U, I, split_item_update=(world > 1), item_chunks=chunks, item_exchange=form, impl="pull")
    for k in range(5): st.step(*pool[k % 2], global_batch=b * world)
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(50): st.step(*pool[k % 2], global_batch=b * world, next_batch=pool[(k + 1) % 2] if world > 1 else None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(f"world {world} chunks {chunks} {form}: users/rank {sh.size}, local batch {b}, per-rank step {dt*1e6:.1f} us "
          f"-> aggregate without the collective {world*b/dt/1e9:.2f} G/s", flush=True)
