"""rank 0 of an 8-rank user-sharded run, emulated on one GPU without the collective (for rocprofv3):
python scratch/emul_prof.py [local batch] [steps]"""
import sys, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
from yelprecommendation_amd.user_shard import UserShard
dev = torch.device('cuda:0')
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii) == 0
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = UserShard(NU, 8, 0)
mine = tr & (iu >= sh.lo) & (iu < sh.hi)
s = TripletSampler(iu[mine] - sh.lo, ii[mine], sh.size, NI, seed=1)
su, sp, sn = s.stream(2 * b)
pool = [tuple(t[:b].contiguous() for t in (su, sp, sn)), tuple(t[b:].contiguous() for t in (su, sp, sn))]
U = torch.randn(sh.size, 64, device=dev) * 0.05; I = torch.randn(NI, 64, device=dev) * 0.05
st = BPRMFStep(U, I, split_item_update=True, impl="pull")
for k in range(steps): st.step(*pool[k % 2], global_batch=b * 8, next_batch=pool[(k + 1) % 2])
torch.cuda.synchronize()
