"""End-to-end CDAE epoch at Yelp2018 size with the device-side batches."""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
from yelprecommendation_amd.data.synthetic import make_interactions_torch, YELP2018_USERS as NU, YELP2018_ITEMS as NI
from yelprecommendation_amd.trainers.cdae_trainer import CDAETrainer
from yelprecommendation_amd.utils import make_config, set_seed
dev = torch.device('cuda:0')
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
u, i = make_interactions_torch(NU, NI, 47.0, device=dev)
t = time.time(); data = CDAEInteractions.from_interactions(u, i, NU, NI, seed=1, device=dev); torch.cuda.synchronize()
print(f"sparse store + split: {time.time()-t:.2f} s; train/valid/test items {[int(data.counts(p).sum()) for p in data.PARTS]}")
cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/yr_cdae", lr=1e-3, batch_size=bs, neg_times=5, top_n=10, loss_name="bce")
set_seed(1)
tr = CDAETrainer(cfg, NI, NU)
mk = lambda mode, seed: CDAEBatchLoader(data, mode, bs, cfg.neg_times, shuffle=mode != 'test', seed=seed)
def timed(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter()-t)*1e3:.1f} ms"); return r
for ep in range(3):
    print("epoch", ep, "batches", len(mk('train', 0)))
    timed("train   ", lambda: tr.train(mk('train', ep)))
    out = timed("validate", lambda: tr.validate(mk('valid', 100 + ep)))
print("valid (loss, P, R, MAP, NDCG)", out)
print("test", timed("evaluate", lambda: tr.evaluate(mk('test', 0))))
ld = mk('train', 9)
timed("loader alone (one epoch of batches)", lambda: [b for b in ld])
