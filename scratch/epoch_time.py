"""End-to-end epoch timing of MFTrainer at Yelp2018 size (synthetic), phases separated."""
import sys, time
import torch
sys.path.insert(0, '.')
from yelprecommendation_amd import train as T
from yelprecommendation_amd.utils import make_config, set_seed
from yelprecommendation_amd.trainers.mf_trainer import MFTrainer
from yelprecommendation_amd.data.triplets import EpochLoader
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = make_config("MF", synthetic="yelp2018", embed_size=64, lr=1e-3, batch_size=bs, epochs=3, device="cuda",
                  model_dir="/tmp/yr_models", fast_loader=True)
t0 = time.time(); args = T.build(cfg); print(f"build (generate + split, host pandas): {time.time()-t0:.1f} s")
set_seed(cfg.seed)
dev = torch.device("cuda"); nu = args.model_info['num_users']
tl = EpochLoader(args.train_dataset.to_sampler(dev, nu, seed=cfg.seed), cfg.batch_size, cfg.shuffle)
vl = EpochLoader(args.valid_dataset.to_sampler(dev, nu, seed=cfg.seed + 1), cfg.batch_size, cfg.shuffle)
tr = MFTrainer(cfg, args.model_info['num_items'], nu)
def timed(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter()-t)*1e3:.2f} ms"); return r
for ep in range(4):
    print("epoch", ep)
    timed("train   ", lambda: tr.train(tl))
    timed("validate", lambda: tr.validate(vl))
    m = timed("evaluate", lambda: tr.evaluate(args.valid_eval_data, 'valid'))
print("metrics", m, "train rows", len(args.train_dataset))
