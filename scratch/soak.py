"""Convergence sanity at Yelp2018 size: MFTrainer for many epochs, losses must fall and Recall@10 rise."""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import train as T
from yelprecommendation_amd.utils import make_config, set_seed
from yelprecommendation_amd.trainers.mf_trainer import MFTrainer
from yelprecommendation_amd.data.triplets import EpochLoader
bs, epochs = int(sys.argv[1]), int(sys.argv[2])
cfg = make_config("MF", synthetic="yelp2018", embed_size=64, lr=float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3, batch_size=bs, epochs=epochs,
                  device="cuda", model_dir="/tmp/yr_models", fast_loader=True)
args = T.build(cfg)
set_seed(cfg.seed)
dev = torch.device("cuda"); nu = args.model_info['num_users']
tl = EpochLoader(args.train_dataset.to_sampler(dev, nu, seed=cfg.seed), cfg.batch_size, cfg.shuffle)
vl = EpochLoader(args.valid_dataset.to_sampler(dev, nu, seed=cfg.seed + 1), cfg.batch_size, cfg.shuffle)
tr = MFTrainer(cfg, args.model_info['num_items'], nu)
t0 = time.time()
for ep in range(epochs):
    tl_, vl_ = tr.train(tl), tr.validate(vl)
    if ep % max(1, epochs // 10) == 0 or ep == epochs - 1:
        p, r, m, n = tr.evaluate(args.valid_eval_data, 'valid')
        print(f"epoch {ep:4d}  train {tl_/len(tl):.4f}  valid {vl_/len(vl):.4f}  P@10 {p:.4f} R@10 {r:.4f} NDCG@10 {n:.4f}  ({time.time()-t0:.1f} s)")
