"""cProfile of CDAETrainer.validate over list batches at Yelp2018 size: where the ~70 us of host time per batch go."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.trainers import CDAETrainer
from yelprecommendation_amd.utils import make_config
dev = torch.device("cuda")
u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
data = CDAEInteractions.from_interactions(u, i, NU, NI, seed=1, device=dev)
cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/yr_cdae_epoch", lr=1e-4, batch_size=256,
                  negative_sampling=True, neg_times=5, loss_name="bce", top_n=10)
trainer = CDAETrainer(cfg, NI, NU)
valid = CDAEBatchLoader(data, "valid", batch_size=256, neg_times=5, seed=4, lists=True)
trainer.validate(valid); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); trainer.validate(valid); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
