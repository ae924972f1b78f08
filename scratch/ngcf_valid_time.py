"""NGCFTrainer.validate at Yelp2018 size: propagate once vs per batch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import EpochLoader, TripletSampler, split_train_rows
from yelprecommendation_amd.graph import LaplacianCSR
from yelprecommendation_amd.trainers import NGCFTrainer
from yelprecommendation_amd.utils import make_config
dev = torch.device("cuda")
u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
label = split_train_rows(u, i, generator=torch.Generator(device=dev).manual_seed(1))
va = label == 1
r = torch.ones(u.numel(), device=dev)
cfg = make_config("NGCF", device="cuda", model_dir="/tmp/yr_ngcf_v", embed_size=64, num_orders=3, batch_size=4096)
graph = LaplacianCSR.from_interactions(u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), NU, NI, dev)
t = NGCFTrainer(cfg, NI, NU, graph)
loader = EpochLoader(TripletSampler(u[va], i[va], NU, NI, seed=3), 4096)
for once in (True, False):
    t.cfg.propagate_once = once
    t.validate(loader)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = t.validate(loader)
    torch.cuda.synchronize()
    print(f"propagate_once={once}: {(time.perf_counter() - t0) * 1e3:.2f} ms for {len(loader)} batches")
