"""How host-bound is an MF training epoch?  MFTrainer.train over the device-side loader at Yelp2018 shape: ms per epoch and
us per step against the step's own GPU time (bench.py batch_sweep: 25 us at 32 and 4,096 triplets).
python scratch/mf_epoch_host.py [batch ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.utils import make_config
from yelprecommendation_amd.trainers import MFTrainer
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows, EpochLoader
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(4321)
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii, generator=gen) == 0
for B in [int(a) for a in sys.argv[1:]] or [32, 4096, 65536]:
    cfg = make_config("MF", embed_size=64, device="cuda", model_dir="/tmp/yr_mf_host", lr=1e-4, batch_size=B)
    trainer = MFTrainer(cfg, NI, NU)
    rows = tr if B >= 1024 else tr & (torch.cumsum(tr.long(), 0) <= 4000 * B)     # small batches: 4,000 steps' worth of rows
    loader = EpochLoader(TripletSampler(iu[rows], ii[rows], NU, NI, seed=99), batch_size=B)
    steps = len(loader)
    trainer.train(loader)
    torch.cuda.synchronize(); t = time.perf_counter()
    trainer.train(loader)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"batch {B}: {steps} steps, {dt * 1e3:.1f} ms per epoch = {dt / steps * 1e6:.1f} us per step", flush=True)
