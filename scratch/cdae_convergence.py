"""CDAE end to end on a learnable synthetic set: list batches + fused step + all-user fused evaluation (and the
dense, reference-shaped route beside it): Recall@10 on the validation split should rise from chance."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
from yelprecommendation_amd.data.synthetic import make_interactions_torch
from yelprecommendation_amd.trainers import CDAETrainer
from yelprecommendation_amd.utils import make_config, set_seed

dev = torch.device("cuda")
NU, NI = 4000, 3000
u, i = make_interactions_torch(NU, NI, 40.0, seed=7, device=dev)
data = CDAEInteractions.from_interactions(u, i, NU, NI, seed=1, device=dev)
for lists in (True, False):
    set_seed(5)
    cfg = make_config("CDAE", hidden_size=64, device="cuda", model_dir="/tmp/yr_cdae_conv", lr=3e-3, batch_size=256,
                      negative_sampling=True, neg_times=5, loss_name="bce", top_n=10, fused_step=lists)
    tr = CDAETrainer(cfg, NI, NU)
    mk = lambda mode, seed: CDAEBatchLoader(data, mode, 256, 5, shuffle=mode == "train", seed=seed, lists=lists,
                                            dropout=tr.model.corruption_level)
    train, valid = mk("train", 1), mk("valid", 2)
    t0 = time.perf_counter()
    hist = []
    for epoch in range(40):
        loss = tr.train(train)
        if epoch % 5 == 4 or epoch == 0:
            v = tr.validate(valid)
            hist.append((epoch + 1, round(loss, 2), round(v[0], 2), round(v[2], 4), round(v[4], 4)))
    torch.cuda.synchronize()
    print("list batches + fused" if lists else "dense batches + autograd", f"{time.perf_counter() - t0:.2f} s")
    for h in hist:
        print("   epoch %3d  train loss %8.2f  valid loss %8.2f  recall@10 %.4f  ndcg@10 %.4f" % h)
