"""End-to-end runs of train.main at Yelp2018 shape (synthetic), a few epochs each: MF (D = 64), CDAE (H = 128, the two-role
evaluation sweep), NGCF (small graph).  python scratch/train_full_smoke.py"""
import sys, time
sys.path.insert(0, '.')
from yelprecommendation_amd.train import main
for name, args in (
    ("MF full size", ["model_name=MF", "synthetic=31668x38048x47", "epochs=3", "batch_size=4096", "lr=0.005", "embed_size=64",
                      "model_dir=/tmp/yr_full_mf", "fast_loader=true"]),
    ("CDAE full size", ["model_name=CDAE", "synthetic=31668x38048x47", "epochs=3", "batch_size=256", "lr=0.001", "hidden_size=128",
                        "model_dir=/tmp/yr_full_cdae", "fast_loader=true", "neg_times=5", "loss_name=bce"]),
    ("NGCF", ["model_name=NGCF", "synthetic=3000x2500x20", "epochs=2", "batch_size=1024", "lr=0.005", "embed_size=64",
              "model_dir=/tmp/yr_full_ngcf"]),
):
    t = time.time()
    m = main(args)
    print(name, m, f"{time.time() - t:.1f} s", flush=True)
