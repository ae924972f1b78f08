"""A longer MF run at Yelp2018 shape (synthetic): 25 epochs, batch 4,096, fast loader: the loss must fall and the metrics rise,
nothing may go non-finite (Adam step counts in the thousands).  python scratch/train_long.py"""
import sys, time, math
sys.path.insert(0, '.')
from yelprecommendation_amd.train import main
t = time.time()
m = main(["model_name=MF", "synthetic=31668x38048x47", "epochs=25", "batch_size=4096", "lr=0.002", "embed_size=64",
          "model_dir=/tmp/yr_long_mf", "fast_loader=true"])
assert all(math.isfinite(x) for x in m), m
print("MF 25 epochs", m, f"{time.time() - t:.1f} s")
