"""Phase timeline of the fused evaluation kernel (library built with -DYR_ET_STAMPS by scratch/eval_phases.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd import _lib, engine
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch

dev = torch.device("cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
u, i = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
order = torch.argsort(u * NI + i)
u, i = u[order], i[order]
ptr = torch.zeros(NU + 1, dtype=torch.int64, device=dev)
ptr[1:] = torch.cumsum(torch.bincount(u, minlength=NU), 0)
U = torch.randn(NU, 64, device=dev) * 0.1
I = torch.randn(NI, 64, device=dev) * 0.1
users = torch.arange(NU, device=dev)
buf = (ctypes.c_ulonglong * 8)()
hint = None
for rep in range(3):
    lib.yr_debug_eval_phases(buf, 1)
    top = engine.mf_eval_topk(U, I, users, ptr, i.contiguous(), int(os.environ.get("YR_K", "10")),
                              precision=os.environ.get("YR_PRECISION", "bf16x3"),
                              prescan={"": None, "0": False, "1": True}[os.environ.get("YR_PRESCAN", "")], hint=hint)
    if os.environ.get("YR_HINT"):                  # the own result as the hint of the next repetition
        hint = top
    torch.cuda.synchronize()
    lib.yr_debug_eval_phases(buf, 0)
v = list(buf)
waves = v[5]
print(f"waves {waves}; per wave, shader-clock cycles: total {v[0] / waves:.0f}  mfma {v[1] / waves:.0f}  mask walk {v[6] / waves:.0f}"
      f"  epilogue(+flush) {v[2] / waves:.0f}  flush {v[3] / waves:.0f}  stash+barrier {v[4] / waves:.0f}")
