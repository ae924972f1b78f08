"""Upper bound of what L2-resident gathers would buy the owner passes: the same step with item ids (user pass
gathers) or user ids (item pass gathers) confined to a quarter of their table."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from yelprecommendation_amd.bpr_step import BPRMFStep
NU, NI, B = 31668, 38048, 1 << 19
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for name, urange, irange in (("full tables", NU, NI), ("items in a quarter", NU, NI // 4), ("users in a quarter", NU // 4, NI),
                             ("both in a quarter", NU // 4, NI // 4)):
    u = torch.randint(0, urange, (B,), generator=g, device=dev)
    p = torch.randint(0, irange, (B,), generator=g, device=dev)
    n = torch.randint(0, irange, (B,), generator=g, device=dev)
    step = BPRMFStep(torch.randn(NU, 64, device=dev) * 0.05, torch.randn(NI, 64, device=dev) * 0.05, lr=1e-4, impl="pull",
                     time_kernels=True)
    step.auto_item_order = False
    for _ in range(10): step.step(u, p, n)
    torch.cuda.synchronize()
    for _ in range(50): step.step(u, p, n, record=True)
    kt = step.kernel_times()
    print(f"{name:22s}", {k: round(v[0], 1) for k, v in kt.items()}, flush=True)
