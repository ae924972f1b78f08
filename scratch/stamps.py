"""Timeline of the first bucket of every owner workgroup (library built with -DYR_STAMPS): python scratch/stamps.py B"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd import _lib
dev = torch.device('cuda:0'); B = int(sys.argv[1]); nu, ni, d = 31668, 38048, 64
u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
step = BPRMFStep(torch.randn(nu, d, device=dev) * 0.05, torch.randn(ni, d, device=dev) * 0.05, lr=1e-4, impl="pull")
for _ in range(20): step.step(u, p, n)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(8192 * 8, np.int64)
lib.yr_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.yr_debug_read_stamps(buf.ctypes.data, buf.size) == 0
for name, lo, nb in (("user", 0, 4096), ("item", 4096, 4096)):
    s = buf.reshape(8192, 8)[lo:lo + nb, :6].astype(np.float64)
    s = s[(s > 0).all(1)]          # owner workgroups only: helper / sizing workgroups (round 3) leave no stamps
    t0 = s[:, 0].min()
    s = (s - t0) / 100.0           # wall_clock64: 100 MHz -> us
    print(name, "pass:", len(s), "owner workgroups; first start -> last end", round(s[:, 5].max(), 2), "us; last start at", round(s[:, 0].max(), 2), "us")
    for q in (0, len(s) // 4, len(s) // 2, len(s) - 1):
        print("  wg", q, " ".join(f"{x:7.2f}" for x in s[q]))
    d = np.diff(s, axis=1)
    print("  mean phase us (start->zeroed, ->desc, ->slabs done, ->stored, ->barrier):", np.round(d.mean(0), 2),
          " per workgroup", round(float((s[:, 5] - s[:, 0]).mean()), 2), "max", round(float((s[:, 5] - s[:, 0]).max()), 2))
