"""Interval timeline of the two-role evaluation sweep (library built with -DYR_PP_TRACE via scratch/inst_build.sh, YR_ENGINE_LIB set):
shader-clock stamps of wave 0 (role 0) and wave 4 (role 1) of one workgroup over 64 tiles.  python scratch/pp_trace.py [hint|cold] [D]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine, _lib
dev = torch.device('cuda:0')
nu, ni, d, k = 31668, 38048, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 10
g = torch.Generator(device=dev).manual_seed(0)
U = torch.randn(nu, d, device=dev, generator=g) * 0.1; I = torch.randn(ni, d, device=dev, generator=g) * 0.1
users = torch.arange(nu, device=dev)
cnt = torch.randint(10, 60, (nu,), device=dev, generator=g)
ptr = torch.zeros(nu + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(cnt, 0)
idx = torch.randint(0, ni, (int(ptr[-1]),), device=dev, generator=g)
sidx = engine.sort_mask_rows(ptr, idx)
top = engine.mf_eval_topk(U, I, users, ptr, sidx, k, form="two_roles")
for _ in range(3):
    top = engine.mf_eval_topk(U, I, users, ptr, sidx, k, form="two_roles", hint=top if len(sys.argv) > 1 and sys.argv[1] == "hint" else None)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * (2 * 64 * 8))()
lib.yr_debug_pp_trace.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.yr_debug_pp_trace.restype = ctypes.c_int
assert lib.yr_debug_pp_trace(buf) == 0
tr = np.array(buf, dtype=np.int64).reshape(2, 64, 8)
t0 = tr[0, 0, 0]
print("per tile, cycles: role 0: scores, wait barrier, rest(+stage+operands), wait barrier | role 1: the same | role 0's tile period")
for t in range(8, 40):
    a, b = tr[0, t], tr[1, t]
    print(f"tile {100 + t}: role0 start {a[0] - t0:7d}  M {a[1] - a[0]:5d} b {a[2] - a[1]:5d} E {a[3] - a[2]:5d} b {a[4] - a[3]:5d} | "
          f"role1 start {b[0] - t0:7d}  M {b[1] - b[0]:5d} b {b[2] - b[1]:5d} E {b[3] - b[2]:5d} b {b[4] - b[3]:5d} | period {tr[0, t + 1, 0] - a[0]:5d}")
m = tr[:, 8:56]
print("inside E (means per role): stash (incl. vmcnt wait)", (m[:, :, 5] - m[:, :, 2]).mean(1), "rest", (m[:, :, 6] - m[:, :, 5]).mean(1),
      "operands", (m[:, :, 7] - m[:, :, 6]).mean(1), "fetch issue", (m[:, :, 3] - m[:, :, 7]).mean(1))
print("means: M", (m[:, :, 1] - m[:, :, 0]).mean(1), "barrier after M", (m[:, :, 2] - m[:, :, 1]).mean(1), "E", (m[:, :, 3] - m[:, :, 2]).mean(1),
      "barrier after E", (m[:, :, 4] - m[:, :, 3]).mean(1), "period", (tr[0, 9:57, 0] - tr[0, 8:56, 0]).mean())
