"""Kernels of the small-batch step in isolation vs interleaved (for rocprofv3): python scratch/kernel_alone.py B mode
mode: fwd (scatter only, repeated), adam (dual Adam only), both (alternating), fwdro (forward only)"""
import sys, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine, _lib
dev = torch.device('cuda:0'); B = int(sys.argv[1]); mode = sys.argv[2]
nu, ni, d = 31668, 38048, 64
U = torch.randn(nu, d, device=dev) * 0.05; I = torch.randn(ni, d, device=dev) * 0.05
gU, gI, mU, vU, mI, vI = (torch.zeros_like(t) for t in (U, I, U, U, I, I))
u = torch.randint(0, nu, (B,), device=dev); p = torch.randint(0, ni, (B,), device=dev); n = torch.randint(0, ni, (B,), device=dev)
part = torch.zeros(2048, device=dev); loss = torch.zeros(1, device=dev); acc = torch.zeros(1, dtype=torch.float64, device=dev)
lib = _lib.load(); s = engine._stream()
def fwd(bwd=True):
    lib.yr_bpr_mf_fwd_bwd(U.data_ptr(), I.data_ptr(), u.data_ptr(), p.data_ptr(), n.data_ptr(), B, d, nu, ni, 1.0 / B,
                          gU.data_ptr() if bwd else None, gI.data_ptr() if bwd else None, part.data_ptr(), None, s)
def adam():
    lib.yr_adam_dense_dual(U.data_ptr(), gU.data_ptr(), mU.data_ptr(), vU.data_ptr(), nu * d, I.data_ptr(), gI.data_ptr(),
                           mI.data_ptr(), vI.data_ptr(), ni * d, d, None, None, 1e-4, 1e-4, 1.0, 0.9, 0.999, 1e-8, 0.0, 0,
                           part.data_ptr(), 1.0 / B, loss.data_ptr(), acc.data_ptr(), s)
for _ in range(60):
    if mode in ("fwd", "both"): fwd()
    if mode == "fwdro": fwd(False)
    if mode in ("adam", "both"): adam()
torch.cuda.synchronize()
