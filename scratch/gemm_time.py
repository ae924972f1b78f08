import sys, time
import torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine
dev = torch.device('cuda:0')
I, H = 38048, 128
def timeit(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
import os
for minb in sys.argv[1:]:
  os.environ['YR_GEMM_MINBLOCKS'] = minb
  print('minblocks', minb)
  for Bsz in (256, 1024):
      x = torch.randn(Bsz, I, device=dev); Wh = torch.randn(H, I, device=dev); Wo = torch.randn(I, H, device=dev)
      z = torch.randn(Bsz, H, device=dev); g = torch.randn(Bsz, I, device=dev); bo = torch.randn(I, device=dev)
      zz = torch.zeros(Bsz, H, device=dev)
      sk = max(1, min(256, I // 256))
      shapes = {
          "enc  x Wh^T (split-K)": lambda: engine.gemm_f32(x, Wh, transB=True, out=zz, accumulate=True, split_k=sk),
          "dec  z Wo^T +b sigm ": lambda: engine.gemm_f32(z, Wo, transB=True, bias=bo, act=1),
          "dWo  g^T z          ": lambda: engine.gemm_f32(g, z, transA=True),
          "dz   g Wo (split-K) ": lambda: engine.gemm_f32(g, Wo, split_k=sk),
          "dWh  dz^T x         ": lambda: engine.gemm_f32(z, x, transA=True),
      }
      flops = 2.0 * Bsz * I * H
      for name, f in shapes.items():
          t = timeit(f)
          print(f"B={Bsz:5d} {name}: {t*1e6:8.1f} us  {flops/t/1e12:6.2f} TFLOP/s")
      # check against torch for the decoder and dWo
      ref = torch.sigmoid(z @ Wo.t() + bo)
      print("  dec max err", (engine.gemm_f32(z, Wo, transB=True, bias=bo, act=1) - ref).abs().max().item(),
            " dWo rel err", ((engine.gemm_f32(g, z, transA=True) - g.t() @ z).abs().max() / (g.t() @ z).abs().max()).item())
  