"""Full-size timing of the NGCF (BASELINE configs[3]) and CDAE (configs[4]) steps on one MI355X."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine
from yelprecommendation_amd.data.synthetic import make_interactions_torch, YELP2018_USERS as NU, YELP2018_ITEMS as NI
from yelprecommendation_amd.graph import LaplacianCSR
from yelprecommendation_amd.loss import BPRLoss, NSBCELoss
from yelprecommendation_amd.models.ngcf import NGCF
from yelprecommendation_amd.models.cdae import CDAE
from yelprecommendation_amd.optim import Adam
from yelprecommendation_amd.utils import make_config
dev = torch.device('cuda:0')

def timeit(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

u, i = make_interactions_torch(NU, NI, 47.0, device=dev)
r = torch.randint(1, 6, u.shape, device=dev)
t0 = time.time()
graph = LaplacianCSR.from_interactions(u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), NU, NI, dev)
print(f"laplacian CSR build (host, sparse): {time.time()-t0:.1f}s  n={graph.n} nnz={graph.nnz} heavy rows={graph.n_heavy}")
for K, B in ((3, 32), (3, 4096), (3, 65536)):
    cfg = make_config("NGCF", embed_size=64, num_orders=K, device="cuda", model_dir="/tmp/m")
    model = NGCF(cfg, NU, NI).to(dev)
    with torch.no_grad(): model.embedding.weight.mul_(0.1)
    opt = Adam(model.parameters(), lr=1e-4); lossf = BPRLoss()
    bu = torch.randint(0, NU, (B,), device=dev); bp = torch.randint(0, NI, (B,), device=dev); bn = torch.randint(0, NI, (B,), device=dev)
    def step():
        pos, neg = model.bpr_forward(bu, bp, bn, graph)
        opt.zero_grad(); l = lossf(pos, neg); l.backward(); opt.step()
    dt = timeit(step)
    X = model.embedding.weight.detach()
    t_spmm = timeit(lambda: engine.spmm_csr(graph, X), 20)
    Z = engine.spmm_csr(graph, X)
    t_dense = timeit(lambda: engine.ngcf_dense_fwd(X, Z, model.W1[0].weight.detach(), model.W2[0].weight.detach()), 20)
    alg = graph.nnz * 8 + (graph.n + 1) * 4 + 2 * graph.n * 64 * 4
    print(f"NGCF K={K} B={B}: step {dt*1e3:.3f} ms -> {B/dt/1e3:.1f} k triplets/s | spmm {t_spmm*1e6:.1f} us ({alg/t_spmm/1e9:.0f} GB/s alg) | dense fwd {t_dense*1e6:.1f} us")
for B in (32, 256, 1024):
    cfg = make_config("CDAE", hidden_size=128, device="cuda", model_dir="/tmp/m", lr=1e-4)
    model = CDAE(cfg, NI, NU)
    opt = Adam(model.parameters(), lr=1e-4); lossf = NSBCELoss()
    users = torch.randperm(NU, device=dev)[:B]
    x = (torch.rand(B, NI, device=dev) < 0.0008).float()
    neg = ((torch.rand(B, NI, device=dev) < 0.004).float() * (1 - x))
    model.train()
    def step():
        pred = model(users, x)
        opt.zero_grad(); l = lossf(pred, x, neg); l.backward(); opt.step()
    dt = timeit(step)
    z = torch.rand(B, 128, device=dev)
    t_dec = timeit(lambda: engine.gemm_f32(z, model.output_layer.weight.detach(), transB=True, bias=model.output_layer.bias.detach(), act=1), 20)
    print(f"CDAE H=128 B={B}: step {dt*1e3:.3f} ms -> {B/dt:.0f} users/s | decoder GEMM {t_dec*1e6:.1f} us ({2*B*NI*128/t_dec/1e12:.2f} TFLOP/s, {(NI*128*4+B*NI*4)/t_dec/1e9:.0f} GB/s)")
