import sys, time, os
import torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine
from yelprecommendation_amd.data.synthetic import make_interactions_torch, YELP2018_USERS as NU, YELP2018_ITEMS as NI
from yelprecommendation_amd.graph import LaplacianCSR
dev = torch.device('cuda:0')
u, i = make_interactions_torch(NU, NI, 47.0, device=dev)
r = torch.randint(1, 6, u.shape, device=dev)
graph = LaplacianCSR.from_interactions(u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), NU, NI, dev)
X = torch.randn(graph.n, 64, device=dev)
def timeit(f, n=50, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for cap in sys.argv[1:] or ["4096"]:
    os.environ["YR_SPMM_CAP"] = cap
    t = timeit(lambda: engine.spmm_csr(graph, X))
    print(f"cap {cap}: spmm {t*1e6:.1f} us  gather {graph.nnz*256/t/1e12:.2f} TB/s")
