"""Both forms of the NGCF SpMM at full graph size: python scratch/spmm_time.py [reps]"""
import sys, time
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
def timeit(f, n=reps, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
a = engine.spmm_csr(graph, X, form="rows"); b = engine.spmm_csr(graph, X, form="sliced")
print("max |rows - sliced|", float((a - b).abs().max()), "deg max", int(torch.diff(graph.rowptr).max()))
alg = graph.nnz * 8 + (graph.n + 1) * 4 + 2 * graph.n * 64 * 4
for form in ("rows", "sliced"):
    t = timeit(lambda: engine.spmm_csr(graph, X, form=form))
    print(f"{form}: spmm {t*1e6:.1f} us  gathered {graph.nnz*256/t/1e12:.2f} TB/s  algorithmic {alg/t/1e12:.2f} TB/s = {alg/t/8e12:.3f} of 8 TB/s")
