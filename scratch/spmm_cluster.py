"""NGCF SpMM at full graph size: index order vs clustered order (one graph cluster per XCD).
python scratch/spmm_cluster.py [reps] [iters]      (scripts/pmc_cmd.sh spmmcl scratch/spmm_cluster.py 6 for the PMC view)"""
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
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
only = sys.argv[3] if len(sys.argv) > 3 else ""          # "index" / "clustered" / "random": that variant alone (PMC passes)
t0 = time.time()
perm, chunk = graph.cluster_order(NU, iters=iters)
print(f"clustering: {time.time() - t0:.1f} s, intra-cluster edge fraction {graph.cluster_intra_fraction:.3f}, chunk {chunk}")
import numpy as np
rnd = np.full((8, chunk), -1, np.int32)
rp = np.random.RandomState(1).permutation(graph.n)
for c in range(8):
    part = rp[c::8]
    rnd[c, :len(part)] = part
rnd = torch.from_numpy(rnd.reshape(-1)).to(dev)


def timeit(f, n=reps, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n


if not only:
    a = engine.spmm_csr(graph, X)
    b = engine.spmm_csr_clustered(graph, X, perm, chunk)
    c = engine.spmm_csr_clustered(graph, X, rnd, chunk)
    print("identical to the index-order product:", torch.equal(a, b), torch.equal(a, c))
tl = graph.tiles()
d = engine.spmm_csr_tiled(graph, X)
print("tiles:", tl.numel() - 1, "tiled identical:", torch.equal(engine.spmm_csr(graph, X), d))
acc1 = engine.spmm_csr(graph, X, out=torch.ones_like(X), accumulate=True)
acc2 = engine.spmm_csr_tiled(graph, X, out=torch.ones_like(X), accumulate=True)
print("tiled accumulate identical:", torch.equal(acc1, acc2))
for name, f in (("index order", lambda: engine.spmm_csr(graph, X)),
                ("tiled", lambda: engine.spmm_csr_tiled(graph, X)),
                ("clustered, cluster x on XCD x", lambda: engine.spmm_csr_clustered(graph, X, perm, chunk)),
                ("random rows per XCD (control)", lambda: engine.spmm_csr_clustered(graph, X, rnd, chunk))):
    if only and not name.startswith(only):
        continue
    t = timeit(f)
    print(f"{name}: {t*1e6:.1f} us  gathered {graph.nnz*256/t/1e12:.2f} TB/s")
