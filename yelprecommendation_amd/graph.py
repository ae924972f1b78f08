"""The NGCF propagation matrix in the layout the SpMM kernel wants.

The reference hands its trainer/model a torch sparse COO tensor
``laplacian_matrix = D^-1/2 A D^-1/2`` ([N, N], N = users + items, float32; built at
data/datasets/ngcf_data_pipeline.py:19-44 through a DENSE (N, N) NumPy array — 19.4 GB at
Yelp2018 size).  ``LaplacianCSR`` converts such a tensor once (or is built straight from the
interaction triples without any dense temporary) into int32 CSR on the device plus the list of
very long rows the kernel gives a whole workgroup to.
"""
from __future__ import annotations

import numpy as np
import torch

HEAVY_THRESHOLD = 256


class LaplacianCSR:
    def __init__(self, rowptr, col, val, n, device, heavy_threshold=HEAVY_THRESHOLD, symmetric=None, split=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        if rowptr.shape[0] != n + 1 or rowptr[-1] >= 2 ** 31:
            raise ValueError("bad rowptr")
        deg = np.diff(rowptr)
        heavy = np.nonzero(deg > heavy_threshold)[0].astype(np.int32)
        self.n, self.nnz = int(n), int(rowptr[-1])
        self.heavy_threshold = int(heavy_threshold)
        self.n_heavy = int(heavy.shape[0])
        self.rowptr = torch.from_numpy(rowptr.astype(np.int32)).to(device)
        self.col = torch.from_numpy(np.ascontiguousarray(col, dtype=np.int32)).to(device)
        self.val = torch.from_numpy(np.ascontiguousarray(val, dtype=np.float32)).to(device)
        self.heavy_rows = torch.from_numpy(heavy).to(device) if self.n_heavy else None
        self.symmetric = symmetric
        # visiting order of the sliced SpMM: rows by falling degree (the longest rows start first); when the
        # bipartite split is known (``split`` = number of user nodes, which come first), each half on its
        # own, so that all workgroups gather from one half table at a time
        split = int(split) if split is not None and 0 < int(split) < n else n
        order = np.concatenate([np.argsort(-deg[:split], kind="stable"), split + np.argsort(-deg[split:], kind="stable")])
        self.row_order = torch.from_numpy(order.astype(np.int32)).to(device)

    def tiles(self, max_rows=32, max_nnz=2048):
        """int32 tile_ptr [n_tiles + 1] on the device for yr_spmm_csr_tiled: consecutive rows cut greedily so that a
        tile has at most ``max_rows`` rows and ``max_nnz`` non-zeros (rows longer than the heavy threshold count as
        empty: the kernel's first workgroups own them)."""
        cached = getattr(self, "_tiles", None)
        if cached is not None and cached[0] == (max_rows, max_nnz):
            return cached[1]
        deg = np.diff(self.rowptr.cpu().numpy().astype(np.int64))
        deg = np.where(deg > self.heavy_threshold, 0, deg)
        if deg.size and deg.max() > max_nnz:
            raise ValueError("a light row does not fit a tile")
        cum = np.concatenate([[0], np.cumsum(deg)])
        cuts, r = [0], 0
        n = self.n
        while r < n:
            hi = min(n, r + max_rows)
            # the furthest end <= hi whose non-zeros fit
            end = int(np.searchsorted(cum, cum[r] + max_nnz, side="right")) - 1
            r = max(r + 1, min(hi, end))
            cuts.append(r)
        out = torch.from_numpy(np.asarray(cuts, dtype=np.int32)).to(self.rowptr.device)
        self._tiles = ((max_rows, max_nnz), out)
        return out

    def cluster_order(self, split, clusters=8, iters=6, seed=0):
        """(row_perm int32[clusters * chunk] on the device, chunk) for yr_spmm_csr_clustered: a balanced
        co-clustering of the bipartite graph (users = nodes below ``split``).  Alternating label propagation — a
        user goes to the cluster most of its items are in, an item to the cluster most of its users are in, every
        cluster capped at 1 / clusters of each side (most decided rows first) — a few bincount passes over the
        edges at load time.  Chunk x lists the users, then the items of cluster x, padded with -1."""
        key = (int(split), int(clusters), int(iters), int(seed))
        cached = getattr(self, "_cluster_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        rp = self.rowptr.cpu().numpy().astype(np.int64)
        col = self.col.cpu().numpy().astype(np.int64)
        n, nu = self.n, int(split)
        ni, C = n - nu, int(clusters)
        rows = np.repeat(np.arange(n), np.diff(rp))
        um = rows < nu                                             # the edges user -> item (the other half mirrors them)
        eu, ei = rows[um], col[um] - nu
        if ((ei < 0) | (ei >= ni)).any():
            raise ValueError("cluster_order: the graph is not bipartite at this split")

        def assign(score, cap):
            pref = np.argsort(-score, axis=1, kind="stable")
            part = np.sort(score, axis=1)
            order = np.argsort(-(part[:, -1] - part[:, -2]), kind="stable") if C > 1 else np.arange(score.shape[0])
            out, load = np.empty(score.shape[0], np.int64), [0] * C
            for r in order.tolist():
                for c in pref[r].tolist():
                    if load[c] < cap:
                        out[r], load[c] = c, load[c] + 1
                        break
            return out

        ci = np.random.RandomState(seed).randint(0, C, ni)
        cu = np.zeros(nu, np.int64)
        for _ in range(max(1, iters)):
            cu = assign(np.bincount(eu * C + ci[ei], minlength=nu * C).reshape(nu, C).astype(np.float64), -(-nu // C))
            ci = assign(np.bincount(ei * C + cu[eu], minlength=ni * C).reshape(ni, C).astype(np.float64), -(-ni // C))
        self.cluster_intra_fraction = float((cu[eu] == ci[ei]).mean()) if eu.size else 1.0
        lists = [np.concatenate([np.flatnonzero(cu == c), nu + np.flatnonzero(ci == c)]) for c in range(C)]
        chunk = max(len(l) for l in lists)
        perm = np.full((C, chunk), -1, np.int32)
        for c, l in enumerate(lists):
            perm[c, :len(l)] = l
        out = torch.from_numpy(perm.reshape(-1)).to(self.rowptr.device)
        self._cluster_cache = (key, out, int(chunk))
        return out, int(chunk)

    @classmethod
    def from_scipy(cls, mat, device, heavy_threshold=HEAVY_THRESHOLD, split=None):
        mat = mat.tocsr()
        mat.sort_indices()
        sym = bool(abs(mat - mat.T).max() <= 1e-6 * max(abs(mat).max(), 1e-30)) if mat.nnz else True
        return cls(mat.indptr, mat.indices, mat.data, mat.shape[0], device, heavy_threshold, sym, split)

    @classmethod
    def from_torch_sparse(cls, L: torch.Tensor, device=None, heavy_threshold=HEAVY_THRESHOLD):
        """From the torch sparse COO tensor the reference's pipeline produces
        (ngcf_data_pipeline.py:38-42)."""
        import scipy.sparse as sp
        device = device if device is not None else L.device
        Lc = L.coalesce().cpu()
        idx, v = Lc.indices().numpy(), Lc.values().numpy()
        mat = sp.csr_matrix((v, (idx[0], idx[1])), shape=tuple(Lc.shape), dtype=np.float32)
        return cls.from_scipy(mat, device, heavy_threshold)

    @classmethod
    def from_interactions(cls, user_id, item_id, rating, num_users, num_items, device,
                          heavy_threshold=HEAVY_THRESHOLD):
        """Straight from the TSV columns, no dense (N, N) temporary; see :func:`laplacian_scipy`."""
        return cls.from_scipy(laplacian_scipy(user_id, item_id, rating, num_users, num_items), device,
                              heavy_threshold, split=num_users)

    def to_torch_sparse(self):
        """Back to the reference's representation (sparse COO, float32)."""
        rp = self.rowptr.cpu().numpy().astype(np.int64)
        rows = np.repeat(np.arange(self.n), np.diff(rp))
        idx = torch.from_numpy(np.stack([rows, self.col.cpu().numpy().astype(np.int64)]))
        return torch.sparse_coo_tensor(idx, self.val.cpu(), (self.n, self.n)).coalesce()


def laplacian_scipy(user_id, item_id, rating, num_users, num_items):
    """reference data/datasets/ngcf_data_pipeline.py:19-44 without the dense (N, N) arrays:
    R = pivot_table(values=rating) (MEAN over duplicate (user, item) rows, not binarised; zero
    means are dropped as to_sparse() drops them), A = [[0, R], [R^T, 0]],
    L = D^-1/2 A D^-1/2 with D = diag(column sums of A), all in float32."""
    import scipy.sparse as sp
    user_id = np.asarray(user_id, dtype=np.int64)
    item_id = np.asarray(item_id, dtype=np.int64)
    rating = np.asarray(rating, dtype=np.float64)
    key = user_id * num_items + item_id
    uniq, inv = np.unique(key, return_inverse=True)
    mean = (np.bincount(inv, weights=rating) / np.bincount(inv)).astype(np.float32)
    u, i = uniq // num_items, uniq % num_items
    keep = mean != 0
    u, i, mean = u[keep], i[keep], mean[keep]
    n = num_users + num_items
    A = sp.csr_matrix((np.concatenate([mean, mean]),
                       (np.concatenate([u, num_users + i]), np.concatenate([num_users + i, u]))),
                      shape=(n, n), dtype=np.float32)
    deg = np.asarray(A.sum(axis=0)).ravel().astype(np.float32)
    with np.errstate(divide="ignore"):
        d = (np.float32(1) / np.sqrt(deg)).astype(np.float32)
    A = A.tocoo()
    val = ((d[A.row] * A.data).astype(np.float32) * d[A.col]).astype(np.float32)
    return sp.csr_matrix((val, (A.row, A.col)), shape=(n, n), dtype=np.float32)
