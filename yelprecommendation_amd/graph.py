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
