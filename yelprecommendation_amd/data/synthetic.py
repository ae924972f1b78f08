"""Synthetic Yelp2018-shaped interaction generator.

The reference ships no data: its input is ``data/yelp_interactions.tsv`` with
columns ``user_id, business_id, rating`` (reference data/data_preprocess.py:37,
:137-140; read back by data/datasets/mf_data_pipeline.py:69-71).  This module
produces a frame of that schema whose ids are dense ``0..U-1`` / ``0..I-1``
(the reference assumes ``nunique == max+1``: mf_data_pipeline.py:73-75) and
which is *learnable* (latent-factor + popularity model), so that a ±1e-3 band on
Recall@10 means something.

Generator: latent factors ``P[U,8], Q[I,8] ~ N(0,1)``, item popularity
``pop ~ min(Zipf(1.5), 50)``; every user draws ``max(5, Poisson(mean_items))``
distinct items from ``softmax(1.5 * P_u.Q^T + log pop)`` (Gumbel top-k, i.e.
sampling without replacement); items nobody drew are dropped and the rest are
relabelled densely; ``rating ~ U{1..5}``.
"""
from __future__ import annotations

import numpy as np

YELP2018_USERS = 31_668
YELP2018_ITEMS = 38_048
YELP2018_NNZ = 1_561_406


def make_interactions(num_users: int = YELP2018_USERS,
                      num_items: int = YELP2018_ITEMS,
                      mean_items: float = 49.0,
                      seed: int = 1234,
                      latent_dim: int = 8,
                      chunk: int = 1024,
                      min_item_degree: int = 0):
    """Return ``(user_id, business_id, rating)`` int64 arrays sorted by user.

    ``num_items`` is the size of the candidate catalogue; the returned item ids
    are relabelled to ``0..I'-1`` with ``I' <= num_items`` (items with no
    interaction vanish, as they would from the reference's 5-core TSV).
    With ``min_item_degree=5`` every candidate item is kept and topped up to
    five interactions, which reproduces the published Yelp2018 shape
    (31,668 x 38,048, ~1.56 M pairs) at ``mean_items=48``.
    """
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((num_users, latent_dim)).astype(np.float32)
    Q = rng.standard_normal((num_items, latent_dim)).astype(np.float32)
    pop = np.minimum(rng.zipf(1.5, size=num_items), 50).astype(np.float32)
    log_pop = np.log(pop)
    n_per_user = np.maximum(5, rng.poisson(mean_items, size=num_users))
    n_per_user = np.minimum(n_per_user, num_items)

    users, items = [], []
    for lo in range(0, num_users, chunk):
        hi = min(lo + chunk, num_users)
        logits = 1.5 * (P[lo:hi] @ Q.T) + log_pop[None, :]
        u = rng.random(logits.shape, dtype=np.float32)
        # Gumbel noise; clip keeps log() finite for u == 0
        np.clip(u, 1e-20, 1.0 - 1e-7, out=u)
        logits -= np.log(-np.log(u))
        kmax = int(n_per_user[lo:hi].max())
        if kmax < num_items:
            top = np.argpartition(-logits, kmax - 1, axis=1)[:, :kmax]
        else:
            top = np.tile(np.arange(num_items), (hi - lo, 1))
        top_val = np.take_along_axis(logits, top, axis=1)
        order = np.argsort(-top_val, axis=1, kind="stable")
        top = np.take_along_axis(top, order, axis=1)
        for r in range(hi - lo):
            k = int(n_per_user[lo + r])
            it = np.sort(top[r, :k])
            items.append(it)
            users.append(np.full(k, lo + r, dtype=np.int64))
    user_id = np.concatenate(users)
    raw_item = np.concatenate(items).astype(np.int64)
    if min_item_degree > 0:
        # k-core-like floor on the item side (reference data_preprocess.yaml:12
        # keeps items with >= 5 reviews): top up rare items with distinct
        # uniformly drawn users that do not hold them yet.
        deg = np.bincount(raw_item, minlength=num_items)
        seen = set(zip(user_id.tolist(), raw_item.tolist())) if (deg < min_item_degree).any() else set()
        add_u, add_i = [], []
        for it in np.nonzero(deg < min_item_degree)[0]:
            need = int(min_item_degree - deg[it])
            while need > 0:
                cand = int(rng.integers(num_users))
                if (cand, int(it)) in seen:
                    continue
                seen.add((cand, int(it)))
                add_u.append(cand)
                add_i.append(int(it))
                need -= 1
        if add_u:
            user_id = np.concatenate([user_id, np.asarray(add_u, dtype=np.int64)])
            raw_item = np.concatenate([raw_item, np.asarray(add_i, dtype=np.int64)])
            order = np.lexsort((raw_item, user_id))
            user_id, raw_item = user_id[order], raw_item[order]
    # dense relabel of the items that survived
    uniq, business_id = np.unique(raw_item, return_inverse=True)
    rating = rng.integers(1, 6, size=user_id.shape[0]).astype(np.int64)
    return user_id, business_id.astype(np.int64), rating


def make_frame(*args, **kwargs):
    """Same as :func:`make_interactions` but as the pandas frame the reference's
    ``_load_df`` yields (mf_data_pipeline.py:69-71)."""
    import pandas as pd
    u, i, r = make_interactions(*args, **kwargs)
    return pd.DataFrame({"user_id": u, "business_id": i, "rating": r})


def make_interactions_torch(num_users: int = YELP2018_USERS, num_items: int = YELP2018_ITEMS,
                            mean_items: float = 47.0, seed: int = 1234, latent_dim: int = 8,
                            device="cpu", chunk: int = 2048, min_item_degree: int = 5):
    """Same generative model as :func:`make_interactions`, on torch tensors (any device),
    for the benchmark's full-size input (seconds on the GPU instead of ~1 min of NumPy).
    Not bit-identical to the NumPy generator (different RNG); same distribution/shape.
    Returns ``(user_id, business_id)`` int64 tensors sorted by (user, item) on ``device``;
    every user has >= 5 items and every item >= ``min_item_degree`` users.
    """
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    P = torch.randn(num_users, latent_dim, generator=g, device=device)
    Q = torch.randn(num_items, latent_dim, generator=g, device=device)
    # Zipf(1.5) popularity by inverse-CDF on the Pareto tail, floored and capped at 50
    uq = torch.rand(num_items, generator=g, device=device).clamp_(1e-7, 1.0)
    pop = torch.floor(uq.pow(-1.0 / 0.5)).clamp_(1.0, 50.0)
    log_pop = pop.log()
    n_per_user = torch.poisson(torch.full((num_users,), float(mean_items), device=device), generator=g)
    n_per_user = n_per_user.clamp_(5, num_items).long()
    users, items = [], []
    for lo in range(0, num_users, chunk):
        hi = min(lo + chunk, num_users)
        logits = 1.5 * (P[lo:hi] @ Q.T) + log_pop[None, :]
        u = torch.rand(logits.shape, generator=g, device=device).clamp_(1e-20, 1.0 - 1e-7)
        logits -= torch.log(-torch.log(u))
        kmax = int(n_per_user[lo:hi].max())
        top = torch.topk(logits, kmax, dim=1).indices                       # Gumbel top-k
        keep = torch.arange(kmax, device=device)[None, :] < n_per_user[lo:hi, None]
        rows = torch.arange(lo, hi, device=device)[:, None].expand(-1, kmax)
        users.append(rows[keep])
        items.append(top[keep])
    user_id = torch.cat(users)
    item_id = torch.cat(items)
    if min_item_degree > 0:
        deg = torch.bincount(item_id, minlength=num_items)
        need = (min_item_degree - deg).clamp_(min=0)
        if int(need.sum()) > 0:
            add_i = torch.repeat_interleave(torch.arange(num_items, device=device), need)
            add_u = torch.randint(0, num_users, (add_i.numel(),), generator=g, device=device)
            user_id = torch.cat([user_id, add_u])
            item_id = torch.cat([item_id, add_i])
    key = torch.unique(user_id * num_items + item_id)                       # sorted, de-duplicated
    return torch.div(key, num_items, rounding_mode="floor"), key % num_items
