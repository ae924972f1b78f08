"""Oracle: full-catalogue scoring + masked top-k (test infrastructure — see oracle/__init__.py).

Restates reference trainers/mf_trainer.py:134-178:
  evaluate():  for every eval user, pred = model([u]*I, arange(I))   (:138-140)
  _generate_top_k_recommendation(): pred[mask_items] = -3.40282e+38 (:166-167),
      np.argpartition(pred, -top_n)[-top_n:] then descending argsort (:170-176).
Ties: the reference's order among EQUAL scores is whatever introselect/quicksort
leave; this oracle breaks ties towards the LOWER item id, and the parity tests use
continuous random weights where exact float ties between distinct items do not occur.
"""
import numpy as np

from . import metric

MASK_VALUE = np.float32(-3.40282e+38)


def scores_for_user(U, I, user_id):
    """mf_trainer.py:140 -> models/mf.py:20-23 on ([u]*I, arange(I))."""
    return np.sum(U[user_id][None, :] * I, axis=1, dtype=np.float32)


def top_k(pred, mask_items, k):
    pred = pred.copy()
    if len(mask_items):
        pred[np.asarray(mask_items, dtype=np.int64)] = MASK_VALUE
    order = np.lexsort((np.arange(pred.shape[0]), -pred))   # by score desc, id asc
    return order[:k]


def recommend(U, I, users, mask_ptr, mask_idx, k):
    """Top-k item ids for each user in ``users`` (mask given as CSR over that list)."""
    out = np.empty((len(users), k), dtype=np.int64)
    for r, u in enumerate(users):
        out[r] = top_k(scores_for_user(U, I, int(u)), mask_idx[mask_ptr[r]:mask_ptr[r + 1]], k)
    return out


def evaluate(U, I, users, pos_ptr, pos_idx, mask_ptr, mask_idx, k):
    """mf_trainer.py:134-161 -> (precision, recall, map, ndcg)@k."""
    pred = recommend(U, I, users, mask_ptr, mask_idx, k)
    actual = [list(pos_idx[pos_ptr[r]:pos_ptr[r + 1]]) for r in range(len(users))]
    predicted = [list(row) for row in pred]
    return (metric.precision_at_k(actual, predicted, k), metric.recall_at_k(actual, predicted, k),
            metric.map_at_k(actual, predicted, k), metric.ndcg_at_k(actual, predicted, k))
