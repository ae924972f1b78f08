"""Ranking metrics — host-side drop-in for reference metric.py:7-109.

Same four entry points and the same (quirky) definitions the reference's numbers
were produced with; tests pin them to the reference's known-answer tests
(test/test_metric.py:9-47) and to golden vectors:

* precision@k   = |set(actual) & set(predicted[:k])| / k, averaged over ALL users
                  (metric.py:20-24);
* recall@k      : users with an empty ``actual`` are skipped AND removed from the
                  denominator (metric.py:40-45);
* MAP@k         : at each hit position i the numerator is
                  |set(actual[:i]) & set(predicted[:i])| — the ACTUAL list is
                  truncated too — and the sum is divided by len(actual)
                  (metric.py:72-77);
* NDCG@k        : DCG scans positions 1..min(len(actual), k) only, and the ideal DCG
                  is the DCG of ``actual`` against itself (metric.py:97-109).

One pass per user computes all four (``ranking_metrics``); the reference-named
functions are thin views of it.
"""
from math import log2

_INV_LOG2 = [0.0] + [1.0 / log2(i + 1) for i in range(1, 4097)]


def _inv_log2(i: int) -> float:
    return _INV_LOG2[i] if i < len(_INV_LOG2) else 1.0 / log2(i + 1)


def _user_terms(user_actual, user_predicted, k):
    """(precision term, recall term | None, AP term | None, NDCG term | None)."""
    actual = list(user_actual)
    topk = list(user_predicted[:k])
    aset = set(actual)
    hits = len(aset & set(topk))
    prec = hits / k
    if len(actual) <= 0:
        return prec, None, None, None
    rec = hits / len(aset)
    # average precision with the reference's double truncation
    ap = 0.0
    for i in range(1, k + 1):
        if user_predicted[i - 1] in aset:
            ap += len(set(actual[:i]) & set(user_predicted[:i])) / i
    ap /= len(actual)
    # DCG over the first min(len(actual), k) positions; ideal = actual vs itself
    span = min(len(actual), k)
    dcg = sum(_inv_log2(i) for i in range(1, span + 1) if user_predicted[i - 1] in aset)
    idcg = sum(_inv_log2(i) for i in range(1, span + 1))
    return prec, rec, ap, dcg / idcg


def ranking_metrics(actual, predicted, k: int = 20):
    """(precision@k, recall@k, map@k, ndcg@k) in one pass over the users."""
    n = len(actual)
    p_sum = r_sum = a_sum = d_sum = 0.0
    n_nonempty = 0
    for u in range(n):
        prec, rec, ap, nd = _user_terms(actual[u], predicted[u], k)
        p_sum += prec
        if rec is not None:
            n_nonempty += 1
            r_sum += rec
            a_sum += ap
            d_sum += nd
    return p_sum / n, r_sum / n_nonempty, a_sum / n_nonempty, d_sum / n_nonempty


def precision_at_k(actual, predicted, k: int = 20) -> float:
    n = len(actual)
    return sum(_user_terms(actual[u], predicted[u], k)[0] for u in range(n)) / n


def recall_at_k(actual, predicted, k: int = 20) -> float:
    return ranking_metrics(actual, predicted, k)[1]


def map_at_k(actual, predicted, k: int = 20) -> float:
    return ranking_metrics(actual, predicted, k)[2]


def ndcg_at_k(actual, predicted, k: int = 20) -> float:
    return ranking_metrics(actual, predicted, k)[3]
