"""Oracle: ranking metrics (test infrastructure — see oracle/__init__.py).

Loop-for-loop restatement of reference metric.py:7-109 INCLUDING its quirks
(SURVEY.md §8 a9):
  * recall / ndcg / map skip users with empty ``actual`` and shrink the denominator
    (metric.py:40-45, :63-68, :97-102); precision does not (:20-24);
  * average precision intersects ``actual[:i]`` with ``predicted[:i]`` — it truncates
    the ACTUAL list too — and divides by ``len(actual)`` (metric.py:72-77);
  * DCG only scans positions 1..min(len(actual), k) (metric.py:106-109) and the
    ideal DCG is ``_dcg_at_k(actual, actual, k)`` (metric.py:102).
Pinned by the reference's own known-answer tests (test/test_metric.py:9-47) and by
tests/golden/metric_cases.npz.
"""
from math import log2


def precision_at_k(actual, predicted, k=20):
    n = len(actual)
    total = sum(len(set(actual[u]) & set(predicted[u][:k])) / k for u in range(n))
    return total / n


def recall_at_k(actual, predicted, k=20):
    n = len(actual)
    total = 0
    for u in range(len(actual)):
        a = set(actual[u])
        if len(a) <= 0:
            n -= 1
            continue
        total += len(a & set(predicted[u][:k])) / len(a)
    return total / n


def _average_precision_at_k(user_actual, user_predicted, k):
    s = sum(len(set(user_actual[:i]) & set(user_predicted[:i])) / i
            for i in range(1, k + 1) if user_predicted[i - 1] in user_actual)
    return s / len(user_actual)


def map_at_k(actual, predicted, k=20):
    n = len(actual)
    total = 0.0
    for u in range(len(actual)):
        if len(actual[u]) <= 0:
            n -= 1
            continue
        total += _average_precision_at_k(actual[u], predicted[u], k)
    return total / n


def _dcg_at_k(user_actual, user_predicted, k):
    return sum(1.0 / log2(i + 1) for i in range(1, min(len(user_actual), k) + 1)
               if user_predicted[i - 1] in user_actual)


def ndcg_at_k(actual, predicted, k=20):
    n = len(actual)
    total = 0.0
    for u in range(len(actual)):
        if len(set(actual[u])) <= 0:
            n -= 1
            continue
        total += _dcg_at_k(actual[u], predicted[u], k) / _dcg_at_k(actual[u], actual[u], k)
    return total / n
