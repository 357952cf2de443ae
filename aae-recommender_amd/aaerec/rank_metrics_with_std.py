"""Ranking metrics with standard deviation (the functions of reference
aaerec/rank_metrics_with_std.py that METRICS uses: 13-41, 73-105, 108-131, 134-154)."""
import numpy as np


def mean_reciprocal_rank(rs, average=True):
    """1 / rank of the first relevant entry per row (0 if none); (mean, std) or the per-row array.

    >>> mean_reciprocal_rank([[0, 0, 1], [0, 1, 0], [1, 0, 0]])[0].round(4)
    0.6111
    >>> mean_reciprocal_rank(np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0]]))[0]
    0.5
    """
    rr = []
    for r in rs:
        hits = np.flatnonzero(np.asarray(r))
        rr.append(1.0 / (hits[0] + 1) if hits.size else 0.0)
    rr = np.asarray(rr, dtype=np.float64)
    return (rr.mean(), rr.std()) if average else rr


def precision_at_k(r, k):
    if k < 1:
        raise AssertionError("k must be >= 1")
    r = np.asarray(r)[:k] != 0
    if r.size != k:
        raise ValueError("Relevance score length < k")
    return float(np.mean(r))


def average_precision(r):
    """
    >>> round(average_precision([1, 1, 0, 1, 0, 1, 0, 0, 0, 1]), 4)
    0.7833
    """
    r = np.asarray(r) != 0
    hits = np.flatnonzero(r)
    if not hits.size:
        return 0.0
    return float(np.mean([r[:k + 1].mean() for k in hits]))


def mean_average_precision(rs):
    aps = np.asarray([average_precision(r) for r in rs], dtype=np.float64)
    return aps.mean(), aps.std()
