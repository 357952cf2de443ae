"""Metric side of the hot path (mirrors reference aaerec/evaluation.py): argtopk 20-58,
RankingMetric/MRR/MAP/P 61-163, METRICS 166-180, remove_non_missing 183-199, evaluate 202-240,
Evaluation 263-404 (wandb hooks dropped)."""
import os
import random
import sys
from datetime import timedelta
from timeit import default_timer as timer

import numpy as np
import scipy.sparse as sp
from sklearn.preprocessing import minmax_scale

from . import rank_metrics_with_std as rm
from .datasets import corrupt_sets
from .transforms import lists2sparse


def argtopk(X, k):
    """Index pair (rows, cols) selecting the k largest entries of every row, descending.

    >>> X = np.arange(20).reshape(2, 10)
    >>> X[argtopk(X, 3)].tolist()
    [[9, 8, 7], [19, 18, 17]]
    >>> np.arange(6).reshape(2, 3)[argtopk(np.arange(6).reshape(2, 3), 100)].tolist()
    [[2, 1, 0], [5, 4, 3]]
    """
    assert len(X.shape) == 2, "X should be two-dimensional array-like"
    rows = np.arange(X.shape[0])[:, np.newaxis]
    if k is None or k >= X.size:
        return rows, np.argsort(X, axis=1)[:, ::-1]
    assert k > 0, "k should be positive integer or None"
    part = np.argpartition(X, -k, axis=1)[:, -k:]
    order = np.argsort(X[rows, part], axis=1)[:, ::-1]
    return rows, part[rows, order]


class RankingMetric:
    """Relevance of the top-k predictions, in rank order."""

    def __init__(self, *args, **kwargs):
        self.k = kwargs.pop("k", None)

    def __call__(self, y_true, y_pred, average=True):
        return y_true[argtopk(y_pred, self.k)]


class MRR(RankingMetric):
    """
    >>> MRR(2)(np.array([[1, 0, 0], [0, 0, 1]]), np.array([[0.2, 0.3, 0.1], [0.2, 0.5, 0.7]]))
    (0.75, 0.25)
    """

    def __init__(self, k=None):
        super().__init__(k=k)

    def __call__(self, y_true, y_pred, average=True):
        return rm.mean_reciprocal_rank(super().__call__(y_true, y_pred), average=average)


class MAP(RankingMetric):
    """
    >>> MAP(2)(np.array([[1, 0, 0], [0, 0, 1]]), np.array([[0.2, 0.3, 0.1], [0.2, 0.5, 0.7]]))
    (0.75, 0.25)
    """

    def __init__(self, k=None):
        super().__init__(k=k)

    def __call__(self, y_true, y_pred, average=True):
        rs = super().__call__(y_true, y_pred)
        if average:
            return rm.mean_average_precision(rs)
        return np.array([rm.average_precision(r) for r in rs])


class P(RankingMetric):
    """
    >>> P(2)(np.array([[1, 0, 1, 0], [1, 0, 1, 0]]), np.array([[.2, .3, .1, .05], [.2, .5, .7, .05]]))
    (0.5, 0.0)
    """

    def __init__(self, k=None):
        super().__init__(k=k)

    def __call__(self, y_true, y_pred, average=True):
        ps = (super().__call__(y_true, y_pred) > 0).mean(axis=1)
        return (ps.mean(), ps.std()) if average else ps


BOUNDED_METRICS = {"{}@{}".format(M.__name__.lower(), k): M(k) for M in (MRR, MAP, P) for k in (5, 10, 20)}
BOUNDED_METRICS["P@1"] = P(1)
UNBOUNDED_METRICS = {M.__name__.lower(): M() for M in (MRR, MAP)}
METRICS = {**BOUNDED_METRICS, **UNBOUNDED_METRICS}


def remove_non_missing(Y_pred, X_test, copy=True):
    """Row-wise min-max scaling to [0,1] (sklearn.preprocessing.minmax_scale(axis=1) semantics:
    constant rows map to 0), then zero the items already present in the input.

    >>> remove_non_missing(np.array([[0.6, 0.5, -1], [40, -20, 10]]), np.array([[1, 0, 1], [0, 1, 0]])).tolist()
    [[0.0, 0.9375, 0.0], [1.0, 0.0, 0.5]]
    """
    Y = minmax_scale(Y_pred, feature_range=(0, 1), axis=1, copy=copy)
    Y[X_test.nonzero()] = 0.0
    return Y


def evaluate(ground_truth, predictions, metrics, batch_size=None):
    """[(mean, std)] per metric; batch_size bounds the dense working set."""
    n = ground_truth.shape[0]
    assert predictions.shape[0] == n
    metrics = [m if callable(m) else METRICS[m] for m in metrics]
    dense = lambda a: a.toarray() if sp.issparse(a) else a   # noqa: E731
    if batch_size is None:
        gt, pr = dense(ground_truth), dense(predictions)
        return [metric(gt, pr) for metric in metrics]
    per_metric = [[] for _ in metrics]
    for start in range(0, n, int(batch_size)):
        stop = min(start + int(batch_size), n)
        gt, pr = dense(ground_truth[start:stop, :]), dense(predictions[start:stop, :])
        for acc, metric in zip(per_metric, metrics):
            acc.extend(metric(gt, pr, average=False))
    return [(np.mean(v), np.std(v)) for v in map(np.asarray, per_metric)]


def evaluate_topk(ground_truth, topk_idx, metrics):
    """The bounded ranking metrics ('mrr@k', 'map@k', 'p@k', 'P@1') from top-k item ids alone:
    a metric at k only ever looks at the relevance of the k best predictions (RankingMetric above),
    so [(mean, std)] equals evaluate() on the full score matrix.  topk_idx: [n, K] ids, best first,
    K >= the largest k asked for; -1 entries count as irrelevant."""
    gt = sp.csr_matrix(ground_truth)
    topk_idx = np.asarray(topk_idx)
    n, K = topk_idx.shape
    assert gt.shape[0] == n
    rel = np.zeros((n, K), dtype=np.int64)
    for r in range(n):
        truth = set(gt.indices[gt.indptr[r]:gt.indptr[r + 1]].tolist())
        rel[r] = [1 if int(i) in truth else 0 for i in topk_idx[r]]
    out = []
    for name in metrics:
        metric = METRICS[name]
        k = metric.k
        if k is None or k > K:
            raise ValueError("metric {} needs the full ranking / more than the {} ids given".format(name, K))
        rs = rel[:, :k]
        if isinstance(metric, MRR):
            out.append(rm.mean_reciprocal_rank(rs))
        elif isinstance(metric, MAP):
            out.append(rm.mean_average_precision(rs))
        else:
            ps = (rs > 0).mean(axis=1)
            out.append((ps.mean(), ps.std()))
    return out


def reevaluate(gold_file, predictions_file, metrics):
    return evaluate(sp.load_npz(gold_file), np.load(predictions_file), metrics)


def maybe_open(logfile, mode="a"):
    return open(logfile, mode) if logfile else sys.stdout


def maybe_close(fh):
    if fh is not sys.stdout:
        fh.close()


class Evaluation:
    """Year split -> vocabulary -> pruning -> drop `drop` items per test bag -> train/predict/score."""

    def __init__(self, dataset, year, metrics=METRICS, logfile=sys.stdout, logdir=None, topk=True):
        self.dataset, self.year, self.metrics = dataset, year, metrics
        self.logfile, self.logdir = logfile, logdir
        # topk: a recommender that can rank on the device (predict_topk: the fused predict -> remove_non_missing -> top-k
        # pass, only [n, k] ids cross PCIe) is asked for its k best items instead of the dense [n, items] score matrix
        # whenever every requested metric is bounded at k (mrr@k, map@k, p@k, P@1: they only ever look at the k best
        # predictions - the numbers are those of the dense pipeline, tests/test_host_gpu.py) and no prediction dump (logdir)
        # is wanted.  topk=False keeps the reference's dense pipeline for every recommender.
        self.topk = topk
        self.train_set = self.test_set = self.x_test = self.y_test = None

    def setup(self, seed=42, min_elements=1, max_features=None, min_count=None, drop=1):
        self.min_elements, self.max_features, self.min_count, self.drop = min_elements, max_features, min_count, drop
        fh = maybe_open(self.logfile)
        random.seed(seed)
        np.random.seed(seed)
        train_set, test_set = self.dataset.train_test_split(on_year=self.year)
        print("=" * 80, file=fh)
        print("Train:", train_set, file=fh)
        print("Test:", test_set, file=fh)
        print("Next Pruning:\n\tmin_count: {}\n\tmax_features: {}\n\tmin_elements: {}"
              .format(min_count, max_features, min_elements), file=fh)
        train_set = train_set.build_vocab(min_count=min_count, max_features=max_features, apply=True)
        test_set = test_set.apply_vocab(train_set.vocab)
        train_set.prune_(min_elements=min_elements)
        test_set.prune_(min_elements=min_elements)
        print("Train:", train_set, file=fh)
        print("Test:", test_set, file=fh)
        print("Drop parameter:", drop, file=fh)
        noisy, missing = corrupt_sets(test_set.data, drop=drop)
        assert len(noisy) == len(missing) == len(test_set)
        test_set.data = noisy
        print("-" * 80, file=fh)
        maybe_close(fh)
        self.y_test = lists2sparse(missing, test_set.size(1)).tocsr(copy=False)
        self.x_test = lists2sparse(noisy, train_set.size(1)).tocsr(copy=False)
        self.train_set, self.test_set = train_set, test_set
        return self

    def _bounded_k(self):
        """The largest k of the requested metrics when ALL of them are bounded names (<= 32: what predict_topk ranks), else None."""
        ks = []
        for m in self.metrics:
            if not isinstance(m, str) or m not in BOUNDED_METRICS:
                return None
            ks.append(BOUNDED_METRICS[m].k)
        return max(ks) if ks and max(ks) <= 32 else None

    def __call__(self, recommenders, batch_size=None):
        if any(v is None for v in (self.train_set, self.test_set, self.x_test, self.y_test)):
            raise UserWarning("Call .setup() before running the experiment")
        if self.logdir:
            os.makedirs(self.logdir, exist_ok=True)
            with open(os.path.join(self.logdir, "vocab.txt"), "w") as vfh:
                print(*self.train_set.index2token, sep="\n", file=vfh)
            sp.save_npz(os.path.join(self.logdir, "gold"), self.y_test)
        all_results = []
        for rec in recommenders:
            fh = maybe_open(self.logfile)
            print(rec, file=fh)
            maybe_close(fh)
            train_set, test_set = self.train_set.clone(), self.test_set.clone()
            t0 = timer()
            rec.train(train_set)
            fh = maybe_open(self.logfile)
            print("Training took {} seconds.".format(timedelta(seconds=timer() - t0)), file=fh)
            t1 = timer()
            kmax = self._bounded_k()
            if self.topk and kmax is not None and not self.logdir and hasattr(rec, "predict_topk"):
                top_ids, _ = rec.predict_topk(test_set, k=kmax)
                print("Prediction took {} seconds.".format(timedelta(seconds=timer() - t1)), file=fh)
                t1 = timer()
                results = evaluate_topk(self.y_test, top_ids, list(self.metrics))
            else:
                y_pred = rec.predict(test_set)
                y_pred = y_pred.toarray() if sp.issparse(y_pred) else np.asarray(y_pred)
                y_pred = remove_non_missing(y_pred, self.x_test, copy=True)
                print("Prediction took {} seconds.".format(timedelta(seconds=timer() - t1)), file=fh)
                if self.logdir:
                    np.save(os.path.join(self.logdir, repr(rec)), y_pred)
                t1 = timer()
                results = evaluate(self.y_test, y_pred, metrics=self.metrics, batch_size=batch_size)
            print("Evaluation took {} seconds.".format(timedelta(seconds=timer() - t1)), file=fh)
            print("\nResults:\n", file=fh)
            for metric, (mean, std) in zip(self.metrics, results):
                print("- {}: {} ({})".format(metric, mean, std), file=fh)
            print("\nOverall time: {} seconds.".format(timedelta(seconds=timer() - t0)), file=fh)
            print("-" * 79, file=fh)
            maybe_close(fh)
            all_results.append(results)
        return all_results
