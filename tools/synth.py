"""Seeded synthetic Bags for benchmarks and scale tests.

throughput_corpus: structure-free rows - row length ~ clipped log-normal (median `median_len`,
max `max_len`), item ids Zipf(1.1) over the vocabulary without replacement inside a row,
values 1.0.  prototype_corpus: learnable structure for MRR checks (each doc = 6-9 items of one
of `n_proto` random 10-item prototype sets).
"""
import numpy as np
import scipy.sparse as sp


def throughput_corpus(n_docs, n_items, median_len=20, max_len=250, zipf_a=1.1, seed=0):
    rng = np.random.RandomState(seed)
    lens = np.clip(np.round(rng.lognormal(np.log(median_len), 0.6, size=n_docs)), 1, min(max_len, n_items)).astype(np.int64)
    p = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_a)
    cdf = np.cumsum(p / p.sum())
    indptr = np.zeros(n_docs + 1, dtype=np.int64)
    rows = []
    for d in range(n_docs):
        k = int(lens[d])
        got = np.empty(0, dtype=np.int64)
        while got.size < k:
            cand = np.searchsorted(cdf, rng.random_sample(2 * k + 8))
            got = np.unique(np.concatenate([got, np.minimum(cand, n_items - 1)]))
        if got.size > k:
            got = np.sort(rng.choice(got, size=k, replace=False))
        rows.append(got)
        indptr[d + 1] = indptr[d] + k
    indices = np.concatenate(rows).astype(np.int32)
    # items are popularity-ranked by id; shuffle ids so hot rows are spread over the table
    perm = rng.permutation(n_items).astype(np.int32)
    indices = perm[indices]
    X = sp.csr_matrix((np.ones(indices.size, dtype=np.float32), indices, indptr), shape=(n_docs, n_items))
    X.sort_indices()
    return X


def prototype_corpus(n_docs, n_items, n_proto, seed=42, hide=1):
    """Returns (X_train_like, X_input, Y_hidden) style rows: full docs; callers split."""
    rng = np.random.RandomState(seed)
    protos = [rng.choice(n_items, size=10, replace=False) for _ in range(n_proto)]
    docs = []
    for _ in range(n_docs):
        pr = protos[rng.randint(n_proto)]
        k = rng.randint(6, 10)
        docs.append(np.sort(rng.choice(pr, size=k, replace=False)))
    return docs


def docs_to_csr(docs, n_items):
    indptr = np.zeros(len(docs) + 1, dtype=np.int64)
    for i, d in enumerate(docs):
        indptr[i + 1] = indptr[i] + len(d)
    indices = np.concatenate(docs).astype(np.int32) if docs else np.zeros(0, np.int32)
    return sp.csr_matrix((np.ones(indices.size, dtype=np.float32), indices, indptr), shape=(len(docs), n_items))


def init_params(n_items, n_hidden, n_code, cond_inc=0, seed=0):
    """nn.Linear default initialisation (U(+-1/sqrt(fan_in)) for weight and bias) of the three nets, keyed like the
    reference's state_dicts ("enc.lin1.weight" ...): random-init weights for benchmarks and tools."""
    import torch
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        k = 1.0 / np.sqrt(in_f)
        return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * k).numpy(), ((torch.rand(out_f, generator=g) * 2 - 1) * k).numpy()
    shapes = {"enc.lin1": (n_hidden, n_items), "enc.lin2": (n_hidden, n_hidden), "enc.lin3": (n_code, n_hidden),
              "dec.lin1": (n_hidden, n_code + cond_inc), "dec.lin2": (n_hidden, n_hidden),
              "dec.lin3": (n_items, n_hidden), "disc.lin1": (n_hidden, n_code),
              "disc.lin2": (n_hidden, n_hidden), "disc.lin3": (1, n_hidden)}
    p = {}
    for name, (o, i) in shapes.items():
        p[name + ".weight"], p[name + ".bias"] = lin(o, i)
    return p
