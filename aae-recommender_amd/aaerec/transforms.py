"""list-of-bags <-> sparse conversions (the subset of reference aaerec/transforms.py that the
AAE path uses: lists2indices 69-87, lists2sparse 122-137, sparse2lists 45-66)."""
import numpy as np
import scipy.sparse as sp


def lists2indices(bags):
    """[[0], [1], [0, 2]] -> (row index per entry, column index per entry)."""
    rows = np.fromiter((r for r, bag in enumerate(bags) for _ in bag), dtype=np.int64)
    cols = np.fromiter((t for bag in bags for t in bag), dtype=np.int64)
    return rows, cols


def _shape_of(bags, size):
    if isinstance(size, (int, np.integer)):
        return len(bags), int(size)
    size = tuple(size)
    if len(size) == 1:
        return len(bags), int(size[0])
    if len(size) == 2:
        if len(bags) != size[0]:
            raise AssertionError("number of bags does not match size[0]")
        return int(size[0]), int(size[1])
    raise ValueError("Incorrect Shape")


def lists2sparse(bags, size):
    """COO matrix with one 1.0 per token occurrence; duplicates add up on .tocsr()
    (reference transforms.py:133-137).

    >>> lists2sparse([[0], [1], [0, 2]], (3, 3)).toarray().tolist()
    [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0, 0.0, 1.0]]
    """
    shape = _shape_of(bags, size)
    rows, cols = lists2indices(bags)
    return sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=shape)


def sparse2lists(X):
    """Inverse of lists2sparse for 0/1 matrices.

    >>> sparse2lists(sp.csr_matrix(np.array([[1, 0, 1], [0, 0, 1]])))
    [[0, 2], [2]]
    """
    X = sp.csr_matrix(X)
    return [X.indices[X.indptr[i]:X.indptr[i + 1]].tolist() for i in range(X.shape[0])]
