"""PyTorch custom ops over the C ABI: ``torch.ops.aaerec.{step, encode, predict, predict_topk}``.

The product is ``libaaerec_hip.so`` (``include/aaerec_hip.h``); these ops are the thin torch-facing
surface BASELINE.json's north star names ("exposed to Python through PyTorch-ROCm custom ops over a
thin C-ABI").  They take the batch as plain device tensors (CSR ``indptr`` int64, ``indices`` int32,
``values`` float32, optional ``rows`` int32 permutation window) and a model id from
:func:`register_model`; all work is enqueued on torch's current stream, nothing synchronises.

    mid = ops.register_model(HipAAE(...))
    losses = torch.ops.aaerec.step(mid, indptr, indices, values, rows, 0, 100, max_row_nnz, None, [], None)
    scores = torch.ops.aaerec.predict(mid, indptr, indices, values, 0, n, max_row_nnz, None)

Reference call sites these replace: ``partial_fit`` (aae.py:745-766), ``predict`` (aae.py:840-870).
There is no CPU implementation: calling an op with host tensors raises (no fallback by design).
"""
import itertools
import weakref

import torch

from . import _hip

_LIB = torch.library.Library("aaerec", "DEF")
_CSR = "Tensor indptr, Tensor indices, Tensor values"
_LIB.define(f"step(int model, {_CSR}, Tensor? rows, int row_start, int n_rows, int max_row_nnz, Tensor? cond, "
            "Tensor?[] masks, Tensor? z_real) -> Tensor")
_LIB.define(f"encode(int model, {_CSR}, int row_start, int n_rows, int max_row_nnz) -> Tensor")
_LIB.define(f"predict(int model, {_CSR}, int row_start, int n_rows, int max_row_nnz, Tensor? cond) -> Tensor")
_LIB.define(f"predict_topk(int model, {_CSR}, int row_start, int n_rows, int max_row_nnz, Tensor? cond, int k, "
            "bool exclude_known) -> (Tensor, Tensor)")

_MODELS = weakref.WeakValueDictionary()
_IDS = itertools.count(1)


def register_model(model):
    """Make a :class:`aaerec._hip.HipAAE` addressable from the ops; returns its integer id."""
    if not isinstance(model, _hip.HipAAE):
        raise TypeError("register_model expects a HipAAE")
    mid = next(_IDS)
    _MODELS[mid] = model
    return mid


def _model(mid):
    try:
        return _MODELS[mid]
    except KeyError:
        raise RuntimeError(f"aaerec: no live model with id {mid} (register_model first; ids die with their model)")


class _CsrView:
    """What HipAAE._batch needs from a resident CSR matrix."""

    def __init__(self, indptr, indices, values, max_row_nnz, n_cols=None):
        if indptr.dtype != torch.int64 or indices.dtype != torch.int32 or values.dtype != torch.float32:
            raise TypeError("aaerec: CSR tensors must be int64 indptr, int32 indices, float32 values")
        if not (indptr.is_cuda and indices.is_cuda and values.is_cuda):
            raise RuntimeError("aaerec: CSR tensors must live on the GPU (there is no CPU path)")
        self.indptr, self.indices, self.values = indptr.contiguous(), indices.contiguous(), values.contiguous()
        # The library sizes its per-batch scratch lists from rows * max_row_nnz: a bound below the longest row would
        # let the kernels write past them, so the caller's value is checked against the true maximum, reduced on the
        # device on EVERY call (one small launch + one host sync).  (r2 cached it by (data_ptr, numel, _version): the
        # caching allocator hands a fresh per-batch indptr the same address and version, and raw-pointer writers such as
        # aae_dense_to_csr never bump the version - a stale maximum either raised for a valid call or let an undersized
        # bound through.  The training loop proper - fit() on a resident corpus - does not come through this shim.)
        # (r4 skipped the reduction when the bound was at least the vocabulary size - true for a canonical row only: the
        #  library also takes rows with duplicate (row, item) pairs (w1_update.h), which can be longer than n_cols.  ADVICE r4.)
        true_max = int((self.indptr[1:] - self.indptr[:-1]).max().item()) if self.indptr.numel() > 1 else 0
        if int(max_row_nnz) < true_max:
            raise ValueError(f"aaerec: max_row_nnz={int(max_row_nnz)} but the CSR matrix has a row of {true_max} entries")
        self.nnz_per_row_max = int(max_row_nnz)


def _step(model, indptr, indices, values, rows, row_start, n_rows, max_row_nnz, cond, masks, z_real):
    """masks (12 uint8 [rows, width] keep-masks in the reference's draw order) and z_real ([B, n_code]) are the
    injected randomness of a model created with rng_mode='inject' (parity runs); pass [] and None otherwise."""
    m = _model(model)
    if rows is not None and rows.dtype != torch.int32:
        raise TypeError("aaerec: rows must be int32")
    m.step(_CsrView(indptr, indices, values, max_row_nnz, m.N), row_start, n_rows, rows=rows, cond=cond,
           masks=list(masks) if masks else None, z_real=z_real)
    return m.tensor(_hip.T_ACT_LOSSES).reshape(-1)[:3].clone()      # (recon, disc, gen); no host sync


def _encode(model, indptr, indices, values, row_start, n_rows, max_row_nnz):
    m = _model(model)
    return m.encode(_CsrView(indptr, indices, values, max_row_nnz, m.N), row_start, n_rows)


def _predict(model, indptr, indices, values, row_start, n_rows, max_row_nnz, cond):
    m = _model(model)
    return m.predict(_CsrView(indptr, indices, values, max_row_nnz, m.N), row_start, n_rows, cond=cond)


def _predict_topk(model, indptr, indices, values, row_start, n_rows, max_row_nnz, cond, k, exclude_known):
    m = _model(model)
    return m.predict_topk(_CsrView(indptr, indices, values, max_row_nnz, m.N), row_start, n_rows, k, cond=cond,
                          exclude_known=exclude_known)


for _name, _fn in (("step", _step), ("encode", _encode), ("predict", _predict), ("predict_topk", _predict_topk)):
    _LIB.impl(_name, _fn, "CUDA")
