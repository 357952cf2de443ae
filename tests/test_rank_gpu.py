"""GPU parity of the fused predict -> rank path (csrc/rank_x3.h, abi_rank.h; reference aae.py:840-870 +
evaluation.py:183-199 remove_non_missing + evaluation.py:20-58 argtopk): aae_predict_topk / aae_decode_topk rank hundreds
of rows per call - more than the training batch, several row blocks of 112 - without a [rows, n_items] score matrix in HBM.
Checked against the reference's host pipeline run on predict()'s dense matrix (the three-GEMM path, itself held to the
oracle in test_fuzz_gpu.py / test_fullsize_gpu.py) and against the oracle's predict directly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_topk(full, known_rows, k, exclude_known):
    """remove_non_missing + argtopk of the reference on a dense score matrix: min-max scale every row over ALL its scores,
    drop the row's input items, the k best (ties: smaller item id first)."""
    n = full.shape[0]
    ids = np.zeros((n, k), dtype=np.int64)
    vals = np.zeros((n, k), dtype=np.float32)
    for b in range(n):
        row = full[b].astype(np.float32)
        lo, hi = row.min(), row.max()
        sc = (row - lo) * (np.float32(1.0) / (hi - lo) if hi > lo else np.float32(1.0))
        rk = row.copy()
        if exclude_known:
            rk[known_rows[b]] = -np.inf
        order = np.lexsort((np.arange(rk.size), -rk))[:k]
        ids[b], vals[b] = order, sc[order]
    return ids, vals


def _corpus(r, N, n_docs, max_len):
    rows = [np.sort(r.choice(N, size=int(r.integers(1, max_len)), replace=False)) for _ in range(n_docs)]
    ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    return ip, np.concatenate(rows).astype(np.int32), np.ones(int(ip[-1]), dtype=np.float32), rows


CASES = [  # N, h, c, inc, max_batch, rows, k, exclude_known, dtype
    (5000, 200, 50, 0, 100, 500, 10, True, "f32"),        # the headline widths, 5 row blocks (112 x 4 + 52)
    (4587, 200, 50, 300, 64, 300, 10, True, "f32"),       # C4: dec.lin1 in two k-parts inside the rank call's chain program
    (3001, 100, 30, 0, 50, 113, 20, True, "f32"),         # 7 column blocks, the 20-entry lists, a ragged last tile
    (2000, 61, 20, 7, 32, 225, 32, False, "f32"),         # 4 column blocks, the 32-entry lists, nothing excluded
    (700, 200, 50, 0, 100, 37, 5, True, "f32"),           # fewer tiles than CUs, a call smaller than the training batch
    (47000, 100, 50, 0, 100, 400, 10, True, "bf16"),      # C2's shape in bf16 mode (operands rounded to bf16)
    (100000, 200, 50, 0, 100, 512, 10, True, "f32"),      # C3 at the rows-per-call the benchmark's predict_topk line uses
    (2900000, 200, 50, 0, 32, 40, 10, True, "f32"),       # PubMed's vocabulary (nmi.txt:85): dec.lin3 beyond 2^31 bytes (r5: the descriptors' window)
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused_rank_equals_host_pipeline_over_predict(case):
    _rank_case(case, CASES[case])


@pytest.mark.parametrize("act", ["GELU", "Softplus", "Hardswish"])
def test_rank_path_of_a_model_with_a_further_activation_class(act):
    """r6: a model with one of the activation classes 6-19 runs its layer programs on the general chain kernel and has no fused
    ranking launch - aae_predict_topk / aae_decode_topk take their two-kernel form (the [rows, N] score matrix + scan): same
    contract, same checks."""
    _rank_case(40 + len(act), (3001, 100, 30, 0, 50, 113, 10, True, "f32"), activation=act)


def _rank_case(case, params_of_case, activation="ReLU"):
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import init_params
    N, h, c, inc, R, rows, k, excl, dtype = params_of_case
    r = np.random.default_rng(100 + case)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3, activation=activation, **({"dtype": "bf16"} if dtype == "bf16" else {}))
    dev = HipAAE(N, h, c, cond_inc=inc, max_batch=R, rng_mode="device", seed=3, **kw)
    params = init_params(N, h, c, cond_inc=inc, seed=case)
    # spread the logits (nn.Linear's initialisation leaves every sigmoid near 0.5): scale the output layer
    params["dec.lin3.weight"] = params["dec.lin3.weight"] * 8.0
    dev.load_params(params)
    ip, idx, val, docs = _corpus(r, N, rows, 30)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    cond = (r.standard_normal((rows, inc)) * 0.4).astype(np.float32) if inc else None
    cdev = torch.as_tensor(cond, device=dev.device) if inc else None
    # a few training steps first: enc.lin1 rows with deferred Adam steps pending, a deferred optimiser launch in flight
    for s in range(3):
        dev.step(csr, s * min(R, rows // 3), min(R, rows // 3), cond=None if cdev is None else cdev[s * min(R, rows // 3):(s + 1) * min(R, rows // 3)])
    cap = dev.rank_max_rows(k)          # (what the arena's scratch holds: [rows][workgroups][k] candidates dominate on tiny models)
    if activation == "ReLU":             # (a model without the fused launch ranks max_batch rows per call)
        assert cap > R and cap >= min(rows, 128), (cap, R, rows)
    else:
        assert cap == R
    rows = min(rows, cap)
    docs = docs[:rows]
    cdev = None if cdev is None else cdev[:rows].contiguous()
    ids, vals = dev.predict_topk(csr, 0, rows, k, cond=cdev, exclude_known=excl)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    # the dense matrix, max_batch rows at a time (aae_predict: the streaming GEMM + sigmoid)
    full = np.concatenate([dev.predict(csr, s, min(R, rows - s), cond=None if cdev is None else cdev[s:s + R]).cpu().numpy()
                           for s in range(0, rows, R)])
    want_ids, want_vals = _host_topk(full, docs, k, excl)
    tol = 2e-6 if dtype == "f32" else 2e-3
    np.testing.assert_allclose(vals, want_vals, atol=tol)
    b, j = np.nonzero(ids != want_ids)
    # positions may differ only where scores tie within the two paths' summation-order difference
    lo, hi = full.min(1), full.max(1)
    scaled = (full - lo[:, None]) / np.where(hi > lo, hi - lo, 1.0)[:, None]
    assert np.all(np.abs(scaled[b, ids[b, j]] - scaled[b, want_ids[b, j]]) <= tol), (case, len(b))
    for row in range(rows):
        assert len(set(ids[row].tolist())) == k
        if excl:
            assert not (set(ids[row].tolist()) & set(docs[row].tolist()))
    # the same through aae_decode_topk (a caller-built decoder input: predict's second half behind any condition plugin)
    z = torch.cat([dev.encode(csr, s, min(R, rows - s)) for s in range(0, rows, R)])
    zc = z if cdev is None else torch.cat([z, cdev], 1)
    ids2, vals2 = dev.decode_topk(zc, csr, 0, k, exclude_known=excl)
    np.testing.assert_allclose(vals2.cpu().numpy(), vals, atol=tol)
    d = ids2.cpu().numpy() != ids
    assert np.all(np.abs(vals2.cpu().numpy()[d] - vals[d]) <= tol)


def test_saturated_scores_tie_and_both_rank_paths_return_a_valid_top_k():
    """for a trained model many of the best logits saturate - sigmoid(x) == 1.0f from x ~ 17 on, and neighbouring
    logits collide in fp32 long before that.  The reference ranks the fp32 sigmoid outputs and leaves ties in whatever order
    np.argpartition produces (evaluation.py:20-58: "ties unordered"); here the dense path (aae_predict + scan, and the host
    pipeline) breaks a tie by the smaller item id, the fused path (rank_x3.h) orders its candidates by LOGIT and applies the
    sigmoid to the winners - a refinement of the same order.  So the two may name different items exactly where their scores
    tie, never elsewhere: with dec.lin3 scaled by 600 (hundreds of saturated items per row) the k scaled SCORES are identical,
    every named item's score is the score of its rank, and no known item is named."""
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import init_params
    N, h, c, R, rows, k = 6000, 200, 50, 100, 160, 10
    r = np.random.default_rng(77)
    dev = HipAAE(N, h, c, max_batch=R, rng_mode="device", seed=5, dropout=(0.0, 0.0))
    params = init_params(N, h, c, seed=9)
    params["dec.lin3.weight"] = params["dec.lin3.weight"] * 600.0
    params["dec.lin2.weight"] = params["dec.lin2.weight"] * 4.0
    dev.load_params(params)
    ip, idx, val, docs = _corpus(r, N, rows, 30)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    ids, vals = dev.predict_topk(csr, 0, rows, k)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    full = np.concatenate([dev.predict(csr, s, min(R, rows - s)).cpu().numpy() for s in range(0, rows, R)])
    saturated = (full == 1.0).sum(1)
    assert np.median(saturated) >= k, ("the case is meant to tie at the top", saturated[:10])
    want_ids, want_vals = _host_topk(full, docs, k, True)
    np.testing.assert_allclose(vals, want_vals, atol=2e-6)
    lo, hi = full.min(1), full.max(1)
    scaled = (full - lo[:, None]) / np.where(hi > lo, hi - lo, 1.0)[:, None]
    got_scores = np.take_along_axis(scaled, ids.astype(np.int64), axis=1)
    np.testing.assert_allclose(got_scores, want_vals, atol=2e-6)          # a named item holds the score of its rank
    differ = int((ids != want_ids).sum())
    print("saturated scores per row (median):", int(np.median(saturated)), "| positions where the two paths name different items:", differ, "of", ids.size)
    for row in range(rows):
        assert len(set(ids[row].tolist())) == k and not (set(ids[row].tolist()) & set(docs[row].tolist()))


def test_fused_rank_matches_oracle_predict_and_chunked_calls_agree():
    """Against the oracle's eval-mode predict (not only against the library's own GEMM path), and: one 300-row call ==
    three 100-row calls == the old two-kernel path (AAE_NO_RANK_FUSED: GEMM + sigmoid into HBM, then a scan per row)."""
    import os
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    N, h, c, R, rows, k = 1500, 64, 24, 100, 300, 10
    r = np.random.default_rng(5)
    params = init_params(N, h, c, seed=2)
    params["dec.lin3.weight"] = params["dec.lin3.weight"] * 6.0
    kw = dict(dropout=(0.0, 0.0), gen_lr=1e-3, reg_lr=1e-3)
    dev = HipAAE(N, h, c, max_batch=R, rng_mode="inject", **kw)
    dev.load_params(params)
    ora = O.OracleAAE(params, **kw)
    ip, idx, val, docs = _corpus(r, N, rows, 12)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    want = ora.predict(ip, idx, val)
    want_ids, want_vals = _host_topk(want, docs, k, True)
    ids, vals = dev.predict_topk(csr, 0, rows, k)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    np.testing.assert_allclose(vals, want_vals, atol=1e-5)
    lo, hi = want.min(1), want.max(1)
    scaled = (want - lo[:, None]) / (hi - lo)[:, None]
    b, j = np.nonzero(ids != want_ids)
    assert np.all(np.abs(scaled[b, ids[b, j]] - scaled[b, want_ids[b, j]]) <= 1e-5)
    parts = [dev.predict_topk(csr, s, 100, k) for s in range(0, rows, 100)]
    assert np.array_equal(torch.cat([p[0] for p in parts]).cpu().numpy(), ids)
    np.testing.assert_array_equal(torch.cat([p[1] for p in parts]).cpu().numpy(), vals)
    os.environ["AAE_NO_RANK_FUSED"] = "1"
    try:
        old = HipAAE(N, h, c, max_batch=R, rng_mode="inject", **kw)
    finally:
        del os.environ["AAE_NO_RANK_FUSED"]
    old.load_params(params)
    assert old.rank_max_rows(k) == R
    parts = [old.predict_topk(csr, s, 100, k) for s in range(0, rows, 100)]
    oi, ov = torch.cat([p[0] for p in parts]).cpu().numpy(), torch.cat([p[1] for p in parts]).cpu().numpy()
    np.testing.assert_allclose(ov, vals, atol=2e-6)
    d = oi != ids
    assert np.all(np.abs(ov[d] - vals[d]) <= 2e-6)
    with pytest.raises(RuntimeError):
        old.predict_topk(csr, 0, rows, k)               # beyond max_batch without the fused path: refused, not truncated


@pytest.mark.parametrize("N,h,c,R,rows,k", [(33, 8, 3, 4, 3, 5), (40, 200, 50, 2, 1, 32), (1000, 32, 8, 16, 1500, 10), (64, 5, 2, 100, 100, 1),
                                             (20000, 120, 20, 10, 9, 20)])
def test_fused_rank_degenerate_shapes(N, h, c, R, rows, k):
    """Edges of the fused rank path: a vocabulary of one (ragged) tile, one row, k = 1 and k = 32, far more rows than the
    training batch on a small vocabulary (14 row blocks, fewer workgroups per block than tiles), rows whose input is most
    of the vocabulary (fewer than k items left to rank: the tail of the list is -1 / 0.0 as in the two-kernel path)."""
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import init_params
    r = np.random.default_rng(N + rows)
    dev = HipAAE(N, h, c, max_batch=R, rng_mode="device", seed=1)
    params = init_params(N, h, c, seed=3)
    params["dec.lin3.weight"] = params["dec.lin3.weight"] * 6.0
    dev.load_params(params)
    cap = dev.rank_max_rows(k)
    rows = min(rows, cap)
    max_len = N - 2 if N <= 64 else 30            # (small vocabularies: rows that name nearly every item)
    ip, idx, val, docs = _corpus(r, N, rows, max_len)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    ids, vals = dev.predict_topk(csr, 0, rows, k)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    full = np.concatenate([dev.predict(csr, s, min(R, rows - s)).cpu().numpy() for s in range(0, rows, R)])
    for b in range(rows):
        row = full[b]
        lo, hi = row.min(), row.max()
        sc = (row - lo) / (hi - lo) if hi > lo else np.zeros_like(row)
        free = np.setdiff1d(np.arange(N), docs[b])
        n_ok = min(k, len(free))
        assert np.all(ids[b, n_ok:] == -1) and np.all(vals[b, n_ok:] == 0.0), (b, ids[b], len(free))
        got = ids[b, :n_ok]
        assert len(set(got.tolist())) == n_ok and not (set(got.tolist()) & set(docs[b].tolist()))
        np.testing.assert_allclose(vals[b, :n_ok], sc[got], atol=2e-6)
        assert np.all(np.diff(vals[b, :n_ok]) <= 1e-6)
        if n_ok:
            kth = np.sort(sc[free])[-n_ok]
            assert np.all(sc[got] >= kth - 2e-6)
