"""GPU parity at BASELINE.json's full sizes (the small fixtures pin every code path; these pin the SHAPES the benchmark
runs: 3 125 item tiles, 12.2 tile rounds per workgroup, 64-bit offsets into 80 MB tensors).

C3 (|items| = 100 000, hidden 200, code 50, batch 100): three full partial_fit steps with injected dropout masks and
prior draws on (a) the fused output-layer kernel, (b) the three-kernel output layer, (c) the NumPy oracle over the WHOLE
vocabulary (it finishes a step at this size in a few seconds, so nothing is sampled): losses and every parameter.

C5, one rank's share (the decoder's output layer over a slice of 275 000 of the 2.2 M items x the global batch of 512
rows, what a rank of the vocabulary-sharded scheme runs per step - aae_output_layer_step): against the NumPy
restatement of the slice (tests/test_parallel_gloo.py::VocabSliceReplica, itself checked against the oracle there).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _masks(rng, B, h, p=(0.2, 0.2)):
    return [(rng.random((B, h)) >= p[i % 2]).astype(np.uint8) for i in range(12)]


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def test_c3_full_size_step_fused_equals_three_kernel_path_equals_oracle():
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from tools.synth import init_params, throughput_corpus
    N, h, c, B, steps = 100000, 200, 50, 100, 3
    params = init_params(N, h, c, seed=3)
    X = throughput_corpus(steps * B, N, median_len=20, seed=77)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3)
    fused = HipAAE(N, h, c, max_batch=B, rng_mode="inject", **kw)
    plain = HipAAE(N, h, c, max_batch=B, rng_mode="inject", unfused_decoder=True, **kw)
    for m in (fused, plain):
        m.load_params(params)
    ora = O.OracleAAE({k: v.copy() for k, v in params.items()}, **kw)
    csr = DeviceCSR(X, fused.device)
    rng = np.random.default_rng(5)
    for s in range(steps):
        masks, z_real = _masks(rng, B, h), rng.standard_normal((B, c)).astype(np.float32)
        for m in (fused, plain):
            m.step(csr, s * B, B, masks=masks, z_real=z_real)
        Xb = X[s * B:(s + 1) * B]
        want = ora.partial_fit(Xb.indptr.astype(np.int64), Xb.indices, Xb.data.astype(np.float32), z_real, masks)
        lf, lp = fused.losses(), plain.losses()
        np.testing.assert_allclose(lf, want, rtol=1e-5, atol=1e-6, err_msg=f"fused losses, step {s}")
        np.testing.assert_allclose(lp, want, rtol=1e-5, atol=1e-6, err_msg=f"three-kernel losses, step {s}")
    sf, sp_ = fused.state_dict(), plain.state_dict()
    worst = {}
    for k, w in ora.p.items():
        worst[k] = (_maxdiff(sf[k], sp_[k]), _maxdiff(sf[k], w), _maxdiff(sp_[k], w))
    print("max |fused - three-kernel|, |fused - oracle|, |three-kernel - oracle| per tensor:", worst)
    for k, (d_fp, d_fo, d_po) in worst.items():
        # the two device paths differ by fp32 summation order only (same kernels elsewhere)
        assert d_fp <= 2e-6, (k, d_fp)
        # against the oracle: the tolerance of the fixture tests (1e-5 absolute on parameters)
        assert d_fo <= 1e-5 and d_po <= 1e-5, (k, d_fo, d_po)
    # and the predictions of the trained weights, over the whole vocabulary (north star: 1e-4)
    Xp = X[:B]
    want = ora.predict(Xp.indptr.astype(np.int64), Xp.indices, Xp.data.astype(np.float32))
    got = fused.predict(csr, 0, B).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_c5_share_output_layer_of_one_rank():
    """Config C5 (|items| = 2.2 M, batch 512, 8 ranks): the item slice one rank owns, 275 000 rows of dec.lin3, against
    the global batch of 512 rows - two consecutive steps (the second one starts from the first one's Adam state)."""
    from aaerec._hip import HipAAE, DeviceCSR
    from test_parallel_gloo import VocabSliceReplica
    from tools.synth import throughput_corpus
    Ns, h, c, B, world = 275000, 200, 50, 512, 8
    rng = np.random.default_rng(11)
    k = 1.0 / np.sqrt(h)
    params = {"dec.lin3.weight": ((rng.random((Ns, h)) * 2 - 1) * k).astype(np.float32),
              "dec.lin3.bias": ((rng.random(Ns) * 2 - 1) * k).astype(np.float32)}
    X = throughput_corpus(2 * B, Ns, median_len=8, seed=9)           # the slice's share of a playlist's ~60 tracks
    sl = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", dropout=(0.2, 0.2))
    full = {"dec.lin3.weight": params["dec.lin3.weight"], "dec.lin3.bias": params["dec.lin3.bias"]}
    sl.load_params(full)
    sl.set_grad_scale(1.0 / world)                                     # slice items / all items
    ref = VocabSliceReplica(params, 0, Ns, 1e-3)
    ref.set_grad_scale(1.0 / world)
    csr = DeviceCSR(X, sl.device)
    for s in range(2):
        dh2 = np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5
        dh2[rng.random((B, h + 1)) < 0.4] = 0.0                       # post-ReLU + dropout sparsity
        dh2[:, h] = 1.0                                                # the bias input column
        sl.dh2_rows(B)[:, :h + 1].copy_(torch.from_numpy(dh2))
        sl.output_layer_step(csr, s * B, B)
        ref.dh2_rows(B)[:] = torch.from_numpy(dh2)
        ref.output_layer_step((X.indptr.astype(np.int64), X.indices, X.data.astype(np.float32)), s * B, B)
        np.testing.assert_allclose(sl.losses()[0], ref.loss, rtol=1e-5)
        got = sl.da2_rows(B)[:, :h].cpu().numpy()
        want = ref.da2_rows(B)[:, :h].numpy()
        # dL/d(dh2): sums over 275 000 items of terms ~1e-9 each: absolute differences of fp32 summation order
        scale = float(np.abs(want).max())
        assert _maxdiff(got, want) <= 2e-5 * scale + 1e-12, (_maxdiff(got, want), scale)
    sd = sl.state_dict()
    dw, db = _maxdiff(sd["dec.lin3.weight"], ref.p["w"]), _maxdiff(sd["dec.lin3.bias"], ref.p["b"])
    print("C5 share: max |dW|", dw, "max |db|", db)
    assert dw <= 1e-5 and db <= 1e-5


@pytest.mark.parametrize("B", [96, 100])
def test_split_output_layer_equals_the_single_launch_and_views_wait_for_the_deferred_launch(B):
    """The fused output layer runs as a critical launch (logits, BCE, dL/d(hidden)) on the caller's stream and a deferred
    one (dV3 + dec_optim) on the handle's side stream (include/aaerec_hip.h: aae_join / aae_set_split).  Same arithmetic
    in the same order at 96 rows (whole 16-row blocks): after the first step (the forward pass has no atomics) dec.lin3
    and both dec_optim moments equal the one-launch form BIT FOR BIT; later steps differ only through the encoder's
    scatter atomics (2e-6, the bound between the fused and the three-kernel path above; a view that did not wait for the
    deferred launch would be one Adam step = 1e-3 off).  At the benchmark's 100 rows the critical launch takes the last 4
    rows on 4x4 matrix-core blocks (another summation order of the same products; the one-launch form has no registers
    left for it, dec_fused.h): 2e-6 from the first step on.  A view taken right behind a step - while the deferred launch
    is still running - must already show its result."""
    from aaerec._hip import HipAAE, DeviceCSR, T_DEC_V3, T_ADAM_DEC
    from tools.synth import init_params, throughput_corpus
    N, h, c, steps = 100000, 200, 50, 6
    params = init_params(N, h, c, seed=4)
    X = throughput_corpus(steps * B, N, median_len=20, seed=78)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3, rng_mode="device", seed=99)
    split, single = HipAAE(N, h, c, max_batch=B, **kw), HipAAE(N, h, c, max_batch=B, **kw)
    single.set_split(0)
    for m in (split, single):
        m.load_params(params)
    csr = DeviceCSR(X, split.device)
    tids = (T_DEC_V3, T_ADAM_DEC + 4, T_ADAM_DEC + 5)

    def views():    # taken right behind the steps: tensor() joins the deferred launch before the copy reads the tensor
        return [(split.tensor(t, padded=True).clone(), single.tensor(t, padded=True).clone()) for t in tids]
    for s in range(steps):
        before = split.tensor(T_DEC_V3, padded=True).clone() if s == 0 else None
        for m in (split, single):
            m.step(csr, s * B, B)
        if s == 0:
            for a, b in views():
                assert torch.equal(a, b) if B % 16 == 0 else float((a - b).abs().max()) <= 2e-6
            assert float((split.tensor(T_DEC_V3, padded=True) - before).abs().max()) > 5e-4      # (the step did move it)
        else:
            for a, b in views():
                assert float((a - b).abs().max()) <= 2e-6
    np.testing.assert_allclose(split.losses(), single.losses(), rtol=1e-5)
    sa, sb = split.state_dict(), single.state_dict()
    for k in sa:
        np.testing.assert_allclose(sa[k], sb[k], atol=2e-6, err_msg=k)
    # short batches and a batch-size change in mid-run (the deferred launch of a 100-row step is pending when a 37-row
    # one starts) keep the order as well
    for s, rows in enumerate((37, 100, 5, 64)):
        for m in (split, single):
            m.step(csr, s * B, min(rows, B))
    np.testing.assert_allclose(split.losses(), single.losses(), rtol=1e-5)
    assert float((split.tensor(T_DEC_V3, padded=True) - single.tensor(T_DEC_V3, padded=True)).abs().max()) <= 2e-6
    # predict reads DEC_V3 through the ABI: joined inside
    np.testing.assert_allclose(split.predict(csr, 0, B).cpu().numpy(), single.predict(csr, 0, B).cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("Ns,B", [(12500, 105), (12500, 800), (25000, 512), (4587, 1000), (50000, 200), (275000, 512)])
def test_row_blocked_output_layer_equals_the_three_kernel_path(Ns, B, monkeypatch):
    """Batches beyond one fused launch (112 rows) on a model created with blocked_output=True - what the item slices of the
    vocabulary-sharded scheme run at every world size > 1 (slice x the global batch, aae_output_layer_step): one critical
    launch per row block of <= 104 rows, the deferred launches accumulating dV3 over the blocks before ONE optimiser pass.
    Against the three-kernel path of the same model (itself pinned by the reference's fixtures and the C5-share test
    above): dL/d(dh2), loss, parameters and both moments after two consecutive steps."""
    from aaerec._hip import HipAAE, DeviceCSR, T_DEC_V3, T_ADAM_DEC
    from tools.synth import throughput_corpus
    h, c = 200, 50
    monkeypatch.setenv("AAE_BLOCKED_ANY", "1")       # (the library keeps the three GEMMs beyond 32 M cells: 275 000 x 512, a C5 slice, is tested all the same)
    rng = np.random.default_rng(Ns + B)
    k = 1.0 / np.sqrt(h)
    full = {"dec.lin3.weight": ((rng.random((Ns, h)) * 2 - 1) * k).astype(np.float32),
            "dec.lin3.bias": ((rng.random(Ns) * 2 - 1) * k).astype(np.float32)}
    X = throughput_corpus(2 * B, Ns, median_len=6, seed=Ns)
    blocked = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", blocked_output=True)
    plain = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", unfused_decoder=True)
    for m in (blocked, plain):
        m.load_params(full)
        m.set_grad_scale(0.25)
    csr = DeviceCSR(X, blocked.device)
    for s in range(2):
        dh2 = np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5
        dh2[rng.random((B, h + 1)) < 0.4] = 0.0
        dh2[:, h] = 1.0
        for m in (blocked, plain):
            m.dh2_rows(B)[:, :h + 1].copy_(torch.from_numpy(dh2))
            m.output_layer_step(csr, s * B, B)
        np.testing.assert_allclose(blocked.losses()[0], plain.losses()[0], rtol=1e-5)
        got, want = blocked.da2_rows(B)[:, :h].cpu().numpy(), plain.da2_rows(B)[:, :h].cpu().numpy()
        scale = float(np.abs(want).max())
        assert _maxdiff(got, want) <= 2e-5 * scale + 1e-12, (s, _maxdiff(got, want), scale)
    for tid in (T_DEC_V3, T_ADAM_DEC + 4, T_ADAM_DEC + 5):
        a, b = blocked.tensor(tid).cpu().numpy(), plain.tensor(tid).cpu().numpy()
        tol = 2e-6 if tid == T_DEC_V3 else 1e-9
        assert _maxdiff(a, b) <= tol + 1e-4 * float(np.abs(b).max()), (tid, _maxdiff(a, b))


@pytest.mark.parametrize("N,B", [(6000, 150), (40000, 224)])
def test_row_blocked_full_steps_equal_the_three_kernel_steps(N, B):
    """Whole training steps (aae_step: ae + disc + gen) on ONE handle with 113..256-row batches - what
    AdversarialAutoEncoder creates for such batch sizes (blocked_output: the output layer as one critical launch for the
    row blocks + the deferred optimiser launch(es); 6 000 items: dec_opt_blocks_kernel, 40 000: one launch per block) -
    against the same steps on the three-kernel output layer: losses and every parameter after 3 steps, short batches
    in between included."""
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import init_params, throughput_corpus
    h, c = 200, 50
    params = init_params(N, h, c, seed=3)
    X = throughput_corpus(4 * B, N, median_len=10, seed=N)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3, rng_mode="device", seed=5)
    blocked = HipAAE(N, h, c, max_batch=B, blocked_output=True, **kw)
    plain = HipAAE(N, h, c, max_batch=B, unfused_decoder=True, **kw)
    for m in (blocked, plain):
        m.load_params(params)
    csr = DeviceCSR(X, blocked.device)
    for s, rows in enumerate((B, B, 97, B)):          # (97 rows: one fused launch on the same handle)
        for m in (blocked, plain):
            m.step(csr, s * B, rows)
        np.testing.assert_allclose(blocked.losses(), plain.losses(), rtol=2e-5, err_msg=f"step {s}")
    sa, sb = blocked.state_dict(), plain.state_dict()
    for k in sa:
        np.testing.assert_allclose(sa[k], sb[k], atol=4e-6, err_msg=k)
