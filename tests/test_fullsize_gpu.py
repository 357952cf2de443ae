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
            for a, b in views():      # (r2: bit-equal at 96 rows; since r3 the critical launch multiplies on the bf16 matrix
                assert float((a - b).abs().max()) <= 2e-6      #  cores - csrc/dec_crit_x3.h: another rounding of the same products)
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


@pytest.mark.parametrize("N,h,B,dtype", [(100000, 200, 100, "f32"), (47000, 100, 100, "f32"), (47000, 100, 100, "bf16"), (20000, 200, 64, "f32")])
def test_late_join_of_the_deferred_launch_changes_no_result(N, h, B, dtype, monkeypatch):
    """Late join (csrc/abi_model.h: join_step_open / join_output_layer): the deferred optimiser launch of step t keeps running
    while step t+1's forward pass goes on - it reads the copies of dh2 and of the layer's step scalars that step t's critical
    launch set aside - and step t+1 waits for it in front of its own output layer.  Against a handle created with
    AAE_NO_LATE_JOIN=1 (every step opens behind the deferred launch, as up to r3) on the same batches and device randomness:
    the same arithmetic in the same order, so dec.lin3 and both Adam moments agree to the scatter-atomics bound of the test
    above (2e-6; a deferred launch that read the NEXT step's dh2 or step scalars would be a whole Adam step, 1e-3, off) at
    every step - views taken right behind a step, with the launch still running -, through batch-size changes, a predict
    call between steps (it reads dec.lin3: joined inside) and the phase-wise entry points."""
    from aaerec._hip import HipAAE, DeviceCSR, T_DEC_V3, T_ADAM_DEC
    from tools.synth import init_params, throughput_corpus
    c, steps = 50, 10
    params = init_params(N, h, c, seed=6)
    X = throughput_corpus(steps * B, N, median_len=20, seed=31)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3, rng_mode="device", seed=5, **({"dtype": "bf16"} if dtype == "bf16" else {}))
    late = HipAAE(N, h, c, max_batch=B, **kw)
    monkeypatch.setenv("AAE_NO_LATE_JOIN", "1")
    early = HipAAE(N, h, c, max_batch=B, **kw)
    monkeypatch.delenv("AAE_NO_LATE_JOIN")
    for m in (late, early):
        m.load_params(params)
    csr = DeviceCSR(X, late.device)
    tids = (T_DEC_V3, T_ADAM_DEC + 4, T_ADAM_DEC + 5)
    tol = 2e-6 if dtype == "f32" else 2e-5

    def same(what):
        for t in tids:
            a, b = late.tensor(t, padded=True).clone(), early.tensor(t, padded=True).clone()
            assert float((a - b).abs().max()) <= tol, (what, t)
    for s in range(steps):
        for m in (late, early):
            m.step(csr, s * B, B)
        if s in (0, 1, 4, steps - 1):
            same(f"step {s}")
        if s == 5:      # a call that reads dec.lin3 between two steps
            np.testing.assert_allclose(late.predict(csr, 0, B).cpu().numpy(), early.predict(csr, 0, B).cpu().numpy(), atol=2e-6 if dtype == "f32" else 2e-4)
    for s, rows in enumerate((37, B, 5, 64, B)):
        for m in (late, early):
            m.step(csr, s * B, min(rows, B))
    same("ragged batches")
    np.testing.assert_allclose(late.losses(), early.losses(), rtol=1e-5 if dtype == "f32" else 1e-3)
    sa, sb = late.state_dict(), early.state_dict()
    for k in sa:
        np.testing.assert_allclose(sa[k], sb[k], atol=tol * (1 if dtype == "f32" else 50), err_msg=k)


@pytest.mark.parametrize("Ns,B,h", [(12500, 105, 200), (12500, 800, 200), (25000, 512, 200), (4587, 1000, 200), (50000, 200, 200), (275000, 512, 200),
                                    (6000, 300, 50), (9000, 230, 100), (40000, 512, 100), (3000, 130, 61)])
def test_row_blocked_output_layer_equals_the_three_kernel_path(Ns, B, h, monkeypatch):
    """Batches beyond one fused launch (112 rows) on a model created with blocked_output=True - what the item slices of the
    vocabulary-sharded scheme run at every world size > 1 (slice x the global batch, aae_output_layer_step): one critical
    launch per row block of <= 104 rows, the deferred launches accumulating dV3 over the blocks before ONE optimiser pass.
    Against the three-kernel path of the same model (itself pinned by the reference's fixtures and the C5-share test
    above): dL/d(dh2), loss, parameters and both moments after two consecutive steps."""
    from aaerec._hip import HipAAE, DeviceCSR, T_DEC_V3, T_ADAM_DEC
    from tools.synth import throughput_corpus
    c = 50        # (h = 50 / 61 / 100: the 4- and 7-column-block instantiations of both launches, 12 / 9 spare waves in the deferred one)
    monkeypatch.setenv("AAE_BLOCKED_ANY", "1")       # (r2: the library kept the three GEMMs beyond 32 M cells; since r3 the row-blocked form covers 275 000 x 512, a C5 slice, by itself)
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


@pytest.mark.parametrize("B", [100, 208])
def test_fused_output_layer_beyond_two_gib(B):
    """vocabularies whose dec.lin3 exceeds 2^31 bytes - PubMed's 2 896 764 and ACM's 2 631 128 items of the
    reference's own dataset table (nmi.txt:68,85) at hidden 200: 2.37 GB per tensor - stay on the fused output layer (r1-r4
    fell back to the three streaming GEMMs with [B, N] in HBM above 2.63 M items).  2 900 000 items, hidden 200, one fused
    launch (100 rows) and the row-blocked form (208 rows), two consecutive steps: dL/d(dh2), loss, EVERY row of dec.lin3 and of
    both dec_optim moments against the three-kernel path; the profile counters say which kernels ran."""
    from aaerec._hip import HipAAE, DeviceCSR, T_DEC_V3, T_ADAM_DEC, K_DEC_CRIT, K_DEC_FUSED, K_DEC_BCE_FWD
    from tools.synth import throughput_corpus
    Ns, h, c = 2900000, 200, 50
    assert (Ns + 64) * 204 * 4 > 2 ** 31
    rng = np.random.default_rng(Ns + B)
    k = 1.0 / np.sqrt(h)
    full = {"dec.lin3.weight": ((rng.random((Ns, h), dtype=np.float32) * 2 - 1) * np.float32(k)),
            "dec.lin3.bias": ((rng.random(Ns, dtype=np.float32) * 2 - 1) * np.float32(k))}
    X = throughput_corpus(2 * B, Ns, median_len=30, seed=Ns)
    fused = HipAAE(Ns, h, c, max_batch=B, max_nnz=B * 256, rng_mode="inject", blocked_output=B > 112)
    plain = HipAAE(Ns, h, c, max_batch=B, max_nnz=B * 256, rng_mode="inject", unfused_decoder=True)
    for m in (fused, plain):
        m.load_params(full)
        m.set_grad_scale(0.25)
    del full
    fused.profile_enable(True, kernels=(K_DEC_CRIT, K_DEC_FUSED))
    plain.profile_enable(True, kernels=(K_DEC_BCE_FWD,))
    csr = DeviceCSR(X, fused.device)
    for s in range(2):
        dh2 = np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5
        dh2[rng.random((B, h + 1)) < 0.4] = 0.0
        dh2[:, h] = 1.0
        for m in (fused, plain):
            m.dh2_rows(B)[:, :h + 1].copy_(torch.from_numpy(dh2))
            m.output_layer_step(csr, s * B, B)
        np.testing.assert_allclose(fused.losses()[0], plain.losses()[0], rtol=1e-5)
        got, want = fused.da2_rows(B)[:, :h].cpu().numpy(), plain.da2_rows(B)[:, :h].cpu().numpy()
        scale = float(np.abs(want).max())
        assert _maxdiff(got, want) <= 2e-5 * scale + 1e-12, (s, _maxdiff(got, want), scale)
    torch.cuda.synchronize()
    # (a layer this large in ONE row block runs the single fused launch - its deferred half would outlast the step -, the
    #  row-blocked form the critical launch + the deferred launch over all blocks: abi_output_layer.h)
    ran = fused.profile_read(K_DEC_FUSED)[1] if B <= 112 else fused.profile_read(K_DEC_CRIT)[1]
    assert ran == 2, "the fused output layer did not run"
    assert plain.profile_read(K_DEC_BCE_FWD)[1] >= 2, "the three-kernel path did not run"
    for tid in (T_DEC_V3, T_ADAM_DEC + 4, T_ADAM_DEC + 5):
        a, b = fused.tensor(tid), plain.tensor(tid)
        d = float((a - b).abs().max().item())
        tol = 2e-6 if tid == T_DEC_V3 else 1e-9
        assert d <= tol + 1e-4 * float(b.abs().max().item()), (tid, d)
        # ... and the rows beyond byte 2^31 did move (an update dropped by a range check would leave them at their start values)
        assert float(a[-1000:].abs().sum().item()) > 0


@pytest.mark.parametrize("N,B", [(6000, 150), (40000, 224), (100000, 512)])
def test_row_blocked_full_steps_equal_the_three_kernel_steps(N, B):
    """Whole training steps (aae_step: ae + disc + gen) on ONE handle with 113..256-row batches - what
    AdversarialAutoEncoder creates for such batch sizes (blocked_output: the output layer as one critical launch for the
    row blocks + ONE deferred optimiser launch for any vocabulary, dec_opt_blocks_x3_kernel; 100 000 x 512 is bench.py's
    extra.b512: four passes of <= 6 tiles per workgroup) - against the same steps on the three-kernel output layer: losses
    and every parameter after 3 steps, short batches in between included."""
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


def test_c2_full_size_bf16_step_fused_equals_three_kernel_path_equals_rounded_oracle():
    """BASELINE configs[1] at size (|items| = 47 000, hidden 100, code 50, batch 100, bf16 matrix-core inputs): three full
    partial_fit steps with injected randomness on (a) the fused bf16 output layer (2 938 units of 16 items, 11.5 unit
    rounds per workgroup of the cross-unit pipeline - what profiles/*_bench_bf16.json times), (b) the three-kernel bf16
    output layer, (c) OracleAAE(bf16=True) over the WHOLE vocabulary.  Mirror of the fp32 C3 test above, at the
    tolerances of tests/test_bf16_gpu.py (an fp32 summation-order difference that straddles a bf16 rounding boundary of
    an operand moves it by 2^-8 relative: rare, bounded, and Adam's first steps turn a sign flip of a near-zero gradient
    into ~2 lr)."""
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from tools.synth import init_params, throughput_corpus
    N, h, c, B, steps = 47000, 100, 50, 100, 3
    params = init_params(N, h, c, seed=13)
    X = throughput_corpus(steps * B, N, median_len=20, seed=79)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3)
    fused = HipAAE(N, h, c, max_batch=B, rng_mode="inject", dtype="bf16", **kw)
    plain = HipAAE(N, h, c, max_batch=B, rng_mode="inject", dtype="bf16", unfused_decoder=True, **kw)
    for m in (fused, plain):
        m.load_params(params)
    ora = O.OracleAAE({k: v.copy() for k, v in params.items()}, bf16=True, **kw)
    csr = DeviceCSR(X, fused.device)
    rng = np.random.default_rng(6)
    for s in range(steps):
        masks, z_real = _masks(rng, B, h), rng.standard_normal((B, c)).astype(np.float32)
        for m in (fused, plain):
            m.step(csr, s * B, B, masks=masks, z_real=z_real)
        Xb = X[s * B:(s + 1) * B]
        want = ora.partial_fit(Xb.indptr.astype(np.int64), Xb.indices, Xb.data.astype(np.float32), z_real, masks)
        np.testing.assert_allclose(fused.losses(), want, rtol=2e-4, atol=2e-6, err_msg=f"fused bf16 losses, step {s}")
        np.testing.assert_allclose(plain.losses(), want, rtol=2e-4, atol=2e-6, err_msg=f"three-kernel bf16 losses, step {s}")
    sf, sp_ = fused.state_dict(), plain.state_dict()
    lr, stats = 1e-3, {}
    for k, w in ora.p.items():
        n_opt = 2 if k.startswith("enc.") else 1           # the encoder takes two optimiser steps per partial_fit
        for tag, sd in (("fused", sf), ("three-kernel", sp_)):
            d = np.abs(sd[k].astype(np.float64) - w)
            stats[(tag, k)] = (float((d > 1e-4).mean()), float(d.max()), 3.0 * lr * steps * n_opt)
        d = np.abs(sf[k].astype(np.float64) - sp_[k])
        stats[("fused vs three-kernel", k)] = (float((d > 1e-4).mean()), float(d.max()), 3.0 * lr * steps * n_opt)
    print({k: (round(f, 6), round(mx, 6)) for k, (f, mx, _) in stats.items()})
    for k, (frac, mx, bound) in stats.items():
        assert frac <= 1e-2 and mx <= bound, (k, frac, mx, bound)
    # dec.lin3 holds 99 % of the parameters and its gradient never passes through a rounded activation twice: tight bound
    for tag, sd in (("fused", sf), ("three-kernel", sp_)):
        d = np.abs(sd["dec.lin3.weight"].astype(np.float64) - ora.p["dec.lin3.weight"])
        assert float((d > 1e-5).mean()) <= 1e-3, (tag, float((d > 1e-5).mean()))
    # reconstructions of the trained weights over the whole vocabulary, against the rounded oracle's
    Xp = X[:B]
    want = ora.predict(Xp.indptr.astype(np.int64), Xp.indices, Xp.data.astype(np.float32))
    got = fused.predict(csr, 0, B).cpu().numpy()
    assert _maxdiff(got, want) <= 2e-3, _maxdiff(got, want)


def test_c5_both_sharded_slice_of_one_rank():
    """Config C5 under the default data-parallel scheme (dp_mode='vocab'): BOTH vocabulary-wide matrices
    live with the item slices, so a rank's slice handle runs, per step and for the GLOBAL batch of 512 documents,
      aae_first_layer_forward   its 275 000 items' share of x * enc.lin1^T (complete-document L1 norms from aae_set_doc_l1),
      aae_output_layer_step     logits / BCE / dV3 + dec_optim / dL/d(dh2) partial over its rows of dec.lin3,
      aae_first_layer_update    x^T * dL/d(a1) on its rows of enc.lin1 + enc_optim, dL/d(a1) read as the ranks' gathered
                                packets (rows_per_block / block_stride), then - Enc_eval of the disc phase -
      aae_first_layer_forward(NULL) with the updated rows, and gen_optim's aae_first_layer_update.
    Items are owned INTERLEAVED (rank r holds items r, r + 8, ...; with the reference's frequency-sorted vocabularies
    every slice then sees 1/8 of each document: ~60 / 8 entries): the corpus below is a 2.2 M-item Zipf corpus of
    median length 60 restricted to the columns of rank 3.  Two consecutive steps against the NumPy stand-in of
    tests/test_parallel_gloo.py (itself checked against the oracle in the gloo tests)."""
    from aaerec._hip import HipAAE, DeviceCSR, O_ENC, O_GEN
    from test_parallel_gloo import BothSliceReplica
    from tools.synth import throughput_corpus
    N, world, rank, h, c, B, Bl = 2200000, 8, 3, 200, 50, 512, 64
    Ns = N // world
    rng = np.random.default_rng(17)
    k1, k3 = 1.0 / np.sqrt(N), 1.0 / np.sqrt(h)
    params = {"dec.lin3.weight": ((rng.random((Ns, h), dtype=np.float32) * 2 - 1) * k3).astype(np.float32),
              "dec.lin3.bias": ((rng.random(Ns, dtype=np.float32) * 2 - 1) * k3).astype(np.float32),
              "enc.lin1.weight": ((rng.random((h, Ns), dtype=np.float32) * 2 - 1) * k1 * 30).astype(np.float32),
              "enc.lin1.bias": np.zeros(h, dtype=np.float32)}             # (the bias lives with the replicas: handed to one share below)
    Xg = throughput_corpus(2 * B, N, median_len=60, seed=19)          # the global batches over the WHOLE vocabulary
    l1 = np.asarray(abs(Xg).sum(1)).reshape(-1).astype(np.float32)     # F.normalize divides by the complete row
    X = Xg[:, rank::world].tocsr()                                     # this rank's columns, ids rebased
    X.sort_indices()
    print("entries per document on the slice: mean", X.nnz / X.shape[0], "max", int(np.diff(X.indptr).max()))
    sl = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", dropout=(0.2, 0.2), blocked_output=True, gen_lr=1e-3, reg_lr=2e-3)
    sl.load_params(params)
    sl.set_grad_scale(1.0 / world)
    sl.set_doc_l1(torch.from_numpy(l1).to(sl.device))
    ref = BothSliceReplica(params, 0, Ns, 1e-3, 2e-3)
    ref.set_grad_scale(1.0 / world)
    ref.set_doc_l1(l1)
    csr = DeviceCSR(X, sl.device)
    host = (X.indptr.astype(np.int64), X.indices, X.data.astype(np.float32))
    bias = torch.from_numpy(((rng.random(h) * 2 - 1) * 0.05).astype(np.float32))
    bias_dev = torch.zeros(sl.first_layer_bias().numel(), device=sl.device)
    bias_dev[:h] = bias.to(sl.device)
    ldg = sl.ga1_rows(1).stride(0)

    def packets(g, ld):     # [B, h] -> 8 blocks of Bl rows of leading dimension ld, each followed by a tail of 256 other
        stride = Bl * ld + 256                                         # floats, as an all-gather of the ranks' packets leaves them
        buf = torch.zeros(world * stride)
        for r in range(world):
            blk = torch.zeros(Bl, ld)
            blk[:, :h] = torch.from_numpy(g[r * Bl:(r + 1) * Bl])
            buf[r * stride:r * stride + Bl * ld] = blk.reshape(-1)
        return buf, stride
    for s in range(2):
        if s == 0:
            sl.prefetch(csr, B, B)                                     # the next global batch, named ahead as fit() does
        sl.first_layer_forward(csr, s * B, B, bias=bias_dev)
        ref.first_layer_forward(host, s * B, B, bias=bias)
        a_got, a_want = sl.a1_rows(B)[:, :h].cpu().numpy(), ref.a1_rows(B).numpy()
        assert _maxdiff(a_got, a_want) <= 1e-5 * max(1.0, float(np.abs(a_want).max())), ("a1", s, _maxdiff(a_got, a_want))
        dh2 = np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5
        dh2[rng.random((B, h + 1)) < 0.4] = 0.0
        dh2[:, h] = 1.0
        sl.dh2_rows(B)[:, :h + 1].copy_(torch.from_numpy(dh2))
        ref.dh2_rows(B)[:] = torch.from_numpy(dh2)
        sl.output_layer_step()
        ref.output_layer_step()
        np.testing.assert_allclose(sl.losses()[0], ref.loss, rtol=1e-5)
        got, want = sl.da2_rows(B)[:, :h].cpu().numpy(), ref.da2_rows(B)[:, :h].numpy()   # (reading the view joins the deferred launch, not the prefetch behind it)
        scale = float(np.abs(want).max())
        assert _maxdiff(got, want) <= 2e-5 * scale + 1e-12, (s, _maxdiff(got, want), scale)
        for which in (O_ENC, O_GEN):
            g = (rng.standard_normal((B, h)) * 1e-3).astype(np.float32)
            pk, stride = packets(g, ldg)
            pk_dev = pk.to(sl.device)
            sl.first_layer_update(which, pk_dev, rows_per_block=Bl, block_stride=stride)
            pk_ref, stride_ref = packets(g, h)                          # (the stand-in reads rows of leading dimension h)
            ref.first_layer_update(which, pk_ref, rows_per_block=Bl, block_stride=stride_ref)
            if which == O_ENC:                                          # Enc_eval of the disc phase: the updated rows, no bias here
                sl.first_layer_forward()
                ref.first_layer_forward()
                a_got, a_want = sl.a1_rows(B)[:, :h].cpu().numpy(), ref.a1_rows(B).numpy()
                assert _maxdiff(a_got, a_want) <= 1e-5 * max(1.0, float(np.abs(a_want).max())), ("a1 eval", s, _maxdiff(a_got, a_want))
            torch.cuda.synchronize()                                   # (pk_dev stays alive until the stream has passed the call)
    sd = sl.state_dict()
    dw, db = _maxdiff(sd["dec.lin3.weight"], ref.p["w"]), _maxdiff(sd["dec.lin3.bias"], ref.p["b"])
    d1 = _maxdiff(sd["enc.lin1.weight"], ref.w1["w1"])
    print("C5 both-sharded slice: max |dV3|", dw, "|db3|", db, "|dW1|", d1)
    assert dw <= 1e-5 and db <= 1e-5 and d1 <= 1e-5


def test_slice_prefetch_is_still_waited_for_after_a_view_was_read_between_the_phases():
    """An item slice of the both-sharded scheme whose NEXT global batch was named ahead (aae_prefetch_batch): the list of
    that batch's distinct items and their deferred-Adam catch-up are enqueued on the handle's side stream BEHIND the
    deferred optimiser launch of aae_output_layer_step.  Reading an activation view between the phases (da2_rows() ->
    aae_join_output_layer) settles the optimiser launch only - the prefetch behind it must stay pending, so that the next
    step's aae_first_layer_forward still waits for it before it swaps the list sets in and gathers enc.lin1 rows (ADVICE
    r2: the join cleared both marks; the race is usually hidden by the length of the step).  40 steps on a 12 500-item
    slice x 800 rows with items coming and going, with the view read, against the same steps without any prefetch: every
    tensor of the slice bit for bit (the first layer's update has no scheduling-dependent summation order since r3)."""
    from aaerec._hip import HipAAE, DeviceCSR, O_ENC, O_GEN, T_ENC_W1T, T_ADAM_ENC, T_ADAM_GEN, T_DEC_V3
    from tools.synth import throughput_corpus
    Ns, h, c, B, steps = 12500, 200, 50, 800, 40
    rng = np.random.default_rng(23)
    k = 1.0 / np.sqrt(h)
    params = {"dec.lin3.weight": ((rng.random((Ns, h)) * 2 - 1) * k).astype(np.float32),
              "dec.lin3.bias": ((rng.random(Ns) * 2 - 1) * k).astype(np.float32),
              "enc.lin1.weight": ((rng.random((h, Ns)) * 2 - 1) * 0.05).astype(np.float32),
              "enc.lin1.bias": np.zeros(h, dtype=np.float32)}
    X = throughput_corpus(steps * B, Ns, median_len=3, seed=29)
    dh2s = [np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5 for _ in range(4)]
    for d in dh2s:
        d[:, h] = 1.0
    gas = [(rng.standard_normal((B, h)) * 1e-3).astype(np.float32) for _ in range(4)]

    def run(prefetch):
        sl = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", blocked_output=True)
        sl.load_params(params)
        sl.set_grad_scale(0.125)
        csr = DeviceCSR(X, sl.device)
        dh = [torch.from_numpy(d).to(sl.device) for d in dh2s]
        ga = []
        for g in gas:
            t = torch.zeros(B, sl.ga1_rows(1).stride(0), device=sl.device)
            t[:, :h] = torch.from_numpy(g).to(sl.device)
            ga.append(t)
        seen = []
        for s in range(steps):
            if prefetch and s + 1 < steps:
                sl.prefetch(csr, (s + 1) * B, B)
            sl.first_layer_forward(csr, s * B, B)
            sl.dh2_rows(B)[:, :h + 1].copy_(dh[s % 4])
            sl.output_layer_step()
            if prefetch:
                seen.append(float(sl.da2_rows(B)[0, 0]))               # a view lookup between the phases (joins the optimiser launch)
            sl.first_layer_update(O_ENC, ga[s % 4], rows_per_block=B, block_stride=ga[s % 4].numel())
            sl.first_layer_forward()
            sl.first_layer_update(O_GEN, ga[(s + 1) % 4], rows_per_block=B, block_stride=ga[s % 4].numel())
        sl.sync()
        torch.cuda.synchronize()
        return {t: sl.tensor(t, padded=True).clone() for t in (T_ENC_W1T, T_ADAM_ENC, T_ADAM_ENC + 1, T_ADAM_GEN, T_ADAM_GEN + 1, T_DEC_V3)}
    a, b = run(True), run(False)
    for t in a:
        assert torch.equal(a[t], b[t]), (t, float((a[t] - b[t]).abs().max()))


class _ConstVectors:
    """Stand-in for the fitted TF-IDF x word2vec vectoriser behind PretrainedWordEmbeddingCondition: the document vectors
    come precomputed (condition.py:345-369 of the reference computes them once per dataset, outside the step)."""

    def __init__(self, dim):
        self.embedding = np.zeros((1, dim), dtype=np.float32)

    def fit(self, x):
        return self

    def transform(self, x):
        return x


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_c4_bench_shape_fit_path_matches_oracle(dtype, monkeypatch):
    """BASELINE configs[3] at its bench shape THROUGH fit(): |items| = 4 587 (EconBiz, nmi.txt:38), hidden 200, code 50, a
    300-d constant title condition (PretrainedWordEmbeddingCondition: condition.py:312-316, 345-369), batch 1000
    (eval/econis.py:45).  AdversarialAutoEncoder(batch_size=1000, conditions=...).fit() creates the handle itself, so this
    is the combination the benchmark's C4 line runs - the row-blocked output layer (blocked_output: one critical launch for
    10 row blocks + one deferred optimiser launch), dec.lin1 as two k-parts of the chain program over 1 000 rows (351 input
    columns), the workgroup-per-item form of the first layer's update (N < 40 x rows), the wide tile-bucket builder, the
    next batch named ahead - with the randomness fit() draws on the host (rng_mode='reference': 12 dropout masks + z_real
    per step) recorded and replayed through OracleAAE(conditions=[ConcatConst(300)]) on the same permutation batches:
    losses after each of 3 steps, every parameter, predictions; fp32 at the fixture tolerances (1e-5 / 1e-5 / 1e-4), bf16
    against OracleAAE(bf16=True) at tests/test_bf16_gpu.py's."""
    from aaerec import _hip
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec.condition import ConditionList, PretrainedWordEmbeddingCondition
    from oracle import aae_oracle as O
    from tools.synth import throughput_corpus
    N, h, c, inc, B, steps = 4587, 200, 50, 300, 1000, 3
    X = throughput_corpus(steps * B, N, median_len=20, seed=45)
    cond_all = (np.random.default_rng(8).standard_normal((steps * B, inc)) * 0.1).astype(np.float32)
    conditions = ConditionList([("title", PretrainedWordEmbeddingCondition(_ConstVectors(inc), use_cuda=True))])
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3)
    model = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, verbose=False, rng_mode="reference",
                                   conditions=conditions, **({"dtype": "bf16"} if dtype == "bf16" else {}), **kw)
    seen = {"params": None, "draws": [], "create": None}
    load0, host0, init0 = _hip.HipAAE.load_params, AdversarialAutoEncoder._host_randomness, _hip.HipAAE.__init__

    def load_params(self, params):
        if seen["params"] is None:
            seen["params"] = {k: np.array(v, copy=True) for k, v in params.items()}
        return load0(self, params)

    def host_randomness(self, rows):
        masks, z_real = host0(self, rows)
        seen["draws"].append(([m.numpy().copy() for m in masks], z_real.numpy().copy()))
        return masks, z_real

    def init(self, *a, **k):
        seen["create"] = dict(k)
        return init0(self, *a, **k)
    monkeypatch.setattr(_hip.HipAAE, "load_params", load_params)
    monkeypatch.setattr(_hip.HipAAE, "__init__", init)
    monkeypatch.setattr(AdversarialAutoEncoder, "_host_randomness", host_randomness)
    np.random.seed(4)
    state = np.random.get_state()
    torch.manual_seed(11)
    losses = [model.hip.losses() for _ in model.fit_steps(X, condition_data=[cond_all])]     # (exhausted: fit())
    np.random.set_state(state)
    perm = np.arange(steps * B)
    np.random.shuffle(perm)                              # fit()'s epoch permutation (aae.py:813-817 of the reference)
    assert len(losses) == steps and len(seen["draws"]) == steps
    assert seen["create"]["blocked_output"] and seen["create"]["cond_inc"] == inc and seen["create"]["max_batch"] == B
    ora = O.OracleAAE(seen["params"], conditions=[O.ConcatConst(inc)], bf16=dtype == "bf16", **kw)
    for s in range(steps):
        idx = perm[s * B:(s + 1) * B]
        Xb = X[idx]
        masks, z_real = seen["draws"][s]
        want = ora.partial_fit(Xb.indptr.astype(np.int64), Xb.indices, Xb.data.astype(np.float32), z_real, masks, [cond_all[idx]])
        if dtype == "f32":
            np.testing.assert_allclose(losses[s], want, rtol=1e-5, atol=1e-6, err_msg=f"losses, step {s}")
        else:
            np.testing.assert_allclose(losses[s], want, rtol=2e-4, atol=2e-6, err_msg=f"bf16 losses, step {s}")
    sd = model.hip.state_dict()
    worst = {k: _maxdiff(sd[k], w) for k, w in ora.p.items()}
    print("C4 through fit(),", dtype, ": max |device - oracle| per tensor:", worst)
    Xp, cp = X[:B], cond_all[:B]
    want = ora.predict(Xp.indptr.astype(np.int64), Xp.indices, Xp.data.astype(np.float32), [cp])
    got = model.predict(Xp, condition_data=[cp])
    if dtype == "f32":
        for k, d in worst.items():
            assert d <= 1e-5, (k, d)
        np.testing.assert_allclose(got, want, atol=1e-5)
    else:
        lr = 1e-3
        for k, w in ora.p.items():
            d = np.abs(sd[k].astype(np.float64) - w)
            n_opt = 2 if k.startswith("enc.") else 1
            # (Adam's first steps move a weight by ~lr whatever its gradient's size: an element whose near-zero gradient changes
            #  sign under another summation order of bf16-rounded operands lands up to 2 lr per step away.  dec.lin1's 300
            #  condition columns sum 1 000 signed terms of |cond| ~ 0.1 per element - more near-zero gradients than any tensor
            #  of the C2 test: 1.2 % of its elements beyond 1e-4 after 3 steps, none beyond 0.8 lr.)
            assert float((d > 1e-4).mean()) <= 2e-2 and d.max() <= 3.0 * lr * steps * n_opt, (k, float((d > 1e-4).mean()), d.max())
        assert _maxdiff(got, want) <= 2e-3, _maxdiff(got, want)


@pytest.mark.parametrize("B", [2000, 10000])
def test_reference_drivers_large_batches_through_fit_match_oracle(B, monkeypatch):
    """The batch sizes of the reference's own drivers beyond one fused launch AND beyond the row-blocked form (1 664 rows):
    eval/aminer.py:62 trains with batch_size 10 000, and 2 000 is the first size past the row-blocked limit.  There the output
    layer is the three streaming GEMMs on the emulated product (gemm_x3_kernel: logits + BCE epilogue, dA2 as split-K slabs,
    dV3 + dec_optim), the hidden stacks run on the 16-row chain kernel, the first layer's update takes the wave-per-item form -
    r1-r5 pinned that path up to 260 rows only.  AdversarialAutoEncoder(batch_size=B).fit() on a 20 000-item
    vocabulary, two steps with the host's draws (rng_mode='reference') recorded and replayed through the oracle on the same
    permutation batches: losses at 1e-5, every parameter at 1e-5 per 1e-3 of learning rate, predictions at 1e-5."""
    from aaerec import _hip
    from aaerec.aae import AdversarialAutoEncoder
    from oracle.dense_torch_port import DenseTorchAAE
    from tools.synth import throughput_corpus
    N, h, c, steps = 20000, 200, 50, 2
    X = throughput_corpus(steps * B, N, median_len=20, seed=61)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3)
    model = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, verbose=False, rng_mode="reference", **kw)
    seen = {"params": None, "draws": [], "create": None}
    load0, host0, init0 = _hip.HipAAE.load_params, AdversarialAutoEncoder._host_randomness, _hip.HipAAE.__init__

    def load_params(self, params):
        if seen["params"] is None:
            seen["params"] = {k: np.array(v, copy=True) for k, v in params.items()}
        return load0(self, params)

    def host_randomness(self, rows):
        masks, z_real = host0(self, rows)
        seen["draws"].append(([m.numpy().copy() for m in masks], z_real.numpy().copy()))
        return masks, z_real

    def init(self, *a, **k):
        seen["create"] = dict(k)
        return init0(self, *a, **k)
    monkeypatch.setattr(_hip.HipAAE, "load_params", load_params)
    monkeypatch.setattr(_hip.HipAAE, "__init__", init)
    monkeypatch.setattr(AdversarialAutoEncoder, "_host_randomness", host_randomness)
    np.random.seed(6)
    state = np.random.get_state()
    torch.manual_seed(12)
    losses = [model.hip.losses() for _ in model.fit_steps(X)]
    np.random.set_state(state)
    perm = np.arange(steps * B)
    np.random.shuffle(perm)                              # fit()'s epoch permutation (aae.py:813-817 of the reference)
    assert len(losses) == steps and len(seen["draws"]) == steps and seen["create"]["max_batch"] == B
    # (the checker is the PyTorch-CPU port in the reference's dense formulation - oracle/dense_torch_port.py, held to the
    #  reference's fixtures by tests/test_oracle_golden.py like the NumPy oracle, and multi-threaded: 2e8 cells per step)
    ora = DenseTorchAAE(seen["params"], **kw)
    for s in range(steps):
        Xb = X[perm[s * B:(s + 1) * B]]
        masks, z_real = seen["draws"][s]
        want = ora.partial_fit(Xb.toarray().astype(np.float32), z_real, masks)
        np.testing.assert_allclose(losses[s], want, rtol=1e-5, atol=1e-6, err_msg=f"B={B}: losses, step {s}")
    sd, ref = model.hip.state_dict(), ora.state_dict()
    worst = {k: _maxdiff(sd[k], w) for k, w in ref.items()}
    print(f"B={B} through fit(): max |device - reference port| per tensor:", worst)
    # (1e-5 per 1e-3 of learning rate, as everywhere; at 10 000 rows the discriminator's first layer sums 20 000 signed terms per
    #  element and Adam's first steps move an element by ~lr whatever its gradient's size - measured 9.5e-6 on ONE element of
    #  disc.lin1 against MKL on this box's thread count: a handful of elements may reach 3e-5, none beyond)
    for k, w in ref.items():
        d = np.abs(sd[k].astype(np.float64) - w)
        assert d.max() <= 3e-5 and int((d > 1e-5).sum()) <= 4, (k, float(d.max()), int((d > 1e-5).sum()))
    Xp = X[:512]
    np.testing.assert_allclose(model.predict(Xp), ora.predict(Xp.toarray().astype(np.float32)), atol=1e-5)


class _IdentityDist:
    """torch.distributed stand-in of ONE rank of `world`: all-reduce = identity (the other ranks' partial sums are zero)."""
    class ReduceOp:
        SUM = "sum"

    def __init__(self, world, backend):
        self.world, self.backend = world, backend

    def get_rank(self, group=None):
        return 0

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return self.backend

    def all_reduce(self, t, op=None, group=None, async_op=False):
        return None


def test_c5_rank_step_of_the_item_sharded_scheme():
    """Config C5 under dp_mode='shard' (the default data-parallel scheme since r4; aae_shard_step): what ONE of the 8 ranks
    runs per step - its 275 000 of the 2.2 M items (interleaved ownership: rank r holds items r, r + 8, ...) of enc.lin1 and
    dec.lin3 with their optimiser states, a full copy of the hidden layers, and the WHOLE global batch of 512 documents
    through the hidden stacks: first layer's share (complete-document L1 norms), chain programs over 512 rows, the
    row-blocked output layer over 275 000 x 512, the per-item first-layer updates, disc and gen phases - as ONE library call
    with the three all-reduces replaced by the identity (aae_echo_collectives: the other ranks' partial sums taken as zero).
    Two consecutive steps against the NumPy stand-in of tests/test_parallel_gloo.py (ShardStandIn, itself held to the
    reference's fixtures over real gloo collectives there) with the same identity all-reduce: losses, the item slices of
    both vocabulary-wide layers, every hidden layer."""
    from aaerec._hip import HipAAE, DeviceCSR
    from aaerec.parallel import ItemShardedAAE
    from test_parallel_gloo import ShardStandIn
    from tools.synth import init_params, throughput_corpus
    N, world, rank, h, c, B = 2200000, 8, 3, 200, 50, 512
    Ns = len(range(rank, N, world))
    rng = np.random.default_rng(31)
    params = init_params(Ns, h, c, seed=7)            # (hidden layers at their widths; the two big layers at the slice's size)
    params["enc.lin1.weight"] = (params["enc.lin1.weight"] * np.float32(20.0)).astype(np.float32)      # (1/sqrt(275 000) initial scale: lift a1 off zero)
    Xg = throughput_corpus(2 * B, N, median_len=60, seed=19)
    l1 = np.asarray(abs(Xg).sum(1)).reshape(-1).astype(np.float32)
    X = Xg[:, rank::world].tocsr()
    X.sort_indices()
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=2e-3)
    sl = HipAAE(Ns, h, c, max_batch=B, rng_mode="inject", blocked_output=True, **kw)
    sl.load_params(params)
    sl.set_doc_l1(torch.from_numpy(l1).to(sl.device))
    sh = ItemShardedAAE(None, sl, _IdentityDist(world, "echo"), N, interleaved=True)
    ref = ShardStandIn({k: v.copy() for k, v in params.items()}, 0, Ns, 0, **kw)
    ref.set_doc_l1(l1)
    rsh = ItemShardedAAE(None, ref, _IdentityDist(world, "none"), N, interleaved=True)
    assert abs(sh.n_slice - Ns) == 0 and rsh.n_slice == Ns
    csr = DeviceCSR(X, sl.device)
    host = (X.indptr.astype(np.int64), X.indices, X.data.astype(np.float32))
    for s in range(2):
        masks = _masks(rng, B, h)
        z_real = rng.standard_normal((B, c)).astype(np.float32)
        sh.step(None, 0, B, csr, s * B, B, masks=masks, z_real=z_real)
        rsh.step(None, 0, B, host, s * B, B, masks=masks, z_real=z_real)
        got, want = sl.losses(), (ref.sl.loss, ref.hid.o.losses[1], ref.hid.o.losses[2])
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6, err_msg=f"step {s}")
    assert sh.comm_stats()["collectives"] == 3
    sd = sl.state_dict()
    worst = {"dec.lin3.weight": _maxdiff(sd["dec.lin3.weight"], ref.sl.p["w"]), "dec.lin3.bias": _maxdiff(sd["dec.lin3.bias"], ref.sl.p["b"]),
             "enc.lin1.weight": _maxdiff(sd["enc.lin1.weight"], ref.sl.w1["w1"])}
    for k, v in ref.hid.o.p.items():
        if not k.startswith("dec.lin3") and k != "enc.lin1.weight":
            worst[k] = _maxdiff(sd[k], v)
    print("C5 rank step, dp_mode='shard': max |device - stand-in| per tensor:", worst)
    for k, d in worst.items():
        assert d <= 1e-5, (k, d)


def test_c3_batch_512_row_blocked_step_matches_oracle():
    """bench.py's extra.b512 path against the ORACLE directly (it was pinned only transitively - blocked ==
    three-kernel, three-kernel == oracle elsewhere): |items| = 100 000, hidden 200, batch 512, the handle created as
    AdversarialAutoEncoder creates it for 113..1664-row batches (blocked_output: ONE critical launch for the 5 row blocks,
    dec_opt_blocks_x3_kernel deferred - four passes of <= 6 tiles per workgroup -, the wide tile-bucket builder, the
    one-wave-per-item first-layer update with its hot list, chains over 512 rows).  Two full partial_fit steps with injected
    masks and prior draws against the NumPy oracle over the WHOLE [512, 100 000] problem: losses and every parameter."""
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from tools.synth import init_params, throughput_corpus
    N, h, c, B, steps = 100000, 200, 50, 512, 2
    params = init_params(N, h, c, seed=5)
    X = throughput_corpus(steps * B, N, median_len=20, seed=81)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=1e-3)
    dev = HipAAE(N, h, c, max_batch=B, rng_mode="inject", blocked_output=True, **kw)
    dev.load_params(params)
    ora = O.OracleAAE({k: v.copy() for k, v in params.items()}, **kw)
    csr = DeviceCSR(X, dev.device)
    rng = np.random.default_rng(9)
    for s in range(steps):
        masks, z_real = _masks(rng, B, h), rng.standard_normal((B, c)).astype(np.float32)
        dev.step(csr, s * B, B, masks=masks, z_real=z_real)
        Xb = X[s * B:(s + 1) * B]
        want = ora.partial_fit(Xb.indptr.astype(np.int64), Xb.indices, Xb.data.astype(np.float32), z_real, masks)
        np.testing.assert_allclose(dev.losses(), want, rtol=1e-5, atol=1e-6, err_msg=f"losses, step {s}")
    sd = dev.state_dict()
    worst = {k: _maxdiff(sd[k], w) for k, w in ora.p.items()}
    print("C3 at batch 512 (row-blocked output layer): max |device - oracle| per tensor:", worst)
    for k, d in worst.items():
        assert d <= 1e-5, (k, d)


class _ChunkedWholeVocabulary:
    """tests/test_parallel_gloo.py::BothSliceReplica's interface over ALL items of a 2.2 M-item vocabulary with bounded host
    memory and time: the same arithmetic (first layer aae.py:132-135, BCE aae.py:693-695 with PyTorch's clamps, the three
    eager dense torch.optim.Adam instances over the two vocabulary-wide layers aae.py:798-804) restated with multi-threaded
    torch-CPU ops on item chunks - the NumPy stand-in's [rows, items] temporaries would be 8 x 4.5 GB here."""

    def __init__(self, params, lr, reg_lr, chunk=200000):
        self.W = torch.from_numpy(params["dec.lin3.weight"]).clone()
        self.b = torch.from_numpy(params["dec.lin3.bias"]).clone()
        self.W1T = torch.from_numpy(np.ascontiguousarray(params["enc.lin1.weight"].T))       # item-major [N, h]
        self.N, self.h = self.W.shape
        self.mW, self.vW, self.mb, self.vb, self.t_dec = torch.zeros_like(self.W), torch.zeros_like(self.W), torch.zeros_like(self.b), torch.zeros_like(self.b), 0
        self.m1 = {0: torch.zeros_like(self.W1T), 2: torch.zeros_like(self.W1T)}
        self.v1 = {0: torch.zeros_like(self.W1T), 2: torch.zeros_like(self.W1T)}
        self.t1 = {0: 0, 2: 0}
        self.lr = {0: float(lr), 2: float(reg_lr)}
        self.chunk, self.scale, self.loss, self.l1 = chunk, 1.0, 0.0, None
        self._a1 = self._dh2 = self._da2 = None
        self.w1 = {"w1": np.empty((self.h, 0), dtype=np.float32)}    # (ShardStandIn reads the hidden width off this)

    @staticmethod
    def _adam(p, m, v, g, lr, t):
        b1, b2, eps = 0.9, 0.999, 1e-8
        m.add_((g - m).mul_(1 - b1))                                  # exp_avg.lerp_(grad, 1 - beta1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        p.addcdiv_(m, v.sqrt().div_(bc2 ** 0.5).add_(eps), value=-(lr / bc1))

    def set_doc_l1(self, l1):
        self.l1 = np.asarray(l1, dtype=np.float32)

    def set_grad_scale(self, s):
        self.scale = float(s)

    def a1_rows(self, n):
        if self._a1 is None or self._a1.shape[0] != n:
            self._a1 = torch.zeros(n, self.h)
        return self._a1

    def dh2_rows(self, n):
        if self._dh2 is None or self._dh2.shape[0] != n:
            self._dh2, self._da2 = torch.zeros(n, self.h + 1), torch.zeros(n, self.h + 1)
        return self._dh2

    def da2_rows(self, n):
        return self._da2

    def _xn(self):
        import scipy.sparse as sp
        X = self._X[self._r0:self._r0 + self._B]
        s = 1.0 / np.maximum(self.l1[self._r0:self._r0 + self._B], 1e-12)
        return sp.diags(s.astype(np.float32)) @ X                     # F.normalize(x, p=1, dim=1)

    def first_layer_forward(self, csr=None, row_start=0, n_rows=0, rows=None, bias=None):
        if csr is not None:
            self._X, self._r0, self._B = csr, row_start, n_rows
        a1 = torch.from_numpy(np.asarray(self._xn().astype(np.float32) @ self.W1T.numpy(), dtype=np.float32))
        if bias is not None:
            a1 += bias
        self.a1_rows(self._B).copy_(a1)

    def output_layer_step(self, *a, **kw):
        B, N, h = self._B, self.N, self.h
        h2 = self._dh2[:, :h].contiguous()
        X = self._X[self._r0:self._r0 + B]
        self.t_dec += 1
        loss, da2 = 0.0, torch.zeros(B, h)
        for lo in range(0, N, self.chunk):
            hi = min(N, lo + self.chunk)
            W, b = self.W[lo:hi], self.b[lo:hi]
            xhat = torch.sigmoid(torch.addmm(b, h2, W.t()))
            T = torch.from_numpy(X[:, lo:hi].toarray().astype(np.float32))
            x, t = xhat + 1e-12, T + 1e-12
            lx, l1x = torch.log(x).clamp_(min=-100.0), torch.log1p(-x).clamp_(min=-100.0)
            loss += float((-(t * lx + (1 - t) * l1x)).sum(dtype=torch.float64))
            gx = (x - t) / ((1 - x) * x).clamp_(min=1e-12) * (self.scale / (B * N))
            glog = gx * xhat * (1 - xhat)
            da2 += glog @ W
            gW, gb = glog.t() @ h2, glog.sum(0)
            self._adam(W, self.mW[lo:hi], self.vW[lo:hi], gW, self.lr[0], self.t_dec)
            self._adam(b, self.mb[lo:hi], self.vb[lo:hi], gb, self.lr[0], self.t_dec)
        self.loss = loss / (B * N)
        self._da2[:, :h] = da2

    def first_layer_update(self, which, ga1=None, rows_per_block=0, block_stride=0):
        ga = ga1.numpy().reshape(-1, block_stride)[:, :rows_per_block * self.h].reshape(-1, self.h)
        g = torch.from_numpy(np.asarray(self._xn().T.tocsr().astype(np.float32) @ ga, dtype=np.float32))      # dense [N, h]: Adam is dense
        self.t1[which] += 1
        for lo in range(0, self.N, self.chunk):
            hi = min(self.N, lo + self.chunk)
            self._adam(self.W1T[lo:hi], self.m1[which][lo:hi], self.v1[which][lo:hi], g[lo:hi], self.lr[which], self.t1[which])


def test_c5_whole_vocabulary_on_one_gpu_matches_the_chunked_stand_in():
    """BASELINE.json configs[4] at FULL size on one MI355X: |items| = 2 200 000, hidden 200, batch 512,
    documents of median length 60 - one handle, the whole vocabulary (19 GB arena; dec.lin3 is 1.8 GB: byte offsets up to
    2^31 in the fused output layer's descriptors), the path bench.py's extra.c5_world1 runs.  Two full partial_fit steps with
    injected masks and prior draws against the stand-in of dp_mode='shard' at world 1 (tests/test_parallel_gloo.py::
    ShardStandIn: the NumPy oracle for every hidden layer, held to the reference's fixtures there) with the two
    vocabulary-wide layers restated in item chunks (_ChunkedWholeVocabulary): losses, every parameter of every layer."""
    import psutil
    from aaerec._hip import HipAAE, DeviceCSR
    from test_parallel_gloo import ShardStandIn, BothLocalReplica
    from tools.synth import init_params, throughput_corpus
    if psutil.virtual_memory().available < 48 * 2 ** 30:
        pytest.skip("needs ~40 GB of host memory for the whole-vocabulary stand-in")
    N, h, c, B, steps = 2200000, 200, 50, 512, 2
    rng = np.random.default_rng(41)
    params = init_params(N, h, c, seed=9)
    params["enc.lin1.weight"] = (params["enc.lin1.weight"] * np.float32(20.0)).astype(np.float32)      # (1/sqrt(2.2 M) initial scale: lift a1 off zero)
    X = throughput_corpus(steps * B, N, median_len=60, seed=23).tocsr()
    X.sort_indices()
    l1 = np.asarray(abs(X).sum(1)).reshape(-1).astype(np.float32)
    kw = dict(dropout=(0.2, 0.2), gen_lr=1e-3, reg_lr=2e-3)
    dev = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, rng_mode="inject", blocked_output=True, **kw)
    dev.load_params(params)
    small = {k: v for k, v in params.items() if k not in ("enc.lin1.weight", "dec.lin3.weight", "dec.lin3.bias")}
    small["enc.lin1.weight"] = np.zeros((h, 8), dtype=np.float32)       # (the hidden-layer stand-in never touches the wide layers)
    small["dec.lin3.weight"], small["dec.lin3.bias"] = np.zeros((8, h), dtype=np.float32), np.zeros(8, dtype=np.float32)
    ref = ShardStandIn.__new__(ShardStandIn)
    ref.hid, ref.rank, ref.collectives = BothLocalReplica(small, **kw), 0, 0
    ref.sl = _ChunkedWholeVocabulary(params, kw["gen_lr"], kw["reg_lr"])
    ref.sl.set_doc_l1(l1)
    del params["dec.lin3.weight"], params["enc.lin1.weight"]
    csr = DeviceCSR(X, dev.device)
    coll = (_IdentityDist(1, "none"), None)
    for s in range(steps):
        masks, z_real = _masks(rng, B, h), rng.standard_normal((B, c)).astype(np.float32)
        dev.step(csr, s * B, B, masks=masks, z_real=z_real)
        ref.shard_step(coll, X, s * B, B, 1.0, masks=masks, z_real=z_real)
        want = (ref.sl.loss, ref.hid.o.losses[1], ref.hid.o.losses[2])
        np.testing.assert_allclose(dev.losses(), want, rtol=2e-5, atol=1e-6, err_msg=f"step {s}")
    assert ref.collectives == 3 * steps
    sd = dev.state_dict()
    worst = {"dec.lin3.weight": _maxdiff(sd["dec.lin3.weight"], ref.sl.W.numpy()), "dec.lin3.bias": _maxdiff(sd["dec.lin3.bias"], ref.sl.b.numpy()),
             "enc.lin1.weight": _maxdiff(sd["enc.lin1.weight"].T, ref.sl.W1T.numpy())}
    for k, v in ref.hid.o.p.items():
        if not k.startswith("dec.lin3") and k != "enc.lin1.weight":
            worst[k] = _maxdiff(sd[k], v)
    print("C5 whole on one GPU: max |device - stand-in| per tensor:", worst)
    for k, d in worst.items():
        # 1e-5 per 1e-3 of learning rate (the fixture tests' bar at lr = 1e-3): an element whose gradient is at Adam's eps moves by
        # up to a whole step either way, and disc_optim / gen_optim step with reg_lr = 2e-3 here (first box: 1.1e-5 on ONE
        # element of disc.lin1.weight, everything else <= 1e-6)
        tol = 2e-5 if k.startswith(("disc.", "enc.")) else 1e-5
        assert d <= tol, (k, d)
