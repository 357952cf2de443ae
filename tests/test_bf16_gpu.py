"""bf16 mode of the build (BASELINE config C2: "bf16 MFMA inputs, fp32 accumulate, fp32 master parameters and Adam",
SURVEY section 7).  The reference has no bf16 path, so parity is established in two steps:

  1. the kernels do exactly what the mode is defined to do: GPU vs the NumPy oracle with the SAME operand rounding
     (oracle/aae_oracle.py, bf16=True: every matrix-core product takes both operands rounded to bf16 - ties to even,
     as v_cvt_pk_bf16_f32 - and accumulates in fp32) at the fp32 tests' tolerances;
  2. the mode stays close to the reference: the fp32 fixtures generated from the real reference, replayed in bf16 mode
     with the recorded randomness, within the stated bf16 bounds (losses 2e-3 relative; parameters 2.5e-3 absolute =
     a few Adam steps of lr = 1e-3, whose direction flips where a gradient is within bf16 rounding of zero), and the
     end-to-end MRR@10 of config C1 in distribution (tests/test_host_gpu.py).
"""
import numpy as np
import pytest
import torch

from golden_util import Fixture

pytestmark = pytest.mark.gpu


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


class _SliceEmu:
    """aae_output_layer_step in bf16 mode, restated: logits = R(dh2) R(V3)^T + R(b3); BCE and its gradient in fp32;
    dA2 = R(G) R(V3); dV3 = R(G)^T R(dh2), db3 = sum_b R(G); Adam in fp32 on the fp32 master weights."""

    def __init__(self, w, b, lr, scale):
        from oracle.aae_oracle import Adam
        self.p = {"w": w.copy(), "b": b.copy()}
        self.opt, self.scale = Adam(lr), scale

    def step(self, dh2, X):
        from oracle.aae_oracle import TINY, bf16_round as R, f32, sigmoid
        B, Ns = X.shape
        h2 = R(dh2[:, :-1])
        logits = (h2 @ R(self.p["w"]).T + R(self.p["b"])).astype(f32)
        xhat = sigmoid(logits)
        T = np.asarray(X.todense(), dtype=f32)
        x, t = xhat + TINY, T + TINY
        with np.errstate(divide="ignore"):
            lx, l1x = np.maximum(np.log(x), f32(-100)), np.maximum(np.log1p(-x), f32(-100))
        loss = float((-(t * lx + (f32(1) - t) * l1x)).mean(dtype=np.float64))
        gx = (x - t) / np.maximum((f32(1) - x) * x, f32(1e-12)) * f32(self.scale / (B * Ns))
        G = R((gx * xhat * (f32(1) - xhat)).astype(f32))
        da2 = (G @ R(self.p["w"])).astype(f32)
        self.opt.step(self.p, {"w": (G.T @ h2).astype(f32), "b": G.sum(0).astype(f32)})
        return loss, da2


@pytest.mark.parametrize("N,h,B", [(1000, 50, 37), (4999, 100, 100), (3333, 200, 104), (70, 20, 5), (6400, 200, 112),
                                   (5000, 100, 300), (9000, 200, 230), (4587, 200, 1000)])
def test_bf16_fused_output_layer_matches_the_rounded_restatement(N, h, B):
    """(B > 112, late r4: the row-blocked form on the rounded-operand kernels - one critical launch for all row blocks, the
    deferred half of every block in one launch - which bf16 mode takes for wide batches since then)"""
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import throughput_corpus
    rng = np.random.default_rng(N + h + B)
    k = 1.0 / np.sqrt(h)
    w = ((rng.random((N, h)) * 2 - 1) * k).astype(np.float32)
    b = ((rng.random(N) * 2 - 1) * k).astype(np.float32)
    X = throughput_corpus(2 * B, N, median_len=min(12, N // 4), seed=N)
    X.data[::3] = 0.5                                                   # targets strictly inside (0, 1) as well
    sl = HipAAE(N, h, 10, max_batch=B, rng_mode="inject", dtype="bf16", blocked_output=B > 112)
    sl.load_params({"dec.lin3.weight": w, "dec.lin3.bias": b})
    sl.set_grad_scale(0.5)
    emu = _SliceEmu(w, b, 1e-3, 0.5)
    csr = DeviceCSR(X, sl.device)
    for s in range(2):
        dh2 = np.abs(rng.standard_normal((B, h + 1))).astype(np.float32) * 0.5
        dh2[rng.random((B, h + 1)) < 0.4] = 0.0
        dh2[:, h] = 1.0
        sl.dh2_rows(B)[:, :h + 1].copy_(torch.from_numpy(dh2))
        sl.output_layer_step(csr, s * B, B)
        loss, da2 = emu.step(dh2, X[s * B:(s + 1) * B])
        np.testing.assert_allclose(sl.losses()[0], loss, rtol=1e-5)
        got = sl.da2_rows(B)[:, :h].cpu().numpy()
        assert _maxdiff(got, da2) <= 2e-4 * float(np.abs(da2).max()) + 1e-12, (s, _maxdiff(got, da2), float(np.abs(da2).max()))
    sd = sl.state_dict()
    assert _maxdiff(sd["dec.lin3.weight"], emu.p["w"]) <= 1e-5
    assert _maxdiff(sd["dec.lin3.bias"], emu.p["b"]) <= 1e-5


# ---- whole partial_fit steps in bf16 mode ------------------------------------------------------------------------------
BF16_STEP_CASES = ["step_masks", "step_cond_concat", "step_selu", "step_categorical_prior", "step_ragged", "step_wide",
                   "step_headline", "step_c4"]


def _run_bf16(name):
    """(device model, rounded oracle, per-step device losses, per-step oracle losses) after replaying fixture `name`"""
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    fx = Fixture(name)
    c = fx.cfg
    kw = fx.model_kwargs()
    dev = HipAAE(c["N"], c["h"], c["c"], cond_inc=c["cond_inc"], max_batch=c["B"], rng_mode="inject", dtype="bf16", **kw)
    dev.load_params(fx.init_params())
    conds = [O.ConcatConst(c["cond_inc"])] if c["cond_inc"] else []
    ora = O.OracleAAE(fx.init_params(), conditions=conds, bf16=True, **kw)
    got, want = [], []
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        csr = DeviceCSR.from_arrays(ip, idx, val, c["N"], dev.device)
        cond = fx.cond_inputs(s)
        dev.step(csr, 0, csr.shape[0], cond=torch.as_tensor(cond[0], device=dev.device) if cond else None,
                 masks=fx.masks(s), z_real=fx.z[f"step{s}.z_real"])
        got.append(dev.losses())
        want.append(ora.partial_fit(ip, idx, val, fx.z[f"step{s}.z_real"], fx.masks(s), cond))
    return fx, dev, ora, np.asarray(got), np.asarray(want)


@pytest.mark.parametrize("name", BF16_STEP_CASES)
def test_bf16_step_matches_the_rounded_oracle(name):
    """Every GEMM-shaped product of the step with bf16-rounded operands (chain layers, per-layer GEMMs, fused and
    three-kernel output layer), against oracle(bf16=True).  An fp32 summation-order difference that straddles a bf16
    rounding boundary of an activation moves that operand by 2^-8 relative: rare (about 1e-5 of the elements), bounded,
    and the reason the tolerances are 10x the fp32 tests' - except for the few parameters whose Adam step direction
    flips because their gradient is within that noise of zero: Adam's first steps move a parameter by ~lr whatever the
    size of its gradient, so such an element ends up to ~2 lr per optimiser step away (at most 1 % of a tensor: the
    encoder's adversarial gradients of gen_step are the smallest and flip most)."""
    fx, dev, ora, got, want = _run_bf16(name)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6, err_msg=f"{name} losses")
    sd = dev.state_dict()
    lr = max(fx.cfg.get("gen_lr", 1e-3), fx.cfg.get("reg_lr", 1e-3))
    stats = {}
    for k, w in ora.p.items():
        d = np.abs(sd[k].astype(np.float64) - w)
        n_opt = 2 if k.startswith("enc.") else 1           # the encoder takes two optimiser steps per partial_fit
        stats[k] = (float((d > 1e-4 * max(1.0, lr / 1e-3)).mean()), float(d.max()), 3.0 * lr * fx.steps * n_opt)
    print(name, {k: (round(f, 5), round(mx, 6)) for k, (f, mx, _) in stats.items()})
    for k, (frac, mx, bound) in stats.items():
        assert frac <= 1e-2 and mx <= bound, (name, k, frac, mx, bound)


@pytest.mark.parametrize("name", BF16_STEP_CASES)
def test_bf16_step_matches_the_rounded_oracle_with_the_chain_kernels_k_slice_form(name, monkeypatch):
    """The same with every linear op of the 4-row chain kernel in the k-slice form (a wave = 64 columns x a slice of K, the
    slices' partial sums through LDS, two barriers per op; AAE_CHAIN_KSLICES=1).  Since r6 bf16 mode takes the column-owner
    form for batches of one fused launch (csrc/chain4.h: a wave owns 16 columns for all of K, the four k-residues of a 4x4x1
    instruction's blocks added across the lanes, the epilogue on the accumulators, one barrier per op) - which the test above
    replays; the k-slice form carries fp32 mode, wide batches, layers in place, matrices without a k4-interleaved copy."""
    monkeypatch.setenv("AAE_CHAIN_KSLICES", "1")
    test_bf16_step_matches_the_rounded_oracle(name)


@pytest.mark.parametrize("name", ["step_masks", "step_headline", "step_wide", "step_c4"])
def test_bf16_stays_within_its_bound_of_the_reference(name):
    """The same replay against what the REFERENCE (fp32) recorded: the stated bf16 bound of the mode - losses within 1 %
    (bf16 has 8 significant bits; the losses are means over thousands of cells), the bulk of the parameters within
    4e-4 (a tenth of an Adam step of lr = 1e-3 ... 2e-3 after 2-3 steps), every parameter within 2.5 lr per step."""
    fx, dev, ora, got, want = _run_bf16(name)
    ref = np.asarray([fx.z[f"step{s}.losses"] for s in range(fx.steps)])
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-5, err_msg=f"{name} losses vs the reference's fp32 run")
    s_last = fx.steps - 1
    sd = dev.state_dict()
    lr = max(fx.cfg.get("gen_lr", 1e-3), fx.cfg.get("reg_lr", 1e-3))
    for k, w in fx.expected_params(s_last).items():
        d = np.abs(sd[k].astype(np.float64) - w)
        n_opt = 2 if k.startswith("enc.") else 1
        assert float(np.median(d)) <= 4e-4 * max(1.0, lr / 1e-3) and d.max() <= 3.0 * lr * fx.steps * n_opt, (name, k, float(np.median(d)), float(d.max()))
    # reconstructions of the trained weights (eval mode, aae.py:840-870): sigmoid outputs within 2e-3 of the reference's
    from aaerec._hip import DeviceCSR
    ip, idx, val = fx.batch(0, prefix="predict")
    pcsr = DeviceCSR.from_arrays(ip, idx, val, fx.cfg["N"], dev.device)
    pc = fx.cond_inputs(0, prefix="predict")
    out = dev.predict(pcsr, 0, pcsr.shape[0], cond=torch.as_tensor(pc[0], device=dev.device) if pc else None).cpu().numpy()
    assert _maxdiff(out, fx.z["predict.out"]) <= 5e-3
