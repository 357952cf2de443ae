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


@pytest.mark.parametrize("N,h,B", [(1000, 50, 37), (4999, 100, 100), (3333, 200, 104), (70, 20, 5), (6400, 200, 112)])
def test_bf16_fused_output_layer_matches_the_rounded_restatement(N, h, B):
    from aaerec._hip import HipAAE, DeviceCSR
    from tools.synth import throughput_corpus
    rng = np.random.default_rng(N + h + B)
    k = 1.0 / np.sqrt(h)
    w = ((rng.random((N, h)) * 2 - 1) * k).astype(np.float32)
    b = ((rng.random(N) * 2 - 1) * k).astype(np.float32)
    X = throughput_corpus(2 * B, N, median_len=min(12, N // 4), seed=N)
    X.data[::3] = 0.5                                                   # targets strictly inside (0, 1) as well
    sl = HipAAE(N, h, 10, max_batch=B, rng_mode="inject", dtype="bf16")
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
