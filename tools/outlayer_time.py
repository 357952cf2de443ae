#!/usr/bin/env python3
"""Time the decoder's output layer alone (aae_output_layer_step: fused kernel + its two small reductions) at a
shape, fp32 vs bf16:  python tools/outlayer_time.py [--items N] [--hidden H] [--batch B] [--steps K]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--hidden", type=int, default=200)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--dtypes", default="f32,bf16")
    a = ap.parse_args()
    from aaerec._hip import HipAAE, DeviceCSR, K_DEC_FUSED, K_DEC_BCE_FWD, K_DEC_DA2, K_DEC_DV3_ADAM
    from tools.synth import throughput_corpus
    N, h, B = a.items, a.hidden, a.batch
    X = throughput_corpus(8 * B, N, seed=1)
    rng = np.random.default_rng(0)
    for dt in a.dtypes.split(","):
        m = HipAAE(N, h, 50, max_batch=B, rng_mode="device", dtype=dt)
        k = 1.0 / np.sqrt(h)
        m.load_params({"dec.lin3.weight": ((rng.random((N, h)) * 2 - 1) * k).astype(np.float32),
                       "dec.lin3.bias": np.zeros(N, dtype=np.float32)})
        csr = DeviceCSR(X, m.device)
        dh2 = torch.rand(B, h + 1, device=m.device)
        dh2[:, h] = 1.0
        m.dh2_rows(B)[:, :h + 1].copy_(dh2)
        for i in range(5):
            m.output_layer_step(csr, (i % 8) * B, B)
        torch.cuda.synchronize()
        m.profile_enable(True, kernels=(K_DEC_FUSED, K_DEC_BCE_FWD, K_DEC_DA2, K_DEC_DV3_ADAM))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.steps):
            m.output_layer_step(csr, (i % 8) * B, B)
        e1.record()
        torch.cuda.synchronize()
        m.profile_enable(False)
        per = {n: m.profile_read(kid) for n, kid in (("fused", K_DEC_FUSED), ("bce", K_DEC_BCE_FWD), ("da2", K_DEC_DA2), ("dv3", K_DEC_DV3_ADAM))}
        msg = ", ".join(f"{n} {ms / max(c, 1) * 1e3:.1f} us" for n, (ms, c) in per.items() if c)
        bytes_ = 24.0 * N * (h + 1)
        fused_us = per["fused"][0] / max(per["fused"][1], 1) * 1e3
        print(f"{dt}: N={N} h={h} B={B}: {e0.elapsed_time(e1) / a.steps * 1e3:.1f} us per output_layer_step; {msg}"
              + (f"; fused kernel {bytes_ / fused_us / 1e3:.0f} GB/s algorithmic" if fused_us else ""), flush=True)
        del m


if __name__ == "__main__":
    main()
