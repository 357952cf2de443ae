#!/usr/bin/env python3
"""Throughput of the host-facing entry points at the headline config (numbers quoted in
DESIGN.md section 6): the raw C-ABI step loop (what bench.py times), AdversarialAutoEncoder.fit
(epoch loop: permutation upload + row-id batches, corpus resident in HBM), and
AdversarialAutoEncoder.partial_fit on host batches (CSR upload over PCIe on every call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from aaerec.aae import AdversarialAutoEncoder
    from tools.synth import throughput_corpus
    N, h, B = 100000, 200, 100
    X = throughput_corpus(64 * B, N, seed=1234)
    out = {}
    m = AdversarialAutoEncoder(n_hidden=h, n_code=50, n_epochs=1, batch_size=B, verbose=False)
    m.fit(X)                                   # warm-up epoch (build, first launches)
    torch.cuda.synchronize()
    m.n_epochs = 3
    t0 = time.perf_counter()
    m.fit(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["fit() docs/s (incl. model build + H2D of the corpus, 3 epochs)"] = 3 * X.shape[0] / dt
    # partial_fit with host batches: CSR slice -> H2D every call
    batches = [X[i * B:(i + 1) * B] for i in range(64)]
    for b in batches[:5]:
        m.partial_fit(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        m.partial_fit(b)
    torch.cuda.synchronize()
    out["partial_fit(host csr batch) docs/s (PCIe-inclusive)"] = 64 * B / (time.perf_counter() - t0)
    dense = [b.toarray().astype(np.float32) for b in batches[:16]]
    t0 = time.perf_counter()
    for d in dense:
        m.partial_fit(d)
    torch.cuda.synchronize()
    out["partial_fit(host dense ndarray, as the reference passes it) docs/s"] = 16 * B / (time.perf_counter() - t0)
    # predict: full [n, N] float32 matrix to the host vs. top-10 ids only
    Xt = X[:2000]
    m.predict(Xt[:200]); m.predict_topk(Xt[:200], k=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.predict(Xt); torch.cuda.synchronize()
    out["predict() docs/s (dense [n,N] scores to host)"] = Xt.shape[0] / (time.perf_counter() - t0)
    t0 = time.perf_counter(); m.predict_topk(Xt, k=10); torch.cuda.synchronize()
    out["predict_topk(k=10) docs/s (scaling, masking, top-k on device)"] = Xt.shape[0] / (time.perf_counter() - t0)
    for k, v in out.items():
        print(f"{k}: {v:,.0f}")


if __name__ == "__main__":
    main()
