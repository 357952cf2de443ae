#!/usr/bin/env python3
"""predict -> rank at the headline shape (C3: 100 000 items, hidden 200): docs/s of the library call alone on a resident
test corpus (aae_predict_topk over `rows` documents per call, launch to completion) and of
AdversarialAutoEncoder.predict_topk (host loop, [n, k] results copied to the host) - fused path and, with
AAE_NO_RANK_FUSED=1 in a second process, the r1-r3 two-kernel path.  Per-call HIP-event time of the rank kernel itself
(AAE_K_RANK) against its floors: 2 rows N (h+1) flop on the matrix cores, 4 N (h+1) bytes of dec.lin3 from HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec.aae import AdversarialAutoEncoder
from aaerec._hip import DeviceCSR
from tools.synth import throughput_corpus
N, h, c, B = int(os.environ.get("RR_ITEMS", 100000)), int(os.environ.get("RR_HIDDEN", 200)), 50, 100
DOCS = int(os.environ.get("RR_DOCS", 8192))
K_RANK = 9
X = throughput_corpus(DOCS, N, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, verbose=False, seed=1)
for _ in zip(range(20), m.fit_steps(X)):
    pass
m._fit_finish()
hip = m.hip
csr = DeviceCSR(X, hip.device)
cap = hip.rank_max_rows(10)
print(f"rank_max_rows(10) = {cap}", flush=True)
ROWS = [int(x) for x in os.environ.get("RR_ROWS", "100,256,512,1024,2048").split(",")]
for rows in [r for r in ROWS if r <= cap]:
    reps = max(3, 4096 // rows)
    for timed in (False, True):
        torch.cuda.synchronize()
        if timed:
            hip.profile_enable(True, kernels=(K_RANK,))
        t0 = time.perf_counter()
        for i in range(reps):
            hip.predict_topk(csr, (i * rows) % (DOCS - rows + 1), rows, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    hip.profile_enable(False)
    ms, n = hip.profile_read(K_RANK)
    line = f"rows/call {rows:5d}: {dt * 1e3:8.3f} ms/call  {rows / dt:10.0f} docs/s  ({dt / rows * 1e5 * 1e3:6.1f} us per 100 docs)"
    if n:
        us = ms / n * 1e3
        fl, by = 2.0 * rows * N * (h + 1), 4.0 * N * (h + 1)
        line += f" | rank kernel {us:7.1f} us = {fl / us * 1e-6:6.1f} TFLOP/s ({fl / us * 1e-6 / 157.3:.2f} of fp32 MFMA, {fl / us * 1e-6 / 416.7:.2f} of the emulated product), {by / us * 1e-3:6.0f} GB/s"
    print(line, flush=True)
for name, fn in ((("predict_topk through the model (k=10)", lambda: m.predict_topk(X, k=10)),) if "RR_ROWS" not in os.environ else ()):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:46s} {DOCS / dt:9.0f} docs/s", flush=True)
