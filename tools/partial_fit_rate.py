#!/usr/bin/env python3
"""docs/s of a caller-driven loop of AdversarialAutoEncoder.partial_fit(X_batch) with host-side scipy batches
(every call uploads its CSR batch), next to fit() on the resident corpus."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus
N, h, c, B, DOCS = 100000, 200, 50, 100, 6400
X = throughput_corpus(DOCS, N, seed=1234)
batches = [X[i:i + B] for i in range(0, DOCS, B)]
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, verbose=False, seed=1)
for b in batches[:20]:
    m.partial_fit(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    for b in batches:
        m.partial_fit(b)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"partial_fit(scipy batch) loop: {10 * DOCS / dt:.0f} docs/s ({1e3 * dt / (10 * len(batches)):.3f} ms/step)")
