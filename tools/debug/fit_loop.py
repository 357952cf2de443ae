#!/usr/bin/env python3
"""A short AdversarialAutoEncoder.fit run at C3 (the path bench.py's `value` times), for tracing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(64 * B, N, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1)
it = m.fit_steps(X)
for _ in range(30):
    next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 200
for _ in range(K):
    next(it)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fit loop: host {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step")
