#!/usr/bin/env python3
"""A short AdversarialAutoEncoder.fit run at C2 with bf16 matrix-core inputs, for tracing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus
N, h, c, B = 47000, 100, 50, 100
X = throughput_corpus(64 * B, N, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1, dtype=os.environ.get("DT", "bf16"))
it = m.fit_steps(X)
for _ in range(230):
    next(it)
torch.cuda.synchronize()
