#!/usr/bin/env python3
"""Soak: a few thousand steps at the headline shape; losses stay finite and move, no NaN anywhere in the arena."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus
from oracle.dense_torch_port import init_params
N, h, c, B, steps = 100000, 200, 50, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
X = throughput_corpus(256 * B, N, seed=7)
m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, seed=3)
m.load_params(init_params(N, h, c, seed=0))
csr = DeviceCSR(X, m.device)
hist = []
for i in range(steps):
    m.step(csr, (i % 256) * B, B)
    if i % 500 == 0 or i == steps - 1:
        hist.append((i, m.losses()))
for i, l in hist:
    print(i, [round(x, 5) for x in l])
sd = m.state_dict()
bad = [k for k, v in sd.items() if not np.isfinite(v).all()]
print("non-finite tensors:", bad)
assert not bad and all(np.isfinite(l).all() for _, l in hist)
print("soak ok")
