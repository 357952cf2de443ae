#!/usr/bin/env python3
"""Per-op in-kernel timeline of the layer-chain programs of one step at C3 (AAE_CHAIN_TS=1 must be set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus, init_params
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(8 * B, N, seed=1234)
m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256)
m.load_params(init_params(N, h, c, seed=0))
csr = DeviceCSR(X, m.device)
for i in range(6):
    if i == 5:
        print("---- step 5", file=sys.stderr)
    m.step(csr, (i % 8) * B, B)
torch.cuda.synchronize()
