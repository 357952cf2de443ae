#!/usr/bin/env python3
"""Host enqueue time per aae_step (no waiting on the GPU inside the loop) next to the GPU-bound step time:
tells whether bench.py's step rate is limited by kernel time or by the launch path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(64 * B, N, seed=1234)
m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256)
from tools.synth import init_params
m.load_params(init_params(N, h, c, seed=0))
csr = DeviceCSR(X, m.device)
for i in range(20):
    m.step(csr, (i % 64) * B, B)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for i in range(K):
    m.step(csr, (i % 64) * B, B)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step, GPU tail after last enqueue {1e3 * (t2 - t1):.2f} ms")
