#!/usr/bin/env python3
"""Pure host cost of enqueuing one training step: the loop runs 40 steps ahead of a synchronise, on a model small enough that
the GPU drains the queue faster than the host fills it (so the host never waits for queue space)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus, init_params
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(64 * B, N, seed=1234)
m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256)
m.load_params(init_params(N, h, c, seed=0))
csr = DeviceCSR(X, m.device)
for i in range(20):
    m.step(csr, (i % 64) * B, B)
torch.cuda.synchronize()
tot, n = 0.0, 0
for rep in range(30):
    t0 = time.perf_counter()
    for i in range(8):          # 8 steps = ~150 packets: far below the queue depth, the GPU is idle at t0
        m.step(csr, (i % 64) * B, B)
    tot += time.perf_counter() - t0; n += 8
    torch.cuda.synchronize()
print(f"host cost of one step's enqueue (8-step bursts into an idle queue): {1e3 * tot / n:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(8):
    m.step(csr, (i % 64) * B, B)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(6)
