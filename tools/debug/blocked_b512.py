#!/usr/bin/env python3
"""Raw-step rate at C3 with batches beyond one fused launch (512 rows) on ONE handle: three-kernel output layer vs the
row-blocked fused form (HipAAE(blocked_output=True)).  Usage on the GPU box: python tools/debug/blocked_b512.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import init_params, throughput_corpus

N, h, c = 100000, 200, 50
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
X = throughput_corpus(16 * B, N, median_len=20, seed=5)
params = init_params(N, h, c, seed=0)
for blocked in (False, True):
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, blocked_output=blocked)
    m.load_params(params)
    csr = DeviceCSR(X, m.device)
    for i in range(10):
        m.step(csr, (i % 16) * B, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        m.step(csr, (i % 16) * B, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"rows {B} blocked={blocked}: {1e3 * dt:.3f} ms/step, {B / dt:.0f} docs/s, losses {m.losses()}", flush=True)
    del m
    torch.cuda.empty_cache()
