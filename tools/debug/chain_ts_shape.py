#!/usr/bin/env python3
"""Per-op in-kernel timeline of the layer-chain programs of one step at any shape (AAE_CHAIN_TS=1 must be set):
    CT_N=4587 CT_B=1000 CT_COND=300 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus, init_params
N, h, c = int(os.environ.get("CT_N", 100000)), int(os.environ.get("CT_H", 200)), 50
B, ci = int(os.environ.get("CT_B", 100)), int(os.environ.get("CT_COND", 0))
X = throughput_corpus(8 * B, N, seed=1234)
m = HipAAE(N, h, c, cond_inc=ci, max_batch=B, max_nnz=B * 256)
m.load_params(init_params(N, h, c, cond_inc=ci, seed=0))
csr = DeviceCSR(X, m.device)
cond = torch.randn(8 * B, ci, device=m.device) * 0.1 if ci else None
for i in range(6):
    if i == 5:
        print(f"---- step 5 (N={N} h={h} B={B} cond={ci})", file=sys.stderr)
    s0 = (i % 8) * B
    m.step(csr, s0, B, cond=None if cond is None else cond[s0:s0 + B])
torch.cuda.synchronize()
