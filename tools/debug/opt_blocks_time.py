#!/usr/bin/env python3
"""Critical and deferred launch of the row-blocked output layer alone (aae_output_layer_step on a blocked_output handle),
timed by the library's event pairs.  Usage on the GPU box: AAE_BLOCKED_ANY=1 python tools/debug/opt_blocks_time.py [rows] [items]
(AAE_DEC_SKIP ablation bits of dec_opt_blocks_x3_kernel: 0x1000 no MFMA, 0x2000 no optimiser traffic, 0x4000 no dh2 reload,
0x8000 no dL/dlogits loads)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec._hip import HipAAE, DeviceCSR, K_DEC_CRIT, K_DEC_OPT
from tools.synth import throughput_corpus

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
h = 200
X = throughput_corpus(8 * B, N, median_len=20, seed=5)
rng = np.random.default_rng(0)
m = HipAAE(N, h, 50, max_batch=B, rng_mode="device", blocked_output=True)
k = 1.0 / np.sqrt(h)
m.load_params({"dec.lin3.weight": ((rng.random((N, h)) * 2 - 1) * k).astype(np.float32), "dec.lin3.bias": np.zeros(N, dtype=np.float32)})
csr = DeviceCSR(X, m.device)
dh2 = torch.rand(B, h + 1, device=m.device); dh2[:, h] = 1.0
m.dh2_rows(B)[:, :h + 1].copy_(dh2)
for i in range(5):
    m.output_layer_step(csr, (i % 8) * B, B)
torch.cuda.synchronize()
m.profile_enable(True, kernels=(K_DEC_CRIT, K_DEC_OPT))
steps = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    m.output_layer_step(csr, (i % 8) * B, B)
e1.record()
torch.cuda.synchronize()
m.profile_enable(False)
per = {n: m.profile_read(kid) for n, kid in (("crit", K_DEC_CRIT), ("opt", K_DEC_OPT))}
print(f"rows {B} items {N} skip {os.environ.get('AAE_DEC_SKIP', '0')}: {e0.elapsed_time(e1) / steps * 1e3:.1f} us per output_layer_step; " +
      ", ".join(f"{n} {ms / max(c, 1) * 1e3:.1f} us ({c // steps} per step)" for n, (ms, c) in per.items()), flush=True)
