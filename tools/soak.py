#!/usr/bin/env python3
"""Soak: a few thousand steps at the headline shape; losses stay finite and move, no NaN anywhere in the arena."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus
from tools.synth import init_params
N, h, c, B, steps = 100000, 200, 50, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"       # fused | vocab | replicated (one rank, gradient-export path)
X = throughput_corpus(256 * B, N, seed=7)
params = init_params(N, h, c, seed=0)
if mode == "fused":
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, seed=3)
    m.load_params(params)
    csr = DeviceCSR(X, m.device)
    step, losses = (lambda i: m.step(csr, (i % 256) * B, B)), m.losses
else:
    from aaerec.parallel import DataParallelAAE, VocabParallelAAE

    class Solo:                      # torch.distributed stand-in for one rank
        class ReduceOp:
            SUM = 0

        def get_world_size(self, group=None): return 1
        def get_rank(self, group=None): return 0
        def get_backend(self, group=None): return "solo"
        def all_reduce(self, t, op=None, group=None, async_op=False): return None
        def all_gather_into_tensor(self, out, inp, group=None, async_op=False): out.copy_(inp.reshape(out.shape))
        def reduce_scatter_tensor(self, out, inp, op=None, group=None, async_op=False): out.copy_(inp.reshape(out.shape))
    w1_cap = int(X.getnnz(1).reshape(-1, B).sum(1).max()) + 8
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, seed=3, grad_mode="export", dp_world=1, w1_cap=w1_cap)
    m.load_params(params)
    csr = DeviceCSR(X, m.device)
    if mode == "vocab":
        sl = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, seed=3)
        sl.load_params(params)
        dp = VocabParallelAAE(m, sl, Solo(), N)
        step = lambda i: dp.step(csr, (i % 256) * B, B, csr, (i % 256) * B, B)        # noqa: E731
        losses = lambda: (dp.recon_loss(),) + tuple(m.losses()[1:])                   # noqa: E731
    else:
        dp = DataParallelAAE(m, Solo(), shard_decoder=False)
        step, losses = (lambda i: dp.step(csr, (i % 256) * B, B, global_rows=B)), m.losses
hist = []
for i in range(steps):
    step(i)
    if i % 5000 == 0 or i == steps - 1:
        hist.append((i, losses()))
for i, l in hist:
    print(i, [round(x, 5) for x in l])
sd = m.state_dict()
bad = [k for k, v in sd.items() if not np.isfinite(v).all()]
print("non-finite tensors:", bad)
assert not bad and all(np.isfinite(l).all() for _, l in hist)
print("soak ok")
