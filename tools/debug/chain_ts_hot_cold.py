"""Debug: per-op in-kernel timeline of the layer-chain programs (AAE_CHAIN_TS=1) in two cache states: inside training steps
(every weight matrix was rewritten by an optimiser kernel since the last program that read it) and in repeated encode /
predict calls (weights untouched between programs: hot in the reading XCD's L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import init_params, throughput_corpus
N, h, c, B = 100000, 200, 50, 100
m = HipAAE(N, h, c, max_batch=B, rng_mode="device", seed=1)
m.load_params(init_params(N, h, c, seed=3))
X = throughput_corpus(8 * B, N, median_len=20, seed=7)
csr = DeviceCSR(X, m.device)
print("---- training steps", file=sys.stderr, flush=True)
for s in range(3):
    m.step(csr, s * B, B)
torch.cuda.synchronize()
print("---- repeated encode (weights untouched)", file=sys.stderr, flush=True)
for s in range(4):
    m.encode(csr, 0, B)
torch.cuda.synchronize()
