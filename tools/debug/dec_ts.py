"""Debug: a few C3 steps with the in-kernel phase timeline of the output layer (AAE_DEC_TS=1: single launch, =x3: the split
form's critical launch on the bf16 matrix cores) printed to stderr."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import init_params, throughput_corpus
N, h, c, B = int(os.environ.get("N", 100000)), 200, 50, int(os.environ.get("B", 100))
m = HipAAE(N, h, c, max_batch=B, rng_mode="device", seed=1)
m.load_params(init_params(N, h, c, seed=3))
X = throughput_corpus(8 * B, N, median_len=20, seed=7)
csr = DeviceCSR(X, m.device)
for s in range(6):
    m.step(csr, s * B, B)
torch.cuda.synchronize()
