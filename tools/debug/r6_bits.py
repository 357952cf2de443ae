"""Debug: a hash of every parameter and of the losses after a few C3-shaped steps - two builds of the library that claim the
same bits (AAE_HIP_LIB) must print the same line.  N / B / STEPS / DTYPE from the environment."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import init_params, throughput_corpus
N, h, c, B = int(os.environ.get("N", 100000)), int(os.environ.get("H", 200)), 50, int(os.environ.get("B", 100))
steps = int(os.environ.get("STEPS", 6))
m = HipAAE(N, h, c, max_batch=B, rng_mode="device", seed=1, dtype=os.environ.get("DTYPE", "f32"))
m.load_params(init_params(N, h, c, seed=3))
X = throughput_corpus(8 * B, N, median_len=20, seed=7)
csr = DeviceCSR(X, m.device)
hs, hl = hashlib.sha256(), hashlib.sha256()
for s in range(steps):
    m.step(csr, (s % 8) * B, B)
    hl.update(np.asarray(m.losses(), dtype=np.float32).tobytes())
torch.cuda.synchronize()
sd = m.state_dict()
for k in sorted(sd):
    hs.update(np.ascontiguousarray(sd[k]).tobytes())
print("bits", f"N={N} B={B} h={h}", "parameters", hs.hexdigest()[:16], "loss series", hl.hexdigest()[:16], "losses", m.losses())
