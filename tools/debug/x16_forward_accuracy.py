#!/usr/bin/env python3
"""Forward accuracy of the hidden stacks: encode() (first layer gather + enc.lin2 + enc.lin3) on chain16x3 (forced) and
on chain4 against a float64 NumPy forward of the same parameters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus, init_params
N, h, c, B = 5000, 200, 50, 100
X = throughput_corpus(B, N, seed=3).tocsr()
p = init_params(N, h, c, seed=1)
for k in p:
    if "weight" in k and not k.startswith("enc.lin1"):
        p[k] = (p[k] * 3.0).astype(np.float32)      # spread the activations
kw = dict(max_batch=B, max_nnz=B * 256, rng_mode="inject", dropout=(0.0, 0.0))
os.environ["AAE_X16_ROWS"] = "1"
a = HipAAE(N, h, c, **kw); a.load_params(p)
del os.environ["AAE_X16_ROWS"]; os.environ["AAE_NO_X16"] = "1"
b = HipAAE(N, h, c, **kw); b.load_params(p)
csr = DeviceCSR(X, a.device)
za, zb = a.encode(csr, 0, B).cpu().numpy().astype(np.float64), b.encode(csr, 0, B).cpu().numpy().astype(np.float64)
Xd = np.asarray(X.todense(), dtype=np.float64)
Xn = Xd / np.maximum(np.abs(Xd).sum(1, keepdims=True), 1e-12)
P = {k: v.astype(np.float64) for k, v in p.items()}
h1 = np.maximum(Xn @ P["enc.lin1.weight"].T + P["enc.lin1.bias"], 0)
h2 = np.maximum(h1 @ P["enc.lin2.weight"].T + P["enc.lin2.bias"], 0)
z = h2 @ P["enc.lin3.weight"].T + P["enc.lin3.bias"]
s = np.abs(z).max()
print("max |z| %.3f;  max |z_x16 - z64| / max|z| = %.3e;  chain4: %.3e;  x16 vs chain4: %.3e" % (s, np.abs(za - z).max() / s, np.abs(zb - z).max() / s, np.abs(za - zb).max() / s))
pa, pb = a.predict(csr, 0, B).cpu().numpy().astype(np.float64), b.predict(csr, 0, B).cpu().numpy().astype(np.float64)
print("predict: max |x16 - chain4| = %.3e (max %.3f)" % (np.abs(pa - pb).max(), pb.max()))
