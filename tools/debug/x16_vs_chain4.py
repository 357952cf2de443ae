#!/usr/bin/env python3
"""Same model, same inputs, the layer chains on chain16x3 (forced) vs chain4: parameter differences per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import throughput_corpus, init_params
N, h, c, B = int(os.environ.get("XN", 5000)), 200, 50, int(os.environ.get("XB", 100))
steps = int(os.environ.get("XS", 30))
X = throughput_corpus(steps * B, N, seed=3)
p = init_params(N, h, c, seed=1)
kw = dict(max_batch=B, max_nnz=B * 256, rng_mode="inject", dropout=(0.0, 0.0))
os.environ["AAE_X16_ROWS"] = "1"
a = HipAAE(N, h, c, **kw); a.load_params(p)
del os.environ["AAE_X16_ROWS"]; os.environ["AAE_NO_X16"] = "1"
b = HipAAE(N, h, c, **kw); b.load_params(p)
csr = DeviceCSR(X, a.device)
rng = np.random.default_rng(0)
for s in range(steps):
    zr = rng.standard_normal((B, c)).astype(np.float32)
    for m in (a, b):
        m.step(csr, s * B, B, z_real=zr)
    if s in (0, 1, 2, 4, 9, 19, steps - 1):
        sa, sb = a.state_dict(), b.state_dict()
        worst = {k: float(np.abs(sa[k] - sb[k]).max() / (np.abs(sb[k]).max() + 1e-30)) for k in sa}
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
        print(f"step {s}: losses {a.losses()} | {b.losses()}")
        print("   max |diff| / max |w|:", ", ".join(f"{k} {v:.2e}" for k, v in top))
