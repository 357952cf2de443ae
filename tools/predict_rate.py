#!/usr/bin/env python3
"""docs/s of AdversarialAutoEncoder.predict (the dense [docs, items] score matrix on the host, as the reference's
API returns it) next to predict_topk (ranking on the device) at the headline shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus
N, h, c, B, DOCS = 100000, 200, 50, 100, 3200
X = throughput_corpus(DOCS, N, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, verbose=False, seed=1)
m.fit(X)
for name, fn in (("predict (dense matrix to the host)", lambda: m.predict(X)),
                 ("predict_topk (k=10, ranking on the device)", lambda: m.predict_topk(X, k=10))):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:46s} {DOCS / dt:9.0f} docs/s", flush=True)
p = m.predict(X[:300])
assert p.shape == (300, N) and np.isfinite(p).all() and 0 <= p.min() and p.max() <= 1
