#!/usr/bin/env python3
"""Run-to-run spread of the three 3-epoch end-to-end parity tests' max |pred - reference| (bound 1e-4)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aae-recommender_amd"))
import numpy as np, torch
import test_host_gpu as T
from aaerec.aae import AdversarialAutoEncoder, AutoEncoder
from aaerec.dae import DenoisingAutoEncoder
z, Xtr, Xin, _ = T._e2e()
reps = int(os.environ.get("REPS", 20))
def spread(tag, make, seed, want):
    out = []
    for _ in range(reps):
        torch.manual_seed(seed); np.random.seed(seed)
        m = make(); m.fit(Xtr)
        out.append(float(np.abs(m.predict(Xin[:40]) - want).max()))
    v = np.asarray(out)
    print(tag, "max diff: min %.3e median %.3e max %.3e" % (v.min(), np.median(v), v.max()), flush=True)
G = T.GOLDEN
for det in (False, True):
    spread(f"AAE det={det}", lambda: AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, gen_lr=0.01, reg_lr=0.001, verbose=False, rng_mode="reference", deterministic=det), int(z["short_seed"]), z["pred_short"])
spread("AutoEncoder", lambda: AutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, lr=0.01, verbose=False, rng_mode="reference"), 7, np.load(os.path.join(G, "e2e_ae_short.npz"))["pred_short"])
spread("DAE", lambda: DenoisingAutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, lr=0.01, verbose=False, rng_mode="reference"), 7, np.load(os.path.join(G, "e2e_dae_short.npz"))["pred_short"])
