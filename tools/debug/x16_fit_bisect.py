#!/usr/bin/env python3
"""fit() of the e2e_c3 fixture with every chain program on chain16x3 (AAE_X16_ROWS=1) vs chain4: where do they part?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, scipy.sparse as sp, torch
import aaerec.aae
from aaerec.aae import AdversarialAutoEncoder
z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_c3.npz"))
N, seed = int(z["N"]), int(z["seed"])
def csr(indptr, indices):
    return sp.csr_matrix((np.ones(len(indices), dtype=np.float32), indices, indptr), shape=(len(indptr) - 1, N))
Xtr, Xin = csr(z["train_indptr"], z["train_indices"]), csr(z["in_indptr"], z["in_indices"])
def run(forced, steps):
    if forced: os.environ["AAE_X16_ROWS"] = "1"; os.environ.pop("AAE_NO_X16", None)
    else: os.environ["AAE_NO_X16"] = "1"; os.environ.pop("AAE_X16_ROWS", None)
    torch.manual_seed(seed); np.random.seed(seed)
    m = AdversarialAutoEncoder(n_hidden=200, n_code=50, n_epochs=3, batch_size=100, dropout=(0., 0.), verbose=False, rng_mode="reference")
    it = m.fit_steps(Xtr)
    out = {}
    for s in range(steps):
        next(it)
        if s in (0, 1, 5, 19, 20, 39, 59):
            out[s] = {k: v.copy() for k, v in m.hip.state_dict().items()}
    m._fit_finish()
    return out, m.predict(Xin), m
a, pa, ma = run(True, int(os.environ.get("XSTEPS", 60)))
b, pb, mb = run(False, int(os.environ.get("XSTEPS", 60)))
for s in sorted(a):
    worst = {k: float(np.abs(a[s][k] - b[s][k]).max() / (np.abs(b[s][k]).max() + 1e-30)) for k in a[s]}
    print("step", s, ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
print("predict after fit: max |x16 - chain4|", float(np.abs(pa - pb).max()), "max", float(pb.max()))
got = np.take_along_axis(pa, z["probe"].astype(np.int64), axis=1)
print("x16 vs reference probes:", float(np.abs(got - z["probe_raw"]).max()), " chain4 vs reference:", float(np.abs(np.take_along_axis(pb, z["probe"].astype(np.int64), axis=1) - z["probe_raw"]).max()))
# the same trained weights through the OTHER kernel's predict
pa2 = mb.predict(Xin)
print("chain4 model predict again:", float(np.abs(pa2 - pb).max()))
