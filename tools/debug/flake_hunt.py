#!/usr/bin/env python3
"""Hunt for the rare alternative trajectory of the short-horizon MRR test: R runs of its 120 steps, a checksum of every
small tensor after every step (device-side, no host sync inside a run); prints, for the runs whose final prediction is off,
the first step and tensor where the run leaves the majority."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R_, "tests")); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "aae-recommender_amd"))
import numpy as np, torch
import test_host_gpu as T
import aaerec.aae
from aaerec import _hip
z, Xtr, Xin, Yout = T._big()
seed = int(z["short_seed"])
TIDS = [(n, getattr(_hip, n)) for n in ("T_ENC_B1", "T_ENC_W2", "T_ENC_W3", "T_DEC_V1", "T_DEC_V2", "T_DEC_V3", "T_DISC_D1", "T_DISC_D2", "T_DISC_D3")]
ACTS = [("A1", _hip.T_ACT_A1), ("DH2", _hip.T_ACT_DH2), ("Z", _hip.T_ACT_Z)]
if os.environ.get("NO_PREFETCH"):
    _hip.HipAAE.prefetch = lambda self, *a, **k: None
runs = []
for rep in range(int(os.environ.get("REPS", 30))):
    torch.manual_seed(seed); np.random.seed(seed)
    m = T._big_model(3, "reference", deterministic=bool(os.environ.get("DET")))
    sums = []
    for step in m.fit_steps(Xtr):
        h = m.hip
        vals = [h.tensor(t, padded=True).double().abs().sum() for _, t in TIDS] + [h.tensor(t, padded=True)[:100].double().abs().sum() for _, t in ACTS]
        sums.append(torch.stack(vals))
    n = z["pred_short"].shape[0]
    pred = m.predict(Xin[:n])
    d = float(np.abs(pred - z["pred_short"]).max())
    runs.append((d, torch.stack(sums).cpu().numpy()))
    print("rep", rep, "max diff", d, flush=True)
names = [n for n, _ in TIDS] + [n for n, _ in ACTS]
ref = np.median(np.stack([r[1] for r in runs]), axis=0)
for i, (d, s) in enumerate(runs):
    rel = np.abs(s - ref) / (np.abs(ref) + 1e-30)
    bad = np.argwhere(rel > 1e-6)
    if d > 3e-5 or len(bad):
        first = bad[0] if len(bad) else None
        print(f"run {i}: final diff {d:.3e}; first departure (> 1e-6 rel) at step {None if first is None else int(first[0])} in "
              f"{None if first is None else names[int(first[1])]}; departures per tensor at that step: "
              f"{ {} if first is None else {names[j]: float(rel[first[0], j]) for j in range(len(names)) if rel[first[0], j] > 1e-7} }")
