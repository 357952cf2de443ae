#!/usr/bin/env python3
"""MRR@10 of the 120-epoch recipe with the reference's draws replayed, deterministic=True vs the production scatter:
is the outcome of a (fragile) host seed reproducible?  python tools/debug/mrr_replay_det.py [seeds...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aae-recommender_amd"))
import numpy as np, torch
import test_host_gpu as T
import aaerec.aae
z, Xtr, Xin, Yout = T._big()
seeds = [int(a) for a in sys.argv[1:]] or [0, 3, 10]
for det in (True, False):
    for s in seeds:
        vals = []
        for rep in range(int(os.environ.get("REPS", 3))):
            torch.manual_seed(s); np.random.seed(s)
            m = T._big_model(120, "reference", deterministic=det)
            m.fit(Xtr)
            vals.append(round(float(T._mrr10(m.predict(Xin), Xin, Yout)), 4))
        print("deterministic" if det else "atomics", "seed", s, vals, "reference", round(float(z["ref_mrr10"][s]), 4), flush=True)
