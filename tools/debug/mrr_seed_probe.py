#!/usr/bin/env python3
"""Debug: MRR@10 of the e2e_c1_big recipe for chosen (host seed, rng mode / device seed) pairs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aaerec.aae
import test_host_gpu as T
z, Xtr, Xin, Yout = T._big()
pairs = [(15, "device", dict(seed=d)) for d in (1014, 1016, 2015, 3015, 15)] + [(h, "device", dict(seed=1015)) for h in (0, 1, 2, 4, 5)]
if len(sys.argv) > 1:
    pairs = [(int(a.split(":")[0]), "device", dict(seed=int(a.split(":")[1]))) for a in sys.argv[1:]]
for host, mode, kw in pairs:
    torch.manual_seed(host); np.random.seed(host)
    m = T._big_model(120, mode, **kw)
    m.fit(Xtr)
    print(host, mode, kw, "MRR@10 %.4f" % T._mrr10(m.predict(Xin), Xin, Yout), "losses", [round(x, 4) for x in m.last_losses], flush=True)
