#!/usr/bin/env python3
"""cProfile of the host side of one training step (enqueue only): where the Python / ctypes / launch time goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec.aae import AdversarialAutoEncoder
from aaerec import condition as C
from tools.synth import throughput_corpus

N, h, c, B, DOCS = 100000, 200, 50, 100, 6400
X = throughput_corpus(DOCS, N, seed=1234)
rng = np.random.default_rng(0)
authors = [[int(a) for a in rng.integers(1, 5000, rng.integers(1, 5))] for _ in range(DOCS)]
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
conds = data = None
if mode == "cat":
    cat = C.CategoricalCondition(32, use_cuda=True, reduce="sum")
    conds = C.ConditionList([("authors", cat)])
    data = conds.fit_transform([authors])
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=2, conditions=conds, verbose=False, seed=1)
m.fit(X, condition_data=data)
torch.cuda.synchronize()
m.n_epochs = 10
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m.fit(X, condition_data=data)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"mode {mode}: host {1e3 * (t1 - t0) / 640:.3f} ms/step (profiled), GPU tail {1e3 * (t2 - t1):.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
