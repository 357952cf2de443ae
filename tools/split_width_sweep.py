#!/usr/bin/env python3
"""The deferred output-layer launch's width (aae_set_split) against the step time, inside the fit() loop at bench.py's C3 shape:
python tools/split_width_sweep.py [widths ...]   (N / B / STEPS from the environment).  The library picks the width by shape
(aae_create); this is the sweep behind that choice, on the current build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import contextlib
import numpy as np
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus

N, h, c, B = int(os.environ.get("N", 100000)), int(os.environ.get("H", 200)), 50, int(os.environ.get("B", 100))
steps = int(os.environ.get("STEPS", 400))
widths = [int(x) for x in sys.argv[1:]] or [0, 96, 104, 112, 120, 128, 136, 144, 160]
X = throughput_corpus(64 * B, N, median_len=20, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1)
with contextlib.redirect_stdout(sys.stderr):
    it = m.fit_steps(X)
    next(it)
for _ in range(50):
    next(it)
torch.cuda.synchronize()
for rep in range(2):
    for w in widths:
        m.hip.sync()
        if w:
            m.hip.set_split(w)
        for _ in range(30):
            next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            next(it)
        torch.cuda.synchronize()
        print(f"pass {rep}: deferred launch on {w if w else 'the default number of'} workgroups: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step", flush=True)
