"""Debug: where a timed region of K steps (barrier + synchronize on both sides, bench.py's bracket) spends its time beyond K x the
steady-state step: device event stamps behind every step of three regions + the host's view.  python tools/debug/region_profile.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import contextlib
import torch
from aaerec.aae import AdversarialAutoEncoder
from tools.synth import throughput_corpus
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(64 * B, N, median_len=20, seed=1234)
m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1)
with contextlib.redirect_stdout(sys.stderr):
    it = m.fit_steps(X)
    next(it)
for _ in range(30):
    next(it)
for rep in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    th = []
    for k in range(K):
        next(it)
        ev[k + 1].record()
        th.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(K)]
    print(f"region {rep}: host {1e6 * (t2 - t0):.0f} us for {K} steps = {1e3 * (t2 - t0) / K:.4f} ms/step; enqueue done after {1e6 * (t1 - t0):.0f} us; "
          f"device us per step (main stream): {' '.join(f'{p:.0f}' for p in per)}; host enqueue done at us: {' '.join(f'{1e6 * x:.0f}' for x in th[:6])} ...")
