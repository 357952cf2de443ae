"""C4's fit() loop: ms/step by corpus length (epoch boundaries drain the queue) and the host's own time per step."""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec.aae import AdversarialAutoEncoder
from aaerec.condition import ConditionList, PretrainedWordEmbeddingCondition
from tools.synth import throughput_corpus
from bench import _ConstVectors
dev = torch.device("cuda", 0)
for nb in (16, 64):
    cl = ConditionList([("title", PretrainedWordEmbeddingCondition(_ConstVectors(300), use_cuda=True))])
    cd = [torch.randn(nb * 1000, 300, device=dev) * 0.1]
    m = AdversarialAutoEncoder(n_hidden=200, n_code=50, batch_size=1000, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1, conditions=cl)
    X = throughput_corpus(nb * 1000, 4587, median_len=20, seed=3456)
    with contextlib.redirect_stdout(sys.stderr):
        it = m.fit_steps(X, condition_data=cd)
        next(it)
    for _ in range(20): next(it)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): next(it)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"batches/epoch {nb}: {dt * 10:.4f} ms/step (host enqueue done after {th * 10:.4f} ms/step)", flush=True)
    del m, it
