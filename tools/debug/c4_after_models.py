"""Why does the 4th model of a process run slower (bench.py extra.c4 after b512 and c2_bf16)?  C4's fit() loop after
N small throw-away models were created and destroyed / after N extra HIP streams were created."""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec.aae import AdversarialAutoEncoder
from aaerec._hip import HipAAE
from aaerec.condition import ConditionList, PretrainedWordEmbeddingCondition
from tools.synth import throughput_corpus
from bench import _ConstVectors
dev = torch.device("cuda", 0)
mode, n = sys.argv[1], int(sys.argv[2])
keep = []
for i in range(n):
    if mode == "models":
        keep.append(HipAAE(2000, 64, 16, max_batch=50))
    elif mode == "models_del":
        HipAAE(2000, 64, 16, max_batch=50).close()
    elif mode == "streams":
        keep.append(torch.cuda.Stream(dev))
    elif mode == "lowprio":
        keep.append(torch.cuda.Stream(dev, priority=0))
nb = 16
cl = ConditionList([("title", PretrainedWordEmbeddingCondition(_ConstVectors(300), use_cuda=True))])
cd = [torch.randn(nb * 1000, 300, device=dev) * 0.1]
m = AdversarialAutoEncoder(n_hidden=200, n_code=50, batch_size=1000, n_epochs=1 << 30, verbose=False, rng_mode="device", seed=1, conditions=cl)
X = throughput_corpus(nb * 1000, 4587, median_len=20, seed=3456)
with contextlib.redirect_stdout(sys.stderr):
    it = m.fit_steps(X, condition_data=cd)
    next(it)
for _ in range(20): next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): next(it)
torch.cuda.synchronize()
print(f"{mode} x {n}: {(time.perf_counter() - t0) * 5:.4f} ms/step", flush=True)
