#!/usr/bin/env python3
"""Host-level fit() throughput (docs/s) of the conditioned model variants at the headline widths:
no condition, a constant concatenated block (pretrained-embedding style), a trainable categorical condition
(embedding on CPU = the reference's default, and on the GPU).  Everything runs through AdversarialAutoEncoder.fit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec.aae import AdversarialAutoEncoder
from aaerec import condition as C
from tools.synth import throughput_corpus

N, h, c, B = 100000, 200, 50, 100
DOCS = 6400
X = throughput_corpus(DOCS, N, seed=1234)
rng = np.random.default_rng(0)


class ConstConcat(C.ConcatenationBasedConditioning):
    constant_concat = True

    def __init__(self, width):
        self.width = width

    def fit(self, raw):
        return self

    def transform(self, raw):
        return raw

    def encode(self, inputs):
        return torch.as_tensor(np.asarray(inputs), dtype=torch.float32)

    def size_increment(self):
        return self.width


def rate(tag, conditions, data, epochs=25):
    m = AdversarialAutoEncoder(n_hidden=h, n_code=c, batch_size=B, n_epochs=1, conditions=conditions, verbose=False, seed=1)
    m.fit(X, condition_data=data)          # builds + warms
    torch.cuda.synchronize()
    m2 = m
    t0 = time.perf_counter()
    # fit() rebuilds the model; time whole calls (what a user sees), then subtract the build measured separately
    m2.n_epochs = epochs
    m2.fit(X, condition_data=data)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m2.n_epochs = 0
    m2.fit(X, condition_data=data)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dt = (t1 - t0) - (t2 - t1)
    print(f"{tag:34s} {epochs * DOCS / dt:10.0f} docs/s   ({1e3 * dt / (epochs * DOCS / B):.3f} ms/step, build {t2 - t1:.2f} s)", flush=True)


rate("no condition", None, None)
vec = rng.standard_normal((DOCS, 50)).astype(np.float32)
rate("constant concat (50)", C.ConditionList([("title", ConstConcat(50))]), [vec])
authors = [[int(a) for a in rng.integers(0, 5000, rng.integers(1, 5))] for _ in range(DOCS)]
for on_gpu in (False, True):
    cat = C.CategoricalCondition(32, use_cuda=True, embedding_on_gpu=on_gpu, reduce="sum")
    cl = C.ConditionList([("authors", cat)])
    data = cl.fit_transform([authors])
    rate(f"categorical (32, emb on {'gpu' if on_gpu else 'cpu'})", cl, data, epochs=25 if on_gpu else 2)
    both = C.ConditionList([("title", ConstConcat(50)), ("authors", C.CategoricalCondition(
        32, use_cuda=True, embedding_on_gpu=on_gpu, reduce="sum"))])
    d2 = both.fit_transform([vec, authors])
    rate(f"concat + categorical (emb on {'gpu' if on_gpu else 'cpu'})", both, d2, epochs=25 if on_gpu else 2)

# the same condition forced through the torch-autograd bridge (the step cut at the condition boundary)
cat = C.CategoricalCondition(32, use_cuda=True, reduce="sum")
cat.device_native = lambda device: False
cl = C.ConditionList([("authors", cat)])
rate("categorical (emb on gpu, autograd bridge)", cl, cl.fit_transform([authors]))
