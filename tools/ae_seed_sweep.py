import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "aae-recommender_amd", ""): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
from test_host_gpu import _e2e
from aaerec.aae import AAERecommender
from aaerec.evaluation import remove_non_missing, METRICS
z, Xtr, Xin, Yout = _e2e()
class Set:
    def __init__(self, X): self.X = X
    def tocsr(self): return self.X
for ep in (100, 200, 400):
    out = []
    for seed in range(6):
        torch.manual_seed(seed); np.random.seed(seed)
        rec = AAERecommender(adversarial=False, n_hidden=50, n_code=50, n_epochs=ep, batch_size=100, lr=0.01, verbose=False)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            rec.train(Set(Xtr))
            pred = remove_non_missing(rec.predict(Set(Xin)), Xin, copy=True)
        out.append(round(float(METRICS["mrr@10"](Yout.toarray(), pred)[0]), 4))
    print(ep, out, flush=True)
