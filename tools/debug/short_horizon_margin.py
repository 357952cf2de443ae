import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "aae-recommender_amd"))
import numpy as np, torch
import test_host_gpu as T
import aaerec.aae
z, Xtr, Xin, Yout = T._big()
seed = int(z["short_seed"])
if os.environ.get("NO_PREFETCH"):
    from aaerec import _hip
    _hip.HipAAE.prefetch = lambda self, *a, **k: None
for rep in range(int(os.environ.get("REPS", 4))):
    torch.manual_seed(seed); np.random.seed(seed)
    m = T._big_model(3, "reference", deterministic=bool(os.environ.get("DET")))
    m.fit(Xtr)
    n = z["pred_short"].shape[0]
    pred = m.predict(Xin[:n])
    d = np.abs(pred - z["pred_short"])
    print("rep", rep, "max diff", d.max(), "n > 5e-5:", int((d > 5e-5).sum()), "n > 8e-5:", int((d > 8e-5).sum()), flush=True)
