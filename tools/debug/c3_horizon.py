"""Debug: replay tests/golden/tmp_e2e_c3_<epochs>.npz (tools/gen_golden.py e2e_c3:<epochs>) and print how far the probe
scores are from the reference's after each horizon."""
import os, sys
import numpy as np
import scipy.sparse as sp
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import aaerec.aae  # noqa
from aaerec.aae import AdversarialAutoEncoder

for ep in sys.argv[1:]:
    z = np.load(os.path.join(ROOT, "tests", "golden", f"tmp_e2e_c3_{ep}.npz"))
    N, seed = int(z["N"]), int(z["seed"])
    def csr(indptr, indices):
        return sp.csr_matrix((np.ones(len(indices), dtype=np.float32), indices, indptr), shape=(len(indptr) - 1, N))
    Xtr, Xin = csr(z["train_indptr"], z["train_indices"]), csr(z["in_indptr"], z["in_indices"])
    import json
    for lr in (json.loads(str(z["recipe"]))["gen_lr"],):
        torch.manual_seed(seed); np.random.seed(seed)
        m = AdversarialAutoEncoder(n_hidden=200, n_code=50, n_epochs=int(z["n_epochs"]), batch_size=100, gen_lr=lr, reg_lr=0.001,
                                   dropout=(0., 0.), verbose=False, rng_mode="reference")
        m.fit(Xtr)
        pred = m.predict(Xin)
        got = np.take_along_axis(pred, z["probe"].astype(np.int64), axis=1)
        want = z["probe_raw"]
        d = np.abs(got - want)
        print(f"epochs {ep}: max |diff| {d.max():.3e}  max rel {np.max(d / np.maximum(want, 1e-9)):.3e}  ref max {want.max():.3e} "
              f"row_max diff {np.abs(pred.max(1) - z['row_max']).max():.3e}", flush=True)
