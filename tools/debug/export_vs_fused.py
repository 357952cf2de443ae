"""debug: export-mode (DataParallelAAE, one rank) vs fused step over many steps on a small recurring vocabulary"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, torch
from aaerec._hip import HipAAE, DeviceCSR
from aaerec.parallel import DataParallelAAE
from test_parity_abi_gpu import _SoloDist
from tools.synth import init_params
rng = np.random.RandomState(3)
N, h, c, B = 500, 48, 16, 40
protos = [rng.choice(N, size=10, replace=False) for _ in range(12)]
rows = [rng.choice(protos[rng.randint(12)], size=rng.randint(3, 9), replace=False) for _ in range(200)]
ind0 = [b for b, r in enumerate(rows) for _ in r]
X = sp.coo_matrix((np.ones(len(ind0), dtype=np.float32), (ind0, np.concatenate(rows))), shape=(200, N)).tocsr()
params = init_params(N, h, c, seed=0)
for ae_only in (True, False):
    kw = dict(dropout=(0.0, 0.0), rng_mode="inject", ae_only=ae_only, gen_lr=0.01, reg_lr=0.01)
    a = HipAAE(N, h, c, max_batch=B, **kw); a.load_params(params)
    b = HipAAE(N, h, c, max_batch=B, grad_mode="export", dp_world=1, **kw); b.load_params(params)
    dp = DataParallelAAE(b, _SoloDist(), shard_decoder=False)
    csr = DeviceCSR(X, a.device)
    for s in range(15):
        zr = rng.standard_normal((B, c)).astype(np.float32)
        a.step(csr, (s % 5) * B, B, z_real=zr)
        dp.step(csr, (s % 5) * B, B, global_rows=B, z_real=zr)
        dp.wait_pending()
        sa, sb = a.state_dict(), b.state_dict()
        worst = max((float(np.abs(sa[k] - sb[k]).max()), k) for k in sa)
        bad = int((np.abs(sa["enc.lin1.weight"] - sb["enc.lin1.weight"]).max(0) > 1e-5).sum())
        print(f"ae_only={ae_only} step {s}: worst {worst[0]:.2e} in {worst[1]}, W1 items off: {bad}", flush=True)
