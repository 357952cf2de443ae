#!/usr/bin/env python3
"""Debug: replay the 3-epoch run of tests/golden/e2e_c1_big.npz step by step on the device model and on the CPU oracle with
the same draws; print where the losses part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, torch
from oracle import aae_oracle as O
from aaerec._hip import HipAAE, DeviceCSR
z = np.load(os.path.join(ROOT, "tests/golden/e2e_c1_big.npz")); N = int(z["N"])
def csr(p):
    ip, idx = z[p + "_indptr"], z[p + "_indices"]
    return sp.csr_matrix((np.ones(len(idx), dtype=np.float32), idx, ip), shape=(len(ip) - 1, N))
Xtr, Xin = csr("train"), csr("in")
torch.manual_seed(7); np.random.seed(7)
params = {}
for net, (i, h, o) in (("enc", (N, 50, 50)), ("dec", (50, 50, N)), ("disc", (50, 50, 1))):
    for layer, (a, b) in enumerate(((i, h), (h, h), (h, o)), start=1):
        lin = torch.nn.Linear(a, b)
        params[f"{net}.lin{layer}.weight"] = lin.weight.detach().numpy(); params[f"{net}.lin{layer}.bias"] = lin.bias.detach().numpy()
kw = dict(gen_lr=0.01, reg_lr=0.001, dropout=(0., 0.))
ora = O.OracleAAE(params, **kw)
dev = HipAAE(N, 50, 50, max_batch=100, rng_mode="inject", **kw); dev.load_params(params)
dcsr = DeviceCSR(Xtr, dev.device)
n = Xtr.shape[0]; step = 0
for ep in range(3):
    perm = np.arange(n); np.random.shuffle(perm)
    pd = torch.as_tensor(perm.astype(np.int32), device=dev.device)
    for s in range(0, n, 100):
        Xb = Xtr[perm[s:s + 100]]
        zr = torch.randn((Xb.shape[0], 50)).numpy()
        lo = ora.partial_fit(Xb.indptr.astype(np.int64), Xb.indices, Xb.data, zr, None, None)
        dev.step(dcsr, 0, 100, rows=pd[s:s + 100], z_real=zr)
        ld = dev.losses()
        d = max(abs(a - b) / max(abs(b), 1e-9) for a, b in zip(ld, lo))
        if step < 5 or step % 20 == 0 or d > 1e-3:
            sd = dev.state_dict()
            pm = max(float(np.abs(sd[k] - ora.p[k]).max()) for k in ora.p)
            print(step, "rel loss diff %.2e" % d, "max param diff %.2e" % pm, [round(x, 5) for x in ld], [round(x, 5) for x in lo], flush=True)
        step += 1
Xp = Xin[:200]
pred = ora.predict(Xp.indptr.astype(np.int64), Xp.indices, Xp.data)
got = dev.predict(DeviceCSR(Xp, dev.device), 0, 100).cpu().numpy()
print("oracle vs fixture", np.abs(pred - z["pred_short"]).max(), "device vs fixture (first 100)", np.abs(got - z["pred_short"][:100]).max())
