"""debug: 2 ranks (threads) x B=20 through DataParallelAAE vs one fused model with B=40, many steps"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, torch
from aaerec._hip import HipAAE, DeviceCSR
from aaerec.parallel import DataParallelAAE
from test_parity_abi_gpu import _ThreadDist
from tools.synth import init_params
rng = np.random.RandomState(3)
N, h, c, B, W = 500, 48, 16, 40, 2
protos = [rng.choice(N, size=10, replace=False) for _ in range(12)]
rows = [rng.choice(protos[rng.randint(12)], size=rng.randint(3, 9), replace=False) for _ in range(200)]
ind0 = [b for b, r in enumerate(rows) for _ in r]
X = sp.coo_matrix((np.ones(len(ind0), dtype=np.float32), (ind0, np.concatenate(rows))), shape=(200, N)).tocsr()
params = init_params(N, h, c, seed=0)
kw = dict(dropout=(0.0, 0.0), rng_mode="inject", ae_only=True, gen_lr=0.01, reg_lr=0.01)
a = HipAAE(N, h, c, max_batch=B, **kw); a.load_params(params)
csr = DeviceCSR(X, a.device)
STEPS = 15
snaps = []
prng = np.random.RandomState(5)
perms = []
for e in range(3):
    p = np.arange(200); prng.shuffle(p); perms.append(torch.as_tensor(p.astype(np.int32), device=a.device))
for s in range(STEPS):
    pd = perms[s // 5]
    a.step(csr, 0, B, rows=pd[(s % 5) * B:(s % 5 + 1) * B])
    snaps.append(a.state_dict())
dist = _ThreadDist(W)
models = [None] * W
log = []
def main(r):
    dist.bind(r)
    m = HipAAE(N, h, c, max_batch=B // W, grad_mode="export", dp_world=W, **kw); m.load_params(params)
    models[r] = m
    dp = DataParallelAAE(m, dist, shard_decoder=(sys.argv[1] == "shard") if len(sys.argv) > 1 else False)
    for s in range(STEPS):
        lo = (s % 5) * B + r * (B // W)
        dp.step(csr, 0, B // W, global_rows=B, rows=perms[s // 5][lo:lo + B // W])
        dp.wait_pending()
        torch.cuda.synchronize()
        dist.bar.wait()
        if r == 0:
            sb = m.state_dict()
            worst = max((float(np.abs(snaps[s][k] - sb[k]).max()), k) for k in sb)
            bad = int((np.abs(snaps[s]["enc.lin1.weight"] - sb["enc.lin1.weight"]).max(0) > 1e-5).sum())
            log.append(f"step {s}: worst {worst[0]:.2e} in {worst[1]}, W1 items off: {bad}")
        dist.bar.wait()
ts = [threading.Thread(target=main, args=(r,)) for r in range(W)]
[t.start() for t in ts]; [t.join() for t in ts]
print("\n".join(log))
