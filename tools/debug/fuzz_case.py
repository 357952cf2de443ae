"""debug: one fuzz configuration step by step; for the elements that disagree print both sides' parameter movement"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_fuzz_gpu as T
from aaerec._hip import HipAAE, DeviceCSR
from oracle import aae_oracle as O
from oracle.dense_torch_port import init_params
seed = int(sys.argv[1])
cfg, r = T._config(seed)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    cfg[k] = type(cfg[k])(v) if not isinstance(cfg[k], bool) else v == "1"
print(cfg)
N, h, c, B, inc = cfg["N"], cfg["h"], cfg["c"], cfg["B"], cfg["inc"]
params = init_params(N, h, c, cond_inc=inc, seed=seed)
p = (0.2, 0.3) if cfg["drop"] else (0.0, 0.0)
lr = (0.05, 0.02) if cfg["opt"] == "sgd" else (2e-3, 1e-3)
kw = dict(gen_lr=lr[0], reg_lr=lr[1], dropout=p, activation=cfg["act"], prior=cfg["prior"], optimizer=cfg["opt"],
          normalize_inputs=cfg["norm"], prior_scale=cfg["scale"] or None)
dev = HipAAE(N, h, c, cond_inc=inc, max_batch=B, rng_mode="inject", **kw); dev.load_params(params)
ora = O.OracleAAE(params, conditions=[O.ConcatConst(inc)] if inc else [], **kw)
prev = {k: v.copy() for k, v in ora.p.items()}
_orig = ora._mlp_fwd
def _rec(net, x0, masks, first_pre=None):
    out, cache = _orig(net, x0, masks, first_pre)
    for name in ("u1", "u2"):
        u = np.abs(cache[name])
        j = np.unravel_index(np.argmin(u), u.shape)
        if u[j] < 1e-5:
            print(f"      oracle: {net}.{name}[{j}] = {cache[name][j]:+.3e}  (activation boundary)")
    return out, cache
ora._mlp_fwd = _rec
for s in range(3):
    Bs = B if s < 2 else max(1, B - int(r.integers(0, min(B, 17))))
    rows = [np.sort(r.choice(N, size=int(r.integers(0 if B > 2 else 1, min(N, 12))), replace=False)) for _ in range(Bs)]
    if not any(len(x) for x in rows): rows[0] = np.array([int(r.integers(0, N))])
    ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    idx = np.concatenate(rows).astype(np.int32); val = np.ones(len(idx), dtype=np.float32)
    masks = [(r.random((Bs, h)) > (p[j % 2])).astype(np.uint8) for j in range(12)] if cfg["drop"] else None
    zr = r.standard_normal((Bs, c)).astype(np.float32) if cfg["prior"] == "gauss" else (np.eye(c, dtype=np.float32)[r.integers(0, c, size=Bs)] if cfg["prior"] == "categorical" else np.zeros((Bs, c), dtype=np.float32))
    cond = (r.standard_normal((Bs, inc)) * 0.4).astype(np.float32) if inc else None
    csr_ = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    cdev_ = torch.as_tensor(cond, device=dev.device) if inc else None
    if cfg["cut"]:
        for name, fn in (("ae_forward", lambda: dev.ae_forward(csr_, 0, Bs, cond=cdev_, masks=masks, z_real=zr)),
                         ("output_layer_step", dev.output_layer_step), ("ae_backward", dev.ae_backward), ("disc_gen", dev.disc_gen)):
            fn(); torch.cuda.synchronize(); print("   ok", name, flush=True)
    else:
        dev.step(csr_, 0, Bs, cond=cdev_, masks=masks, z_real=zr)
    want = ora.partial_fit(ip, idx, val, zr, masks, [cond] if inc else None)
    print(f"step {s}: losses dev {dev.losses()} oracle {want}")
    got = dev.state_dict()
    for k, w in ora.p.items():
        d = np.abs(got[k] - w)
        if d.max() > 2e-5:
            bad = np.argwhere(d > 2e-5)
            print(f"   {k}: {len(bad)} off, max {d.max():.2e}; first: idx {bad[0]}, oracle moved {w[tuple(bad[0])] - prev[k][tuple(bad[0])]:+.3e}, device moved {got[k][tuple(bad[0])] - prev[k][tuple(bad[0])]:+.3e}")
            if k == "enc.lin1.weight":
                items = sorted(set(int(b[1]) for b in bad)); print("      items:", items[:10], "in batch:", [int(i in set(idx.tolist())) for i in items[:10]])
    prev = {k: v.copy() for k, v in ora.p.items()}
