import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR
from tools.synth import init_params
from test_rank_gpu import _corpus, _host_topk
N, h, c, R = int(os.environ.get("N", 5000)), 200, 50, 100
for rows, excl, train in ((37, False, False), (37, True, False), (100, True, False), (112, True, False), (113, True, False), (300, True, False), (300, True, True)):
    r = np.random.default_rng(1)
    dev = HipAAE(N, h, c, max_batch=R, rng_mode="device", seed=3, dropout=(0.2, 0.2))
    params = init_params(N, h, c, seed=0)
    params["dec.lin3.weight"] = params["dec.lin3.weight"] * 8.0
    dev.load_params(params)
    ip, idx, val, docs = _corpus(r, N, rows, 30)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    if train:
        for s in range(3):
            dev.step(csr, s * 100, 100)
    ids, vals = dev.predict_topk(csr, 0, rows, 10, exclude_known=excl)
    torch.cuda.synchronize()
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    full, dh2s = [], []
    for s in range(0, rows, R):
        n = min(R, rows - s)
        full.append(dev.predict(csr, s, n).cpu().numpy())
        dh2s.append(dev.dh2_rows(n).clone().cpu().numpy())
    full, dh2 = np.concatenate(full), np.concatenate(dh2s)
    wi, wv = _host_topk(full, docs, 10, excl)
    print(f"rows {rows} excl {excl} train {train}: vals diff {np.abs(vals - wv).max():.2e}  ids differ {(ids != wi).sum()}", flush=True)
    if np.abs(vals - wv).max() > 1e-4:
        b = int(np.argmax(np.abs(vals - wv).max(1)))
        print("  row", b, "got", ids[b], vals[b], "\n  want", wi[b], wv[b])
        lo, hi = full[b].min(), full[b].max()
        print("  scaled score of got ids in dense:", (full[b][ids[b]] - lo) / (hi - lo), " known?", [int(i) in set(docs[b].tolist()) for i in ids[b]])
