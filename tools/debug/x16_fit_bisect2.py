#!/usr/bin/env python3
"""Which ingredient of fit() makes chain16x3 (forced) part from chain4 on the e2e_c3 fixture's data?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, scipy.sparse as sp, torch
import aaerec.aae
from aaerec import _hip
from aaerec.aae import AdversarialAutoEncoder
z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_c3.npz"))
N, seed = int(z["N"]), int(z["seed"])
Xtr = sp.csr_matrix((np.ones(len(z["train_indices"]), dtype=np.float32), z["train_indices"], z["train_indptr"]), shape=(len(z["train_indptr"]) - 1, N))
def run(forced, variant, steps=4):
    if forced: os.environ["AAE_X16_ROWS"] = "1"; os.environ.pop("AAE_NO_X16", None)
    else: os.environ["AAE_NO_X16"] = "1"; os.environ.pop("AAE_X16_ROWS", None)
    torch.manual_seed(seed); np.random.seed(seed)
    m = AdversarialAutoEncoder(n_hidden=200, n_code=50, n_epochs=3, batch_size=100, dropout=(0., 0.), verbose=False, rng_mode="reference")
    m._build(N, 0, max_row_nnz=4096)
    csr = _hip.DeviceCSR(Xtr, m.hip.device)
    perm = np.random.permutation(Xtr.shape[0]).astype(np.int32)
    perm_dev = torch.as_tensor(perm, device=m.hip.device)
    out = []
    for s in range(steps):
        zr = torch.randn(100, 50)
        rows = _hip.row_ids(perm_dev[s * 100:(s + 1) * 100], 7) if "rows" in variant else None
        if "prefetch" in variant and s + 1 < steps:
            nr = _hip.row_ids(perm_dev[(s + 1) * 100:(s + 2) * 100], 7) if "rows" in variant else None
            m.hip.prefetch(csr, 0 if nr is not None else (s + 1) * 100, 100, nr)
        m.hip.step(csr, 0 if rows is not None else s * 100, 100, rows=rows, z_real=zr)
        out.append({k: v.copy() for k, v in m.hip.state_dict().items()} if "nosd" not in variant or s == steps - 1 else None)
    return out
for variant in ("plain", "rows", "prefetch", "rows+prefetch", "rows+prefetch+nosd"):
    a, b = run(True, variant), run(False, variant)
    line = []
    for s in range(len(a)):
        if a[s] is None: continue
        w = max(float(np.abs(a[s][k] - b[s][k]).max() / (np.abs(b[s][k]).max() + 1e-30)) for k in a[s])
        line.append(f"step {s}: {w:.1e}")
    print(f"{variant:20s}", "  ".join(line), flush=True)

print("---- activations after each step, 'rows' variant")
def run2(forced, steps=3):
    if forced: os.environ["AAE_X16_ROWS"] = "1"; os.environ.pop("AAE_NO_X16", None)
    else: os.environ["AAE_NO_X16"] = "1"; os.environ.pop("AAE_X16_ROWS", None)
    torch.manual_seed(seed); np.random.seed(seed)
    m = AdversarialAutoEncoder(n_hidden=200, n_code=50, n_epochs=3, batch_size=100, dropout=(0., 0.), verbose=False, rng_mode="reference")
    m._build(N, 0, max_row_nnz=4096)
    csr = _hip.DeviceCSR(Xtr, m.hip.device)
    perm = np.random.permutation(Xtr.shape[0]).astype(np.int32)
    perm_dev = torch.as_tensor(perm, device=m.hip.device)
    out = []
    for s in range(steps):
        zr = torch.randn(100, 50)
        m.hip.step(csr, 0, 100, rows=_hip.row_ids(perm_dev[s * 100:(s + 1) * 100], 7), z_real=zr)
        acts = {n: m.hip.tensor(t).clone().cpu().numpy() for n, t in (("z", _hip.T_ACT_Z), ("a1", _hip.T_ACT_A1), ("dh2", _hip.T_ACT_DH2), ("da2", _hip.T_ACT_DA2), ("ga1", _hip.T_ACT_GA1))}
        out.append((m.hip.losses(), acts))
    return out
a, b = run2(True), run2(False)
for s in range(len(a)):
    print("step", s, "losses", a[s][0], b[s][0])
    for n in a[s][1]:
        x, y = a[s][1][n][:100], b[s][1][n][:100]
        d = np.abs(x - y)
        r, cidx = np.unravel_index(d.argmax(), d.shape)
        print(f"   {n}: max |diff| {d.max():.2e} at row {r} col {cidx} (max |value| {np.abs(y).max():.2e}); rows with diff > 1e-5 * max: {np.unique(np.nonzero(d > 1e-5 * np.abs(y).max())[0])[:20].tolist()}")
