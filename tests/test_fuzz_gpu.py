"""Randomised differential test: the HIP step against the CPU oracle (itself pinned to the reference's fixtures) over
random model shapes and options - widths that are not multiples of the kernels' 16 / 32 / 64 blockings, every
activation / prior / optimiser, a constant condition block, ragged batches with empty rows, both step forms (fused
aae_step and the cut at the output layer).  Small sizes: the oracle is NumPy."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("AAE_FUZZ_SEEDS", "28"))       # (a wider hunt: AAE_FUZZ_SEEDS=1000 pytest tests/test_fuzz_gpu.py)
# configurations the wide hunts turned up, kept as regressions:
#   228  max_batch 113 with a 103-row last batch - the short batch qualifies for the fused output-layer kernel on a
#        model whose arena was sized for the three-kernel path only (out-of-bounds slab writes before the fix)
#   347  a decoder pre-activation 2e-8 from the LeakyReLU kink (see the tolerance note below)
SEEDS = sorted(set(range(N_SEEDS)) | {228, 347})

ACTS = ["ReLU", "SELU", "Tanh", "Sigmoid", "ELU", "LeakyReLU"]


def _config(seed):
    r = np.random.default_rng(1000 + seed)
    cfg = dict(N=int(r.integers(17, 2500)), h=int(r.integers(3, 208)), c=int(r.integers(2, 100)),
               B=int(r.integers(1, 105)), inc=int(r.choice([0, 0, 5, 30, 64])), act=str(r.choice(ACTS)),
               prior=str(r.choice(["gauss", "gauss", "categorical", "bernoulli"])),
               opt=str(r.choice(["adam", "adam", "sgd"])), drop=bool(r.integers(0, 2)), norm=bool(r.integers(0, 4)),
               scale=float(r.choice([0.0, 0.0, 2.0])), cut=bool(r.integers(0, 2)))
    cfg["inc"] = min(cfg["inc"], 206 - cfg["c"])
    if r.random() < 0.2:                      # batches past the fused output-layer kernel's 104 rows: the three-kernel path
        cfg["B"] = int(r.integers(105, 260))
        cfg["N"] = min(cfg["N"], 1200)
    return cfg, r


@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration_matches_oracle(seed):
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    cfg, r = _config(seed)
    N, h, c, B, inc = cfg["N"], cfg["h"], cfg["c"], cfg["B"], cfg["inc"]
    params = init_params(N, h, c, cond_inc=inc, seed=seed)
    p = (0.2, 0.3) if cfg["drop"] else (0.0, 0.0)
    lr = (0.05, 0.02) if cfg["opt"] == "sgd" else (2e-3, 1e-3)
    kw = dict(gen_lr=lr[0], reg_lr=lr[1], dropout=p, activation=cfg["act"], prior=cfg["prior"], optimizer=cfg["opt"],
              normalize_inputs=cfg["norm"], prior_scale=cfg["scale"] or None)
    dev = HipAAE(N, h, c, cond_inc=inc, max_batch=B, rng_mode="inject", **kw)
    dev.load_params(params)
    ora = O.OracleAAE(params, conditions=[O.ConcatConst(inc)] if inc else [], **kw)
    for s in range(3):
        Bs = B if s < 2 else max(1, B - int(r.integers(0, min(B, 17))))          # a shorter last batch
        rows = [np.sort(r.choice(N, size=int(r.integers(0 if B > 2 else 1, min(N, 12))), replace=False)) for _ in range(Bs)]
        if not any(len(x) for x in rows):
            rows[0] = np.array([int(r.integers(0, N))])
        ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
        idx = np.concatenate(rows).astype(np.int32)
        val = np.ones(len(idx), dtype=np.float32)
        masks = None
        if cfg["drop"]:
            masks = [(r.random((Bs, h)) > (p[j % 2])).astype(np.uint8) for j in range(12)]
        if cfg["prior"] == "gauss":
            zr = r.standard_normal((Bs, c)).astype(np.float32)
        elif cfg["prior"] == "categorical":
            zr = np.eye(c, dtype=np.float32)[r.integers(0, c, size=Bs)]
        else:
            zr = np.zeros((Bs, c), dtype=np.float32)             # the reference's randint(0, 1) bernoulli prior
        cond = (r.standard_normal((Bs, inc)) * 0.4).astype(np.float32) if inc else None
        csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
        cdev = torch.as_tensor(cond, device=dev.device) if inc else None
        if cfg["cut"] and h + 1 <= 208:
            dev.ae_forward(csr, 0, Bs, cond=cdev, masks=masks, z_real=zr)
            dev.output_layer_step()
            dev.ae_backward()
            dev.disc_gen()
        else:
            dev.step(csr, 0, Bs, cond=cdev, masks=masks, z_real=zr)
        want = ora.partial_fit(ip, idx, val, zr, masks, [cond] if inc else None)
        np.testing.assert_allclose(dev.losses(), want, rtol=5e-5, atol=2e-6, err_msg=f"{cfg} step {s}")
    got = dev.state_dict()
    # Parameters: 5e-5 absolute.  Two fp32 effects can move single elements further without anything being wrong, and
    # a random search over hundreds of configurations does find them (tools/debug/fuzz_case.py prints both):
    # a pre-activation within ~1e-7 of an activation kink (ReLU / LeakyReLU / SELU at 0) takes the other branch under a
    # different summation order, and Adam turns a gradient of the order of its eps (1e-8) into a step of the order of
    # the learning rate.  So: at most 1 % of a tensor's elements (8 for the small ones) may exceed the tolerance, none by
    # more than 3 lr.
    for k, w in ora.p.items():
        d = np.abs(got[k] - w)
        assert (d > 5e-5).sum() <= max(8, 0.01 * d.size) and d.max() <= 3 * max(lr), \
            f"{cfg} {k}: {(d > 5e-5).sum()} of {d.size} off, max {d.max():.2e}"
