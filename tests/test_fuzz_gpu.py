"""Randomised differential test: the HIP step against the CPU oracle (itself pinned to the reference's fixtures) over
random model shapes and options - widths that are not multiples of the kernels' 16 / 32 / 64 blockings, every
activation / prior / optimiser, a constant condition block, ragged batches with empty rows, both step forms (fused
aae_step and the cut at the output layer).  Small sizes: the oracle is NumPy."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("AAE_FUZZ_SEEDS", "28"))       # (a wider hunt: AAE_FUZZ_SEEDS=1000 pytest tests/test_fuzz_gpu.py)
# (the wide hunts found: a model with max_batch 113 whose 103-row last batch took the fused output-layer kernel without
#  room for its slabs - fixed, tests/test_parity_abi_gpu.py::test_short_batch_on_a_model_sized_for_long_ones - and the
#  single-element sensitivities described at the tolerance below)
SEEDS = list(range(N_SEEDS))

ACTS = ["ReLU", "SELU", "Tanh", "Sigmoid", "ELU", "LeakyReLU"]


def _config(seed, tiny=False):
    r = np.random.default_rng((3000 if tiny else 1000) + seed)
    dims = dict(N=int(r.integers(2, 40)), h=int(r.integers(1, 9)), c=int(r.integers(1, 7)), B=int(r.integers(1, 6)),
                inc=int(r.choice([0, 0, 1, 3]))) if tiny else \
        dict(N=int(r.integers(17, 2500)), h=int(r.integers(3, 208)), c=int(r.integers(2, 100)),
             B=int(r.integers(1, 105)), inc=int(r.choice([0, 0, 5, 30, 64])))
    cfg = dict(**dims, act=str(r.choice(ACTS)),
               prior=str(r.choice(["gauss", "gauss", "categorical", "bernoulli"])),
               opt=str(r.choice(["adam", "adam", "sgd"])), drop=bool(r.integers(0, 2)), norm=bool(r.integers(0, 4)),
               scale=float(r.choice([0.0, 0.0, 2.0])), cut=bool(r.integers(0, 2)))
    cfg["inc"] = max(0, min(cfg["inc"], 206 - cfg["c"]))
    if not tiny and r.random() < 0.2:         # batches past the fused output-layer kernel's 104 rows: the three-kernel path
        cfg["B"] = int(r.integers(105, 260))
        cfg["N"] = min(cfg["N"], 1200)
    cfg["long_rows"] = bool(r.random() < 0.15)       # documents with hundreds of items (per-row chunking, >1024 entries per tile)
    cfg["window"] = bool(r.random() < 0.3)           # the batch is a permutation window into a larger resident corpus
    if not tiny and r.random() < 0.2:         # a condition that makes the decoder's input wider than a chain slot (two k-parts)
        cfg["inc"] = int(r.integers(208 - cfg["c"], 415 - cfg["c"]))
    return cfg, r


def _close_enough(got, want, tol, dmax, tag):
    """Element-wise |got - want| <= tol, except for what a single activation kink / an Adam-eps gradient can move (see
    the note in test_random_configuration_matches_oracle): one hidden unit taking the other branch of its activation
    for one document changes that unit's whole weight-gradient ROW, so up to two rows' worth of elements (at least 8,
    at most 1 % of a large tensor) may exceed tol - none by more than dmax."""
    d = np.abs(got - want)
    allowed = max(8, 0.01 * d.size, 2 * (d.shape[-1] if d.ndim > 1 else 1))
    assert (d > tol).sum() <= allowed and d.max() <= dmax, f"{tag}: {(d > tol).sum()} of {d.size} off (allowed {allowed:.0f}), max {d.max():.2e}"


@pytest.mark.parametrize("tiny", [False, True])
@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration_matches_oracle(seed, tiny):
    cfg, r = _config(seed, tiny)           # tiny: degenerate sizes (one hidden unit, a code of one, two items, one document)
    _check_configuration(cfg, r, seed)


# r6: the fourteen further activation classes (csrc/device_common.h; the general program kernel chain_kernel<.., true> and the
# per-layer epilogues carry them): the same random configurations - widths, conditions incl. the wide ones, priors, optimisers,
# ragged batches, permutation windows, the step cut at the output layer - with the class cycling over the seeds
ACTS_R6 = ["Softplus", "Hardtanh", "ReLU6", "CELU", "Softsign", "Hardsigmoid", "LogSigmoid", "Softshrink", "Hardshrink", "Identity",
           "GELU", "SiLU", "Mish", "Hardswish"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "28"))))
def test_random_configuration_with_a_further_activation_class_matches_oracle(seed):
    cfg, r = _config(500 + seed)
    cfg["act"] = ACTS_R6[seed % len(ACTS_R6)]
    # (the step cut at the output layer - aae_ae_forward / _backward - is the layer programs'; with one of these classes a decoder
    #  input wider than a slot runs layer by layer, where the library refuses the cut form with AAE_ESTATE: the whole step there)
    if cfg["c"] + cfg["inc"] + 1 > 208:
        cfg["cut"] = False
    _check_configuration(cfg, r, 500 + seed)


# A decoder input [z | condition | 1] wider than a row of the layer-chain kernel's LDS slots (208 columns): dec.lin1 runs as
# two k-parts, its dX as two column parts (csrc/abi_chains.h: add_dec_in_fwd / add_dec_in_dx).  C4's shape (code 50 +
# 300-d condition), the narrowest wide input (209 columns: the second part is the bias column alone), the widest (416),
# a part boundary inside the condition / right behind z, a batch beyond one fused launch.
WIDE = [dict(N=300, h=40, c=50, B=37, inc=300), dict(N=120, h=200, c=10, B=100, inc=198), dict(N=500, h=64, c=100, B=64, inc=315),
        dict(N=257, h=207, c=207, B=5, inc=33), dict(N=900, h=100, c=50, B=150, inc=300), dict(N=64, h=7, c=2, B=3, inc=206),
        dict(N=200, h=30, c=300, B=9, inc=0), dict(N=150, h=20, c=210, B=6, inc=100)]      # (a code wider than a slot: layer by layer)


@pytest.mark.parametrize("cut", [False, True])
@pytest.mark.parametrize("case", range(len(WIDE)))
def test_condition_wider_than_a_chain_slot_matches_oracle(case, cut):
    r = np.random.default_rng(7000 + case)
    cfg = dict(**WIDE[case], act=ACTS[case % len(ACTS)], prior=["gauss", "categorical", "gauss"][case % 3],
               opt=["adam", "adam", "sgd"][case % 3], drop=bool(case % 2 == 0), norm=bool(case % 3), scale=0.0, cut=cut,
               long_rows=False, window=bool(case % 2), predict=True)
    _check_configuration(cfg, r, 7000 + case)


def _check_configuration(cfg, r, seed):
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
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
        max_len = int(min(N, 600)) if cfg["long_rows"] else int(min(N + 1, 12))
        n_docs = 3 * Bs if cfg["window"] else Bs
        rows = [np.sort(r.choice(N, size=int(r.integers(0 if B > 2 else 1, max_len)), replace=False)) for _ in range(n_docs)]
        if not any(len(x) for x in rows):
            rows[0] = np.array([int(r.integers(0, N))])
        ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
        idx = np.concatenate(rows).astype(np.int32)
        val = np.ones(len(idx), dtype=np.float32)
        csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
        sel = None
        if cfg["window"]:                      # rows = a window of a permutation, as fit() passes it
            pick = r.permutation(n_docs)[:Bs].astype(np.int32)
            sel = torch.as_tensor(pick, device=dev.device)
            rows = [rows[j] for j in pick]
            ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
            idx = np.concatenate(rows).astype(np.int32) if ip[-1] else np.zeros(0, dtype=np.int32)
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
        cdev = torch.as_tensor(cond, device=dev.device) if inc else None
        if cfg["cut"] and h + 1 <= 208 and c + 1 <= 208 and c + inc + 1 <= 416:     # (the cut form is the layer-chain models')
            dev.ae_forward(csr, 0, Bs, rows=sel, cond=cdev, masks=masks, z_real=zr)
            dev.output_layer_step()
            dev.ae_backward()
            dev.disc_gen()
        else:
            dev.step(csr, 0, Bs, rows=sel, cond=cdev, masks=masks, z_real=zr)
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
        _close_enough(got[k], w, 5e-5, 3 * max(lr), f"{cfg} {k}")
    if cfg.get("predict"):                     # the last batch's reconstructions with the trained weights (no dropout)
        out = dev.predict(csr, 0, Bs, cond=cdev) if sel is None else None
        if out is not None:
            want = ora.predict(ip, idx, val, [cond] if inc else None)
            np.testing.assert_allclose(out.cpu().numpy(), want, atol=1e-4, err_msg=f"{cfg} predict")


def _batch(r, N, Bs, B, max_len=12):
    rows = [np.sort(r.choice(N, size=int(r.integers(0 if B > 2 else 1, min(N, max_len))), replace=False)) for _ in range(Bs)]
    if not any(len(x) for x in rows):
        rows[0] = np.array([int(r.integers(0, N))])
    ip = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64)
    idx = np.concatenate(rows).astype(np.int32)
    return ip, idx, np.ones(len(idx), dtype=np.float32)


class _Solo:
    """torch.distributed stand-in for one rank."""
    class ReduceOp:
        SUM = 0

    def get_world_size(self, group=None):
        return 1

    def get_rank(self, group=None):
        return 0

    def get_backend(self, group=None):
        return "solo"

    def all_reduce(self, t, op=None, group=None, async_op=False):
        return None

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        out.copy_(inp.reshape(out.shape))

    def reduce_scatter_tensor(self, out, inp, op=None, group=None, async_op=False):
        out.copy_(inp.reshape(out.shape))


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "16"))))
def test_random_long_run_and_execution_paths_match_oracle(seed):
    """25-40 steps on a SMALL vocabulary, so items come and go with irregular gaps (the deferred Adam on enc.lin1
    replays the steps a row missed), through a randomly chosen execution path: the fused step, the gradient-export
    path of the two data-parallel drivers on one rank, or hidden widths beyond the layer-chain kernels (per-layer
    GEMM path); then eval-mode predict and the on-device top-k against the oracle."""
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from aaerec.parallel import DataParallelAAE, VocabParallelAAE
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    r = np.random.default_rng(5000 + seed)
    path = str(r.choice(["fused", "replicated", "vocab", "wide"]))
    if path == "vocab" and seed % 2:
        path = "vocab2"                             # enc.lin1 with the item slice as well (aae_first_layer_*)
    N = int(r.integers(40, 400))
    h = int(r.integers(208, 300)) if path == "wide" else int(r.integers(8, 120))
    c = int(r.integers(2, 40))
    B = int(r.integers(2, 60))
    steps = int(r.integers(25, 41))
    act = str(r.choice(["ReLU", "Tanh", "ELU"]))
    params = init_params(N, h, c, seed=seed)
    kw = dict(gen_lr=2e-3, reg_lr=1e-3, dropout=(0.0, 0.0), activation=act)
    ora = O.OracleAAE(params, **kw)
    if path in ("fused", "wide"):
        dev = HipAAE(N, h, c, max_batch=B, rng_mode="inject", **kw)
        dev.load_params(params)
        run = lambda csr, Bs, zr: dev.step(csr, 0, Bs, z_real=zr)                                   # noqa: E731
        losses = dev.losses
    else:
        dev = HipAAE(N, h, c, max_batch=B, rng_mode="inject", grad_mode="export", dp_world=1, w1_cap=min(N, B * 12), **kw)
        dev.load_params(params)
        if path == "replicated":
            dp = DataParallelAAE(dev, _Solo(), shard_decoder=bool(r.integers(0, 2)) and "force")
            run = lambda csr, Bs, zr: (dp.step(csr, 0, Bs, global_rows=Bs, z_real=zr), dp.wait_pending())   # noqa: E731
            losses = dev.losses
        else:
            sl = HipAAE(N, h, c, max_batch=B, rng_mode="inject", **kw)
            sl.load_params(params)
            dp = VocabParallelAAE(dev, sl, _Solo(), N, shard_first_layer=path == "vocab2")
            run = lambda csr, Bs, zr: dp.step(csr, 0, Bs, csr, 0, Bs, z_real=zr)                   # noqa: E731
            losses = lambda: (dp.recon_loss(),) + tuple(dev.losses()[1:])                          # noqa: E731
    for s in range(steps):
        Bs = B if r.random() < 0.8 else int(r.integers(1, B + 1))
        ip, idx, val = _batch(r, N, Bs, B, max_len=6)
        zr = r.standard_normal((Bs, c)).astype(np.float32)
        run(DeviceCSR.from_arrays(ip, idx, val, N, dev.device), Bs, zr)
        want = ora.partial_fit(ip, idx, val, zr)
        if s % 8 == 0 or s == steps - 1:
            np.testing.assert_allclose(losses(), want, rtol=2e-4, atol=5e-6, err_msg=f"{path} N={N} h={h} c={c} B={B} step {s}")
    if path in ("vocab", "vocab2"):
        dp.gather_output_layer()
    got = dev.state_dict()
    for k, w in ora.p.items():
        # (fp32 rounding differences compound over the run; single units may take an activation kink differently)
        _close_enough(got[k], w, 2e-4, 0.02, f"{path} N={N} h={h} c={c} B={B} {k}")
    # eval-mode reconstruction and the on-device ranking of it
    ip, idx, val = _batch(r, N, B, B, max_len=6)
    csr = DeviceCSR.from_arrays(ip, idx, val, N, dev.device)
    pred = dev.predict(csr, 0, B).cpu().numpy()
    want = ora.predict(ip, idx, val)
    np.testing.assert_allclose(pred, want, atol=5e-4)
    k = int(min(r.integers(1, 33), N - 6))
    excl = bool(r.integers(0, 2))
    ids, vals = dev.predict_topk(csr, 0, B, k, exclude_known=excl)
    ids, vals = ids.cpu().numpy(), vals.cpu().numpy()
    for b in range(B):
        known = set(idx[ip[b]:ip[b + 1]].tolist())
        assert len(set(ids[b].tolist())) == k and (np.diff(vals[b]) <= 1e-6).all()      # k distinct items, best first
        score = pred[b].copy()
        if excl:
            assert not (set(ids[b].tolist()) & known)                          # items of the input row are excluded
            score[list(known)] = -1.0
        kth = np.sort(score)[-k]
        assert (score[ids[b]] >= kth - 1e-4).all(), f"{path} row {b}: not the top {k} (exclude_known={excl})"


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "12"))))
def test_random_decoder_and_vae_steps_match_oracle(seed):
    """The sibling models' entry points over random shapes: aae_decoder_step (DecodingRecommender: decoder only, input
    block from the conditions, dL/d(input) returned) and aae_vae_step / aae_vae_predict (VAE)."""
    _decoder_vae_case(seed)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "14"))))
def test_random_decoder_and_vae_steps_with_a_further_activation_class_match_oracle(seed):
    """... with the activation classes of r6 (the VAE's programs and the decoder-only step on chain_kernel<.., true>)."""
    _decoder_vae_case(seed, ACTS_R6[seed % len(ACTS_R6)])


def _decoder_vae_case(seed, act_override=None):
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    r = np.random.default_rng(9000 + seed)
    N, h, B = int(r.integers(17, 1500)), int(r.integers(3, 208)), int(r.integers(1, 150))
    act = str(r.choice(["ReLU", "Tanh", "SELU", "ELU"]))
    if act_override:
        act = act_override
    if seed % 2 == 0:
        # ---- decoder only: the input block is what the condition plugins produced (width n_code here) ----
        c = int(r.integers(2, 200))
        params = init_params(N, h, c, seed=seed)
        p = (0.2, 0.2) if r.integers(0, 2) else (0.0, 0.0)
        dev = HipAAE(N, h, c, max_batch=B, rng_mode="inject", gen_lr=2e-3, reg_lr=2e-3, dropout=p, activation=act)
        dev.load_params(params)
        ora = O.OracleDecoder(params, lr=2e-3, dropout=p, activation=act, conditions=[O.ConcatConst(c)])
        for s in range(3):
            ip, idx, val = _batch(r, N, B, B)
            zin = (r.standard_normal((B, c)) * 0.5).astype(np.float32)
            masks = [(r.random((B, h)) > 0.2).astype(np.uint8) for _ in range(2)] if p[0] else None
            dz = dev.decoder_step(DeviceCSR.from_arrays(ip, idx, val, N, dev.device), 0, B,
                                  torch.as_tensor(zin, device=dev.device), masks=masks)
            want = ora.partial_fit([zin], ip, idx, val, masks)
            np.testing.assert_allclose(dev.losses()[0], want, rtol=5e-5, atol=2e-6)
            np.testing.assert_allclose(dz.cpu().numpy(), ora.last_dzin, rtol=5e-4, atol=1e-7)     # (entries are sums with cancellation, ~1e-4 in size)
        got = dev.state_dict()
        for k, w in ora.p.items():
            _close_enough(got[k], w, 5e-5, 6e-3, f"decoder N={N} h={h} c={c} B={B} {k}")
        zp = (r.standard_normal((B, c)) * 0.5).astype(np.float32)
        np.testing.assert_allclose(dev.decode(torch.as_tensor(zp, device=dev.device)).cpu().numpy(), ora.predict([zp]), atol=2e-5)
        return
    # ---- VAE ----
    c, inc = int(r.integers(2, 100)), int(r.choice([0, 0, 9, 40]))
    inc = min(inc, 206 - c)
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        k = 1.0 / np.sqrt(i)
        return ((torch.rand(o, i, generator=g) * 2 - 1) * k).numpy(), ((torch.rand(o, generator=g) * 2 - 1) * k).numpy()
    vp = {}
    for name, (o, i) in (("fc1", (h, N)), ("fc21", (c, h)), ("fc22", (c, h)), ("fc3", (h, c + inc)), ("fc4", (N, h))):
        vp[name + ".weight"], vp[name + ".bias"] = lin(o, i)
    dev = HipAAE(N, h, c, cond_inc=inc, max_batch=B, rng_mode="inject", gen_lr=2e-3, reg_lr=2e-3, dropout=(0.0, 0.0),
                 activation=act, vae=True)
    dev.load_params({"enc.lin1.weight": vp["fc1.weight"], "enc.lin1.bias": vp["fc1.bias"],
                     "enc.lin3.weight": np.vstack([vp["fc21.weight"], vp["fc22.weight"]]),
                     "enc.lin3.bias": np.concatenate([vp["fc21.bias"], vp["fc22.bias"]]),
                     "dec.lin1.weight": vp["fc3.weight"], "dec.lin1.bias": vp["fc3.bias"],
                     "dec.lin3.weight": vp["fc4.weight"], "dec.lin3.bias": vp["fc4.bias"]})
    ora = O.OracleVAE(vp, lr=2e-3, activation=act, conditions=[O.ConcatConst(inc)] if inc else [])
    for s in range(3):
        Bs = B if s < 2 else max(1, B - int(r.integers(0, min(B, 9))))
        ip, idx, val = _batch(r, N, Bs, B)
        eps = r.standard_normal((Bs, c)).astype(np.float32)
        cond = (r.standard_normal((Bs, inc)) * 0.4).astype(np.float32) if inc else None
        dev.vae_step(DeviceCSR.from_arrays(ip, idx, val, N, dev.device), 0, Bs,
                     cond=torch.as_tensor(cond, device=dev.device) if inc else None, eps=eps)
        want = ora.partial_fit(ip, idx, val, eps, [cond] if inc else None)
        l = dev.losses()
        np.testing.assert_allclose((l[0] + l[1]) / Bs, want, rtol=5e-5)
    got = dev.state_dict()
    want = {"enc.lin1.weight": ora.p["fc1.weight"], "enc.lin3.weight": np.vstack([ora.p["fc21.weight"], ora.p["fc22.weight"]]),
            "dec.lin1.weight": ora.p["fc3.weight"], "dec.lin3.weight": ora.p["fc4.weight"], "dec.lin3.bias": ora.p["fc4.bias"]}
    for k, w in want.items():
        _close_enough(got[k], w, 5e-5, 6e-3, f"vae N={N} h={h} c={c} inc={inc} B={B} {k}")


@pytest.mark.parametrize("scheme", ["vocab", "replicated", "vocab2", "shard"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "6"))))
def test_random_multi_rank_runs_match_oracle(seed, scheme):
    """Both data-parallel drivers with 2-4 ranks as threads on this GPU over random shapes: vocabularies that do not
    divide by the world size (uneven item slices / the un-sharded fallback), global batches on both sides of the fused
    output-layer kernel's row limit, dropout masks and a condition block sliced per rank - against the oracle's
    single-process step."""
    import threading
    import scipy.sparse as sp
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from aaerec.parallel import DataParallelAAE, VocabParallelAAE, ItemShardedAAE, item_items
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    from test_parity_abi_gpu import _ThreadDist
    r = np.random.default_rng(12000 + seed)
    world = int(r.integers(2, 5))
    Bl = int(r.integers(1, 50))
    B = Bl * world
    N, h, c = int(r.integers(world + 20, 1200)), int(r.integers(4, 150)), int(r.integers(2, 60))
    inc = int(r.choice([0, 0, 11]))
    drop = bool(r.integers(0, 2))
    p = (0.2, 0.2) if drop else (0.0, 0.0)
    kw = dict(gen_lr=2e-3, reg_lr=1e-3, dropout=p, activation=str(r.choice(["ReLU", "Tanh", "ELU"])))
    if os.environ.get("AAE_FUZZ_ACT"):              # (hunts: every multi-rank scheme with one activation class, e.g. GELU)
        kw["activation"] = os.environ["AAE_FUZZ_ACT"]
    params = init_params(N, h, c, cond_inc=inc, seed=seed)
    ora = O.OracleAAE(params, conditions=[O.ConcatConst(inc)] if inc else [], **kw)
    steps = []
    for s in range(3):
        ip, idx, val = _batch(r, N, B, B)
        masks = [(r.random((B, h)) > 0.2).astype(np.uint8) for _ in range(12)] if drop else None
        zr = r.standard_normal((B, c)).astype(np.float32)
        cond = (r.standard_normal((B, inc)) * 0.4).astype(np.float32) if inc else None
        steps.append((ip, idx, val, masks, zr, cond, ora.partial_fit(ip, idx, val, zr, masks, [cond] if inc else None)))
    dist = _ThreadDist(world)
    locals_, slices, errors = [None] * world, [None] * world, []

    def rank_main(rk):
        try:
            dist.bind(rk)
            inter = scheme in ("vocab2", "shard") and seed % 3 != 0      # items rk, rk + world, ... (what fit() uses) or [lo, hi)
            items = item_items(N, rk, world, inter)
            m = HipAAE(N, h, c, cond_inc=inc, max_batch=Bl, rng_mode="inject", grad_mode="fused" if scheme == "shard" else "export",
                       dp_world=world, **kw)
            m.load_params(params)
            sp_params = dict(params)
            sp_params["dec.lin3.weight"], sp_params["dec.lin3.bias"] = params["dec.lin3.weight"][items], params["dec.lin3.bias"][items]
            sp_params["enc.lin1.weight"] = params["enc.lin1.weight"][:, items]
            locals_[rk] = m
            if scheme == "replicated":
                dp = DataParallelAAE(m, dist, shard_decoder=True)
                for ip, idx, val, masks, zr, cond, want in steps:
                    X = sp.csr_matrix((val, idx, ip), shape=(B, N))
                    mk = None if masks is None else [k[rk * Bl:(rk + 1) * Bl] for k in masks]
                    cond_fn = None
                    if inc:
                        ct = torch.as_tensor(cond[rk * Bl:(rk + 1) * Bl], device=m.device)
                        cond_fn = lambda z, ct=ct: (torch.cat([z, ct], 1), lambda dzc: dzc[:, :c].contiguous())   # noqa: E731
                    dp.step(DeviceCSR(X, m.device), rk * Bl, Bl, global_rows=B, cond_fn=cond_fn, masks=mk,
                            z_real=zr[rk * Bl:(rk + 1) * Bl])
                    dp.wait_pending()
                return
            sl = HipAAE(len(sp_params["dec.lin3.bias"]), h, c, cond_inc=inc, max_batch=B, rng_mode="inject", blocked_output=bool(seed % 2), **kw)
            sl.load_params(sp_params)
            slices[rk] = sl
            if scheme == "shard":
                # ONE training handle per rank (its item slice + the hidden layers), the WHOLE batch through it, three
                # all-reduces of partial sums per step (aae_shard_step); `m` only receives the result
                sh = ItemShardedAAE(m, sl, dist, N, interleaved=inter)
                for ip, idx, val, masks, zr, cond, want in steps:
                    X = sp.csr_matrix((val, idx, ip), shape=(B, N))
                    sl.set_doc_l1(torch.as_tensor(np.asarray(abs(X).sum(1), dtype=np.float32).reshape(-1), device=m.device))
                    # (even seeds: a batch named ahead in the documented (row_start, rows, n_rows) form - here a hint that
                    #  the next step does not honour: the library must fall back, not trust it)
                    sh.step(None, 0, B, DeviceCSR(X[:, items].tocsr(), m.device), 0, B,
                            cond=torch.as_tensor(cond, device=m.device) if inc else None, masks=masks, z_real=zr,
                            next_rows=(0, None, B) if seed % 2 == 0 else None)
                    loss = sh.recon_loss()
                    if rk == 0:
                        np.testing.assert_allclose(loss, want[0], rtol=5e-5)
                        np.testing.assert_allclose(sl.losses()[1:], want[1:], rtol=5e-5, atol=2e-6)
                assert sh.comm_stats()["collectives"] == 3
                sh.gather_output_layer()
                return
            vp = VocabParallelAAE(m, sl, dist, N, shard_first_layer=scheme == "vocab2", interleaved=inter)
            for ip, idx, val, masks, zr, cond, want in steps:
                X = sp.csr_matrix((val, idx, ip), shape=(B, N))
                mk = None if masks is None else [k[rk * Bl:(rk + 1) * Bl] for k in masks]
                if scheme == "vocab2":                   # the slice sees its columns only: the rows' complete L1 norms
                    sl.set_doc_l1(torch.as_tensor(np.asarray(abs(X).sum(1), dtype=np.float32).reshape(-1), device=m.device))
                vp.step(DeviceCSR(X, m.device), rk * Bl, Bl, DeviceCSR(X[:, items].tocsr(), m.device), 0, B,
                        cond=torch.as_tensor(cond[rk * Bl:(rk + 1) * Bl], device=m.device) if inc else None, masks=mk,
                        z_real=zr[rk * Bl:(rk + 1) * Bl])
                loss = vp.recon_loss()
                if rk == 0:
                    np.testing.assert_allclose(loss, want[0], rtol=5e-5)
            if scheme == "vocab2":
                vp.gather_output_layer()                 # enc.lin1 and dec.lin3 of every slice into every replica
        except BaseException as e:              # noqa: B902 - a dead rank must not leave the others at a barrier
            errors.append((rk, e))
            dist.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(rk,)) for rk in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    got0 = locals_[0].state_dict()
    tag = f"{scheme} world={world} N={N} h={h} c={c} B={B} inc={inc}"
    for k, w in ora.p.items():
        if k.startswith("dec.lin3") and scheme == "vocab":
            got = np.concatenate([slices[rk].state_dict()[k] for rk in range(world)])
        else:
            got = got0[k]
            for rk in range(1, world):
                np.testing.assert_array_equal(locals_[rk].state_dict()[k], got, err_msg=f"{tag} rank {rk} {k}")
        _close_enough(got, w, 5e-5, 6e-3, f"{tag} {k}")
    if scheme == "shard":
        # the optimisers' state travels with the trained model (r5): enc_optim.state_dict() & co. read the
        # full-vocabulary handle, which took no part in a step
        for which, opt, net in (("enc", ora.opt_enc, "enc"), ("dec", ora.opt_dec, "dec"), ("gen", ora.opt_gen, "enc"),
                                ("disc", ora.opt_disc, "disc")):
            st = locals_[0].adam_state(which)
            for k in opt.m:
                if not k.startswith(net + "."):
                    continue
                gm, gv = st[k.split(".", 1)[1]]
                assert st["step"] == opt.t[k], f"{tag} {which} step"
                _close_enough(gm, opt.m[k], 2e-6 + 2e-3 * float(np.abs(opt.m[k]).max()), 1.0, f"{tag} {which} m {k}")
                _close_enough(gv, opt.v[k], 1e-9 + 4e-3 * float(np.abs(opt.v[k]).max()), 1.0, f"{tag} {which} v {k}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "4"))))
def test_random_large_vocabulary_matches_oracle(seed):
    """Vocabularies of 9-60 k items: more 32-item tiles than compute units, so every workgroup of the fused output-layer
    kernel walks several tiles (its cross-tile software pipeline: prefetched entries, moments and parameter stores of
    the previous tile in flight) - against the oracle, plus the three-kernel path on the same batches."""
    import torch
    from aaerec._hip import HipAAE, DeviceCSR
    from oracle import aae_oracle as O
    from oracle.dense_torch_port import init_params
    r = np.random.default_rng(15000 + seed)
    N, h, c, B = int(r.integers(9000, 60000)), int(r.integers(20, 208)), int(r.integers(4, 60)), int(r.integers(3, 105))
    drop = bool(r.integers(0, 2))
    p = (0.2, 0.2) if drop else (0.0, 0.0)
    kw = dict(gen_lr=2e-3, reg_lr=1e-3, dropout=p, activation=str(r.choice(["ReLU", "Tanh"])))
    params = init_params(N, h, c, seed=seed)
    fused = HipAAE(N, h, c, max_batch=B, rng_mode="inject", **kw)
    plain = HipAAE(N, h, c, max_batch=B, rng_mode="inject", unfused_decoder=True, **kw)
    fused.load_params(params)
    plain.load_params(params)
    ora = O.OracleAAE(params, **kw)
    for s in range(3):
        ip, idx, val = _batch(r, N, B, B, max_len=40)
        masks = [(r.random((B, h)) > 0.2).astype(np.uint8) for _ in range(12)] if drop else None
        zr = r.standard_normal((B, c)).astype(np.float32)
        csr = DeviceCSR.from_arrays(ip, idx, val, N, fused.device)
        fused.step(csr, 0, B, masks=masks, z_real=zr)
        plain.step(csr, 0, B, masks=masks, z_real=zr)
        want = ora.partial_fit(ip, idx, val, zr, masks)
        np.testing.assert_allclose(fused.losses(), want, rtol=5e-5, atol=2e-6, err_msg=f"N={N} h={h} c={c} B={B} step {s}")
        np.testing.assert_allclose(plain.losses(), want, rtol=5e-5, atol=2e-6)
    gf, gp = fused.state_dict(), plain.state_dict()
    for k, w in ora.p.items():
        for name, got in (("fused", gf[k]), ("three-kernel", gp[k])):
            _close_enough(got, w, 5e-5, 6e-3, f"{name} N={N} h={h} c={c} B={B} {k}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("AAE_FUZZ_SEEDS", "10"))))
def test_random_categorical_condition_kernels_match_oracle(seed):
    """aae_cat_encode / aae_cat_update (+ aae_csr_embed) over random shapes: table widths 1..256, list widths 1..80,
    vocabularies smaller than a batch (every row shared by many slots), sum / mean, SparseAdam / dense Adam."""
    import scipy.sparse as sp
    import torch
    from aaerec import _hip
    from oracle import aae_oracle as O
    r = np.random.default_rng(20000 + seed)
    rows, width = int(r.integers(1, 200)), int(r.choice([1, 1, 2, 5, 17, 80]))
    vocab, dim = int(r.integers(2, 400)), int(r.choice([1, 3, 32, 64, 65, 200, 256]))
    mean, sparse = bool(r.integers(0, 2)), bool(r.integers(0, 2))
    dev = torch.device("cuda:0")
    table0 = (r.standard_normal((vocab, dim)) * 0.1).astype(np.float32)
    table0[0] = 0
    ora = O.CategoricalEmbedding(table0, lr=5e-3, reduce="mean" if mean else "sum", sparse=sparse)
    table = torch.from_numpy(table0.copy()).to(dev)
    m, v = torch.zeros_like(table), torch.zeros_like(table)
    scratch = None if sparse else torch.zeros_like(table)
    for step in range(1, 5):
        idx = r.integers(0, vocab, size=(rows, width))
        idx[r.random((rows, width)) < 0.25] = 0
        d = (r.standard_normal((rows, dim)) * 0.05).astype(np.float32)
        idx_dev = torch.from_numpy(idx.astype(np.int32)).to(dev)
        out = torch.empty(rows, dim, device=dev)
        _hip.cat_encode(table, idx_dev, out, mean=mean)
        np.testing.assert_allclose(out.cpu().numpy(), ora.encode(idx), atol=2e-6, rtol=1e-5)
        _hip.cat_update(table, m, v, idx_dev, torch.from_numpy(d).to(dev), 5e-3, step, mean=mean, grad_scratch=scratch)
        ora._c = 0
        ora.bwd(d)
        ora.step()
        tag = f"rows={rows} width={width} vocab={vocab} dim={dim} mean={mean} sparse={sparse} step {step}"
        np.testing.assert_allclose(table.cpu().numpy(), ora.params["w"], atol=5e-6, rtol=0, err_msg=tag)
        om, ov = (ora.opt.m, ora.opt.v) if sparse else (ora.opt.m["w"], ora.opt.v["w"])
        np.testing.assert_allclose(m.cpu().numpy(), om, atol=1e-8, rtol=1e-4, err_msg=tag)
        np.testing.assert_allclose(v.cpu().numpy(), ov, atol=1e-12, rtol=1e-4, err_msg=tag)
    # the TF-IDF x embedding product on the same table: a random sparse weight matrix times the table
    S = sp.random(rows, vocab, density=min(1.0, 6.0 / vocab), format="csr", random_state=int(seed), dtype=np.float32)
    got = _hip.csr_embed(_hip.DeviceCSR(S, dev), table).cpu().numpy()
    np.testing.assert_allclose(got, S @ table.cpu().numpy(), atol=1e-5)
