"""Pins the CPU oracle (oracle/aae_oracle.py) to the golden vectors produced by the
real reference (tools/gen_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from golden_util import ACT_CASES, CAT_CASES, STEP_CASES, Fixture
from oracle import aae_oracle as O

TOL_LOSS = 2e-6     # relative, fp32 summation order only
# absolute on parameters (values are O(0.1); Adam steps are O(lr)=1e-3).  Adam's first steps
# compute lr*g/(|g|+1e-8): where |g| is itself ~1e-8 an fp32 summation-order difference in g
# moves the update by a few 1e-6, so the bound is a small multiple of lr*1e-3, not 1 ulp.
TOL_PARAM = 5e-6


def build_oracle(fx):
    conds = []
    cond = fx.cfg["cond"]
    if cond.startswith("concat") and cond.split("+")[0][6:].isdigit() and "cat" not in fx.cfg:
        conds.append(O.ConcatConst(int(cond.split("+")[0][6:])))               # "concat30", "concat300", "concat30+bias"
    if cond == "concat30+bias":
        conds.append(O.BiasConst())
    if cond == "categorical":
        conds.append(O.CategoricalSum(fx.z["init.cond.embedding"], lr=1e-2))
    elif "cat" in fx.cfg:
        k = fx.cfg["cat"]
        if k["concat"]:
            conds.append(O.ConcatConst(30))
        conds.append(O.CategoricalEmbedding(fx.z["init.cond.embedding"], lr=k["lr"], reduce=k["reduce"], sparse=k["sparse"]))
    return O.OracleAAE(fx.init_params(), conditions=conds, **fx.model_kwargs())


@pytest.mark.parametrize("name", STEP_CASES + CAT_CASES + ACT_CASES)
def test_oracle_reproduces_reference_steps(name):
    fx = Fixture(name)
    m = build_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        losses = m.partial_fit(ip, idx, val, fx.z[f"step{s}.z_real"], fx.masks(s), fx.cond_inputs(s))
        want = fx.z[f"step{s}.losses"]
        np.testing.assert_allclose(losses, want, rtol=TOL_LOSS, atol=1e-7, err_msg=f"{name} step {s} losses")
        if s == 0 and "step0.act.enc_a1_ae" in fx.z.files:
            np.testing.assert_allclose(m.last["enc_a1"], fx.z["step0.act.enc_a1_ae"], atol=1e-6)
            np.testing.assert_allclose(m.last["z"], fx.z["step0.act.enc_z_ae"], atol=1e-6)
            np.testing.assert_allclose(m.last["xhat"], fx.z["step0.act.dec_xhat"], atol=1e-6)
            np.testing.assert_allclose(m.last["z_disc"], fx.z["step0.act.enc_z_disc"], atol=1e-6)
            np.testing.assert_allclose(m.last["z_gen"], fx.z["step0.act.enc_z_gen"], atol=1e-6)
        if not fx.has_state(s):
            continue
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(m.p[k], w, atol=TOL_PARAM, rtol=0, err_msg=f"{name} step {s} {k}")
        opts = {"A_enc": m.opt_enc, "A_dec": m.opt_dec, "A_gen": m.opt_gen, "A_disc": m.opt_disc}
        for (tag, k), (em, ev, et) in fx.expected_adam(s).items():
            o = opts[tag]
            assert o.t[k] == et
            np.testing.assert_allclose(o.m[k], em, atol=1e-9, rtol=2e-5, err_msg=f"{name} {tag} m {k}")
            np.testing.assert_allclose(o.v[k], ev, atol=1e-13, rtol=5e-5, err_msg=f"{name} {tag} v {k}")
        if fx.cfg["cond"] == "categorical" or "cat" in fx.cfg:
            cat = m.conditions[-1]
            np.testing.assert_allclose(cat.params["w"], fx.z[f"step{s}.cond.embedding"], atol=TOL_PARAM)
            if f"step{s}.cond.m" in fx.z.files:       # the condition's own optimiser (SparseAdam: touched rows only)
                assert cat.opt.t == float(fx.z[f"step{s}.cond.t"])
                np.testing.assert_allclose(cat.opt.m, fx.z[f"step{s}.cond.m"], atol=1e-9, rtol=2e-5)
                np.testing.assert_allclose(cat.opt.v, fx.z[f"step{s}.cond.v"], atol=1e-13, rtol=5e-5)


@pytest.mark.parametrize("name", ["step_nodrop_gauss", "step_cond_concat", "step_cond_concat_bias",
                                  "step_selu", "step_categorical_prior", "step_ragged"] + CAT_CASES)
def test_oracle_predict(name):
    fx = Fixture(name)
    m = build_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        m.partial_fit(ip, idx, val, fx.z[f"step{s}.z_real"], fx.masks(s), fx.cond_inputs(s))
    ip, idx, val = fx.batch(0, prefix="predict")
    got = m.predict(ip, idx, val, fx.cond_inputs(0, prefix="predict"))
    np.testing.assert_allclose(got, fx.z["predict.out"], atol=1e-6)


def test_oracle_ae_step_matches_reference_autoencoder():
    """The reference's plain AutoEncoder (aae.py:221-458) is the oracle's ae_step alone."""
    fx = Fixture("step_ae_only")
    m = build_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        loss = m.ae_step(ip, idx, val, fx.masks(s))
        np.testing.assert_allclose(loss, fx.z[f"step{s}.losses"][0], rtol=TOL_LOSS)
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(m.p[k], w, atol=TOL_PARAM, rtol=0, err_msg=k)


def test_oracle_rejects_counts_above_one():
    """Reference behaviour (torch BCE): duplicate items give value 2.0 -> RuntimeError."""
    fx = Fixture("step_nodrop_gauss")
    m = build_oracle(fx)
    ip, idx, val = fx.batch(0)
    val = val.copy()
    val[0] = 2.0
    with pytest.raises(RuntimeError):
        m.partial_fit(ip, idx, val, fx.z["step0.z_real"])


# ---------------------------------------------------------------------------------------------
# the PyTorch-CPU dense port (bench.py's cpu_baseline) is pinned to the same golden vectors
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["step_nodrop_gauss", "step_masks", "step_selu", "step_categorical_prior",
                                  "step_prior_scale", "step_sgd", "step_nonorm", "step_ragged", "step_cond_concat"])
def test_dense_port_reproduces_reference_steps(name):
    import scipy.sparse as sp
    from oracle.dense_torch_port import DenseTorchAAE
    fx = Fixture(name)
    m = DenseTorchAAE(fx.init_params(), **fx.model_kwargs())
    N = fx.cfg["N"]
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, N)).toarray()
        cond = fx.cond_inputs(s)
        losses = m.partial_fit(X, z_real=fx.z[f"step{s}.z_real"], masks=fx.masks(s) or [None] * 12,
                               cond=cond[0] if cond else None)
        np.testing.assert_allclose(losses, fx.z[f"step{s}.losses"], rtol=2e-6, atol=1e-7)
        if fx.has_state(s):
            got = m.state_dict()
            for k, w in fx.expected_params(s).items():
                np.testing.assert_allclose(got[k], w, atol=TOL_PARAM, rtol=0, err_msg=f"{name} step {s} {k}")


DECODING_CASES = ["step_decoding", "step_decoding_trainable"]


def build_decoding_oracle(fx):
    c = fx.cfg
    incs = c["incs"]
    conds = [O.CategoricalSum(fx.z["init.cond.embedding"], lr=1e-2) if c["trainable"] else O.ConcatConst(incs[0])]
    conds += [O.ConcatConst(i) for i in incs[1:]]
    return O.OracleDecoder(fx.init_params(), lr=c["gen_lr"], dropout=tuple(c["dropout"]), conditions=conds)


@pytest.mark.parametrize("name", DECODING_CASES)
def test_oracle_reproduces_reference_decoding_recommender(name):
    """DecodingRecommender.partial_fit / predict (aae.py:461-584) against fixtures made from the reference."""
    fx = Fixture(name)
    m = build_decoding_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        loss = m.partial_fit(fx.cond_inputs(s), ip, idx, val, fx.masks(s))
        np.testing.assert_allclose(loss, fx.z[f"step{s}.losses"][0], rtol=TOL_LOSS, atol=1e-7)
        for k in ("lin1.weight", "lin1.bias", "lin2.weight", "lin2.bias", "lin3.weight", "lin3.bias"):
            np.testing.assert_allclose(m.p["dec." + k], fx.z[f"step{s}.dec.{k}"], atol=TOL_PARAM, rtol=0,
                                       err_msg=f"{name} step {s} {k}")
        for (tag, k), (em, ev, et) in fx.expected_adam(s).items():
            assert tag == "A_dec" and m.opt_dec.t[k] == et
            np.testing.assert_allclose(m.opt_dec.m[k], em, atol=1e-9, rtol=2e-5)
            np.testing.assert_allclose(m.opt_dec.v[k], ev, atol=1e-13, rtol=5e-5)
        if fx.cfg["trainable"]:
            np.testing.assert_allclose(m.conditions[0].params["w"], fx.z[f"step{s}.cond.embedding"], atol=TOL_PARAM)
    out = m.predict(fx.cond_inputs(0, prefix="predict"))
    np.testing.assert_allclose(out, fx.z["predict.out"], atol=2e-6)


def test_oracle_ae_step_matches_reference_denoising_autoencoder():
    """DenoisingAutoEncoder, corrupt='zeros' (dae.py:48-52, 189-210): the reference thins the batch tensor in
    place, so its step is ae_step on the thinned bag - the fixture records which CSR entries survived."""
    fx = Fixture("step_dae")
    m = build_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        loss = m.ae_step(ip, idx, val * fx.z[f"step{s}.keep"], fx.masks(s))
        np.testing.assert_allclose(loss, fx.z[f"step{s}.losses"][0], rtol=TOL_LOSS)
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(m.p[k], w, atol=TOL_PARAM, rtol=0, err_msg=k)
    ip, idx, val = fx.batch(0, prefix="predict")
    np.testing.assert_allclose(m.predict(ip, idx, val), fx.z["predict.out"], atol=2e-6)


def test_oracle_ae_step_matches_reference_denoising_autoencoder_gauss():
    """DenoisingAutoEncoder, corrupt='gauss' (dae.py:40-45, 191): the encoder reads the dense batch + the recorded N(0,
    noise_factor) on all N columns, the BCE target is the clean batch."""
    fx = Fixture("step_dae_gauss")
    m = build_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        loss = m.ae_step(ip, idx, val, fx.masks(s), input_noise=fx.z[f"step{s}.noise"])
        np.testing.assert_allclose(loss, fx.z[f"step{s}.losses"][0], rtol=TOL_LOSS)
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(m.p[k], w, atol=TOL_PARAM, rtol=0, err_msg=k)
    ip, idx, val = fx.batch(0, prefix="predict")
    np.testing.assert_allclose(m.predict(ip, idx, val), fx.z["predict.out"], atol=2e-6)


VAE_CASES = ["step_vae", "step_vae_cond", "step_vae_cat"]
VAE_NAMES = ("fc1", "fc21", "fc22", "fc3", "fc4")


def build_vae_oracle(fx):
    conds = [O.ConcatConst(30)] if fx.cfg["cond"] == "concat30" else []
    if fx.cfg["cond"] == "cat":       # trainable CategoricalCondition (embedding sum, SparseAdam) behind the code
        k = fx.cfg["cat"]
        conds = [O.CategoricalEmbedding(fx.z["init.cond.embedding"], lr=k["lr"], reduce=k["reduce"], sparse=k["sparse"])]
    params = {f"{n}.{t}": fx.z[f"init.{n}.{t}"] for n in VAE_NAMES for t in ("weight", "bias")}
    return O.OracleVAE(params, lr=fx.cfg["gen_lr"], conditions=conds)


@pytest.mark.parametrize("name", VAE_CASES)
def test_oracle_reproduces_reference_vae(name):
    """VAE.partial_fit / predict (vae.py:47-266) with the recorded eps of reparametrize()."""
    fx = Fixture(name)
    m = build_vae_oracle(fx)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        loss = m.partial_fit(ip, idx, val, fx.z[f"step{s}.eps"], fx.cond_inputs(s))
        np.testing.assert_allclose(loss, fx.z[f"step{s}.losses"][0], rtol=5e-6)
        for n in VAE_NAMES:
            for t in ("weight", "bias"):
                k = f"{n}.{t}"
                np.testing.assert_allclose(m.p[k], fx.z[f"step{s}.{k}"], atol=TOL_PARAM, rtol=0, err_msg=f"{name} {s} {k}")
                assert m.opt.t[k] == float(fx.z[f"step{s}.A.{k}.t"])
                np.testing.assert_allclose(m.opt.m[k], fx.z[f"step{s}.A.{k}.m"], atol=2e-8, rtol=2e-4)
                np.testing.assert_allclose(m.opt.v[k], fx.z[f"step{s}.A.{k}.v"], atol=1e-12, rtol=2e-4)
        if fx.cfg["cond"] == "cat":
            np.testing.assert_allclose(m.conditions[0].params["w"], fx.z[f"step{s}.cond.embedding"], atol=TOL_PARAM)
    ip, idx, val = fx.batch(0, prefix="predict")
    out = m.predict(ip, idx, val, fx.z["predict.eps"], fx.cond_inputs(0, prefix="predict"))
    np.testing.assert_allclose(out, fx.z["predict.out"], atol=2e-6)
