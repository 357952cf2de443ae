"""GPU tests of the host-side mirror (aaerec.aae) - through AdversarialAutoEncoder /
AAERecommender exactly as the reference's drivers call them."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import free_port, spawn_ranks, CAT_CASES, Fixture

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _e2e():
    z = np.load(os.path.join(GOLDEN, "e2e_c1.npz"))
    N = int(z["N"])

    def csr(p, i):
        return sp.csr_matrix((np.ones(len(i)), i, p), shape=(len(p) - 1, N))
    return z, csr(z["train_indptr"], z["train_indices"]), csr(z["in_indptr"], z["in_indices"]), \
        csr(z["out_indptr"], z["out_indices"])


def test_fit_predict_tracks_reference_with_reference_rng():
    """Same seeds, rng_mode='reference': initial weights, epoch permutations, dropout masks and
    z_real are the reference's own draws, so after 3 epochs (51 steps incl. short last batches)
    the predictions must agree with the reference's to fp32 accumulation noise."""
    from aaerec.aae import AdversarialAutoEncoder
    z, Xtr, Xin, _ = _e2e()
    seed = int(z["short_seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, gen_lr=0.01, reg_lr=0.001,
                               verbose=False, rng_mode="reference")
    m.fit(Xtr)
    pred = m.predict(Xin[:40])
    assert pred.dtype == np.float32 and pred.shape == z["pred_short"].shape
    # north star: reconstructions within 1e-4 (fp32)
    np.testing.assert_allclose(pred, z["pred_short"], atol=1e-4)


def _c1_mrr(seed, rng_mode):
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec.evaluation import remove_non_missing, METRICS
    z, Xtr, Xin, Yout = _e2e()
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=100, batch_size=100, gen_lr=0.01, reg_lr=0.001,
                               verbose=False, rng_mode=rng_mode)
    m.fit(Xtr)
    pred = remove_non_missing(m.predict(Xin), Xin, copy=True)
    return METRICS["mrr@10"](Yout.toarray(), pred)[0]


def test_c1_mrr_parity():
    """Config C1 end to end (1k items, 1800 train docs, h=50, 100 epochs = 1800 steps, 200 test
    docs), MRR@10 against the reference's own runs (tests/golden/e2e_c1.npz, 8 seeds).

    The training is noisy: the reference's MRR@10 over its 8 seeds spans 0.13 .. 0.29 (std 0.05),
    so a +-0.001 comparison is only meaningful between runs that share their randomness:
      (a) rng_mode='reference' re-uses the reference's draws (init, shuffles, masks, z_real); the
          two trajectories then differ by fp32 rounding only, which 1800 Adam steps amplify -
          the mean over 3 seeds is compared with a tolerance of 0.06 and printed;
      (b) the production device RNG is a different random stream: compared in distribution
          (mean over 8 seeds within 2 standard errors of the reference's mean)."""
    z = np.load(os.path.join(GOLDEN, "e2e_c1.npz"))
    ref = z["ref_mrr10"]
    same = [_c1_mrr(s, "reference") for s in range(3)]
    print("MRR@10 reference-rng", same, "reference", ref[:3].tolist())
    # (51 steps in, predictions still agree to 1e-4 - test above; over 1800 steps fp32 rounding
    # differences grow chaotically, observed per-seed wander between builds of this repo: +-0.03)
    assert abs(np.mean(same) - ref[:3].mean()) < 0.06
    dev = [_c1_mrr(s, "device") for s in range(8)]
    print("MRR@10 device-rng", dev, "reference", ref.tolist())
    se = np.sqrt(ref.std() ** 2 / len(ref) + np.std(dev) ** 2 / len(dev))
    assert abs(np.mean(dev) - ref.mean()) < max(2.5 * se, 0.03), (np.mean(dev), ref.mean(), se)
    # every run learned (random ranking gives ~0.003); one slow seed in eight is within what the reference shows
    assert min(dev) > 0.01 and sorted(dev)[1] > 0.05, dev


@pytest.mark.parametrize("name", ["step_cond_categorical", "step_cond_concat_bias", "step_cond_concat"])
def test_condition_plugins_through_autograd_bridge(name):
    """Conditions as real plugin objects (trainable embedding / bias / constant concat) driven
    through AdversarialAutoEncoder.partial_fit; compared with the reference fixtures."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec import condition as C
    fx = Fixture(name)
    cfg = fx.cfg

    class ConstConcat(C.ConcatenationBasedConditioning):
        def size_increment(self):
            return 30

        def encode(self, inputs):
            return torch.as_tensor(inputs, dtype=torch.float32, device="cuda")

    class ConstBias(C.ConditionalBiasing):
        def encode(self, inputs):
            return torch.as_tensor(inputs, dtype=torch.float32, device="cuda")

    if cfg["cond"] == "categorical":
        cat = C.CategoricalCondition(8, sparse=False, use_cuda=True, reduce="sum", lr=1e-2)
        V = fx.z["init.cond.embedding"].shape[0]
        cat.vocab = {"a%d" % i: i for i in range(1, V)}           # indices are given pre-transformed
        cat.embedding = torch.nn.Embedding(V, 8, padding_idx=0)
        with torch.no_grad():
            cat.embedding.weight.copy_(torch.from_numpy(fx.z["init.cond.embedding"]))
        cat.optimizer = torch.optim.Adam(cat.embedding.parameters(), lr=1e-2)
        conds = C.ConditionList([("authors", cat)])
    elif cfg["cond"] == "concat30+bias":
        conds = C.ConditionList([("title", ConstConcat()), ("b", ConstBias())])
    else:
        conds = C.ConditionList([("title", ConstConcat())])

    kw = fx.model_kwargs()
    m = AdversarialAutoEncoder(n_hidden=cfg["h"], n_code=cfg["c"], batch_size=cfg["B"], conditions=conds,
                               verbose=True, rng_mode="reference", **kw)
    m._build(cfg["N"], cfg["cond_inc"])
    m.hip.load_params(fx.init_params())
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        cin = fx.cond_inputs(s)
        if cfg["cond"] == "categorical":
            cin = [[[int(j) for j in row if j != 0] or [0] for row in cin[0]]]
        # inject the fixture's randomness instead of drawing it
        masks, zr = fx.masks(s), fx.z[f"step{s}.z_real"]
        m._host_randomness = lambda B, masks=masks, zr=zr: (masks, torch.from_numpy(zr))
        m.partial_fit(X, condition_data=cin)
        np.testing.assert_allclose(m.last_losses, fx.z[f"step{s}.losses"], rtol=1e-5, atol=1e-6)
        got = m.hip.state_dict()
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=f"{name} step {s} {k}")
        if cfg["cond"] == "categorical":
            np.testing.assert_allclose(cat.embedding.weight.detach().cpu().numpy(),
                                       fx.z[f"step{s}.cond.embedding"], atol=1e-5)
    # predict with conditions (ragged batches of 7, as the fixture was produced)
    ip, idx, val = fx.batch(0, prefix="predict")
    Xp = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
    pc = fx.cond_inputs(0, prefix="predict")
    if cfg["cond"] == "categorical":
        pc = [[[int(j) for j in row if j != 0] or [0] for row in pc[0]]]
    m.batch_size = 7
    np.testing.assert_allclose(m.predict(Xp, condition_data=pc), fx.z["predict.out"], atol=1e-5)


@pytest.mark.parametrize("name", ["step_cond_categorical"] + CAT_CASES)
def test_categorical_condition_device_native(name):
    """A CategoricalCondition with its table on the GPU is trained by the library's own kernels (aae_cat_encode before
    the fused step, aae_cat_update = backward + SparseAdam / Adam after it) instead of the autograd bridge: same
    fixtures from the reference, including the condition's optimiser state."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec import condition as C
    fx = Fixture(name)
    cfg = fx.cfg
    kind = cfg.get("cat", dict(sparse=False, reduce="sum", concat=False, lr=1e-2))

    class ConstConcat(C.ConcatenationBasedConditioning):
        constant_concat = True

        def size_increment(self):
            return 30

        def encode(self, inputs):
            return torch.as_tensor(inputs, dtype=torch.float32, device="cuda")

    cat = C.CategoricalCondition(8, sparse=kind["sparse"], use_cuda=True, reduce=kind["reduce"], lr=kind["lr"])
    assert cat.embedding_on_gpu
    V = fx.z["init.cond.embedding"].shape[0]
    cat.vocab = {"a%d" % i: i for i in range(1, V)}               # indices are given pre-transformed
    cat.embedding = torch.nn.Embedding(V, 8, padding_idx=0, sparse=kind["sparse"]).cuda()
    with torch.no_grad():
        cat.embedding.weight.copy_(torch.from_numpy(fx.z["init.cond.embedding"]))
    opt = torch.optim.SparseAdam if kind["sparse"] else torch.optim.Adam
    cat.optimizer = opt(cat.embedding.parameters(), lr=kind["lr"])
    items = [("authors", cat)]
    if kind["concat"]:
        items.insert(0, ("title", ConstConcat()))
    conds = C.ConditionList(items)

    def cat_inputs(arr):
        if kind["reduce"] is None:
            return [int(j) for j in arr]
        # the fixture stores the batch-padded lists; strip the padding again but keep the padded width (one row
        # always spans it), which 'mean' divides by
        return [[int(j) for j in row if j != 0] or [0] for row in arr]

    m = AdversarialAutoEncoder(n_hidden=cfg["h"], n_code=cfg["c"], batch_size=cfg["B"], conditions=conds,
                               verbose=True, rng_mode="reference", **fx.model_kwargs())
    m._build(cfg["N"], cfg["cond_inc"])
    assert m._is_device_native() and not m._is_constant_concat()
    m.hip.load_params(fx.init_params())
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        cin = fx.cond_inputs(s)
        cin[-1] = cat_inputs(cin[-1])
        masks, zr = fx.masks(s), fx.z[f"step{s}.z_real"]
        m._host_randomness = lambda B, masks=masks, zr=zr: (masks, torch.from_numpy(zr))
        m.partial_fit(X, condition_data=cin)
        np.testing.assert_allclose(m.last_losses, fx.z[f"step{s}.losses"], rtol=1e-5, atol=1e-6)
        got = m.hip.state_dict()
        for k, w in fx.expected_params(s).items():
            np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=f"{name} step {s} {k}")
        np.testing.assert_allclose(cat.embedding.weight.detach().cpu().numpy(), fx.z[f"step{s}.cond.embedding"],
                                   atol=1e-5, err_msg=f"{name} step {s} embedding")
        if f"step{s}.cond.m" in fx.z.files:
            st = cat.optimizer.state[cat.embedding.weight]
            assert float(st["step"]) == float(fx.z[f"step{s}.cond.t"])
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), fx.z[f"step{s}.cond.m"], atol=1e-8, rtol=1e-4)
            np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), fx.z[f"step{s}.cond.v"], atol=1e-12, rtol=1e-4)
    # the padding row never moves
    assert float(cat.embedding.weight.detach()[0].abs().max()) == 0.0
    ip, idx, val = fx.batch(0, prefix="predict")
    Xp = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
    pc = fx.cond_inputs(0, prefix="predict")
    pc[-1] = cat_inputs(pc[-1])
    m.batch_size = 7
    np.testing.assert_allclose(m.predict(Xp, condition_data=pc), fx.z["predict.out"], atol=1e-5)
    # the optimiser state is torch's own: the same condition keeps training through the autograd bridge
    sd = cat.optimizer.state_dict()
    assert sd["state"][0]["exp_avg"].shape == (V, 8)


def test_recommender_with_bags_and_evaluation_harness(capsys):
    """The path the reference's main.py drives: Bags -> Evaluation.setup -> AAERecommender.train /
    predict -> metrics (evaluation.py:330-404)."""
    from aaerec.aae import AAERecommender
    from aaerec.datasets import Bags
    from aaerec.evaluation import Evaluation
    rng = np.random.RandomState(0)
    protos = [rng.choice(300, size=8, replace=False) for _ in range(30)]
    data, owners, years = [], [], {}
    for i in range(600):
        p = protos[rng.randint(30)]
        data.append(["i%d" % t for t in rng.choice(p, size=rng.randint(4, 8), replace=False)])
        owners.append("d%d" % i)
        years["d%d" % i] = 2000 + (i * 10) // 600
    bags = Bags(data, owners, {"year": years})
    ev = Evaluation(bags, 2009, metrics=["mrr@10", "map@10"], logfile=None).setup(min_elements=2, drop=1)
    res = ev([AAERecommender(n_hidden=40, n_code=16, n_epochs=60, batch_size=50, gen_lr=0.01, verbose=False)])
    mrr = res[0][0][0]
    assert 0.15 < mrr <= 1.0, mrr
    out = capsys.readouterr().out
    assert "Training took" in out and "- mrr@10:" in out
    # (every metric bounded at k: the harness asked the recommender for its top 10 on the device - predict_topk - instead of
    #  the dense score matrix.)  The reference's dense pipeline on the same run of the same seeds gives the same numbers:
    def run(topk):
        np.random.seed(3)
        torch.manual_seed(3)
        e = Evaluation(bags, 2009, metrics=["mrr@10", "map@10", "p@5", "P@1"], logfile=None, topk=topk).setup(min_elements=2, drop=1)
        return e([AAERecommender(n_hidden=40, n_code=16, n_epochs=10, batch_size=50, gen_lr=0.01, verbose=False, seed=11)])[0]
    fast, dense = run(True), run(False)
    np.testing.assert_allclose(np.asarray(fast), np.asarray(dense), atol=1e-12)
    ev2 = Evaluation(bags, 2009, metrics=["mrr@10", "mrr"], logfile=None)
    assert ev2._bounded_k() is None and Evaluation(bags, 2009, metrics=["mrr@5", "p@20"], logfile=None)._bounded_k() == 20


def test_per_step_buffers_are_released_every_step():
    """the list that keeps a step's device operands alive (condition blocks, packets, decoder inputs) must not
    grow with the number of steps - 120 conditioned steps in the default rng mode, on every path that appends to it."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec import condition as C
    z, Xtr, Xin, Yout = _e2e()
    rng = np.random.RandomState(0)

    class ConstConcat(C.ConcatenationBasedConditioning):
        constant_concat = True

        def size_increment(self):
            return 6

        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")

    class Bias(C.ConditionalBiasing):
        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")
    cases = [([("t", ConstConcat())], [rng.standard_normal((Xtr.shape[0], 6)).astype(np.float32)]),
             ([("t", ConstConcat()), ("b", Bias())], [rng.standard_normal((Xtr.shape[0], 6)).astype(np.float32),
                                                      0.1 * rng.standard_normal((Xtr.shape[0], 56)).astype(np.float32)])]
    for items, data in cases:
        m = AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=1, batch_size=20, conditions=C.ConditionList(items),
                                   verbose=False)
        sizes = []
        for k, _ in enumerate(m.fit_steps(Xtr[:2400], condition_data=[d[:2400] for d in data])):
            sizes.append(len(m.hip._keep))
            if k >= 119:
                break
        assert max(sizes) <= 8 and sizes[-1] <= max(sizes[:10]), sizes[-5:]


def test_partial_fit_rejects_duplicate_items():
    from aaerec.aae import AdversarialAutoEncoder
    m = AdversarialAutoEncoder(n_hidden=8, n_code=4, batch_size=4, verbose=False)
    X = np.zeros((2, 20), dtype=np.float32)
    X[0, 3] = 2.0
    with pytest.raises(RuntimeError, match="between 0 and 1"):
        m.partial_fit(X)


def test_predict_topk_equals_host_pipeline():
    """On-device remove_non_missing + argtopk against the reference's host pipeline on the full
    score matrix: same top-10 ids (score ties aside), same scaled scores, same MRR/MAP/P."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec.evaluation import remove_non_missing, argtopk, evaluate, evaluate_topk
    z, Xtr, Xin, Yout = _e2e()
    torch.manual_seed(3)
    np.random.seed(3)
    m = AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=30, batch_size=100, gen_lr=0.01, reg_lr=0.001,
                               verbose=False)
    m.fit(Xtr)
    full = remove_non_missing(m.predict(Xin), Xin, copy=True)
    rows, cols = argtopk(full, 10)
    ids, vals = m.predict_topk(Xin, k=10)
    assert ids.shape == (Xin.shape[0], 10) and ids.dtype == np.int32
    np.testing.assert_allclose(vals, full[rows, cols], atol=2e-6)
    same = (ids == cols)
    # positions may only differ where the scores tie
    assert np.all(same | np.isclose(full[rows, ids], full[rows, cols], atol=1e-7))
    names = ["mrr@10", "map@10", "p@10", "mrr@5", "P@1"]
    np.testing.assert_allclose(np.asarray(evaluate_topk(Yout, ids, names)),
                               np.asarray(evaluate(Yout.toarray(), full, names)), atol=1e-12)
    # known items never come back
    Xd = Xin.toarray()
    assert Xd[np.arange(Xin.shape[0])[:, None], ids].sum() == 0


@pytest.mark.parametrize("kind", ["categorical_native", "concat_plus_bias_generic"])
def test_predict_topk_with_trainable_and_generic_conditions(kind):
    """predict_topk behind every kind of condition plugin predict() accepts (aae.py:844-865): the device-native
    CategoricalCondition rides in the fused call, any other plugin imposes itself between aae_encode and aae_decode_topk.
    Same ids / scaled scores as the host pipeline over predict()'s dense matrix."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec import condition as C
    from aaerec.evaluation import remove_non_missing, argtopk
    z, Xtr, Xin, Yout = _e2e()
    rng = np.random.RandomState(5)
    torch.manual_seed(5)
    np.random.seed(5)

    class ConstConcat(C.ConcatenationBasedConditioning):
        def size_increment(self):
            return 6

        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")

    class ConstBias(C.ConditionalBiasing):
        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")

    def attrs(n):
        if kind == "categorical_native":
            return [[["a%d" % rng.randint(1, 30) for _ in range(rng.randint(1, 4))] for _ in range(n)]]
        return [rng.standard_normal((n, 6)).astype(np.float32), 0.1 * rng.standard_normal((n, 56)).astype(np.float32)]     # (bias over code + the 6 concatenated columns)
    if kind == "categorical_native":
        conds = C.ConditionList([("authors", C.CategoricalCondition(8, use_cuda=True, reduce="sum", lr=0.01))])
        ctr = conds.fit_transform(attrs(Xtr.shape[0]))
        cte = conds.transform(attrs(Xin.shape[0]))
    else:
        conds = C.ConditionList([("title", ConstConcat()), ("b", ConstBias())])
        ctr, cte = attrs(Xtr.shape[0]), attrs(Xin.shape[0])
    m = AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=5, batch_size=100, gen_lr=0.01, reg_lr=0.001,
                               conditions=conds, verbose=False)
    m.fit(Xtr, condition_data=ctr)
    assert m._is_device_native() == (kind == "categorical_native")
    full = remove_non_missing(m.predict(Xin, condition_data=cte), Xin, copy=True)
    rows, cols = argtopk(full, 10)
    ids, vals = m.predict_topk(Xin, k=10, condition_data=cte)
    np.testing.assert_allclose(vals, full[rows, cols], atol=2e-6)
    assert np.all((ids == cols) | np.isclose(full[rows, ids], full[rows, cols], atol=1e-7))


def test_autoencoder_recommender_learns():
    """AAERecommender(adversarial=False) -> AutoEncoder (reference aae.py:221-458, 953-957)."""
    from aaerec.aae import AAERecommender, AutoEncoder
    from aaerec.evaluation import remove_non_missing, METRICS
    z, Xtr, Xin, Yout = _e2e()

    class Set:
        def __init__(self, X):
            self.X = X

        def tocsr(self):
            return self.X
    # The reference's AutoEncoder on this data learns late and seed-dependently (reference, seed 0: MRR@10
    # 0.022 after 40 epochs, 0.153 after 100).  Ours after 200 epochs over seeds 0..5: 0.05 - 0.42 (tools/
    # ae_seed_sweep.py), so the check is on the median of three seeds, not on one lucky or unlucky draw.
    mrr = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        np.random.seed(seed)
        rec = AAERecommender(adversarial=False, n_hidden=50, n_code=50, n_epochs=200, batch_size=100, lr=0.01,
                             verbose=False)
        rec.train(Set(Xtr))
        assert isinstance(rec.model, AutoEncoder)
        pred = remove_non_missing(rec.predict(Set(Xin)), Xin, copy=True)
        mrr.append(METRICS["mrr@10"](Yout.toarray(), pred)[0])
    assert np.median(mrr) > 0.08, mrr


def test_autoencoder_fit_tracks_reference_with_reference_rng():
    """Plain AutoEncoder, same seeds, rng_mode='reference': 3 epochs of the reference's AutoEncoder.fit
    (fixture tests/golden/e2e_ae_short.npz, written by tools/gen_golden.py e2e: AutoEncoder(n_hidden=50,
    n_code=50, n_epochs=3, batch_size=100, lr=0.01), seed 7)."""
    from aaerec.aae import AutoEncoder
    z, Xtr, Xin, _ = _e2e()
    want = np.load(os.path.join(GOLDEN, "e2e_ae_short.npz"))["pred_short"]
    torch.manual_seed(7)
    np.random.seed(7)
    m = AutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, lr=0.01, verbose=False, rng_mode="reference")
    m.fit(Xtr)
    np.testing.assert_allclose(m.predict(Xin[:40]), want, atol=1e-4)


@pytest.mark.parametrize("name", ["step_decoding", "step_decoding_trainable"])
def test_decoding_recommender_tracks_reference(name):
    """DecodingRecommender (reference aae.py:461-584) with real condition plugins (constant concatenated blocks;
    a trainable CategoricalCondition that gets dL/d(inputs) through autograd and steps its own Adam), replaying
    the reference fixtures with their recorded dropout masks."""
    from aaerec.aae import DecodingRecommender
    from aaerec import condition as C
    fx = Fixture(name)
    cfg = fx.cfg
    incs = cfg["incs"]

    def const_concat(inc):
        class ConstConcat(C.ConcatenationBasedConditioning):
            def size_increment(self):
                return inc

            def encode(self, inputs):
                return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")
        return ConstConcat()
    items, cat = [], None
    if cfg["trainable"]:
        cat = C.CategoricalCondition(incs[0], sparse=False, use_cuda=True, reduce="sum", lr=1e-2)
        V = fx.z["init.cond.embedding"].shape[0]
        cat.vocab = {"a%d" % i: i for i in range(1, V)}           # indices are given pre-transformed
        cat.embedding = torch.nn.Embedding(V, incs[0], padding_idx=0)
        with torch.no_grad():
            cat.embedding.weight.copy_(torch.from_numpy(fx.z["init.cond.embedding"]))
        cat.optimizer = torch.optim.Adam(cat.embedding.parameters(), lr=1e-2)
        items.append(("authors", cat))
    else:
        items.append(("title", const_concat(incs[0])))
    items += [(f"c{j}", const_concat(inc)) for j, inc in enumerate(incs[1:])]
    conds = C.ConditionList(items)
    m = DecodingRecommender(conds, n_epochs=1, batch_size=cfg["B"], n_hidden=cfg["h"], lr=cfg["gen_lr"], verbose=True,
                            rng_mode="reference", dropout=tuple(cfg["dropout"]))
    assert str(m).startswith("MLP-2 Decoder with %d hidden units" % cfg["h"])
    m._build(cfg["N"])
    m.hip.load_params(fx.init_params())

    def cin_of(prefix, s=0):
        cin = fx.cond_inputs(s, prefix=prefix)
        if cfg["trainable"]:
            cin[0] = [[int(j) for j in row if j != 0] or [0] for row in cin[0]]
        return cin
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        Y = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        masks = [torch.from_numpy(k) for k in fx.masks(s)]
        m._masks = lambda B, masks=masks: masks                  # the fixture's randomness instead of a fresh draw
        m.partial_fit(cin_of(None, s), torch.FloatTensor(Y.toarray()), step=s)
        np.testing.assert_allclose(m.last_loss, fx.z[f"step{s}.losses"][0], rtol=1e-5, atol=1e-6)
        got = m.mlp.state_dict()
        for k in ("lin1.weight", "lin1.bias", "lin2.weight", "lin2.bias", "lin3.weight", "lin3.bias"):
            np.testing.assert_allclose(got[k].numpy(), fx.z[f"step{s}.dec.{k}"], atol=1e-5, rtol=0,
                                       err_msg=f"{name} step {s} {k}")
        if cat is not None:
            np.testing.assert_allclose(cat.embedding.weight.detach().cpu().numpy(), fx.z[f"step{s}.cond.embedding"],
                                       atol=1e-5)
    m.batch_size = 7                                             # ragged predict batches
    out = m._predict_conditions(cin_of("predict"), cfg["B"])
    np.testing.assert_allclose(out, fx.z["predict.out"], atol=1e-5)
    with pytest.raises(TypeError):
        DecodingRecommender(conds, not_a_decoder_kwarg=1)


def test_decoding_recommender_learns_from_conditions():
    """End to end through train()/predict() on Bags-like sets: items are a noisy function of a 16-d condition
    vector, the decoder has to pick that up (MRR@10 far above chance)."""
    from aaerec.aae import DecodingRecommender
    from aaerec import condition as C
    from aaerec.evaluation import METRICS
    rs = np.random.RandomState(5)
    n, N, d = 1200, 400, 16
    proto = (rs.rand(d, N) < 0.03).astype(np.float32)             # each latent topic owns ~12 items
    topic = rs.randint(0, d, size=n)
    cond = np.eye(d, dtype=np.float32)[topic] + 0.05 * rs.randn(n, d).astype(np.float32)
    Y = sp.csr_matrix(((proto[topic] > 0) & (rs.rand(n, N) < 0.7)).astype(np.float32))

    class Vec(C.ConcatenationBasedConditioning):
        def fit(self, raw):
            return self

        def transform(self, raw):
            return np.asarray(raw, dtype=np.float32)

        def size_increment(self):
            return d

        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")

    class Set:
        def __init__(self, Y, c):
            self.Y, self.c = Y, c

        def tocsr(self):
            return self.Y

        def size(self, dim=0):
            return self.Y.shape[dim]

        def get_attributes(self, keys):
            return [self.c for _ in keys]
    torch.manual_seed(0)
    np.random.seed(0)
    rec = DecodingRecommender(C.ConditionList([("vec", Vec())]), n_epochs=30, batch_size=100, n_hidden=64, lr=0.01,
                              verbose=False)
    rec.train(Set(Y[:1000], cond[:1000]))
    pred = rec.predict(Set(Y[1000:], cond[1000:]))
    assert pred.shape == (200, N)
    assert METRICS["mrr@10"](Y[1000:].toarray(), pred)[0] > 0.5


def test_denoising_autoencoder_gauss_tracks_reference():
    """DenoisingAutoEncoder(corrupt='gauss') (dae.py:40-45, 191): the encoder reads the DENSE batch + N(0, noise_factor)
    on all N columns - a dense first layer, a dense weight gradient and an eager Adam over every row of enc.lin1 - while
    the BCE target stays the clean batch.  (a) the reference's recorded steps with the recorded noise and dropout masks;
    (b) production randomness: the model still learns, steps of both kinds of first layer mix (predict is sparse)."""
    from aaerec.dae import DenoisingAutoEncoder
    from aaerec.evaluation import remove_non_missing, METRICS
    fx = Fixture("step_dae_gauss")
    cfg = fx.cfg
    m = DenoisingAutoEncoder(n_hidden=cfg["h"], n_code=cfg["c"], lr=cfg["gen_lr"], batch_size=cfg["B"],
                             dropout=tuple(cfg["dropout"]), noise_factor=cfg["noise_factor"], corrupt="gauss", verbose=True,
                             rng_mode="reference")
    m._build(cfg["N"], 0)
    m.hip.load_params(fx.init_params())
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        masks = fx.masks(s)
        m._host_randomness = lambda B, masks=masks: (masks + [None] * 8, None)
        m.partial_fit(X, noise=fx.z[f"step{s}.noise"])
        np.testing.assert_allclose(m.last_losses[0], fx.z[f"step{s}.losses"][0], rtol=1e-5)
        got = m.hip.state_dict()
        for k, w in fx.expected_params(s).items():
            if not k.startswith("disc."):
                np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=f"dae gauss step {s} {k}")
        st = m.hip.adam_state("enc")
        em, ev, et = fx.expected_adam(s)[("A_enc", "enc.lin1.weight")]
        assert st["step"] == et
        np.testing.assert_allclose(st["lin1.weight"][0], em, atol=2e-9, rtol=1e-4)
        np.testing.assert_allclose(st["lin1.weight"][1], ev, atol=1e-12, rtol=2e-4)
    ip, idx, val = fx.batch(0, prefix="predict")
    Xp = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
    np.testing.assert_allclose(m.predict(Xp), fx.z["predict.out"], atol=1e-5)
    # (b)
    z, Xtr, Xin, Yout = _e2e()
    torch.manual_seed(3)
    np.random.seed(3)
    # (dense noise on 1 000 columns swamps a bag of 8 items after L1 normalisation unless it is small: 0.002 * 1 000
    # columns * 0.8 = 1.6 against 8; the plain autoencoder itself needs ~100 epochs on this corpus, see below)
    d = DenoisingAutoEncoder(n_hidden=50, n_code=50, n_epochs=100, batch_size=100, lr=0.01, noise_factor=0.002, corrupt="gauss",
                             verbose=False)
    d.fit(Xtr)
    pred = remove_non_missing(d.predict(Xin), Xin, copy=True)
    assert METRICS["mrr@10"](Yout.toarray(), pred)[0] > 0.02         # (a random ranking gives ~0.003)


def test_denoising_autoencoder_tracks_reference():
    """aaerec.dae.DenoisingAutoEncoder against the reference's dae.py: (a) recorded steps with the recorded
    corruption and dropout masks injected; (b) 3 epochs of fit() with rng_mode='reference' and the reference's
    seeds reproduce the reference's predictions (fixture e2e_dae_short.npz: corruption mask, shuffles, dropout
    draws all come off the same generators in the same order)."""
    from aaerec.dae import DenoisingAutoEncoder, DAERecommender
    fx = Fixture("step_dae")
    cfg = fx.cfg
    m = DenoisingAutoEncoder(n_hidden=cfg["h"], n_code=cfg["c"], lr=cfg["gen_lr"], batch_size=cfg["B"],
                             dropout=tuple(cfg["dropout"]), noise_factor=cfg["noise_factor"], verbose=True,
                             rng_mode="reference")
    m._build(cfg["N"], 0)
    m.hip.load_params(fx.init_params())
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        masks = fx.masks(s)
        m._host_randomness = lambda B, masks=masks: (masks + [None] * 8, None)
        m.partial_fit(X, keep=fx.z[f"step{s}.keep"])
        np.testing.assert_allclose(m.last_losses[0], fx.z[f"step{s}.losses"][0], rtol=1e-5)
        got = m.hip.state_dict()
        for k, w in fx.expected_params(s).items():
            if not k.startswith("disc."):
                np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=f"dae step {s} {k}")
    with pytest.raises(ValueError):
        m.partial_fit(X, y=1)
    with pytest.raises(KeyError):
        DenoisingAutoEncoder(corrupt="salt")
    # (b)
    z, Xtr, Xin, Yout = _e2e()
    want = np.load(os.path.join(GOLDEN, "e2e_dae_short.npz"))["pred_short"]
    torch.manual_seed(7)
    np.random.seed(7)
    d = DenoisingAutoEncoder(n_hidden=50, n_code=50, n_epochs=3, batch_size=100, lr=0.01, verbose=False,
                             rng_mode="reference")
    d.fit(Xtr)
    np.testing.assert_allclose(d.predict(Xin[:40]), want, atol=1e-4)
    # (c) production randomness through the recommender surface: it learns, and thins ~noise_factor of the entries
    from aaerec.evaluation import remove_non_missing, METRICS

    class Set:
        def __init__(self, X):
            self.X = X

        def tocsr(self):
            return self.X
    mrr = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        np.random.seed(seed)
        rec = DAERecommender(n_hidden=50, n_code=50, n_epochs=150, batch_size=100, lr=0.01, verbose=False)
        rec.train(Set(Xtr))
        pred = remove_non_missing(rec.predict(Set(Xin)), Xin, copy=True)
        mrr.append(METRICS["mrr@10"](Yout.toarray(), pred)[0])
    assert np.median(mrr) > 0.05, mrr
    from aaerec import _hip
    csr = _hip.DeviceCSR(Xtr, rec.dae.hip.device)
    kept = float((rec.dae._epoch_csr(csr).values != 0).float().mean())
    assert abs(kept - 0.8) < 0.02


@pytest.mark.parametrize("name", ["step_vae", "step_vae_cond", "step_vae_cat", "step_vae_cat:host"])
def test_vae_tracks_reference(name):
    """aaerec.vae.VAE against the reference's vae.py fixtures: recorded steps with the recorded eps (step_vae_cat: behind
    a trainable CategoricalCondition - embedding sum + SparseAdam, condition.py:397-508 - whose table the library's
    kernels encode and train around aae_vae_step)."""
    from aaerec.vae import VAE
    from aaerec import condition as C
    from test_parity_abi_gpu import _vae_params
    # ":host" = the same fixture with the condition's table in HOST memory (the reference's default placement): not a
    # block the kernels can produce, so the step is cut at the condition boundary and the plugin runs under torch autograd
    name, _, where = name.partition(":")
    fx = Fixture(name)
    cfg = fx.cfg
    conds = None
    if cfg["cond"] == "concat30":
        class ConstConcat(C.ConcatenationBasedConditioning):
            constant_concat = True

            def size_increment(self):
                return 30

            def encode(self, inputs):
                return torch.as_tensor(np.asarray(inputs), dtype=torch.float32, device="cuda")
        conds = C.ConditionList([("title", ConstConcat())])
    cat = None
    if cfg["cond"] == "cat":
        kind = cfg["cat"]
        on_gpu = where != "host"
        cat = C.CategoricalCondition(8, sparse=kind["sparse"], use_cuda=on_gpu, reduce=kind["reduce"], lr=kind["lr"])
        V = fx.z["init.cond.embedding"].shape[0]
        cat.vocab = {"a%d" % i: i for i in range(1, V)}               # indices are given pre-transformed
        cat.embedding = torch.nn.Embedding(V, 8, padding_idx=0, sparse=kind["sparse"])
        if on_gpu:
            cat.embedding = cat.embedding.cuda()
        with torch.no_grad():
            cat.embedding.weight.copy_(torch.from_numpy(fx.z["init.cond.embedding"]))
        cat.optimizer = torch.optim.SparseAdam(cat.embedding.parameters(), lr=kind["lr"])
        conds = C.ConditionList([("authors", cat)])

    def cin(s, prefix="step"):
        c = fx.cond_inputs(s, prefix=prefix) if prefix != "step" else fx.cond_inputs(s)
        if cat is not None:     # the fixture stores the batch-padded index lists: strip the padding again
            c = [[[int(j) for j in row if j != 0] or [0] for row in c[0]]]
        return c or None
    m = VAE(cfg["N"], cfg["N"], n_hidden=cfg["h"], n_code=cfg["c"], lr=cfg["gen_lr"], batch_size=cfg["B"],
            conditions=conds, verbose=True, rng_mode="reference")
    m.hip.load_params(_vae_params(fx, "init"))
    assert m._cond_native == (where != "host")
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        X = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
        eps = fx.z[f"step{s}.eps"]
        m._eps = lambda B, eps=eps: torch.from_numpy(eps)
        m.partial_fit(X, condition_data=cin(s))
        if cat is not None:
            np.testing.assert_allclose(cat.embedding.weight.detach().cpu().numpy(), fx.z[f"step{s}.cond.embedding"],
                                       atol=1e-5, err_msg=f"{name} step {s} embedding")
            st = cat.optimizer.state[cat.embedding.weight]
            assert float(st["step"]) == float(fx.z[f"step{s}.cond.t"])
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), fx.z[f"step{s}.cond.m"], atol=1e-8, rtol=1e-4)
        np.testing.assert_allclose(m.last_loss, fx.z[f"step{s}.losses"][0], rtol=2e-5)
        sd = m.state_dict()
        for n in ("fc1", "fc21", "fc22", "fc3", "fc4"):
            for t in ("weight", "bias"):
                np.testing.assert_allclose(sd[f"{n}.{t}"].numpy(), fx.z[f"step{s}.{n}.{t}"], atol=1e-5, rtol=0,
                                           err_msg=f"{name} step {s} {n}.{t}")
    ip, idx, val = fx.batch(0, prefix="predict")
    Xp = sp.csr_matrix((val, idx, ip), shape=(len(ip) - 1, cfg["N"]))
    m._eps = lambda B: torch.from_numpy(fx.z["predict.eps"])
    np.testing.assert_allclose(m.predict(Xp, condition_data=cin(0, prefix="predict")), fx.z["predict.out"], atol=1e-5)
    with pytest.raises(ValueError):
        m.partial_fit(X, y=1, condition_data=cin(0))
    with pytest.raises(NotImplementedError):
        VAE(cfg["N"], cfg["N"], final_activation="Tanh")


def test_vae_fit_with_reference_rng_and_recommender():
    """(a) 3 epochs of VAE.fit with rng_mode='reference' and the reference's seeds give the reference's trained
    encoder/decoder: fixture e2e_vae_short.npz holds mu, logvar and decode(mu) of 40 test rows computed by the
    reference from ITS weights; here they are computed from OUR weights.  (b) VAERecommender learns."""
    from aaerec.vae import VAE, VAERecommender
    from aaerec.evaluation import remove_non_missing, METRICS
    z, Xtr, Xin, Yout = _e2e()
    want = np.load(os.path.join(GOLDEN, "e2e_vae_short.npz"))
    N = Xtr.shape[1]
    torch.manual_seed(7)
    np.random.seed(7)
    v = VAE(N, N, n_hidden=50, n_code=50, n_epochs=3, batch_size=100, lr=0.01, verbose=False, rng_mode="reference")
    v.fit(Xtr)
    sd = {k: t.numpy().astype(np.float64) for k, t in v.state_dict().items()}
    x = Xin[:40].toarray().astype(np.float64)
    x /= np.maximum(np.abs(x).sum(1, keepdims=True), 1e-12)
    h1 = np.maximum(x @ sd["fc1.weight"].T + sd["fc1.bias"], 0)
    mu = h1 @ sd["fc21.weight"].T + sd["fc21.bias"]
    lv = h1 @ sd["fc22.weight"].T + sd["fc22.bias"]
    h3 = np.maximum(mu @ sd["fc3.weight"].T + sd["fc3.bias"], 0)
    rec = 1.0 / (1.0 + np.exp(-(h3 @ sd["fc4.weight"].T + sd["fc4.bias"])))
    np.testing.assert_allclose(mu, want["mu"], atol=2e-4)
    np.testing.assert_allclose(lv, want["logvar"], atol=2e-4)
    np.testing.assert_allclose(rec, want["recon_mu"], atol=1e-4)

    class Set:
        def __init__(self, X):
            self.X = X

        def tocsr(self):
            return self.X
    mrr = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        np.random.seed(seed)
        rec = VAERecommender(n_hidden=50, n_code=50, n_epochs=60, batch_size=100, lr=0.01, verbose=False)
        rec.train(Set(Xtr))
        pred = remove_non_missing(rec.predict(Set(Xin)), Xin, copy=True)
        mrr.append(METRICS["mrr@10"](Yout.toarray(), pred)[0])
    print("VAE MRR@10", mrr)
    # The reference's VAE adds a MEAN BCE to a SUMMED KL term (vae.py:133-145), so the KL term dominates and the
    # posterior collapses: on this corpus the reference itself reaches MRR@10 0.010 / 0.018 / 0.020 after 60 epochs
    # and 0.020 / 0.014 / 0.013 after 200 (seeds 1..3, measured with the reference in this repo's container;
    # chance is ~0.003).  Parity with that behaviour is the point, not a good recommender.
    assert 0.006 < np.median(mrr) < 0.1, mrr


def test_embedded_vectorizer_product_on_the_gpu_matches_reference_fixture():
    """EmbeddedVectorizer.transform (TF-IDF on the host, `aae_csr_embed` for the product) against the reference's
    outputs; then as the vectoriser of a PretrainedWordEmbeddingCondition built from gensim-like vectors."""
    import json
    from aaerec import condition as C
    from aaerec.ub import EmbeddedVectorizer
    z = np.load(os.path.join(GOLDEN, "embedded_vectorizer.npz"))
    words, docs, test = (json.loads(str(z[k])) for k in ("words", "docs", "test"))
    for tag in ("default", "sublinear"):
        v = EmbeddedVectorizer(z["embedding"], words, **json.loads(str(z[f"{tag}.kwargs"])))
        got = v.fit_transform(docs)
        assert got.dtype == np.float32 and got.shape == (len(docs), 300)
        np.testing.assert_allclose(got, z[f"{tag}.train"], atol=2e-6)
        np.testing.assert_allclose(v.transform(test), z[f"{tag}.test"], atol=2e-6)

    class KV:                      # gensim < 4 attribute names
        index2word, vectors = words, z["embedding"]
    cond = C.PretrainedWordEmbeddingCondition(KV(), use_cuda=True)
    data = C.ConditionList([("title", cond)]).fit_transform([docs])[0]
    np.testing.assert_allclose(data, z["default.train"], atol=2e-6)
    enc = cond.encode(data[:7])
    assert enc.is_cuda and enc.shape == (7, 300) and cond.size_increment() == 300


# ---------------------------------------------------------------------------------------------------------------
# fit() under data parallelism, two processes sharing the one GPU of the test box.  RCCL refuses two ranks on one
# device, so the ranks talk over gloo with the operands staged through the host; everything else - both models per
# rank, the kernels, aaerec.parallel - is what runs on a multi-GPU node.
# ---------------------------------------------------------------------------------------------------------------
def _dp_conditions(reduce):
    """A constant 6-wide block + a trainable CategoricalCondition (SparseAdam, table on the GPU) and their data."""
    from aaerec import condition as C

    class Const(C.ConcatenationBasedConditioning):
        constant_concat = True

        def size_increment(self):
            return 6

        def encode(self, inputs):
            return torch.as_tensor(np.asarray(inputs), dtype=torch.float32)

    rng = np.random.RandomState(11)
    vec = (rng.standard_normal((200, 6)) * 0.3).astype(np.float32)
    raw = [["a%d" % a for a in rng.randint(0, 30, size=rng.randint(1, 5))] for _ in range(200)]
    cat = C.CategoricalCondition(8, use_cuda=True, reduce=reduce, lr=0.01)
    cat.fit(raw)
    conds = C.ConditionList([("title", Const()), ("authors", cat)])
    return conds, [vec, cat.transform(raw)], cat


def _dp_model(adversarial, **kw):
    """AutoEncoder without dropout (nothing random but the shared shuffles), or the full adversarial model with
    dropout and a gauss prior in rng_mode='device' with an explicit seed: the device generator is keyed by the row of
    the global batch, so the ranks of a data-parallel run draw what the single process draws."""
    from aaerec.aae import AdversarialAutoEncoder, AutoEncoder
    if adversarial:
        return AdversarialAutoEncoder(n_hidden=48, n_code=16, gen_lr=0.01, reg_lr=0.005, batch_size=40, n_epochs=3,
                                      dropout=(0.2, 0.2), verbose=False, seed=4242, **kw)
    return AutoEncoder(n_hidden=48, n_code=16, lr=0.01, batch_size=40, n_epochs=3, dropout=(0.0, 0.0), verbose=False, **kw)


def _fit_worker(rank, world, port, mode, ret, reduce=None, adversarial=False):
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "aae-recommender_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aaerec.aae                               # (seeds torch at import, like the reference: import before seeding)
    from aaerec.parallel import HostStagedCollectives
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    conds, cdata, cat = _dp_conditions(reduce) if reduce else (None, None, None)
    m = _dp_model(adversarial, conditions=conds, data_parallel=HostStagedCollectives(dist), dp_mode=mode)   # batch_size = the GLOBAL batch
    m.fit(X, condition_data=cdata)
    pred = m.predict(X[:33], condition_data=[c[:33] for c in cdata] if cdata else None)
    if rank == 0:
        ret["state"] = m.hip.state_dict()
        ret["pred"] = pred
        ret["loss"] = m.last_losses[0]
        ret["sliced"] = m._slice is not None
        if cat is not None:
            ret["embedding"] = cat.embedding.weight.detach().cpu().numpy()
    flat = torch.from_numpy(pred.copy())
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(other[0], o) for o in other)          # every replica predicts the same
    dist.destroy_process_group()


def _dp_corpus():
    rng = np.random.RandomState(3)
    protos = [rng.choice(500, size=10, replace=False) for _ in range(12)]
    rows = [rng.choice(protos[rng.randint(12)], size=rng.randint(3, 9), replace=False) for _ in range(200)]
    ind0 = [b for b, r in enumerate(rows) for _ in r]
    return sp.coo_matrix((np.ones(len(ind0), dtype=np.float32), (ind0, np.concatenate(rows))), shape=(200, 500)).tocsr()


@pytest.mark.parametrize("adversarial", [False, True])
@pytest.mark.parametrize("mode", ["vocab", "vocab_out", "replicated", "shard"])
def test_fit_on_two_ranks_equals_single_process(mode, adversarial):
    """fit() on two ranks - batches of 40 documents, 20 per rank, in 'vocab' mode each rank owning 250 of the 500 items'
    rows of dec.lin3 AND columns of enc.lin1 ('vocab_out': of dec.lin3 only; 'shard': the same item slices, but ONE handle
    per rank that runs all 40 documents through its own copy of the hidden layers - three all-reduces of partial sums per
    step, no gradient exchange: aae_shard_step) - against one process: same parameters, same
    predictions.  Plain AutoEncoder without dropout, and the
    adversarial model with dropout + prior drawn by the device generator (keyed by global row, one seed)."""
    import torch.multiprocessing as mp
    import aaerec.aae                               # noqa: F401  (seeds torch at import: import before seeding below)
    with mp.get_context("spawn").Manager() as mgr:       # (a fork()ed manager process would inherit this process's GPU objects)
        ret = mgr.dict()
        spawn_ranks(_fit_worker, 2, lambda port: (2, port, mode, ret, None, adversarial))
        got = dict(ret)
    assert got["sliced"] == (mode != "replicated")
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    one = _dp_model(adversarial)
    one.fit(X)
    want = one.hip.state_dict()
    tol = 2e-4 if adversarial else 2e-5          # (15 adversarial steps with dropout amplify the summation-order noise)
    for k, w in want.items():
        d = np.abs(got["state"][k] - w)
        assert (d > tol).sum() <= max(8, 0.01 * d.size) and d.max() < 0.02, f"{mode} {k}: {(d > tol).sum()} off, max {d.max():.2e}"
    np.testing.assert_allclose(got["pred"], one.predict(X[:33]), atol=10 * tol)
    if mode != "replicated":     # (the replicated scheme reports each rank's loss over its own share)
        assert abs(got["loss"] - one.last_losses[0]) < 1e-5


@pytest.mark.parametrize("mode", ["vocab", "shard"])
@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_fit_on_two_ranks_with_trainable_categorical_condition(reduce, mode):
    """The vocabulary-sharded scheme with a constant block and a device-native CategoricalCondition: every rank
    gathers dL/d(condition block) of the whole batch and applies the identical SparseAdam update ('mean' pads each
    share to the whole batch's width, as the single process does)."""
    import torch.multiprocessing as mp
    from aaerec.aae import AutoEncoder
    with mp.get_context("spawn").Manager() as mgr:       # (a fork()ed manager process would inherit this process's GPU objects)
        ret = mgr.dict()
        spawn_ranks(_fit_worker, 2, lambda port: (2, port, mode, ret, reduce))
        got = dict(ret)
    assert got["sliced"]
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    conds, cdata, cat = _dp_conditions(reduce)
    one = AutoEncoder(n_hidden=48, n_code=16, lr=0.01, batch_size=40, n_epochs=3, dropout=(0.0, 0.0), verbose=False,
                      conditions=conds)
    one.fit(X, condition_data=cdata)
    assert one._is_device_native()
    for k, w in one.hip.state_dict().items():
        np.testing.assert_allclose(got["state"][k], w, atol=2e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(got["embedding"], cat.embedding.weight.detach().cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(got["pred"], one.predict(X[:33], condition_data=[c[:33] for c in cdata]), atol=2e-5)


def test_conditioned_recommender_through_bags_attributes():
    """The conditioned path of the reference's drivers end to end: Bags with owner attributes -> Evaluation.setup ->
    AAERecommender(conditions=ConditionList([...])).train / predict.  The 'venue' attribute fully determines a
    document's item prototype, so the model conditioned on it (a CategoricalCondition trained by the library's kernels)
    must rank the held-out item far better than the unconditioned one can."""
    from aaerec.aae import AAERecommender
    from aaerec import condition as C
    from aaerec.datasets import Bags
    from aaerec.evaluation import Evaluation
    rng = np.random.RandomState(1)
    protos = [rng.choice(400, size=6, replace=False) for _ in range(40)]
    data, owners, years, venue = [], [], {}, {}
    for i in range(800):
        k = rng.randint(40)
        data.append(["i%d" % t for t in rng.choice(protos[k], size=rng.randint(2, 4), replace=False)])
        owners.append("d%d" % i)
        years["d%d" % i] = 2000 + (i * 10) // 800
        venue["d%d" % i] = "v%d" % k
    bags = Bags(data, owners, {"year": years, "venue": venue})
    ev = Evaluation(bags, 2009, metrics=["mrr@10"], logfile=None).setup(min_elements=2, drop=1)
    torch.manual_seed(0)
    np.random.seed(0)
    cat = C.CategoricalCondition(16, use_cuda=True, lr=0.01)                 # one venue per document (reduce=None)
    rec = AAERecommender(conditions=C.ConditionList([("venue", cat)]), n_hidden=40, n_code=16, n_epochs=40, batch_size=50,
                         gen_lr=0.01, verbose=False)
    plain = AAERecommender(n_hidden=40, n_code=16, n_epochs=40, batch_size=50, gen_lr=0.01, verbose=False)
    res = ev([rec, plain])
    mrr_cond, mrr_plain = res[0][0][0], res[1][0][0]
    assert rec.model._is_device_native() and cat.embedding.weight.is_cuda
    assert float(cat.optimizer.state[cat.embedding.weight]["step"]) > 100          # trained by aae_cat_update
    assert mrr_cond > 0.3 and mrr_cond > mrr_plain + 0.05, (mrr_cond, mrr_plain)


@pytest.mark.parametrize("dp", ["shard", "vocab", "replicated"])
def test_bench_spawns_its_own_ranks(dp):
    """`python bench.py --gpus 2` the way the driver starts it - no launcher: bench.py spawns one process per rank before
    touching the GPU and relays rank 0's single JSON line.  On this one-GPU box the ranks share device 0 and their
    collectives go over gloo staged through the host (AAE_BENCH_GLOO_ONE_GPU=1): the numbers mean nothing, the protocol
    (rendezvous, rank-0 broadcast of weights / seed / permutation, both exchange schemes, one JSON line, exit code 0) does."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AAE_BENCH_GLOO_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu", "--items", "6000", "--hidden", "64", "--dp", dp], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["global_batch"] == 200 and out["value"] > 0
    assert np.all(np.isfinite(out["losses_last_step"]))
    if dp == "shard":            # the default scheme: three all-reduces per step, the library's step driver
        assert out["collectives_per_step"]["collectives"] == 3 and "3 all-reduces" in out["config"]["parallelism"]


def test_dense_batches_are_compacted_on_the_device():
    """partial_fit / predict on the DENSE batch the reference passes (X_shuf[start:end].toarray(), aae.py:823, 848-853),
    float64 as toarray() gives it and float32: the same step as the CSR call (the device-side compaction gives scipy's CSR
    bit for bit: ascending columns, float32 values); targets outside [0, 1] raise as the reference's BCE does."""
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec._hip import DeviceCSR
    rng = np.random.default_rng(3)
    N, B = 700, 37
    X = sp.random(3 * B, N, density=0.02, random_state=5, format="csr", dtype=np.float32)
    X.data[:] = 1.0
    X.data[::5] = 0.25
    # the compaction itself, against scipy
    d = DeviceCSR.from_dense(X[:B].toarray().astype(np.float64), "cuda", 10000)
    ref = X[:B].tocsr()
    assert d.nnz == ref.nnz and d.nnz_per_row_max == int(np.diff(ref.indptr).max())
    np.testing.assert_array_equal(d.indptr.cpu().numpy(), ref.indptr)
    np.testing.assert_array_equal(d.indices[:ref.nnz].cpu().numpy(), ref.indices)
    np.testing.assert_array_equal(d.values[:ref.nnz].cpu().numpy(), ref.data)
    models = []
    for form in ("csr", "dense64", "dense32"):
        torch.manual_seed(11)
        m = AdversarialAutoEncoder(n_hidden=40, n_code=12, batch_size=B, verbose=False, rng_mode="device", seed=5)
        for s in range(3):
            Xb = X[s * B:(s + 1) * B]
            m.partial_fit(Xb if form == "csr" else Xb.toarray().astype(np.float64 if form == "dense64" else np.float32))
        models.append(m)
    sd0 = models[0].hip.state_dict()
    for m in models[1:]:
        for k, v in m.hip.state_dict().items():
            np.testing.assert_allclose(v, sd0[k], atol=1e-6, rtol=0, err_msg=k)    # (float atomics in the first layer's scatter: no two runs are bitwise equal)
    p0 = models[0].predict(X[:50])
    np.testing.assert_allclose(models[1].predict(X[:50].toarray()), p0, atol=1e-6, rtol=0)
    bad = X[:B].toarray()
    bad[3, 5] = 2.0
    with pytest.raises(RuntimeError, match="between 0 and 1"):
        models[0].partial_fit(bad)


# ---- MRR@10 parity on a corpus big enough for the north star's +-0.001 to mean something (tests/golden/e2e_c1_big.npz) ----
def _big():
    z = np.load(os.path.join(GOLDEN, "e2e_c1_big.npz"))
    N = int(z["N"])

    def csr(p):
        ip, idx = z[p + "_indptr"], z[p + "_indices"]
        return sp.csr_matrix((np.ones(len(idx), dtype=np.float32), idx, ip), shape=(len(ip) - 1, N))
    return z, csr("train"), csr("in"), csr("out")


def _mrr10(pred, Xin, Yout):
    from aaerec.evaluation import remove_non_missing, METRICS
    return METRICS["mrr@10"](Yout.toarray(), remove_non_missing(pred, Xin, copy=True))[0]


def _big_model(n_epochs, rng_mode, **kw):
    from aaerec.aae import AdversarialAutoEncoder
    return AdversarialAutoEncoder(n_hidden=50, n_code=50, n_epochs=n_epochs, batch_size=100, gen_lr=0.01, reg_lr=0.001,
                                  dropout=(0., 0.), verbose=False, rng_mode=rng_mode, **kw)


def test_ranking_metrics_identical_at_the_short_horizon():
    """North star: "reconstructions match the reference within 1e-4 fp32 (ranking metrics identical)".  Three epochs (120
    steps) of fit() with the reference's random draws replayed: the predictions for 200 test docs agree to 1e-4, and
    MRR@10 / MAP@10 / P@5 computed from them are IDENTICAL to the ones computed from the reference's own predictions
    (every top-10 list ranks its relevant item the same).
    The path is the production path: since r3 the sparse first layer's weight gradient is a per-item sum in row order
    (csrc/w1_update.h), no float atomics anywhere in the parameter path, so two runs of the same seeds agree bit for bit -
    asserted below.  (r1/r2 scattered with float atomics: a swapped pair of adds moved one weight by an ulp, and 40-120
    adversarial steps later that was 4e-5 (5 % of runs) or 1.09e-4 (3 %) in the predictions.)"""
    from aaerec.evaluation import remove_non_missing, METRICS
    import aaerec.aae  # noqa: F401  (its import seeds torch, as the reference's does, aae.py:27: import BEFORE seeding)
    z, Xtr, Xin, Yout = _big()
    seed = int(z["short_seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = _big_model(3, "reference")
    m.fit(Xtr)
    n = z["pred_short"].shape[0]
    pred = m.predict(Xin[:n])
    np.testing.assert_allclose(pred, z["pred_short"], atol=1e-4)
    # the same run again: bit for bit the same predictions (no scheduling-dependent summation order in the parameter path)
    torch.manual_seed(seed)
    np.random.seed(seed)
    m2 = _big_model(3, "reference")
    m2.fit(Xtr)
    assert np.array_equal(m2.predict(Xin[:n]), pred)
    Y = Yout[:n].toarray()
    ours, ref = remove_non_missing(pred, Xin[:n], copy=True), remove_non_missing(z["pred_short"], Xin[:n], copy=True)
    for name in ("mrr@10", "map@10", "p@5"):
        a, b = METRICS[name](Y, ours), METRICS[name](Y, ref)
        assert a[0] == b[0] and a[1] == b[1], (name, a, b)
    assert METRICS["mrr@10"](Y, ours)[0] > 0.01            # (not a degenerate 0 == 0; a random ranking gives ~0.003)


def test_mrr_parity_at_10k_test_docs():
    """MRR@10 on 10 000 test docs (sampling s.e. 0.003) after the full 120-epoch recipe (4 800 steps), against the
    reference's 16 runs stored in the fixture (mean 0.5226, seed-to-seed s.d. 0.023 - the REFERENCE's own noise, so a
    +-0.001 band between independent random streams is not a testable statement; what is:)
      (a) the reference's draws replayed (rng_mode='reference'): the runs share every draw for 4 800 steps, yet fp32
          rounding differences decorrelate the trajectories long before the end (observed, r2: seed 0 0.506 vs 0.535,
          seed 3 0.525 vs 0.521, seed 10 0.548 vs 0.447) - per-seed equality is exactly what the short-horizon test above
          checks, here each replayed run only has to land inside the reference's own spread;
      (b) the production device generator, 16 seeds: the mean over the converged runs lies within 2.5 standard errors
          of the reference's mean (s.e. of the difference ~0.007), the medians within 0.015;
      (c) bf16 mode (config C2's arithmetic), 8 seeds: the same bands;
    Every run of a seed is the same run (the parameter path has no float atomics since r3: host seeds 0 / 3 / 10 replay to
    0.5400 / 0.5144 / 0.5431 each time); r2's atomic scatter let the fragile seed 10 land between 0.35 and 0.55."""
    import aaerec.aae  # noqa: F401  (import before seeding: the module seeds torch at import, as the reference's does)
    z, Xtr, Xin, Yout = _big()
    ref = z["ref_mrr10"]

    def run(host_seed, rng_mode, **kw):
        torch.manual_seed(host_seed)
        np.random.seed(host_seed)
        m = _big_model(120, rng_mode, **kw)
        m.fit(Xtr)
        return _mrr10(m.predict(Xin), Xin, Yout)
    same = {s: run(s, "reference") for s in (0, 3, 10)}
    print("MRR@10, reference draws replayed:", {s: (round(v, 4), round(float(ref[s]), 4)) for s, v in same.items()})
    sd = float(ref.std(ddof=1))
    for s, v in same.items():
        # inside the reference's spread - or next to the reference's OWN outcome for this seed where that one is itself in
        # the tail (seed 10: the reference's lowest run, 0.447; replays of it ended at 0.548 and at 0.426 in r2, when the
        # first-layer scatter's float atomics made two runs of this library differ in the last bits)
        assert abs(v - ref.mean()) < 4 * sd or abs(v - float(ref[s])) < 2 * sd, (s, v, float(ref[s]), float(ref.mean()), sd)
    # The recipe has a failure mode: for some initialisations the adversarial game wrecks the autoencoder for most prior
    # streams (host seed 15: the REFERENCE itself drops to MRR@10 0.09 when its z_real draws come from another generator,
    # tools/debug notes in DESIGN.md; its own 16 runs happened to avoid it, min 0.447).  Means are therefore compared
    # over the converged runs and the medians over all of them.
    def summary(tag, vals):
        vals = np.asarray(vals)
        ok = vals[vals > 0.4]
        se = np.sqrt(ref.var(ddof=1) / len(ref) + ok.var(ddof=1) / len(ok))
        print(f"MRR@10, {tag}:", np.round(vals, 4).tolist(), "| converged", len(ok), "of", len(vals), "mean", round(float(ok.mean()), 4),
              "median", round(float(np.median(vals)), 4), "| reference mean", round(float(ref.mean()), 4), "median",
              round(float(np.median(ref)), 4), "| s.e. of the mean difference", round(float(se), 4))
        assert len(ok) >= len(vals) - 2, vals                       # at most 2 collapsed runs in the sample
        assert abs(ok.mean() - ref.mean()) < 2.5 * se, (ok.mean(), ref.mean(), se)
        assert abs(np.median(vals) - np.median(ref)) < 0.015, (np.median(vals), np.median(ref))
    summary("device generator", [run(s, "device", seed=1000 + s) for s in range(16)])
    summary("bf16 mode", [run(s, "device", seed=2000 + s, dtype="bf16") for s in range(8)])


def test_c3_scale_ranking_matches_reference():
    """The ranking check at config C3's SHAPE (|items| = 100 000, hidden 200, code 50, batch 100: 3 125 item tiles in the
    output layer, 13-block layer chains - the layer sizes bench.py times), VERDICT r2 item 7.  tests/golden/e2e_c3.npz
    (tools/gen_golden.py e2e_c3) holds one REFERENCE run on a prototype-structured corpus - 2 000 training docs, 3 epochs
    = 60 partial_fit steps at the reference's default learning rates without dropout, seed 11 - reduced to what its
    evaluation makes of the [200, 100 000] prediction matrix: evaluation.remove_non_missing + argtopk (evaluation.py:183-199,
    20-58) -> the 12 best items per test doc with their scaled scores, MRR@10 / MAP@10 / P@5, plus raw sigmoid outputs at
    32 probe items per doc.
    Here: fit() with the reference's draws replayed (rng_mode='reference'), then
      * predict() at the probe items within 1e-6 absolute / 1e-4 relative (north star: 1e-4 absolute; observed 5e-8 / 2e-6);
      * predict_topk() - remove_non_missing + top-k on the device, only [200, 12] leaves the GPU - names the SAME items in
        the SAME order as the reference wherever the reference's own scaled scores separate them by more than 2e-5 (20x
        the observed difference of the scaled scores), and the same SET of ten otherwise; scaled scores within 2e-5;
      * MRR@10, MAP@10 and P@5 computed from the device's top-k equal the reference's."""
    from aaerec.evaluation import evaluate_topk
    import aaerec.aae  # noqa: F401  (import before seeding: the module seeds torch at import, as the reference's does)
    from aaerec.aae import AdversarialAutoEncoder
    import json
    z = np.load(os.path.join(GOLDEN, "e2e_c3.npz"))
    N, seed = int(z["N"]), int(z["seed"])

    def csr(indptr, indices):
        return sp.csr_matrix((np.ones(len(indices), dtype=np.float32), indices, indptr), shape=(len(indptr) - 1, N))
    Xtr, Xin, Yout = csr(z["train_indptr"], z["train_indices"]), csr(z["in_indptr"], z["in_indices"]), csr(z["out_indptr"], z["out_indices"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    recipe = json.loads(str(z["recipe"]))
    m = AdversarialAutoEncoder(n_hidden=200, n_code=50, n_epochs=int(z["n_epochs"]), batch_size=100, gen_lr=recipe["gen_lr"],
                               reg_lr=recipe["reg_lr"], dropout=(0., 0.), verbose=False, rng_mode="reference")
    m.fit(Xtr)
    # raw reconstructions at the probe items
    pred = m.predict(Xin)
    got = np.take_along_axis(pred, z["probe"].astype(np.int64), axis=1)
    want = z["probe_raw"]
    print("probe items: max |diff|", float(np.abs(got - want).max()), "max reference score", float(want.max()))
    np.testing.assert_allclose(got, want, atol=1e-6)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(pred.max(1), z["row_max"], atol=1e-6)
    # on-device remove_non_missing + top-k
    idx, val = m.predict_topk(Xin, k=12)
    idx, val = np.asarray(idx), np.asarray(val)
    ref_idx, ref_val = z["top_idx"], z["top_val"]
    np.testing.assert_allclose(val, ref_val, atol=2e-5)
    gaps = ref_val[:, :-1] - ref_val[:, 1:]                # separation of rank r from rank r + 1 in the reference
    same_order = 0
    for r in range(idx.shape[0]):
        clear = gaps[r, :10].min() > 2e-5                  # ranks 1..11 all separated
        if clear:
            assert idx[r, :10].tolist() == ref_idx[r, :10].tolist(), (r, idx[r, :10], ref_idx[r, :10], ref_val[r])
            same_order += 1
        elif gaps[r, 9] > 2e-5:                            # near-ties inside the top ten only: the same ten items
            assert set(idx[r, :10].tolist()) == set(ref_idx[r, :10].tolist()), (r, idx[r, :10], ref_idx[r, :10])
    print("identical top-10 lists:", same_order, "of", idx.shape[0])
    assert same_order >= 0.9 * idx.shape[0]
    ref_metrics = json.loads(str(z["metrics"]))
    names = ("mrr@10", "map@10", "p@5")
    ours = evaluate_topk(Yout, idx[:, :10], list(names))
    for name, (mean, std) in zip(names, ours):
        assert abs(mean - ref_metrics[name][0]) < 1e-12 and abs(std - ref_metrics[name][1]) < 1e-9, (name, mean, std, ref_metrics[name])


def _rccl_world1_worker(rank, port, ret):
    """One rank over REAL RCCL (backend nccl, world 1): fit() in dp_mode='vocab' takes the native step driver - aae_dp_step
    with the collectives table aae_rccl_init builds on a communicator the library creates."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "aae-recommender_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import aaerec.aae  # noqa: F401
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1, device_id=torch.device("cuda", 0))
    np.random.seed(5)
    torch.manual_seed(5)
    m = _dp_model(True, data_parallel=True, dp_mode="vocab")
    m.fit(_dp_corpus())
    from aaerec.parallel import RcclTable
    ret["native"] = m._dp._native is not None and isinstance(m._dp._native_keep, RcclTable)   # (the RCCL table, not Python callbacks)
    ret["state"] = m.hip.state_dict()
    ret["pred"] = m.predict(_dp_corpus()[:33])
    ret["stats"] = m._dp.comm_stats()
    # SURVEY 8e's bar: 3 steps, 1e-5 (default learning rates; what differs from the single process is summation order only)
    np.random.seed(6)
    torch.manual_seed(6)
    m3 = _three_step_model(data_parallel=True, dp_mode="vocab")
    for _ in zip(range(3), m3.fit_steps(_dp_corpus())):
        pass
    m3._fit_finish()
    ret["state3"] = m3.hip.state_dict()
    ret["losses3"] = tuple(float(x) for x in m3.last_losses)
    # the third scheme (dp_mode='shard': aae_shard_step, three all-reduces) over the library's RCCL communicator
    np.random.seed(6)
    torch.manual_seed(6)
    ms = _three_step_model(data_parallel=True, dp_mode="shard")
    for _ in zip(range(3), ms.fit_steps(_dp_corpus())):
        pass
    ms._fit_finish()
    ret["shard_native"] = isinstance(ms._dp._native_keep, RcclTable)
    ret["shard_state3"] = ms.hip.state_dict()
    ret["shard_losses3"] = tuple(float(x) for x in ms.last_losses)
    ret["shard_stats"] = ms._dp.comm_stats()
    m._dp._native_keep.close()                      # aae_rccl_destroy (the communicator was never handed back)
    m3._dp._native_keep.close()
    ms._dp._native_keep.close()
    dist.destroy_process_group()


def _three_step_model(**kw):
    from aaerec.aae import AdversarialAutoEncoder
    return AdversarialAutoEncoder(n_hidden=48, n_code=16, batch_size=40, n_epochs=1, dropout=(0.2, 0.2), verbose=False,
                                  seed=4243, **kw)


def test_native_step_driver_over_rccl_on_one_rank():
    """aae_dp_step over real RCCL collectives (world 1: the only multi-rank-capable transport a one-GPU box can run): the
    whole both-sharded step - 7 collectives on a library-owned communicator and every kernel - is ONE library call per
    partial_fit; parameters and predictions equal the plain single-process fit()."""
    import torch.multiprocessing as mp
    import aaerec.aae                               # noqa: F401
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_rccl_world1_worker, 1, lambda port: (port, ret))
        got = dict(ret)
    assert got["native"]
    assert got["stats"]["collectives"] == 7
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    one = _dp_model(True)
    one.fit(X)
    for k, w in one.hip.state_dict().items():
        d = np.abs(got["state"][k] - w)
        assert (d > 2e-4).sum() <= max(8, 0.01 * d.size) and d.max() < 0.02, f"{k}: {(d > 2e-4).sum()} off, max {d.max():.2e}"
    np.testing.assert_allclose(got["pred"], one.predict(X[:33]), atol=2e-3)
    # (above: 15 steps at gen_lr 0.01 with dropout - adversarial steps amplify an ulp of summation order, hence the wide
    #  bound.)  The bar SURVEY 8e names - 3 steps within 1e-5 - at the reference's default learning rates:
    np.random.seed(6)
    torch.manual_seed(6)
    one3 = _three_step_model()
    for _ in zip(range(3), one3.fit_steps(X)):
        pass
    one3._fit_finish()
    np.testing.assert_allclose(got["losses3"], one3.last_losses, rtol=1e-5, atol=1e-6)
    for k, w in one3.hip.state_dict().items():
        np.testing.assert_allclose(got["state3"][k], w, atol=1e-5, err_msg=k)
    # ... and the same bar for dp_mode='shard' (3 all-reduces per step on the library's communicator)
    assert got["shard_native"] and got["shard_stats"]["collectives"] == 3
    np.testing.assert_allclose(got["shard_losses3"], one3.last_losses, rtol=1e-5, atol=1e-6)
    for k, w in one3.hip.state_dict().items():
        np.testing.assert_allclose(got["shard_state3"][k], w, atol=1e-5, err_msg="shard " + k)


def test_fit_on_two_ranks_through_the_python_step_driver(monkeypatch):
    """The phase-by-phase Python driver of the both-sharded scheme (VocabParallelAAE._step_both_sharded: what the CPU
    stand-ins of tests/test_parallel_gloo.py run, and what AAE_DP_PYTHON=1 selects on the GPU) on two ranks sharing the
    GPU: same result as the default - the native aae_dp_step - checks against (test_fit_on_two_ranks_equals_...)."""
    monkeypatch.setenv("AAE_DP_PYTHON", "1")
    test_fit_on_two_ranks_equals_single_process("vocab", True)


def _ipc_worker(rank, world, port, ret):
    import ctypes as C
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "aae-recommender_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec._hip import HipAAE
    from aaerec.parallel import ipc_collectives
    m = HipAAE(300, 20, 10, max_batch=16)                  # (the library handle, device and stream the table is built on)
    cap = 800 * 204
    owner = ipc_collectives(m, dist, cap)
    assert owner is not None, "the mailboxes could not be shared between the two processes of this GPU"
    tab = owner.table
    gen = torch.Generator().manual_seed(100 + rank)
    ok, worst = True, 0.0
    for k, n in enumerate([4, 204 * 100, cap, 204 * 37, cap, 1024, 204 * 800, 8]):     # slots alternate by call; sizes up to the capacity
        x = (torch.rand(n, generator=gen) * 2 - 1) * 10.0 ** (k % 3 - 1)
        every = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(every, x)
        want = every[0].clone()
        for r in range(1, world):
            want += every[r]                                # rank order: what the kernel adds, on every rank
        buf = x.to(m.device)
        rc = tab.all_reduce(tab.ctx, C.c_void_p(buf.data_ptr()), n, C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream))
        assert rc == 0, m.lib.aae_last_error()
        got = buf.cpu()
        ok = ok and bool(torch.equal(got, want))
        worst = max(worst, float((got - want).abs().max()))
    ret[f"ok{rank}"], ret[f"worst{rank}"] = ok, worst
    # an operand beyond the capacity / a collective the table does not carry: refused, not truncated
    big = torch.zeros(cap + 4, device=m.device)
    ret[f"refused{rank}"] = (tab.all_reduce(tab.ctx, C.c_void_p(big.data_ptr()), cap + 4, None) != 0,
                             tab.all_gather(tab.ctx, C.c_void_p(big.data_ptr()), C.c_void_p(big.data_ptr()), 4, None) != 0)
    owner.close()
    dist.destroy_process_group()


def test_one_shot_all_reduce_over_ipc_mailboxes_equals_the_rank_ordered_sum():
    """r6: aae_ipc_* - the all-reduce of dp_mode='shard' as ONE launch over mailboxes every rank maps
    (hipIpc), summed in rank order.  Two processes sharing this box's one GPU; eight calls of different sizes (the two slots
    of a mailbox alternate) against the sum of the gloo-gathered operands in rank order: BITWISE equal on both ranks."""
    import torch.multiprocessing as mp
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_ipc_worker, 2, lambda port: (2, port, ret))
        got = dict(ret)
    assert got["ok0"] and got["ok1"], (got["worst0"], got["worst1"])
    assert got["refused0"] == (True, True) and got["refused1"] == (True, True)


def _fit_ipc_worker(rank, world, port, ret):
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "aae-recommender_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aaerec.aae                               # noqa: F401
    from aaerec.parallel import HostStagedCollectives, IpcTable
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    m = _dp_model(True, data_parallel=HostStagedCollectives(dist), dp_mode="shard")
    m.dp_collectives = "ipc"
    m.fit(X)
    pred = m.predict(X[:33])
    if rank == 0:
        ret["state"], ret["pred"], ret["loss"] = m.hip.state_dict(), pred, m.last_losses[0]
    ret[f"ipc{rank}"] = isinstance(m._dp._native_keep, IpcTable)
    m._dp._native_keep.close()
    dist.destroy_process_group()


def test_fit_on_two_ranks_through_the_ipc_all_reduce_equals_single_process():
    """dp_mode='shard' with its three all-reduces per step as one-shot launches over the peers' mailboxes (dp_collectives='ipc')
    instead of the host-staged gloo ones: the adversarial model, 15 steps on two ranks sharing the GPU, against one process -
    the bounds of test_fit_on_two_ranks_equals_single_process."""
    import torch.multiprocessing as mp
    import aaerec.aae                               # noqa: F401
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_fit_ipc_worker, 2, lambda port: (2, port, ret))
        got = dict(ret)
    assert got["ipc0"] and got["ipc1"]
    X = _dp_corpus()
    np.random.seed(5)
    torch.manual_seed(5)
    one = _dp_model(True)
    one.fit(X)
    want = one.hip.state_dict()
    tol = 2e-4
    for k, w in want.items():
        d = np.abs(got["state"][k] - w)
        assert (d > tol).sum() <= max(8, 0.01 * d.size) and d.max() < 0.02, f"{k}: {(d > tol).sum()} off, max {d.max():.2e}"
    np.testing.assert_allclose(got["pred"], one.predict(X[:33]), atol=10 * tol)
    assert abs(got["loss"] - one.last_losses[0]) < 1e-5
