"""CPU tests of the host-side mirror: data model, metrics (known answers copied from the
reference's doctests / produced by the reference's functions, tests/golden/metrics.npz),
condition plugins, and the C-ABI library's exported symbols.  No GPU, no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---- C ABI -------------------------------------------------------------------------------------
def test_library_exports_every_symbol_of_the_header():
    from aaerec import _build, _hip
    _build.build()                       # no-op when up to date; hipcc cross-compiles without a GPU
    lib = ctypes.CDLL(_hip.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "aaerec_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(aae_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_hip._PROTOS), "ctypes prototypes out of sync with the header"
    assert _hip.load_library().aae_abi_version() == _hip.ABI_VERSION


def test_arena_size_and_validation_without_gpu():
    from aaerec import _hip
    lib = _hip.load_library()
    cfg = _hip.AaeConfig()
    cfg.abi_version = _hip.ABI_VERSION
    cfg.n_items, cfg.n_hidden, cfg.n_code, cfg.max_batch, cfg.max_nnz = 100000, 200, 50, 100, 25000
    n = ctypes.c_size_t()
    assert lib.aae_arena_bytes(ctypes.byref(cfg), ctypes.byref(n)) == 0
    # 2 item-sized layers x (param + 2 Adam x {1,2}) + gradient + [B,N] logits gradient, ~0.85 GB
    assert 7e8 < n.value < 1.2e9
    cfg.n_hidden = 0
    assert lib.aae_arena_bytes(ctypes.byref(cfg), ctypes.byref(n)) == -1
    assert b"positive" in lib.aae_last_error()


def test_options_are_one_table_read_per_handle_and_set_through_the_abi(monkeypatch):
    """r6: every switch of the library is a field of aae_options (csrc/abi_model.h), read ONCE per handle in aae_create /
    aae_arena_bytes - a value set with aae_set_option wins over the environment variable AAE_<NAME>, NULL hands the name back,
    an unknown name is refused.  Observable without a GPU: X16_ROWS decides whether the arena carries the wide-batch chain
    kernel's split weight copies (aae_arena_bytes)."""
    from aaerec import _hip
    lib = _hip.load_library()
    cfg = _hip.AaeConfig()
    cfg.abi_version = _hip.ABI_VERSION
    cfg.n_items, cfg.n_hidden, cfg.n_code, cfg.max_batch, cfg.max_nnz = 5000, 200, 50, 100, 25000

    def arena():
        n = ctypes.c_size_t()
        assert lib.aae_arena_bytes(ctypes.byref(cfg), ctypes.byref(n)) == 0
        return n.value
    monkeypatch.delenv("AAE_X16_ROWS", raising=False)
    base = arena()                                         # 200 stacked rows < 1024: no split copies
    assert lib.aae_set_option(b"X16_ROWS", b"1") == 0
    with_copies = arena()
    assert with_copies > base + 6 * 200 * 200 * 2          # six layers' three-plane copies, forward and dX
    monkeypatch.setenv("AAE_X16_ROWS", "100000")           # (the table wins over the environment ...)
    assert arena() == with_copies
    assert lib.aae_set_option(b"X16_ROWS", None) == 0      # (... until the name is handed back)
    assert arena() == base
    monkeypatch.setenv("AAE_X16_ROWS", "1")
    assert arena() == with_copies
    assert lib.aae_set_option(b"NO_SUCH_SWITCH", b"1") == -1 and b"unknown option" in lib.aae_last_error()
    assert lib.aae_set_option(None, b"1") == -1


def test_no_cpu_fallback():
    from aaerec import _hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.AaeHipError):
        _hip.HipAAE(100, 8, 4)


# ---- metrics -------------------------------------------------------------------------------------
def test_metric_known_answers_from_reference_doctests():
    from aaerec.evaluation import MRR, MAP, P, argtopk, remove_non_missing
    yt = np.array([[1, 0, 0], [0, 0, 1]])
    yp = np.array([[0.2, 0.3, 0.1], [0.2, 0.5, 0.7]])
    assert MRR(2)(yt, yp) == (0.75, 0.25)                       # evaluation.py:100-103
    assert MAP(2)(yt, yp) == (0.75, 0.25)                       # evaluation.py:125-128
    assert MRR(3)(np.array([[1, 0, 1], [1, 0, 1]]), np.array([[0.4, 0.3, 0.2], [0.4, 0.3, 0.2]])) == (1.0, 0.0)
    m, s = MAP(3)(np.array([[1, 0, 1], [1, 1, 1]]), np.array([[0.4, 0.3, 0.2], [0.4, 0.3, 0.2]]))
    assert abs(m - 0.9166666666666666) < 1e-12 and abs(s - 0.08333333333333337) < 1e-12
    yt4, yp4 = np.array([[1, 0, 1, 0], [1, 0, 1, 0]]), np.array([[0.2, 0.3, 0.1, 0.05], [0.2, 0.5, 0.7, 0.05]])
    assert P(2)(yt4, yp4) == (0.5, 0.0) and P(4)(yt4, yp4) == (0.5, 0.0)       # evaluation.py:151-156
    X = np.arange(20).reshape(2, 10)
    assert X[argtopk(X, 3)].tolist() == [[9, 8, 7], [19, 18, 17]]           # evaluation.py:30-40
    got = remove_non_missing(np.array([[0.6, 0.5, -1], [40, -20, 10]]), np.array([[1, 0, 1], [0, 1, 0]]))
    np.testing.assert_allclose(got, [[0, 0.9375, 0], [1, 0, 0.5]])          # evaluation.py:187-191
    from aaerec.rank_metrics_with_std import mean_reciprocal_rank
    assert abs(mean_reciprocal_rank([[0, 0, 1], [0, 1, 0], [1, 0, 0]])[0] - 0.6111111111111111) < 1e-12


def test_metrics_match_reference_outputs():
    from aaerec.evaluation import METRICS, remove_non_missing, evaluate
    z = np.load(os.path.join(GOLDEN, "metrics.npz"))
    removed = remove_non_missing(z["y_pred"], sp.csr_matrix(z["x_test"]), copy=True)
    np.testing.assert_allclose(removed, z["removed"], atol=1e-15)
    for key in z.files:
        if key.startswith("metric."):
            mean, std = METRICS[key[len("metric."):]](z["y_true"], removed)
            np.testing.assert_allclose([mean, std], z[key], atol=1e-12, err_msg=key)
    # batched == unbatched (reference tests/test_evaluation.py:8-24)
    names = ["mrr", "map", "P@1", "p@5"]
    a = evaluate(z["y_true"], removed, names)
    b = evaluate(sp.csr_matrix(z["y_true"]), removed, names, batch_size=7)
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), atol=1e-12)


# ---- data model ----------------------------------------------------------------------------------
def _toy_bags():
    from aaerec.datasets import Bags
    data = [["a", "b", "c"], ["b", "c"], ["a", "d", "b"], ["c"], ["a", "b"]]
    owners = ["o%d" % i for i in range(5)]
    attrs = {"year": {o: 2000 + i for i, o in enumerate(owners)}, "title": {o: "t " + o for o in owners}}
    return Bags(data, owners, attrs)


def test_bags_vocab_split_tocsr():
    bags = _toy_bags()
    assert len(bags) == 5 and bags.numel() == 11 and bags.maxlen() == 3
    train, test = bags.train_test_split(on_year=2003)
    assert len(train) == 3 and len(test) == 2
    train = train.build_vocab(min_count=None, max_features=None, apply=True)
    assert train.vocab["b"] == 0                      # most frequent token first
    test = test.apply_vocab(train.vocab)
    X = train.tocsr()
    assert X.shape == (3, 4) and X.dtype == np.float64 and X.sum() == 8
    assert test.data == [[train.vocab["c"]], [train.vocab["a"], train.vocab["b"]]]
    assert train.get_attributes(["title"]) == [["t o0", "t o1", "t o2"]]
    clone = train.clone()
    clone.data[0].append(0)
    assert len(train.data[0]) == 3
    with pytest.raises(ValueError):
        train.build_vocab()
    train.prune_(min_elements=3)
    assert len(train) == 2 and len(train.bag_owners) == 2
    # duplicates add up (transforms.py:133-137)
    from aaerec.transforms import lists2sparse, sparse2lists
    assert lists2sparse([[0, 0, 2]], (1, 3)).tocsr().toarray().tolist() == [[2.0, 0.0, 1.0]]
    assert sparse2lists(lists2sparse([[0], [1], [0, 2]], (3, 3))) == [[0], [1], [0, 2]]


def test_corrupt_sets_and_tabcomma(tmp_path):
    import random
    from aaerec.datasets import corrupt_sets, Bags
    random.seed(0)
    kept, dropped = corrupt_sets([[1, 2, 3], [4, 5]], drop=1)
    assert all(len(d) == 1 for d in dropped) and [len(k) for k in kept] == [2, 1]
    assert all(set(k) | set(d) == set(o) for k, d, o in zip(kept, dropped, ([1, 2, 3], [4, 5])))
    path = tmp_path / "d.tsv"
    path.write_text("owner\tyear\tset\ttitle\nA\t2010\tx,y,x\tfoo bar\nB\t2012\tz\tbaz\n")
    bags = Bags.load_tabcomma_format(str(path), unique=True)
    assert bags.data == [["x", "y"], ["z"]] and bags.owner_attributes["year"]["B"] == 2012
    assert bags.get_single_attribute("title") == ["foo bar", "baz"]


# ---- conditions (the intent of reference tests/test_condition.py) ----------------------------------
def test_condition_list_and_concat_pipeline():
    from aaerec.condition import (ConditionList, ConditionBase, EmbeddingBagCondition, CategoricalCondition,
                                  ConcatenationBasedConditioning, ConditionalBiasing, Condition, _check_conditions)
    assert issubclass(EmbeddingBagCondition, ConditionBase) and issubclass(Condition, ConditionBase)
    ebc = EmbeddingBagCondition(10, 7)
    cat = CategoricalCondition(5, sparse=False, use_cuda=False, reduce="sum")
    conds = ConditionList([("title", ebc), ("authors", cat)])
    assert list(conds.keys()) == ["title", "authors"] and conds.size_increment() == 12
    raw = [torch.randint(0, 10, (4, 3)), [["x", "y"], ["y"], ["z", "x", "w"], ["q"]]]
    data = conds.fit_transform(raw)
    assert data[1][1] == [cat.vocab["y"]] and len(cat.vocab) == 5
    assert cat.transform([["unseen"]]) == [[0]]                       # OOV -> padding index
    code = torch.randn(4, 6, requires_grad=True)
    out = conds.encode_impose(code, data)
    assert out.shape == (4, 18)
    w0 = cat.embedding.weight.detach().clone()
    conds.zero_grad()
    out.pow(2).sum().backward()
    conds.step()
    assert not torch.equal(w0, cat.embedding.weight.detach())        # trainable condition learned
    assert torch.equal(cat.embedding.weight[0], torch.zeros(5))      # padding row frozen
    assert _check_conditions(conds, data) and not _check_conditions(None, None)
    with pytest.raises(AssertionError):
        _check_conditions(conds, data[:1])
    with pytest.raises(AssertionError):
        ConditionList([("x", object())])

    class Const(ConcatenationBasedConditioning):
        def size_increment(self):
            return 2
    z = torch.zeros(3, 4)
    assert Const().encode_impose(z, torch.ones(3, 2)).shape == (3, 6)
    assert torch.equal(ConditionalBiasing().encode_impose(z, torch.ones(3, 4)), torch.ones(3, 4))
    generic = Condition(encoder=torch.nn.Linear(3, 2), mode="concat", size_increment=2)
    assert generic.encode_impose(z, torch.ones(3, 3)).shape == (3, 6) and generic.size_increment() == 2


def test_pretrained_embedding_condition_with_vectoriser_object():
    from aaerec.condition import PretrainedWordEmbeddingCondition, ConditionList

    class Vect:
        embedding = np.zeros((11, 300), dtype=np.float32)

        def fit(self, raw):
            return self

        def transform(self, raw):
            return np.ones((len(raw), 300), dtype=np.float32)

        def fit_transform(self, raw):
            return self.transform(raw)
    cond = PretrainedWordEmbeddingCondition(Vect(), use_cuda=False)
    cl = ConditionList([("title", cond)])
    assert cl.size_increment() == 300 and cond.constant_concat
    data = cl.fit_transform([["a b", "c"]])
    assert cl.encode_impose(torch.zeros(2, 5), data).shape == (2, 305)


# ---- model surface that needs no GPU ----------------------------------------------------------------
def test_aae_constructor_surface_and_errors():
    from aaerec.aae import AdversarialAutoEncoder, AAERecommender, PRIOR_ACTIVATIONS
    from aaerec.base import Recommender
    m = AdversarialAutoEncoder(n_hidden=20, n_code=5, prior="Categorical", optimizer="SGD", verbose=False)
    assert m.prior == "categorical" and m.encoder_activation == "softmax" and m.optimizer == "sgd"
    assert "Adversarial Autoencoder (20, 20, 5, 20, 20) optimized by sgd" in str(m)
    assert PRIOR_ACTIVATIONS == {"categorical": "softmax", "bernoulli": "sigmoid", "gauss": "linear"}
    with pytest.raises(NotImplementedError):
        m.fit(sp.identity(3, format="csr"), y=[1, 2, 3])
    with pytest.raises(NotImplementedError):
        m.partial_fit(np.eye(3), y=[1, 2, 3])
    with pytest.raises(KeyError):
        AdversarialAutoEncoder(optimizer="rmsprop")
    from aaerec.aae import AutoEncoder
    ae = AutoEncoder(n_hidden=20, n_code=5, lr=0.01, verbose=False)
    assert ae.gen_lr == ae.reg_lr == 0.01 and ae.encoder_activation == "linear" and str(ae).startswith("Autoencoder (20, 20, 5")
    rec = AAERecommender(n_hidden=10, n_epochs=1)
    assert isinstance(rec, Recommender) and str(rec).startswith("Adversarial Autoencoder\nModel Params:")
    with pytest.raises(RuntimeError):                # duplicate items -> 2.0 -> the reference's BCE error
        from aaerec.aae import _validate_targets
        _validate_targets(sp.csr_matrix(np.array([[2.0, 0.0]])))


def test_torch_custom_ops_are_registered_and_have_no_cpu_path():
    """torch.ops.aaerec.* (aaerec/ops.py) exist after import and refuse host tensors: the product has no CPU
    fallback (the dispatcher has no CPU kernel for them)."""
    import torch
    from aaerec import ops  # noqa: F401  (registers the library)
    for name in ("step", "encode", "predict", "predict_topk"):
        assert hasattr(torch.ops.aaerec, name)
    ip = torch.zeros(2, dtype=torch.int64)
    idx = torch.zeros(1, dtype=torch.int32)
    val = torch.ones(1)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.aaerec.encode(1, ip, idx, val, 0, 1, 1)


def test_embedded_vectorizer_host_side_matches_reference_fixture():
    """aaerec.ub.EmbeddedVectorizer: the TF-IDF half (host, scikit-learn) against the fixture made with the
    reference's own fit / transform (tools/gen_golden.py vectorizer); the product itself is a GPU kernel and
    transform() refuses to run without one."""
    import json
    from aaerec.ub import EmbeddedVectorizer, GensimEmbeddedVectorizer
    z = np.load(os.path.join(GOLDEN, "embedded_vectorizer.npz"))
    words, docs, test = (json.loads(str(z[k])) for k in ("words", "docs", "test"))
    for tag in ("default", "sublinear"):
        v = EmbeddedVectorizer(z["embedding"], words, **json.loads(str(z[f"{tag}.kwargs"])))
        assert v.fit(docs) is v and v.vocabulary_["w7"] == 7 and "oov" not in v.vocabulary_
        np.testing.assert_allclose(v.tfidf.transform(docs) @ z["embedding"], z[f"{tag}.train"], atol=1e-12)
        np.testing.assert_allclose(v.tfidf.transform(test) @ z["embedding"], z[f"{tag}.test"], atol=1e-12)
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError, match="no CPU path"):
                v.transform(docs)
    with pytest.raises(ValueError):
        EmbeddedVectorizer(z["embedding"], words[:-1])
    with pytest.raises(TypeError):
        EmbeddedVectorizer(z["embedding"], words, vocabulary=words)

    class KV:                      # gensim >= 4 attribute names
        index_to_key, vectors = words, z["embedding"]
    g = GensimEmbeddedVectorizer(KV())
    assert repr(g) == "Gensim Embedded Vectorizer with embedding shape (400, 300)"


# ---- boundary: the reference's drivers keep importing what this build does not restate ----------
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "aaerec")), reason="the reference checkout exists in the build container only")
def test_driver_import_lines_resolve_with_this_package_in_front_of_the_reference():
    """sys.path = [aae-recommender_amd, <the user's reference checkout>].  The import lines of the
    reference's main.py:11-20 must all resolve - the modules this build mirrors from HERE, the out-of-scope ones
    (aaerec.baselines, aaerec.svd: main.py:14-15) from the user's checkout through aaerec.__path__ - in a fresh
    interpreter (gensim, main.py:18, is not part of either package and not in this image)."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "aae-recommender_amd")
    code = (
        "import sys, os\n"
        f"sys.path[:0] = [{pkg!r}, {REFERENCE!r}]\n"
        "from aaerec.datasets import Bags\n"
        "from aaerec.evaluation import Evaluation\n"
        "from aaerec.aae import AAERecommender, DecodingRecommender\n"
        "from aaerec.baselines import RandomBaseline, Countbased, MostPopular\n"
        "from aaerec.svd import SVDRecommender\n"
        "from aaerec.vae import VAERecommender\n"
        "from aaerec.dae import DAERecommender\n"
        "from aaerec.condition import ConditionList, PretrainedWordEmbeddingCondition, CategoricalCondition\n"
        "import aaerec, aaerec.svd, aaerec.baselines\n"
        "here = lambda m: os.path.realpath(sys.modules[m].__file__).startswith(os.path.realpath(sys.argv[1]))\n"
        "mine = ['aaerec.datasets', 'aaerec.evaluation', 'aaerec.aae', 'aaerec.vae', 'aaerec.dae', 'aaerec.condition', 'aaerec.base', 'aaerec.ub']\n"
        "assert all(here(m) for m in mine), [m for m in mine if not here(m)]\n"
        "assert not here('aaerec.svd') and not here('aaerec.baselines')\n"
        "assert issubclass(SVDRecommender, aaerec.base.Recommender) and issubclass(Countbased, aaerec.base.Recommender)\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code, pkg], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_package_alone_reports_missing_out_of_scope_modules_plainly():
    """Without a reference checkout on sys.path the out-of-scope modules are simply absent (ModuleNotFoundError naming the
    module): nothing of them is restated or stubbed here."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "aae-recommender_amd")
    code = (f"import sys\nsys.path.insert(0, {pkg!r})\n"
            "try:\n    import aaerec.svd\nexcept ModuleNotFoundError as e:\n    print('missing', e.name)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "missing aaerec.svd", (out.stdout, out.stderr[-2000:])
