"""Condition (side-information) plugin API - mirrors reference aaerec/condition.py.

A ConditionList is an ordered dict {attribute name: condition}.  On the hot path only
`encode_impose / zero_grad / step / train / eval / size_increment` are used
(reference aae.py:688-709, 773-780, 844-865): the encoder output z leaves the HIP kernels as a
torch tensor, every condition encodes its batch input and imposes it (concatenate / add /
multiply) with ordinary differentiable torch ops, the result goes back into the decoder
kernels, and the gradient the kernels return for it is back-propagated through this small torch
graph so trainable conditions (embedding tables with their own optimiser) keep learning.

Conditions whose encoded input is a constant block that is concatenated
(`PretrainedWordEmbeddingCondition`, anything with `constant_concat = True`) take the fully
fused kernel path instead: their block is copied next to z on the device.  So does a
`CategoricalCondition` whose table lives on the GPU: its lookup, backward and SparseAdam / Adam step
are two small kernels of the library around the fused step (`device_native`).
"""
import itertools as it
from abc import ABC, abstractmethod
from collections import Counter, OrderedDict

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
from torch import optim


def _check_conditions(conditions, condition_data):
    """True iff conditions are in use; asserts that spec and data match (reference 31-57)."""
    if not conditions and not condition_data:
        return False
    assert isinstance(conditions, ConditionList), "`conditions` no instance of ConditionList"
    assert condition_data and conditions, "Mismatch between condition spec and supplied condition data."
    assert len(condition_data) == len(conditions), "Unexpected number of supplied condition data"
    return True


class ConditionList(OrderedDict):
    def __init__(self, items):
        super().__init__(items)
        assert all(isinstance(v, ConditionBase) for v in self.values())

    def fit(self, raw_inputs):
        assert len(raw_inputs) == len(self)
        for cond, inp in zip(self.values(), raw_inputs):
            cond.fit(inp)
        return self

    def transform(self, raw_inputs):
        assert len(raw_inputs) == len(self)
        return [cond.transform(inp) for cond, inp in zip(self.values(), raw_inputs)]

    def fit_transform(self, raw_inputs):
        assert len(raw_inputs) == len(self)
        return [cond.fit_transform(inp) for cond, inp in zip(self.values(), raw_inputs)]

    def encode_impose(self, x, condition_inputs, dim=None):
        assert len(condition_inputs) == len(self)
        for cond, inp in zip(self.values(), condition_inputs):
            x = cond.encode_impose(x, inp, dim)
        return x

    def encode(self, condition_inputs):
        assert len(condition_inputs) == len(self)
        return [cond.encode(inp) for cond, inp in zip(self.values(), condition_inputs)]

    def zero_grad(self):
        for cond in self.values():
            cond.zero_grad()
        return self

    def step(self):
        for cond in self.values():
            cond.step()
        return self

    def size_increment(self):
        return sum(cond.size_increment() for cond in self.values())

    def train(self):
        for cond in self.values():
            if hasattr(cond, "train"):
                cond.train()

    def eval(self):
        for cond in self.values():
            if hasattr(cond, "eval"):
                cond.eval()


_PROTOCOL = ("encode", "impose", "encode_impose", "size_increment", "fit", "transform", "fit_transform",
             "zero_grad", "step", "train", "eval")


class ConditionBase(ABC):
    """fit/transform once on the whole attribute column; encode + impose per batch; optional own
    parameters updated through zero_grad/step."""

    def fit(self, raw_inputs):
        return self

    def transform(self, raw_inputs):
        return raw_inputs

    def fit_transform(self, raw_inputs):
        return self.fit(raw_inputs).transform(raw_inputs)

    @abstractmethod
    def size_increment(self):
        """How many columns the condition appends to the code (0 for bias / scale)."""

    def encode(self, inputs):
        return inputs

    @abstractmethod
    def impose(self, inputs, encoded_condition, dim=None):
        raise NotImplementedError

    def encode_impose(self, inputs, condition_input, dim=None):
        return self.impose(inputs, self.encode(condition_input), dim=None)

    def zero_grad(self):
        return self

    def step(self):
        return self

    def train(self):
        return self

    def eval(self):
        return self

    @classmethod
    def __subclasshook__(cls, C):
        if cls is ConditionBase:
            if all(any(name in B.__dict__ for B in C.__mro__) for name in _PROTOCOL):
                return True
        return NotImplemented


def _to_device_like(encoded, ref):
    if torch.is_tensor(encoded) and torch.is_tensor(ref) and encoded.device != ref.device:
        return encoded.to(ref.device)
    return encoded


class ConcatenationBasedConditioning(ConditionBase):
    dim = 1

    @abstractmethod
    def size_increment(self):
        """Subclasses say how wide their block is."""

    def impose(self, inputs, encoded_condition, dim=None):
        return torch.cat([inputs, _to_device_like(encoded_condition, inputs)], dim=self.dim if dim is None else dim)


class ConditionalBiasing(ConditionBase):
    def impose(self, inputs, encoded_condition, dim=None):
        return inputs + _to_device_like(encoded_condition, inputs)

    def size_increment(self):
        return 0


class ConditionalScaling(ConditionBase):
    def impose(self, inputs, encoded_condition, dim=None):
        return inputs * _to_device_like(encoded_condition, inputs)

    def size_increment(self):
        return 0


class CountCondition(ConditionBase):
    """Binary bag-of-words of a text attribute, hstacked onto sparse inputs (reference 258-281)."""

    def __init__(self, **cv_params):
        from sklearn.feature_extraction.text import CountVectorizer
        self.cv = CountVectorizer(binary=True, **cv_params)

    def fit(self, raw_inputs):
        self.cv.fit(raw_inputs)
        return self

    def transform(self, raw_inputs):
        return self.cv.transform(raw_inputs)

    def fit_transform(self, raw_inputs):
        return self.cv.fit_transform(raw_inputs)

    def impose(self, x, encoded_inputs, dim=None):
        assert dim is None, "dim not supported for scipy.sparse based imposing"
        return sp.hstack([x, encoded_inputs])

    def size_increment(self):
        return len(self.cv.vocabulary_)


class PretrainedWordEmbeddingCondition(ConcatenationBasedConditioning):
    """Fixed document vectors (TF-IDF weighted mean of pre-trained word vectors) concatenated to
    the code (reference 345-369).  `vectors` is either a gensim-KeyedVectors-like object (needs
    the optional aaerec.ub vectoriser) or any object with fit/transform/fit_transform and an
    `embedding` array of shape [vocab, dim]."""
    constant_concat = True

    def __init__(self, vectors, dim=1, use_cuda=None, **tfidf_params):
        if all(hasattr(vectors, a) for a in ("fit", "transform", "embedding")):
            self.vect = vectors
        else:
            try:
                from .ub import GensimEmbeddedVectorizer
            except ImportError as exc:   # the TF-IDF x embedding preprocessing is outside the hot path
                raise ImportError("PretrainedWordEmbeddingCondition with raw word vectors needs the optional "
                                  "aaerec.ub vectoriser; pass a fitted vectoriser object instead") from exc
            self.vect = GensimEmbeddedVectorizer(vectors, **tfidf_params)
        self.dim = dim
        use_cuda = torch.cuda.is_available() if use_cuda is None else use_cuda
        self.device = torch.device("cuda") if use_cuda else torch.device("cpu")

    def fit(self, raw_inputs):
        self.vect.fit(raw_inputs)
        return self

    def transform(self, raw_inputs):
        return self.vect.transform(raw_inputs)

    def fit_transform(self, raw_inputs):
        return self.vect.fit_transform(raw_inputs)

    def encode(self, inputs):
        if self.device.type == "cuda":
            from . import _hip
            return _hip.upload(inputs, self.device, torch.float32)
        return torch.as_tensor(inputs, dtype=torch.float32, device=self.device)

    def size_increment(self):
        return self.vect.embedding.shape[1]


class EmbeddingBagCondition(ConcatenationBasedConditioning):
    """Trainable nn.EmbeddingBag + its own Adam (reference 372-394)."""

    def __init__(self, num_embeddings, embedding_dim, **kwargs):
        self.embedding_bag = nn.EmbeddingBag(num_embeddings, embedding_dim, **kwargs)
        self.optimizer = optim.Adam(self.embedding_bag.parameters())
        self.embedding_dim = embedding_dim

    def encode(self, inputs):
        return self.embedding_bag(inputs)

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        self.optimizer.step()

    def size_increment(self):
        return self.embedding_dim


class CategoricalCondition(ConcatenationBasedConditioning):
    """Trainable embedding of a categorical attribute; index 0 = padding / out of vocabulary
    (reference 397-508).

    embedding_on_gpu: the reference keeps the table in host memory unless asked (False) "to save GPU memory" and
    pays a host round trip per step; with 288 GB of HBM the default here (None) is the GPU whenever use_cuda, which
    also lets the AAE step train the condition with the library's own kernels (device_native).  An explicit False
    is honoured."""
    padding_idx = 0

    def __init__(self, embedding_dim, vocab_size=None, sparse=True, use_cuda=None, embedding_on_gpu=None,
                 lr=1e-3, reduce=None, **embedding_params):
        self.vocab_size, self.embedding_dim = vocab_size, embedding_dim
        self.vocab = self.embedding = self.optimizer = None
        self.lr, self.sparse = lr, sparse
        self.use_cuda = torch.cuda.is_available() if use_cuda is None else use_cuda
        self.embedding_on_gpu = self.use_cuda if embedding_on_gpu is None else embedding_on_gpu
        assert "padding_idx" not in embedding_params, "Padding is fixed with token 0"
        self.embedding_params = embedding_params
        assert reduce is None or reduce in ("mean", "sum", "max"), "Reduce neither None nor in 'mean','sum','max'"
        self.reduce = reduce

    def fit(self, raw_inputs):
        flat = raw_inputs if self.reduce is None else list(it.chain.from_iterable(raw_inputs))
        if self.vocab_size is None:
            cutoff = len(flat)
        elif isinstance(self.vocab_size, float):
            cutoff = int(self.vocab_size * len(flat))
        else:
            cutoff = int(self.vocab_size)
        self.vocab = {value: i + 1 for i, (value, _) in enumerate(Counter(flat).most_common(cutoff))}
        self.embedding = nn.Embedding(len(self.vocab) + 1, self.embedding_dim, padding_idx=self.padding_idx,
                                      **self.embedding_params, sparse=self.sparse)
        if self.use_cuda and self.embedding_on_gpu:
            self.embedding = self.embedding.cuda()
        opt = optim.SparseAdam if self.sparse else optim.Adam
        self.optimizer = opt(self.embedding.parameters(), lr=self.lr)
        return self

    def transform(self, raw_inputs):
        get = self.vocab.get
        if self.reduce is None:
            return [get(x, self.padding_idx) for x in raw_inputs]
        return [[get(x, self.padding_idx) for x in row] for row in raw_inputs]

    def _pad_batch(self, batch_inputs):
        width = max(len(row) for row in batch_inputs)
        return [list(row) + [self.padding_idx] * (width - len(row)) for row in batch_inputs]

    def encode(self, inputs):
        if self.reduce is not None:
            inputs = self._pad_batch(inputs)
        idx = torch.tensor(inputs, device=self.embedding.weight.device)
        hid = self.embedding(idx)
        if self.reduce is not None:
            hid = getattr(hid, self.reduce)(1)
            if self.reduce == "max":
                hid = hid[0]
        if self.use_cuda:
            hid = hid.cuda()
        return hid

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        self.optimizer.step()

    def size_increment(self):
        return self.embedding_dim

    # ---- device-native training path (libaaerec_hip: aae_cat_encode / aae_cat_update) ------------------------
    # With the table in HBM the whole condition - lookup + reduction, its backward and its SparseAdam / Adam step -
    # runs in two small kernels around aae_step instead of cutting the step at the condition boundary for autograd.
    # The optimiser state lives in self.optimizer.state (torch's own layout), so state_dict() / a later torch-side
    # step see the same tensors.
    def device_native(self, device):
        w = self.embedding.weight if self.embedding is not None else None
        return (w is not None and w.is_cuda and w.device == torch.device(device) and self.reduce in (None, "sum", "mean")
                and not self.embedding_params and self.embedding_dim <= 256)

    def _index_block(self, inputs, device, width=None):
        if self.reduce is None:
            arr = np.asarray(inputs, dtype=np.int32).reshape(-1, 1)
        else:
            width = max(1, width or 0, max(len(row) for row in inputs))
            arr = np.zeros((len(inputs), width), dtype=np.int32)
            for r, row in enumerate(inputs):
                arr[r, :len(row)] = row
        from . import _hip
        return _hip.upload(arr, device)

    def padded_width(self, inputs):
        """Width the reference pads a batch of value lists to (its longest list); 'mean' divides by it."""
        return 1 if self.reduce is None else max(1, max(len(row) for row in inputs))

    def encode_into(self, out, inputs, width=None):
        """encode(inputs) written into `out` ([rows, embedding_dim], may be a column slice of a wider block).
        width: pad to at least this many values (a rank's share of a batch uses the whole batch's width)."""
        from . import _hip
        w = self.embedding.weight
        self._idx = self._index_block(inputs, w.device, width)
        _hip.cat_encode(w.data, self._idx, out, mean=self.reduce == "mean")

    def update_from(self, dout, inputs=None):
        """zero_grad + backward + step from dL/d(encoded block) of the rows last handed to encode_into, or of
        `inputs` (data parallel: every rank applies the gathered gradient of the whole batch)."""
        from . import _hip
        w = self.embedding.weight
        if inputs is not None:
            self._idx = self._index_block(inputs, w.device)
        st = self.optimizer.state[w]
        if "exp_avg" not in st:
            st["step"] = 0 if self.sparse else torch.tensor(0.0)
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(w.data), torch.zeros_like(w.data)
        if not self.sparse and getattr(self, "_grad_scratch", None) is None:
            self._grad_scratch = torch.zeros_like(w.data)
        st["step"] += 1
        _hip.cat_update(w.data, st["exp_avg"], st["exp_avg_sq"], self._idx, dout, self.optimizer.param_groups[0]["lr"],
                        int(st["step"]), mean=self.reduce == "mean",
                        grad_scratch=None if self.sparse else self._grad_scratch)


class Condition(ConditionBase):
    """Generic condition assembled from a preprocessor, an encoder module and an optimiser
    (reference 514-603)."""

    def __init__(self, preprocessor=None, encoder=None, optimizer=None, mode="concat", size_increment=0, dim=1):
        if encoder is not None:
            assert callable(encoder)
        assert mode in ("concat", "bias", "scale")
        if mode == "concat":
            assert size_increment > 0, "Specify size increment in concat mode"
        else:
            assert size_increment == 0, "Size increment should be zero in bias or scale modes"
        if preprocessor is not None:
            for meth in ("fit", "transform", "fit_transform"):
                assert hasattr(preprocessor, meth), "Preprocessor has no {} method".format(meth)
        if optimizer is not None:
            assert hasattr(optimizer, "zero_grad") and hasattr(optimizer, "step")
        self.preprocessor, self.encoder, self.optimizer = preprocessor, encoder, optimizer
        self.mode_, self.dim, self._inc = mode, dim, size_increment

    def fit(self, raw_inputs):
        if self.preprocessor is not None:
            self.preprocessor.fit(raw_inputs)
        return self

    def transform(self, raw_inputs):
        return self.preprocessor.transform(raw_inputs) if self.preprocessor is not None else raw_inputs

    def fit_transform(self, raw_inputs):
        return self.preprocessor.fit_transform(raw_inputs) if self.preprocessor is not None else raw_inputs

    def encode(self, inputs):
        return self.encoder(inputs) if self.encoder is not None else inputs

    def impose(self, inputs, encoded_condition, dim=None):
        enc = _to_device_like(encoded_condition, inputs)
        if self.mode_ == "concat":
            return torch.cat([inputs, enc], dim=self.dim)
        return inputs + enc if self.mode_ == "bias" else inputs * enc

    def size_increment(self):
        return self._inc

    def zero_grad(self):
        if self.optimizer is not None:
            self.optimizer.zero_grad()

    def step(self):
        if self.optimizer is not None:
            self.optimizer.step()

    def train(self):
        if self.encoder is not None and hasattr(self.encoder, "train"):
            self.encoder.train()

    def eval(self):
        if self.encoder is not None and hasattr(self.encoder, "eval"):
            self.encoder.eval()
