"""Weighted bag of embedded words: the document vectors behind PretrainedWordEmbeddingCondition
(reference aaerec/ub.py:38-89, used from condition.py:345-369).

    document -> TF-IDF over the embedding's own vocabulary (scikit-learn, host) -> sparse [docs, V]
             -> times the embedding matrix [V, D]                               -> dense  [docs, D]

The TF-IDF weighting is host-side text processing exactly as in the reference; the product runs on the GPU
(`aae_csr_embed` of libaaerec_hip.so: the embedding matrix is uploaded once and stays resident, GoogleNews-sized
tables of 3 M x 300 are 3.6 GB of the 288 GB).  There is no CPU product: transform() without a GPU raises.
"""
import numpy as np
import scipy.sparse as sp
import torch
from sklearn.feature_extraction.text import TfidfVectorizer

from . import _hip


class AutoEncoderMixin:
    """Protocol mixin of the reference's sklearn-style autoencoders (reference aaerec/ub.py:5-11): a model with
    transform() and inverse_transform() reconstructs by chaining them.  Kept here because out-of-scope modules of a user's
    own reference checkout that fall through aaerec.__path__ (aaerec/svd.py:7) import it from this module."""

    def reconstruct(self, X, y=None):
        return self.inverse_transform(self.transform(X))


class EmbeddedVectorizer:
    """ Weighted Bag-of-embedded-Words

    embedding   [V, D] matrix (anything np.asarray takes)
    index2word  the V words, row i of the embedding belonging to index2word[i]
    device      GPU the table lives on (default: the current one)
    other keyword arguments go to sklearn's TfidfVectorizer (the vocabulary is fixed to index2word)
    """

    def __init__(self, embedding, index2word, device=None, **tfidf_params):
        self.embedding = np.ascontiguousarray(np.asarray(embedding), dtype=np.float32)
        if self.embedding.ndim != 2 or self.embedding.shape[0] != len(index2word):
            raise ValueError("embedding must be [len(index2word), dim]")
        if "vocabulary" in tfidf_params:
            raise TypeError("the vocabulary is the embedding's: index2word")
        self.tfidf = TfidfVectorizer(vocabulary=list(index2word), **tfidf_params)
        self.device = device
        self._table = None

    # sklearn-style surface the conditions call -------------------------------------------------------
    def fit(self, raw_documents, y=None):
        self.tfidf.fit(raw_documents)
        return self

    def transform(self, raw_documents, __y=None):
        scores = self.tfidf.transform(raw_documents)               # sparse [docs, V], float64
        return self._times_embedding(scores)

    def fit_transform(self, raw_documents, y=None):
        return self.fit(raw_documents, y).transform(raw_documents, y)

    @property
    def vocabulary_(self):
        return self.tfidf.vocabulary_

    def __repr__(self):
        return f"Embedded Vectorizer with embedding shape {self.embedding.shape}"

    # -----------------------------------------------------------------------------------------------------
    def _times_embedding(self, scores):
        if not torch.cuda.is_available():
            raise RuntimeError("EmbeddedVectorizer.transform multiplies on the GPU (libaaerec_hip aae_csr_embed); "
                               "no GPU is visible and there is no CPU path")
        dev = torch.device(self.device if self.device is not None else ("cuda:%d" % torch.cuda.current_device()))
        if self._table is None or self._table.device != dev:
            self._table = torch.from_numpy(self.embedding).to(dev)
        scores = sp.csr_matrix(scores, dtype=np.float32)
        scores.sort_indices()                                      # accumulate in column order, as scipy does
        out = np.empty((scores.shape[0], self.embedding.shape[1]), dtype=np.float32)
        step = 1 << 16                                             # documents per launch
        for start in range(0, scores.shape[0], step):
            part = scores[start:start + step]
            out[start:start + step] = _hip.csr_embed(_hip.DeviceCSR(part, dev), self._table).cpu().numpy()
        return out


class GensimEmbeddedVectorizer(EmbeddedVectorizer):
    """ Shorthand for a gensim KeyedVectors-like object: `.vectors` plus `.index2word` (gensim < 4) or
    `.index_to_key` (gensim >= 4), so that the vocabularies match. """

    def __init__(self, gensim_vectors, **kwargs):
        words = getattr(gensim_vectors, "index2word", None)
        if words is None:
            words = gensim_vectors.index_to_key
        super().__init__(gensim_vectors.vectors, words, **kwargs)

    def __repr__(self):
        return "Gensim " + super().__repr__()
