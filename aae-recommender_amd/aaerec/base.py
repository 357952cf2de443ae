"""Recommender plugin protocol: the surface the evaluation drivers call (reference aaerec/base.py:5-19).

A plugin is trained on one ``Bags`` and asked to score every item for the rows of another; the HIP-backed
recommenders (``AAERecommender``, ``DecodingRecommender``, ``DAERecommender``, ``VAERecommender``) all subclass
this.  Like the reference's abstract base, a class that leaves ``train`` or ``predict`` undefined cannot be
instantiated (TypeError).
"""

_REQUIRED = ("train", "predict")


class Recommender:
    use_wandb = False           # the reference's drivers read this flag; experiment tracking is out of scope here

    def __new__(cls, *args, **kwargs):
        missing = [name for name in _REQUIRED if getattr(cls, name) is getattr(Recommender, name)]
        if missing:
            raise TypeError("Can't instantiate {} without an implementation of: {}".format(
                cls.__name__, ", ".join(missing)))
        return super().__new__(cls)

    def train(self, X_train):
        """Fit on the training Bags."""
        raise NotImplementedError

    def predict(self, X_test):
        """Score every item for every row of the test Bags: [n_docs, n_items], ndarray or scipy sparse."""
        raise NotImplementedError
