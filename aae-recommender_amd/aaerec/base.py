"""Recommender plugin protocol (mirrors reference aaerec/base.py:5-19)."""
from abc import ABC, abstractmethod


class Recommender(ABC):
    """A recommender is trained on a Bags instance and scores every item for the rows of another."""
    use_wandb = False

    def __init__(self):
        super().__init__()

    @abstractmethod
    def train(self, X_train):
        """Fit on the training Bags."""
        raise NotImplementedError

    @abstractmethod
    def predict(self, X_test):
        """Return an [n_docs, n_items] score matrix (ndarray or scipy sparse)."""
        raise NotImplementedError
