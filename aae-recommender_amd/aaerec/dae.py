""" Denoising Autoencoders (mirror of the reference's aaerec/dae.py:144-382 on the HIP kernels).

`DenoisingAutoEncoder` is the plain autoencoder step (`aae_step` with cfg.reserved[2] = 1, i.e. the
reference's ae_step: aae.py:676-711 == dae.py:189-210) on a corrupted batch.  The reference's default
corruption, ``corrupt='zeros'`` (dae.py:48-52), sets a random subset of the batch tensor to zero IN PLACE,
so the encoder input and the BCE target are both the thinned bag: on CSR data that is a per-entry keep
flag, applied to the values on the device.  ``corrupt='gauss'`` (dae.py:40-45) adds dense N(0, noise_factor)
noise to every one of the N input columns and returns a NEW tensor: the encoder's first layer becomes a dense
[B,N]x[N,h] product with a dense weight gradient and an eager Adam over every row of enc.lin1 for those steps
(`aae_set_input_noise` before the step: `dense_input_kernel` + the split-K GEMM of csrc/gemm_f32.h), the BCE target
stays the clean batch.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import _hip
from .aae import AutoEncoder, log_losses, _validate_targets   # noqa: F401  (log_losses: same format, dae.py:33-37)
from .base import Recommender
from .condition import _check_conditions

TINY = 1e-12
NOISE_TYPES = ("gauss", "zeros")


class DenoisingAutoEncoder(AutoEncoder):
    def __init__(self, n_hidden=100, n_code=50, lr=0.001, batch_size=100, n_epochs=500, optimizer='adam',
                 normalize_inputs=True, activation='ReLU', dropout=(.2, .2), noise_factor=0.2, corrupt='zeros',
                 conditions=None, verbose=True, device=None, rng_mode="device", seed=None):
        super().__init__(n_hidden=n_hidden, n_code=n_code, lr=lr, batch_size=batch_size, n_epochs=n_epochs,
                         optimizer=optimizer, normalize_inputs=normalize_inputs, activation=activation,
                         dropout=dropout, conditions=conditions, verbose=verbose, device=device,
                         rng_mode=rng_mode, seed=seed)
        self.noise_factor = noise_factor
        if corrupt.lower() not in NOISE_TYPES:
            raise KeyError(corrupt.lower())                      # NOISE_TYPES[corrupt.lower()], dae.py:173
        self.corrupt = corrupt.lower()
        # corrupt='gauss' adds N(0, noise_factor) to ALL N input columns (dae.py:40-45): the encoder's first layer runs as
        # a dense product for those steps (aae_set_input_noise); the BCE target stays the clean batch
        self._dense_noise = self.corrupt == "gauss"

    def __str__(self):
        return "Denoising " + super().__str__()

    # ---- corruption ------------------------------------------------------------------------------
    def _thinned(self, csr, keep):
        out = _hip.DeviceCSR.__new__(_hip.DeviceCSR)
        out.touch()                                              # (new values: a content id of its own, aae_batch.generation)
        out.shape, out.nnz_per_row_max = csr.shape, csr.nnz_per_row_max
        out.indptr, out.indices = csr.indptr, csr.indices
        out.values = csr.values * keep.to(csr.values.dtype)
        return out

    def _gauss_noise(self, n_rows):
        """gauss_noise's `torch.randn(batch.size()) * noise_factor` (dae.py:42): off the global CPU generator in the
        reference's draw order (rng_mode='reference'), else drawn on the device."""
        n_items = self.hip.N
        if self.rng_mode == "reference":
            return torch.randn(n_rows, n_items) * self.noise_factor
        return torch.randn(n_rows, n_items, device=self.hip.device) * self.noise_factor

    def _epoch_csr(self, csr):
        # every row is visited once per epoch, so thinning the whole resident corpus once per epoch with fresh
        # randomness is the per-batch zeros_noise of the reference (dae.py:48-52, 191) in distribution
        if self.corrupt == "gauss" or self.rng_mode == "reference":
            return csr                                           # drawn per batch, in the reference's order
        return self._thinned(csr, torch.rand_like(csr.values) >= self.noise_factor)

    def _reference_keep(self, X_batch):
        """`torch.rand(batch.size()) < noise_factor` over the DENSE batch (dae.py:50), as per-entry keep flags."""
        mask = torch.rand(X_batch.shape[0], X_batch.shape[1]) < self.noise_factor
        rows = np.repeat(np.arange(X_batch.shape[0]), np.diff(X_batch.indptr))
        return ~mask[torch.from_numpy(rows), torch.from_numpy(X_batch.indices.astype(np.int64))]

    def _run_step(self, csr, row_start, n_rows, rows, c_batch):
        if self.corrupt == "gauss":
            # the encoder of THIS step reads the dense batch + noise (corruption is drawn before the dropout masks);
            # the BCE target stays the clean batch (gauss_noise returns a new tensor, dae.py:40-45)
            self.hip.set_input_noise(self._gauss_noise(n_rows))
            return super()._run_step(csr, row_start, n_rows, rows, c_batch)
        if self.rng_mode == "reference" and getattr(self, "_in_fit", False):
            # the reference's draw order within a step: corruption mask first, then the dropout masks
            idx = rows.cpu().numpy() if rows is not None else np.arange(row_start, row_start + n_rows)
            Xb = self._fit_X[idx].tocsr()
            Xb.sort_indices()
            b = _hip.DeviceCSR(Xb, self.hip.device)
            b.nnz_per_row_max = max(b.nnz_per_row_max, 1)
            b = self._thinned(b, self._reference_keep(Xb).to(self.hip.device))
            return super()._run_step(b, 0, n_rows, None, c_batch)
        return super()._run_step(csr, row_start, n_rows, rows, c_batch)

    # ---- public API (dae.py:212-314) ---------------------------------------------------------------
    def partial_fit(self, X, y=None, condition_data=None, step=None, keep=None, noise=None):
        """ Performs one denoising reconstruction step.  keep: optional per-entry keep flags (CSR order) that
        replace the random corruption (parity runs); noise: the same for corrupt='gauss' ([rows, n_items], scaled). """
        use_condition = _check_conditions(self.conditions, condition_data)
        if y is not None:
            raise ValueError("(Semi-)supervised usage not supported")
        Xs = sp.csr_matrix(X) if not sp.issparse(X) else X.tocsr()
        Xs.sort_indices()
        if self.hip is None:
            self._build(Xs.shape[1], self.conditions.size_increment() if use_condition else 0)
        _validate_targets(Xs)
        if Xs.shape[0] > self.hip.max_batch:
            raise ValueError("batch of {} rows exceeds batch_size={}".format(Xs.shape[0], self.hip.max_batch))
        csr = _hip.DeviceCSR(Xs, self.hip.device)
        if self.corrupt == "gauss":
            self.train()
            self.hip.set_input_noise(torch.as_tensor(noise, dtype=torch.float32) if noise is not None
                                     else self._gauss_noise(Xs.shape[0]))
            AutoEncoder._run_step(self, csr, 0, Xs.shape[0], None, condition_data if use_condition else None)
            if self.verbose:
                self.last_losses = self.hip.losses()
                log_losses(self.last_losses[0], 0, 0)
            return self
        if keep is not None:
            kp = torch.as_tensor(np.asarray(keep), device=self.hip.device) != 0
        elif self.rng_mode == "reference":
            kp = self._reference_keep(Xs).to(self.hip.device)
        else:
            kp = torch.rand_like(csr.values) >= self.noise_factor
        csr = self._thinned(csr, kp)
        self.train()
        AutoEncoder._run_step(self, csr, 0, Xs.shape[0], None, condition_data if use_condition else None)
        if self.verbose:
            self.last_losses = self.hip.losses()
            log_losses(self.last_losses[0], 0, 0)
        return self

    def fit(self, X, y=None, condition_data=None):
        self._in_fit = True
        try:
            return super().fit(X, y=y, condition_data=condition_data)
        finally:
            self._in_fit = False


class DAERecommender(Recommender):
    """
    Denoising Recommender
    =====================================
    Keyword arguments are forwarded to DenoisingAutoEncoder (n_hidden, n_code, n_epochs, batch_size, lr,
    noise_factor, corrupt, normalize_inputs, verbose, ...).
    """

    def __init__(self, conditions=None, **kwargs):
        super().__init__()
        self.verbose = kwargs.get('verbose', True)
        self.model_params = kwargs
        self.conditions = conditions
        self.dae = None

    def __str__(self):
        desc = "Denoising Autoencoder"
        if self.conditions:
            desc += " conditioned on: " + ', '.join(self.conditions.keys())
        desc += '\nDAE Params: ' + str(self.model_params)
        return desc

    def train(self, training_set):
        X = training_set.tocsr()
        if self.conditions:
            condition_data = self.conditions.fit_transform(training_set.get_attributes(self.conditions.keys()))
        else:
            condition_data = None
        self.dae = DenoisingAutoEncoder(conditions=self.conditions, **self.model_params)
        print(self)
        print(self.dae)
        print(self.conditions)
        self.dae.fit(X, condition_data=condition_data)

    def predict(self, test_set):
        X = test_set.tocsr()
        if self.conditions:
            # Important to not call fit here, but just transform
            condition_data = self.conditions.transform(test_set.get_attributes(self.conditions.keys()))
        else:
            condition_data = None
        return self.dae.predict(X, condition_data=condition_data)
