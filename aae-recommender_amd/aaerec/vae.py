""" Variational autoencoder recommender (mirror of the reference's aaerec/vae.py:47-345 on the HIP kernels).

x -> L1 normalise -> fc1 -> act -> (fc21 = mu, fc22 = logvar) -> z = mu + eps * exp(logvar / 2) -> [conditions]
-> fc3 -> act -> fc4 -> sigmoid;  loss = nn.BCELoss() (mean) + KL sum (vae.py:132-145);  one optimiser.
The step is `aae_vae_step` of libaaerec_hip.so (cfg.reserved[2] = 3): the sparse first-layer gather/scatter with
lazy Adam, the fused vocabulary-wide output layer and the layer-chain kernel with two VAE ops (reparametrise and
its backward incl. the KL gradient).  Parameter names in the kernels' model: enc.lin1 = fc1, enc.lin3 =
[fc21; fc22], dec.lin1 = fc3, dec.lin3 = fc4.

Like the reference, predict() samples eps as well (vae.py:229-266 runs the same forward in eval mode).
Conditions: constant concatenated blocks (e.g. PretrainedWordEmbeddingCondition) and CategoricalConditions whose
table lives on this GPU ride in aae_vae_step (encoded and trained by aae_cat_encode / aae_cat_update around the step, as
in the AAE - condition.py:397-508); any other plugin runs with torch autograd between the code and the decoder: the step
is cut at the condition boundary (aae_vae_encode -> encode_impose -> aae_vae_decode_backward -> the plugins' backward and
step -> aae_vae_encoder_backward).
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import _hip
from .aae import TORCH_OPTIMIZERS, AdversarialAutoEncoder, _take, _validate_targets
from .base import Recommender
from .condition import _check_conditions

STATUS_FORMAT = "[ R: {:.4f}]"


def log_losses(loss):
    print('\r' + STATUS_FORMAT.format(loss), end='', flush=True)


class VAE:
    def __init__(self, inp, out, n_hidden=100, n_code=50, lr=0.001, batch_size=100, n_epochs=500, optimizer='adam',
                 normalize_inputs=True, activation='ReLU', final_activation='Sigmoid', conditions=None, verbose=True,
                 log_interval=1, device=None, rng_mode="device", seed=None):
        if inp != out:
            raise ValueError("the bag-of-items VAE reconstructs its input: inp must equal out")
        if final_activation != 'Sigmoid':
            raise NotImplementedError("final_activation='{}': the fused output layer is sigmoid + BCE".format(
                final_activation))
        if optimizer.lower() not in TORCH_OPTIMIZERS:
            raise KeyError(optimizer.lower())
        if rng_mode not in ("device", "reference"):
            raise ValueError("rng_mode must be 'device' or 'reference'")
        self.normalize_inputs = normalize_inputs
        self.inp, self.n_hidden, self.n_code = inp, n_hidden, n_code
        self.n_epochs, self.verbose, self.batch_size, self.lr = n_epochs, verbose, batch_size, lr
        self.activation, self.conditions, self.log_interval = activation, conditions, log_interval
        self.rng_mode = rng_mode
        self.training = True
        self.last_loss = None
        code_inc = int(conditions.size_increment()) if conditions else 0
        dev_probe = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # conditions the kernels produce / train themselves ride in aae_vae_step; any other plugin gets the step cut at
        # the condition boundary and runs with torch autograd in between (aae_vae_encode / _decode_backward / ...)
        self._cond_native = not conditions or all(
            getattr(c, "constant_concat", False) or (hasattr(c, "device_native") and c.device_native(dev_probe))
            for c in conditions.values())
        self._dp = None
        seed = seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) \
            if rng_mode == "device" else 0
        self.hip = _hip.HipAAE(inp, n_hidden, n_code, cond_inc=code_inc, max_batch=batch_size, max_nnz=batch_size * 4096,
                               activation=activation, optimizer=optimizer.lower(), normalize_inputs=normalize_inputs,
                               dropout=(0.0, 0.0), gen_lr=lr, reg_lr=lr,
                               rng_mode="device" if rng_mode == "device" else "inject", seed=seed, device=device,
                               vae=True)
        self.device = self.hip.device
        # nn.Linear default initialisation in the reference's construction order: fc1, fc21, fc22, fc3, fc4
        fc1, fc21, fc22 = torch.nn.Linear(inp, n_hidden), torch.nn.Linear(n_hidden, n_code), torch.nn.Linear(n_hidden, n_code)
        fc3, fc4 = torch.nn.Linear(n_code + code_inc, n_hidden), torch.nn.Linear(n_hidden, out)
        n = lambda t: t.detach().numpy()                                              # noqa: E731
        self.hip.load_params({
            "enc.lin1.weight": n(fc1.weight), "enc.lin1.bias": n(fc1.bias),
            "enc.lin3.weight": np.vstack([n(fc21.weight), n(fc22.weight)]),
            "enc.lin3.bias": np.concatenate([n(fc21.bias), n(fc22.bias)]),
            "dec.lin1.weight": n(fc3.weight), "dec.lin1.bias": n(fc3.bias),
            "dec.lin3.weight": n(fc4.weight), "dec.lin3.bias": n(fc4.bias)})

    def __str__(self):
        return "VAE ({} -> {} -> 2x{} -> {} -> {}), lr {}".format(self.inp, self.n_hidden, self.n_code, self.n_hidden,
                                                                 self.inp, self.lr)

    # ---- nn.Module-like surface the drivers touch ----------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def cuda(self):
        return self

    def state_dict(self):
        """The reference's parameter names (fc1, fc21, fc22, fc3, fc4)."""
        sd, c = self.hip.state_dict(), self.n_code
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))                      # noqa: E731
        return {"fc1.weight": t(sd["enc.lin1.weight"]), "fc1.bias": t(sd["enc.lin1.bias"]),
                "fc21.weight": t(sd["enc.lin3.weight"][:c]), "fc21.bias": t(sd["enc.lin3.bias"][:c]),
                "fc22.weight": t(sd["enc.lin3.weight"][c:]), "fc22.bias": t(sd["enc.lin3.bias"][c:]),
                "fc3.weight": t(sd["dec.lin1.weight"]), "fc3.bias": t(sd["dec.lin1.bias"]),
                "fc4.weight": t(sd["dec.lin3.weight"]), "fc4.bias": t(sd["dec.lin3.bias"])}

    # ---- internals ------------------------------------------------------------------------------------
    def _eps(self, B):
        """reparametrize()'s torch.randn_like(std) (vae.py:117) off the global CPU generator in 'reference' mode."""
        return torch.randn(B, self.n_code) if self.rng_mode == "reference" else None

    # the condition block of a batch / the update of trainable tables from its gradient: the AAE's helpers
    _is_constant_concat = AdversarialAutoEncoder._is_constant_concat
    _native_cond_block = AdversarialAutoEncoder._native_cond_block
    _native_cond_update = AdversarialAutoEncoder._native_cond_update

    def _cond(self, c_batch):
        return self._native_cond_block(c_batch, len(c_batch[0]) if not hasattr(c_batch[0], "shape") else c_batch[0].shape[0])

    _cond_fn = AdversarialAutoEncoder._cond_fn

    def _step(self, csr, n_rows, rows, c_batch):
        if c_batch is not None and not self._cond_native:
            # the step cut at the condition boundary (vae.py:120-130, 175-181)
            z = self.hip.vae_encode(csr, 0, n_rows, rows=rows, eps=self._eps(n_rows), train=True)
            zc, back = self._cond_fn(c_batch)(z)
            self.hip.vae_encoder_backward(back(self.hip.vae_decode_backward(zc)))
            self._last_rows = n_rows
            if self.verbose:
                log_losses(self.loss())
            return
        self.hip.vae_step(csr, 0, n_rows, rows=rows, cond=self._cond(c_batch) if c_batch is not None else None,
                          eps=self._eps(n_rows))
        if c_batch is not None and not self._is_constant_concat():
            self._native_cond_update(n_rows)           # conditions.zero_grad / backward / step, vae.py:175-181
        self._last_rows = n_rows
        if self.verbose:
            log_losses(self.loss())

    def loss(self):
        """(mean BCE + KL sum) / rows of the last step: loss.item() / len(X) as the reference logs it (vae.py:185)."""
        l = self.hip.losses()
        self.last_loss = (l[0] + l[1]) / self._last_rows
        return self.last_loss

    # ---- public API (vae.py:147-266) --------------------------------------------------------------------
    def partial_fit(self, X, y=None, condition_data=None):
        use_condition = _check_conditions(self.conditions, condition_data)
        if y is not None:
            raise ValueError("(Semi-)supervised usage not supported")
        Xs = sp.csr_matrix(X) if not sp.issparse(X) else X.tocsr()
        _validate_targets(Xs)
        if Xs.shape[0] > self.hip.max_batch:
            raise ValueError("batch of {} rows exceeds batch_size={}".format(Xs.shape[0], self.hip.max_batch))
        self.train()
        if self.conditions:
            self.conditions.train()
        self._step(_hip.DeviceCSR(Xs, self.device), Xs.shape[0], None, condition_data if use_condition else None)
        return self

    def fit(self, X, y=None, condition_data=None):
        if y is not None:
            raise NotImplementedError("(Semi-)supervised usage not supported")
        use_condition = _check_conditions(self.conditions, condition_data)
        X = X.tocsr()
        _validate_targets(X)
        csr = _hip.DeviceCSR(X, self.device)             # the corpus stays resident in HBM
        n_docs = X.shape[0]
        self.train()
        if self.conditions:
            self.conditions.train()
        for epoch in range(self.n_epochs):
            if self.verbose:
                print("Epoch", epoch + 1)
            perm = np.arange(n_docs)                     # sklearn.utils.shuffle(X, *condition_data), vae.py:207-211
            np.random.shuffle(perm)
            perm_dev = torch.as_tensor(perm.astype(np.int32), device=self.device)
            for start in range(0, n_docs, self.batch_size):
                idx = perm[start:start + self.batch_size]
                c_batch = [_take(c, idx) for c in condition_data] if use_condition else None
                self._step(csr, len(idx), perm_dev[start:start + len(idx)], c_batch)
            if self.verbose:
                print()
        self.loss()
        return self

    def predict(self, X, condition_data=None):
        use_condition = _check_conditions(self.conditions, condition_data)
        self.eval()
        if self.conditions:
            self.conditions.eval()
        Xs = sp.csr_matrix(X) if not sp.issparse(X) else X.tocsr()
        csr = _hip.DeviceCSR(Xs, self.device)
        pred = _hip.HostRows(Xs.shape[0], Xs.shape[1], self.device)
        with torch.no_grad():
            for start in range(0, Xs.shape[0], self.batch_size):
                n = min(self.batch_size, Xs.shape[0] - start)
                cond = None
                c_batch = [_take(c, slice(start, start + n)) for c in condition_data] if use_condition else None
                if use_condition and not self._cond_native:
                    z = self.hip.vae_encode(csr, start, n, eps=self._eps(n), train=False)
                    pred.put(start, self.hip.decode(self.conditions.encode_impose(z, c_batch)))
                    continue
                if use_condition:
                    cond = self._cond(c_batch)
                pred.put(start, self.hip.vae_predict(csr, start, n, cond=cond, eps=self._eps(n)))
        return pred.numpy()


class VAERecommender(Recommender):
    """
    Varietional Autoencoder Recommender
    =====================================
    Keyword arguments are forwarded to VAE (n_hidden, n_code, n_epochs, batch_size, lr, normalize_inputs, verbose, ...).
    """

    def __init__(self, conditions=None, **kwargs):
        super().__init__()
        self.verbose = kwargs.get('verbose', True)
        self.conditions = conditions
        self.model_params = kwargs
        self.model = None

    def __str__(self):
        desc = "Variational Autoencoder"
        if self.conditions:
            desc += " conditioned on: " + ', '.join(self.conditions.keys())
        desc += '\nModel Params: ' + str(self.model_params)
        return desc

    def train(self, training_set):
        X = training_set.tocsr()
        if self.conditions:
            condition_data = self.conditions.fit_transform(training_set.get_attributes(self.conditions.keys()))
        else:
            condition_data = None
        self.model = VAE(X.shape[1], X.shape[1], conditions=self.conditions, **self.model_params)
        print(self)
        print(self.model)
        print(self.conditions)
        self.model.fit(X, condition_data=condition_data)

    def predict(self, test_set):
        X = test_set.tocsr()
        if self.conditions:
            # Important to not call fit here, but just transform
            condition_data = self.conditions.transform(test_set.get_attributes(self.conditions.keys()))
        else:
            condition_data = None
        return self.model.predict(X, condition_data=condition_data)
