"""Adversarially regularised autoencoder recommender on MI355X.

Same classes, constructor keywords and method semantics as reference aaerec/aae.py
(`AdversarialAutoEncoder` 589-870: fit / partial_fit / predict / train / eval / zero_grad;
`AAERecommender` 873-977: train / predict), but the step itself - Encoder / Decoder /
Discriminator forward and backward, BCE, the three losses and the four Adam optimisers - runs in
libaaerec_hip.so (hand-written gfx950 kernels, C ABI include/aaerec_hip.h) instead of
torch.nn modules.  This file is host-side orchestration only: shuffling, batching, condition
plugins, logging.  It fails loudly without the built library or without a GPU.

Build-only keyword arguments (all optional, defaults keep the reference behaviour):
    device      torch device string, default the current HIP device
    rng_mode    'device'    dropout masks and z_real from the counter generator in the kernels
                'reference' draw them on the host from torch's global CPU generator in exactly
                            the order the reference's modules do, so a run is comparable with
                            the reference step by step (slower: masks are uploaded every step)
    seed        seed of the device generator (default: drawn from torch's global generator)
    data_parallel  torch.distributed process group (or True for the default group): shard every
                batch over the ranks and all-reduce gradients over RCCL
    dtype       'f32' (default: the reference's arithmetic) or 'bf16': the GEMM-shaped products take bf16 matrix-core
                inputs with fp32 accumulation; master weights, optimiser state, losses stay fp32 (BASELINE config C2)
"""
import os

import numpy as np
import scipy.sparse as sp
import sklearn  # noqa: F401  (sklearn.utils.shuffle semantics are reproduced with np.random below)
import torch

from . import _hip
from .base import Recommender
from .condition import _check_conditions

torch.manual_seed(42)            # reference aae.py:27
TINY = 1e-12
STATUS_FORMAT = "[ R: {:.4f} | D: {:.4f} | G: {:.4f} ]"


def log_losses(*losses):
    print("\r" + STATUS_FORMAT.format(*losses), end="", flush=True)


def sample_categorical(size):
    batch_size, n_classes = size
    return torch.from_numpy(np.eye(n_classes)[np.random.randint(0, n_classes, batch_size)].astype("float32"))


def sample_bernoulli(size):
    return torch.from_numpy(np.random.randint(0, 1, size).astype("float32"))   # sic: always 0 (aae.py:86-88)


PRIOR_SAMPLERS = {"categorical": sample_categorical, "bernoulli": sample_bernoulli, "gauss": torch.randn}
PRIOR_ACTIVATIONS = {"categorical": "softmax", "bernoulli": "sigmoid", "gauss": "linear"}
TORCH_OPTIMIZERS = ("sgd", "adam")


class _NetView:
    """What the reference exposes as model.enc / .dec / .disc: here a named view of the kernel
    library's parameters with the nn.Module methods the drivers touch."""

    def __init__(self, owner, name):
        self._owner, self._name, self.training = owner, name, True

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        return None      # gradients never outlive the fused kernels

    def state_dict(self):
        self._owner._dp_settle()
        sd = self._owner.hip.state_dict()
        pre = self._name + "."
        return {k[len(pre):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if k.startswith(pre)}

    def load_state_dict(self, sd):
        self._owner.hip.load_params({self._name + "." + k: np.asarray(v) for k, v in sd.items()})


class _OptimView:
    def __init__(self, owner, name):
        self._owner, self._name = owner, name

    def state_dict(self):
        self._owner._dp_settle()
        return self._owner.hip.adam_state(self._name)


class AdversarialAutoEncoder:
    """ Adversarial Autoencoder """

    def __init__(self, n_hidden=100, n_code=50, gen_lr=0.001, reg_lr=0.001, prior="gauss", prior_scale=None,
                 batch_size=100, n_epochs=500, optimizer="adam", normalize_inputs=True, activation="ReLU",
                 dropout=(.2, .2), conditions=None, verbose=True,
                 device=None, rng_mode="device", seed=None, data_parallel=None, dp_mode="shard", dtype="f32"):
        self.prior = prior.lower()
        self.prior_scale = prior_scale
        self.prior_sampler = PRIOR_SAMPLERS[self.prior]
        self.encoder_activation = PRIOR_ACTIVATIONS[self.prior]
        self.optimizer = optimizer.lower()
        if self.optimizer not in TORCH_OPTIMIZERS:
            raise KeyError(self.optimizer)
        self.n_hidden, self.n_code = n_hidden, n_code
        self.gen_lr, self.reg_lr = gen_lr, reg_lr
        self.batch_size, self.n_epochs, self.verbose = batch_size, n_epochs, verbose
        self.normalize_inputs, self.dropout, self.activation = normalize_inputs, dropout, activation
        self.conditions = conditions
        self.enc = self.dec = self.disc = None
        self.enc_optim = self.dec_optim = self.gen_optim = self.disc_optim = None
        if rng_mode not in ("device", "reference"):
            raise ValueError("rng_mode must be 'device' or 'reference'")
        self.device, self.rng_mode, self.seed, self.data_parallel = device, rng_mode, seed, data_parallel
        if dp_mode not in ("vocab", "vocab_out", "replicated", "shard"):
            raise ValueError("dp_mode must be 'vocab', 'vocab_out', 'replicated' or 'shard'")
        # data_parallel (torch.distributed, one process per GPU): 'shard' (default, r4) gives every rank ONE handle with its
        # item slice of both vocabulary-wide matrices and a full copy of the hidden layers, runs the whole global batch
        # through it and all-reduces three [global batch, n_hidden] partial sums per step (parallel.ItemShardedAAE);
        # 'vocab' shards both vocabulary-wide matrices - the
        # decoder's output layer and the encoder's first layer - over the items (aaerec.parallel.VocabParallelAAE: the
        # ranks exchange [global batch, n_hidden] activations only), 'vocab_out' the output layer alone (the first
        # layer's row-sparse gradient travels as packed rows), 'replicated' keeps everything on every rank
        self.dp_mode = dp_mode
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' (the reference's arithmetic) or 'bf16' (bf16 matrix-core inputs, fp32 "
                             "accumulation, fp32 master weights and optimiser state)")
        self.dtype = dtype
        self._unfused_decoder = False                  # A/B switch of bench.py: keep the three-kernel output layer
        self.hip = None
        self._dp = None
        self._slice = self._slice_csr = self._g_rows = self._g_c_batch = None     # dp_mode='vocab': this rank's item slice of dec.lin3
        self.last_losses = None
        self._ae_only = False

    def __str__(self):
        desc = "Adversarial Autoencoder"
        desc += " ({}, {}, {}, {}, {})".format(self.n_hidden, self.n_hidden, self.n_code, self.n_hidden, self.n_hidden)
        desc += " optimized by " + self.optimizer
        desc += " with learning rates Gen, Reg = {}, {}".format(self.gen_lr, self.reg_lr)
        desc += ", using a batch size of {}".format(self.batch_size)
        desc += "\nMatching the {} distribution".format(self.prior)
        desc += " by {} activation.".format(self.encoder_activation)
        if self.conditions:
            desc += "\nConditioned on " + ", ".join(self.conditions.keys())
        return desc

    # ---- nn.Module-like switches (the kernels take the mode per call) ----------------------
    def eval(self):
        for net in (self.enc, self.dec, self.disc):
            if net is not None:
                net.eval()
        if self.conditions:
            self.conditions.eval()

    def train(self):
        for net in (self.enc, self.dec, self.disc):
            if net is not None:
                net.train()
        if self.conditions:
            self.conditions.train()

    def zero_grad(self):
        return None

    def _dp_settle(self):
        """Data parallel: block until a still-travelling exchange of updated parameters (the asynchronous all-gather
        of dec.lin3 rows in dp_mode='replicated') has landed, before anything outside the step reads them."""
        if self._dp is not None and hasattr(self._dp, "wait_pending"):
            self._dp.wait_pending()

    # ---- construction: the nets + 4 optimisers of the reference's fit() (aae.py:782-804) ----
    def _build(self, n_items, code_inc, max_row_nnz=None, w1_cap=None):
        dist = dist_group = None
        dist_world = 1
        if self.data_parallel is not None and self.data_parallel is not False:
            if hasattr(self.data_parallel, "all_gather_into_tensor"):
                dist = self.data_parallel                # an object with torch.distributed's collective interface
            else:
                import torch.distributed as dist
                if not dist.is_initialized():
                    raise RuntimeError("data_parallel needs torch.distributed to be initialised")
                dist_group = None if self.data_parallel is True else self.data_parallel
            dist_world = dist.get_world_size(dist_group)
        # nn.Linear default initialisation, drawn from torch's global CPU generator in the
        # reference's construction order (Encoder, Decoder, Discriminator; lin1, lin2, lin3 each),
        # so equal seeds give equal initial weights (the device generator's seed is drawn AFTER them)
        params = {}
        shapes = [("enc", (n_items, self.n_hidden, self.n_code)),
                  ("dec", (self.n_code + code_inc, self.n_hidden, n_items)),
                  ("disc", (self.n_code, self.n_hidden, 1))]
        for net, (n_in, n_hid, n_out) in shapes:
            for layer, (i, o) in enumerate(((n_in, n_hid), (n_hid, n_hid), (n_hid, n_out)), start=1):
                if net == "disc" and self._ae_only:
                    # the reference's AutoEncoder builds no discriminator (aae.py:367-383): do not
                    # consume the global generator for the (unused) one the kernels carry
                    with torch.random.fork_rng():
                        lin = torch.nn.Linear(i, o)
                else:
                    lin = torch.nn.Linear(i, o)
                params["{}.lin{}.weight".format(net, layer)] = lin.weight.detach().numpy()
                params["{}.lin{}.bias".format(net, layer)] = lin.bias.detach().numpy()
        seed = self.seed if self.seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) \
            if self.rng_mode == "device" else 0
        if dist is not None and dist_world > 1 and hasattr(dist, "broadcast"):
            # rank 0 is authoritative for everything the replicas must agree on: initial weights and the device
            # generator's seed (each process seeds torch's generator by itself; nothing else guarantees equality)
            from .parallel import broadcast_array
            dev = self.device if self.device is not None else "cuda:{}".format(torch.cuda.current_device())
            keys = sorted(params)
            flat = np.concatenate([params[k].ravel() for k in keys] + [np.array([seed], dtype=np.float64).view(np.float32)])
            flat = broadcast_array(dist, dist_group, flat.astype(np.float32, copy=False), dev)
            off = 0
            for k in keys:
                n = params[k].size
                params[k] = flat[off:off + n].reshape(params[k].shape).copy()
                off += n
            seed = int(flat[off:off + 2].view(np.float64)[0])
        # dp_mode='shard' (the default): the full-vocabulary handle only receives the trained model (predict, state_dict);
        # conditions the kernels cannot train themselves fall back to the replicated scheme, as under 'vocab'
        shard = dist is not None and self.dp_mode == "shard" and self._vocab_sharded(code_inc)
        self.hip = _hip.HipAAE(
            n_items, self.n_hidden, self.n_code, cond_inc=code_inc, max_batch=self.batch_size,
            max_nnz=None if max_row_nnz is None else self.batch_size * max(1, int(max_row_nnz)),
            activation=self.activation, prior=self.prior, prior_scale=self.prior_scale, optimizer=self.optimizer,
            normalize_inputs=self.normalize_inputs, dropout=self.dropout, gen_lr=self.gen_lr, reg_lr=self.reg_lr,
            rng_mode="device" if self.rng_mode == "device" else "inject", seed=seed,
            grad_mode="export" if dist is not None and not shard else "fused", device=self.device,
            dp_world=dist_world, w1_cap=w1_cap, ae_only=self._ae_only, dtype=self.dtype,
            unfused_decoder=self._unfused_decoder, dense_noise=getattr(self, "_dense_noise", False),
            # batches of 113..1664 rows: the output layer as row blocks of the fused kernel (one critical launch for all
            # blocks + one deferred optimiser launch, both on the bf16-emulated product: 0.70 -> 0.64 ms/step at 208 rows,
            # 0.93 -> 0.78 at 512); beyond 16 blocks the library takes the three streaming GEMMs
            blocked_output=16 * 7 < self.batch_size <= 16 * 104 and os.environ.get("AAE_SLICE_THREE_KERNEL") is None)
        self.hip.load_params(params)
        self.enc, self.dec, self.disc = (_NetView(self, n) for n in ("enc", "dec", "disc"))
        self.enc_optim, self.dec_optim = _OptimView(self, "enc"), _OptimView(self, "dec")
        self.gen_optim, self.disc_optim = _OptimView(self, "gen"), _OptimView(self, "disc")
        if dist is not None:
            from .parallel import DataParallelAAE, VocabParallelAAE, ItemShardedAAE, item_items
            if self._vocab_sharded(code_inc):
                # this rank's items: rank, rank + world, ... (the vocabulary is frequency-sorted: interleaving gives
                # every rank the same share of each batch's entries)
                items = item_items(n_items, dist.get_rank(dist_group), dist_world, True)
                sl_params = dict(params)
                sl_params["dec.lin3.weight"] = np.ascontiguousarray(params["dec.lin3.weight"][items])
                sl_params["dec.lin3.bias"] = np.ascontiguousarray(params["dec.lin3.bias"][items])
                sl_params["enc.lin1.weight"] = np.ascontiguousarray(params["enc.lin1.weight"][:, items])
                self._slice = _hip.HipAAE(
                    len(sl_params["dec.lin3.bias"]), self.n_hidden, self.n_code, cond_inc=code_inc, max_batch=self.batch_size,
                    max_nnz=None if max_row_nnz is None else self.batch_size * max(1, int(max_row_nnz)),
                    activation=self.activation, prior=self.prior, prior_scale=self.prior_scale,
                    optimizer=self.optimizer, normalize_inputs=self.normalize_inputs, dropout=self.dropout,
                    gen_lr=self.gen_lr, reg_lr=self.reg_lr, rng_mode="device" if self.rng_mode == "device" else "inject",
                    seed=seed, device=self.device, ae_only=self._ae_only, dtype=self.dtype,
                    unfused_decoder=self._unfused_decoder,
                    # the slice sees world x batch rows: beyond one fused launch's 112 they run as row blocks of the
                    # split form (one critical launch for all blocks, one deferred optimiser launch), DESIGN.md 7.3
                    # (with the output layer ALONE sharded the packed-row exchange of the first layer reads the slice's
                    #  tensors between the phases and would wait for the deferred launch: three GEMMs stay faster there)
                    blocked_output=self.dp_mode in ("vocab", "shard") and os.environ.get("AAE_SLICE_THREE_KERNEL") is None)
                self._slice.load_params(sl_params)
                if shard:
                    # ONE training handle per rank: its item slice of both vocabulary-wide layers + the hidden layers, the
                    # whole global batch through it, three all-reduces of partial sums per step (parallel.ItemShardedAAE)
                    # (dp_collectives = 'ipc', an attribute like _unfused_decoder: the three all-reduces as one-shot launches over
                    #  peer-mapped mailboxes - parallel.ipc_collectives, the ranks of one node)
                    self._dp = ItemShardedAAE(self.hip, self._slice, dist, n_items, group=dist_group, interleaved=True,
                                              collectives=getattr(self, "dp_collectives", None), max_rows=self.batch_size)
                else:
                    self._dp = VocabParallelAAE(self.hip, self._slice, dist, n_items, group=dist_group,
                                                shard_first_layer=self.dp_mode == "vocab", interleaved=True)
            else:
                self._dp = DataParallelAAE(self.hip, dist, group=dist_group)

    def _vocab_sharded(self, code_inc):
        """dp_mode='vocab' applies when the step has no cut at the condition boundary (no conditions, constant
        concatenated blocks, CategoricalConditions the kernels train themselves) and the batches are the corpus' own
        rows (no per-epoch corruption hook)."""
        if self.dp_mode not in ("vocab", "vocab_out", "shard") or type(self)._epoch_csr is not AdversarialAutoEncoder._epoch_csr:
            return False
        return not self.conditions or code_inc == 0 or self._is_constant_concat() or self._is_device_native()

    # ---- randomness in the reference's draw order (rng_mode='reference') -------------------
    def _host_randomness(self, B):
        h, p1, p2 = self.n_hidden, self.dropout[0], self.dropout[1]

        def mask(p):
            return None if p == 0 else torch.empty(B, h).bernoulli_(1 - p).to(torch.uint8)
        masks = [mask(p1), mask(p2), mask(p1), mask(p2)]                 # ae_step: enc, dec
        z_real = self.prior_sampler((B, self.n_code))                    # disc_step (aae.py:716)
        masks += [mask(p1), mask(p2), mask(p1), mask(p2)]                # D(z_real), D(z_fake)
        masks += [mask(p1), mask(p2), mask(p1), mask(p2)]                # gen_step: enc, disc
        return masks, z_real.to(torch.float32)

    def _is_constant_concat(self):
        return all(getattr(c, "constant_concat", False) for c in self.conditions.values())

    def _is_device_native(self):
        """Every condition is a concatenated block the kernels can produce and train themselves: a constant block or
        a CategoricalCondition with its table on this GPU."""
        dev = self.hip.device if self.hip is not None else \
            torch.device(self.device if self.device is not None else "cuda:{}".format(torch.cuda.current_device()))
        return all(getattr(c, "constant_concat", False) or (hasattr(c, "device_native") and c.device_native(dev))
                   for c in self.conditions.values())

    def _native_cond_block(self, c_batch, n_rows, whole_batch=None):
        """[n_rows, size_increment] block of the encoded conditions, in ConditionList order (encode_impose's
        concatenation order, condition.py:90-99).  whole_batch: the condition inputs of the global batch this is a
        share of (data parallel) - value lists are padded to ITS width, as the single process would."""
        dev = self.hip.device
        block = torch.empty(n_rows, self.hip.cond_inc, dtype=torch.float32, device=dev)
        off = 0
        for j, (cond, x) in enumerate(zip(self.conditions.values(), c_batch)):
            w = cond.size_increment()
            if getattr(cond, "constant_concat", False):
                block[:, off:off + w] = _hip.upload(cond.encode(x), dev)
            else:
                cond.encode_into(block[:, off:off + w], x,
                                 width=None if whole_batch is None else cond.padded_width(whole_batch[j]))
            off += w
        return block

    def _native_cond_update(self, n_rows, dblock=None, whole_batch=None):
        """conditions.zero_grad / backward / step (aae.py:699-709) from the step's dL/d(condition block); data
        parallel: from the gathered gradient of the whole batch and its inputs, identically on every rank."""
        if dblock is None:
            dblock = self.hip.cond_grad(n_rows)
        off = 0
        for j, cond in enumerate(self.conditions.values()):
            w = cond.size_increment()
            if not getattr(cond, "constant_concat", False):
                cond.update_from(dblock[:, off:off + w], None if whole_batch is None else whole_batch[j])
            off += w

    def _run_step(self, csr, row_start, n_rows, rows, c_batch):
        """One partial_fit on rows of a device-resident CSR."""
        masks = z_real = None
        if self.rng_mode == "reference":
            masks, z_real = self._host_randomness(n_rows)
            if self._dp is not None and hasattr(self._dp, "agree_randomness"):
                # dp_mode='shard': every rank runs the whole batch through its own copy of the hidden layers - rank 0's draws
                masks, z_real = self._dp.agree_randomness(masks, z_real)
        hip = self.hip
        use_condition = c_batch is not None
        if self._dp is not None and self._slice is not None:
            if self._g_rows is None:
                raise NotImplementedError("partial_fit on a rank's own batch is not available with dp_mode='vocab' (the "
                                          "output-layer slices need the global batch): use fit() or dp_mode='replicated'")
            cond, trainable = None, False
            if use_condition:
                trainable = not self._is_constant_concat()
                cond = self._native_cond_block(c_batch, n_rows, whole_batch=self._g_c_batch if trainable else None)
            self._dp.step(csr, row_start, n_rows, self._slice_csr, 0, self._dp.global_rows, rows=rows, g_rows=self._g_rows,
                          cond=cond, masks=masks, z_real=z_real)
            if trainable:
                # every rank gathers dL/d(condition block) of the whole batch and applies the identical update
                # (dp_mode='shard': the training handle saw the whole batch - nothing to gather)
                src = self._slice if self.dp_mode == "shard" else hip
                self._native_cond_update(n_rows, self._dp.gather_rows(src.cond_grad(n_rows)), self._g_c_batch)
        elif self._dp is not None:
            cond_fn = self._cond_fn(c_batch) if use_condition else None
            self._dp.step(csr, row_start, n_rows, global_rows=getattr(self._dp, "global_rows", None), rows=rows,
                          cond_fn=cond_fn, masks=masks, z_real=z_real)
        elif not use_condition:
            hip.step(csr, row_start, n_rows, rows=rows, masks=masks, z_real=z_real)
        elif self._is_constant_concat():
            blocks = [_hip.upload(c.encode(x), hip.device) for c, x in zip(self.conditions.values(), c_batch)]
            # (one block: as it is - torch.cat of a single tensor is a copy launch of its own, every step)
            hip.step(csr, row_start, n_rows, rows=rows, cond=blocks[0] if len(blocks) == 1 else torch.cat(blocks, 1),
                     masks=masks, z_real=z_real)
        elif self._is_device_native():
            hip.step(csr, row_start, n_rows, rows=rows, cond=self._native_cond_block(c_batch, n_rows), masks=masks,
                     z_real=z_real)
            self._native_cond_update(n_rows)
        else:
            z = hip.ae_encode(csr, row_start, n_rows, rows=rows, masks=masks, z_real=z_real)
            zc, back = self._cond_fn(c_batch)(z)
            hip.ae_encoder_backward(back(hip.ae_decode_backward(zc)))
            if not self._ae_only:
                hip.disc_gen()

    def _cond_fn(self, c_batch):
        """z -> (zc, backward) through the condition plugins with torch autograd
        (aae.py:688-690, 699-700, 708-709)."""
        conditions, dp = self.conditions, self._dp

        def fn(z):
            z = z.detach().requires_grad_(True)
            zc = conditions.encode_impose(z, c_batch)

            def back(dzc):
                conditions.zero_grad()
                zc.backward(dzc.to(zc.device))
                if dp is not None:
                    dp.sync_conditions(conditions)       # data parallel: the plugins' gradients summed over ranks
                conditions.step()
                return z.grad
            return zc, back
        return fn

    # ---- public API ------------------------------------------------------------------------
    def partial_fit(self, X, y=None, condition_data=None, step=None):
        """ Performs reconstrction, discimination, generator training steps on one dense batch """
        if y is not None:
            raise NotImplementedError("(Semi-)supervised usage not supported")
        use_condition = _check_conditions(self.conditions, condition_data)
        if self.hip is None:
            self._build(X.shape[1], self.conditions.size_increment() if use_condition else 0)
        if X.shape[0] > self.hip.max_batch:
            raise ValueError("batch of {} rows exceeds batch_size={}".format(X.shape[0], self.hip.max_batch))
        if sp.issparse(X):
            Xs = X.tocsr()
            Xs.sum_duplicates()                         # a non-canonical CSR hides a target of 2.0 in two entries
            _validate_targets(Xs)
            csr = _hip.DeviceCSR(Xs, self.hip.device)
        else:
            # the reference's call form: the dense batch of X_shuf[start:end].toarray() (aae.py:823).  It crosses PCIe as
            # it is and becomes CSR on the device (target validation included)
            csr = _hip.DeviceCSR.from_dense(np.asarray(X), self.hip.device, self.hip.cfg.max_nnz)
        self.train()
        self._run_step(csr, 0, X.shape[0], None, condition_data if use_condition else None)
        if self.verbose:
            self.last_losses = self.hip.losses()
            log_losses(*self.last_losses)
        return self

    def fit(self, X, y=None, condition_data=None):
        if y is not None:
            raise NotImplementedError("(Semi-)supervised usage not supported")
        for _step in self.fit_steps(X, condition_data=condition_data):
            pass
        return self

    def fit_steps(self, X, condition_data=None, n_epochs=None):
        """fit() as a generator: builds the model, uploads the corpus and yields after every partial_fit of the epoch
        loop (reference aae.py:808-831), so a caller can meter the loop (bench.py times exactly K of its steps, host
        batch assembly included).  Exhausting it is fit()."""
        use_condition = _check_conditions(self.conditions, condition_data)
        code_inc = self.conditions.size_increment() if use_condition else 0
        print(("Using condition, code size:" if use_condition else "Not using condition, code size:"),
              self.n_code + code_inc)
        X = X.tocsr()
        if not X.has_canonical_format:
            X = X.copy()
            X.sum_duplicates()
        _validate_targets(X)
        row_nnz = np.sort(X.getnnz(1))[::-1]
        # most distinct items any batch can touch: the batch_size longest rows (bounds the packed
        # first-layer gradient a data-parallel rank sends)
        w1_cap = int(min(X.shape[1], max(1, row_nnz[:self.batch_size].sum())))
        self._build(X.shape[1], code_inc, max_row_nnz=max(int(row_nnz[0]) if X.shape[0] else 1, 4096), w1_cap=w1_cap)
        csr0 = _hip.DeviceCSR(X, self.hip.device)      # the corpus stays resident in HBM
        self._fit_csr = csr0
        self._fit_X = X                                 # (host copy; subclasses with host-side randomness use it)
        row_len = X.getnnz(1)
        if self._slice is not None:                     # this rank's items of the corpus, ids rebased to the slice
            self._slice_csr = _hip.DeviceCSR(X[:, self._dp.items].tocsr(), self.hip.device)
            if self._dp.shard_first:                    # the slice holds its columns only: F.normalize's whole-row norms
                self._slice.set_doc_l1(torch.as_tensor(np.asarray(abs(X).sum(1), dtype=np.float32).reshape(-1),
                                                       device=self.hip.device))
        n_docs = X.shape[0]
        self.train()
        step = 0
        for epoch in range(self.n_epochs if n_epochs is None else n_epochs):
            if self.verbose:
                print("Epoch", epoch + 1)
            # sklearn.utils.shuffle(X, *condition_data) == one permutation from np.random's
            # global state applied to every array (aae.py:813-817); only the permutation moves
            perm = np.arange(n_docs)
            np.random.shuffle(perm)
            if self._dp is not None and self._dp.world > 1 and hasattr(self._dp.dist, "broadcast"):
                # every rank walks rank 0's permutation (np.random's global state is per process and nothing else
                # makes the ranks' draws agree; in dp_mode='vocab' a rank pairs OTHER ranks' hidden activations with
                # targets taken from this order)
                from .parallel import broadcast_array
                perm = broadcast_array(self._dp.dist, self._dp.group, perm.astype(np.int64), self.hip.device)
            perm_dev = torch.as_tensor(perm.astype(np.int32), device=self.hip.device)
            # content id of this epoch's row ids (aae_batch.generation): every window of perm_dev below carries it, so the
            # batch a step runs is matched with the one named ahead by content, not by the address the allocator gave it
            perm_gen = next(_hip._GENERATION)
            perm_i64 = perm_dev.to(torch.int64) if use_condition else None      # (what index_select wants: once per epoch, not per step)

            def window(a, b):
                w = _hip.row_ids(perm_dev[a:b], perm_gen)
                if perm_i64 is not None:
                    w._aae_i64 = perm_i64[a:b]
                return w
            csr = self._epoch_csr(csr0)
            for start in range(0, n_docs, self.batch_size):
                stop = min(start + self.batch_size, n_docs)
                if self._dp is not None:
                    # every rank walks the same permutation (same np.random state) and takes its
                    # contiguous share of the global batch; a tail batch with fewer rows than
                    # ranks is skipped on all ranks
                    if self._slice is not None and self.dp_mode != "shard":
                        # vocabulary-sharded output layer: equal shares (a tail batch loses < world documents)
                        stop = start + (stop - start) // self._dp.world * self._dp.world
                    lo, hi = self._dp.shard(start, stop)
                    if lo is None:
                        continue
                    self._dp.global_rows = stop - start
                    if self._slice is not None and self._dp.shard_first:
                        # the slice model's next (global) batch: its distinct items and their deferred-Adam catch-up
                        # run behind this step's deferred optimiser launch (_hip.prefetch)
                        gn = start + self.batch_size
                        gs = min(gn + self.batch_size, n_docs)
                        if self.dp_mode != "shard":
                            gs = gn + (gs - gn) // self._dp.world * self._dp.world
                        if gs > gn:
                            self._slice.prefetch(self._slice_csr, 0, gs - gn, window(gn, gs))
                    # first-layer packets: no share of this batch names more distinct items than it has entries
                    shares = np.array_split(row_len[perm[start:stop]], self._dp.world)
                    self._dp.w1_rows = int(max(sh.sum() for sh in shares)) + 8
                    if self._slice is not None:
                        self._g_rows = window(start, stop)
                        if use_condition and not self._is_constant_concat():
                            self._g_c_batch = [_take(c, perm[start:stop]) for c in condition_data]
                    start, stop = lo, hi
                rows = window(start, stop)
                nxt = start + self.batch_size           # (single process) the batch after this one: see _hip.prefetch
                if self._dp is None and nxt < n_docs:
                    self.hip.prefetch(csr, 0, min(nxt + self.batch_size, n_docs) - nxt,
                                      window(nxt, min(nxt + self.batch_size, n_docs)))
                c_batch = None
                if use_condition:
                    idx = perm[start:stop]
                    c_batch = [_take(c, idx, rows) for c in condition_data]
                self._run_step(csr, 0, int(rows.numel()), rows, c_batch)
                if self.verbose:
                    self.last_losses = self._losses()
                    log_losses(*self.last_losses)
                step += 1
                yield step
            if self.verbose:
                print()
        self._fit_finish()

    def _fit_finish(self):
        """What fit() does after its last step (also for a caller that stops fit_steps() early)."""
        self._dp_settle()                               # replicated scheme: the last step's parameter all-gather
        self.last_losses = self._losses()
        self._g_rows = None
        if self._slice is not None:
            self._dp.gather_output_layer()              # every replica ends with the whole decoder (predict, state_dict)

    def _losses(self):
        """(recon, disc, gen) of the last step; under dp_mode='vocab' the reconstruction loss lives in the item slices
        (a collective: every rank calls this at the same points)."""
        losses = (self._slice if self.dp_mode == "shard" and self._slice is not None else self.hip).losses()
        if self._slice is not None:
            losses = (self._dp.recon_loss(),) + tuple(losses[1:])
        return losses

    def _epoch_csr(self, csr):
        """The resident corpus as this epoch's batches see it (hook: DenoisingAutoEncoder thins it)."""
        return csr

    def predict(self, X, condition_data=None):
        self.eval()
        use_condition = _check_conditions(self.conditions, condition_data)
        if self.conditions:
            self.conditions.eval()
        if sp.issparse(X):
            Xs = X.tocsr()
            csr = _hip.DeviceCSR(Xs, self.hip.device)
        else:                                           # dense test matrix: compacted on the device, batch-size rows at a time
            Xs = np.asarray(X)
            csr = None
        self._dp_settle()
        if self._slice is not None:
            self._dp.gather_output_layer()              # (a no-op after fit(); collective otherwise)
        fused = (not use_condition) or self._is_constant_concat()
        native = use_condition and not fused and self._is_device_native()
        pred = _hip.HostRows(Xs.shape[0], Xs.shape[1], self.hip.device)
        with torch.no_grad():
            for start in range(0, Xs.shape[0], self.batch_size):
                n = min(self.batch_size, Xs.shape[0] - start)
                c_batch = [_take(c, slice(start, start + n)) for c in condition_data] if use_condition else None
                if not sp.issparse(X):
                    dcsr, dstart = _hip.DeviceCSR.from_dense(Xs[start:start + n], self.hip.device, self.hip.cfg.max_nnz), 0
                else:
                    dcsr, dstart = csr, start
                if fused:
                    cond = None
                    if use_condition:
                        cond = torch.cat([_hip.upload(c.encode(x), self.hip.device)
                                          for c, x in zip(self.conditions.values(), c_batch)], 1)
                    out = self.hip.predict(dcsr, dstart, n, cond=cond)
                elif native:
                    out = self.hip.predict(dcsr, dstart, n, cond=self._native_cond_block(c_batch, n))
                else:
                    z = self.hip.encode(dcsr, dstart, n)
                    out = self.hip.decode(self.conditions.encode_impose(z, c_batch))
                pred.put(start, out)
        return pred.numpy()


class AutoEncoder(AdversarialAutoEncoder):
    """The reference's plain (non-adversarial) autoencoder, aae.py:221-458: the same encoder and
    decoder trained with the reconstruction step only, one learning rate for both optimisers.
    Same kernels, the discriminator / generator phases are simply not run."""

    def __init__(self, n_hidden=100, n_code=50, lr=0.001, batch_size=100, n_epochs=500, optimizer="adam",
                 normalize_inputs=True, activation="ReLU", dropout=(.2, .2), conditions=None, verbose=True,
                 device=None, rng_mode="device", seed=None, data_parallel=None, dp_mode="shard"):
        super().__init__(n_hidden=n_hidden, n_code=n_code, gen_lr=lr, reg_lr=lr, prior="gauss", batch_size=batch_size,
                         n_epochs=n_epochs, optimizer=optimizer, normalize_inputs=normalize_inputs,
                         activation=activation, dropout=dropout, conditions=conditions, verbose=verbose,
                         device=device, rng_mode=rng_mode, seed=seed, data_parallel=data_parallel, dp_mode=dp_mode)
        self.lr = lr
        self._ae_only = True

    def __str__(self):
        return "Autoencoder ({0}, {0}, {1}, {0}, {0}) optimized by {2} with learning rate {3}".format(
            self.n_hidden, self.n_code, self.optimizer, self.lr)

    def _host_randomness(self, B):
        h, p1, p2 = self.n_hidden, self.dropout[0], self.dropout[1]

        def mask(p):
            return None if p == 0 else torch.empty(B, h).bernoulli_(1 - p).to(torch.uint8)
        return [mask(p1), mask(p2), mask(p1), mask(p2)] + [None] * 8, None


def _predict_topk(self, X, k=10, condition_data=None, exclude_known=True):
    """Top-k recommendations without materialising the [n, N] score matrix on the host: the
    reference's predict -> remove_non_missing -> argtopk pipeline (aae.py:840-870,
    evaluation.py:183-199, 20-58) with only [n, k] ids and scaled scores crossing PCIe.
    Conditions as in predict(): constant concatenation and the device-native CategoricalCondition ride in the fused
    call, any other plugin imposes itself on the code between aae_encode and aae_decode_topk."""
    self.eval()
    use_condition = _check_conditions(self.conditions, condition_data)
    if self.conditions:
        self.conditions.eval()
    fused = (not use_condition) or self._is_constant_concat()
    native = use_condition and not fused and self._is_device_native()
    Xs = sp.csr_matrix(X) if not sp.issparse(X) else X.tocsr()
    csr = _hip.DeviceCSR(Xs, self.hip.device)
    self._dp_settle()
    if self._slice is not None:
        self._dp.gather_output_layer()
    ids, vals = [], []
    # rows per call: what one fused predict -> rank launch takes (aae_rank_max_rows: hundreds to thousands of rows, the
    # parameter stream of dec.lin3 is read once per call), the training batch size otherwise
    chunk = max(self.batch_size, min(self.hip.rank_max_rows(k), 2048))
    for start in range(0, Xs.shape[0], chunk):
        n = min(chunk, Xs.shape[0] - start)
        cond = None
        c_batch = [_take(c, slice(start, start + n)) for c in condition_data] if use_condition else None
        if use_condition and fused:
            cond = torch.cat([_hip.upload(c.encode(x), self.hip.device) for c, x in zip(self.conditions.values(), c_batch)], 1)
        elif native:
            cond = self._native_cond_block(c_batch, n)
        if fused or native:
            i, v = self.hip.predict_topk(csr, start, n, k, cond=cond, exclude_known=exclude_known)
        else:
            with torch.no_grad():
                # (aae_encode takes the handle's per-batch buffers: batch_size rows at a time; the ranking is one call)
                z = torch.cat([self.hip.encode(csr, s0, min(self.batch_size, start + n - s0))
                               for s0 in range(start, start + n, self.batch_size)])
                i, v = self.hip.decode_topk(self.conditions.encode_impose(z, c_batch), csr, start, k, exclude_known=exclude_known)
        ids.append(i)
        vals.append(v)
    return torch.cat(ids).cpu().numpy(), torch.cat(vals).cpu().numpy()


AdversarialAutoEncoder.predict_topk = _predict_topk


def _take(c, idx, rows_dev=None):
    """Row selection on whatever a condition's transform produced (ndarray, sparse, list, tensor).  rows_dev: the same
    selection as a device tensor - condition data that already lives in HBM is then selected there (indexing a device
    tensor with a host array is a blocking copy on the compute stream: host and GPU in lock-step, every step)."""
    if rows_dev is not None and torch.is_tensor(c) and c.is_cuda:
        # (the epoch's permutation is kept as int64 as well, fit_steps' window(): no cast launch per step)
        i64 = getattr(rows_dev, "_aae_i64", None)
        return c.index_select(0, i64 if i64 is not None else rows_dev.to(torch.int64))
    if isinstance(c, (list, tuple)):
        if isinstance(idx, slice):
            return list(c[idx])
        return [c[i] for i in idx]
    return c[idx]


class DecodingRecommender(Recommender):
    """ Only the decoder part of the AAE, basically 2-MLP (reference aae.py:461-584): the encoded conditions
    (the first one as is, the others imposed on it) -> Decoder -> BCE against the item rows.  The decoder runs
    on the HIP kernels (aae_decoder_step / aae_decode); the condition plugins stay torch modules and receive
    dL/d(inputs) through autograd, then take their own optimiser step. """

    def __init__(self, conditions, n_epochs=100, batch_size=100, optimizer='adam', n_hidden=100, lr=0.001,
                 verbose=True, device=None, rng_mode="device", seed=None, **mlp_params):
        super().__init__()
        self.n_epochs = n_epochs
        self.batch_size = batch_size
        self.lr = lr
        self.optimizer = optimizer.lower()
        self.model_params = mlp_params
        self.verbose = verbose
        self.n_hidden = n_hidden
        assert len(conditions), "Minimum 1 condition is necessary for MLP"
        self.conditions = conditions
        unknown = set(mlp_params) - {"dropout", "activation"}
        if unknown:        # the reference forwards **mlp_params to Decoder(...), which accepts exactly these
            raise TypeError("__init__() got an unexpected keyword argument '{}'".format(sorted(unknown)[0]))
        self.dropout = tuple(mlp_params.get("dropout", (.2, .2)))
        self.activation = mlp_params.get("activation", "ReLU")
        if rng_mode not in ("device", "reference"):
            raise ValueError("rng_mode must be 'device' or 'reference'")
        self.device, self.rng_mode, self.seed = device, rng_mode, seed
        self.hip = None
        self.mlp, self.mlp_optim, self.vect = None, None, None
        self.last_loss = None

    def __str__(self):
        desc = "MLP-2 Decoder with " + str(self.n_hidden) + " hidden units"
        desc += " training for " + str(self.n_epochs)
        desc += " optimized by " + self.optimizer
        desc += " with learning rate " + str(self.lr)
        desc += " with %d conditions: %s " % (len(self.conditions), ', '.join(self.conditions.keys()))
        desc += "\n MLP Params: " + str(self.model_params)
        return desc

    # ---- internals ---------------------------------------------------------------------------
    def _build(self, n_items, max_row_nnz=None):
        n_in = int(self.conditions.size_increment())
        # Decoder(size_increment, n_hidden, n_items): nn.Linear default init in construction order (aae.py:521-524);
        # the device generator's seed is drawn after them, so equal torch seeds give the reference's initial weights
        params = {}
        for layer, (i, o) in enumerate(((n_in, self.n_hidden), (self.n_hidden, self.n_hidden),
                                        (self.n_hidden, n_items)), start=1):
            lin = torch.nn.Linear(i, o)
            params["dec.lin{}.weight".format(layer)] = lin.weight.detach().numpy()
            params["dec.lin{}.bias".format(layer)] = lin.bias.detach().numpy()
        seed = self.seed if self.seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) \
            if self.rng_mode == "device" else 0
        self.hip = _hip.HipAAE(
            n_items, self.n_hidden, n_in, cond_inc=0, max_batch=self.batch_size,
            max_nnz=None if max_row_nnz is None else self.batch_size * max(1, int(max_row_nnz)),
            activation=self.activation, optimizer=self.optimizer, dropout=self.dropout, gen_lr=self.lr,
            reg_lr=self.lr, rng_mode="device" if self.rng_mode == "device" else "inject", seed=seed,
            device=self.device)
        self.hip.load_params(params)
        self.mlp, self.mlp_optim = _NetView(self, "dec"), _OptimView(self, "dec")

    def _dp_settle(self):
        return None      # single device: nothing travels

    def _inputs(self, condition_data):
        """ Encode ALL condition data with the respective condition, start with the first encoded condition
        and impose all remaining ones (aae.py:494-502, 570-576) """
        encoded = self.conditions.encode(condition_data)
        inputs = encoded[0]
        for cond, cdata in zip(list(self.conditions.values())[1:], encoded[1:]):
            inputs = cond.impose(inputs, cdata)
        return inputs

    def _masks(self, B):
        if self.rng_mode != "reference":
            return None
        return [None if p == 0 else torch.empty(B, self.n_hidden).bernoulli_(1 - p).to(torch.uint8)
                for p in self.dropout]

    def _step(self, csr, n_rows, rows, condition_data):
        self.conditions.train()
        inputs = self._inputs(condition_data)
        masks = self._masks(n_rows)
        dz = self.hip.decoder_step(csr, 0, n_rows, inputs, rows=rows, masks=masks, want_grad=inputs.requires_grad)
        self.conditions.zero_grad()
        if inputs.requires_grad:
            inputs.backward(dz.to(inputs.device))
        self.conditions.step()

    # ---- public API ----------------------------------------------------------------------------
    def partial_fit(self, condition_data, y, step=None):
        ys = sp.csr_matrix(y.numpy() if torch.is_tensor(y) else y) if not sp.issparse(y) else y.tocsr()
        if self.hip is None:
            self._build(ys.shape[1])
        _validate_targets(ys)
        if ys.shape[0] > self.hip.max_batch:
            raise ValueError("batch of {} rows exceeds batch_size={}".format(ys.shape[0], self.hip.max_batch))
        csr = _hip.DeviceCSR(ys, self.hip.device)
        self._step(csr, ys.shape[0], None, condition_data)
        if self.verbose:
            self.last_loss = self.hip.losses()[0]
            print("\rLoss: {}".format(self.last_loss), flush=True, end='')
        return self

    def fit(self, condition_data, Y):
        Y = Y.tocsr() if sp.issparse(Y) else sp.csr_matrix(Y)
        _validate_targets(Y)
        row_nnz = Y.getnnz(1)
        self._build(Y.shape[1], max_row_nnz=max(int(row_nnz.max()) if Y.shape[0] else 1, 4096))
        csr = _hip.DeviceCSR(Y, self.hip.device)       # the targets stay resident in HBM
        n_docs = Y.shape[0]
        step = 0
        for __epoch in range(self.n_epochs):
            # sklearn.utils.shuffle(Y, *condition_data): one permutation for all arrays (aae.py:530)
            perm = np.arange(n_docs)
            np.random.shuffle(perm)
            perm_dev = torch.as_tensor(perm.astype(np.int32), device=self.hip.device)
            for start in range(0, n_docs, self.batch_size):
                idx = perm[start:start + self.batch_size]
                c_batch = [_take(c, idx) for c in condition_data]
                self._step(csr, len(idx), perm_dev[start:start + len(idx)], c_batch)
                if self.verbose:
                    self.last_loss = self.hip.losses()[0]
                    print("\rLoss: {}".format(self.last_loss), flush=True, end='')
                step += 1
            if self.verbose:
                print()
        self.last_loss = self.hip.losses()[0]
        return self

    def train(self, training_set):
        # Fit function from condition to X
        Y = training_set.tocsr()
        condition_data_raw = training_set.get_attributes(self.conditions.keys())
        condition_data = self.conditions.fit_transform(condition_data_raw)
        self.fit(condition_data, Y)

    def _predict_conditions(self, condition_data, n_users):
        self.conditions.eval()
        batch_results = _hip.HostRows(n_users, self.hip.N, self.hip.device)
        with torch.no_grad():
            for start in range(0, n_users, self.batch_size):
                c_batch = [_take(c, slice(start, start + self.batch_size)) for c in condition_data]
                batch_results.put(start, self.hip.decode(self._inputs(c_batch)))
        return batch_results.numpy()

    def predict(self, test_set):
        n_users = test_set.size(0)
        condition_data_raw = test_set.get_attributes(self.conditions.keys())
        condition_data = self.conditions.transform(condition_data_raw)
        return self._predict_conditions(condition_data, n_users)


def _validate_targets(X):
    """The reference's F.binary_cross_entropy rejects targets outside [0,1] (duplicate items in a
    bag give 2.0 after tocsr()); keep that error behaviour."""
    if X.nnz and (X.data.max() > 1 or X.data.min() < 0):
        raise RuntimeError("all elements of target should be between 0 and 1")


class AAERecommender(Recommender):
    """
    Adversarially Regularized Recommender
    =====================================
    Keyword arguments are forwarded to AdversarialAutoEncoder (n_hidden, n_code, n_epochs,
    batch_size, gen_lr, reg_lr, prior, activation, dropout, normalize_inputs, verbose, ...).
    """

    def __init__(self, adversarial=True, conditions=None, **kwargs):
        super().__init__()
        self.verbose = kwargs.get("verbose", True)
        self.conditions = conditions
        self.model_params = kwargs
        self.adversarial = adversarial
        self.model = None

    def __str__(self):
        desc = "Adversarial Autoencoder" if self.adversarial else "Autoencoder"
        if self.conditions:
            desc += " conditioned on: " + ", ".join(self.conditions.keys())
        desc += "\nModel Params: " + str(self.model_params)
        return desc

    def train(self, training_set):
        print(self)
        X = training_set.tocsr()
        if self.conditions:
            print("Fit transforming conditions:", self.conditions)
            condition_data = self.conditions.fit_transform(training_set.get_attributes(self.conditions.keys()))
        else:
            print("Start of training, not using condition...", self.conditions)
            condition_data = None
        if self.adversarial:
            self.model = AdversarialAutoEncoder(conditions=self.conditions, **self.model_params)
        else:
            self.model = AutoEncoder(conditions=self.conditions, **self.model_params)
        print(self.model)
        print(self.conditions)
        self.model.fit(X, condition_data=condition_data)

    def predict(self, test_set):
        X = test_set.tocsr()
        condition_data = None
        if self.conditions:
            condition_data = self.conditions.transform(test_set.get_attributes(self.conditions.keys()))
        return self.model.predict(X, condition_data=condition_data)

    def predict_topk(self, test_set, k=10):
        """(item ids [n, k], scaled scores [n, k]) of the k best new items per test bag."""
        X = test_set.tocsr()
        condition_data = None
        if self.conditions:
            condition_data = self.conditions.transform(test_set.get_attributes(self.conditions.keys()))
        return self.model.predict_topk(X, k=k, condition_data=condition_data)
