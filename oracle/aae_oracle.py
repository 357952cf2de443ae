"""CPU oracle for the adversarial-autoencoder training step.  TEST INFRASTRUCTURE.

This file is the checker, not the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The shipped path
(aae-recommender_amd/aaerec/aae.py -> libaaerec_hip.so) never does.

It restates, in plain NumPy float32 with explicit hand-derived backward passes
(no autograd), what one `AdversarialAutoEncoder.partial_fit` of the reference
computes (reference aaerec/aae.py:745-766):

    ae_step   aae.py:676-711   Encoder(train) -> conditions -> Decoder -> BCE -> Adam(enc), Adam(dec)
    disc_step aae.py:713-732   Encoder(eval) ; D(z_real), D(z_fake) -> Adam(disc)
    gen_step  aae.py:734-743   Encoder(train) -> D -> Adam#2(enc)
    Encoder   aae.py:104-146   L1-normalise, 3 Linear, dropout BEFORE activation
    Decoder   aae.py:149-178   3 Linear + sigmoid
    Discriminator aae.py:181-213
    predict   aae.py:840-870   eval-mode enc -> conditions -> dec

The arithmetic delegated by the reference to PyTorch (torch is unpinned in the
reference's setup.py:3-14; the fixtures were produced with torch 2.10.0) is
restated from PyTorch's published semantics:
  F.normalize(x, p=1, dim=1, eps=1e-12)      x / max(sum|x|, eps)
  nn.Dropout(p)                              x * (keep / (1-p))
  nn.AlphaDropout(p)                         x * (keep*a) + ((keep-1)*alpha*a + alpha*a*p)
  F.binary_cross_entropy(mean)               -(t*max(log x,-100) + (1-t)*max(log1p(-x),-100));
                                             grad (x-t)/max((1-x)*x, 1e-12)/numel
  optim.Adam (single-tensor, defaults)       m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
                                             p += -(lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
  optim.SGD (momentum 0)                     p += -lr * g

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which tools/gen_golden.py produced by running the
real reference (imported from /root/reference) in the build container: losses,
every parameter and all four Adam states after each of 3-5 steps, across
dropout / SELU / priors / SGD / conditions / ragged batches.
"""
import numpy as np

f32 = np.float32
TINY = f32(1e-12)          # aae.py:28
SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946
ADROP_ALPHA = 1.7580993408473766     # = SELU_ALPHA * SELU_SCALE, as torch's alpha_dropout


# ---------------------------------------------------------------------------
# activations  (getattr(nn, activation)(), aae.py:110)
# ---------------------------------------------------------------------------
def act_fwd(name, x):
    if name == "ReLU":
        return np.maximum(x, f32(0))
    if name == "SELU":
        neg = f32(SELU_SCALE * SELU_ALPHA) * np.expm1(np.minimum(x, f32(0)))
        return np.where(x > 0, f32(SELU_SCALE) * x, neg).astype(f32)
    if name == "Tanh":
        return np.tanh(x)
    if name == "Sigmoid":
        return sigmoid(x)
    if name == "ELU":
        return np.where(x > 0, x, np.expm1(np.minimum(x, f32(0)))).astype(f32)
    if name == "LeakyReLU":
        return np.where(x > 0, x, f32(0.01) * x).astype(f32)
    # r6: the other parameter-free element-wise torch.nn classes at their default arguments (torch/nn/modules/activation.py)
    if name == "Softplus":          # beta = 1, threshold = 20
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, f32(20))))).astype(f32)
    if name == "Hardtanh":
        return np.clip(x, f32(-1), f32(1)).astype(f32)
    if name == "ReLU6":
        return np.clip(x, f32(0), f32(6)).astype(f32)
    if name == "CELU":              # alpha = 1: max(0, x) + min(0, exp(x) - 1)
        return np.where(x > 0, x, np.expm1(np.minimum(x, f32(0)))).astype(f32)
    if name == "Softsign":
        return (x / (f32(1) + np.abs(x))).astype(f32)
    if name == "Hardsigmoid":       # relu6(x + 3) / 6
        return np.clip(x / f32(6) + f32(0.5), f32(0), f32(1)).astype(f32)
    if name == "LogSigmoid":
        return (np.minimum(x, f32(0)) - np.log1p(np.exp(-np.abs(x)))).astype(f32)
    if name == "Softshrink":        # lambda = 0.5
        return np.where(x > 0.5, x - f32(0.5), np.where(x < -0.5, x + f32(0.5), f32(0))).astype(f32)
    if name == "Hardshrink":        # lambda = 0.5
        return np.where(np.abs(x) > 0.5, x, f32(0)).astype(f32)
    if name == "Identity":
        return x.astype(f32)
    if name == "GELU":              # approximate = 'none': x Phi(x)
        return (x * f32(0.5) * (f32(1) + _erf(x * f32(0.7071067811865476)))).astype(f32)
    if name == "SiLU":
        return (x * sigmoid(x)).astype(f32)
    if name == "Mish":              # x tanh(softplus(x))
        return (x * np.tanh(act_fwd("Softplus", x))).astype(f32)
    if name == "Hardswish":         # x relu6(x + 3) / 6
        return (x * np.clip(x + f32(3), f32(0), f32(6)) / f32(6)).astype(f32)
    raise ValueError("activation not covered by the oracle: " + name)


def _erf(x):
    from scipy.special import erf
    return erf(x.astype(np.float64)).astype(f32)


def act_bwd(name, x, y, g):
    """dL/dx given pre-activation x, output y, upstream g."""
    if name == "ReLU":
        return np.where(x > 0, g, f32(0)).astype(f32)
    if name == "SELU":
        d = np.where(x > 0, f32(SELU_SCALE), y + f32(SELU_SCALE * SELU_ALPHA))
        return (g * d).astype(f32)
    if name == "Tanh":
        return (g * (f32(1) - y * y)).astype(f32)
    if name == "Sigmoid":
        return (g * y * (f32(1) - y)).astype(f32)
    if name == "ELU":
        return (g * np.where(x > 0, f32(1), y + f32(1))).astype(f32)
    if name == "LeakyReLU":
        return np.where(x > 0, g, f32(0.01) * g).astype(f32)
    # r6 (derivatives as ATen's backward formulas state them, in the pre-activation x)
    if name == "Softplus":
        return (g * np.where(x > 20, f32(1), sigmoid(x))).astype(f32)
    if name == "Hardtanh":
        return np.where((x > -1) & (x < 1), g, f32(0)).astype(f32)
    if name == "ReLU6":
        return np.where((x > 0) & (x < 6), g, f32(0)).astype(f32)
    if name == "CELU":
        return (g * np.where(x > 0, f32(1), y + f32(1))).astype(f32)
    if name == "Softsign":
        return (g / np.square(f32(1) + np.abs(x))).astype(f32)
    if name == "Hardsigmoid":
        return np.where((x > -3) & (x < 3), g / f32(6), f32(0)).astype(f32)
    if name == "LogSigmoid":
        return (g * sigmoid(-x)).astype(f32)
    if name == "Softshrink" or name == "Hardshrink":
        return np.where(np.abs(x) > 0.5, g, f32(0)).astype(f32)
    if name == "Identity":
        return g.astype(f32)
    if name == "GELU":
        cdf = f32(0.5) * (f32(1) + _erf(x * f32(0.7071067811865476)))
        pdf = f32(0.3989422804014327) * np.exp(f32(-0.5) * x * x)
        return (g * (cdf + x * pdf)).astype(f32)
    if name == "SiLU":
        sg = sigmoid(x)
        return (g * sg * (f32(1) + x * (f32(1) - sg))).astype(f32)
    if name == "Mish":
        t = np.tanh(act_fwd("Softplus", x))
        return (g * (t + x * sigmoid(x) * (f32(1) - t * t))).astype(f32)
    if name == "Hardswish":
        return np.where(x < -3, f32(0), np.where(x <= 3, g * (x / f32(3) + f32(0.5)), g)).astype(f32)
    raise ValueError(name)


def sigmoid(x):
    x = x.astype(f32)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = f32(1) / (f32(1) + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (f32(1) + e)
    return out


def softmax_rows(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=1, keepdims=True)).astype(f32)


# ---------------------------------------------------------------------------
# dropout (applied to the PRE-activation, aae.py:135-137)
# ---------------------------------------------------------------------------
class Drop:
    """Holds one keep-mask; fwd/bwd for Dropout or AlphaDropout."""

    def __init__(self, p, keep, alpha_mode):
        self.p, self.alpha_mode = float(p), alpha_mode
        self.keep = None if (keep is None or self.p == 0.0) else keep.astype(f32)
        if self.keep is not None:
            if alpha_mode:
                a = 1.0 / np.sqrt((ADROP_ALPHA ** 2 * self.p + 1) * (1 - self.p))
                self.mul = self.keep * f32(a)
                self.add = ((self.keep - f32(1)) * f32(ADROP_ALPHA * a) + f32(ADROP_ALPHA * a * self.p)).astype(f32)
            else:
                self.mul = self.keep / f32(1 - self.p)
                self.add = None

    def fwd(self, x):
        if self.keep is None:
            return x
        y = x * self.mul
        return y if self.add is None else (y + self.add).astype(f32)

    def bwd(self, g):
        return g if self.keep is None else (g * self.mul).astype(f32)


# ---------------------------------------------------------------------------
# optimisers
# ---------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8), one state per tensor name."""

    def __init__(self, lr):
        self.lr, self.m, self.v, self.t = float(lr), {}, {}, {}

    def step(self, params, grads):
        b1, b2, eps = 0.9, 0.999, 1e-8
        for k, g in grads.items():
            p = params[k]
            if k not in self.m:
                self.m[k], self.v[k], self.t[k] = np.zeros_like(p), np.zeros_like(p), 0
            self.t[k] += 1
            t = self.t[k]
            m, v = self.m[k], self.v[k]
            m += f32(1 - b1) * (g - m)                     # lerp_
            v *= f32(b2)
            v += (f32(1 - b2) * g) * g                     # addcmul_
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = np.sqrt(v) / f32(bc2 ** 0.5) + f32(eps)
            p += (f32(-(self.lr / bc1)) * m) / denom       # addcdiv_


class SGD:
    def __init__(self, lr):
        self.lr, self.m, self.v, self.t = float(lr), {}, {}, {}

    def step(self, params, grads):
        for k, g in grads.items():
            params[k] += f32(-self.lr) * g


# ---------------------------------------------------------------------------
# condition stand-ins for the fixtures (condition.py:90-99, 300-342, 397-508)
# ---------------------------------------------------------------------------
class ConcatConst:
    """ConcatenationBasedConditioning with a constant encoded input [B, inc]."""

    def __init__(self, inc):
        self.inc = inc

    def fwd(self, z, inp, train=True):
        self._c = z.shape[1]
        return np.concatenate([z, np.asarray(inp, dtype=f32)], axis=1)

    def bwd(self, dz):
        return dz[:, :self._c]

    def step(self):
        pass


class BiasConst:
    """ConditionalBiasing with a constant encoded input (condition.py:319-329)."""
    inc = 0

    def fwd(self, z, inp, train=True):
        return (z + np.asarray(inp, dtype=f32)).astype(f32)

    def bwd(self, dz):
        return dz

    def step(self):
        pass


class SparseAdam:
    """torch.optim.SparseAdam defaults on one row-sparse tensor: only the rows present in the (coalesced) gradient
    move, and eps is added to sqrt(exp_avg_sq) before the bias corrections are applied as one step size
    (torch/optim/_functional.py sparse_adam)."""

    def __init__(self, lr):
        self.lr, self.m, self.v, self.t = float(lr), None, None, 0

    def step(self, p, rows, g):
        """p [vocab, dim] updated in place; rows: unique indices; g [len(rows), dim] their summed gradients."""
        b1, b2, eps = 0.9, 0.999, 1e-8
        if self.m is None:
            self.m, self.v = np.zeros_like(p), np.zeros_like(p)
        self.t += 1
        if len(rows) == 0:
            return
        m_old, v_old = self.m[rows], self.v[rows]
        mu = ((g - m_old) * f32(1 - b1)).astype(f32)
        self.m[rows] = m_old + mu
        vu = ((g * g - v_old) * f32(1 - b2)).astype(f32)
        self.v[rows] = v_old + vu
        numer = mu + m_old
        denom = np.sqrt(vu + v_old) + f32(eps)
        step_size = self.lr * (1 - b2 ** self.t) ** 0.5 / (1 - b1 ** self.t)
        p[rows] += f32(-step_size) * (numer / denom)


class CategoricalEmbedding:
    """CategoricalCondition (condition.py:397-508): a trainable embedding of a categorical attribute concatenated
    to the code.  Index 0 = padding / out of vocabulary: its row stays zero and gets no gradient.
      reduce  None: one index per document;  'sum' / 'mean': over the batch-padded list (mean divides by the padded
              width, padding included - hid.mean(1) in the reference)
      sparse  True: nn.Embedding(sparse=True) + SparseAdam (the reference's default);  False: dense Adam."""

    def __init__(self, weight, lr, reduce="sum", sparse=False):
        self.params = {"w": weight.astype(f32).copy()}
        self.reduce, self.sparse = reduce, sparse
        self.opt = SparseAdam(lr) if sparse else Adam(lr)
        self.inc = weight.shape[1]

    def encode(self, idx):
        idx = np.asarray(idx)
        if idx.ndim == 1:
            idx = idx[:, None]
        self._idx = idx
        e = self.params["w"][idx].sum(axis=1).astype(f32)
        if self.reduce == "mean":
            e = (e / f32(idx.shape[1])).astype(f32)
        return e

    def fwd(self, z, idx, train=True):
        self._c = z.shape[1]
        return np.concatenate([z, self.encode(idx)], axis=1)

    def bwd(self, dz):
        de = dz[:, self._c:self._c + self.inc]
        if self.reduce == "mean":
            de = (de / f32(self._idx.shape[1])).astype(f32)
        g = np.zeros_like(self.params["w"])
        touched = np.zeros(len(g), dtype=bool)
        for b in range(self._idx.shape[0]):
            for j in self._idx[b]:
                if j != 0:
                    g[j] += de[b]
                    touched[j] = True
        self._g, self._rows = g, np.nonzero(touched)[0]
        return dz[:, :self._c]

    def step(self):
        if self.sparse:
            self.opt.step(self.params["w"], self._rows, self._g[self._rows])
        else:
            self.opt.step(self.params, {"w": self._g})


def CategoricalSum(weight, lr):
    """CategoricalCondition(reduce='sum', sparse=False)."""
    return CategoricalEmbedding(weight, lr, reduce="sum", sparse=False)


# ---------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------
def _lin(x, W, b):
    return (x @ W.T + b).astype(f32)


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what v_cvt_pk_bf16_f32 does to a matrix-core
    operand in the build's bf16 mode (BASELINE config C2; the reference has no bf16 path, SURVEY section 7 defines it
    as bf16 MFMA inputs, fp32 accumulate, fp32 master parameters and optimiser)."""
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(f32)


class OracleAAE:
    """State = torch-layout parameters ([out,in]) + four optimiser states."""
    bf16 = False        # (subclasses with constructors of their own run in the reference's fp32)

    def __init__(self, params, gen_lr=1e-3, reg_lr=1e-3, prior="gauss", prior_scale=None,
                 optimizer="adam", normalize_inputs=True, activation="ReLU",
                 dropout=(0.2, 0.2), conditions=None, bf16=False):
        # params: dict "enc.lin1.weight" -> ndarray (copied)
        self.p = {k: np.array(v, dtype=f32) for k, v in params.items()}
        # bf16 mode of the build (not of the reference): every dense product that runs on the matrix cores takes its
        # two operands rounded to bf16 and accumulates in fp32 - the Linear layers' forward (bias included: it is a
        # column of the augmented weight matrix) and dX products, and all three products of the decoder's output
        # layer; the sparse first encoder layer, the hidden layers' weight gradients, losses, optimisers stay fp32
        self.bf16 = bool(bf16)
        self.N = self.p["enc.lin1.weight"].shape[1]
        self.c = self.p["enc.lin3.weight"].shape[0]
        self.prior, self.prior_scale = prior, prior_scale
        self.final = {"gauss": None, "categorical": "softmax", "bernoulli": "sigmoid"}[prior]
        self.normalize, self.act, self.dropout = normalize_inputs, activation, tuple(dropout)
        mk = Adam if optimizer == "adam" else SGD
        self.opt_enc, self.opt_dec = mk(gen_lr), mk(gen_lr)
        self.opt_gen, self.opt_disc = mk(reg_lr), mk(reg_lr)
        self.conditions = conditions or []
        self.alpha_mode = activation == "SELU"

    # -- pieces ------------------------------------------------------------
    def _row_scale(self, indptr, values):
        B = len(indptr) - 1
        s = np.ones(B, dtype=f32)
        if self.normalize:
            for b in range(B):
                l1 = np.abs(values[indptr[b]:indptr[b + 1]]).sum(dtype=f32)
                s[b] = f32(1) / max(l1, TINY)
        return s

    def _first_layer(self, indptr, indices, values, s):
        """a1[b] = b1 + sum_i (v_i * s_b) * W1[:, i]   (K2+K3 of SURVEY 2.1)."""
        W, b1 = self.p["enc.lin1.weight"], self.p["enc.lin1.bias"]
        B = len(indptr) - 1
        a1 = np.empty((B, W.shape[0]), dtype=f32)
        for b in range(B):
            lo, hi = indptr[b], indptr[b + 1]
            xn = (values[lo:hi] * s[b]).astype(f32) if self.normalize else values[lo:hi]
            a1[b] = W[:, indices[lo:hi]] @ xn + b1
        return a1

    def _R(self, x):
        return bf16_round(x) if self.bf16 else x

    def _linear(self, x, W, b):
        return _lin(self._R(x), self._R(W), self._R(b))

    def _mlp_fwd(self, net, x0, masks, first_pre=None):
        """3-layer stack; returns (output_pre_final, cache). masks: (keep1, keep2) or None."""
        P = self.p
        k1, k2 = masks if masks is not None else (None, None)
        d1, d2 = Drop(self.dropout[0], k1, self.alpha_mode), Drop(self.dropout[1], k2, self.alpha_mode)
        a1 = first_pre if first_pre is not None else self._linear(x0, P[net + ".lin1.weight"], P[net + ".lin1.bias"])
        u1 = d1.fwd(a1)
        h1 = act_fwd(self.act, u1)
        a2 = self._linear(h1, P[net + ".lin2.weight"], P[net + ".lin2.bias"])
        u2 = d2.fwd(a2)
        h2 = act_fwd(self.act, u2)
        if net == "disc":     # the discriminator's 1-unit output layer is a dot product on the vector units: fp32 in every mode
            a3 = _lin(h2, P[net + ".lin3.weight"], P[net + ".lin3.bias"])
        else:
            a3 = self._linear(h2, P[net + ".lin3.weight"], P[net + ".lin3.bias"])
        return a3, dict(x0=x0, d1=d1, d2=d2, u1=u1, h1=h1, u2=u2, h2=h2)

    def _mlp_bwd(self, net, g3, cache, need_dx=True):
        """Given dL/da3 -> grads of lin3/lin2/lin1 (lin1 weight grad returned as da1), dL/dx0."""
        P, R = self.p, self._R
        if net == "dec" and self.bf16:       # the decoder's output layer: its weight gradient is a matrix-core product too
            G = {net + ".lin3.weight": (R(g3).T @ R(cache["h2"])).astype(f32), net + ".lin3.bias": R(g3).sum(0).astype(f32)}
        else:
            G = {net + ".lin3.weight": (g3.T @ cache["h2"]).astype(f32), net + ".lin3.bias": g3.sum(0).astype(f32)}
        gh2 = (g3 @ P[net + ".lin3.weight"]).astype(f32) if net == "disc" else (R(g3) @ R(P[net + ".lin3.weight"])).astype(f32)
        ga2 = cache["d2"].bwd(act_bwd(self.act, cache["u2"], cache["h2"], gh2))
        G[net + ".lin2.weight"] = (ga2.T @ cache["h1"]).astype(f32)
        G[net + ".lin2.bias"] = ga2.sum(0).astype(f32)
        gh1 = (R(ga2) @ R(P[net + ".lin2.weight"])).astype(f32)
        ga1 = cache["d1"].bwd(act_bwd(self.act, cache["u1"], cache["h1"], gh1))
        G[net + ".lin1.bias"] = ga1.sum(0).astype(f32)
        gx = (R(ga1) @ R(P[net + ".lin1.weight"])).astype(f32) if need_dx else None
        return G, ga1, gx

    def _enc_final_fwd(self, a3):
        if self.final is None:
            return a3
        return softmax_rows(a3) if self.final == "softmax" else sigmoid(a3)

    def _enc_final_bwd(self, z, gz):
        if self.final is None:
            return gz
        if self.final == "sigmoid":
            return (gz * z * (f32(1) - z)).astype(f32)
        dot = (gz * z).sum(axis=1, keepdims=True)
        return (z * (gz - dot)).astype(f32)

    def _enc_w1_grad(self, indptr, indices, values, s, ga1):
        """dW1[:, i] += (v_i * s_b) * ga1[b]  (K9): dense [h, N] result."""
        g = np.zeros_like(self.p["enc.lin1.weight"])
        for b in range(len(indptr) - 1):
            lo, hi = indptr[b], indptr[b + 1]
            xn = (values[lo:hi] * s[b]).astype(f32) if self.normalize else values[lo:hi]
            np.add.at(g.T, indices[lo:hi], np.outer(xn, ga1[b]).astype(f32))
        return g

    def encode(self, indptr, indices, values, masks=None, input_noise=None):
        """input_noise [B, N]: the DenoisingAutoEncoder's corrupt='gauss' (dae.py:40-45, 191): the encoder sees the DENSE
        batch + noise (already scaled by noise_factor); F.normalize(x, 1) then runs over all N columns (aae.py:132-133)."""
        if input_noise is not None:
            B = len(indptr) - 1
            X = np.asarray(input_noise, dtype=f32).copy()
            for b in range(B):
                X[b, indices[indptr[b]:indptr[b + 1]]] += values[indptr[b]:indptr[b + 1]]
            if self.normalize:
                l1 = np.abs(X).sum(axis=1, dtype=f32)
                X = (X / np.maximum(l1, TINY)[:, None]).astype(f32)
            a1 = (X @ self.p["enc.lin1.weight"].T + self.p["enc.lin1.bias"]).astype(f32)
            a3, cache = self._mlp_fwd("enc", None, masks, first_pre=a1)
            cache["a1"], cache["s"], cache["xdense"] = a1, None, X
            return self._enc_final_fwd(a3), cache
        s = self._row_scale(indptr, values)
        a1 = self._first_layer(indptr, indices, values, s)
        a3, cache = self._mlp_fwd("enc", None, masks, first_pre=a1)
        cache["a1"], cache["s"] = a1, s
        return self._enc_final_fwd(a3), cache

    def _disc(self, z, masks):
        a3, cache = self._mlp_fwd("disc", z, masks)
        return sigmoid(a3), cache

    # -- the three sub-steps ----------------------------------------------
    def _bce(self, logits, indptr, indices, values):
        """F.binary_cross_entropy(sigmoid(logits) + TINY, X + TINY), mean over B*N, and its gradient w.r.t. the
        logits exactly as autograd forms it (aae.py:693-695; clamped logs, 1e-12 floor in the backward)."""
        B, N = logits.shape
        xhat = sigmoid(logits)
        T = np.zeros((B, N), dtype=f32)
        for b in range(B):
            T[b, indices[indptr[b]:indptr[b + 1]]] = values[indptr[b]:indptr[b + 1]]
        x = xhat + TINY
        t = T + TINY
        with np.errstate(divide="ignore"):
            lx = np.maximum(np.log(x), f32(-100))
            l1x = np.maximum(np.log1p(-x), f32(-100))
        loss = float((-(t * lx + (f32(1) - t) * l1x)).mean(dtype=np.float64))
        gx = (x - t) / np.maximum((f32(1) - x) * x, f32(1e-12)) / f32(B * N)
        glog = (gx * xhat * (f32(1) - xhat)).astype(f32)
        return loss, glog, xhat

    def ae_step(self, indptr, indices, values, masks, cond_inputs=None, input_noise=None):
        """masks = [enc.drop1, enc.drop2, dec.drop1, dec.drop2] keep-masks or None."""
        mk = masks if masks is not None else [None] * 4
        B, N = len(indptr) - 1, self.N
        z, ec = self.encode(indptr, indices, values, (mk[0], mk[1]), input_noise=input_noise)
        zc = z
        for cond, inp in zip(self.conditions, cond_inputs or []):
            zc = cond.fwd(zc, inp)
        logits, dc = self._mlp_fwd("dec", zc, (mk[2], mk[3]))
        loss, glog, xhat = self._bce(logits, indptr, indices, values)
        Gd, gda1, gzc = self._mlp_bwd("dec", glog, dc)
        Gd["dec.lin1.weight"] = (gda1.T @ zc).astype(f32)
        gz = gzc
        for cond in reversed(self.conditions):
            gz = cond.bwd(gz)
        ga3 = self._enc_final_bwd(z, gz)
        Ge, ga1, _ = self._mlp_bwd("enc", ga3, ec, need_dx=False)
        if "xdense" in ec:
            Ge["enc.lin1.weight"] = (ga1.T @ ec["xdense"]).astype(f32)
        else:
            Ge["enc.lin1.weight"] = self._enc_w1_grad(indptr, indices, values, ec["s"], ga1)
        self.opt_enc.step(self.p, Ge)
        self.opt_dec.step(self.p, Gd)
        for cond in self.conditions:
            cond.step()
        self.last = dict(enc_a1=ec["a1"], z=z, xhat=xhat)
        return loss

    def disc_step(self, indptr, indices, values, z_real, masks):
        """enc in eval mode; masks = [disc.drop1, disc.drop2 (real)], [.. (fake)]."""
        mk = masks if masks is not None else [None] * 4
        B = len(indptr) - 1
        zr = np.asarray(z_real, dtype=f32)
        if self.prior_scale is not None:
            zr = (zr * f32(self.prior_scale)).astype(f32)
        zf, _ = self.encode(indptr, indices, values, None)
        dr, cr = self._disc(zr, (mk[0], mk[1]))
        df, cf = self._disc(zf, (mk[2], mk[3]))
        loss = float(-(np.log(dr + TINY) + np.log(f32(1) - df + TINY)).mean(dtype=np.float64))
        g_dr = (f32(-1.0 / B) / (dr + TINY)).astype(f32)
        g_df = (f32(1.0 / B) / (f32(1) - df + TINY)).astype(f32)
        G = {}
        for d, g, cache in ((dr, g_dr, cr), (df, g_df, cf)):
            ga3 = (g * d * (f32(1) - d)).astype(f32)
            Gi, ga1, _ = self._mlp_bwd("disc", ga3, cache, need_dx=False)
            Gi["disc.lin1.weight"] = (ga1.T @ cache["x0"]).astype(f32)
            for k, v in Gi.items():
                G[k] = v if k not in G else (G[k] + v).astype(f32)
        self.opt_disc.step(self.p, G)
        self.last["z_disc"] = zf
        return loss

    def gen_step(self, indptr, indices, values, masks):
        """masks = [enc.drop1, enc.drop2, disc.drop1, disc.drop2]."""
        mk = masks if masks is not None else [None] * 4
        B = len(indptr) - 1
        z, ec = self.encode(indptr, indices, values, (mk[0], mk[1]))
        d, cd = self._disc(z, (mk[2], mk[3]))
        loss = float(-np.log(d + TINY).mean(dtype=np.float64))
        g_d = (f32(-1.0 / B) / (d + TINY)).astype(f32)
        ga3d = (g_d * d * (f32(1) - d)).astype(f32)
        _, _, gz = self._mlp_bwd("disc", ga3d, cd, need_dx=True)
        ga3 = self._enc_final_bwd(z, gz)
        Ge, ga1, _ = self._mlp_bwd("enc", ga3, ec, need_dx=False)
        Ge["enc.lin1.weight"] = self._enc_w1_grad(indptr, indices, values, ec["s"], ga1)
        self.opt_gen.step(self.p, Ge)
        self.last["z_gen"] = z
        return loss

    def partial_fit(self, indptr, indices, values, z_real, masks=None, cond_inputs=None):
        """masks: the 12 keep-masks in the reference's call order (SURVEY 7 'RNG parity'):
        enc.d1 enc.d2 dec.d1 dec.d2 | disc.d1 disc.d2 (real) disc.d1 disc.d2 (fake) |
        enc.d1 enc.d2 disc.d1 disc.d2.  None = dropout off."""
        if values.size and (values.max() > 1 or values.min() < 0):
            raise RuntimeError("all elements of target should be between 0 and 1")
        mk = masks if masks is not None else [None] * 12
        r = self.ae_step(indptr, indices, values, mk[0:4], cond_inputs)
        d = self.disc_step(indptr, indices, values, z_real, mk[4:8])
        g = self.gen_step(indptr, indices, values, mk[8:12])
        return r, d, g

    def predict(self, indptr, indices, values, cond_inputs=None):
        z, _ = self.encode(indptr, indices, values, None)
        for cond, inp in zip(self.conditions, cond_inputs or []):
            z = cond.fwd(z, inp, train=False)
        logits, _ = self._mlp_fwd("dec", z, None)
        return sigmoid(logits)



class OracleDecoder(OracleAAE):
    """DecodingRecommender (aae.py:461-584): the encoded conditions (first one as is, the others imposed on it,
    aae.py:494-502) -> the 3-layer Decoder -> BCE against the item rows; one optimiser for the decoder, the
    conditions step their own.  Condition stand-ins start from an empty code block."""

    def __init__(self, params, lr=1e-3, optimizer="adam", activation="ReLU", dropout=(0.2, 0.2), conditions=None):
        self.p = {k: np.array(v, dtype=f32) for k, v in params.items() if k.startswith("dec.")}
        self.N = self.p["dec.lin3.weight"].shape[0]
        self.act, self.dropout = activation, tuple(dropout)
        self.alpha_mode = activation == "SELU"
        self.opt_dec = (Adam if optimizer == "adam" else SGD)(lr)
        self.conditions = conditions or []

    def inputs(self, cond_inputs, train=True):
        B = len(cond_inputs[0])
        z = np.zeros((B, 0), dtype=f32)
        for cond, inp in zip(self.conditions, cond_inputs):
            z = cond.fwd(z, inp, train=train)
        return z

    def partial_fit(self, cond_inputs, indptr, indices, values, masks=None):
        """masks = (dec.drop1, dec.drop2) keep-masks or None (aae.py:489-517)."""
        zin = self.inputs(cond_inputs)
        logits, dc = self._mlp_fwd("dec", zin, masks)
        loss, glog, _ = self._bce(logits, indptr, indices, values)
        Gd, gda1, gz = self._mlp_bwd("dec", glog, dc)
        Gd["dec.lin1.weight"] = (gda1.T @ zin).astype(f32)
        self.last_dzin = gz
        for cond in reversed(self.conditions):
            gz = cond.bwd(gz)
        self.opt_dec.step(self.p, Gd)
        for cond in self.conditions:
            cond.step()
        return loss

    def predict(self, cond_inputs):
        logits, _ = self._mlp_fwd("dec", self.inputs(cond_inputs, train=False), None)
        return sigmoid(logits)


class OracleVAE(OracleAAE):
    """VAE (vae.py:47-266): x -> L1 normalise -> fc1 -> act -> (fc21 = mu, fc22 = logvar) -> z = mu + eps*exp(logvar/2)
    -> [conditions] -> fc3 -> act -> fc4 -> sigmoid; loss = nn.BCELoss() (mean: the `size_average = False` the
    reference sets afterwards is not read by torch, vae.py:133-136) + KL sum (vae.py:141-145); ONE optimiser over
    all five Linears.  Parameters are keyed fc1 / fc21 / fc22 / fc3 / fc4 (.weight [out,in], .bias)."""
    NAMES = ("fc1", "fc21", "fc22", "fc3", "fc4")

    def __init__(self, params, lr=1e-3, optimizer="adam", normalize_inputs=True, activation="ReLU", conditions=None):
        self.p = {k: np.array(v, dtype=f32) for k, v in params.items()}
        self.p["enc.lin1.weight"], self.p["enc.lin1.bias"] = self.p["fc1.weight"], self.p["fc1.bias"]   # aliases
        self.N = self.p["fc1.weight"].shape[1]
        self.normalize, self.act = normalize_inputs, activation
        self.opt = (Adam if optimizer == "adam" else SGD)(lr)
        self.conditions = conditions or []

    def forward(self, indptr, indices, values, eps, cond_inputs=None, train=True):
        P = self.p
        s = self._row_scale(indptr, values)
        a1 = self._first_layer(indptr, indices, values, s)
        h1 = act_fwd(self.act, a1)
        mu, lv = _lin(h1, P["fc21.weight"], P["fc21.bias"]), _lin(h1, P["fc22.weight"], P["fc22.bias"])
        std = np.exp(f32(0.5) * lv).astype(f32)
        z = (np.asarray(eps, dtype=f32) * std + mu).astype(f32)
        zc = z
        for cond, inp in zip(self.conditions, cond_inputs or []):
            zc = cond.fwd(zc, inp, train=train)
        a3 = _lin(zc, P["fc3.weight"], P["fc3.bias"])
        h3 = act_fwd(self.act, a3)
        logits = _lin(h3, P["fc4.weight"], P["fc4.bias"])
        return sigmoid(logits), dict(s=s, a1=a1, h1=h1, mu=mu, lv=lv, std=std, eps=np.asarray(eps, dtype=f32), zc=zc,
                                     a3=a3, h3=h3)

    def partial_fit(self, indptr, indices, values, eps, cond_inputs=None):
        """Returns (BCE mean + KL sum) / B, the number the reference logs (vae.py:185)."""
        P = self.p
        B, N = len(indptr) - 1, self.N
        xhat, c = self.forward(indptr, indices, values, eps, cond_inputs)
        T = np.zeros((B, N), dtype=f32)
        for b in range(B):
            T[b, indices[indptr[b]:indptr[b + 1]]] = values[indptr[b]:indptr[b + 1]]
        with np.errstate(divide="ignore"):
            lx = np.maximum(np.log(xhat), f32(-100))
            l1x = np.maximum(np.log1p(-xhat), f32(-100))
        bce = float((-(T * lx + (f32(1) - T) * l1x)).mean(dtype=np.float64))
        mu, lv = c["mu"], c["lv"]
        kld = float(-0.5 * (1.0 + lv.astype(np.float64) - mu.astype(np.float64) ** 2 - np.exp(lv.astype(np.float64))).sum())
        gx = (xhat - T) / np.maximum((f32(1) - xhat) * xhat, f32(1e-12)) / f32(B * N)
        glog = (gx * xhat * (f32(1) - xhat)).astype(f32)
        G = {"fc4.weight": (glog.T @ c["h3"]).astype(f32), "fc4.bias": glog.sum(0).astype(f32)}
        ga3 = act_bwd(self.act, c["a3"], c["h3"], (glog @ P["fc4.weight"]).astype(f32))
        G["fc3.weight"], G["fc3.bias"] = (ga3.T @ c["zc"]).astype(f32), ga3.sum(0).astype(f32)
        gz = (ga3 @ P["fc3.weight"]).astype(f32)
        for cond in reversed(self.conditions):
            gz = cond.bwd(gz)
        gmu = (gz + mu).astype(f32)
        glv = (gz * c["eps"] * f32(0.5) * c["std"] + f32(0.5) * (np.exp(lv) - f32(1))).astype(f32)
        G["fc21.weight"], G["fc21.bias"] = (gmu.T @ c["h1"]).astype(f32), gmu.sum(0).astype(f32)
        G["fc22.weight"], G["fc22.bias"] = (glv.T @ c["h1"]).astype(f32), glv.sum(0).astype(f32)
        ga1 = act_bwd(self.act, c["a1"], c["h1"], (gmu @ P["fc21.weight"] + glv @ P["fc22.weight"]).astype(f32))
        G["fc1.weight"] = self._enc_w1_grad(indptr, indices, values, c["s"], ga1)
        G["fc1.bias"] = ga1.sum(0).astype(f32)
        self.opt.step(self.p, G)
        for cond in self.conditions:
            cond.step()
        self.last = dict(bce=bce, kld=kld)
        return (bce + kld) / B

    def predict(self, indptr, indices, values, eps, cond_inputs=None):
        return self.forward(indptr, indices, values, eps, cond_inputs, train=False)[0]


# ---------------------------------------------------------------------------------------------
# the same step cut into the phases of the C ABI with gradients EXPORTED instead of applied
# (aae_ae_encode / aae_ae_decode_backward / aae_ae_encoder_backward / aae_disc_step /
# aae_gen_step / aae_apply_updates / aae_set_grad_scale, include/aaerec_hip.h) - the stand-in
# that lets the world_size-2 gloo tests drive aaerec.parallel.DataParallelAAE on CPU.
# ---------------------------------------------------------------------------------------------
class OraclePhases(OracleAAE):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.grad_scale = 1.0
        self.G = {}          # optimiser id -> {param name: gradient}
        self.losses = [0.0, 0.0, 0.0]

    def set_grad_scale(self, s):
        self.grad_scale = float(s)

    def ae_encode(self, indptr, indices, values, masks=None, z_real=None):
        self._b = (indptr, indices, values)
        self._mk = masks if masks is not None else [None] * 12
        self._zr = z_real
        z, self._ec = self.encode(indptr, indices, values, (self._mk[0], self._mk[1]))
        self._z = z
        return z

    def ae_decode_backward(self, zc):
        indptr, indices, values = self._b
        B, N = len(indptr) - 1, self.N
        zc = np.asarray(zc, dtype=f32)
        logits, dc = self._mlp_fwd("dec", zc, (self._mk[2], self._mk[3]))
        xhat = sigmoid(logits)
        T = np.zeros((B, N), dtype=f32)
        for b in range(B):
            T[b, indices[indptr[b]:indptr[b + 1]]] = values[indptr[b]:indptr[b + 1]]
        x, t = xhat + TINY, T + TINY
        with np.errstate(divide="ignore"):
            lx = np.maximum(np.log(x), f32(-100))
            l1x = np.maximum(np.log1p(-x), f32(-100))
        self.losses[0] = float((-(t * lx + (f32(1) - t) * l1x)).mean(dtype=np.float64))
        gx = (x - t) / np.maximum((f32(1) - x) * x, f32(1e-12)) * f32(self.grad_scale / (B * N))
        glog = (gx * xhat * (f32(1) - xhat)).astype(f32)
        Gd, gda1, gzc = self._mlp_bwd("dec", glog, dc)
        Gd["dec.lin1.weight"] = (gda1.T @ zc).astype(f32)
        self.G[1] = Gd
        return gzc

    def ae_encoder_backward(self, dz):
        indptr, indices, values = self._b
        ga3 = self._enc_final_bwd(self._z, np.asarray(dz, dtype=f32))
        Ge, ga1, _ = self._mlp_bwd("enc", ga3, self._ec, need_dx=False)
        Ge["enc.lin1.weight"] = self._enc_w1_grad(indptr, indices, values, self._ec["s"], ga1)
        self.G[0] = Ge

    def disc_step(self):
        indptr, indices, values = self._b
        mk = self._mk[4:8]
        B = len(indptr) - 1
        zr = np.asarray(self._zr, dtype=f32)
        if self.prior_scale is not None:
            zr = (zr * f32(self.prior_scale)).astype(f32)
        zf, _ = self.encode(indptr, indices, values, None)
        dr, cr = self._disc(zr, (mk[0], mk[1]))
        df, cf = self._disc(zf, (mk[2], mk[3]))
        self.losses[1] = float(-(np.log(dr + TINY) + np.log(f32(1) - df + TINY)).mean(dtype=np.float64))
        gs = f32(self.grad_scale / B)
        G = {}
        for d, g, cache in ((dr, -gs / (dr + TINY), cr), (df, gs / (f32(1) - df + TINY), cf)):
            ga3 = (g * d * (f32(1) - d)).astype(f32)
            Gi, ga1, _ = self._mlp_bwd("disc", ga3, cache, need_dx=False)
            Gi["disc.lin1.weight"] = (ga1.T @ cache["x0"]).astype(f32)
            for k, v in Gi.items():
                G[k] = v if k not in G else (G[k] + v).astype(f32)
        self.G[3] = G

    def gen_step(self):
        indptr, indices, values = self._b
        mk = self._mk[8:12]
        B = len(indptr) - 1
        z, ec = self.encode(indptr, indices, values, (mk[0], mk[1]))
        d, cd = self._disc(z, (mk[2], mk[3]))
        self.losses[2] = float(-np.log(d + TINY).mean(dtype=np.float64))
        ga3d = ((f32(-self.grad_scale / B) / (d + TINY)) * d * (f32(1) - d)).astype(f32)
        _, _, gz = self._mlp_bwd("disc", ga3d, cd, need_dx=True)
        Ge, ga1, _ = self._mlp_bwd("enc", self._enc_final_bwd(z, gz), ec, need_dx=False)
        Ge["enc.lin1.weight"] = self._enc_w1_grad(indptr, indices, values, ec["s"], ga1)
        self.G[2] = Ge

    def apply_updates(self, which):
        opt = {0: self.opt_enc, 1: self.opt_dec, 2: self.opt_gen, 3: self.opt_disc}[which]
        opt.step(self.p, self.G[which])
