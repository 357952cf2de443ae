"""PyTorch-CPU port of the reference step in the reference's own DENSE formulation.
TEST / BASELINE INFRASTRUCTURE - never imported by the shipped path.

Purpose: bench.py's `cpu_baseline` leg.  The reference's Python cannot travel to the GPU box,
so the number the GPU result is reported next to is this port, which issues the same stock
ATen operators in the same order as the reference does per batch:

    toarray() -> FloatTensor                          aae.py:823, 751
    F.normalize(x, 1) -> Linear -> Dropout -> act ... aae.py:129-146 (Encoder), 164-178, 195-213
    F.binary_cross_entropy(x+TINY, t+TINY)            aae.py:693-695
    zero_grad / backward / four torch.optim.Adam      aae.py:697-707, 728-730, 739-742, 798-804
    three .item() calls                               aae.py:711, 732, 743

It is written functionally (parameter dicts + torch.nn.functional), not as nn.Module classes.
tests/test_oracle_golden.py::test_dense_port_* pins it to the golden vectors (dropout off, or
masks injected), so its arithmetic is the reference's.
"""
import numpy as np
import torch
import torch.nn.functional as F

TINY = 1e-12
_ACTS = {"ReLU": F.relu, "SELU": F.selu, "Tanh": torch.tanh, "Sigmoid": torch.sigmoid,
         "ELU": F.elu, "LeakyReLU": F.leaky_relu,
         # r6: the other parameter-free element-wise classes, as getattr(nn, name)() evaluates them (aae.py:110)
         "Softplus": F.softplus, "Hardtanh": F.hardtanh, "ReLU6": F.relu6, "CELU": F.celu, "Softsign": F.softsign,
         "Hardsigmoid": F.hardsigmoid, "LogSigmoid": F.logsigmoid, "Softshrink": F.softshrink, "Hardshrink": F.hardshrink,
         "Identity": lambda x: x, "GELU": F.gelu, "SiLU": F.silu, "Mish": F.mish, "Hardswish": F.hardswish}


def init_params(n_items, n_hidden, n_code, cond_inc=0, seed=0):
    """nn.Linear default init (U(+-1/sqrt(fan_in)) for weight and bias), as fit() gets from
    Encoder/Decoder/Discriminator construction (aae.py:782-793)."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        k = 1.0 / np.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * k
        b = (torch.rand(out_f, generator=g) * 2 - 1) * k
        return w, b
    shapes = {"enc.lin1": (n_hidden, n_items), "enc.lin2": (n_hidden, n_hidden), "enc.lin3": (n_code, n_hidden),
              "dec.lin1": (n_hidden, n_code + cond_inc), "dec.lin2": (n_hidden, n_hidden),
              "dec.lin3": (n_items, n_hidden), "disc.lin1": (n_hidden, n_code),
              "disc.lin2": (n_hidden, n_hidden), "disc.lin3": (1, n_hidden)}
    p = {}
    for name, (o, i) in shapes.items():
        w, b = lin(o, i)
        p[name + ".weight"], p[name + ".bias"] = w.numpy(), b.numpy()
    return p


class DenseTorchAAE:
    def __init__(self, params, gen_lr=1e-3, reg_lr=1e-3, prior="gauss", prior_scale=None, optimizer="adam",
                 normalize_inputs=True, activation="ReLU", dropout=(.2, .2)):
        self.p = {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=True) for k, v in params.items()}
        self.act = _ACTS[activation]
        self.alpha = activation == "SELU"
        self.dropout, self.normalize = dropout, normalize_inputs
        self.final = {"gauss": None, "categorical": "softmax", "bernoulli": "sigmoid"}[prior]
        self.prior_scale = prior_scale
        mk = torch.optim.Adam if optimizer == "adam" else torch.optim.SGD
        grp = lambda net: [self.p[k] for k in self.p if k.startswith(net + ".")]   # noqa: E731
        self.opt_enc, self.opt_dec = mk(grp("enc"), lr=gen_lr), mk(grp("dec"), lr=gen_lr)
        self.opt_gen, self.opt_disc = mk(grp("enc"), lr=reg_lr), mk(grp("disc"), lr=reg_lr)
        self.n_code = self.p["enc.lin3.weight"].shape[0]

    def _drop(self, x, p, train, keep):
        if not train or p == 0.0:
            return x
        if keep is None:
            return F.alpha_dropout(x, p, True) if self.alpha else F.dropout(x, p, True)
        keep = torch.as_tensor(keep, dtype=torch.float32)
        if self.alpha:
            al = 1.7580993408473766
            a = 1.0 / np.sqrt((al * al * p + 1) * (1 - p))
            return x * (keep * a) + ((keep - 1) * (al * a) + al * a * p)
        return x * (keep / (1 - p))

    def _stack(self, net, x, train, keeps):
        P = self.p
        k1, k2 = keeps if keeps is not None else (None, None)
        x = self.act(self._drop(F.linear(x, P[net + ".lin1.weight"], P[net + ".lin1.bias"]), self.dropout[0], train, k1))
        x = self.act(self._drop(F.linear(x, P[net + ".lin2.weight"], P[net + ".lin2.bias"]), self.dropout[1], train, k2))
        return F.linear(x, P[net + ".lin3.weight"], P[net + ".lin3.bias"])

    def _enc(self, X, train, keeps=None):
        if self.normalize:
            X = F.normalize(X, 1)
        z = self._stack("enc", X, train, keeps)
        if self.final == "softmax":
            z = torch.softmax(z, dim=1)
        elif self.final == "sigmoid":
            z = torch.sigmoid(z)
        return z

    def partial_fit(self, X_dense, z_real=None, masks=None, cond=None):
        """X_dense: ndarray [B, N] (what X_shuf[start:end].toarray() yields)."""
        X = torch.FloatTensor(X_dense)
        mk = masks if masks is not None else [None] * 12
        use = masks is not None
        # ae_step
        z = self._enc(X, True, (mk[0], mk[1]) if use else None)
        if cond is not None:
            z = torch.cat([z, torch.as_tensor(cond, dtype=torch.float32)], 1)
        xh = torch.sigmoid(self._stack("dec", z, True, (mk[2], mk[3]) if use else None))
        recon = F.binary_cross_entropy(xh + TINY, X + TINY)
        self.opt_enc.zero_grad(); self.opt_dec.zero_grad()
        recon.backward()
        self.opt_enc.step(); self.opt_dec.step()
        # disc_step
        B = X.shape[0]
        zr = torch.randn(B, self.n_code) if z_real is None else torch.as_tensor(z_real, dtype=torch.float32)
        if self.prior_scale is not None:
            zr = zr * self.prior_scale
        zf = self._enc(X, False)
        dr = torch.sigmoid(self._stack("disc", zr, True, (mk[4], mk[5]) if use else None))
        df = torch.sigmoid(self._stack("disc", zf, True, (mk[6], mk[7]) if use else None))
        dl = -torch.mean(torch.log(dr + TINY) + torch.log(1 - df + TINY))
        self.opt_disc.zero_grad()
        dl.backward()
        self.opt_disc.step()
        # gen_step
        zf = self._enc(X, True, (mk[8], mk[9]) if use else None)
        dg = torch.sigmoid(self._stack("disc", zf, True, (mk[10], mk[11]) if use else None))
        gl = -torch.mean(torch.log(dg + TINY))
        self.opt_gen.zero_grad()
        gl.backward()
        self.opt_gen.step()
        return recon.item(), dl.item(), gl.item()

    def predict(self, X_dense, cond=None):
        with torch.no_grad():
            z = self._enc(torch.FloatTensor(X_dense), False)
            if cond is not None:
                z = torch.cat([z, torch.as_tensor(cond, dtype=torch.float32)], 1)
            return torch.sigmoid(self._stack("dec", z, False, None)).numpy()

    def state_dict(self):
        return {k: v.detach().numpy() for k, v in self.p.items()}
