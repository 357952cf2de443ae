"""Data-parallel exchange (aaerec/parallel.py) on CPU: world_size 2, gloo, the kernels replaced by
the oracle's phase-split stand-in (oracle.aae_oracle.OraclePhases).  Two ranks with half a batch
each must end up with the parameters a single process gets from the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import free_port, spawn_ranks, Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleReplica:
    """Adapts OraclePhases to the interface DataParallelAAE drives (aaerec._hip.HipAAE's)."""

    def __init__(self, params, **kw):
        from oracle.aae_oracle import OraclePhases
        self.o = OraclePhases(params, **kw)

    def set_grad_scale(self, s):
        self.o.set_grad_scale(s)

    def ae_encode(self, csr, row_start, n_rows, rows=None, masks=None, z_real=None):
        ip, idx, val = csr
        lo, hi = ip[row_start], ip[row_start + n_rows]
        return torch.from_numpy(self.o.ae_encode(ip[row_start:row_start + n_rows + 1] - lo, idx[lo:hi], val[lo:hi],
                                                 masks, z_real))

    def ae_decode_backward(self, zc):
        return torch.from_numpy(self.o.ae_decode_backward(zc.detach().numpy()))

    def ae_encoder_backward(self, dz):
        self.o.ae_encoder_backward(dz.numpy())

    def disc_step(self):
        self.o.disc_step()

    def gen_step(self):
        self.o.gen_step()

    def apply_updates(self, which):
        self.o.apply_updates(which)

    def grad_buckets(self, which):
        # torch.from_numpy aliases the oracle's gradient arrays: all_reduce updates them in place
        return [torch.from_numpy(g) for g in self.o.G[which].values()]


def _worker(rank, world, port, name, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import DataParallelAAE
    fx = Fixture(name)
    model = OracleReplica(fx.init_params(), **fx.model_kwargs())
    dp = DataParallelAAE(model, dist)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        lo, hi = dp.shard(0, B)
        masks = fx.masks(s)
        if masks is not None:
            masks = [m[lo:hi] for m in masks]
        dp.step((ip, idx, val), lo, hi - lo, global_rows=B, masks=masks, z_real=fx.z[f"step{s}.z_real"][lo:hi])
    if rank == 0:
        ret.update({k: v.copy() for k, v in model.o.p.items()})
    # replicas stay identical
    flat = torch.from_numpy(np.concatenate([v.ravel() for v in model.o.p.values()]))
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(other[0], o) for o in other)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_masks", "step_masks_uneven", "step_ragged"])
def test_two_ranks_equal_single_process(name):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker, 2, lambda port: (2, port, name, ret))
        got = dict(ret)
    fx = Fixture(name)
    # the single-process result is the golden fixture itself (the reference's own parameters)
    want = fx.expected_params(fx.steps - 1)
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)


def test_shard_bounds():
    from aaerec.parallel import DataParallelAAE

    class FakeDist:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

        def get_world_size(self, group=None):
            return self.world

        def get_rank(self, group=None):
            return self.rank
    spans = [DataParallelAAE(None, FakeDist(r, 4)).shard(10, 23) for r in range(4)]
    assert spans == [(10, 14), (14, 17), (17, 20), (20, 23)]
    assert DataParallelAAE(None, FakeDist(0, 4)).shard(0, 3) == (None, None)


class SparseW1Replica(OracleReplica):
    """Same stand-in, but the first encoder layer's gradient travels as packed rows
    (w1_export / w1_import), the path the GPU model takes (its small layers still all-reduce)."""
    CAP = 64
    packet_has_small = False

    def grad_buckets(self, which):
        return [torch.from_numpy(g) for k, g in self.o.G[which].items() if k != "enc.lin1.weight"]

    def w1_export(self):
        g = self.o.G[self._which]["enc.lin1.weight"]          # dense [h, N]
        rows = np.flatnonzero(np.abs(g).sum(0))
        h = g.shape[0]
        pk = np.zeros(1 + self.CAP + self.CAP * h, dtype=np.float32)
        pk[0] = len(rows)
        pk[1:1 + len(rows)] = rows
        pk[1 + self.CAP:1 + self.CAP + len(rows) * h] = g[:, rows].T.ravel()
        return torch.from_numpy(pk)

    def w1_import(self, packets, n_peers, which):
        g = self.o.G[which]["enc.lin1.weight"]
        h = g.shape[0]
        g[:] = 0
        per = 1 + self.CAP + self.CAP * h
        for p in range(n_peers):
            pk = packets[p * per:(p + 1) * per].numpy()
            n = int(pk[0])
            rows = pk[1:1 + n].astype(np.int64)
            g[:, rows] += pk[1 + self.CAP:1 + self.CAP + n * h].reshape(n, h).T
        self.o.opt_enc.step(self.o.p, {"enc.lin1.weight": g}) if which == 0 else \
            self.o.opt_gen.step(self.o.p, {"enc.lin1.weight": g})

    def ae_encoder_backward(self, dz):
        super().ae_encoder_backward(dz)
        self._which = 0

    def gen_step(self):
        super().gen_step()
        self._which = 2

    def apply_updates(self, which):
        opt = {0: self.o.opt_enc, 1: self.o.opt_dec, 2: self.o.opt_gen, 3: self.o.opt_disc}[which]
        opt.step(self.o.p, {k: v for k, v in self.o.G[which].items() if k != "enc.lin1.weight"})


def _worker_sparse(rank, world, port, name, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import DataParallelAAE
    fx = Fixture(name)
    model = SparseW1Replica(fx.init_params(), **fx.model_kwargs())
    dp = DataParallelAAE(model, dist)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        lo, hi = dp.shard(0, B)
        masks = fx.masks(s)
        if masks is not None:
            masks = [m[lo:hi] for m in masks]
        dp.step((ip, idx, val), lo, hi - lo, global_rows=B, masks=masks, z_real=fx.z[f"step{s}.z_real"][lo:hi])
    if rank == 0:
        ret.update({k: v.copy() for k, v in model.o.p.items()})
    dist.destroy_process_group()


def test_two_ranks_with_packed_first_layer_rows_equal_single_process():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_sparse, 2, lambda port: (2, port, "step_masks", ret))
        got = dict(ret)
    fx = Fixture("step_masks")
    for k, w in fx.expected_params(fx.steps - 1).items():
        np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)


def _worker_cond(rank, world, port, name, ret):
    """Two ranks, half a batch each, with a trainable CategoricalCondition (real torch plugin, CPU table) behind the
    autograd bridge: the plugin's gradients have to be summed over the ranks before its optimiser steps."""
    import types
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import DataParallelAAE
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec import condition as C
    fx = Fixture(name)
    kind = fx.cfg.get("cat", dict(sparse=False, reduce="sum", lr=1e-2))
    model = OracleReplica(fx.init_params(), **fx.model_kwargs())
    dp = DataParallelAAE(model, dist)
    cat = C.CategoricalCondition(8, sparse=kind["sparse"], use_cuda=False, reduce=kind["reduce"], lr=kind["lr"])
    V = fx.z["init.cond.embedding"].shape[0]
    cat.vocab = {"a%d" % i: i for i in range(1, V)}
    cat.embedding = torch.nn.Embedding(V, 8, padding_idx=0, sparse=kind["sparse"])
    with torch.no_grad():
        cat.embedding.weight.copy_(torch.from_numpy(fx.z["init.cond.embedding"]))
    cat.optimizer = (torch.optim.SparseAdam if kind["sparse"] else torch.optim.Adam)(cat.embedding.parameters(), lr=kind["lr"])
    conds = C.ConditionList([("authors", cat)])
    host = types.SimpleNamespace(conditions=conds, _dp=dp)
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        lo, hi = dp.shard(0, B)
        masks = fx.masks(s)
        if masks is not None:
            masks = [m[lo:hi] for m in masks]
        # every rank pads to the GLOBAL batch's width, as the single process does
        lists = [[int(j) for j in row] for row in fx.cond_inputs(s)[0][lo:hi]]
        cond_fn = AdversarialAutoEncoder._cond_fn(host, [lists])
        dp.step((ip, idx, val), lo, hi - lo, global_rows=B, cond_fn=cond_fn, masks=masks,
                z_real=fx.z[f"step{s}.z_real"][lo:hi])
    emb = cat.embedding.weight.detach().clone()
    if rank == 0:
        ret.update({k: v.copy() for k, v in model.o.p.items()})
        ret["cond.embedding"] = emb.numpy().copy()
    other = [torch.empty_like(emb) for _ in range(world)]
    dist.all_gather(other, emb)
    assert all(torch.equal(other[0], o) for o in other)          # the replicas' tables stay identical
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_cond_categorical", "step_cat_sparse_sum"])
def test_two_ranks_with_trainable_condition_equal_single_process(name):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_cond, 2, lambda port: (2, port, name, ret))
        got = dict(ret)
    fx = Fixture(name)
    last = fx.steps - 1
    for k, w in fx.expected_params(last).items():
        np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(got["cond.embedding"], fx.z[f"step{last}.cond.embedding"], atol=1e-5)


# -------------------------------------------------------------------------------------------------------------
# vocabulary-sharded decoder output layer (aaerec.parallel.VocabParallelAAE) over gloo, oracle stand-ins
# -------------------------------------------------------------------------------------------------------------
class VocabLocalReplica(OracleReplica):
    """The document-sharded replica of VocabParallelAAE's interface: ae_forward / dh2_rows / da2_rows / ae_backward
    (include/aaerec_hip.h: aae_ae_forward, AAE_T_ACT_DH2, AAE_T_ACT_DA2, aae_ae_backward) on the oracle."""
    big_tensor_id = 6

    def ae_forward(self, csr, row_start, n_rows, rows=None, cond=None, masks=None, z_real=None):
        z = self.ae_encode(csr, row_start, n_rows, masks=masks, z_real=z_real).numpy()
        o = self.o
        _, self._dc = o._mlp_fwd("dec", z, (o._mk[2], o._mk[3]))      # (its own output layer result is not used)
        self._zc = z
        h = self._dc["h2"].shape[1]
        self._dh2 = torch.zeros(n_rows, h + 1)
        self._dh2[:, :h] = torch.from_numpy(self._dc["h2"])
        self._dh2[:, h] = 1.0
        self._da2 = torch.zeros(n_rows, h + 1)

    def dh2_rows(self, n):
        return self._dh2[:n]

    def da2_rows(self, n):
        return self._da2[:n]

    def ae_backward(self):
        from oracle.aae_oracle import act_bwd, f32
        o, dc = self.o, self._dc
        P = o.p
        gh2 = self._da2[:, :-1].numpy().astype(f32)
        ga2 = dc["d2"].bwd(act_bwd(o.act, dc["u2"], dc["h2"], gh2))
        G = {"dec.lin2.weight": (ga2.T @ dc["h1"]).astype(f32), "dec.lin2.bias": ga2.sum(0).astype(f32)}
        gh1 = (ga2 @ P["dec.lin2.weight"]).astype(f32)
        ga1 = dc["d1"].bwd(act_bwd(o.act, dc["u1"], dc["h1"], gh1))
        G["dec.lin1.bias"] = ga1.sum(0).astype(f32)
        G["dec.lin1.weight"] = (ga1.T @ self._zc).astype(f32)
        o.G[1] = G
        o.ae_encoder_backward((ga1 @ P["dec.lin1.weight"]).astype(f32))

    def grad_buckets(self, which):
        return super().grad_buckets(1 if which == "dec_small" else which)

    def apply_updates(self, which, skip=-1):
        self.o.apply_updates(which)          # (O_DEC: G[1] holds the two small layers only)


class VocabSliceReplica:
    """The item-slice model: aae_output_layer_step on rows [lo, hi) of dec.lin3 with its own (fused) Adam."""

    def __init__(self, params, lo, hi, lr):
        from oracle.aae_oracle import Adam
        self.p = {"w": params["dec.lin3.weight"][lo:hi].copy(), "b": params["dec.lin3.bias"][lo:hi].copy()}
        self.lo, self.hi, self.opt, self.scale = lo, hi, Adam(lr), 1.0
        self._dh2 = self._da2 = None
        self.loss = 0.0

    def set_grad_scale(self, s):
        self.scale = float(s)

    def dh2_rows(self, n):
        if self._dh2 is None or self._dh2.shape[0] != n:
            h = self.p["w"].shape[1]
            self._dh2, self._da2 = torch.zeros(n, h + 1), torch.zeros(n, h + 1)
        return self._dh2

    def da2_rows(self, n):
        return self._da2

    def output_layer_step(self, slice_csr, g_row_start, global_rows, rows=None):
        from oracle.aae_oracle import TINY, f32, sigmoid
        ip, idx, val = slice_csr
        B, Ns = global_rows, self.hi - self.lo
        h2 = self._dh2[:, :-1].numpy().astype(f32)
        xhat = sigmoid((h2 @ self.p["w"].T + self.p["b"]).astype(f32))
        T = np.zeros((B, Ns), dtype=f32)
        for b in range(B):
            r = g_row_start + b
            T[b, idx[ip[r]:ip[r + 1]]] = val[ip[r]:ip[r + 1]]
        x, t = xhat + TINY, T + TINY
        lx, l1x = np.maximum(np.log(x), f32(-100)), np.maximum(np.log1p(-x), f32(-100))
        self.loss = float((-(t * lx + (f32(1) - t) * l1x)).mean(dtype=np.float64))
        gx = (x - t) / np.maximum((f32(1) - x) * x, f32(1e-12)) * f32(self.scale / (B * Ns))
        glog = (gx * xhat * (f32(1) - xhat)).astype(f32)
        self._da2[:, :-1] = torch.from_numpy((glog @ self.p["w"]).astype(f32))
        self.opt.step(self.p, {"w": (glog.T @ h2).astype(f32), "b": glog.sum(0).astype(f32)})

    def losses(self):
        return (self.loss, 0.0, 0.0)


def _worker_vocab(rank, world, port, name, ret):
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import VocabParallelAAE, item_slice
    fx = Fixture(name)
    N = fx.cfg["N"]
    lo, hi = item_slice(N, rank, world)
    model = VocabLocalReplica(fx.init_params(), **fx.model_kwargs())
    sl = VocabSliceReplica(fx.init_params(), lo, hi, fx.model_kwargs().get("gen_lr", 1e-3))
    vp = VocabParallelAAE(model, sl, dist, N)
    losses = []
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        Bl = B // world
        X = sp.csr_matrix((val, idx, ip), shape=(B, N))
        Xs = X[:, lo:hi].tocsr()
        masks = fx.masks(s)
        if masks is not None:
            masks = [m[rank * Bl:(rank + 1) * Bl] for m in masks]
        vp.step((ip, idx, val), rank * Bl, Bl, (Xs.indptr, Xs.indices, Xs.data), 0, B, masks=masks,
                z_real=fx.z[f"step{s}.z_real"][rank * Bl:(rank + 1) * Bl])
        losses.append(vp.recon_loss())
    if rank == 0:
        ret.update({k: v.copy() for k, v in model.o.p.items() if not k.startswith("dec.lin3")})
        ret["recon_losses"] = losses
    ret[f"v3w{rank}"], ret[f"v3b{rank}"] = sl.p["w"].copy(), sl.p["b"].copy()
    flat = torch.from_numpy(np.concatenate([v.ravel() for k, v in model.o.p.items() if not k.startswith("dec.lin3")]))
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(other[0], o) for o in other)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_masks", "step_nodrop_gauss"])
def test_two_ranks_with_vocabulary_sharded_output_layer_equal_single_process(name):
    """VocabParallelAAE over real gloo collectives: rank r holds half the documents and half the items' output rows;
    parameters (replicas + the two slices of dec.lin3) and the reconstruction loss equal the reference's."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_vocab, 2, lambda port: (2, port, name, ret))
        got = dict(ret)
    fx = Fixture(name)
    want = fx.expected_params(fx.steps - 1)
    for k, w in want.items():
        if not k.startswith("dec.lin3"):
            np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(np.concatenate([got["v3w0"], got["v3w1"]]), want["dec.lin3.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(np.concatenate([got["v3b0"], got["v3b1"]]), want["dec.lin3.bias"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(got["recon_losses"], [fx.z[f"step{s}.losses"][0] for s in range(fx.steps)], rtol=1e-5)


# ---- both vocabulary-wide layers with the item slices (VocabParallelAAE(shard_first_layer=True)) -------------------
class BothLocalReplica(VocabLocalReplica):
    """The replica with an EXTERNAL first layer (include/aaerec_hip.h: aae_set_first_layer_external): a1_rows() is filled
    by the caller before ae_forward / disc_step, dL/d(a1) comes back in ga1_rows(); enc.lin1.weight is never touched, its
    bias stays a replicated small parameter."""
    stable_views = False            # (numpy buffers, re-created per step: the driver must not cache views of them)

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        h = params["enc.lin1.weight"].shape[0]
        self._h, self._a1, self._ga1 = h, None, None
        o = self.o
        o.encode = self._encode_ext                      # disc_step / gen_step: Enc(...) from the caller's a1
        o._enc_w1_grad = self._take_ga1                   # ... and dL/d(a1) instead of the first layer's weight gradient

    def set_first_layer_external(self, on=True):
        pass

    def _buf(self, n):
        if self._a1 is None or self._a1.shape[0] != n:
            self._a1, self._ga1 = torch.zeros(n, self._h), torch.zeros(n, self._h)

    def a1_rows(self, n):
        self._buf(n)
        return self._a1

    def ga1_rows(self, n):
        self._buf(n)
        return self._ga1

    def first_layer_bias(self):
        return torch.from_numpy(self.o.p["enc.lin1.bias"])

    def _encode_ext(self, indptr, indices, values, masks=None, input_noise=None):
        o = self.o
        a1 = self._a1.numpy().copy()
        a3, cache = o._mlp_fwd("enc", None, masks, first_pre=a1)
        cache["a1"], cache["s"] = a1, None
        return o._enc_final_fwd(a3), cache

    def _take_ga1(self, indptr, indices, values, s, ga1):
        self._ga1.copy_(torch.from_numpy(np.ascontiguousarray(ga1)))
        return None

    def ae_forward(self, csr, row_start, n_rows, rows=None, cond=None, masks=None, z_real=None):
        o = self.o
        o._b = (np.arange(n_rows + 1), None, None)        # (only the row count is read downstream)
        o._mk = masks if masks is not None else [None] * 12
        o._zr = z_real
        z, o._ec = self._encode_ext(None, None, None, (o._mk[0], o._mk[1]))
        o._z = z
        _, self._dc = o._mlp_fwd("dec", z, (o._mk[2], o._mk[3]))
        self._zc = z
        h = self._dc["h2"].shape[1]
        self._dh2 = torch.zeros(n_rows, h + 1)
        self._dh2[:, :h] = torch.from_numpy(self._dc["h2"])
        self._dh2[:, h] = 1.0
        self._da2 = torch.zeros(n_rows, h + 1)

    def ae_backward(self):
        super().ae_backward()                             # (-> o.ae_encoder_backward -> _take_ga1)
        del self.o.G[0]["enc.lin1.weight"]

    def gen_step(self):
        self.o.gen_step()
        del self.o.G[2]["enc.lin1.weight"]

    def grad_buckets(self, which):
        G = self.o.G
        if which == "enc_dec_small":
            return [torch.from_numpy(g) for g in list(G[0].values()) + list(G[1].values())]
        if which == "enc_small":
            return [torch.from_numpy(g) for g in G[2].values()]
        return super().grad_buckets(which)


class BothSliceReplica(VocabSliceReplica):
    """The item-slice model with its columns of enc.lin1 as well: aae_first_layer_forward / aae_first_layer_update
    (enc_optim and gen_optim keep separate moments, as torch's two Adam instances over the encoder do)."""

    def __init__(self, params, lo, hi, lr, reg_lr, normalize=True):
        from oracle.aae_oracle import Adam
        super().__init__(params, lo, hi, lr)
        self.w1 = {"w1": params["enc.lin1.weight"][:, lo:hi].copy()}
        self.opt_w1 = {0: Adam(lr), 2: Adam(reg_lr)}
        self.normalize, self.l1 = normalize, None
        self._a1 = None

    def join(self):
        pass

    def set_doc_l1(self, l1):
        self.l1 = np.asarray(l1, dtype=np.float32)

    def a1_rows(self, n):
        if self._a1 is None or self._a1.shape[0] != n:
            self._a1 = torch.zeros(n, self.w1["w1"].shape[0])
        return self._a1

    def _rows(self):
        from oracle.aae_oracle import TINY, f32
        ip, idx, val = self._csr
        for b in range(self._B):
            r = self._r0 + b
            lo, hi = ip[r], ip[r + 1]
            s = f32(1) / max(self.l1[r], TINY) if self.normalize else f32(1)
            yield b, idx[lo:hi], (val[lo:hi] * s).astype(f32)

    def first_layer_forward(self, csr=None, row_start=0, n_rows=0, rows=None, bias=None):
        from oracle.aae_oracle import f32
        if csr is not None:
            self._csr, self._r0, self._B = csr, row_start, n_rows
        W = self.w1["w1"]
        a1 = np.zeros((self._B, W.shape[0]), dtype=f32)
        for b, cols, xn in self._rows():
            a1[b] = W[:, cols] @ xn
        if bias is not None:
            a1 += bias.numpy()
        self.a1_rows(self._B).copy_(torch.from_numpy(a1))

    def output_layer_step(self, *a, **kw):
        super().output_layer_step(self._csr, self._r0, self._B)

    def first_layer_update(self, which, ga1=None, rows_per_block=0, block_stride=0):
        from oracle.aae_oracle import f32
        h = self.w1["w1"].shape[0]
        blocks = ga1.numpy().reshape(-1, block_stride)[:, :rows_per_block * h].reshape(-1, h)      # rank-major rows
        g = np.zeros_like(self.w1["w1"])
        for b, cols, xn in self._rows():
            np.add.at(g.T, cols, np.outer(xn, blocks[b]).astype(f32))
        self.opt_w1[which].step(self.w1, {"w1": g})


def _worker_both(rank, world, port, name, ret):
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import VocabParallelAAE, item_slice
    fx = Fixture(name)
    N = fx.cfg["N"]
    lo, hi = item_slice(N, rank, world)
    kw = fx.model_kwargs()
    model = BothLocalReplica(fx.init_params(), **kw)
    sl = BothSliceReplica(fx.init_params(), lo, hi, kw.get("gen_lr", 1e-3), kw.get("reg_lr", 1e-3),
                          normalize=kw.get("normalize_inputs", True))
    vp = VocabParallelAAE(model, sl, dist, N, shard_first_layer=True)
    losses = []
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        Bl = B // world
        X = sp.csr_matrix((val, idx, ip), shape=(B, N))
        Xs = X[:, lo:hi].tocsr()
        sl.set_doc_l1(np.asarray(abs(X).sum(1)).reshape(-1))
        masks = fx.masks(s)
        if masks is not None:
            masks = [m[rank * Bl:(rank + 1) * Bl] for m in masks]
        vp.step((ip, idx, val), rank * Bl, Bl, (Xs.indptr, Xs.indices, Xs.data), 0, B, masks=masks,
                z_real=fx.z[f"step{s}.z_real"][rank * Bl:(rank + 1) * Bl])
        losses.append(vp.recon_loss())
    if rank == 0:
        ret.update({k: v.copy() for k, v in model.o.p.items() if not k.startswith("dec.lin3") and k != "enc.lin1.weight"})
        ret["recon_losses"] = losses
    ret[f"v3w{rank}"], ret[f"v3b{rank}"], ret[f"w1{rank}"] = sl.p["w"].copy(), sl.p["b"].copy(), sl.w1["w1"].copy()
    flat = torch.from_numpy(np.concatenate([v.ravel() for k, v in model.o.p.items()
                                            if not k.startswith("dec.lin3") and k != "enc.lin1.weight"]))
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(other[0], o) for o in other)          # replicated small layers: bitwise the same
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_masks", "step_nodrop_gauss"])
def test_two_ranks_with_both_vocabulary_wide_layers_sharded_equal_single_process(name):
    """VocabParallelAAE(shard_first_layer=True) over real gloo collectives (reduce-scatter of the first layer's shares,
    all-gather of dL/d(a1) with the small layers' gradients behind it, ...): rank r holds half the documents, half the
    items' rows of dec.lin3 AND half the items' columns of enc.lin1; every parameter and the reconstruction loss equal the
    reference's single-process fixtures."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_both, 2, lambda port: (2, port, name, ret))
        got = dict(ret)
    fx = Fixture(name)
    want = fx.expected_params(fx.steps - 1)
    for k, w in want.items():
        if not k.startswith("dec.lin3") and k != "enc.lin1.weight":
            np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(np.concatenate([got["v3w0"], got["v3w1"]]), want["dec.lin3.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(np.concatenate([got["v3b0"], got["v3b1"]]), want["dec.lin3.bias"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(np.concatenate([got["w10"], got["w11"]], axis=1), want["enc.lin1.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(got["recon_losses"], [fx.z[f"step{s}.losses"][0] for s in range(fx.steps)], rtol=1e-5)


class ShardStandIn:
    """dp_mode='shard' on the oracle (include/aaerec_hip.h: aae_shard_step): ONE model per rank = an item slice of both
    vocabulary-wide layers (BothSliceReplica) + every hidden layer with its own optimisers (BothLocalReplica, fed the
    WHOLE batch), three all-reduces of [rows, n_hidden] partial sums and no gradient exchange."""

    def __init__(self, params, lo, hi, rank, **kw):
        self.hid = BothLocalReplica(params, **kw)
        self.sl = BothSliceReplica(params, lo, hi, kw.get("gen_lr", 1e-3), kw.get("reg_lr", 1e-3),
                                   normalize=kw.get("normalize_inputs", True))
        self.rank = rank
        self.collectives = 0

    def set_first_layer_external(self, on=True):
        pass

    def set_doc_l1(self, l1):
        self.sl.set_doc_l1(l1)

    def a1_rows(self, n):
        return self.hid.a1_rows(n)

    def losses(self):
        return (self.sl.loss, 0.0, 0.0)

    def _all_reduce(self, coll, t):
        d, group = coll
        d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
        self.collectives += 1

    def shard_step(self, coll, slice_csr, row_start, n_rows, item_share, rows=None, next_rows=None, cond=None, masks=None,
                   z_real=None):
        hid, sl, B = self.hid, self.sl, n_rows
        h = sl.w1["w1"].shape[0]
        bias = hid.first_layer_bias() if self.rank == 0 else None
        hid.set_grad_scale(1.0)
        # ae phase
        sl.first_layer_forward(slice_csr, row_start, B, bias=bias)
        a1 = sl.a1_rows(B)
        self._all_reduce(coll, a1)
        hid.a1_rows(B).copy_(a1)
        hid.ae_forward(None, 0, B, masks=masks, z_real=z_real)
        sl.dh2_rows(B)[:] = hid.dh2_rows(B)
        sl.set_grad_scale(item_share)
        sl.output_layer_step()
        da2 = sl.da2_rows(B)
        self._all_reduce(coll, da2)
        hid.da2_rows(B).copy_(da2)
        hid.ae_backward()
        hid.apply_updates(0)
        hid.apply_updates(1)
        ga1 = hid.ga1_rows(B).clone().reshape(-1)
        sl.first_layer_update(0, ga1, rows_per_block=B, block_stride=B * h)
        # disc phase (Enc_eval with the updated first layer), gen phase
        sl.first_layer_forward(bias=bias)
        a1 = sl.a1_rows(B)
        self._all_reduce(coll, a1)
        hid.a1_rows(B).copy_(a1)
        hid.disc_step()
        hid.apply_updates(3)
        hid.gen_step()
        hid.apply_updates(2)
        ga1 = hid.ga1_rows(B).clone().reshape(-1)
        sl.first_layer_update(2, ga1, rows_per_block=B, block_stride=B * h)


def _worker_shard(rank, world, port, name, ret):
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.parallel import ItemShardedAAE, item_slice
    fx = Fixture(name)
    N = fx.cfg["N"]
    lo, hi = item_slice(N, rank, world)
    model = ShardStandIn(fx.init_params(), lo, hi, rank, **fx.model_kwargs())
    sh = ItemShardedAAE(None, model, dist, N, interleaved=False)
    losses = []
    for s in range(fx.steps):
        ip, idx, val = fx.batch(s)
        B = len(ip) - 1
        X = sp.csr_matrix((val, idx, ip), shape=(B, N))
        Xs = X[:, lo:hi].tocsr()
        model.set_doc_l1(np.asarray(abs(X).sum(1)).reshape(-1))
        sh.step(None, 0, B, (Xs.indptr, Xs.indices, Xs.data), 0, B, masks=fx.masks(s), z_real=fx.z[f"step{s}.z_real"])
        losses.append(sh.recon_loss())
    small = {k: v.copy() for k, v in model.hid.o.p.items() if not k.startswith("dec.lin3") and k != "enc.lin1.weight"}
    if rank == 0:
        ret.update(small)
        ret["recon_losses"] = losses
        ret["collectives"] = model.collectives
        ret["stats"] = sh.comm_stats()
    ret[f"v3w{rank}"], ret[f"v3b{rank}"], ret[f"w1{rank}"] = model.sl.p["w"].copy(), model.sl.p["b"].copy(), model.sl.w1["w1"].copy()
    flat = torch.from_numpy(np.concatenate([v.ravel() for v in small.values()]))
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert all(torch.equal(other[0], o) for o in other)          # the hidden layers, never exchanged: bitwise the same
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_masks", "step_nodrop_gauss"])
def test_two_ranks_item_sharded_with_replicated_hidden_stacks_equal_single_process(name):
    """ItemShardedAAE (dp_mode='shard') over real gloo collectives, world 2: every rank owns half the items' rows of
    dec.lin3 and columns of enc.lin1 and runs the WHOLE batch through its own copy of the hidden layers; three all-reduces
    of [rows, n_hidden] partial sums per step and nothing else.  Every parameter and the reconstruction loss equal the
    reference's single-process fixtures; the never-exchanged hidden layers are bitwise identical on both ranks."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_shard, 2, lambda port: (2, port, name, ret))
        got = dict(ret)
    fx = Fixture(name)
    want = fx.expected_params(fx.steps - 1)
    for k, w in want.items():
        if not k.startswith("dec.lin3") and k != "enc.lin1.weight":
            np.testing.assert_allclose(got[k], w, atol=1e-5, rtol=0, err_msg=k)
    np.testing.assert_allclose(np.concatenate([got["v3w0"], got["v3w1"]]), want["dec.lin3.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(np.concatenate([got["v3b0"], got["v3b1"]]), want["dec.lin3.bias"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(np.concatenate([got["w10"], got["w11"]], axis=1), want["enc.lin1.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(got["recon_losses"], [fx.z[f"step{s}.losses"][0] for s in range(fx.steps)], rtol=1e-5)
    assert got["collectives"] == 3 * fx.steps and got["stats"]["collectives"] == 3


def _worker_shard_rng(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaerec.aae import AdversarialAutoEncoder
    from aaerec.parallel import ItemShardedAAE, item_slice
    fx = Fixture("step_masks")
    N = fx.cfg["N"]
    lo, hi = item_slice(N, rank, world)
    sh = ItemShardedAAE(None, ShardStandIn(fx.init_params(), lo, hi, rank, **fx.model_kwargs()), dist, N, interleaved=False)
    # the host model of rng_mode='reference', never built (no GPU here): its draw routine and the step's agreement hook
    host = AdversarialAutoEncoder(n_hidden=fx.cfg["h"], n_code=fx.cfg["c"], dropout=(0.2, 0.0), rng_mode="reference", verbose=False)
    torch.manual_seed(1000 + rank)                     # the ranks' generators do NOT coincide
    own = host._host_randomness(24)
    masks, z_real = sh.agree_randomness(*own)
    assert [m is None for m in masks] == [m is None for m in own[0]]
    flat = torch.cat([m.reshape(-1).float() for m in masks if m is not None] + [z_real.reshape(-1)])
    mine = torch.cat([m.reshape(-1).float() for m in own[0] if m is not None] + [own[1].reshape(-1)])
    every = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(every, flat)
    ret[f"agreed{rank}"] = all(torch.equal(every[0], e) for e in every)
    ret[f"own{rank}"] = bool(torch.equal(flat, mine))
    ret[f"dtypes{rank}"] = (str(masks[0].dtype), str(z_real.dtype), tuple(z_real.shape), tuple(masks[0].shape))
    # a model without a prior: AutoEncoder._host_randomness returns (masks, None) - only the masks travel
    from aaerec.aae import AutoEncoder
    ae = AutoEncoder(n_hidden=fx.cfg["h"], n_code=fx.cfg["c"], dropout=(0.2, 0.0), rng_mode="reference", verbose=False)
    own_ae = ae._host_randomness(24)
    assert own_ae[1] is None
    m_ae, z_ae = sh.agree_randomness(*own_ae)
    flat = torch.cat([m.reshape(-1).float() for m in m_ae if m is not None])
    mine = torch.cat([m.reshape(-1).float() for m in own_ae[0] if m is not None])
    every = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(every, flat)
    ret[f"ae{rank}"] = (z_ae is None, all(torch.equal(every[0], e) for e in every), bool(torch.equal(flat, mine)),
                        [m is None for m in m_ae] == [m is None for m in own_ae[0]])
    dist.destroy_process_group()


def test_item_sharded_ranks_with_different_generators_apply_rank_zeros_draws():
    """dp_mode='shard' + rng_mode='reference' (ADVICE r4): the hidden stacks are replicated with no gradient exchange, so
    every rank must apply the SAME dropout masks and prior sample to the whole batch.  Two ranks seeded differently: after
    ItemShardedAAE.agree_randomness both hold rank 0's draws (rank 1's own differ), shapes and dtypes as drawn."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        spawn_ranks(_worker_shard_rng, 2, lambda port: (2, port, ret))
        got = dict(ret)
    assert got["agreed0"] and got["agreed1"]
    assert got["own0"] and not got["own1"]
    assert got["ae0"] == (True, True, True, True) and got["ae1"] == (True, True, False, True)
    fx = Fixture("step_masks")
    assert got["dtypes0"] == got["dtypes1"] == ("torch.uint8", "torch.float32", (24, fx.cfg["c"]), (24, fx.cfg["h"]))


def test_item_ownership_partitions_the_vocabulary():
    """item_items: every item has exactly one owner, interleaved or contiguous, for vocabularies that do not divide by the
    world size; the slice object indexes NumPy arrays and SciPy CSR columns alike."""
    import scipy.sparse as sp
    sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
    from aaerec.parallel import item_items, item_slice, _slice_len
    rng = np.random.default_rng(0)
    for n, world in [(10, 3), (1000, 8), (7, 8), (4587, 4)]:
        X = sp.random(5, n, density=0.3, format="csr", random_state=1, dtype=np.float32)
        for inter in (True, False):
            owned = np.zeros(n, dtype=int)
            cols = 0
            for r in range(world):
                it = item_items(n, r, world, inter)
                owned[it] += 1
                assert _slice_len(it, n) == len(np.arange(n)[it]) == X[:, it].shape[1]
                cols += X[:, it].nnz
                if not inter:
                    assert (it.start, it.stop) == item_slice(n, r, world)
            assert (owned == 1).all() and cols == X.nnz
        w = rng.standard_normal((n, 3))
        np.testing.assert_array_equal(w[item_items(n, 1, world, True)], w[1::world])
