"""Data-parallel AAE step: one process per GPU, gradients summed over RCCL (xGMI).

New functionality defined by BASELINE.json (the reference is single-device, aae.py:794-797).
Every rank holds a full replica (parameters + four optimiser states) and a shard of the global
batch.  The loss is a mean over the GLOBAL batch, so each rank scales its loss gradients by
local_rows / global_rows (aae_set_grad_scale) and the all-reduce is a plain SUM; after it every
rank runs the identical optimiser update, so replicas stay bit-identical.

Exchange points per step (the order the reference's single-process step imposes):
    after decoder backward      dec grads  (V3 [N,h+1] dense, V1, V2): asynchronous all-reduce,
                                waited for at the end of the step          -> dec_optim
    after encoder backward      enc small grads all-reduce; W1T row-sparse:
                                all-gather of packed rows                  -> enc_optim
    after disc_step             disc grads (~50 k floats)                  -> disc_optim
    after gen_step              enc grads again (small + packed rows)      -> gen_optim

`model` is anything with the export-mode phase interface of aaerec._hip.HipAAE
(ae_encode / ae_decode_backward / ae_encoder_backward / disc_step / gen_step / apply_updates /
set_grad_scale / grad_buckets), which is what lets the gloo CPU tests drive this file with an
oracle-backed stand-in.
"""
O_ENC, O_DEC, O_GEN, O_DISC = 0, 1, 2, 3


class DataParallelAAE:
    def __init__(self, model, dist, group=None, shard_decoder=True):
        self.model, self.dist, self.group = model, dist, group
        self.shard_decoder = shard_decoder
        self._shard_grad = self._shard_p = self._pending = None
        self.world = dist.get_world_size(group)
        self.global_rows = None
        self._w1_all = self._w1_ext = None
        self._coll = [0, 0]          # collectives / payload bytes (this rank's send side) of the running step
        self._coll_last = (0, 0)
        # upper bound on the distinct items of any rank's share of the running batch, agreed by all ranks (e.g. the
        # largest entry count of a share): sizes the first-layer packets of this step; None = the model-wide worst case
        self.w1_rows = None

    def shard(self, start, stop):
        """Contiguous share [lo, hi) of the global batch [start, stop) for this rank, or
        (None, None) when the batch has fewer rows than ranks (skipped on every rank)."""
        n = stop - start
        if n < self.world:
            return None, None
        rank = self.dist.get_rank(self.group)
        base, extra = divmod(n, self.world)
        lo = start + rank * base + min(rank, extra)
        return lo, lo + base + (1 if rank < extra else 0)

    def sync_conditions(self, conditions):
        """Sum the parameter gradients of trainable condition plugins (embedding tables with their own optimiser,
        condition.py:372-508) over the ranks, between their backward and their optimiser step: every rank saw its
        share of the global batch only, and the loss gradients are already scaled to the global mean.  Row-sparse
        gradients (nn.Embedding(sparse=True) + SparseAdam) are summed densely and handed back row-sparse, so the
        optimiser moves the rows the GLOBAL batch named - what the single-process step does."""
        import torch
        d = self.dist
        nccl = str(d.get_backend(self.group)).lower() == "nccl"
        for cond in conditions.values():
            opt = getattr(cond, "optimizer", None)
            if opt is None:
                continue
            want_sparse = isinstance(opt, torch.optim.SparseAdam)
            for group in opt.param_groups:
                for p in group["params"]:
                    if not p.requires_grad:
                        continue
                    g = p.grad
                    dense = torch.zeros_like(p.data) if g is None else (g.to_dense() if g.is_sparse else g)
                    buf = dense.to(self.model.device) if nccl and not dense.is_cuda else dense.contiguous()
                    d.all_reduce(buf, op=d.ReduceOp.SUM, group=self.group)
                    dense = buf.to(p.device)
                    p.grad = dense.to_sparse(1) if want_sparse else dense

    def _set_rng_rows(self, n_rows, global_rows):
        """Tell the model which rows of the global batch it holds (contiguous shares in rank order, as shard() deals
        them): its device RNG then draws what a single process draws for those rows."""
        if hasattr(self.model, "set_rng_rows"):
            rank = self.dist.get_rank(self.group)
            base, extra = divmod(int(global_rows), self.world)
            self.model.set_rng_rows(rank * base + min(rank, extra), global_rows)

    def _count(self, t):
        self._coll[0] += 1
        self._coll[1] += t.numel() * t.element_size()

    def comm_stats(self):
        """{'collectives': n, 'bytes': b} of the last completed step: how many collectives this rank took part in and the
        payload it contributed to them (bench.py reports it)."""
        return {"collectives": self._coll_last[0], "bytes": self._coll_last[1]}

    def _allreduce(self, which, async_op=False):
        out = []
        for t in self.model.grad_buckets(which):
            self._count(t)
            out.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        return out

    def _exchange_encoder(self, which, ride=None):
        """All encoder gradients of one optimiser step (enc_optim after the ae phases, gen_optim after
        gen_step).  GPU model: one all-gather of packets (packed first-layer rows + the small layers).
        Stand-ins without the packed path: all-reduce of dense buckets.
        ride: (gradient views, then()) - other small gradients that are due at the same point of the step travel in the
        same packets (one collective instead of two: at ~20-40 us of latency each on a ~0.5 ms step the count matters
        more than the bytes); then() runs once they are summed."""
        packed = hasattr(self.model, "w1_export")
        if not packed or getattr(self.model, "packet_has_small", True) is False:
            self._allreduce(which)
            self.model.apply_updates(which)
        if ride is not None and not packed:
            for t in ride[0]:
                self._count(t)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            ride[1]()
            ride = None
        self._exchange_w1(which, ride)

    def _exchange_w1(self, which, ride=None):
        """First encoder layer: only the rows of the items in the global batch carry gradient, so the
        ranks all-gather their packed rows (a few MB) instead of all-reducing the dense [N, h] tensor."""
        m = self.model
        if not hasattr(m, "w1_export"):
            return                      # dense stand-in: the gradient is part of grad_buckets()
        cap = self.w1_rows
        pk = m.w1_export() if cap is None else m.w1_export(cap)
        total = pk.numel()
        extra = [t.reshape(-1) for t in ride[0]] if ride is not None else []
        if extra:                       # the riders sit behind the packet proper
            n_ext = total + sum(t.numel() for t in extra)
            if self._w1_ext is None or self._w1_ext.numel() < n_ext:
                self._w1_ext = pk.new_empty(n_ext)
            ext = self._w1_ext[:n_ext]
            ext[:total].copy_(pk)
            off = total
            for t in extra:
                ext[off:off + t.numel()].copy_(t)
                off += t.numel()
            pk = ext
        if self.world == 1:
            allp = pk
        else:
            need = pk.numel() * self.world
            if self._w1_all is None or self._w1_all.numel() < need:
                self._w1_all = pk.new_empty(need)
            allp = self._w1_all[:need]
            self._count(pk)
            self.dist.all_gather_into_tensor(allp, pk, group=self.group)
        if extra:
            # summed over the peers in one fixed order on identical inputs: bitwise the same on every rank
            peers = allp.view(self.world, pk.numel())
            off = total
            for t in extra:
                t.copy_(peers[:, off:off + t.numel()].sum(0))
                off += t.numel()
            ride[1]()
        if extra:
            m.w1_import(allp, self.world, which, cap, stride_floats=pk.numel())
        elif cap is None:
            m.w1_import(allp, self.world, which)
        else:
            m.w1_import(allp, self.world, which, cap)

    # ---- decoder output layer: reduce-scatter -> optimiser on this rank's rows -> all-gather ------
    def _dec_start(self):
        """Start the exchange of the decoder gradients; returns the state _dec_finish needs."""
        m, d = self.model, self.dist
        shardable = ((self.world > 1 or self.shard_decoder == "force") and self.shard_decoder and hasattr(m, "big_grad")
                     and hasattr(d, "reduce_scatter_tensor"))
        if shardable:
            tid, g, p = m.big_grad()
            if g.shape[0] % self.world == 0:
                rows = g.shape[0] // self.world
                if self._shard_grad is None:
                    self._shard_grad = g.new_empty((rows, g.shape[1]))
                    self._shard_p = g.new_empty((rows, g.shape[1]))
                work = [d.reduce_scatter_tensor(self._shard_grad, g, op=d.ReduceOp.SUM, group=self.group, async_op=True)]
                work += [d.all_reduce(t, op=d.ReduceOp.SUM, group=self.group, async_op=True)
                         for t in m.grad_buckets("dec_small")]
                return ("shard", tid, p, rows, work)
        return ("dense", self._allreduce(O_DEC, async_op=True))

    def _dec_finish(self, st):
        m, d = self.model, self.dist
        if st[0] == "dense":
            for w in st[1]:
                if w is not None:
                    w.wait()
            m.apply_updates(O_DEC)
            return
        _, tid, p, rows, work = st
        for w in work:
            if w is not None:
                w.wait()
        rank = d.get_rank(self.group)
        m.apply_updates(O_DEC, skip=tid)                              # the small decoder layers
        m.apply_shard(tid, rank * rows, (rank + 1) * rows, self._shard_grad, O_DEC)
        self._shard_p.copy_(p[rank * rows:(rank + 1) * rows])
        # the updated rows travel while the next step's encoder forward runs; waited for before the
        # decoder is used again
        self._pending = d.all_gather_into_tensor(p.view(-1), self._shard_p.view(-1), group=self.group, async_op=True)

    def wait_pending(self):
        """Block the stream until a still-travelling parameter all-gather has landed (call before
        anything reads the decoder's output layer: next decode, predict, state export)."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def step(self, csr, row_start, n_rows, global_rows=None, rows=None, cond_fn=None, masks=None, z_real=None):
        """cond_fn(z) -> (zc, backward(dzc) -> dz) for condition plugins; None = no condition."""
        m = self.model
        self._coll = [0, 0]
        if global_rows is None:
            global_rows = n_rows * self.world
        m.set_grad_scale(n_rows / float(global_rows))
        self._set_rng_rows(n_rows, global_rows)
        z = m.ae_encode(csr, row_start, n_rows, rows=rows, masks=masks, z_real=z_real)
        self.wait_pending()
        if cond_fn is None:
            dz = m.ae_decode_backward(z)
        else:
            zc, back = cond_fn(z)
            dz = back(m.ae_decode_backward(zc))
        # the decoder is not touched again before the next step: its (large) gradient travels while
        # the encoder backward, disc_step and gen_step run
        dec_state = self._dec_start()
        m.ae_encoder_backward(dz)
        self._exchange_encoder(O_ENC)
        if not getattr(m, "ae_only", False):         # the plain AutoEncoder has no disc_step / gen_step (aae.py:221-458)
            m.disc_step()
            self._allreduce(O_DISC)
            m.apply_updates(O_DISC)
            m.gen_step()
            self._exchange_encoder(O_GEN)
        self._dec_finish(dec_state)
        self._coll_last = tuple(self._coll)


def broadcast_array(dist, group, arr, device=None, src=0):
    """numpy array of rank `src` on every rank (same shape / dtype everywhere).  RCCL moves device tensors only."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    nccl = str(dist.get_backend(group)).lower() == "nccl"
    buf = t.to(device) if nccl else t.clone()
    dist.broadcast(buf, src, group=group)
    return buf.cpu().numpy()


def item_slice(n_items, rank, world):
    """[lo, hi) of the items whose decoder output rows rank `rank` of `world` owns (contiguous, sizes differ by <= 1)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def item_items(n_items, rank, world, interleaved=True):
    """The items rank `rank` of `world` owns, as a slice object (indexes NumPy / torch / SciPy-CSR columns alike).
    interleaved: items rank, rank + world, rank + 2 world, ... - vocabularies are frequency-sorted (datasets.py: the most
    frequent token first), so contiguous slices would hand rank 0 the popular head and most of every batch's entries
    (the slowest rank sets the step: 0.443 -> 0.432 ms of per-rank compute at world 8); else item_slice()'s [lo, hi)."""
    if interleaved:
        return slice(rank, n_items, world)
    return slice(*item_slice(n_items, rank, world))


def _slice_len(sl, n):
    return len(range(*sl.indices(n)))


class _Views:
    """name -> view, looked up once (stable) or on every access."""

    def __init__(self, make, stable):
        self.make, self.stable, self.have = make, stable, {}

    def __getitem__(self, k):
        if not self.stable:
            return self.make[k]()
        if k not in self.have:
            self.have[k] = self.make[k]()
        return self.have[k]


class VocabParallelAAE(DataParallelAAE):
    """Documents are sharded over the ranks for everything except the decoder's output layer, which is sharded over
    the VOCABULARY.

    dec.lin3 holds ~all of the decoder's parameters ([n_items, n_hidden + 1]) and, because the BCE runs over every
    item, ALL of its rows receive gradient every step: replicated data parallelism has to move that dense gradient
    (80 MB at 100 k items, reduce-scatter + all-gather of the updated rows) per step.  Here each rank owns the rows
    of a contiguous slice of the items in a second model (`slice_model`, created with n_items = slice size and the
    fused optimiser) and the ranks exchange hidden activations instead - [global rows, n_hidden] floats, 0.6 MB at
    8 x 100 rows:

        model.ae_forward             encoder + decoder hidden layers on this rank's documents        -> dh2
        all-gather dh2               -> the slice model sees the hidden activations of the GLOBAL batch
        slice.output_layer_step      logits for its items, BCE against the global batch restricted to them, dV3 + Adam
                                     on its rows (no exchange: the rows are nobody else's), partial dL/d(dh2)
        reduce-scatter dL/d(dh2)     summed over the item slices, this rank's documents back
        model.ae_backward            decoder hidden + encoder backward on this rank's documents
        ... small decoder layers all-reduced, encoder packets, disc_step, gen_step exactly as DataParallelAAE.

    The global batch must divide evenly over the ranks (callers trim a tail batch).  `slice_csr` is the corpus
    restricted to this rank's items with ids rebased to the slice; every rank walks the same permutation, so the
    global batch needs no exchange."""

    def __init__(self, model, slice_model, dist, n_items, group=None, shard_first_layer=False, interleaved=False):
        super().__init__(model, dist, group=group, shard_decoder=False)
        self.slice = slice_model
        self.n_items = n_items
        # interleaved: the slice model holds items rank, rank + world, ... (item_items) instead of a contiguous range
        self.interleaved = bool(interleaved)
        self.items = item_items(n_items, dist.get_rank(group), self.world, self.interleaved)
        self.n_slice = _slice_len(self.items, n_items)
        self.item_lo, self.item_hi = item_slice(n_items, dist.get_rank(group), self.world)     # (contiguous form only)
        slice_model.set_grad_scale(self.n_slice / float(n_items))
        # shard_first_layer: enc.lin1 - the other [n_items, n_hidden] matrix - lives with the item slices as well (see
        # _step_both_sharded); the caller gives the slice model the documents' complete L1 norms (set_doc_l1)
        self.shard_first = bool(shard_first_layer)
        self._pk = self._pk_all = None
        self._pkt = {}
        self._view_cache = {}
        if self.shard_first:
            model.set_first_layer_external(True)
        # the step as one library call (aae_dp_step) when both models are library handles: RCCL collectives on a
        # communicator the library creates (backend nccl), or - any other torch.distributed-like object: the host-staged
        # gloo wrapper of the one-GPU tests, single-process stand-ins - Python callbacks.  AAE_DP_PYTHON=1 keeps the
        # phase-by-phase Python driver (_step_both_sharded), which is also what CPU stand-in models run.
        self._native = self._native_keep = None
        import os
        if self.shard_first and hasattr(model, "dp_step") and hasattr(slice_model, "handle") and os.environ.get("AAE_DP_PYTHON") is None:
            if str(dist.get_backend(group)).lower() == "nccl":
                # every rank takes the same driver: rccl_collectives agrees (in lock-step collectives) on whether the library's
                # communicator came up everywhere; if not, all keep the phase-by-phase driver over torch.distributed (RCCL too)
                self._native_keep = rccl_collectives(model, dist, group)
                self._native = None if self._native_keep is None else self._native_keep.table
            elif str(dist.get_backend(group)).lower() == "echo":     # (tools/vocab_rank_time.py: device-side stand-ins)
                import ctypes as C
                from . import _hip
                self._native = _hip.AaeCollectives()
                _hip._check(model.lib.aae_echo_collectives(self.world, C.byref(self._native)))
            else:
                self._native, self._native_keep = python_collectives(model, dist, group)

    def _native_stats(self, n_rows, global_rows):
        """(collectives, payload bytes this rank contributes) of one native step: the seven exchanges of DESIGN.md 5.0"""
        m = self.model
        ld = m.ga1_rows(1).stride(0)
        blk = n_rows * ld * 4
        pk_ae = m.ga1_packet(n_rows, True)[0].numel() * 4
        pk_gen = m.ga1_packet(n_rows, False)[0].numel() * 4
        disc = sum(t.numel() for t in m.grad_buckets(O_DISC)) * 4
        if getattr(m, "ae_only", False):
            return (4, 2 * global_rows * ld * 4 + blk + pk_ae)
        return (7, 3 * global_rows * ld * 4 + blk + pk_ae + pk_gen + disc)

    def step(self, csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows=None, g_rows=None, cond=None,
             masks=None, z_real=None):
        """rows / row_start select this rank's documents in `csr`; g_rows / g_row_start the global batch in
        `slice_csr` (rank-major: rank r's documents are global rows [r * n_rows, (r + 1) * n_rows))."""
        m, sl, d = self.model, self.slice, self.dist
        if global_rows != n_rows * self.world:
            raise ValueError("vocabulary-sharded step: the global batch must divide evenly over the ranks")
        self._coll = [0, 0]
        m.set_grad_scale(n_rows / float(global_rows))
        self._set_rng_rows(n_rows, global_rows)
        if self.shard_first and self._native is not None:
            # the whole step - kernels and collectives - enqueued by ONE library call (aae_dp_step, csrc/dp_step.h)
            if getattr(self, "_reserved", None) != n_rows:      # (set-up, once per batch shape: no allocation inside a step)
                m.dp_reserve(n_rows, self.world)
                self._reserved = n_rows
            m.dp_step(sl, self._native, csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows=rows, g_rows=g_rows,
                      cond=cond, masks=masks, z_real=z_real)
            self._gathered = False
            self._coll_last = self._native_stats(n_rows, global_rows)
            return None
        if self.shard_first:
            return self._step_both_sharded(csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows, g_rows, cond,
                                           masks, z_real)
        m.ae_forward(csr, row_start, n_rows, rows=rows, cond=cond, masks=masks, z_real=z_real)
        self._count(m.dh2_rows(n_rows))
        d.all_gather_into_tensor(sl.dh2_rows(global_rows).view(-1), m.dh2_rows(n_rows).view(-1), group=self.group)
        sl.output_layer_step(slice_csr, g_row_start, global_rows, rows=g_rows)
        self._count(sl.da2_rows(global_rows))
        d.reduce_scatter_tensor(m.da2_rows(n_rows).view(-1), sl.da2_rows(global_rows).view(-1), op=d.ReduceOp.SUM,
                                group=self.group)
        m.ae_backward()
        # the decoder's hidden-layer gradients ride in the encoder's packets: 5 collectives per step, each on the
        # critical path of the next phase (dh2 -> dA2 -> encoder + decoder-hidden -> discriminator -> encoder again)
        self._exchange_encoder(O_ENC, ride=(m.grad_buckets("dec_small"),
                                            lambda: m.apply_updates(O_DEC, skip=m.big_tensor_id)))
        if not getattr(m, "ae_only", False):
            m.disc_step()
            self._allreduce(O_DISC)
            m.apply_updates(O_DISC)
            m.gen_step()
            self._exchange_encoder(O_GEN)
        self._gathered = False
        self._coll_last = tuple(self._coll)

    # ---- both vocabulary-wide matrices with their item slices -----------------------------------------------------
    def _views(self, n_rows, global_rows):
        """The arena views a step exchanges, looked up once per batch shape (each lookup is a few library calls, and
        the step is a chain of ~40 short launches).  Views do not wait for a deferred optimiser launch of the slice
        model: step() joins it once.  (A model whose buffers are not fixed arena views - the CPU stand-ins of the gloo
        tests - says so with stable_views = False and is asked every time.)"""
        key = (n_rows, global_rows)
        v = self._view_cache.get(key)
        if v is None:
            m, sl = self.model, self.slice
            v = _Views(dict(
                a1=lambda: m.a1_rows(n_rows).view(-1), a1_all=lambda: sl.a1_rows(global_rows).view(-1),
                dh2=lambda: m.dh2_rows(n_rows).view(-1), dh2_all=lambda: sl.dh2_rows(global_rows).view(-1),
                da2=lambda: m.da2_rows(n_rows).view(-1), da2_all=lambda: sl.da2_rows(global_rows).view(-1),
                ga1=lambda: m.ga1_rows(n_rows).reshape(-1),
                bias=lambda: m.first_layer_bias() if self.dist.get_rank(self.group) == 0 else None,
                small_ae=lambda: [t.reshape(-1) for t in m.grad_buckets("enc_dec_small")],
                small_gen=lambda: [t.reshape(-1) for t in m.grad_buckets("enc_small")],
                disc=lambda: m.grad_buckets(O_DISC)), getattr(m, "stable_views", True))
            self._view_cache = {key: v}               # (one shape at a time: the tail batch of an epoch replaces it)
        return v

    def _first_layer(self, v, open_step=None):
        """The slices' shares of x * enc.lin1^T for the global batch, summed, this rank's rows -> model.a1_rows
        (rank 0's share carries the bias)."""
        sl, d = self.slice, self.dist
        if open_step is not None:
            slice_csr, g_row_start, global_rows, g_rows = open_step
            sl.first_layer_forward(slice_csr, g_row_start, global_rows, rows=g_rows, bias=v["bias"])
        else:
            sl.first_layer_forward(bias=v["bias"])
        self._count(v["a1_all"])
        d.reduce_scatter_tensor(v["a1"], v["a1_all"], op=d.ReduceOp.SUM, group=self.group)

    def _exchange_packet(self, n_rows, with_decoder):
        """The same exchange when the model keeps dL/d(a1) and the small layers' gradient span contiguous in its arena
        (HipAAE.ga1_packet): the packet is a view - no packing launch; returns what first_layer_update / apply_gathered
        take (buffer, rows per block, block stride, offset of the gradient span)."""
        d = self.dist
        ent = self._pkt.get((n_rows, with_decoder))
        if ent is None:                               # (looked up once per batch shape, like _views)
            pk, n0, span_off = self.model.ga1_packet(n_rows, with_decoder)
            ent = (pk, span_off, pk.new_empty(pk.numel() * self.world) if self.world > 1 else pk)
            self._pkt = {k: v for k, v in self._pkt.items() if k[0] == n_rows}
            self._pkt[(n_rows, with_decoder)] = ent
        pk, span_off, allp = ent
        if self.world > 1:
            self._count(pk)
            d.all_gather_into_tensor(allp, pk, group=self.group)
        return allp, n_rows, pk.numel(), span_off

    def _exchange_ga1(self, n_rows, ga1, riders, fused_apply=False):
        """dL/d(a1) of every rank's documents gathered (rank-major = global batch order; returned as (buffer, rows per
        block, block stride) for first_layer_update); `riders` (small gradient spans due at the same point) travel behind
        the rows and come back summed over the ranks in one fixed order - bitwise the same everywhere (fused_apply: the
        caller's optimiser launch sums them itself, model.apply_gathered)."""
        import torch
        d = self.dist
        n0 = ga1.numel()
        total = n0 + sum(t.numel() for t in riders)
        if self._pk is None or self._pk.numel() != total:
            self._pk = ga1.new_empty(total)
            self._pk_all = ga1.new_empty(total * self.world)
        pk, allp = self._pk, self._pk_all
        torch.cat([ga1] + riders, out=pk)
        if self.world == 1:
            allp = pk
        else:
            self._count(pk)
            d.all_gather_into_tensor(allp, pk, group=self.group)
        if not (fused_apply and len(riders) == 1):
            peers = allp.view(self.world, total)
            off = n0
            for t in riders:
                torch.sum(peers[:, off:off + t.numel()], dim=0, out=t)
                off += t.numel()
        return allp, n_rows, total          # the slice reads the rows where they landed (first_layer_update(ga1=...))

    def _step_both_sharded(self, csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows, g_rows, cond, masks,
                           z_real):
        """enc.lin1 ([n_hidden, n_items]) with the item slices like dec.lin3: under replication its row-sparse gradient
        is the largest exchange of the step (packed rows of every rank: ~60 MB gathered at 8 x 100 documents) and every
        rank carries the whole matrix with two optimisers' moments.  Here a slice computes its items' share of the first
        layer's pre-activations for the GLOBAL batch (it holds those columns of the corpus anyway), the shares are
        reduce-scattered ([global rows, n_hidden]: 0.6 MB), and dL/d(a1) travels back by all-gather with the small
        layers' gradients behind it (the first layer's bias is one of them: it stays replicated, the slice of rank 0 adds
        it to its share); every owner applies enc_optim / gen_optim to its rows.  7 collectives per step, all small:
            a1 reduce-scatter | dh2 all-gather | dA2 reduce-scatter | ga1 + enc/dec small all-gather |
            a1 reduce-scatter (Enc_eval) | disc all-reduce | ga1 + enc small all-gather."""
        m, sl, d = self.model, self.slice, self.dist
        sl.join()                                   # (a deferred optimiser launch of the slice model's last step, if any)
        v = self._views(n_rows, global_rows)
        self._first_layer(v, open_step=(slice_csr, g_row_start, global_rows, g_rows))
        m.ae_forward(csr, row_start, n_rows, rows=rows, cond=cond, masks=masks, z_real=z_real)
        self._count(v["dh2"])
        d.all_gather_into_tensor(v["dh2_all"], v["dh2"], group=self.group)
        sl.output_layer_step()                      # continues the slice model's step (opened by first_layer_forward)
        self._count(v["da2_all"])
        d.reduce_scatter_tensor(v["da2"], v["da2_all"], op=d.ReduceOp.SUM, group=self.group)
        m.ae_backward()
        fused = hasattr(m, "apply_gathered")        # one launch: the peers' sum + enc_optim + dec_optim's small layers
        packed = fused and hasattr(m, "ga1_packet")  # ... and the packet a view of the arena: no packing launch
        if packed:
            g, rpb, bs, off = self._exchange_packet(n_rows, True)
            m.apply_gathered(O_ENC, O_DEC, g, bs, self.world, off)
        elif fused:
            g, rpb, bs = self._exchange_ga1(n_rows, v["ga1"], v["small_ae"], fused)
            m.apply_gathered(O_ENC, O_DEC, g, bs, self.world, v["ga1"].numel())
        else:
            g, rpb, bs = self._exchange_ga1(n_rows, v["ga1"], v["small_ae"], fused)
            m.apply_updates(O_ENC)
            m.apply_updates(O_DEC, skip=m.big_tensor_id)
        sl.first_layer_update(O_ENC, g, rpb, bs)
        if not getattr(m, "ae_only", False):
            self._first_layer(v)
            m.disc_step()
            for t in v["disc"]:
                self._count(t)
                d.all_reduce(t, op=d.ReduceOp.SUM, group=self.group)
            m.apply_updates(O_DISC)
            m.gen_step()
            if packed:
                g, rpb, bs, off = self._exchange_packet(n_rows, False)
                m.apply_gathered(O_GEN, -1, g, bs, self.world, off)
            elif fused:
                g, rpb, bs = self._exchange_ga1(n_rows, v["ga1"], v["small_gen"], fused)
                m.apply_gathered(O_GEN, -1, g, bs, self.world, v["ga1"].numel())
            else:
                g, rpb, bs = self._exchange_ga1(n_rows, v["ga1"], v["small_gen"], fused)
                m.apply_updates(O_GEN)
            sl.first_layer_update(O_GEN, g, rpb, bs)
        self._gathered = False
        self._coll_last = tuple(self._coll)

    def gather_rows(self, t):
        """[world * rows, cols] tensor of every rank's [rows, cols] block, in rank (= global batch) order."""
        import torch
        t = t.contiguous()
        out = torch.empty(self.world * t.shape[0], t.shape[1], dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=self.group)
        return out

    def gather_output_layer(self):
        """Copy every rank's rows of dec.lin3 into each replica's full copy (predict, state_dict; the optimiser state
        of those rows stays with their owner).  Collective; a no-op when nothing changed since the last call."""
        import torch
        if getattr(self, "_gathered", False):
            return
        m, sl, d = self.model, self.slice, self.dist
        full = m.tensor(m.big_tensor_id, padded=True)
        mine = sl.tensor(m.big_tensor_id, padded=True)
        rows = -(-self.n_items // self.world)
        send = torch.zeros(rows, full.shape[1], dtype=full.dtype, device=full.device)
        send[:mine.shape[0]] = mine
        recv = torch.empty(self.world * rows, full.shape[1], dtype=full.dtype, device=full.device)
        d.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        for r in range(self.world):
            it = item_items(self.n_items, r, self.world, self.interleaved)
            full[it] = recv[r * rows:r * rows + _slice_len(it, self.n_items)]
        if self.shard_first:
            # enc.lin1 the same way (stored transposed: one row per item), and the bias every owner keeps
            from ._hip import T_ENC_W1T
            sl.sync()                                   # (the slice's deferred Adam on its rows, replayed)
            m.sync()
            full, mine = m.tensor(T_ENC_W1T, padded=True), sl.tensor(T_ENC_W1T, padded=True)
            send = torch.zeros(rows, full.shape[1], dtype=full.dtype, device=full.device)
            send[:mine.shape[0]] = mine
            recv = torch.empty(self.world * rows, full.shape[1], dtype=full.dtype, device=full.device)
            d.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
            for r in range(self.world):
                it = item_items(self.n_items, r, self.world, self.interleaved)
                full[it] = recv[r * rows:r * rows + _slice_len(it, self.n_items)]
            m.params_changed()
        self._gathered = True

    def recon_loss(self):
        """Reconstruction loss of the last step over all items: the slices' means weighted by their sizes."""
        import torch
        part = torch.tensor([self.slice.losses()[0] * self.n_slice / float(self.n_items)],
                            dtype=torch.float64)
        nccl = str(self.dist.get_backend(self.group)).lower() == "nccl"
        buf = part.to(self.model.device) if nccl else part
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(buf.item())


class ItemShardedAAE:
    """dp_mode='shard' (r4; csrc/dp_step.h aae_shard_step): the third data-parallel scheme.

    Every rank holds ONE training handle (`slice_model`): its item slice of the two vocabulary-wide layers (rows of
    dec.lin3, columns of enc.lin1, with their optimiser states) and a full copy of every hidden layer, and runs the WHOLE
    global batch through the hidden stacks.  Same inputs -> same activations, same small-layer gradients, same optimiser
    updates on every rank: the hidden layers stay identical with NO gradient exchange.  What crosses the ranks are the
    three partial sums over the item slices, each ONE all-reduce of [global rows, n_hidden] floats:

        x * enc.lin1^T (ae phase)  |  dL/d(dh2) of the output layer  |  x * enc.lin1^T (Enc_eval of the disc phase)

    3 collectives per partial_fit (the both-sharded scheme: 7, plus gradient packets and a second handle's step), at the
    price of the hidden stacks running on world x the rows - launches that are latency-bound on a mostly idle chip.
    `model` is the full-vocabulary handle the replica API reads after fit() (predict, state_dict): gather_output_layer()
    fills it from the slices; it takes no part in a step."""
    shard_first = True

    def __init__(self, model, slice_model, dist, n_items, group=None, interleaved=True, collectives=None, max_rows=None):
        """collectives='ipc' (r6): the three all-reduces as one-shot launches over peer-mapped mailboxes (ipc_collectives: the
        ranks of ONE node; max_rows = the largest global batch) instead of the backend's own - RCCL's ring under 'nccl', host
        staging under 'gloo'; falls back to those when a mailbox cannot be shared."""
        self.model, self.slice, self.dist, self.group = model, slice_model, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_items = n_items
        self.interleaved = bool(interleaved)
        self.items = item_items(n_items, self.rank, self.world, self.interleaved)
        self.n_slice = _slice_len(self.items, n_items)
        self.global_rows = None
        self.w1_rows = None
        self._gathered = False
        self._coll_last = (0, 0)
        slice_model.set_first_layer_external(True)
        self._native = self._native_keep = None
        backend = str(dist.get_backend(group)).lower()
        if not hasattr(slice_model, "handle"):
            # a stand-in without the library behind it (the CPU models of tests/test_parallel_gloo.py): it gets the
            # torch.distributed-like object itself and all-reduces its three partial sums through it
            self._native = (dist, group)
        elif collectives == "ipc" and self.world > 1:
            ld = slice_model.a1_rows(1).stride(0)
            rows = int(max_rows if max_rows is not None else slice_model.max_batch)
            self._native_keep = ipc_collectives(slice_model, getattr(dist, "d", dist), rows * ld, group)     # (HostStagedCollectives: its gloo group)
            if self._native_keep is not None:
                self._native = self._native_keep.table
        if self._native is not None:
            pass
        elif backend == "nccl":
            self._native_keep = rccl_collectives(slice_model, dist, group)
            if self._native_keep is not None:
                self._native = self._native_keep.table
        elif backend == "echo":                     # (tools/vocab_rank_time.py: device-side stand-ins)
            import ctypes as C
            from . import _hip
            self._native = _hip.AaeCollectives()
            _hip._check(slice_model.lib.aae_echo_collectives(self.world, C.byref(self._native)))
        if self._native is None:                    # any other torch.distributed-like object (host-staged gloo, stand-ins;
            self._native, self._native_keep = python_collectives(slice_model, dist, group)     # RCCL through callbacks as the fallback)

    def shard(self, start, stop):
        """Every rank takes the whole global batch [start, stop)."""
        return (start, stop) if stop > start else (None, None)

    def step(self, csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows=None, g_rows=None, cond=None,
             masks=None, z_real=None, next_rows=None):
        """g_rows / g_row_start select the global batch in `slice_csr` (this rank's columns of the corpus); cond, masks,
        z_real cover all of its rows.  (csr / rows / row_start - a rank's own share in the other schemes - are unused.)"""
        self.slice.shard_step(self._native, slice_csr, g_row_start, global_rows, self.n_slice / float(self.n_items),
                              rows=g_rows, next_rows=next_rows, cond=cond, masks=masks, z_real=z_real)
        self._gathered = False
        ld = self.slice.a1_rows(1).stride(0)
        self._coll_last = (2 if getattr(self.slice, "ae_only", False) else 3,
                           (2 if getattr(self.slice, "ae_only", False) else 3) * global_rows * ld * 4)

    def comm_stats(self):
        return {"collectives": self._coll_last[0], "bytes": self._coll_last[1]}

    def agree_randomness(self, masks, z_real):
        """rng_mode='reference': rank 0's host draws (the 12 dropout masks and z_real of the WHOLE batch) on every rank.
        The hidden stacks are replicated with no gradient exchange, so they stay identical only while every rank applies the
        same masks and the same prior sample; each process draws from its own torch generator, and nothing else makes
        those agree (a per-rank torch.manual_seed(rank) is enough to break it).  One broadcast per step; the device
        generator (rng_mode='device') is keyed by the global row and needs none."""
        if self.world == 1 or not hasattr(self.dist, "broadcast"):
            return masks, z_real
        import numpy as np
        import torch
        # (a model without a prior - AutoEncoder._host_randomness - hands z_real = None: only the masks travel)
        z = None if z_real is None else np.ascontiguousarray(z_real.detach().cpu().numpy(), dtype=np.float32)
        parts = [np.ascontiguousarray(m.cpu().numpy(), dtype=np.uint8).ravel() for m in masks if m is not None]
        if z is not None:
            parts.append(z.view(np.uint8).ravel())
        if not parts:
            return masks, z_real
        flat = np.concatenate(parts)
        flat = broadcast_array(self.dist, self.group, flat, getattr(self.slice, "device", None))
        out, off = [], 0
        for m in masks:
            if m is None:
                out.append(None)
                continue
            n = m.numel()
            out.append(torch.from_numpy(flat[off:off + n].reshape(tuple(m.shape)).copy()))
            off += n
        zr = None if z is None else torch.from_numpy(flat[off:off + z.nbytes].view(np.float32).reshape(z.shape).copy())
        return out, zr

    def sync_conditions(self, conditions):
        """Condition plugins see the whole batch on every rank: their gradients are complete and identical - nothing to sum."""
        return None

    def gather_rows(self, t):
        return t                                    # (every rank already holds all rows)

    def recon_loss(self):
        """Reconstruction loss of the last step over all items: the slices' means weighted by their sizes."""
        import torch
        part = torch.tensor([self.slice.losses()[0] * self.n_slice / float(self.n_items)], dtype=torch.float64)
        nccl = str(self.dist.get_backend(self.group)).lower() == "nccl"
        buf = part.to(self.slice.device) if nccl else part
        if self.world > 1:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(buf.item())

    def gather_output_layer(self):
        """The trained model into the full-vocabulary handle: every rank's rows of dec.lin3 and enc.lin1, and the hidden
        layers (identical on all ranks: copied from this rank's training handle).  Collective."""
        import torch
        from ._hip import T_ENC_W1T, T_ENC_B1, T_ENC_W2, T_ENC_W3, T_DEC_V1, T_DEC_V2, T_DEC_V3, T_DISC_D1, T_DISC_D2, T_DISC_D3
        if self._gathered:
            return
        m, sl, d = self.model, self.slice, self.dist
        sl.sync()
        m.sync()
        rows = -(-self.n_items // self.world)
        for tid in (T_DEC_V3, T_ENC_W1T):
            full, mine = m.tensor(tid, padded=True), sl.tensor(tid, padded=True)
            send = torch.zeros(rows, full.shape[1], dtype=full.dtype, device=full.device)
            send[:mine.shape[0]] = mine
            recv = torch.empty(self.world * rows, full.shape[1], dtype=full.dtype, device=full.device)
            if self.world > 1:
                d.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
            else:
                recv = send
            for r in range(self.world):
                it = item_items(self.n_items, r, self.world, self.interleaved)
                full[it] = recv[r * rows:r * rows + _slice_len(it, self.n_items)]
        for tid in (T_ENC_B1, T_ENC_W2, T_ENC_W3, T_DEC_V1, T_DEC_V2, T_DISC_D1, T_DISC_D2, T_DISC_D3):
            m.tensor(tid, padded=True).copy_(sl.tensor(tid, padded=True))
        m.params_changed()
        self._gather_optimiser_state()
        self._gathered = True

    def _gather_optimiser_state(self):
        """The four optimisers' state into the full-vocabulary handle (enc_optim.state_dict() & co. read it there):
        the moments of the two vocabulary-wide layers travel with their rows, those of the hidden layers are this rank's own
        (identical on every rank), the step counters follow.  sl.sync() has replayed the postponed first-layer updates, so
        the rows carry the counters' step."""
        import ctypes as C
        import torch
        from ._hip import T_ADAM_ENC, T_ADAM_GEN, T_ADAM_DEC, T_ADAM_DISC, O_ENC, O_DEC, O_GEN, O_DISC, _check
        m, sl, d = self.model, self.slice, self.dist
        rows = -(-self.n_items // self.world)
        wide = [T_ADAM_ENC, T_ADAM_ENC + 1, T_ADAM_GEN, T_ADAM_GEN + 1, T_ADAM_DEC + 4, T_ADAM_DEC + 5]
        small = [T_ADAM_ENC + k for k in range(2, 8)] + [T_ADAM_GEN + k for k in range(2, 8)] + \
                [T_ADAM_DEC + k for k in range(4)] + [T_ADAM_DISC + k for k in range(6)]
        for tid in wide:
            full, mine = m.tensor(tid, padded=True), sl.tensor(tid, padded=True)
            send = torch.zeros(rows, full.shape[1], dtype=full.dtype, device=full.device)
            send[:mine.shape[0]] = mine
            recv = torch.empty(self.world * rows, full.shape[1], dtype=full.dtype, device=full.device)
            if self.world > 1:
                d.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
            else:
                recv = send
            for r in range(self.world):
                it = item_items(self.n_items, r, self.world, self.interleaved)
                full[it] = recv[r * rows:r * rows + _slice_len(it, self.n_items)]
        for tid in small:
            m.tensor(tid, padded=True).copy_(sl.tensor(tid, padded=True))
        torch.cuda.synchronize(m.device)
        for oid in (O_ENC, O_DEC, O_GEN, O_DISC):
            step = C.c_int64()
            _check(sl.lib.aae_store_adam(sl.handle, oid, 2, None, None, None, None, C.byref(step)))
            _check(m.lib.aae_load_adam(m.handle, oid, 2, None, None, None, None, step.value))


class RcclTable:
    """Owner of an aae_collectives table over a communicator the library created (aae_rccl_init): close() - also at
    garbage collection - hands it back (aae_rccl_destroy).  `.table` is what aae_dp_step takes."""

    def __init__(self, lib, table):
        self.lib, self.table = lib, table

    def close(self):
        tab, self.table = self.table, None
        if tab is not None and self.lib is not None:
            import ctypes as C
            self.lib.aae_rccl_destroy(C.byref(tab))

    def __del__(self):
        try:
            self.close()
        except Exception:               # noqa: BLE001 - interpreter shutdown: the library may be gone already
            pass


def rccl_collectives(model, dist, group=None):
    """An aae_collectives table (include/aaerec_hip.h) over an RCCL communicator the LIBRARY creates: rank 0 draws the
    ncclUniqueId (aae_rccl_unique_id), torch.distributed carries its 128 bytes to the other ranks, every rank joins
    (aae_rccl_init).  Returns an RcclTable (the communicator lives as long as it does) or None when the ranks agree that
    the library's communicator is not available everywhere.

    Every rank runs the SAME sequence of torch.distributed collectives whatever fails locally (a rank that skipped one
    would leave its peers waiting in it for ever): 1. each rank probes librccl by itself (aae_rccl_unique_id: dlopen +
    ncclGetUniqueId, no communication; rank 0's draw is the one used), 2. all_reduce(MIN) of the probe status, 3. only if
    every probe succeeded: broadcast of the id and the collective ncclCommInitRank, 4. all_reduce(MIN) of the init status
    (a rank whose init failed after its peers' succeeded; the peers drop their communicator again)."""
    import ctypes as C
    import sys
    import torch
    from . import _hip
    lib = model.lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=model.device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    def give_up(stage, err):
        print("aaerec: the library's RCCL communicator is not available on every rank (%s%s); using the phase-by-phase "
              "driver over torch.distributed" % (stage, "" if err is None else ": %s" % (err,)), file=sys.stderr, flush=True)
        return None
    buf, err = C.create_string_buffer(128), None
    try:
        _hip._check(lib.aae_rccl_unique_id(buf))
    except Exception as e:              # noqa: BLE001 - agreed on below, by every rank
        err = e
    if not agree(err is None):
        return give_up("librccl probe", err)
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(model.device)
    if world > 1:
        dist.broadcast(t, src=0, group=group)
    ident = bytes(t.cpu().numpy().tobytes())
    tab = _hip.AaeCollectives()
    try:
        with model._on_device():
            _hip._check(lib.aae_rccl_init(C.create_string_buffer(ident, 128), world, rank, C.byref(tab)))
    except Exception as e:              # noqa: BLE001
        err = e
    owner = RcclTable(lib, tab) if err is None else None
    if not agree(err is None):
        if owner is not None:
            owner.close()
        return give_up("ncclCommInitRank", err)
    return owner


class IpcTable:
    """Owner of an aae_collectives table over peer-mapped mailboxes (aae_ipc_*): close() is COLLECTIVE - a barrier first, so
    that no peer still reads this rank's mailbox when it is freed."""

    def __init__(self, lib, table, mailbox, dist, group):
        self.lib, self.table, self.mailbox, self.dist, self.group = lib, table, mailbox, dist, group

    def close(self):
        tab, self.table = self.table, None
        if tab is None:
            return
        import ctypes as C
        from . import _hip
        import torch
        torch.cuda.synchronize()
        self.dist.barrier(group=self.group)
        _hip._check(self.lib.aae_ipc_destroy(C.byref(tab), self.mailbox))


def ipc_collectives(model, dist, max_floats, group=None):
    """An aae_collectives table whose all_reduce is ONE launch over mailboxes every rank of the node maps (csrc/ipc_collectives.h):
    for the three small all-reduces of dp_mode='shard'.  Every rank allocates a mailbox, torch.distributed carries the 64-byte
    IPC handles to the peers (all_gather of uint8 tensors - any backend), every rank maps the others'.  Returns an IpcTable (call
    close() on every rank when done) or None when the ranks agree that a mailbox could not be shared everywhere."""
    import ctypes as C
    import torch
    from . import _hip
    lib = model.lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nccl = str(dist.get_backend(group)).lower() == "nccl"
    dev = model.device if nccl else torch.device("cpu")

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1
    handle, mailbox, err = C.create_string_buffer(64), C.c_void_p(), None
    try:
        with model._on_device():
            _hip._check(lib.aae_ipc_create(int(max_floats), handle, C.byref(mailbox)))
    except Exception as e:              # noqa: BLE001 - agreed on below, by every rank
        err = e
    if not agree(err is None):
        if err is None:
            lib.aae_ipc_destroy(None, mailbox)
        return None
    mine = torch.frombuffer(bytearray(handle.raw), dtype=torch.uint8).to(dev)
    every = [torch.empty_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(every, mine, group=group)
    else:
        every = [mine]
    blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in every)
    tab = _hip.AaeCollectives()
    try:
        with model._on_device():
            _hip._check(lib.aae_ipc_init(mailbox, C.create_string_buffer(blob, 64 * world), world, rank, int(max_floats), C.byref(tab)))
    except Exception as e:              # noqa: BLE001
        err = e
    if not agree(err is None):
        if err is None:
            dist.barrier(group=group)
            lib.aae_ipc_destroy(C.byref(tab), mailbox)
        else:
            lib.aae_ipc_destroy(None, mailbox)
        return None
    # self-test before anybody depends on it: ONE all-reduce of a known operand through the table.  Its wait for the peers' flags
    # spins on the device for at most 2 s - on a box where a peer's mailbox is mapped but its flag never becomes visible that is
    # 2 s here, once, instead of 2 s per exchange of a training run; every rank must see the rank-ordered sum, in time
    import sys
    import time
    probe = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=model.device)
    with model._on_device():
        torch.cuda.synchronize(model.device)
        t0 = time.perf_counter()
        rc = tab.all_reduce(tab.ctx, C.c_void_p(probe.data_ptr()), 1024, C.c_void_p(torch.cuda.current_stream(model.device).cuda_stream))
        torch.cuda.synchronize(model.device)
        dt = time.perf_counter() - t0
    good = rc == 0 and dt < 0.5 and bool((probe == float(world * (world + 1) // 2)).all().item())
    if not agree(good):
        dist.barrier(group=group)
        lib.aae_ipc_destroy(C.byref(tab), mailbox)          # (its report of the timed-out wait is the reason we are here)
        print("aaerec: the one-shot all-reduce over mapped mailboxes failed its self-test on some rank (%.3f s, rc %d); using the "
              "backend's collectives" % (dt, rc), file=sys.stderr, flush=True)
        return None
    return IpcTable(lib, tab, mailbox, dist, group)


def python_collectives(model, dist, group=None):
    """The same table over any object with torch.distributed's collective interface (HostStagedCollectives, the
    single-process stand-ins of the tools): the library calls back at the step's exchange points; operands are staged
    through host memory (aae_memcpy_sync).  Functional, not fast - the production table is rccl_collectives.
    Returns (table, keep-alive): the callbacks must outlive the table's users."""
    import ctypes as C
    import numpy as np
    import torch
    from . import _hip
    lib = model.lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    raw = getattr(dist, "d", dist)                 # (HostStagedCollectives wraps the gloo module: CPU tensors go to it directly)

    def fetch(ptr, count, stream):
        host = np.empty(int(count), dtype=np.float32)
        _hip._check(lib.aae_memcpy_sync(host.ctypes.data, ptr, host.nbytes, stream))
        return torch.from_numpy(host)

    def put(ptr, t, stream):
        host = np.ascontiguousarray(t.numpy(), dtype=np.float32)
        _hip._check(lib.aae_memcpy_sync(ptr, host.ctypes.data, host.nbytes, stream))

    def guarded(fn):
        def wrapped(*a):
            try:
                fn(*a)
                return 0
            except Exception as e:                 # (an exception must not unwind through the C frames)
                import traceback
                traceback.print_exc()
                return -3
        return wrapped

    def all_gather(ctx, send, recv, count, stream):
        x = fetch(send, count, stream)
        out = torch.empty(world * int(count), dtype=torch.float32)
        raw.all_gather_into_tensor(out, x, group=group)
        put(recv, out, stream)

    def reduce_scatter(ctx, send, recv, count, stream):
        x = fetch(send, world * int(count), stream)
        raw.all_reduce(x, group=group)
        put(recv, x[rank * int(count):(rank + 1) * int(count)], stream)

    def all_reduce(ctx, buf, count, stream):
        x = fetch(buf, count, stream)
        raw.all_reduce(x, group=group)
        put(buf, x, stream)

    cbs = (_hip._COLL_AG(guarded(all_gather)), _hip._COLL_AG(guarded(reduce_scatter)), _hip._COLL_AR(guarded(all_reduce)))
    tab = _hip.AaeCollectives()
    tab.ctx, tab.all_gather, tab.reduce_scatter, tab.all_reduce = None, cbs[0], cbs[1], cbs[2]
    tab.world, tab.rank = world, rank
    return tab, cbs


class HostStagedCollectives:
    """torch.distributed's collective interface (the subset this module uses) over a gloo group for GPU tensors: every
    operand is staged through host memory.  Not a production path - RCCL over xGMI is - but it lets several ranks
    share ONE GPU (RCCL refuses two ranks on a device), which is how the multi-rank logic of fit() and bench.py is
    exercised on a single-GPU test box (tests/test_host_gpu.py, AAE_BENCH_GLOO_ONE_GPU=1)."""

    def __init__(self, dist):
        self.d, self.ReduceOp = dist, dist.ReduceOp

    def get_rank(self, group=None):
        return self.d.get_rank()

    def get_world_size(self, group=None):
        return self.d.get_world_size()

    def get_backend(self, group=None):
        return "gloo"

    def barrier(self):
        self.d.barrier()

    def destroy_process_group(self):
        self.d.destroy_process_group()

    def all_reduce(self, t, op=None, group=None, async_op=False):
        c = t.cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        t.copy_(c)

    def broadcast(self, t, src=0, group=None, async_op=False):
        c = t.cpu()
        self.d.broadcast(c, src)
        t.copy_(c)

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        import torch
        c = torch.empty(out.shape, dtype=out.dtype)
        self.d.all_gather_into_tensor(c, inp.cpu())
        out.copy_(c)

    def reduce_scatter_tensor(self, out, inp, op=None, group=None, async_op=False):
        c = inp.cpu()
        self.d.all_reduce(c)
        n, r = out.numel(), self.d.get_rank()
        out.copy_(c.view(-1)[r * n:(r + 1) * n].view(out.shape))
