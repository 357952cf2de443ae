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
        self._w1_all = None

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

    def _allreduce(self, which, async_op=False):
        return [self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                for t in self.model.grad_buckets(which)]

    def _exchange_encoder(self, which):
        """All encoder gradients of one optimiser step (enc_optim after the ae phases, gen_optim after
        gen_step).  GPU model: one all-gather of packets (packed first-layer rows + the small layers).
        Stand-ins without the packed path: all-reduce of dense buckets."""
        if not hasattr(self.model, "w1_export") or getattr(self.model, "packet_has_small", True) is False:
            self._allreduce(which)
            self.model.apply_updates(which)
        self._exchange_w1(which)

    def _exchange_w1(self, which):
        """First encoder layer: only the rows of the items in the global batch carry gradient, so the
        ranks all-gather their packed rows (a few MB) instead of all-reducing the dense [N, h] tensor."""
        m = self.model
        if not hasattr(m, "w1_export"):
            return                      # dense stand-in: the gradient is part of grad_buckets()
        pk = m.w1_export()
        if self.world == 1:
            allp = pk
        else:
            if self._w1_all is None or self._w1_all.numel() != pk.numel() * self.world:
                self._w1_all = pk.new_empty(pk.numel() * self.world)
            self.dist.all_gather_into_tensor(self._w1_all, pk, group=self.group)
            allp = self._w1_all
        m.w1_import(allp, self.world, which)

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
        if global_rows is None:
            global_rows = n_rows * self.world
        m.set_grad_scale(n_rows / float(global_rows))
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
        m.disc_step()
        self._allreduce(O_DISC)
        m.apply_updates(O_DISC)
        m.gen_step()
        self._exchange_encoder(O_GEN)
        self._dec_finish(dec_state)
