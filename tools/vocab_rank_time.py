#!/usr/bin/env python3
"""Per-rank COMPUTE time of one training step at `world` ranks, measured on one GPU: the collectives are replaced by
local copies of the same shapes (all-gather = this rank's operand repeated, reduce-scatter / all-reduce = identity),
so what is timed is every kernel a rank launches - the replicated scheme (DataParallelAAE: dense dec.lin3 gradient,
optimiser on 1/world of its rows) next to the vocabulary-sharded one (VocabParallelAAE: output layer over
n_items / world items x the global batch).  Communication time is NOT in these numbers; the operand sizes are printed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np
import torch
from aaerec._hip import HipAAE, DeviceCSR
from aaerec.parallel import DataParallelAAE, VocabParallelAAE, ItemShardedAAE, item_slice
from tools.synth import init_params
from tools.synth import throughput_corpus

N, h, c, B = int(os.environ.get("VR_N", 100000)), 200, 50, int(os.environ.get("VR_B", 100))     # B = docs per rank
world = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("VR_WORLD", 8))
NB = int(os.environ.get("VR_BATCHES", 64))
MEDIAN_LEN = int(os.environ.get("VR_MEDIAN_LEN", 20))


class EchoDist:
    class ReduceOp:
        SUM = "sum"

    def __init__(self, world):
        self.world, self.bytes = world, {}

    def get_rank(self, group=None):
        return 0

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return "echo"

    def _count(self, kind, t):
        self.bytes[kind] = self.bytes.get(kind, 0) + t.numel() * 4

    def all_reduce(self, t, op=None, group=None, async_op=False):
        self._count("all_reduce", t)

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        self._count("all_gather(out)", out)
        out.reshape(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))

    def reduce_scatter_tensor(self, out, inp, op=None, group=None, async_op=False):
        self._count("reduce_scatter(in)", inp)
        out.view(-1).copy_(inp.reshape(self.world, -1)[0])


Bg = B * world
X = throughput_corpus(NB * Bg, N, median_len=MEDIAN_LEN, seed=1234)
params = init_params(N, h, c, seed=0)
dev = torch.device("cuda:0")
w1_cap = int(X.getnnz(1).reshape(-1, B).sum(1).max()) + 8       # rows of a first-layer packet: the fullest share


def timeit(step, steps=int(os.environ.get('VR_STEPS', 200)), warm=int(os.environ.get('VR_WARM', 30))):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    t_host = time.perf_counter() - t0           # the loop has ENQUEUED everything
    torch.cuda.synchronize()
    if os.environ.get("VR_HOST"):
        print(f"   [host: {1e3 * t_host / steps:.3f} ms/step to enqueue, {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step to finish]", flush=True)
    return 1e3 * (time.perf_counter() - t0) / steps


csr = DeviceCSR(X, dev)
SCHEMES = os.environ.get("VR_SCHEMES", "replicated,vocab,both,shard").split(",")
# ---- replicated decoder (the current default for N > 1) -------------------------------------------------------
if "replicated" in SCHEMES:
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, grad_mode="export", dp_world=world, w1_cap=w1_cap)
    m.load_params(params)
    d1 = EchoDist(world)
    dp = DataParallelAAE(m, d1, shard_decoder=True if N % world == 0 else False)
    t_dp = timeit(lambda i: dp.step(csr, (i % NB) * Bg, B, global_rows=Bg))
    dp.wait_pending()
    per_step = {k: v / (int(os.environ.get('VR_STEPS', 200)) + int(os.environ.get('VR_WARM', 30))) for k, v in d1.bytes.items()}
    print(f"world {world}: replicated decoder   {t_dp:.3f} ms/step of compute per rank; exchanged per step: "
          + ", ".join(f"{k} {v / 1e6:.2f} MB" for k, v in per_step.items()), flush=True)
    del dp, m
    torch.cuda.empty_cache()
# ---- vocabulary-sharded output layer ---------------------------------------------------------------------------
lo, hi = item_slice(N, 0, world)
if os.environ.get("VR_INTERLEAVE", "1") != "0":     # rank 0 owns items 0, world, 2 world, ... (what fit() uses); 0: the first N / world
    ITEMS = slice(0, N, world)                      # (= the popular head of a frequency-sorted vocabulary: most of every batch's entries)
    lo, hi = 0, len(range(0, N, world))
else:
    ITEMS = slice(lo, hi)
INTER = ITEMS.step is not None and ITEMS.step > 1
sp = dict(params)
sp["dec.lin3.weight"], sp["dec.lin3.bias"] = params["dec.lin3.weight"][ITEMS], params["dec.lin3.bias"][ITEMS]
sp["enc.lin1.weight"] = params["enc.lin1.weight"][:, ITEMS]
BLOCKED = os.environ.get("VR_BLOCKED", "1") != "0"      # 'both' scheme: row-blocked fused launches for the slice (what fit() uses); 0: three GEMMs
BLOCKED_OUT = os.environ.get("VR_BLOCKED_OUT", "0") != "0"   # 'vocab' (output layer alone) scheme: three GEMMs is what fit() uses there
slice_csr = DeviceCSR(X[:, ITEMS].tocsr(), dev)
if "vocab" in SCHEMES:
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, grad_mode="export", dp_world=world, w1_cap=w1_cap)
    m.load_params(params)
    sl = HipAAE(hi - lo, h, c, max_batch=Bg, max_nnz=Bg * 256, blocked_output=BLOCKED_OUT)
    sl.load_params(sp)
    d2 = EchoDist(world)
    vp = VocabParallelAAE(m, sl, d2, N, interleaved=INTER)
    t_vp = timeit(lambda i: vp.step(csr, (i % NB) * Bg, B, slice_csr, (i % NB) * Bg, Bg))
    sl.profile_enable(True)
    for i in range(50):
        vp.step(csr, (i % NB) * Bg, B, slice_csr, (i % NB) * Bg, Bg)
    torch.cuda.synchronize()
    names = ["enc_gather", "dec_bce_fwd", "dec_da2", "dec_dv3_adam", "enc_w1_adam", "dec_fused", "chain", "dec_crit", "dec_opt"]
    parts = []
    for k in range(9):
        ms, n = sl.profile_read(k)
        if n:
            parts.append(f"{names[k]} {n // 50} x {1e3 * ms / n:.1f} us")
    sl.profile_enable(False)
    print(f"world {world}: slice handle ({hi - lo} items x {Bg} rows) output-layer kernels: " + ", ".join(parts), flush=True)
    per_step = {k: v / (int(os.environ.get('VR_STEPS', 200)) + int(os.environ.get('VR_WARM', 30))) for k, v in d2.bytes.items()}
    print(f"world {world}: vocabulary-sharded ({'row-blocked fused' if BLOCKED_OUT else 'three-kernel'} output layer)   {t_vp:.3f} ms/step of compute per rank; exchanged per step: "
          + ", ".join(f"{k} {v / 1e6:.2f} MB" for k, v in per_step.items()), flush=True)
    del vp, m, sl
    torch.cuda.empty_cache()
# ---- both vocabulary-wide matrices with the item slices (enc.lin1 too) ------------------------------------------
if "both" in SCHEMES:
    m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256, grad_mode="export", dp_world=world, w1_cap=w1_cap)
    m.load_params(params)
    sl = HipAAE(hi - lo, h, c, max_batch=Bg, max_nnz=Bg * 256, blocked_output=BLOCKED)
    sl.load_params(sp)
    sl.set_doc_l1(torch.as_tensor(np.asarray(abs(X).sum(1), dtype=np.float32).reshape(-1), device=dev))
    d3 = EchoDist(world)
    vp = VocabParallelAAE(m, sl, d3, N, shard_first_layer=True, interleaved=INTER)
    PF = os.environ.get("VR_PREFETCH", "1") != "0"     # name the slice model's next batch ahead (what fit() does)
    t_vp2 = timeit(lambda i: ((sl.prefetch(slice_csr, ((i + 1) % NB) * Bg, Bg) if PF else None),
                              vp.step(csr, (i % NB) * Bg, B, slice_csr, (i % NB) * Bg, Bg)))
    per_step = {k: v / (int(os.environ.get('VR_STEPS', 200)) + int(os.environ.get('VR_WARM', 30))) for k, v in d3.bytes.items()}
    print(f"world {world}: both vocabulary-wide layers sharded   {t_vp2:.3f} ms/step of compute per rank; exchanged per step: "
          + ", ".join(f"{k} {v / 1e6:.2f} MB" for k, v in per_step.items()), flush=True)
    del vp, m, sl
    torch.cuda.empty_cache()
# ---- item slices + REPLICATED hidden stacks: one handle per rank, the whole global batch through it, 3 all-reduces ----------
if "shard" in SCHEMES:
    sl = HipAAE(hi - lo, h, c, max_batch=Bg, max_nnz=Bg * 256, blocked_output=True)
    sl.load_params(sp)
    sl.set_doc_l1(torch.as_tensor(np.asarray(abs(X).sum(1), dtype=np.float32).reshape(-1), device=dev))
    d4 = EchoDist(world)
    sh = ItemShardedAAE(None, sl, d4, N, interleaved=INTER)
    PF = os.environ.get("VR_PREFETCH", "1") != "0"
    t_sh = timeit(lambda i: ((sl.prefetch(slice_csr, ((i + 1) % NB) * Bg, Bg) if PF else None),
                             sh.step(None, 0, Bg, slice_csr, (i % NB) * Bg, Bg)))
    st = sh.comm_stats()
    print(f"world {world}: item slices + replicated hidden stacks (dp_mode='shard')   {t_sh:.3f} ms/step of compute per rank; "
          f"{st['collectives']} all-reduces of {st['bytes'] / st['collectives'] / 1e6:.2f} MB per step", flush=True)
