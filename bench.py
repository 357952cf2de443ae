#!/usr/bin/env python3
"""Headline benchmark: AAE training docs/sec at |items|=100k, hidden=200 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--items N] [--hidden H]

A "step" is one AdversarialAutoEncoder.partial_fit (ae_step + disc_step + gen_step with all
four optimiser updates, reference aaerec/aae.py:745-766) over one batch of B synthetic docs per
GPU, inputs (the CSR corpus) already resident in HBM.  For N > 1 launch with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU, RCCL); every rank processes its own B docs per step (weak scaling) and the gradients are
summed across ranks before the optimisers run.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel that takes the most time in the
step, from hipEvent pairs recorded around its launches inside the timed region;
`cpu_baseline` is the PyTorch-CPU dense port of the reference step (oracle/dense_torch_port.py)
timed on this box's host cores over a bounded sample of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=100, help="docs per GPU per step (reference default 100, aae.py:599)")
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--hidden", type=int, default=200)
    ap.add_argument("--code", type=int, default=50)
    ap.add_argument("--cond-inc", type=int, default=0, help="width of a constant concatenated condition block (config C4: 300)")
    ap.add_argument("--median-len", type=int, default=20, help="median items per synthetic doc")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-shard", action="store_true", help="data parallel: all-reduce + replicated Adam for the decoder output layer")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel (gradient export) path even on 1 rank")
    ap.add_argument("--dp", choices=("vocab", "replicated"), default="vocab",
                    help="N > 1: 'vocab' shards the decoder's output layer over the vocabulary (ranks exchange hidden "
                         "activations, aaerec.parallel.VocabParallelAAE); 'replicated' keeps a replica of it on every "
                         "rank and exchanges its dense gradient (DataParallelAAE)")
    ap.add_argument("--unfused-decoder", action="store_true", help="A/B: keep the three-kernel decoder path")
    return ap.parse_args()


def kernel_models(N, h, B, nnz_per_batch):
    """ALGORITHMIC bytes / flops per launch of the instrumented kernels."""
    P3 = N * (h + 1)      # decoder output layer, augmented with its bias column
    P1 = N * h            # encoder first layer
    return {
        "enc_gather":  dict(bytes=nnz_per_batch * h * 4, flops=2 * nnz_per_batch * h),
        "dec_bce_fwd": dict(bytes=4 * P3 + 4 * B * N, flops=2 * B * P3),
        "dec_da2":     dict(bytes=4 * P3 + 4 * B * N, flops=2 * B * N * h),
        "dec_dv3_adam": dict(bytes=24 * P3 + 4 * B * N, flops=2 * B * P3),
        # deferred Adam: only the rows of the items in the batch move (read p,g,m,v; write p,m,v,g=0)
        "enc_w1_adam": dict(bytes=32 * nnz_per_batch * h, flops=0),
        # fused decoder output layer: V3a read once (4 B) + m, v read (8 B) + p, m, v written (12 B) per
        # parameter; logits/dL/dlogits never leave the chip; three GEMMs of 2*B*N*(h+1) flop each
        "dec_fused": dict(bytes=24 * P3, flops=6 * B * P3),
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the AAE step has no CPU fallback")
    # debugging aid for a single-GPU box: all ranks on device 0, collectives over gloo staged through the host
    # (aaerec.parallel.HostStagedCollectives) - exercises this file's multi-rank logic; the numbers mean nothing
    one_gpu = os.environ.get("AAE_BENCH_GLOO_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dp = world > 1 or a.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            from aaerec.parallel import HostStagedCollectives
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist = HostStagedCollectives(dist)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from aaerec._hip import HipAAE, DeviceCSR
    from aaerec import _hip
    from tools.synth import init_params, throughput_corpus

    N, h, c, B = a.items, a.hidden, a.code, a.batch
    n_batches = 64
    X = throughput_corpus(n_batches * B, N, median_len=a.median_len, seed=1234 + rank)
    nnz_per_batch = X.nnz / n_batches
    csr = DeviceCSR(X, dev)
    params = init_params(N, h, c, cond_inc=a.cond_inc, seed=0)
    cond_all = None
    if a.cond_inc:
        cond_all = (torch.randn(n_batches * B, a.cond_inc, device=dev) * 0.1)
    # rows of the packed first-layer gradient per exchange: must be the same on every rank
    w1_cap = int(X.getnnz(1).reshape(n_batches, B).sum(1).max()) + 8
    if use_dp and world > 1:
        capt = torch.tensor([w1_cap], dtype=torch.int64, device=dev)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
        w1_cap = int(capt.item())
    # one seed on every rank: the device generator is keyed by the row of the global batch (aae_set_rng_rows), so the
    # ranks together draw what one process would for the whole batch
    model = HipAAE(N, h, c, cond_inc=a.cond_inc, max_batch=B, rng_mode="device", seed=1,
                   grad_mode="export" if use_dp else "fused", device=dev, unfused_decoder=a.unfused_decoder,
                   dp_world=world, w1_cap=w1_cap)
    model.load_params(params)
    vocab = use_dp and a.dp == "vocab"
    slice_model = None
    if vocab:
        import scipy.sparse as sp
        from aaerec.parallel import VocabParallelAAE, item_slice
        lo, hi = item_slice(N, rank, world)
        # every rank walks the same global batches (in fit(): one shared permutation of one corpus; here: the ranks'
        # synthetic corpora regenerated from their seeds): rank r's documents are rows [r*B, (r+1)*B) of global batch i
        Xs = [X if r == rank else throughput_corpus(n_batches * B, N, median_len=a.median_len, seed=1234 + r)
              for r in range(world)]
        Xg = sp.vstack([Xs[r][i * B:(i + 1) * B] for i in range(n_batches) for r in range(world)]).tocsr()
        slice_csr = DeviceCSR(Xg[:, lo:hi], dev)
        del Xs, Xg
        sp_params = dict(params)
        sp_params["dec.lin3.weight"], sp_params["dec.lin3.bias"] = params["dec.lin3.weight"][lo:hi], params["dec.lin3.bias"][lo:hi]
        sp_params["enc.lin1.weight"] = params["enc.lin1.weight"][:, lo:hi]
        slice_model = HipAAE(hi - lo, h, c, cond_inc=a.cond_inc, max_batch=B * world, rng_mode="device", seed=1,
                             device=dev, unfused_decoder=a.unfused_decoder)
        slice_model.load_params(sp_params)
        runner = VocabParallelAAE(model, slice_model, dist, N)
        Bg = B * world
        step = lambda i: runner.step(csr, (i % n_batches) * B, B, slice_csr, (i % n_batches) * Bg, Bg,   # noqa: E731
                                     cond=None if cond_all is None else cond_all[(i % n_batches) * B:(i % n_batches + 1) * B])
    elif use_dp:
        from aaerec.parallel import DataParallelAAE
        runner = DataParallelAAE(model, dist, shard_decoder=False if a.no_shard else ("force" if world == 1 else True))
        step = lambda i: runner.step(csr, (i % n_batches) * B, B, global_rows=B * world)   # noqa: E731
    else:
        if cond_all is not None:
            step = lambda i: model.step(csr, (i % n_batches) * B, B,   # noqa: E731
                                        cond=cond_all[(i % n_batches) * B:(i % n_batches + 1) * B])
        else:
            step = lambda i: model.step(csr, (i % n_batches) * B, B)   # noqa: E731

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        step(i)
    # Live HIP-event timing inside the timed region, on the launch stream (C ABI: aae_profile_*), of the decoder
    # output-layer kernels only - the candidates for the dominant kernel the roofline block reports.  An event
    # pair costs a few microseconds of stream time, so the other kernels (2 gathers + 2 sparse-Adam launches per
    # step) are timed in a short pass AFTER the timed region; that pass does not enter `value`.
    K_GATHER, K_BCE, K_DA2, K_DV3, K_W1, K_FUSED = range(6)
    # (vocabulary-sharded runs: the output-layer kernels run on the slice model, over n_items / world items and the
    # global batch)
    out_model = slice_model if vocab else model
    out_model.profile_enable(True, kernels=(K_BCE, K_DA2, K_DV3, K_FUSED))
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    if use_dp:
        runner.wait_pending()
    barrier()
    dt = time.perf_counter() - t0
    out_model.profile_enable(False)
    losses = model.losses()
    if vocab:
        losses = (runner.recon_loss(),) + tuple(losses[1:])
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    docs_per_s = a.steps * B * world / dt

    names = ["enc_gather", "dec_bce_fwd", "dec_da2", "dec_dv3_adam", "enc_w1_adam", "dec_fused"]
    km = kernel_models(N, h, B, nnz_per_batch)
    if vocab:       # the output-layer kernels' algorithmic work on this rank: its item slice x the global batch
        km_out = kernel_models(slice_model.N, h, B * world, nnz_per_batch)
        for k in ("dec_bce_fwd", "dec_da2", "dec_dv3_adam", "dec_fused"):
            km[k] = km_out[k]
    kstats = {}

    def collect(kids, steps, wall, src=None):
        for kid in kids:
            ms, n = (src or model).profile_read(kid)
            if n:
                avg_s = ms / n * 1e-3
                kstats[names[kid]] = dict(launches_per_step=n / steps, avg_us=round(avg_s * 1e6, 2),
                                          step_share=round(ms * 1e-3 / wall, 4),
                                          GBps=round(km[names[kid]]["bytes"] / avg_s / 1e9, 1),
                                          TFLOPs=round(km[names[kid]]["flops"] / avg_s / 1e12, 2))
    collect((K_BCE, K_DA2, K_DV3, K_FUSED), a.steps, dt, out_model)
    extra = min(a.steps, 40)
    model.profile_enable(True, kernels=(K_GATHER, K_W1))
    for i in range(extra):
        step(a.warmup + a.steps + i)
    if use_dp:
        runner.wait_pending()
    barrier()
    model.profile_enable(False)
    collect((K_GATHER, K_W1), extra, dt * extra / a.steps)
    roofline = None
    if kstats:
        dom = max(kstats, key=lambda k: kstats[k]["step_share"])
        ks = kstats[dom]
        hbm_frac = ks["GBps"] / HBM_PEAK_GBS
        mfma_frac = ks["TFLOPs"] / MFMA_F32_PEAK_TF
        if hbm_frac >= mfma_frac:
            roofline = dict(kernel=dom, bound="hbm", achieved=ks["GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(hbm_frac, 4), traffic=None)
        else:
            roofline = dict(kernel=dom, bound="mfma", achieved=ks["TFLOPs"], peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                            frac=round(mfma_frac, 4), traffic=None)
        roofline["avg_us"] = ks["avg_us"]
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as
        # MI355X_MICROARCH.md prescribes), collected separately and committed under profiles/
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            if pmc["config"] == {"n_items": N, "n_hidden": h, "batch": B} and dom in pmc["kernels"]:
                roofline["traffic"] = pmc["kernels"][dom]["traffic_bytes"]
        except (OSError, ValueError, KeyError):
            pass

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        host_cores = os.cpu_count() or 1
        from oracle.dense_torch_port import DenseTorchAAE      # the checker's dense port, timed here as the CPU baseline only
        ref = DenseTorchAAE(params)
        Xc = X[:B * 8]
        # intra-op thread count: the dense step's ATen ops stop scaling (and collapse) well below the
        # core count of a 100+-core host, so time one step at a few counts and keep the fastest
        cands = [t for t in (8, 16, 32, 64) if t <= host_cores] or [host_cores]
        torch.set_num_threads(cands[0])
        ref.partial_fit(Xc[0:B].toarray())               # warm-up
        best_t, best = cands[0], None
        for t in cands:
            torch.set_num_threads(t)
            t0c = time.perf_counter()
            ref.partial_fit(Xc[B:2 * B].toarray())
            el = time.perf_counter() - t0c
            if best is None or el < best:
                best_t, best = t, el
            elif el > 1.5 * best:
                break
        cores = best_t
        torch.set_num_threads(cores)
        done, t0c = 0, time.perf_counter()
        while True:
            s0 = (done % 7 + 1) * B
            ref.partial_fit(Xc[s0:s0 + B].toarray())     # toarray() is part of the reference's loop (aae.py:823)
            done += 1
            el = time.perf_counter() - t0c
            if (el > a.cpu_seconds and done >= 3) or done >= 200:
                break
        cpu = dict(value=round(done * B / el, 1), unit="docs/s", cores=cores, kind="port",
                   sample=f"{done} partial_fit steps of batch {B} (toarray + dense PyTorch-CPU step), "
                          f"{el:.1f} s, {cores} intra-op threads (fastest of {cands} on a {host_cores}-core host)")

    if rank == 0:
        out = {
            "metric": "train docs/sec at |items|=100k h=200",
            "value": round(docs_per_s, 1), "unit": "docs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3 PubMed-scale synthetic Bags: |items|={N}, hidden={h}, code={c}, fp32, "
                                   f"batch={B} docs/GPU/step, full partial_fit (ae+disc+gen, 4 Adam)",
                       "n_items": N, "n_hidden": h, "n_code": c, "batch_per_gpu": B, "global_batch": B * world,
                       "cond_inc": a.cond_inc, "nnz_per_batch": round(nnz_per_batch, 1), "rng": "device",
                       "parallelism": (f"dp{world}" if not use_dp else
                                       f"dp{world}, decoder output layer sharded over the vocabulary" if vocab else
                                       f"dp{world}, replicated decoder")},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kstats,
            "losses_last_step": [round(x, 5) for x in losses],
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(docs_per_s / cpu["value"], 1)
        result_line = json.dumps(out)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio, which is block-buffered when stdout is a pipe and
        # would otherwise land after the result: flush it first so the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
