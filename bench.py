#!/usr/bin/env python3
"""Headline benchmark: AAE training docs/sec at |items|=100k, hidden=200 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--items N] [--hidden H] [--dtype f32|bf16]

A "step" is one AdversarialAutoEncoder.partial_fit (ae_step + disc_step + gen_step with all four optimiser
updates, reference aaerec/aae.py:745-766) over one batch of B synthetic docs per GPU, driven by the epoch loop of
AdversarialAutoEncoder.fit (reference aae.py:808-831: per-epoch permutation, batches = windows of it) over a corpus
that is resident in HBM - SURVEY section 8d's "docs/s over fit epochs incl. host batch assembly".  `value` is that.
`raw_step` (N = 1) is the same step called directly on contiguous row windows (no fit loop around it).

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), or `python bench.py --gpus N` spawns
them itself (before anything touches the GPU) and relays rank 0's line.  Every rank processes B docs per step (weak
scaling): the global batch of N * B rows is sharded over the ranks by AdversarialAutoEncoder(data_parallel=...).

Rank 0 prints ONE JSON line.  `roofline` is for the kernel that takes the most time in the step, from hipEvent
pairs recorded around its launches inside the timed region; `cpu_baseline` is the PyTorch-CPU dense port of the
reference step (oracle/dense_torch_port.py) timed on this box's host cores over a bounded sample of the same
workload (N = 1 only); `extra.b512` is the MFMA-bound batch-512 variant of the same configuration (N = 1, f32),
`extra.c2_bf16` BASELINE.json's configs[1] (|items| = 47 000, hidden 100, bf16 matrix-core inputs) through the same loop.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA peak
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=100, help="docs per GPU per step (reference default 100, aae.py:599)")
    ap.add_argument("--items", type=int, default=None, help="default: 100000 (config C3), 47000 with --dtype bf16 (config C2)")
    ap.add_argument("--hidden", type=int, default=None, help="default: 200 (C3), 100 with --dtype bf16 (C2)")
    ap.add_argument("--code", type=int, default=50)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="arithmetic of the GEMM-shaped products: f32 (reference precision, config C3) or bf16 MFMA inputs "
                         "with fp32 accumulation, fp32 master weights and fp32 Adam (config C2)")
    ap.add_argument("--cond-inc", type=int, default=0, help="width of a constant concatenated condition block (config C4: 300)")
    ap.add_argument("--median-len", type=int, default=20, help="median items per synthetic doc")
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of K steps, the median is reported (0 = auto: 15 for a region under 20 ms, 5 when "
                                                           "K steps take well under a second, else 1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the batch-512 extra line")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel (gradient export) path even on 1 rank")
    ap.add_argument("--dp", choices=("vocab", "vocab_out", "replicated", "shard"), default="shard",
                    help="N > 1: 'shard' (default) = one handle per rank with its item slice of both vocabulary-wide matrices + "
                         "the hidden layers, the global batch through it, three all-reduces of [global batch, n_hidden] partial "
                         "sums per step (aaerec.parallel.ItemShardedAAE); 'vocab' shards both vocabulary-wide matrices (decoder output layer, encoder first "
                         "layer) over the items, the ranks exchange [global batch, n_hidden] blocks only "
                         "(aaerec.parallel.VocabParallelAAE); 'vocab_out' shards the output layer alone; 'replicated' keeps "
                         "everything on every rank and exchanges dense gradients (DataParallelAAE)")
    ap.add_argument("--unfused-decoder", action="store_true", help="A/B: keep the three-kernel decoder path")
    a = ap.parse_args()
    if a.items is None:
        a.items = 47000 if a.dtype == "bf16" else 100000
    if a.hidden is None:
        a.hidden = 100 if a.dtype == "bf16" else 200
    return a


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one child per GPU (this process has not touched the GPU and
    never will), watch ALL of them, relay rank 0's JSON line.  Children get the torch.distributed.run environment.  The
    library is built once here, before the ranks start (a hipcc subprocess, no GPU call): N ranks racing to write
    libaaerec_hip.so was one way for a rank other than 0 to die at start-up and leave rank 0 in the rendezvous."""
    from aaerec import _build
    _build.build(verbose=False)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout carries the result line; the other ranks' output goes to stderr (their diagnostics stay visible)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
                break
        time.sleep(0.2)
    if failed is not None:
        # a rank died: the others would wait for it in a collective for ever - stop them (our own children, by handle)
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}; stopping the other ranks", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = (buf[0] if buf else b"").decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines and failed is None:
        print(lines[-1], flush=True)
    rc = abs(failed[1]) if failed is not None else max((abs(r) for r in rcs), default=0)
    sys.exit(rc if rc or lines else 1)


def kernel_models(N, h, B, nnz_per_batch, c=50):
    """ALGORITHMIC bytes / flops per launch of the instrumented kernels."""
    P3 = N * (h + 1)      # decoder output layer, augmented with its bias column
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
        # the same layer as two launches (dec_fused.h kDecCrit / kDecOpt): the critical one reads V3a once and does
        # GEMM1 + GEMM3 (logits, dL/d(hidden)), leaving dL/dlogits behind as [B, N] floats; the deferred one (side stream,
        # half the CUs, concurrent with the rest of the step) reads those, does GEMM2 and streams the optimiser state
        "dec_crit": dict(bytes=4 * P3 + 4 * B * N, flops=4 * B * P3),
        "dec_opt": dict(bytes=24 * P3 + 4 * B * N, flops=2 * B * P3),
        # a layer-chain program (5 per step: the hidden stacks of ae forward, ae backward, the discriminator's encoder
        # pass, the discriminator step, gen_step).  Per step: 13 passes of B rows through an h x (h+1) layer and 11 through
        # an (h or c)-wide x (c or h)+1 one, forward and dX; the weights are the algorithmic bytes (each program reads its
        # matrices once).  Latency-, not roofline-bound: the figure is reported, not a target.
        "chain": dict(bytes=4 * (13 * h * (h + 1) + 11 * c * (h + 1)) / 5.0, flops=2.0 * B * (13 * h * (h + 1) + 11 * c * (h + 1)) / 5.0),
    }


def step_floor(Nr, h, c, cond_inc, B_out, B, nnz_per_batch, peak_tf, ms_per_step):
    """One partial_fit against its own floor: ALGORITHMIC bytes (the output layer's parameter + optimiser stream, 24 B per
    element of dec.lin3; the first layer's touched rows: 3 forward gathers + 2 optimisers of 32 B per element; the hidden
    layers' Adam, 28 B per element, the encoder's twice) and flops, against max(bytes / HBM peak, flops / matrix peak of the
    line's dtype).  Nr items x B_out rows in the output layer (a rank's item slice x the global batch when sharded)."""
    small = 2 * (h * (h + 1) + c * (h + 1)) + (h * (c + cond_inc + 1) + h * (h + 1)) + (h * (c + 1) + h * (h + 1) + h + 1)
    step_bytes = 24.0 * Nr * (h + 1) + nnz_per_batch * h * (3 * 4 + 2 * 32) + 28.0 * small
    step_flops = 6.0 * B_out * Nr * (h + 1) + 2.0 * B * (13 * h * (h + 1) + 11 * c * (h + 1)) + 10.0 * nnz_per_batch * h
    floor_ms = max(step_bytes / (HBM_PEAK_GBS * 1e9), step_flops / (peak_tf * 1e12)) * 1e3
    return dict(bytes=round(step_bytes), flops=round(step_flops), floor_ms=round(floor_ms, 4),
                bound="hbm" if step_bytes / (HBM_PEAK_GBS * 1e9) >= step_flops / (peak_tf * 1e12) else "mfma",
                ms_per_step=round(ms_per_step, 4), frac=round(floor_ms / ms_per_step, 4))


NAMES = ["enc_gather", "dec_bce_fwd", "dec_da2", "dec_dv3_adam", "enc_w1_adam", "dec_fused", "chain", "dec_crit", "dec_opt", "rank", "collective"]
K_GATHER, K_BCE, K_DA2, K_DV3, K_W1, K_FUSED, K_CHAIN, K_CRIT, K_OPT, K_RANK, K_COLL = range(11)
K_OUT = (K_BCE, K_DA2, K_DV3, K_FUSED, K_CRIT, K_OPT)       # the decoder output layer's kernels, whichever path runs


class _ConstVectors:
    """Stand-in for the fitted TF-IDF x word2vec vectoriser behind PretrainedWordEmbeddingCondition (config C4): the
    document vectors are precomputed once per dataset (condition.py:345-369), here they are synthetic N(0, 0.1)."""

    def __init__(self, dim):
        self.embedding = np.zeros((1, dim), dtype=np.float32)

    def fit(self, x):
        return self

    def transform(self, x):
        return x


def make_model(a, B_global, dist, dtype=None, conditions=None):
    from aaerec.aae import AdversarialAutoEncoder
    kw = {}
    if (dtype or a.dtype) != "f32":
        kw["dtype"] = dtype or a.dtype
    m = AdversarialAutoEncoder(n_hidden=a.hidden, n_code=a.code, batch_size=B_global, n_epochs=1 << 30, verbose=False,
                               rng_mode="device", seed=1, data_parallel=dist, dp_mode=a.dp, conditions=conditions, **kw)
    if a.unfused_decoder:
        m._unfused_decoder = True
    return m


def timed_steps(it, k, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        next(it)
    barrier()
    return time.perf_counter() - t0


def collective_breakdown(model, it, extra_steps, barrier, dist, world, dev):
    """N > 1: what the step spends in its collectives (event pairs on the step's stream around every call of the collectives
    table, a pass of its own behind the timed region): per rank the time from "this rank reaches the collective" to "it has
    the result" - the transfer AND the wait for the slowest rank.  The rank that waits least is the one the others wait for:
    min over ranks ~ the collective itself, max - min ~ the ranks' skew; step time minus a rank's collective time = its compute."""
    handles = [hh for hh in (model.hip, model._slice) if hh is not None]
    for hh in handles:
        hh.profile_enable(True, kernels=(K_COLL,))
    t0 = time.perf_counter()
    for _ in range(extra_steps):
        next(it)
    barrier()
    wall_c = time.perf_counter() - t0
    ms_c = n_c = 0
    for hh in handles:
        hh.profile_enable(False)
        ms, n = hh.profile_read(K_COLL)
        ms_c, n_c = ms_c + ms, n_c + n
    if not n_c:
        return None
    per_step = ms_c / extra_steps
    lo = hi = per_step
    if dist is not None and world > 1:
        t = torch.tensor([per_step, -per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo = float(t[0].item()), -float(t[1].item())
    step_ms = wall_c / extra_steps * 1e3
    return dict(collectives_per_step=round(n_c / extra_steps, 2), collective_us_avg_rank0=round(ms_c / n_c * 1e3, 2),
                collective_ms_per_step_rank0=round(per_step, 4), collective_ms_per_step_min_over_ranks=round(lo, 4),
                collective_ms_per_step_max_over_ranks=round(hi, 4), ms_per_step_this_pass=round(step_ms, 4),
                compute_ms_per_step_slowest_rank=round(step_ms - lo, 4),
                note="event pairs around the collectives table's calls, a pass of its own (not the timed region); a rank's "
                     "collective time includes its wait for the slowest rank")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)          # does not return
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
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

    from aaerec import _hip
    from aaerec.condition import ConditionList, PretrainedWordEmbeddingCondition
    from tools.synth import init_params, throughput_corpus

    N, h, c, B = a.items, a.hidden, a.code, a.batch
    Bg = B * world
    n_batches = 64
    # one corpus, the same on every rank (as in fit(): every rank holds the corpus and walks rank 0's permutation)
    X = throughput_corpus(n_batches * Bg, N, median_len=a.median_len, seed=1234)
    nnz_per_batch = X.nnz / n_batches / world            # per rank and step
    conditions = cond_data = None
    if a.cond_inc:
        conditions = ConditionList([("title", PretrainedWordEmbeddingCondition(_ConstVectors(a.cond_inc), use_cuda=True))])
        cond_data = [torch.randn(n_batches * Bg, a.cond_inc, device=dev) * 0.1]
    model = make_model(a, Bg, dist if use_dp else None, conditions=conditions)
    with contextlib.redirect_stdout(sys.stderr):
        it = model.fit_steps(X, condition_data=cond_data)
        next(it)                                          # construction + corpus upload + first step
    vocab = model._slice is not None
    out_model = model._slice if vocab else model.hip     # the handle that runs the decoder's output layer

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        next(it)
    # Live HIP-event timing inside the timed region, on the launch stream (C ABI: aae_profile_*), of the decoder
    # output-layer kernels only - the candidates for the dominant kernel the roofline block reports.  An event
    # pair costs a few microseconds of stream time, so the other kernels (2 gathers + 2 sparse-Adam launches per
    # step) are timed in a short pass AFTER the timed region; that pass does not enter `value`.
    # The event pairs ride in the FIRST timed repeat of K steps only (`profiled_repeat`): they cost that repeat ~4 % (a
    # timing event carries a system-scope release the plain launch does not), the other repeats run as production does and
    # the median over all of them is `value`.
    if os.environ.get("AAE_BENCH_NO_PROF") is None:          # (debugging aid: a kernel trace of the loop without the timing events)
        out_model.profile_enable(True, kernels=K_OUT)
    dts = [timed_steps(it, a.steps, barrier)]
    out_model.profile_enable(False)
    # (a region of 20 steps - the driver's K - is 5 ms: the first three such regions behind the warm-up run 3-4 % slower than the
    #  ones after them (the chip settles its clocks over the first ~15 ms of the loop; region_profile.py), so a median of five sat
    #  on the settling ones.  Fifteen repeats of so short a region cost 70 ms and every one is listed in config.repeat_ms_per_step)
    repeats = a.repeats or (15 if dts[0] < 0.02 else 5 if dts[0] < 0.25 else 1)
    for _ in range(repeats - 1):
        dts.append(timed_steps(it, a.steps, barrier))
    dt = float(np.median(dts))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = model._losses()
    docs_per_s = a.steps * Bg / dt

    km = kernel_models(N, h, B, nnz_per_batch, c)
    if vocab:       # the output-layer kernels' algorithmic work on this rank: its item slice x the global batch
        km_out = kernel_models(out_model.N, h, Bg, nnz_per_batch, c)
        for k in ("dec_bce_fwd", "dec_da2", "dec_dv3_adam", "dec_fused", "dec_crit", "dec_opt"):
            km[k] = km_out[k]
    kstats = {}

    def collect(kids, steps, wall, src, models=km):
        for kid in kids:
            ms, n = src.profile_read(kid)
            if n:
                avg_s = ms / n * 1e-3
                kstats[NAMES[kid]] = dict(launches_per_step=round(n / steps, 3), avg_us=round(avg_s * 1e6, 2),
                                          step_share=round(ms * 1e-3 / wall, 4),
                                          GBps=round(models[NAMES[kid]]["bytes"] / avg_s / 1e9, 1),
                                          TFLOPs=round(models[NAMES[kid]]["flops"] / avg_s / 1e12, 2))
    collect(K_OUT, a.steps, dts[0], out_model)
    extra_steps = min(a.steps, 40)
    model.hip.profile_enable(True, kernels=(K_GATHER, K_W1, K_CHAIN))
    t0 = time.perf_counter()
    for _ in range(extra_steps):
        next(it)
    barrier()
    model.hip.profile_enable(False)
    collect((K_GATHER, K_W1, K_CHAIN), extra_steps, time.perf_counter() - t0, model.hip)

    dp_breakdown = collective_breakdown(model, it, extra_steps, barrier, dist, world, dev) if use_dp else None

    peak_tf = MFMA_BF16_PEAK_TF if a.dtype == "bf16" else MFMA_F32_PEAK_TF
    roofline = roofline_critical = None

    def roof(name):
        ks = kstats[name]
        hbm_frac = ks["GBps"] / HBM_PEAK_GBS
        mfma_frac = ks["TFLOPs"] / peak_tf
        if hbm_frac >= mfma_frac:
            r = dict(kernel=name, bound="hbm", achieved=ks["GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_frac, 4),
                     traffic=None)
        else:
            r = dict(kernel=name, bound="mfma", achieved=ks["TFLOPs"], peak=peak_tf, unit="TFLOP/s", frac=round(mfma_frac, 4),
                     traffic=None)
        r["avg_us"] = ks["avg_us"]
        if name == "dec_crit" and a.dtype == "f32" and r["bound"] == "mfma":
            # (r3: the launch multiplies on the bf16 matrix cores, six bf16 products per fp32 product - csrc/dec_crit_x3.h.  r6,
            #  `peak` / `frac` are the ceiling of THAT arithmetic - the pipe the kernel runs on; the fp32 pipe's figure,
            #  which it no longer uses, stays beside them)
            r["peak_fp32_mfma"], r["frac_fp32_mfma"] = r["peak"], r["frac"]
            r["peak"] = round(MFMA_BF16_PEAK_TF / 6.0, 1)
            r["frac"] = round(ks["TFLOPs"] / (MFMA_BF16_PEAK_TF / 6.0), 4)
            r["peak_note"] = "dense bf16 MFMA peak / 6: an fp32 product is six bf16 matrix instructions (three-term split)"
        return r
    if kstats:
        # the dominant kernel = the one with the largest share of launch-to-completion time.  In the split form of the
        # output layer that is the deferred optimiser launch, which runs on half the CUs BESIDE the rest of the step (its
        # duration is not on the step's critical path); the launch the step waits for is reported as roofline_critical
        dom = max(kstats, key=lambda k: kstats[k]["step_share"])
        roofline = roof(dom)
        if dom == "dec_opt" and "dec_crit" in kstats:
            roofline_critical = roof("dec_crit")
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as
        # MI355X_MICROARCH.md prescribes) of this same command, collected in separate runs and committed under
        # profiles/ (a counter pass cannot share a run with the timed region); null when no pass matches the config
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if not name.endswith("_pmc_traffic.json"):
                continue
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                want = {"n_items": N, "n_hidden": h, "batch": B}
                if a.dtype != "f32":
                    want["dtype"] = a.dtype
                if pmc["config"] == want and world == 1:
                    for r in (roofline, roofline_critical):
                        if r is not None and r["traffic"] is None and r["kernel"] in pmc["kernels"]:
                            r["traffic"] = pmc["kernels"][r["kernel"]]["traffic_bytes"]
                            r["traffic_source"] = "profiles/" + name
                            r["traffic_in_run"] = False      # (a counter pass cannot share a run with the timed region)
            except (OSError, ValueError, KeyError):
                pass

    # The whole step against its own floor: ALGORITHMIC bytes and flops of one partial_fit - the output layer's
    # parameter + optimiser stream (24 B per element of dec.lin3, read once / written once), the first layer's touched rows
    # (3 forward gathers + 2 optimisers of 32 B per element), the hidden layers' Adam (28 B per element, the encoder's twice) -
    # against max(bytes / HBM peak, flops / matrix peak of the line's dtype); per rank: its documents, its share of the layer
    Nr = out_model.N if vocab else N
    step_roofline = step_floor(Nr, h, c, a.cond_inc, (Bg if vocab else B), B, nnz_per_batch, peak_tf, dt / a.steps * 1e3)

    raw = None
    extra = {}
    if world == 1 and not use_dp:
        # the same step called directly on contiguous row windows of the resident corpus (r1's headline figure)
        csr = model._fit_csr
        hip = model.hip
        cond_all = cond_data[0] if cond_data else None

        def raw_step(i):
            s0 = (i % n_batches) * B
            hip.step(csr, s0, B, cond=None if cond_all is None else cond_all[s0:s0 + B])
        for i in range(5):
            raw_step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            raw_step(i)
        barrier()
        rdt = time.perf_counter() - t0
        raw = dict(docs_per_s=round(a.steps * B / rdt, 1), ms_per_step=round(rdt / a.steps * 1e3, 4))
        want_extra = set(os.environ.get("AAE_BENCH_EXTRAS", "b512,c2_bf16,c4,c5_world1,predict_topk").split(","))   # (debugging aid)
        if not a.no_extra and a.dtype == "f32" and B != 512 and not a.cond_inc and "b512" in want_extra:
            # SURVEY 8d asks for the MFMA-bound batch (512) next to the reference's default batch
            a2 = argparse.Namespace(**vars(a))
            m2 = make_model(a2, 512, None)
            X2 = throughput_corpus(16 * 512, N, median_len=a.median_len, seed=4321)
            with contextlib.redirect_stdout(sys.stderr):
                it2 = m2.fit_steps(X2)
                next(it2)
            for _ in range(5):
                next(it2)
            k2 = max(10, min(a.steps, 50))
            for _ in range(10):
                next(it2)
            d2 = timed_steps(it2, k2, barrier)            # (timed without the per-kernel events, as the headline loop is)
            m2.hip.profile_enable(True, kernels=K_OUT)
            timed_steps(it2, 10, barrier)
            m2.hip.profile_enable(False)
            km2 = kernel_models(N, h, 512, X2.nnz / 16, c)
            ks2 = {}
            for kid in K_OUT:
                ms, n = m2.hip.profile_read(kid)
                if n:
                    avg_s = ms / n * 1e-3
                    ks2[NAMES[kid]] = dict(avg_us=round(avg_s * 1e6, 2), TFLOPs=round(km2[NAMES[kid]]["flops"] / avg_s / 1e12, 2),
                                           frac_mfma=round(km2[NAMES[kid]]["flops"] / avg_s / 1e12 / MFMA_F32_PEAK_TF, 4))
            extra["b512"] = dict(docs_per_s=round(k2 * 512 / d2, 1), ms_per_step=round(d2 / k2 * 1e3, 4), steps=k2,
                                 kernels=ks2)
            m2.hip.close()      # (destroy the handle - and its side stream - now: the model object sits in a reference cycle, and an idle
            del m2, it2       #  handle's low-priority stream keeps a hardware queue the next model's deferred launches would otherwise get)
        if not a.no_extra and a.dtype == "f32" and (N, h) == (100000, 200) and not a.cond_inc and "c2_bf16" in want_extra:
            # BASELINE.json configs[1] (C2, RCV1-scale: |items| = 47 000, hidden 100, bf16 matrix-core inputs) next to the
            # headline config, so that the driver's line carries it: the same fit() loop at the reference's batch 100
            a3 = argparse.Namespace(**vars(a))
            a3.hidden, a3.items = 100, 47000
            m3 = make_model(a3, B, None, dtype="bf16")
            X3 = throughput_corpus(n_batches * B, a3.items, median_len=a.median_len, seed=2345)
            with contextlib.redirect_stdout(sys.stderr):
                it3 = m3.fit_steps(X3)
                next(it3)
            for _ in range(a.warmup):
                next(it3)
            m3.hip.profile_enable(True, kernels=K_OUT)
            d3 = [timed_steps(it3, a.steps, barrier)]
            m3.hip.profile_enable(False)
            for _ in range(2):
                d3.append(timed_steps(it3, a.steps, barrier))
            km3 = kernel_models(a3.items, a3.hidden, B, X3.nnz / n_batches, c)
            ks3 = {}
            for kid in K_OUT:
                ms, n = m3.hip.profile_read(kid)
                if n:
                    avg_s = ms / n * 1e-3
                    ks3[NAMES[kid]] = dict(avg_us=round(avg_s * 1e6, 2), GBps=round(km3[NAMES[kid]]["bytes"] / avg_s / 1e9, 1),
                                           frac_hbm=round(km3[NAMES[kid]]["bytes"] / avg_s / 1e9 / HBM_PEAK_GBS, 4))
            dm = float(np.median(d3))
            extra["c2_bf16"] = dict(workload=f"C2 RCV1-scale synthetic Bags: |items|={a3.items}, hidden={a3.hidden}, code={c}, bf16 MFMA "
                                             f"inputs / fp32 accumulate, master weights and Adam, batch={B}, through fit()",
                                    docs_per_s=round(a.steps * B / dm, 1), ms_per_step=round(dm / a.steps * 1e3, 4), steps=a.steps,
                                    dtype="bf16", kernels=ks3)
            m3.hip.close()      # (destroy the handle - and its side stream - now: the model object sits in a reference cycle, and an idle
            del m3, it3       #  handle's low-priority stream keeps a hardware queue the next model's deferred launches would otherwise get)
        if not a.no_extra and a.dtype == "f32" and (N, h) == (100000, 200) and not a.cond_inc and "c4" in want_extra:
            # BASELINE.json configs[3] (C4, EconBiz-scale + title condition: |items| = 4 587 - nmi.txt:38 of the reference -,
            # hidden 200, a 300-d constant document vector concatenated to the code - condition.py:312-316, 345-369 -, batch
            # 1000 - eval/econis.py:45): the same fit() loop; tests/test_fullsize_gpu.py::test_c4_bench_shape_fit_path_matches_oracle
            # holds exactly this combination of launches against the oracle
            a4 = argparse.Namespace(**vars(a))
            a4.hidden, a4.items, a4.cond_inc = 200, 4587, 300
            B4, nb4 = 1000, 16
            cl4 = ConditionList([("title", PretrainedWordEmbeddingCondition(_ConstVectors(a4.cond_inc), use_cuda=True))])
            cd4 = [torch.randn(nb4 * B4, a4.cond_inc, device=dev) * 0.1]
            m4 = make_model(a4, B4, None, conditions=cl4)
            X4 = throughput_corpus(nb4 * B4, a4.items, median_len=a.median_len, seed=3456)
            with contextlib.redirect_stdout(sys.stderr):
                it4 = m4.fit_steps(X4, condition_data=cd4)
                next(it4)
            for _ in range(max(5, a.warmup)):
                next(it4)
            k4 = max(10, min(a.steps, 100))
            d4 = [timed_steps(it4, k4, barrier) for _ in range(3)]
            dm = float(np.median(d4))
            extra["c4"] = dict(workload=f"C4 EconBiz-scale synthetic Bags + 300-d title condition (constant concatenated block): "
                                        f"|items|={a4.items}, hidden={a4.hidden}, code={c}, fp32, batch={B4}, through fit()",
                               docs_per_s=round(k4 * B4 / dm, 1), ms_per_step=round(dm / k4 * 1e3, 4), steps=k4, dtype="f32",
                               cond_inc=a4.cond_inc, repeat_ms_per_step=[round(d / k4 * 1e3, 4) for d in d4])
            m4.hip.close()      # (destroy the handle - and its side stream - now: the model object sits in a reference cycle, and an idle
            del m4, it4       #  handle's low-priority stream keeps a hardware queue the next model's deferred launches would otherwise get)
        if not a.no_extra and a.dtype == "f32" and (N, h) == (100000, 200) and not a.cond_inc and "c5_world1" in want_extra:
            # BASELINE.json configs[4] (C5, MPD-scale: |items| = 2 200 000 tracks, hidden 200, batch 512, playlists of median
            # length 60) WHOLE on one MI355X (19 GB of its 288): the N = 1 anchor of the 8-GPU configuration, through the same
            # fit() loop; tests/test_fullsize_gpu.py::test_c5_whole_vocabulary_on_one_gpu_matches_the_chunked_stand_in holds
            # this shape against the oracle's stand-in
            a5 = argparse.Namespace(**vars(a))
            a5.hidden, a5.items = 200, 2200000
            B5, nb5 = 512, 8
            m5 = make_model(a5, B5, None)
            X5 = throughput_corpus(nb5 * B5, a5.items, median_len=60, seed=5678)
            with contextlib.redirect_stdout(sys.stderr):
                it5 = m5.fit_steps(X5)
                next(it5)
            for _ in range(3):
                next(it5)
            k5 = 10
            d5 = [timed_steps(it5, k5, barrier) for _ in range(2)]
            m5.hip.profile_enable(True, kernels=K_OUT)
            timed_steps(it5, 5, barrier)
            m5.hip.profile_enable(False)
            km5 = kernel_models(a5.items, a5.hidden, B5, X5.nnz / nb5, c)
            ks5 = {}
            for kid in K_OUT:
                ms, n = m5.hip.profile_read(kid)
                if n:
                    avg_s = ms / n * 1e-3
                    ks5[NAMES[kid]] = dict(avg_us=round(avg_s * 1e6, 2), TFLOPs=round(km5[NAMES[kid]]["flops"] / avg_s / 1e12, 2),
                                           GBps=round(km5[NAMES[kid]]["bytes"] / avg_s / 1e9, 1))
            dm = float(np.median(d5))
            extra["c5_world1"] = dict(workload=f"C5 MPD-scale synthetic Bags WHOLE on one GPU: |items|={a5.items}, hidden={a5.hidden}, code={c}, "
                                               f"fp32, batch={B5}, median document length 60, through fit()",
                                      docs_per_s=round(k5 * B5 / dm, 1), ms_per_step=round(dm / k5 * 1e3, 4), steps=k5, dtype="f32",
                                      step_roofline=step_floor(a5.items, a5.hidden, c, 0, B5, B5, X5.nnz / nb5, MFMA_F32_PEAK_TF, dm / k5 * 1e3),
                                      kernels=ks5)
            m5.hip.close()
            del m5, it5, X5
        if not a.no_extra and not a.cond_inc and "predict_topk" in want_extra:
            # SURVEY 8f rank 1 / VERDICT r3: predict -> remove_non_missing -> top-k on the device (reference aae.py:840-870,
            # evaluation.py:183-199, 20-58) with the headline model, documents of the resident corpus, 512 rows per library
            # call (aae_predict_topk: the fused rank kernels of csrc/rank_x3.h - no [rows, items] matrix in HBM); k = 10.
            # docs/s = rows ranked / wall time of back-to-back calls; the rank kernel itself from HIP events on the launch stream
            rows_pc = min(512, hip.rank_max_rows(10), X.shape[0])
            calls = max(8, min(40, (X.shape[0] - rows_pc) // rows_pc))
            for i in range(3):
                hip.predict_topk(csr, i * rows_pc, rows_pc, 10)
            barrier()
            t0 = time.perf_counter()
            for i in range(calls):
                hip.predict_topk(csr, (i * rows_pc) % (X.shape[0] - rows_pc + 1), rows_pc, 10)
            barrier()
            dpt = (time.perf_counter() - t0) / calls
            hip.profile_enable(True, kernels=(K_RANK,))
            for i in range(8):
                hip.predict_topk(csr, i * rows_pc, rows_pc, 10)
            barrier()
            hip.profile_enable(False)
            ms_r, n_r = hip.profile_read(K_RANK)
            pt = dict(workload=f"predict -> known-item mask -> top-10 on the device, |items|={N}, hidden={h}, {rows_pc} documents per call",
                      docs_per_s=round(rows_pc / dpt, 1), ms_per_call=round(dpt * 1e3, 4), us_per_100_docs=round(dpt / rows_pc * 1e8, 2),
                      rows_per_call=rows_pc, k=10, fused=bool(hip.rank_max_rows(10) > B))
            if n_r:
                us = ms_r / n_r * 1e3
                fl, by = 2.0 * rows_pc * N * (h + 1), 4.0 * N * (h + 1)
                tf = fl / us * 1e-6
                emu = MFMA_BF16_PEAK_TF / 6.0     # (f32: six bf16 matrix instructions per product, as roofline_critical)
                pk = emu if a.dtype == "f32" else peak_tf
                pt["roofline"] = dict(kernel="rank_x3", bound="mfma", achieved=round(tf, 2), peak=round(pk, 1), unit="TFLOP/s",
                                      frac=round(tf / pk, 4), avg_us=round(us, 2), traffic=None,
                                      algorithmic_flops=fl, algorithmic_bytes=by, GBps=round(by / us * 1e-3, 1),
                                      **({"peak_fp32_mfma": peak_tf, "frac_fp32_mfma": round(tf / peak_tf, 4)} if a.dtype == "f32" else {}))
            extra["predict_topk"] = pt

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        host_cores = os.cpu_count() or 1
        from oracle.dense_torch_port import DenseTorchAAE      # the checker's dense port, timed here as the CPU baseline only
        ref = DenseTorchAAE(init_params(N, h, c, cond_inc=0, seed=0))
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
                   sample=f"{done} partial_fit steps of batch {B} (toarray + dense fp32 PyTorch-CPU step, no conditions), "
                          f"{el:.1f} s, {cores} intra-op threads (fastest of {cands} on a {host_cores}-core host)")

    # N > 1 (r6): the OTHER data-parallel scheme in the same run - the default line is `shard` (item slices, three
    # all-reduces of partial sums), the north star's wording is `replicated` (full replicas, RCCL reduce-scatter / all-gather of
    # the decoder gradient overlapped with backward): the first multi-GPU run measures both.  Same steps, same bracketing
    # (barrier + synchronize, MAX over ranks), a full dp_breakdown; AAE_BENCH_EXTRAS without "dp_other" skips it.
    dp_desc = None
    if use_dp:          # (what the output says about the MAIN line's scheme: read before its model goes)
        dp_desc = dict(shard_first=bool(getattr(model._dp, "shard_first", False)),
                       comm=model._dp.comm_stats() if hasattr(model._dp, "comm_stats") else None,
                       native=getattr(model._dp, "_native", None) is not None)
    dp_other = None
    if use_dp and world > 1 and not a.no_extra and "dp_other" in os.environ.get("AAE_BENCH_EXTRAS", "dp_other").split(","):
        try:
            other = "replicated" if a.dp != "replicated" else "shard"
            del it
            for hh in (model.hip, model._slice):
                if hh is not None:
                    hh.close()
            del model
            torch.cuda.empty_cache()
            a2 = argparse.Namespace(**vars(a))
            a2.dp = other
            model2 = make_model(a2, Bg, dist, conditions=conditions)
            with contextlib.redirect_stdout(sys.stderr):
                it2 = model2.fit_steps(X, condition_data=cond_data)
                next(it2)
            for _ in range(a.warmup):
                next(it2)
            dts2 = [timed_steps(it2, a.steps, barrier) for _ in range(min(repeats, 3))]
            dt2 = float(np.median(dts2))
            t = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
            dp_other = dict(dp=other, value=round(a.steps * Bg / dt2, 1), unit="docs/s", ms_per_step=round(dt2 / a.steps * 1e3, 4),
                            timed_repeats=len(dts2), dp_breakdown=collective_breakdown(model2, it2, min(a.steps, 40), barrier, dist, world, dev),
                            **({"collectives_per_step": model2._dp.comm_stats()} if hasattr(model2._dp, "comm_stats") else {}))
            # ... and `shard` with its three all-reduces as one-shot launches over peer-mapped mailboxes (aae_ipc_*) in
            # place of the backend's ring: taken only when the table passed its self-test on every rank (parallel.ipc_collectives)
            # (opt-in: AAE_BENCH_EXTRAS=dp_other,dp_ipc - this path has run between two processes of ONE GPU only, and a rank that
            #  faults in it would take the whole line with it: the default N > 1 run keeps to the two schemes that RCCL carries)
            if (a.dp == "shard" or other == "shard") and "dp_ipc" in os.environ.get("AAE_BENCH_EXTRAS", "dp_other").split(","):
                del it2
                for hh in (model2.hip, model2._slice):
                    if hh is not None:
                        hh.close()
                del model2
                torch.cuda.empty_cache()
                from aaerec.parallel import IpcTable
                a3 = argparse.Namespace(**vars(a))
                a3.dp = "shard"
                model3 = make_model(a3, Bg, dist, conditions=conditions)
                model3.dp_collectives = "ipc"
                with contextlib.redirect_stdout(sys.stderr):
                    it3 = model3.fit_steps(X, condition_data=cond_data)
                    next(it3)
                keep = getattr(model3._dp, "_native_keep", None)
                if isinstance(keep, IpcTable):
                    for _ in range(a.warmup):
                        next(it3)
                    dts3 = [timed_steps(it3, a.steps, barrier) for _ in range(min(repeats, 3))]
                    t = torch.tensor([float(np.median(dts3))], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt3 = float(t.item())
                    dp_other["shard_over_ipc_all_reduce"] = dict(
                        value=round(a.steps * Bg / dt3, 1), unit="docs/s", ms_per_step=round(dt3 / a.steps * 1e3, 4), timed_repeats=len(dts3),
                        dp_breakdown=collective_breakdown(model3, it3, min(a.steps, 40), barrier, dist, world, dev))
                    del it3
                    torch.cuda.synchronize(dev)
                    keep.close()
                else:
                    dp_other["shard_over_ipc_all_reduce"] = None       # (a mailbox could not be shared / the self-test failed on some rank)
        except Exception as e:       # noqa: BLE001 - the main line is measured already: it is printed whatever happens to this one
            dp_other = dict(dp_other or {}, error=f"{type(e).__name__}: {e}")

    if rank == 0:
        cfg_name = ("C2 RCV1-scale" if a.dtype == "bf16" else "C4 EconBiz-scale + 300-d title condition" if a.cond_inc
                    else "C3 PubMed-scale")
        cfg_id = ("C3" if (N, h, a.dtype, a.cond_inc) == (100000, 200, "f32", 0) else
                  "C2" if (N, h, a.dtype, a.cond_inc) == (47000, 100, "bf16", 0) else
                  "C4" if a.cond_inc else "custom")
        kitems = f"{N // 1000}k" if N % 1000 == 0 else str(N)
        out = {
            # (BASELINE.json's metric names the headline shape; any other shape names its own so that two lines are never
            #  compared across configurations)
            "metric": f"train docs/sec at |items|={kitems} h={h}" + ("" if a.dtype == "f32" else f" {a.dtype}"),
            "config_id": cfg_id,
            "value": round(docs_per_s, 1), "unit": "docs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg_name} synthetic Bags: |items|={N}, hidden={h}, code={c}, "
                                   f"{'fp32' if a.dtype == 'f32' else 'bf16 MFMA inputs / fp32 accumulate, master weights and Adam'}, "
                                   f"batch={B} docs/GPU/step, full partial_fit (ae+disc+gen, 4 Adam) inside the "
                                   f"AdversarialAutoEncoder.fit epoch loop (permutation batches of a corpus resident in HBM)",
                       "n_items": N, "n_hidden": h, "n_code": c, "batch_per_gpu": B, "global_batch": Bg,
                       "cond_inc": a.cond_inc, "nnz_per_batch": round(nnz_per_batch, 1), "rng": "device",
                       "timed_repeats": len(dts), "repeat_ms_per_step": [round(d / a.steps * 1e3, 4) for d in dts],
                       "profiled_repeat": 0,
                       "parallelism": (f"dp{world}" if not use_dp else
                                       (f"dp{world}, item slices of both vocabulary-wide layers + replicated hidden stacks on the global batch, "
                                        f"3 all-reduces of partial sums per step" if a.dp == "shard" else
                                        f"dp{world}, decoder output layer and encoder first layer sharded over the vocabulary"
                                        if dp_desc["shard_first"] else
                                        f"dp{world}, decoder output layer sharded over the vocabulary") if vocab else
                                       f"dp{world}, replicated decoder")},
            "roofline": roofline, "step_roofline": step_roofline, "cpu_baseline": cpu, "kernels": kstats,
            **({"roofline_critical": roofline_critical} if roofline_critical else {}),
            "losses_last_step": [round(float(x), 5) for x in losses],
        }
        if raw:
            out["raw_step"] = raw
        if extra:
            out["extra"] = extra
        if dp_breakdown:
            out["dp_breakdown"] = dp_breakdown
        if use_dp:
            out["dp"] = a.dp
        if dp_other:
            out["dp_other"] = dp_other
        if use_dp and dp_desc["comm"] is not None:
            out["collectives_per_step"] = dp_desc["comm"]
            native = dp_desc["native"]
            rccl = dist is not None and str(dist.get_backend()).lower() == "nccl"
            out["dp_step_driver"] = (("library call (" + ("aae_shard_step" if a.dp == "shard" else "aae_dp_step") + "), " + ("RCCL communicator of the library" if rccl else
                                                                         "collectives through host-staged callbacks (functional check)"))
                                     if native else "python phases over torch.distributed")
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
