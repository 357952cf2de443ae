#!/usr/bin/env python3
"""How to bring a [docs, items] float32 score matrix back to the host: timings of the candidate schemes."""
import time, torch, numpy as np
dev = torch.device("cuda:0")
B, N, NB = 100, 100000, 32
src = torch.randn(B, N, device=dev)
torch.cuda.synchronize()

def t(name, fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:58s} {NB * B / dt:9.0f} docs/s  ({1e3 * dt / NB:.2f} ms/batch)", flush=True)
    return out

def cur():
    return np.vstack([src.cpu().numpy() for _ in range(NB)])
t("batch.cpu().numpy() + np.vstack (current)", cur)

def prealloc_pageable():
    out = torch.empty(NB * B, N)
    for i in range(NB):
        out[i * B:(i + 1) * B].copy_(src)
    return out.numpy()
t("preallocated pageable output, slice.copy_(dev)", prealloc_pageable)

def pinned_full():
    out = torch.empty(NB * B, N, pin_memory=True)
    for i in range(NB):
        out[i * B:(i + 1) * B].copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return out.numpy()
t("pinned full output (allocated per call), async copies", pinned_full)

pin = torch.empty(NB * B, N, pin_memory=True)
def pinned_reuse():
    for i in range(NB):
        pin[i * B:(i + 1) * B].copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return pin.numpy()
t("pinned full output (already allocated), async copies", pinned_reuse)

stage = [torch.empty(B, N, pin_memory=True) for _ in range(2)]
ev = [torch.cuda.Event() for _ in range(2)]
def staged():
    out = np.empty((NB * B, N), dtype=np.float32)
    for i in range(NB):
        k = i & 1
        if i >= 2:
            ev[k].synchronize(); out[(i - 2) * B:(i - 1) * B] = stage[k].numpy()
        stage[k].copy_(src, non_blocking=True); ev[k].record()
    for i in range(NB - 2, NB):
        k = i & 1
        ev[k].synchronize(); out[i * B:(i + 1) * B] = stage[k].numpy()
    return out
t("2 pinned staging buffers + memcpy into pageable output", staged)
t0 = time.perf_counter(); x = torch.empty(NB * B, N, pin_memory=True); print(f"pin alloc of {x.numel() * 4 / 2**30:.2f} GiB: {time.perf_counter() - t0:.3f} s")
