#!/usr/bin/env python3
"""Phase clocks of the hybrid first-layer item form (w1_update.h, library built with -DW1_TS: tools/debug/w1_ts.sh): per workgroup
and wave - start | list / scalars read | item + tile range | entries scanned | rows added + optimiser | items of all rounds
done | barrier | deferred (many-row) items done.  C3, the LAST weight-gradient launch with such workgroups before the read-out
(the generator phase's)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aae-recommender_amd"))
import numpy as np, torch
from aaerec._hip import HipAAE, DeviceCSR, load_library
from tools.synth import throughput_corpus, init_params
N, h, c, B = 100000, 200, 50, 100
X = throughput_corpus(64 * B, N, seed=1234)
m = HipAAE(N, h, c, max_batch=B, max_nnz=B * 256)
m.load_params(init_params(N, h, c, seed=0))
csr = DeviceCSR(X, m.device)
stop_after = sys.argv[1] if len(sys.argv) > 1 else "gen"
for i in range(40):
    m.prefetch(csr, ((i + 1) % 64) * B, B)
    m.step(csr, (i % 64) * B, B)
torch.cuda.synchronize()
lib = load_library()
nb = 1024
buf = (C.c_ulonglong * (nb * 32))()
lib.aae_debug_w1_ts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.aae_debug_w1_ts(buf, nb * 32) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4, 8).astype(np.float64) * 0.01
live = t[:, 0, 0] > 0
t = t[live]
t0 = t[:, :, 0].min()
dur = t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1)
names = ["list / scalars", "item + tile range", "scan", "rows + optimiser", "later rounds", "barrier", "deferred items"]
print(f"{len(t)} workgroups; duration mean {dur.mean():.2f} us, longest {dur.max():.2f} us; last end {t[:, :, 7].max() - t0:.2f} us after the first start")
ph = np.diff(t, axis=2)                    # [wg, wave, 7]
print("mean per phase and wave (us):")
for k, nme in enumerate(names):
    print(f"   {nme:18s}" + "".join(f" {ph[:, w, k].mean():6.2f}" for w in range(4)))
for idx in np.argsort(-dur)[:4]:
    print(f"workgroup with duration {dur[idx]:.2f} us (start {t[idx, :, 0].min() - t0:.2f}):")
    for w in range(4):
        print("   wave %d: " % w + " | ".join(f"{nme} {ph[idx, w, k]:.2f}" for k, nme in enumerate(names)))
