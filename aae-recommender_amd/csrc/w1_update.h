// K9 + K10 of the sparse first encoder layer: weight gradient and optimiser in ONE pass over the batch's distinct items,
// without float atomics.  reference: the backward of enc.lin1 on the dense [B, N] input followed by
// enc_optim.step() / gen_optim.step() (aae.py:703-706, 740-742); only the rows of items present in the batch have a
// gradient (the zero-gradient updates of all other rows are deferred, kernels.h).
//
//   gW1T[item, :] = sum over the batch rows b that hold the item, IN ASCENDING ROW ORDER, of (v_be * s_b) * ga1[b, :]
//
// One WAVEFRONT per distinct item (grid-stride over the step's unique-item list; no workgroup barrier anywhere).  The
// batch's entries come bucketed by 32-item tile (buckets.h - the counting sort the fused output layer needs anyway, built
// off the critical path); the order INSIDE a bucket is whatever the sort's LDS atomics produced, so the wave first marks
// its item's rows in an LDS bitmap, ranks them by prefix popcount and only then walks them: the summation order is a
// function of the batch alone and two runs of a training loop agree bit for bit.  (r1/r2 scattered the products with
// global_atomic_add_f32: a swapped pair of adds moves a weight by an ulp, and the adversarial dynamics turned that into
// 4e-5 .. 1e-4 in the predictions 40-120 steps later in 3-12 % of the runs of the 120-step parity recipe.)
// The optimiser runs on the sum in registers: the gradient rows never exist in HBM (fused mode), or are written once
// (export mode: aae_w1_export packs them for the data-parallel exchange).  A lane owns 4 consecutive columns (one
// 16-byte access per row); the row's parameter and moments are requested before the entry lists are read, so the kernel
// is four dependent memory round trips deep: item id -> tile range (+ row state) -> entries -> dL/d(a1) rows.
//
// Preconditions (aae_batch): column indices unique within a row - a repeated (row, item) pair collapses to one term.
#pragma once
#include "buckets.h"

namespace aae {

struct W1Items {
    const int* ulist; const int* ucount;                              // the step's distinct items
    const int* tstart; const int* eb; const int* en; const float* ev; // tile buckets of the running batch
    const float* ga1; int ld;                                         // dL/d(a1) [rows][ld], ld % 4 == 0
    int rpb; size_t bstride;                                          // rpb > 0: blocks of rpb rows, bstride floats apart
                                                                      // (the ranks' packets of an all-gather, read in place)
    const float* rscale; int rows; int h;
    float* W; float* M; float* V; int ldw;                            // enc.lin1 item-major + the optimiser's moments (ldw % 4 == 0)
    float* gout;                                                      // != NULL: export the gradient rows, no update
    const OptScalars* sc; int* tsync; const long long* step_ctr; int mark_synced;
};

// LDS words one wavefront needs for a batch of `rows` documents: bitmap + prefix + (row, value) list
__host__ __device__ inline size_t w1_items_wave_words(int rows) { return 2 * (size_t)((rows + 31) >> 5) + 2 * (size_t)rows; }

// (LDS traffic of one wave: the hardware runs a wave's DS instructions in order; the fence keeps the compiler from moving
//  a lane's read above another lane's write)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// vblock / vgrid: this workgroup's index among the workgroups that share the item list (the kernel below, or the first-layer
// workgroups of the grouped weight-gradient launch, chain.h)
__device__ __forceinline__ void w1_item_update_body(const W1Items& a, unsigned* w1_lds, int vblock, int vgrid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int nw = (a.rows + 31) >> 5;
    unsigned* bm = w1_lds + (size_t)wave * w1_items_wave_words(a.rows);   // [nw]   bit b: row b holds the item
    int* pre = reinterpret_cast<int*>(bm + nw);                            // [nw]   rows below word w that hold it
    int* rl = pre + nw;                                                    // [rows] its rows, ascending
    float* xl = reinterpret_cast<float*>(rl + a.rows);                     // [rows] their values
    const int cnt = *a.ucount;
    const bool upd = a.gout == nullptr;
    OptScalars s;
    if (upd) s = *a.sc;
    const int c0 = 4 * lane;                               // this lane's columns of a 256-column chunk
    for (int u = vblock * nwave + wave; u < cnt; u += vgrid * nwave) {
        const int item = a.ulist[u];
        const int tile = item / kTI, it = item - tile * kTI;
        const int e0 = a.tstart[tile], e1 = a.tstart[tile + 1];
        // the row's state for the first 256 columns travels while the lists are built (clamped: lanes beyond h re-read the
        // last float4 of the row and never store)
        const size_t o0 = (size_t)item * a.ldw + min(c0, a.ldw - 4);
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pm = pw, pv = pw;
        if (upd) {
            pw = *reinterpret_cast<const float4*>(a.W + o0);
            if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o0); pv = *reinterpret_cast<const float4*>(a.V + o0); }
        }
        for (int i = lane; i < nw; i += 64) bm[i] = 0u;
        wave_lds_sync();
        // pass 1: mark the rows; a lane keeps its first match in registers (tiles beyond 64 entries: re-read in pass 2)
        int my_r = -1; float my_x = 0.f;
        for (int base = e0; base < e1; base += 64) {
            const int e = base + lane;
            const int ec = min(e, e1 - 1);
            const int en_e = a.en[ec], r = a.eb[ec]; const float x = a.ev[ec];
            if (e < e1 && en_e == it) {
                atomicOr(&bm[r >> 5], 1u << (r & 31));
                if (base == e0) { my_r = r; my_x = x; }
            }
        }
        wave_lds_sync();
        int n = 0;                                         // exclusive prefix of the words' popcounts
        for (int base = 0; base < nw; base += 64) {
            const int i = base + lane;
            const int c = i < nw ? __popc(bm[i]) : 0;
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            if (i < nw) pre[i] = n + inc - c;
            n += __shfl(inc, 63, 64);
        }
        wave_lds_sync();
        if (my_r >= 0) {
            const int k = pre[my_r >> 5] + __popc(bm[my_r >> 5] & ((1u << (my_r & 31)) - 1u));
            rl[k] = my_r; xl[k] = my_x;
        }
        for (int base = e0 + 64; base < e1; base += 64) {  // (hot tiles only)
            const int e = base + lane;
            const int ec = min(e, e1 - 1);
            const int en_e = a.en[ec], r = a.eb[ec]; const float x = a.ev[ec];
            if (e < e1 && en_e == it) {
                const int k = pre[r >> 5] + __popc(bm[r >> 5] & ((1u << (r & 31)) - 1u));
                rl[k] = r; xl[k] = x;
            }
        }
        wave_lds_sync();
        for (int cb = 0; cb < a.h; cb += 256) {            // 256-column chunks (one for n_hidden <= 256)
            const int c = cb + c0;
            const int cc = min(c, a.ld - 4);               // (lanes beyond the row: clamped, never stored)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // kU rows at a time: their loads are independent and travel together (a popular item sits in most of the batch's
            // rows - one dependent L2 round trip per row was the kernel's critical path), the adds stay in row order
            constexpr int kU = 16;
            for (int i0 = 0; i0 < n; i0 += kU) {
                float4 g[kU]; float x[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    const int i = min(i0 + j, n - 1);
                    const int r = rl[i];
                    x[j] = (i0 + j < n ? xl[i] : 0.f) * a.rscale[r];  // (beyond the list: the last row again, times zero; the row
                                                                      //  scale travels with the row itself: one round trip)
                    const float* grow = a.rpb > 0 ? a.ga1 + (size_t)(r / a.rpb) * a.bstride + (size_t)(r % a.rpb) * a.ld
                                                  : a.ga1 + (size_t)r * a.ld;
                    g[j] = *reinterpret_cast<const float4*>(grow + cc);
                }
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    acc.x += x[j] * g[j].x; acc.y += x[j] * g[j].y; acc.z += x[j] * g[j].z; acc.w += x[j] * g[j].w;
                }
            }
            const size_t o = (size_t)item * a.ldw + min(c, a.ldw - 4);
            if (cb > 0 && upd) {
                pw = *reinterpret_cast<const float4*>(a.W + o);
                if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o); pv = *reinterpret_cast<const float4*>(a.V + o); }
            }
            if (c >= a.h) continue;
            // (columns h .. ldw - 1 of a row are padding: a lane's float4 may straddle h - the pad columns get a zero gradient)
            if (c + 1 >= a.h) acc.y = 0.f;
            if (c + 2 >= a.h) acc.z = 0.f;
            if (c + 3 >= a.h) acc.w = 0.f;
            if (!upd) { *reinterpret_cast<float4*>(a.gout + o) = acc; continue; }
            adam_update(pw.x, pm.x, pv.x, acc.x, s); adam_update(pw.y, pm.y, pv.y, acc.y, s);
            adam_update(pw.z, pm.z, pv.z, acc.z, s); adam_update(pw.w, pm.w, pv.w, acc.w, s);
            *reinterpret_cast<float4*>(a.W + o) = pw;
            if (!s.is_sgd) { *reinterpret_cast<float4*>(a.M + o) = pm; *reinterpret_cast<float4*>(a.V + o) = pv; }
        }
        if (upd && a.mark_synced && lane == 0) a.tsync[item] = (int)*a.step_ctr;
        wave_lds_sync();                                   // the lists are rebuilt for the next item
    }
}

__global__ __launch_bounds__(256) void w1_item_update_kernel(W1Items a) {
    extern __shared__ unsigned w1_lds_dyn[];
    w1_item_update_body(a, w1_lds_dyn, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace aae
