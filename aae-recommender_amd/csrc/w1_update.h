// K9 + K10 of the sparse first encoder layer: weight gradient and optimiser in ONE pass over the batch's distinct items,
// without float atomics.  reference: the backward of enc.lin1 on the dense [B, N] input followed by
// enc_optim.step() / gen_optim.step() (aae.py:703-706, 740-742); only the rows of items present in the batch have a
// gradient (the zero-gradient updates of all other rows are deferred, kernels.h).
//
//   gW1T[item, :] = sum over the batch rows b that hold the item, IN ASCENDING ROW ORDER, of (v_be * s_b) * ga1[b, :]
//
// One 256-thread workgroup per distinct item (grid-stride over the step's unique-item list).  The batch's entries come
// bucketed by 32-item tile (buckets.h - the counting sort the fused output layer needs anyway, built off the critical
// path); the order INSIDE a bucket is whatever the sort's LDS atomics produced, so the workgroup first marks its item's
// rows in an LDS bitmap, ranks them by prefix popcount and only then walks them - the summation order is a function of
// the batch alone and two runs of a training loop agree bit for bit.  (r1/r2 scattered the products with
// global_atomic_add_f32: a swapped pair of adds moves a weight by an ulp, and the adversarial dynamics turned that into
// 4e-5 .. 1e-4 in the predictions 40-120 steps later in 3-12 % of the runs of the 120-step parity recipe.)
// The optimiser runs on the sum in registers: the gradient rows never exist in HBM (fused mode), or are written once
// (export mode: aae_w1_export packs them for the data-parallel exchange).
//
// Preconditions (aae_batch): column indices unique within a row - a repeated (row, item) pair collapses to one term.
#pragma once
#include "buckets.h"

namespace aae {

struct W1Items {
    const int* ulist; const int* ucount;                              // the step's distinct items
    const int* tstart; const int* eb; const int* en; const float* ev; // tile buckets of the running batch
    const float* ga1; int ld;                                         // dL/d(a1) [rows][ld]
    int rpb; size_t bstride;                                          // rpb > 0: blocks of rpb rows, bstride floats apart
                                                                      // (the ranks' packets of an all-gather, read in place)
    const float* rscale; int rows; int h;
    float* W; float* M; float* V; int ldw;                            // enc.lin1 item-major + the optimiser's moments
    float* gout;                                                      // != NULL: export the gradient rows, no update
    const OptScalars* sc; int* tsync; const long long* step_ctr; int mark_synced;
};

// dynamic LDS of w1_item_update_kernel for a batch of `rows` documents
inline size_t w1_items_lds_bytes(int rows) { return sizeof(int) * (2 * (size_t)((rows + 31) >> 5) + 2 * (size_t)rows); }

__global__ __launch_bounds__(256) void w1_item_update_kernel(W1Items a) {
    extern __shared__ unsigned w1_lds[];
    __shared__ int s_n;
    const int nw = (a.rows + 31) >> 5;
    unsigned* bm = w1_lds;                                 // [nw]   bit b: row b holds the item
    int* pre = reinterpret_cast<int*>(bm + nw);            // [nw]   rows below word w that hold it
    int* rl = pre + nw;                                    // [rows] its rows, ascending
    float* xl = reinterpret_cast<float*>(rl + a.rows);     // [rows] value * row scale
    const int tid = threadIdx.x, lane = tid & 63;
    const int cnt = *a.ucount;
    OptScalars s;
    if (!a.gout) s = *a.sc;
    for (int u = blockIdx.x; u < cnt; u += gridDim.x) {
        const int item = a.ulist[u];
        const int tile = item / kTI, it = item - tile * kTI;
        const int e0 = a.tstart[tile], e1 = a.tstart[tile + 1];
        for (int i = tid; i < nw; i += 256) bm[i] = 0u;
        __syncthreads();
        for (int e = e0 + tid; e < e1; e += 256)
            if (a.en[e] == it) { const int r = a.eb[e]; atomicOr(&bm[r >> 5], 1u << (r & 31)); }
        __syncthreads();
        if (tid < 64) {                                    // exclusive prefix of the words' popcounts
            int run = 0;
            for (int base = 0; base < nw; base += 64) {
                const int i = base + lane;
                const int c = i < nw ? __popc(bm[i]) : 0;
                int inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
                if (i < nw) pre[i] = run + inc - c;
                run += __shfl(inc, 63, 64);
            }
            if (tid == 0) s_n = run;
        }
        __syncthreads();
        for (int e = e0 + tid; e < e1; e += 256)
            if (a.en[e] == it) {
                const int r = a.eb[e];
                const int k = pre[r >> 5] + __popc(bm[r >> 5] & ((1u << (r & 31)) - 1u));
                rl[k] = r; xl[k] = a.ev[e] * a.rscale[r];
            }
        __syncthreads();
        const int n = s_n;
        for (int c = tid; c < a.h; c += 256) {
            float acc = 0.f;
            for (int i = 0; i < n; ++i) {
                const int r = rl[i];
                const float* grow = a.rpb > 0 ? a.ga1 + (size_t)(r / a.rpb) * a.bstride + (size_t)(r % a.rpb) * a.ld
                                              : a.ga1 + (size_t)r * a.ld;
                acc += xl[i] * grow[c];
            }
            const size_t o = (size_t)item * a.ldw + c;
            if (a.gout) { a.gout[o] = acc; continue; }
            float p = a.W[o], m = s.is_sgd ? 0.f : a.M[o], v = s.is_sgd ? 0.f : a.V[o];
            adam_update(p, m, v, acc, s);
            a.W[o] = p;
            if (!s.is_sgd) { a.M[o] = m; a.V[o] = v; }
        }
        if (!a.gout && a.mark_synced && tid == 0) a.tsync[item] = (int)*a.step_ctr;
        __syncthreads();                                   // the lists are rebuilt for the next item
    }
}

}  // namespace aae
