// K9 + K10 of the sparse first encoder layer: weight gradient and optimiser in ONE pass over the batch's distinct items,
// without float atomics.  reference: the backward of enc.lin1 on the dense [B, N] input followed by
// enc_optim.step() / gen_optim.step() (aae.py:703-706, 740-742); only the rows of items present in the batch have a
// gradient (the zero-gradient updates of all other rows are deferred, kernels.h).
//
//   gW1T[item, :] = sum over the batch rows b that hold the item, IN ASCENDING ROW ORDER, of (v_be * s_b) * ga1[b, :]
//
// One WORKGROUP (4 waves) per distinct item (grid-stride over the step's unique-item list).  The batch's entries come
// bucketed by 32-item tile (buckets.h - the counting sort the fused output layer needs anyway, built off the critical
// path); the order INSIDE a bucket is whatever the sort's LDS atomics produced, so the workgroup first marks its item's
// rows in an LDS bitmap, ranks them by prefix popcount and only then walks them: the summation order is a function of the
// batch alone and two runs of a training loop agree bit for bit.  (r1/r2 scattered the products with
// global_atomic_add_f32: a swapped pair of adds moves a weight by an ulp, and the adversarial dynamics turned that into
// 4e-5 .. 1e-4 in the predictions 40-120 steps later in 3-12 % of the runs of the 120-step parity recipe.)
// The optimiser runs on the sum in registers: the gradient rows never exist in HBM (fused mode), or are written once
// (export mode: aae_w1_export packs them for the data-parallel exchange).  The row's parameter and moments are requested
// before the entry lists are read, so the kernel is four dependent memory round trips deep: item id -> tile range (+ row
// state) -> entries -> dL/d(a1) rows.  (Measured on the way, r3: one wavefront per item with its rows added one dependent
// round trip at a time - 29 us per launch, the most popular item's rows being the critical path; 16 rows' loads in
// flight at a time - 17 us at batch 100, but 43 us on an item slice that sees 800 rows; this form: see DESIGN.md.)
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

// LDS words one workgroup needs for a batch of `rows` documents: bitmap + prefix + (row, value) list + the waves' partial sums
__host__ __device__ inline size_t w1_items_lds_words(int rows) { return 2 * (size_t)((rows + 31) >> 5) + 2 * (size_t)rows + 4 * 256 + 4; }

// One 256-thread workgroup per distinct item (grid-stride over the step's unique-item list).
// vblock / vgrid: this workgroup's index among the workgroups that share the item list (the kernel below, or the first-layer
// workgroups of the grouped weight-gradient launch, chain.h).
//   rows of the item   marked in an LDS bitmap, ranked by prefix popcount -> (row, value) list in ascending row order;
//   the sum            up to 16 rows: wave 0 adds them in row order.  More (a popular item sits in most of a batch's rows:
//                      hundreds on an item slice that sees the global batch of 8 ranks): the list is cut into four
//                      contiguous quarters, wave w adds its quarter in row order - 16 rows' loads in flight at a time -
//                      and the four partial sums are added in wave order: still a function of the batch alone;
//   the update         thread t owns column t of a 256-column chunk; its parameter and moments were requested before the
//                      lists were built.
__device__ __forceinline__ void w1_item_update_one(const W1Items& a, unsigned* w1_lds, int u, const OptScalars& s, bool upd) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (a.rows + 31) >> 5;
    unsigned* bm = w1_lds;                                 // [nw]   bit b: row b holds the item
    int* pre = reinterpret_cast<int*>(bm + nw);            // [nw]   rows below word w that hold it
    int* rl = pre + nw;                                    // [rows] its rows, ascending
    float* xl = reinterpret_cast<float*>(rl + a.rows);     // [rows] their values
    float* part = xl + a.rows;                             // [4][256] the waves' partial sums of a column chunk
    int* s_n = reinterpret_cast<int*>(part + 4 * 256);
    {
        const int item = a.ulist[u];
        const int tile = item / kTI, it = item - tile * kTI;
        const int e0 = a.tstart[tile], e1 = a.tstart[tile + 1];
        // this thread's column of the first chunk: its state travels while the lists are built
        const size_t o0 = (size_t)item * a.ldw + min(tid, a.ldw - 1);
        float pw = 0.f, pm = 0.f, pv = 0.f;
        if (upd) {
            pw = a.W[o0];
            if (!s.is_sgd) { pm = a.M[o0]; pv = a.V[o0]; }
        }
        for (int i = tid; i < nw; i += 256) bm[i] = 0u;
        __syncthreads();
        // pass 1: mark the rows; a thread keeps its first match in registers (tiles beyond 256 entries: re-read in pass 2)
        // (r5: four groups of 256 entries requested together.  The tile of the 32 most popular items holds a third to a half of a
        //  wide batch's entries - 9 000 at C4, 3 400 at C3 with 512 rows - and each of its items' workgroups walked them 256 at a
        //  time, one memory round trip per group and pass: the long pole of the weight-gradient launches that carry these blocks)
        int my_r = -1; float my_x = 0.f;
        {
            const int ec = min(e0 + tid, e1 - 1);          // the first group also keeps its value
            const int en_e = a.en[ec], r = a.eb[ec]; const float x = a.ev[ec];
            if (e0 + tid < e1 && en_e == it) { atomicOr(&bm[r >> 5], 1u << (r & 31)); my_r = r; my_x = x; }
        }
        for (int base = e0 + 256; base < e1; base += 1024) {
            int en4[4], r4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int ec = min(base + 256 * j + tid, e1 - 1); en4[j] = a.en[ec]; r4[j] = a.eb[ec]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (base + 256 * j + tid < e1 && en4[j] == it) atomicOr(&bm[r4[j] >> 5], 1u << (r4[j] & 31));
        }
        __syncthreads();
        if (wave == 0) {                                   // exclusive prefix of the words' popcounts
            int n = 0;
            for (int base = 0; base < nw; base += 64) {
                const int i = base + lane;
                const int c = i < nw ? __popc(bm[i]) : 0;
                int inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
                if (i < nw) pre[i] = n + inc - c;
                n += __shfl(inc, 63, 64);
            }
            if (lane == 0) *s_n = n;
        }
        __syncthreads();
        if (my_r >= 0) {
            const int k = pre[my_r >> 5] + __popc(bm[my_r >> 5] & ((1u << (my_r & 31)) - 1u));
            rl[k] = my_r; xl[k] = my_x;
        }
        for (int base = e0 + 256; base < e1; base += 1024) {    // (hot tiles only)
            int en4[4], r4[4]; float x4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int ec = min(base + 256 * j + tid, e1 - 1); en4[j] = a.en[ec]; r4[j] = a.eb[ec]; x4[j] = a.ev[ec]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (base + 256 * j + tid < e1 && en4[j] == it) {
                    const int r = r4[j];
                    const int k = pre[r >> 5] + __popc(bm[r >> 5] & ((1u << (r & 31)) - 1u));
                    rl[k] = r; xl[k] = x4[j];
                }
        }
        __syncthreads();
        const int n = *s_n;
        // this wave's share of the list
        const int q = n <= 16 ? n : (n + 3) >> 2;
        const int i_lo = min(wave * q, n), i_hi = min(i_lo + q, n);
        for (int cb = 0; cb < a.h; cb += 256) {            // 256-column chunks (one for n_hidden <= 256)
            const int c4 = cb + 4 * lane;                  // the accumulating lanes own 4 consecutive columns
            const int cc = min(c4, a.ld - 4);              // (lanes beyond the row: clamped, masked below)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int kU = 16;                         // rows whose loads travel together; the adds stay in row order
            for (int i0 = i_lo; i0 < i_hi; i0 += kU) {
                float4 g[kU]; float x[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    const int i = min(i0 + j, i_hi - 1);
                    const int r = rl[i];
                    x[j] = (i0 + j < i_hi ? xl[i] : 0.f) * a.rscale[r];   // (beyond the share: its last row again, times zero; the
                                                                          //  row scale travels with the row: one round trip)
                    const float* grow = a.rpb > 0 ? a.ga1 + (size_t)(r / a.rpb) * a.bstride + (size_t)(r % a.rpb) * a.ld
                                                  : a.ga1 + (size_t)r * a.ld;
                    g[j] = *reinterpret_cast<const float4*>(grow + cc);
                }
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    acc.x += x[j] * g[j].x; acc.y += x[j] * g[j].y; acc.z += x[j] * g[j].z; acc.w += x[j] * g[j].w;
                }
            }
            // (columns h .. of a row are padding: a lane's float4 may straddle h - the pad columns get a zero gradient)
            if (c4 >= a.h) acc.x = 0.f;
            if (c4 + 1 >= a.h) acc.y = 0.f;
            if (c4 + 2 >= a.h) acc.z = 0.f;
            if (c4 + 3 >= a.h) acc.w = 0.f;
            *reinterpret_cast<float4*>(part + wave * 256 + 4 * lane) = acc;
            __syncthreads();
            const int c = cb + tid;
            const float g = ((part[tid] + part[256 + tid]) + part[512 + tid]) + part[768 + tid];      // wave order
            const size_t o = (size_t)item * a.ldw + min(c, a.ldw - 1);
            if (cb > 0 && upd) {
                pw = a.W[o];
                if (!s.is_sgd) { pm = a.M[o]; pv = a.V[o]; }
            }
            if (c < a.ldw) {
                if (!upd) a.gout[o] = g;
                else {
                    adam_update(pw, pm, pv, g, s);
                    a.W[o] = pw;
                    if (!s.is_sgd) { a.M[o] = pm; a.V[o] = pv; }
                }
            }
            __syncthreads();                               // part / the lists are rewritten next
        }
        if (upd && a.mark_synced && tid == 0) a.tsync[item] = (int)*a.step_ctr;
    }
}
__device__ __forceinline__ void w1_item_update_body(const W1Items& a, unsigned* w1_lds, int vblock, int vgrid) {
    const int cnt = *a.ucount;
    const bool upd = a.gout == nullptr;
    OptScalars s;
    if (upd) s = *a.sc;
    for (int u = vblock; u < cnt; u += vgrid) w1_item_update_one(a, w1_lds, u, s, upd);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same update with one WAVE per item, for WIDE batches (beyond one fused launch's 112 rows: batch 512 on one GPU,
// the global batch of an item slice).  A batch of 512 documents has ~7 000 distinct items, nearly all of them in one or two
// rows: with a 4-wave workgroup each, a CU holds 8 items at a time and the launch ran in ~9 rounds of a latency chain four
// memory round trips deep (75-98 us beside the deferred output-layer launch, r3).  A wave needs no barrier and no partial
// sums: lane l owns the columns 4 l .. 4 l + 3 of a 256-column chunk for the sum AND for the optimiser (float4 state), and
// a CU holds 32 items.  The rows of the item are collected from its tile's entries (lanes = entries, ballot + prefix
// popcount), ranked by row (every listed row compared with every other: n <= kW1WaveRows) and added in ascending row order
// - the summation order is a function of the batch alone, as in the workgroup form.  An item with more rows than
// kW1WaveRows (a handful of head items of the Zipf vocabulary: the one item in every row would be 512 dependent-ish
// loads for ONE wave) goes to the HOT list instead, which a launch of the workgroup form (w1_item_update_kernel with
// ulist / ucount = the hot list) works off behind this one.
// LDS: 4 x kW1WaveRows words per wave.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kW1WaveRows = 48;
__host__ __device__ inline size_t w1_wave_lds_words(int waves) { return (size_t)waves * 4 * kW1WaveRows; }

__device__ __forceinline__ void w1_item_wave_body(const W1Items& a, int* __restrict__ hot, int* __restrict__ hot_count,
                                                  unsigned* lds, int vwave, int vwaves) {
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6);
    int* rl = reinterpret_cast<int*>(lds) + wv * 4 * kW1WaveRows;      // rows as found
    float* xl = reinterpret_cast<float*>(rl + kW1WaveRows);             // their values
    int* rs = reinterpret_cast<int*>(xl + kW1WaveRows);                 // rows, ascending
    float* xs = reinterpret_cast<float*>(rs + kW1WaveRows);             // their values x row scale
    const int cnt = *a.ucount;
    const bool upd = a.gout == nullptr;
    OptScalars s;
    if (upd) s = *a.sc;
    for (int u = vwave; u < cnt; u += vwaves) {
        const int item = a.ulist[u];
        const int tile = item / kTI, it = item - tile * kTI;
        const int e0 = a.tstart[tile], e1 = a.tstart[tile + 1];
        // the first column chunk's state travels while the rows are collected
        const int c4 = 4 * lane;
        const size_t o0 = (size_t)item * a.ldw + min(c4, a.ldw - 4);
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pm = pw, pv = pw;
        if (upd) {
            pw = *reinterpret_cast<const float4*>(a.W + o0);
            if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o0); pv = *reinterpret_cast<const float4*>(a.V + o0); }
        }
        int n = 0;
        for (int base = e0; base < e1; base += 256) {      // (r5: four groups of 64 entries requested together, taken in order)
            int en4[4], r4[4]; float x4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int ec = min(base + 64 * j + lane, e1 - 1); en4[j] = a.en[ec]; r4[j] = a.eb[ec]; x4[j] = a.ev[ec]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = base + 64 * j + lane < e1 && en4[j] == it;
                const unsigned long long bal = __ballot(hit);
                const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
                if (hit && k < kW1WaveRows) { rl[k] = r4[j]; xl[k] = x4[j]; }
                n += __popcll(bal);
            }
        }
        if (n > kW1WaveRows) {                          // a head item: the workgroup form takes it
            if (lane == 0) hot[atomicAdd(hot_count, 1)] = item;
            continue;
        }
        // rank by row (distinct rows: the batch's CSR is canonical) -> ascending lists
        if (lane < n) {
            const int r = rl[lane];
            int k = 0;
            for (int j = 0; j < n; ++j) k += (rl[j] < r || (rl[j] == r && j < lane)) ? 1 : 0;   // total order: a duplicate (row, item) pair of a non-canonical CSR still gets a slot of its own
            rs[k] = r; xs[k] = xl[lane];        // (its row scale is requested with the row itself below: one round trip less)
        }
        // (one wave: its LDS writes are in order before its reads - no barrier)
        for (int cb = 0; cb < a.h; cb += 256) {
            const int cc = min(cb + c4, a.ld - 4);      // (lanes beyond the row: clamped, masked below)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int kU = 8;                       // rows whose loads travel together; the adds stay in row order
            for (int i0 = 0; i0 < n; i0 += kU) {
                float4 g[kU]; float x[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    const int i = min(i0 + j, n - 1);
                    const int r = rs[i];
                    x[j] = (i0 + j < n ? xs[i] : 0.f) * a.rscale[r];
                    const float* grow = a.rpb > 0 ? a.ga1 + (size_t)(r / a.rpb) * a.bstride + (size_t)(r % a.rpb) * a.ld
                                                  : a.ga1 + (size_t)r * a.ld;
                    g[j] = *reinterpret_cast<const float4*>(grow + cc);
                }
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    acc.x += x[j] * g[j].x; acc.y += x[j] * g[j].y; acc.z += x[j] * g[j].z; acc.w += x[j] * g[j].w;
                }
            }
            const int c = cb + c4;
            if (c >= a.h) acc.x = 0.f;                  // (columns h .. of a row are padding: a zero gradient)
            if (c + 1 >= a.h) acc.y = 0.f;
            if (c + 2 >= a.h) acc.z = 0.f;
            if (c + 3 >= a.h) acc.w = 0.f;
            const size_t o = (size_t)item * a.ldw + min(c, a.ldw - 4);
            if (cb > 0 && upd) {
                pw = *reinterpret_cast<const float4*>(a.W + o);
                if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o); pv = *reinterpret_cast<const float4*>(a.V + o); }
            }
            if (c < a.ldw) {
                if (!upd) *reinterpret_cast<float4*>(a.gout + o) = acc;
                else {
                    adam_update(pw.x, pm.x, pv.x, acc.x, s); adam_update(pw.y, pm.y, pv.y, acc.y, s);
                    adam_update(pw.z, pm.z, pv.z, acc.z, s); adam_update(pw.w, pm.w, pv.w, acc.w, s);
                    *reinterpret_cast<float4*>(a.W + o) = pw;
                    if (!s.is_sgd) { *reinterpret_cast<float4*>(a.M + o) = pm; *reinterpret_cast<float4*>(a.V + o) = pv; }
                }
            }
        }
        if (upd && a.mark_synced && lane == 0) a.tsync[item] = (int)*a.step_ctr;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Batches of one fused launch (<= 112 rows), inside the grouped weight-gradient launch: BOTH forms in one workgroup.  Nearly
// every item of such a batch sits in one or two rows; with a 4-wave workgroup per item the ~1 800 items of 100 documents are
// 3 200 workgroups, and beside the deferred output-layer launch (which holds 9/16 of the CUs) they run in ~7 rounds of a
// latency chain four round trips deep: 27 us for the ae phase's launch against 12.5 us for the gen phase's, which runs alone
// (kernel trace, r3).  Here a workgroup takes FOUR items, one per wave, in the wave form when the item has at most 16 rows -
// exactly the case in which the workgroup form lets wave 0 add them all, in the same order: the results are the workgroup
// form's bit for bit - and notes the others in an LDS bitmap; behind a barrier the four waves take those together in the
// workgroup form.  No hot list, no second launch.
// LDS: max(the workgroup form's, 4 x 4 x 16 words) + kW1HybWords.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef W1_TS       // (debug builds: hybrid-form phase clocks per workgroup and wave; read by tools/debug/w1_ts.py through aae_debug_w1_ts)
__device__ unsigned long long w1_ts[4096 * 4 * 8];
#define W1_STAMP(i) do { if (lane == 0 && vblock < 4096) w1_ts[(vblock * 4 + wv) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define W1_STAMP(i) do { } while (0)
#endif
constexpr int kW1HybRows = 16;
constexpr int kW1HybWords = 64;        // the deferred items' bitmap: (round, wave) pairs, 512 rounds
__host__ __device__ inline size_t w1_hybrid_lds_words(int rows) {
    const size_t a = w1_items_lds_words(rows), b = 4 * 4 * kW1HybRows;
    return (a > b ? a : b) + kW1HybWords;
}
__device__ __forceinline__ void w1_item_hybrid_body(const W1Items& a, unsigned* lds, int vblock, int vgrid) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* defer = lds + (w1_hybrid_lds_words(a.rows) - kW1HybWords);
    W1_STAMP(0);
    for (int i = tid; i < kW1HybWords; i += 256) defer[i] = 0u;
    __syncthreads();
    int* rl = reinterpret_cast<int*>(lds) + wv * 4 * kW1HybRows;       // rows as found
    float* xl = reinterpret_cast<float*>(rl + kW1HybRows);              // their values
    int* rs = reinterpret_cast<int*>(xl + kW1HybRows);                  // rows, ascending
    float* xs = reinterpret_cast<float*>(rs + kW1HybRows);              // their values x row scale
    const int cnt = *a.ucount;
    const bool upd = a.gout == nullptr;
    OptScalars s;
    if (upd) s = *a.sc;
    // wave w of workgroup g takes items g + w * vgrid (+ rounds of 4 * vgrid): NEIGHBOURS in the list go to different workgroups.
    // (The list is in arrival order of the distinct-item pass: the items every document holds - the ones with many rows, which
    //  cost their workgroup a pass of the workgroup form each - sit together at its head.)
    const int ustep = vgrid * 4;
    int round = 0;
    W1_STAMP(1);
    constexpr int kRounds = kW1HybWords * 32 / 4;      // rounds the bitmap of deferred items holds
    for (int u = vblock + wv * vgrid; u < cnt && round < kRounds; u += ustep, ++round) {
        const int item = a.ulist[u];
        const int tile = item / kTI, it = item - tile * kTI;
        const int e0 = a.tstart[tile], e1 = a.tstart[tile + 1];
        if (round == 0) W1_STAMP(2);
        const int c4 = 4 * lane;
        const size_t o0 = (size_t)item * a.ldw + min(c4, a.ldw - 4);
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pm = pw, pv = pw;
        if (upd) {
            pw = *reinterpret_cast<const float4*>(a.W + o0);
            if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o0); pv = *reinterpret_cast<const float4*>(a.V + o0); }
        }
        int n = 0;
        for (int base = e0; base < e1; base += 256) {      // (r5: four groups of 64 entries requested together, taken in order)
            int en4[4], r4[4]; float x4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int ec = min(base + 64 * j + lane, e1 - 1); en4[j] = a.en[ec]; r4[j] = a.eb[ec]; x4[j] = a.ev[ec]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = base + 64 * j + lane < e1 && en4[j] == it;
                const unsigned long long bal = __ballot(hit);
                const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
                if (hit && k < kW1HybRows) { rl[k] = r4[j]; xl[k] = x4[j]; }
                n += __popcll(bal);
            }
        }
        if (round == 0) W1_STAMP(3);
        if (n > kW1HybRows) {                           // more rows than one wave adds in the workgroup form: all four take it below
            const int b = round * 4 + wv;
            if (lane == 0) atomicOr(&defer[b >> 5], 1u << (b & 31));
            continue;
        }
        if (lane < n) {                                 // rank by row (distinct rows: the batch's CSR is canonical)
            const int r = rl[lane];
            int k = 0;
            for (int j = 0; j < n; ++j) k += (rl[j] < r || (rl[j] == r && j < lane)) ? 1 : 0;   // total order: a duplicate (row, item) pair of a non-canonical CSR still gets a slot of its own
            rs[k] = r; xs[k] = xl[lane];        // (its row scale is requested with the row itself below: one round trip less)
        }
        for (int cb = 0; cb < a.h; cb += 256) {
            const int cc = min(cb + c4, a.ld - 4);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int kU = 8;
            for (int i0 = 0; i0 < n; i0 += kU) {
                float4 g[kU]; float x[kU];
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    const int i = min(i0 + j, n - 1);
                    const int r = rs[i];
                    x[j] = (i0 + j < n ? xs[i] : 0.f) * a.rscale[r];
                    const float* grow = a.rpb > 0 ? a.ga1 + (size_t)(r / a.rpb) * a.bstride + (size_t)(r % a.rpb) * a.ld
                                                  : a.ga1 + (size_t)r * a.ld;
                    g[j] = *reinterpret_cast<const float4*>(grow + cc);
                }
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    acc.x += x[j] * g[j].x; acc.y += x[j] * g[j].y; acc.z += x[j] * g[j].z; acc.w += x[j] * g[j].w;
                }
            }
            const int c = cb + c4;
            if (c >= a.h) acc.x = 0.f;
            if (c + 1 >= a.h) acc.y = 0.f;
            if (c + 2 >= a.h) acc.z = 0.f;
            if (c + 3 >= a.h) acc.w = 0.f;
            const size_t o = (size_t)item * a.ldw + min(c, a.ldw - 4);
            if (cb > 0 && upd) {
                pw = *reinterpret_cast<const float4*>(a.W + o);
                if (!s.is_sgd) { pm = *reinterpret_cast<const float4*>(a.M + o); pv = *reinterpret_cast<const float4*>(a.V + o); }
            }
            if (c < a.ldw) {
                if (!upd) *reinterpret_cast<float4*>(a.gout + o) = acc;
                else {
                    adam_update(pw.x, pm.x, pv.x, acc.x, s); adam_update(pw.y, pm.y, pv.y, acc.y, s);
                    adam_update(pw.z, pm.z, pv.z, acc.z, s); adam_update(pw.w, pm.w, pv.w, acc.w, s);
                    *reinterpret_cast<float4*>(a.W + o) = pw;
                    if (!s.is_sgd) { *reinterpret_cast<float4*>(a.M + o) = pm; *reinterpret_cast<float4*>(a.V + o) = pv; }
                }
            }
        }
        if (upd && a.mark_synced && lane == 0) a.tsync[item] = (int)*a.step_ctr;
        if (round == 0) W1_STAMP(4);
    }
    W1_STAMP(5);
    __syncthreads();
    W1_STAMP(6);
    // the deferred items, in (round, wave) order, with all four waves
    const int rounds = vblock < cnt ? (cnt - vblock + ustep - 1) / ustep : 0;      // (of wave 0, the one with the most)
    const int nbits = min(rounds, kRounds) * 4;
    for (int b = 0; b < nbits; ++b) {
        if (!((defer[b >> 5] >> (b & 31)) & 1u)) continue;             // (uniform: LDS word read by every thread)
        w1_item_update_one(a, lds, vblock + (b & 3) * vgrid + (b >> 2) * ustep, s, upd);
    }
    //  a launch so narrow that its list takes more rounds than the bitmap holds - the host sizes launches so that none
    // is (abi_chains.h), but no item may be lost to a caller that does not: the rest of the list in the workgroup form, in order
    for (int u = vblock + kRounds * ustep; u < cnt; u += vgrid) w1_item_update_one(a, lds, u, s, upd);
    W1_STAMP(7);
}

// (hot / hot_count: this launch's list; other_count: the list of the NEXT use, zeroed here - the two alternate)
__global__ __launch_bounds__(256) void w1_item_wave_kernel(W1Items a, int* hot, int* hot_count) {
    extern __shared__ unsigned w1w_lds_dyn[];
    w1_item_wave_body(a, hot, hot_count, w1w_lds_dyn, (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), (int)gridDim.x * 4);
}

__global__ __launch_bounds__(256) void w1_item_update_kernel(W1Items a) {
    extern __shared__ unsigned w1_lds_dyn[];
    w1_item_update_body(a, w1_lds_dyn, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace aae
