// The non-GEMM kernels of the AAE step (gfx950).  All are HBM/latency-bound row or element
// kernels: coalesced float4 row segments, one wave (64 lanes) per gathered row, LDS partial sums.
#pragma once
#include "device_common.h"
#include "gemm_f32.h"

namespace aae {

struct BatchView {
    const int64_t* indptr; const int32_t* indices; const float* values;
    const int32_t* rows; int row_start; int n_rows;
    __device__ __forceinline__ int doc(int b) const { return rows ? rows[b] : row_start + b; }
};

constexpr int kLazyReplay = 128;
constexpr int kLazyTabCap = 65536;   // ring of the last 65536 steps' scalars (power of two; a replay reads <= kLazyReplay of them,
                                     // so a learning rate changed by aae_set_lr at any step is honoured)

struct LazyTab { float nss_gen, nss_reg, ibc2, pad; };   // per step t: -lr_gen/bc1, -lr_reg/bc1, 1/sqrt(bc2)

// The step-opening bookkeeping (its own launch: advance_step_kernel at the end of this file; or the first 4 threads of
// one workgroup of the step's first gather when nothing in between needs it - a launch costs ~4.5 us however little it does)
struct AdvanceJob { OptScalars* sc; long long* ctr; LazyTab* tab; float* losses; int enabled;
                    int* stamp2 = nullptr; int* ucount2 = nullptr; };   // != NULL: new stamp + empty list for the NEXT batch's distinct-item pass (bump_stamp_kernel's work)
__device__ __forceinline__ void advance_step_body(OptScalars* sc, long long* ctr, LazyTab* tab, int* stamp, int* ucount,
                                                  float* losses, int i) {
    // (stamp == NULL: the step's unique-item list was built ahead of time, aae_prefetch_batch)
    if (i == 0) { *ctr += 1; if (stamp) { *stamp += 1; *ucount = 0; } losses[1] = 0.f; losses[2] = 0.f; }
    if (i >= 4) return;
    OptScalars s = sc[i];
    s.t += 1;
    if (s.is_sgd) { s.neg_step_size = (float)(-s.lr); s.bc2_sqrt = 1.f; s.inv_bc2_sqrt = 1.f; }
    else {
        double bc1 = 1.0 - pow(0.9, (double)s.t), bc2 = 1.0 - pow(0.999, (double)s.t);
        s.neg_step_size = (float)(-(s.lr / bc1));
        s.bc2_sqrt = (float)sqrt(bc2);
        s.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    sc[i] = s;
    if (tab) {                             // optimiser 0 = enc_optim (gen_lr), 2 = gen_optim (reg_lr)
        const int slot = (int)(s.t & (kLazyTabCap - 1));
        if (i == 0) { tab[slot].nss_gen = s.neg_step_size; tab[slot].ibc2 = s.inv_bc2_sqrt; }
        if (i == 2) tab[slot].nss_reg = s.neg_step_size;
        // ... and the entry of the step AFTER this one, one step early (the same expressions at t + 1: the next call writes the
        // same bits again): the next batch's catch-up may then start before that step's own bookkeeping has run
        // (launch_prefetch, early).  A learning rate changed from the host in between makes it stale: the host falls back.
        float nss1 = (float)(-s.lr), ib1 = 1.f;
        if (!s.is_sgd) {
            const double b1n = 1.0 - pow(0.9, (double)(s.t + 1)), b2n = 1.0 - pow(0.999, (double)(s.t + 1));
            nss1 = (float)(-(s.lr / b1n)); ib1 = (float)(1.0 / sqrt(b2n));
        }
        const int slot1 = (int)((s.t + 1) & (kLazyTabCap - 1));
        if (i == 0) { tab[slot1].nss_gen = nss1; tab[slot1].ibc2 = ib1; }
        if (i == 2) tab[slot1].nss_reg = nss1;
    }
}

// ---------------------------------------------------------------------------------------
// K2+K3 (+K4): sparse multi-hot -> first encoder Linear.  reference: F.normalize(inp, 1) +
// enc.lin1 (+ drop1, act1), aae.py:132-137.
//   a1[b,:] = b1 + sum_{e in row b} (v_e * s_b) * W1T[idx_e, :],  s_b = 1/max(sum|v|, 1e-12)
// One workgroup per document; its 16 waves take every 16th entry (a typical document is one
// dependent HBM round trip deep), each lane a float4 of the 800-byte (h=200) weight row, partial
// sums meet in LDS.
// ---------------------------------------------------------------------------------------
template <int NW>      // waves per document: 16, or 4 for wide batches (late r4: eight documents per CU instead of two)
__global__ __launch_bounds__(64 * NW) void enc_gather_kernel_t(BatchView bv, const float* __restrict__ W1T, int ldw,
                                                         const float* __restrict__ b1, int h, int normalize,
                                                         float* __restrict__ a1, float* __restrict__ y, int ld,
                                                         int act, DropSpec d, uint64_t seed, const long long* step_ctr,
                                                         float* __restrict__ rscale, const float* __restrict__ doc_l1 = nullptr,
                                                         AdvanceJob adv = AdvanceJob{nullptr, nullptr, nullptr, nullptr, 0},
                                                         long long step_val = -1) {
    // adv.enabled: the last workgroup also opens the step (advance_step_body); every workgroup then takes the step number
    // from step_val (the host's count of opened steps = what *step_ctr holds once the step is open), not from memory
    // doc_l1 != NULL: the L1 norms of the COMPLETE documents, indexed by document (the batch holds only the columns of one
    // item slice of them, aae_set_doc_l1); b1 == NULL: no bias (a partial sum that meets the other slices' elsewhere)
    extern __shared__ __attribute__((aligned(16))) float part[];   // [NW][hp]
    __shared__ float red[NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hp = (h + 3) & ~3;
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    if constexpr (NW >= 16) {
        // (r6) The FIRST pass's ids, values and weight rows are requested before the L1 norm is reduced: the scale is needed by the
        // multiply-adds only, so the chain of dependent round trips is indptr -> {ids -> rows | values -> norm} -> sum: three deep where
        // r1-r5 walked five (indptr -> values -> norm -> id -> row, and the 16-wave form one (id, row) pair per entry after the other:
        // a document of 24 entries was two more).  The 16-wave form only: the 4-wave form of wide batches keeps r5's loop - eight entries per
        // pass - where this one's extra registers (130) cost it a third of its rate (C4: 15.8 -> 20.9 us per launch).  Same products, added
        // in the same order: same bits.
        constexpr int EP = 4;       // (16 waves x 4 = 64 entries in the first pass)
        const int c00 = lane * 4;
        const bool col = c00 < hp;
        int idx0[EP]; float x0[EP]; float4 w0[EP];
        {
            const int64_t e0 = lo + wave;
    #pragma unroll
            for (int j = 0; j < EP; ++j) {
                const int64_t e = e0 + (int64_t)NW * j;
                const int64_t ec = e < hi ? e : (e0 < hi ? e0 : (hi > lo ? lo : 0));
                idx0[j] = (hi > lo) ? bv.indices[ec] : 0;
                x0[j] = e < hi ? bv.values[ec] : 0.f;
            }
    #pragma unroll
            for (int j = 0; j < EP; ++j) w0[j] = col ? *reinterpret_cast<const float4*>(W1T + (size_t)idx0[j] * ldw + c00) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 1.f;
        if (normalize && doc_l1) {
            s = 1.f / fmaxf(doc_l1[dc], 1e-12f);
        } else if (normalize) {
            float l1 = 0.f;
            for (int64_t e = lo + tid; e < hi; e += 64 * NW) l1 += fabsf(bv.values[e]);
            l1 = wave_sum(l1);
            if (lane == 0) red[wave] = l1;
            __syncthreads();
            l1 = 0.f;
    #pragma unroll
            for (int w = 0; w < NW; ++w) l1 += red[w];
            s = 1.f / fmaxf(l1, 1e-12f);
        }
        if (tid == 0) rscale[b] = s;
        for (int c0 = c00; c0 < hp; c0 += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t e0 = lo + wave; e0 < hi; e0 += EP * NW) {
                const bool first = e0 == lo + wave && c0 == c00;
                int idx[EP]; float x[EP]; float4 w[EP];
                if (first) {
    #pragma unroll
                    for (int j = 0; j < EP; ++j) { x[j] = x0[j]; w[j] = w0[j]; }
                } else {
    #pragma unroll
                    for (int j = 0; j < EP; ++j) {
                        const int64_t e = e0 + (int64_t)NW * j;
                        const int64_t ec = e < hi ? e : e0;
                        idx[j] = bv.indices[ec];
                        x[j] = e < hi ? bv.values[ec] : 0.f;
                    }
    #pragma unroll
                    for (int j = 0; j < EP; ++j) w[j] = *reinterpret_cast<const float4*>(W1T + (size_t)idx[j] * ldw + c0);
                }
    #pragma unroll
                for (int j = 0; j < EP; ++j)
                    if (e0 + (int64_t)NW * j < hi) {
                        float xv = x[j];
                        if (normalize) xv *= s;
                        acc.x += xv * w[j].x; acc.y += xv * w[j].y; acc.z += xv * w[j].z; acc.w += xv * w[j].w;
                    }
            }
            *reinterpret_cast<float4*>(&part[wave * hp + c0]) = acc;
        }
    } else {
        float s = 1.f;
        if (normalize && doc_l1) {
            s = 1.f / fmaxf(doc_l1[dc], 1e-12f);
        } else if (normalize) {
            float l1 = 0.f;
            for (int64_t e = lo + tid; e < hi; e += 64 * NW) l1 += fabsf(bv.values[e]);
            l1 = wave_sum(l1);
            if (lane == 0) red[wave] = l1;
            __syncthreads();
            l1 = 0.f;
    #pragma unroll
            for (int w = 0; w < NW; ++w) l1 += red[w];
            s = 1.f / fmaxf(l1, 1e-12f);
        }
        if (tid == 0) rscale[b] = s;
        for (int c0 = lane * 4; c0 < hp; c0 += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NW < 16) {
                // r5, the four-wave form (wide batches: a wave owns ~5 of a document's ~20 entries): eight entries per pass - their
                // ids and values in one round trip, their weight rows in a second - where the loop below is two dependent round
                // trips PER entry (17 us per launch at 1 000 rows x 20 entries).  Same products, added in the same order.
                for (int64_t e0 = lo + wave; e0 < hi; e0 += 8 * NW) {
                    int idx[8]; float x[8];
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int64_t e = e0 + (int64_t)NW * j;
                        const int64_t ec = e < hi ? e : e0;
                        idx[j] = bv.indices[ec];
                        x[j] = e < hi ? bv.values[ec] : 0.f;
                    }
                    float4 w[8];
    #pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const float4*>(W1T + (size_t)idx[j] * ldw + c0);
    #pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (e0 + (int64_t)NW * j < hi) {
                            float xv = x[j];
                            if (normalize) xv *= s;
                            acc.x += xv * w[j].x; acc.y += xv * w[j].y; acc.z += xv * w[j].z; acc.w += xv * w[j].w;
                        }
                }
            } else
            for (int64_t e = lo + wave; e < hi; e += NW) {
                const int idx = bv.indices[e];
                float x = bv.values[e];
                if (normalize) x *= s;
                const float4 w = *reinterpret_cast<const float4*>(W1T + (size_t)idx * ldw + c0);
                acc.x += x * w.x; acc.y += x * w.y; acc.z += x * w.z; acc.w += x * w.w;
            }
            *reinterpret_cast<float4*>(&part[wave * hp + c0]) = acc;
        }
    }
    __syncthreads();
    if (adv.enabled && blockIdx.x == gridDim.x - 1 && tid < 64) advance_step_body(adv.sc, adv.ctr, adv.tab, nullptr, nullptr, adv.losses, tid);
    if (adv.stamp2 && blockIdx.x == 0 && tid == 0) { *adv.stamp2 += 1; *adv.ucount2 = 0; }
    const uint64_t key = d.device_rng ? rng_key(seed, (uint64_t)(step_val >= 0 ? step_val : *step_ctr), 0) : 0;
    for (int c = tid; c < h; c += 64 * NW) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += part[w * hp + c];
        if (b1) v += b1[c];
        a1[(size_t)b * ld + c] = v;
        if (y) {
            if (d.enabled) v = drop_fwd(d, drop_keep(d, key, b, c), v);
            y[(size_t)b * ld + c] = act_fwd(act, v);
        }
    }
}

#define enc_gather_kernel enc_gather_kernel_t<16>

// y = act(dropout(a))  elementwise over [rows][h]  (gen_step re-uses disc_step's a1)
__global__ void drop_act_kernel(const float* __restrict__ a, float* __restrict__ y, int rows, int h, int ld,
                                int act, DropSpec d, uint64_t seed, const long long* step_ctr) {
    const uint64_t key = d.device_rng ? rng_key(seed, (uint64_t)*step_ctr, 0) : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * h; i += gridDim.x * blockDim.x) {
        int r = i / h, c = i - r * h;
        float v = a[(size_t)r * ld + c];
        if (d.enabled) v = drop_fwd(d, drop_keep(d, key, r, c), v);
        y[(size_t)r * ld + c] = act_fwd(act, v);
    }
}

// bias gradient of the first encoder layer for 64 columns, 256 threads (4 waves stride the rows), then its
// optimiser update or export: the body of colsum_adam_kernel for use inside the grouped update launch
__device__ __forceinline__ void colsum_adam_body(const float* __restrict__ ga, int rows, int h, int ld, float* p,
                                                 float* m, float* v, float* gout, const OptScalars* sc, int cblock,
                                                 float* red /* [4][64] LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = cblock * 64 + lane;
    float g = 0.f;
    if (c < h) {
        // rows wave, wave + 4, ... added in ascending order (the order the r1-r3 loop used: results bit for bit the same), 16
        // loads in flight at a time: one load per add was one memory round trip per row - 200 of them in a row for a batch
        // of 800 rows, 40 of the 47 us the weight-gradient launch took there
        // (r5: 32 at a time from 256 rows on - 7 us of a 12 us launch at 800 rows were this loop)
        int r0 = wave;
        if (rows >= 256)
            for (; r0 + 4 * 31 < rows; r0 += 128) {
                float t[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) t[i] = ga[(size_t)(r0 + 4 * i) * ld + c];
#pragma unroll
                for (int i = 0; i < 32; ++i) g += t[i];
            }
        for (; r0 < rows; r0 += 64) {
            float t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = ga[(size_t)min(r0 + 4 * i, rows - 1) * ld + c];
#pragma unroll
            for (int i = 0; i < 16; ++i) g += (r0 + 4 * i < rows) ? t[i] : 0.f;
        }
    }
    red[wave * 64 + lane] = g;
    __syncthreads();
    if (wave != 0 || c >= h) return;
    g = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    if (gout) { gout[c] = g; return; }
    const OptScalars s = *sc;
    float pp = p[c], mm = s.is_sgd ? 0.f : m[c], vv = s.is_sgd ? 0.f : v[c];
    adam_update(pp, mm, vv, g, s);
    p[c] = pp;
    if (!s.is_sgd) { m[c] = mm; v[c] = vv; }
}

// dense torch.optim.Adam/SGD over a flat tensor with a materialised gradient (K10)
__global__ void adam_dense_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                  float* __restrict__ g, size_t n4, const OptScalars* sc, int zero_g) {
    const OptScalars s = *sc;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = make_float4(0, 0, 0, 0), vv = mm;
        if (!s.is_sgd) { mm = reinterpret_cast<float4*>(m)[i]; vv = reinterpret_cast<float4*>(v)[i]; }
        adam_update(pp.x, mm.x, vv.x, gg.x, s); adam_update(pp.y, mm.y, vv.y, gg.y, s);
        adam_update(pp.z, mm.z, vv.z, gg.z, s); adam_update(pp.w, mm.w, vv.w, gg.w, s);
        reinterpret_cast<float4*>(p)[i] = pp;
        if (!s.is_sgd) { reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv; }
        if (zero_g) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// The same update for up to 8 small tensors of one optimiser in ONE launch (data-parallel replicas apply the
// all-reduced gradients of every hidden layer after each phase: twelve ~4 us launches per step otherwise).
struct AdamJob { float* p; float* m; float* v; float* g; unsigned n4; unsigned blk0;       // blk0: first block of the job
                 W4Copies w4; int ld;                  // w4.f4 != NULL: the k4-interleaved copies of p kept in step (p rows are ld wide)
                 const OptScalars* sc;                 // != NULL: this job's optimiser (a launch that serves two of them)
                 int npeers; long long pstride; };     // npeers > 0: g = the first of npeers gathered copies, pstride floats apart:
                                                       // the gradient is their sum in peer order (identical on every rank)
struct AdamGroup { int njobs; AdamJob jobs[8]; };

__global__ __launch_bounds__(256) void adam_group_kernel(AdamGroup grp, const OptScalars* sc) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < grp.njobs && blockIdx.x >= grp.jobs[i].blk0) j = i;
    const AdamJob job = grp.jobs[j];
    const OptScalars s = job.sc ? *job.sc : *sc;
    const unsigned i = (blockIdx.x - job.blk0) * 256u + threadIdx.x;
    if (i >= job.n4) return;
    float4 pp = reinterpret_cast<float4*>(job.p)[i];
    float4 gg = reinterpret_cast<const float4*>(job.g)[i];
    for (int q = 1; q < job.npeers; ++q) {
        const float4 t = reinterpret_cast<const float4*>(job.g + (size_t)q * job.pstride)[i];
        gg.x += t.x; gg.y += t.y; gg.z += t.z; gg.w += t.w;
    }
    float4 mm = make_float4(0, 0, 0, 0), vv = mm;
    if (!s.is_sgd) { mm = reinterpret_cast<float4*>(job.m)[i]; vv = reinterpret_cast<float4*>(job.v)[i]; }
    adam_update(pp.x, mm.x, vv.x, gg.x, s); adam_update(pp.y, mm.y, vv.y, gg.y, s);
    adam_update(pp.z, mm.z, vv.z, gg.z, s); adam_update(pp.w, mm.w, vv.w, gg.w, s);
    reinterpret_cast<float4*>(job.p)[i] = pp;
    if (!s.is_sgd) { reinterpret_cast<float4*>(job.m)[i] = mm; reinterpret_cast<float4*>(job.v)[i] = vv; }
    if (job.w4.f4) {
        const unsigned row = (i * 4u) / (unsigned)job.ld, col = (i * 4u) % (unsigned)job.ld;     // ld is a multiple of 4
        if ((int)col + 3 < job.w4.N) w4_put4(job.w4, (int)row, (int)col, pp);
        else
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((int)col + k < job.w4.N) w4_put1(job.w4, (int)row, (int)col + k, (&pp.x)[k]);
    }
}

// both k4-interleaved copies of w [c.M][c.N] (row stride ld) from scratch: after a writer that is not an optimiser epilogue
__global__ __launch_bounds__(256) void interleave4_kernel(const float* __restrict__ w, int ld, W4Copies c) {
    const int per_row = (c.N + 3) >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < c.M * per_row; i += gridDim.x * 256) {
        const int o = i / per_row, c0 = (i - o * per_row) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < c.N) w4_put1(c, o, c0 + k, w[(size_t)o * ld + c0 + k]);
    }
}

// bias gradient of the first encoder layer: db1[c] = sum_b ga1[b][c], then its optimiser
// update (or export).  Block = 64 columns x 16 waves striding the rows, LDS tree over waves.
__global__ __launch_bounds__(1024) void colsum_adam_kernel(const float* __restrict__ ga, int rows, int h, int ld,
                                                           float* p, float* m, float* v, float* gout,
                                                           const OptScalars* sc) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float g = 0.f;
    if (c < h)
        for (int r = wave; r < rows; r += 16) g += ga[(size_t)r * ld + c];
    red[wave][lane] = g;
    __syncthreads();
    if (wave != 0 || c >= h) return;
    g = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) g += red[w][lane];
    if (gout) { gout[c] = g; return; }
    const OptScalars s = *sc;
    float pp = p[c], mm = s.is_sgd ? 0.f : m[c], vv = s.is_sgd ? 0.f : v[c];
    adam_update(pp, mm, vv, g, s);
    p[c] = pp;
    if (!s.is_sgd) { m[c] = mm; v[c] = vv; }
}

// ---------------------------------------------------------------------------------------
// BCE fix-up for the non-zero targets (the GEMM epilogue assumed target 0 everywhere):
// one wave per CSR entry recomputes that logit as a (h+1)-long dot product, rewrites dL/dlogit
// with the reference's exact formula and emits the loss difference.  grid (docs, chunks).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_fixup_kernel(BatchView bv, const float* __restrict__ hdec, int ldh,
                                                        const float* __restrict__ V3a, int ldv, int kdim,
                                                        float* __restrict__ G, int ldg, float gscale,
                                                        float* __restrict__ partials) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    float dl = 0.f;
    for (int64_t e0 = lo + 16 * (int64_t)blockIdx.y; e0 < hi; e0 += 16 * (int64_t)gridDim.y) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t e = e0 + 4 * j + wave;
            if (e >= hi) break;
            const int n = bv.indices[e];
            float acc = 0.f;
            for (int k = lane; k < kdim; k += 64) acc += hdec[(size_t)b * ldh + k] * V3a[(size_t)n * ldv + k];
            acc = wave_sum(acc);
            if (lane == 0) {
                float g0, l0, g1, l1;
                bce_elem_t0(acc, gscale, g0, l0);
                bce_elem(acc, bv.values[e], gscale, g1, l1);
                G[(size_t)b * ldg + n] = g1;
                dl += l1 - l0;
            }
        }
    }
    if (lane == 0) red[wave] = dl;
    __syncthreads();
    if (tid == 0) partials[b * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

// first stage of a wide slab reduction: out[g][i] = sum of the slabs z = g, g+G, g+2G, ... of element i
// (float4 per thread, the loads of one thread are independent and stay in flight together)
// The last workgroup also folds the fused decoder's per-workgroup loss partials into losses[slot]
// (fixed order, double accumulation: what loss_finalize_kernel does as a launch of its own).
__global__ __launch_bounds__(256) void slab_partial_kernel(const float* __restrict__ slabs, int nslab,
                                                           size_t slab_stride, size_t n4,
                                                           float* __restrict__ out, size_t out_stride,
                                                           const float* __restrict__ loss_partials, int n_partials,
                                                           float loss_scale, float* losses, int slot) {
    const int g = blockIdx.y, G = gridDim.y;
    if (loss_partials && blockIdx.x == gridDim.x - 1 && g == G - 1) {
        __shared__ double red[256];
        double acc = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 256) acc += (double)loss_partials[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) losses[slot] = (float)(red[0] * (double)loss_scale);
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int z = g; z < nslab; z += G) {
            const float4 v = reinterpret_cast<const float4*>(slabs + (size_t)z * slab_stride)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(out + (size_t)g * out_stride)[i] = acc;
    }
}

// out = (sum_z slab[z]) * act'(y) * dropout_scale      (consumer of the split-K dA2 GEMM)
__global__ void slab_reduce_actbwd_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                          int rows, int h, int ld, const float* __restrict__ y, int ldy,
                                          float* __restrict__ out, int act, DropSpec d, uint64_t seed,
                                          const long long* step_ctr) {
    const uint64_t key = d.device_rng ? rng_key(seed, (uint64_t)*step_ctr, 0) : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * h; i += gridDim.x * blockDim.x) {
        int r = i / h, c = i - r * h;
        float acc = 0.f;
        for (int z = 0; z < nslab; ++z) acc += slabs[(size_t)z * slab_stride + (size_t)r * ld + c];
        acc *= act_grad_from_y<true>(act, y[(size_t)r * ldy + c]);
        if (d.enabled) acc *= drop_bwd_mul(d, drop_keep(d, key, r, c));
        out[(size_t)r * ld + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// DenoisingAutoEncoder(corrupt='gauss'), reference dae.py:40-45, 191: the encoder reads the DENSE batch + noise on all
// N columns, so its first layer is a dense [B, N] x [N, h] product (and its weight gradient a dense [N, B] x [B, h]
// one with an eager Adam over every row) instead of the sparse gather / scatter.
//   dense_input_kernel: x[b][:] = noise[b][:] + the batch's row b, L1-normalised over ALL columns (F.normalize(x, 1),
//                       aae.py:132-133) - one workgroup per document;
//   slab_reduce_fwd_kernel: a1 = sum of the split-K slabs of x * W1T + b1, y = act(dropout(a1)) - the dense
//                       counterpart of enc_gather_kernel's epilogue;
//   fill_tsync_kernel:  every row of W1T has received this step's update (the dense Adam touched all of them).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dense_input_kernel(BatchView bv, const float* __restrict__ noise, int64_t ld_noise,
                                                          int N, int normalize, float* __restrict__ X, int ldx) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* row = X + (size_t)b * ldx;
    const float* nz = noise + (size_t)b * ld_noise;
    for (int n = tid; n < ldx; n += 1024) row[n] = n < N ? nz[n] : 0.f;
    __syncthreads();
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    for (int64_t e = lo + tid; e < hi; e += 1024) row[bv.indices[e]] += bv.values[e];      // (column ids are unique within a row)
    __syncthreads();
    if (!normalize) return;
    float acc = 0.f;
    for (int n = tid; n < N; n += 1024) acc += fabsf(row[n]);
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    float l1 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) l1 += red[w];
    const float sc = 1.f / fmaxf(l1, 1e-12f);
    for (int n = tid; n < N; n += 1024) row[n] *= sc;
}

__global__ void slab_reduce_fwd_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, int rows, int h,
                                       int ld, const float* __restrict__ b1, float* __restrict__ a1, float* __restrict__ y,
                                       int act, DropSpec d, uint64_t seed, const long long* step_ctr) {
    const uint64_t key = d.device_rng ? rng_key(seed, (uint64_t)*step_ctr, 0) : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * h; i += gridDim.x * blockDim.x) {
        const int r = i / h, c = i - r * h;
        float v = 0.f;
        for (int z = 0; z < nslab; ++z) v += slabs[(size_t)z * slab_stride + (size_t)r * ld + c];
        v += b1[c];
        a1[(size_t)r * ld + c] = v;
        if (d.enabled) v = drop_fwd(d, drop_keep(d, key, r, c), v);
        y[(size_t)r * ld + c] = act_fwd(act, v);
    }
}

__global__ void fill_tsync_kernel(int* __restrict__ tsync, int n, const long long* step_ctr) {
    const int t = (int)*step_ctr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) tsync[i] = t;
}

// losses[slot] = scale * sum(partials[0..n))  in a fixed order (deterministic), one workgroup
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ pa, int na,
                                                            const float* __restrict__ pb, int nb, float scale,
                                                            float* losses, int slot) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < na; i += 256) acc += (double)pa[i];
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)pb[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) losses[slot] = (float)(red[0] * (double)scale);
}

// ---------------------------------------------------------------------------------------
// encoder output activation (PRIOR_ACTIVATIONS, aae.py:97-101): rows of width c, in place.
// One wave per row.  fwd: z = softmax(a) | sigmoid(a) | a.   bwd: ga from gz and z.
// ---------------------------------------------------------------------------------------
__global__ void final_act_fwd_kernel(float* __restrict__ z, int rows, int c, int ld, int kind,
                                     float* __restrict__ copy_out, int ld_copy) {
    int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* zr = z + (size_t)r * ld;
    if (kind == 1) {
        float mx = -INFINITY;
        for (int j = lane; j < c; j += 64) mx = fmaxf(mx, zr[j]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < c; j += 64) sum += expf(zr[j] - mx);
        sum = wave_sum(sum);
        for (int j = lane; j < c; j += 64) zr[j] = expf(zr[j] - mx) / sum;
    } else if (kind == 2) {
        for (int j = lane; j < c; j += 64) zr[j] = sigmoidf_(zr[j]);
    }
    if (copy_out)
        for (int j = lane; j < c; j += 64) copy_out[(size_t)r * ld_copy + j] = zr[j];
}

__global__ void final_act_bwd_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ gz, int ldg,
                                     float* __restrict__ ga, int lda, int rows, int c, int kind) {
    int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* zr = z + (size_t)r * ldz;
    const float* gr = gz + (size_t)r * ldg;
    float* o = ga + (size_t)r * lda;
    if (kind == 1) {
        float dot = 0.f;
        for (int j = lane; j < c; j += 64) dot += gr[j] * zr[j];
        dot = wave_sum(dot);
        for (int j = lane; j < c; j += 64) o[j] = zr[j] * (gr[j] - dot);
    } else if (kind == 2) {
        for (int j = lane; j < c; j += 64) o[j] = gr[j] * zr[j] * (1.f - zr[j]);
    } else {
        for (int j = lane; j < c; j += 64) o[j] = gr[j];
    }
}

// strided 2-D copy: dst[r][c] = src[r][c] * scale
__global__ void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows,
                              int cols, float scale) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c] * scale;
    }
}

__global__ void fill_col_kernel(float* dst, int ld, int rows, int col, float v) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) dst[(size_t)r * ld + col] = v;
}

// z_real ~ prior (PRIOR_SAMPLERS, aae.py:78-94) from the counter generator, times prior_scale
__global__ void prior_kernel(float* __restrict__ z, int ld, int rows, int c, int prior, float scale, uint64_t seed,
                             const long long* step_ctr, int grow0) {
    const uint64_t key = rng_key(seed, (uint64_t)*step_ctr, 100);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * c; i += gridDim.x * blockDim.x) {
        int r = i / c, j = i - r * c;
        const uint64_t gr = (uint64_t)(r + grow0);       // keyed by the row of the global batch
        float v = 0.f;
        if (prior == 0) {   // gauss: Box-Muller
            uint32_t u1 = hash_u32(key, (gr << 32) | (uint32_t)(2 * j));
            uint32_t u2 = hash_u32(key, (gr << 32) | (uint32_t)(2 * j + 1));
            float f1 = ((float)u1 + 1.f) * 2.3283064365386963e-10f;   // (0,1]
            float f2 = (float)u2 * 2.3283064365386963e-10f;
            v = sqrtf(-2.f * logf(f1)) * cosf(6.283185307179586f * f2);
        } else if (prior == 1) {   // categorical: one-hot of a uniform class per row
            uint32_t u = hash_u32(key, (gr << 32) | 0xFFFFFFFFu);
            v = ((int)(u % (uint32_t)c) == j) ? 1.f : 0.f;
        }                     // bernoulli: the reference's randint(0,1) is always 0 (aae.py:86-88)
        z[(size_t)r * ld + j] = v * scale;
    }
}

// ---------------------------------------------------------------------------------------
// adversarial losses on the discriminator outputs (aae.py:726-727, 738) and their gradients
// w.r.t. the discriminator's last pre-activation.  Single workgroup, B <= a few thousand.
//   mode 0 (disc_step): rows [0,B) real, [B,2B) fake
//       D = -mean(log(dr+T) + log(1-df+T));  g_r = -1/B/(dr+T)*dr(1-dr);  g_f = +1/B/(1-df+T)*df(1-df)
//   mode 1 (gen_step): rows [0,B) fake
//       G = -mean(log(df+T));                g   = -1/B/(df+T)*df(1-df)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adv_loss_kernel(const float* __restrict__ dout, int ldo, int B, int mode,
                                                       float gscale, float* __restrict__ ga, int lda,
                                                       float* losses, int slot) {
    __shared__ double red[256];
    double acc = 0.0;
    const float invB = 1.f / (float)B;
    const int rows = mode == 0 ? 2 * B : B;
    for (int r = threadIdx.x; r < rows; r += 256) {
        float dv = dout[(size_t)r * ldo];
        float g, l;
        if (mode == 0 && r >= B) { l = logf(1.f - dv + kTiny); g = invB / (1.f - dv + kTiny); }
        else { l = logf(dv + kTiny); g = -invB / (dv + kTiny); }
        ga[(size_t)r * lda] = g * dv * (1.f - dv) * gscale;
        acc += (double)l;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) losses[slot] = (float)(-red[0] / (double)B);
}

// ---------------------------------------------------------------------------------------
// Deferred ("lazy") Adam on the encoder's first layer W1T [N][h].
//
// torch.optim.Adam touches every row of enc.lin1.weight in every enc_optim.step() and
// gen_optim.step() (aae.py:706,742) although only the rows of the items present in the batch
// have a gradient: for all other rows g = 0 and the update
//     m <- m + 0.1(0 - m);  v <- 0.999 v;  p <- p - step_size_t * m / (sqrt(v)/bc2_t + eps)
// depends on nothing but the row's own (p, m, v) and the step number.  Rows are independent, so
// those zero-gradient updates are postponed until a row is read again (gathered by a batch, or
// exported): a row carries `tsync` = the last step whose two updates (enc_optim then gen_optim)
// it has received, and w1_catchup replays steps tsync+1 .. upto with the SAME fp32 operations
// in the SAME order the dense kernel would have used.  Since m shrinks by 0.9 per replayed step
// while sqrt(v) shrinks by 0.9995, the parameter increments die out geometrically: after
// kLazyReplay steps their sum is < 32 * lr * 0.9005^kLazyReplay (|m| / sqrt(v) <= (1 - b1) / sqrt(1 - b2) = 3.2 and the
// ratio shrinks by 0.9 / 0.9995 per step: ~5e-8 for lr = 1e-3 at 128 steps, below one ulp of any weight that matters and
// 200x below the 1e-5 parity tolerance; r1 replayed 192 steps, 5e-11), so the replay stops there and the rest of the
// gap only decays m and v in closed form.  The replay is one dependent chain per element (two rsqrt-class operations per
// step): its length IS the kernel's time (14 us at 192, the step's first kernel after the unique-item list).  HBM traffic per step drops from 2 * 28 B * N * h to the
// touched rows.
// ---------------------------------------------------------------------------------------
// (kLazyReplay / kLazyTabCap / LazyTab: defined at the top of this file, next to advance_step_body)

// distinct items of the batch -> ulist (order arbitrary), *ucount
__global__ __launch_bounds__(256) void uniq_items_kernel(BatchView bv, int* __restrict__ mark,
                                                         const int* __restrict__ stamp_p, int* __restrict__ ulist,
                                                         int* __restrict__ ucount) {
    const int b = blockIdx.x;
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    const int stamp = *stamp_p;
    for (int64_t e = lo + (int64_t)blockIdx.y * 256 + threadIdx.x; e < hi; e += (int64_t)gridDim.y * 256) {
        const int idx = bv.indices[e];
        if (atomicExch(&mark[idx], stamp) != stamp) ulist[atomicAdd(ucount, 1)] = idx;
    }
}

// stab: the per-step scalars of steps t0+1 .. t0+nl, staged in LDS by the caller (a chain of dependent scalar
// loads from the global table cost ~50 ns per replayed step - the whole kernel's time at 192 steps)
__device__ __forceinline__ void lazy_replay(float& p, float& m1, float& v1, float& m3, float& v3, int t0, int upto,
                                            const LazyTab* stab) {
    const int n = upto - t0;
    const int nl = n < kLazyReplay ? n : kLazyReplay;
    for (int j = 1; j <= nl; ++j) {
        const LazyTab T = stab[j - 1];
        m1 = m1 + 0.1f * (0.f - m1);
        v1 = v1 * 0.999f;
        p = p + (T.nss_gen * m1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v1) * T.ibc2 + 1e-8f);
        m3 = m3 + 0.1f * (0.f - m3);
        v3 = v3 * 0.999f;
        p = p + (T.nss_reg * m3) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v3) * T.ibc2 + 1e-8f);
    }
    const int rest = n - nl;
    if (rest > 0) {
        const float dm = exp2f((float)rest * -0.15200309344504997f);    // 0.9^rest
        const float dv = exp2f((float)rest * -0.0014434168696687186f);  // 0.999^rest
        m1 *= dm; m3 *= dm; v1 *= dv; v3 *= dv;
    }
}

// rows = ulist[0..*ucount) (or every row when ulist == NULL); one workgroup per row.
// upto = *step_ctr + upto_off  (training gather: -1 = through the previous step; export/predict: 0)
__global__ __launch_bounds__(256) void w1_catchup_kernel(const int* __restrict__ ulist, const int* __restrict__ ucount,
                                                         int n_rows_all, int* __restrict__ tsync,
                                                         float* __restrict__ W, float* __restrict__ M1,
                                                         float* __restrict__ V1, float* __restrict__ M3,
                                                         float* __restrict__ V3, int ld, int h,
                                                         const LazyTab* __restrict__ tab, const long long* step_ctr,
                                                         int upto_off, const int* __restrict__ skip_mark = nullptr,
                                                         const int* __restrict__ skip_stamp = nullptr, long long upto_abs = -1) {
    // (upto_abs >= 0: the step number from the host - the early prefetch runs beside the launch that advances *step_ctr)
    __shared__ LazyTab stab[kLazyReplay];
    const int upto = upto_abs >= 0 ? (int)upto_abs : (int)*step_ctr + upto_off;
    const int cnt = ulist ? *ucount : n_rows_all;
    // skip_mark (the prefetch of the NEXT batch's rows while a step runs): rows of the RUNNING batch are left alone -
    // the step's own two updates bring them to `upto`
    const int skip = skip_mark ? *skip_stamp : 0;
    for (int r = blockIdx.x; r < cnt; r += gridDim.x) {
        const int row = ulist ? ulist[r] : r;
        const int t0 = (skip_mark && skip_mark[row] == skip) ? upto : tsync[row];
        if (t0 < upto) {
            const int nl = min(upto - t0, kLazyReplay);
            for (int j = threadIdx.x; j < nl; j += 256) stab[j] = tab[(t0 + 1 + j) & (kLazyTabCap - 1)];
            __syncthreads();
            for (int c = threadIdx.x; c < h; c += 256) {
                const size_t o = (size_t)row * ld + c;
                float p = W[o], m1 = M1[o], v1 = V1[o], m3 = M3[o], v3 = V3[o];
                lazy_replay(p, m1, v1, m3, v3, t0, upto, stab);
                W[o] = p; M1[o] = m1; V1[o] = v1; M3[o] = m3; V3[o] = v3;
            }
        }
        __syncthreads();                       // every thread has read tsync[row]
        if (threadIdx.x == 0 && t0 < upto) tsync[row] = upto;
    }
}

// optimiser step on the touched rows with the scattered gradient; clears the gradient rows.
// mark_synced: this was gen_optim (the second update of the step) -> tsync[row] = current step
__global__ __launch_bounds__(256) void w1_sparse_adam_kernel(const int* __restrict__ ulist,
                                                             const int* __restrict__ ucount, float* __restrict__ W,
                                                             float* __restrict__ M, float* __restrict__ V,
                                                             float* __restrict__ G, int ld, int h, const OptScalars* sc,
                                                             int* __restrict__ tsync, const long long* step_ctr,
                                                             int mark_synced) {
    const OptScalars s = *sc;
    const int cnt = *ucount;
    for (int r = blockIdx.x; r < cnt; r += gridDim.x) {
        const int row = ulist[r];
        for (int c = threadIdx.x; c < h; c += 256) {
            const size_t o = (size_t)row * ld + c;
            float p = W[o], m = s.is_sgd ? 0.f : M[o], v = s.is_sgd ? 0.f : V[o];
            adam_update(p, m, v, G[o], s);
            W[o] = p; G[o] = 0.f;
            if (!s.is_sgd) { M[o] = m; V[o] = v; }
        }
        if (mark_synced && threadIdx.x == 0) tsync[row] = (int)*step_ctr;
    }
}

// ---------------------------------------------------------------------------------------
// data parallel, row-sparse exchange of the first encoder layer's gradient.
// pack:   hdr[0] = count, hdr[1 + r] = item id, vals[r][0:h] = gW1T[item][0:h]; the gradient row is
//         cleared.  One workgroup per listed row.
// unpack: gW1T[item] += vals[r] for one peer's packet (rows are unique inside a packet, so plain
//         read-modify-write; peers are applied by consecutive launches in rank order, which makes
//         the sum bitwise identical on every rank) and the item joins the union list.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void w1_pack_kernel(const int* __restrict__ ulist, const int* __restrict__ ucount,
                                                      float* __restrict__ gW1T, int ldw, int h, int cap,
                                                      int* __restrict__ hdr, float* __restrict__ vals) {
    const int cnt = min(*ucount, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = (*ucount > cap) ? -1 : cnt;
    for (int r = blockIdx.x; r < cnt; r += gridDim.x) {
        const int row = ulist[r];
        if (threadIdx.x == 0) hdr[1 + r] = row;
        for (int c = threadIdx.x; c < h; c += 256) {
            vals[(size_t)r * h + c] = gW1T[(size_t)row * ldw + c];
            gW1T[(size_t)row * ldw + c] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void w1_unpack_kernel(const int* __restrict__ hdr, const float* __restrict__ vals,
                                                        int h, float* __restrict__ gW1T, int ldw,
                                                        int* __restrict__ mark, const int* __restrict__ stamp_p,
                                                        int* __restrict__ ulist, int* __restrict__ ucount) {
    const int cnt = hdr[0];
    const int stamp = *stamp_p;
    for (int r = blockIdx.x; r < cnt; r += gridDim.x) {
        const int row = hdr[1 + r];
        for (int c = threadIdx.x; c < h; c += 256) gW1T[(size_t)row * ldw + c] += vals[(size_t)r * h + c];
        if (threadIdx.x == 0 && atomicExch(&mark[row], stamp) != stamp) ulist[atomicAdd(ucount, 1)] = row;
    }
}

// ---------------------------------------------------------------------------------------
// On-device tail of the evaluation path: what the reference does on the host
// with the full [n_test, N] score matrix - remove_non_missing (row-wise min-max scaling, items
// already in the input set to 0; evaluation.py:183-199) followed by argtopk (evaluation.py:20-58)
// - reduced to the k best items per row, so only [rows, k] leaves the GPU.
// One workgroup per row.  Known items are masked to -inf in place (their original scores still
// enter the row minimum / maximum), every thread keeps a sorted top-K of its strided share in
// registers, the 256 lists are merged by K rounds of block-wide argmax.
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void topk_rows_kernel(float* __restrict__ scores, int ld, int n_items, BatchView bv,
                                                        int exclude_known, int k_out, int* __restrict__ idx_out,
                                                        float* __restrict__ val_out) {
    __shared__ float s_val[4]; __shared__ int s_idx[4]; __shared__ int s_who[4];
    __shared__ float s_min[4], s_max[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sc = scores + (size_t)row * ld;
    float vmin = INFINITY, vmax = -INFINITY;
    if (exclude_known) {
        const int dc = bv.doc(row);
        const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
        for (int64_t e = lo + tid; e < hi; e += 256) {
            const int i = bv.indices[e];
            const float v = sc[i];
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            sc[i] = -INFINITY;
        }
        __syncthreads();
    }
    float tv[K]; int ti[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    for (int i = tid; i < n_items; i += 256) {
        const float v = sc[i];
        if (v != -INFINITY) { vmin = fminf(vmin, v); vmax = fmaxf(vmax, v); }
        if (v > tv[K - 1]) {
            tv[K - 1] = v; ti[K - 1] = i;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                if (tv[j] > tv[j - 1]) {
                    const float fv = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = fv;
                    const int iv = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = iv;
                }
            }
        }
    }
    // row minimum / maximum
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64)); }
    if (lane == 0) { s_min[wave] = vmin; s_max[wave] = vmax; }
    __syncthreads();
    vmin = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    vmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const float span = vmax - vmin;
    const float inv = span > 0.f ? 1.f / span : 1.f;
    // K rounds: block-wide argmax over every thread's current head (lists are sorted descending)
    for (int r = 0; r < k_out; ++r) {
        float bv_ = tv[0]; int bi = ti[0]; int who = tid;
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv_, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ow = __shfl_xor(who, o, 64);
            if (ov > bv_ || (ov == bv_ && (unsigned)oi < (unsigned)bi)) { bv_ = ov; bi = oi; who = ow; }
        }
        if (lane == 0) { s_val[wave] = bv_; s_idx[wave] = bi; s_who[wave] = who; }
        __syncthreads();
        float gv = s_val[0]; int gi = s_idx[0], gw = s_who[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > gv || (s_val[w] == gv && (unsigned)s_idx[w] < (unsigned)gi)) { gv = s_val[w]; gi = s_idx[w]; gw = s_who[w]; }
        if (tid == 0) {
            idx_out[(size_t)row * k_out + r] = gi;
            val_out[(size_t)row * k_out + r] = gi >= 0 ? (gv - vmin) * inv : 0.f;
        }
        if (tid == gw) {            // pop the winner's head
#pragma unroll
            for (int j = 0; j < K - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = -1;
        }
        __syncthreads();
    }
}

// dst[i] = (first ? 0 : dst[i]) + src[i]   (summing the peers' small-layer gradients in rank order)
__global__ void accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, int first) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (first ? 0.f : dst[i]) + src[i];
}

__global__ void fill_int_kernel(int* p, size_t n, int v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- all peers in ONE launch each (a launch per peer is ~17 us x world, twice per step) ----------------------
// map:  for every packed row of every peer: pslot / ptag [item][peer] = its slot in that peer's packet (tag = the
//       exchange's stamp, so nothing has to be cleared), and the item joins the union list.  grid (x, peers).
// sum:  one workgroup per item of the union list adds the peers' rows IN RANK ORDER (bitwise the same sum on every
//       rank, and the same as consecutive per-peer launches) into the gradient image.
__global__ __launch_bounds__(256) void w1_map_kernel(const char* __restrict__ pk, long long stride_bytes, int W,
                                                     int* __restrict__ pslot, int* __restrict__ ptag,
                                                     int* __restrict__ mark, const int* __restrict__ stamp_p,
                                                     int* __restrict__ ulist, int* __restrict__ ucount) {
    const int p = blockIdx.y;
    const int* hdr = reinterpret_cast<const int*>(pk + (size_t)p * stride_bytes);
    const int cnt = hdr[0], stamp = *stamp_p;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < cnt; r += gridDim.x * 256) {
        const int row = hdr[1 + r];
        pslot[(size_t)row * W + p] = r;
        ptag[(size_t)row * W + p] = stamp;
        if (atomicExch(&mark[row], stamp) != stamp) ulist[atomicAdd(ucount, 1)] = row;
    }
}

__global__ __launch_bounds__(256) void w1_sum_kernel(const char* __restrict__ vals0, long long stride_bytes, int h,
                                                     int n_peers, int W, const int* __restrict__ pslot,
                                                     const int* __restrict__ ptag, const int* __restrict__ stamp_p,
                                                     const int* __restrict__ ulist, const int* __restrict__ ucount,
                                                     float* __restrict__ gW1T, int ldw) {
    const int cnt = *ucount, stamp = *stamp_p;
    for (int u = blockIdx.x; u < cnt; u += gridDim.x) {
        const int row = ulist[u];
        for (int c = threadIdx.x; c < h; c += 256) {
            float acc = gW1T[(size_t)row * ldw + c];
            for (int p = 0; p < n_peers; ++p)
                if (ptag[(size_t)row * W + p] == stamp)
                    acc += reinterpret_cast<const float*>(vals0 + (size_t)p * stride_bytes)[(size_t)pslot[(size_t)row * W + p] * h + c];
            gW1T[(size_t)row * ldw + c] = acc;
        }
    }
}

// dst[i] = sum over peers (rank order) of src_p[i]: the small encoder layers behind the packed rows
__global__ void accumulate_peers_kernel(float* __restrict__ dst, const char* __restrict__ src0, long long stride_bytes,
                                        int n_peers, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int p = 0; p < n_peers; ++p) acc += reinterpret_cast<const float*>(src0 + (size_t)p * stride_bytes)[i];
        dst[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// Dense batch -> CSR on the device (the reference's call form: partial_fit / predict take the dense [B, N] matrix that
// X[start:end].toarray() produced, aae.py:745-754, 823).  The matrix crosses PCIe once, as it is (float32 or float64);
// three small launches compact it: per-row counts (+ the [0, 1] target check of F.binary_cross_entropy), a one-block
// scan -> indptr, an ordered fill (ascending columns per row, as scipy's tocsr()).  stats: [0] longest row, [1] total
// entries, [2] != 0: a value outside [0, 1], [3] != 0: more entries than `capacity`.
// ---------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void dense_count_kernel(const T* __restrict__ X, long long ld, int n_cols,
                                                          int* __restrict__ rowcnt, int* __restrict__ stats) {
    __shared__ int red[4];
    const T* x = X + (size_t)blockIdx.x * ld;
    int cnt = 0, bad = 0;
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        const float v = (float)x[c];
        cnt += v != 0.f;
        bad |= !(v >= 0.f && v <= 1.f);
    }
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); bad |= __shfl_xor(bad, o, 64); }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    if (bad && (threadIdx.x & 63) == 0) atomicOr(&stats[2], 1);
    __syncthreads();
    if (threadIdx.x == 0) rowcnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void dense_scan_kernel(const int* __restrict__ rowcnt, int rows, long long capacity,
                                                          long long* __restrict__ indptr, int* __restrict__ stats) {
    __shared__ long long part[1024];
    __shared__ int mx[1024];
    const int t = threadIdx.x, per = (rows + 1023) / 1024;
    const int lo = min(t * per, rows), hi = min(rows, lo + per);
    long long s = 0; int m = 0;
    for (int i = lo; i < hi; ++i) { s += rowcnt[i]; m = max(m, rowcnt[i]); }
    part[t] = s; mx[t] = m;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = t >= o ? part[t - o] : 0;
        const int w = t >= o ? mx[t - o] : 0;
        __syncthreads();
        part[t] += v; mx[t] = max(mx[t], w);
        __syncthreads();
    }
    long long run = part[t] - s;
    for (int i = lo; i < hi; ++i) { indptr[i] = run; run += rowcnt[i]; }
    if (t == 1023) {
        indptr[rows] = part[1023];
        stats[0] = mx[1023];
        stats[1] = (int)min(part[1023], (long long)0x7FFFFFFF);
        if (part[1023] > capacity) stats[3] = 1;
    }
}

template <class T>
__global__ __launch_bounds__(256) void dense_fill_kernel(const T* __restrict__ X, long long ld, int n_cols,
                                                         const long long* __restrict__ indptr, const int* __restrict__ stats,
                                                         int* __restrict__ indices, float* __restrict__ values) {
    __shared__ int wsum[4];
    if (stats[3]) return;                                  // over capacity: nothing is written, the host raises
    const T* x = X + (size_t)blockIdx.x * ld;
    long long base = indptr[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < n_cols; c0 += 256) {             // 256 consecutive columns per pass: positions stay ordered
        const int c = c0 + threadIdx.x;
        const float v = c < n_cols ? (float)x[c] : 0.f;
        const unsigned long long m = __ballot(v != 0.f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (v != 0.f) { indices[base + off + before] = c; values[base + off + before] = v; }
        base += total;
        __syncthreads();
    }
}

// predict-time prologue of the unique-row pass: new stamp, empty list
__global__ void bump_stamp_kernel(int* stamp, int* ucount) { *stamp += 1; *ucount = 0; }

// once per step, one launch: rng step counter += 1 and the four optimisers' step counts ->
// scalars (see OptScalars) + this step's row of the lazy-Adam table.  Thread i = optimiser i.
__global__ void advance_step_kernel(OptScalars* sc, long long* ctr, LazyTab* tab, int* stamp, int* ucount,
                                    float* losses) {
    advance_step_body(sc, ctr, tab, stamp, ucount, losses, (int)threadIdx.x);
}

}  // namespace aae
