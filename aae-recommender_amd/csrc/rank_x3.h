// Fused predict -> rank (reference aaerec/aae.py:840-870 predict, evaluation.py:183-199 remove_non_missing,
// evaluation.py:20-58 argtopk): the decoder's output layer, its sigmoid, the row minimum / maximum the reference scales with,
// the known-item mask and the top-k selection in ONE pass over dec.lin3 - the [rows, n_items] score matrix never exists in
// HBM (r1-r3: aae_predict wrote it, 40 MB per 100 documents at C3, and topk_rows_kernel read it back with one workgroup per
// row).  Any number of rows per launch (row blocks of 112), so the parameter stream is amortised over the whole call.
//
//   known_mask_kernel   the rows' known items as a bitmap [rows][ceil(N / 32)] (one workgroup per row) + dh2's bias column
//   rank_x3_kernel      per tile of 32 items, one persistent 1024-thread workgroup per CU (the critical launch of the training
//                       step, dec_crit_x3.h, minus everything backward): V3a tile -> three bf16 images (3-term split of the
//                       fp32 values: fp32 products emulated on the bf16 matrix cores; one_term: bf16 mode), logits = dh2 * V3a^T
//                       with dh2 split in registers, then - in the same phase as the NEXT tile's image build, two LDS-only
//                       barriers per tile - the epilogue on the LOGITS (the sigmoid is monotone: it is applied to the winners
//                       and to the row's extremes only, in the merge): minimum / maximum over ALL cells of the row, and every
//                       thread's sorted top-K (registers) of its 4 items x the workgroup's tiles, known items skipped.  At
//                       the end the 8 threads of a row merge their lists: K candidates + (min, max) per (row, workgroup).
//   rank_merge_kernel   one wave per row: K rounds of wave-wide argmax over the workgroups' sorted candidate lists, the
//                       winners' sigmoids scaled (v - min) / (max - min) as topk_rows_kernel (kernels.h) emits them.
// Ties go to the smaller item id at every level, as in topk_rows_kernel.
#pragma once
#include "dec_crit_x3.h"

namespace aae {

struct RankArgs {
    const float* dh2; int ldh;          // [B][ldh] decoder hidden activations, constant-1 column at h
    const float* V3a; int ldv;          // [N][ldv] dec.lin3 augmented with its bias column
    int N, B;                           // items, rows of this call
    int nblk, Bb;                       // row blocks of Bb rows: workgroup w -> block w % nblk, tiles w / nblk, + gridDim.x / nblk, ...
    const unsigned* known; int kw;      // [B][kw] bit (i & 31) of word (i >> 5): item i is one of the row's inputs; NULL: rank everything
    float* cand_v; int* cand_i;         // [B][wgs][K] per-workgroup candidates, sorted descending
    float* mm;                          // [B][wgs][2] per-workgroup (min, max) over all cells
    int one_term;                       // bf16 mode: operands rounded to bf16 (first term of the split only)
    int dbg;                            // timing-only ablation mask (AAE_RANK_SKIP), 0 in production
};

// One workgroup per row: the row's known items -> its bitmap (known != NULL), and the bias input of the output layer: column
// h of the row's dh2 = 1, the padding columns behind it = 0 (the chain program stores the layer's h outputs only; the
// workspace held other data before)
__global__ __launch_bounds__(256) void known_mask_kernel(BatchView bv, unsigned* __restrict__ known, int kw,
                                                         float* __restrict__ dh2, int ldh, int h) {
    const int row = blockIdx.x, tid = threadIdx.x;
    if (h + tid < ldh) dh2[(size_t)row * ldh + h + tid] = tid == 0 ? 1.f : 0.f;
    if (!known) return;
    unsigned* k = known + (size_t)row * kw;
    for (int i = tid; i < kw; i += 256) k[i] = 0u;
    __syncthreads();
    const int dc = bv.doc(row);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    for (int64_t e = lo + tid; e < hi; e += 256) {
        const int i = bv.indices[e];
        atomicOr(&k[i >> 5], 1u << (i & 31));
    }
}

constexpr int kRRS = 36;       // row stride (floats) of the raw-logit halves [b][n]: 16-byte aligned rows for the epilogue's float4 reads

// the rank kernels' row block: 8 MFMA row blocks = 128 rows (r5; its four wave pairs hold two 16-row blocks each - the training
// kernels' 7 left the last pair half empty - and its epilogue has a thread per (row, item quad) of 128 rows): 512 rows per call
// are 4 row blocks instead of 5, every tile's images are built 4 times instead of 5
constexpr int kRankMB2 = 8, kRankGR2 = 16 * kRankMB2;
inline size_t rank_x3v2_lds_bytes(int NB) {
    const int KC1 = (NB + 1) / 2, S1 = x3_stride(KC1);
    return sizeof(float) * ((size_t)3 * kTI * S1 + (size_t)4 * kRankGR2 * kRRS);
}

inline size_t rank_x3_lds_bytes(int NB) {          // (128-row blocks as well: its waves are (row block, k half) pairs - 16 of them now, not 14)
    const int KC1 = (NB + 1) / 2, NKS = (KC1 + 1) / 2, S1 = x3_stride(KC1);
    const int lsteps = NKS > kXRegSteps ? NKS - kXRegSteps : 0;
    return sizeof(float) * ((size_t)3 * kTI * S1 + (size_t)2 * kRankGR2 * kRRS + (size_t)lsteps * kRankMB2 * 3 * 64 * 4);
}

template <int NB, int K, bool WIN = false>      // WIN: dec.lin3 beyond 2^31 bytes (dec_fused.h X3WindowT)
__global__ __launch_bounds__(kNT) void rank_x3_kernel(RankArgs a) {
    const bool one = a.one_term != 0;
    constexpr int KC1 = (NB + 1) / 2, NKS = (KC1 + 1) / 2;
    constexpr int NKR = NKS > kXRegSteps ? kXRegSteps : NKS, NKL = NKS - NKR;
    constexpr int S1 = x3_stride(KC1);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* v3K = reinterpret_cast<unsigned*>(lds);           // [3][32][S1] V3a tile, k = hidden column
    constexpr int kGR = kRankGR2, kMB = kRankMB2;               // (the rank kernels' row block: 128 rows)
    float* raw = reinterpret_cast<float*>(v3K + 3 * kTI * S1);  // [2][kGR][kRRS] the two k halves of the logits
    u32x4_t* dAl = reinterpret_cast<u32x4_t*>(raw + 2 * kGR * kRRS);    // [kMB][NKL][3][64] dh2 fragments beyond the register steps

    const int nblk = a.nblk > 1 ? a.nblk : 1;
    const int blk = (int)blockIdx.x % nblk, wgi = (int)blockIdx.x / nblk, wgs = (int)gridDim.x / nblk;
    if (wgi >= wgs) return;
    const int erow0 = blk * a.Bb;
    const int B = min(a.Bb, a.B - erow0);
    const float* dh2_blk = a.dh2 + (size_t)erow0 * a.ldh;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldv = a.ldv, N = a.N;
    const int ntiles = (N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;

    // the parameter stream: tensor base in a buffer descriptor, tile offset scalar, slot offset in one vector register;
    // reads beyond the tensor return zero
    const unsigned lane_off = (unsigned)tid * 16u;
    const unsigned tile_bytes = (unsigned)(kTI * ldv) * 4u;
    X3WindowT<WIN> win(N, ldv, tile_bytes);            // (the descriptor's window of the tensor, dec_crit_x3.h: layers beyond 2^31 bytes)
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a);
    auto load_span = [&](int tile, float4* r) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            r[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, tid + kNT * j < tile_f4 ? lane_off + (unsigned)(kNT * 16 * j) : 0x80000000u, win.so(tile), 0));      // (a slot beyond the tile's span: no access, dec_crit_x3.h)
    };
    // epilogue: thread -> row eb, items 4 eq .. 4 eq + 3 of the tile
    const int eb = tid >> 3, eq = tid & 7;
    const bool erow = eb < B;
    const unsigned* kn = a.known ? a.known + (size_t)(erow0 + min(eb, B - 1)) * a.kw : nullptr;
    float4 vreg[NV];
    int tile = wgi;
    const int stride = wgs;
    unsigned kw_next = 0u;
    if (tile < ntiles) {
        load_span(tile, vreg);
        if (kn) kw_next = kn[tile];
    }
    for (int i = tid; i < 3 * kTI * S1 + 2 * kGR * kRRS; i += kNT) v3K[i] = 0u;

    // dh2 -> split A fragments: wave w < 14 = (row block w % 7, k half w / 7), lane (fr, fk) holds row 16 mb + fr, k = 32 kc + 8 fk + {0..7}
    const bool g1 = wave < 2 * kMB;
    const int mb1 = wave % kMB, kh = wave / kMB;
    bf16x8 dA[NKR][3];
    {
        const int row = 16 * mb1 + fr;
        const float* src = dh2_blk + (size_t)min(row, B - 1) * a.ldh;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            const int k0 = 32 * (kh + 2 * j) + 8 * fk;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
            if (g1 && row < B && k0 < a.ldh) x = *reinterpret_cast<const float4*>(src + k0);
            if (g1 && row < B && k0 + 4 < a.ldh) y = *reinterpret_cast<const float4*>(src + k0 + 4);
            unsigned p[3][4];
            split3_pair(x.x, x.y, p[0][0], p[1][0], p[2][0], one);
            split3_pair(x.z, x.w, p[0][1], p[1][1], p[2][1], one);
            split3_pair(y.x, y.y, p[0][2], p[1][2], p[2][2], one);
            split3_pair(y.z, y.w, p[0][3], p[1][3], p[2][3], one);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const u32x4_t v = {p[t][0], p[t][1], p[t][2], p[t][3]};
                if (j < NKR) dA[j < NKR ? j : 0][t] = __builtin_bit_cast(bf16x8, v);
                else if (g1 && kh == 0) dAl[((mb1 * NKL + (j - NKR)) * 3 + t) * 64 + lane] = v;
            }
        }
    }
    const int nks = (KC1 - kh + 1) / 2;

    int s_rc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int fc = min(tid + kNT * j, tile_f4 - 1), row = fc / f4_per_row;
        s_rc[j] = row * 64 + (fc - row * f4_per_row);
    }
    float tv[K]; int ti[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    float vmin = INFINITY, vmax = -INFINITY;

    // the epilogue of one finished tile: the thread's four cells
    auto epilogue = [&](int i0, unsigned kword) {
        if (!erow || (a.dbg & 2)) return;
        const float4 lA = *reinterpret_cast<const float4*>(raw + eb * kRRS + 4 * eq);
        const float4 lB = *reinterpret_cast<const float4*>(raw + kGR * kRRS + eb * kRRS + 4 * eq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * eq + j, item = i0 + n;
            const float v = (&lA.x)[j] + (&lB.x)[j];       // the LOGIT: sigmoid is monotone - applied to the winners only (merge)
            if (item < N) {
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
                if (!((kword >> n) & 1u) && v > tv[K - 1] && !(a.dbg & 1)) {
                    tv[K - 1] = v; ti[K - 1] = item;
#pragma unroll
                    for (int s = K - 1; s > 0; --s) {
                        if (tv[s] > tv[s - 1]) {
                            const float fv = tv[s]; tv[s] = tv[s - 1]; tv[s - 1] = fv;
                            const int iv = ti[s]; ti[s] = ti[s - 1]; ti[s - 1] = iv;
                        }
                    }
                }
            }
        }
    };
    __syncthreads();

    int prev_i0 = -1; unsigned kw_prev = 0u;
    for (; tile < ntiles; tile += stride) {
        if (win.moves(tile)) rP = win.desc(a.V3a);
        const int i0 = tile * kTI;
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int frz = fr + oz, fkz = fk + oz;
        lds_barrier();                          // the previous tile's products are in `raw`, its readers of v3K are done
        // ---- the previous tile's epilogue and this tile's images, one VALU phase
        if (prev_i0 >= 0) epilogue(prev_i0, kw_prev);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (tid + kNT * j < tile_f4 && !(a.dbg & 8)) {
                float4 p = vreg[j];
                if (i0 + (s_rc[j] >> 6) >= N) p = make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned q0[3], q1[3];
                split3_pair(p.x, p.y, q0[0], q0[1], q0[2], one);
                split3_pair(p.z, p.w, q1[0], q1[1], q1[2], one);
                unsigned* d = v3K + (s_rc[j] >> 6) * S1 + 2 * (s_rc[j] & 63);
#pragma unroll
                for (int t = 0; t < 3; ++t) *reinterpret_cast<uint2*>(d + t * (kTI * S1)) = make_uint2(q0[t], q1[t]);
            }
        }
        prev_i0 = i0; kw_prev = kw_next;
        {
            const int nt = min(tile + stride, ntiles - 1);
            load_span(nt, vreg);
            if (kn) kw_next = kn[nt];
        }
        lds_barrier();
        // ---- logits of the wave's row block x both item halves over its k-steps -> its half's raw tile
        if (g1 && !(a.dbg & 4)) {
            float* rw = raw + kh * (kGR * kRRS) + (16 * mb1 + 4 * fkz) * kRRS + frz;
#pragma unroll
            for (int nb2 = 0; nb2 < 2; ++nb2) {
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NKS; ++j)
                    if (j < nks) {
                        bf16x8 bb[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) bb[t] = x3_frag(v3K + t * (kTI * S1), 16 * nb2 + frz, S1, kh + 2 * j, fkz);
                        if (j < NKR) c = mfma_x3(dA[j < NKR ? j : 0], bb, c);
                        else {
                            bf16x8 al[3];
#pragma unroll
                            for (int t = 0; t < 3; ++t) al[t] = __builtin_bit_cast(bf16x8, dAl[((mb1 * NKL + (j - NKR)) * 3 + t) * 64 + lane + oz]);
                            c = mfma_x3(al, bb, c);
                        }
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r * kRRS + 16 * nb2] = c[r];
            }
        }
    }
    lds_barrier();
    if (prev_i0 >= 0) epilogue(prev_i0, kw_prev);

    // ---- the 8 threads of a row merge their lists (K rounds of an 8-lane argmax, ties to the smaller item) -> K candidates
    // of this workgroup for the row; its minimum / maximum
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64)); }
    const size_t slot = (size_t)(erow0 + eb) * wgs + wgi;
    if (erow && eq == 0) { a.mm[2 * slot] = vmin; a.mm[2 * slot + 1] = vmax; }
    for (int r = 0; r < K; ++r) {
        float bv = tv[0]; int bi = ti[0]; int who = eq;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ow = __shfl_xor(who, o, 64);
            if (ov > bv || (ov == bv && (unsigned)oi < (unsigned)bi)) { bv = ov; bi = oi; who = ow; }
        }
        if (erow && eq == 0) { a.cand_v[slot * K + r] = bv; a.cand_i[slot * K + r] = bi; }
        if (eq == who) {
#pragma unroll
            for (int j = 0; j < K - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = -1;
        }
    }
}

// v2 (r4): the GEMM's wave mapping.  v1 (the critical launch's: wave = (row block, k half)) has every one of 14 waves read
// the tile's B fragments for itself - 0.5 LDS operand reads per matrix instruction, and the phase ran at half the pipe's rate
// (ablation: 2.2 us per tile for 1.0 us of pipe time).  Here wave = (row-block PAIR rp = wave / 4, k QUARTER kq): one B
// fragment serves both row blocks of the pair (0.25 reads per instruction), all 16 waves multiply, the four k quarters'
// partial sums meet in the epilogue (4 raw tiles instead of 2), and the quarter a wave takes is rotated by its pair,
// kq = (wave + rp) & 3, so that every SIMD (wave & 3) gets 12-13 of the 49 (row block, k-step) units.
template <int NB, int K, bool WIN = false>      // WIN: dec.lin3 beyond 2^31 bytes (dec_fused.h X3WindowT)
__global__ __launch_bounds__(kNT) void rank_x3v2_kernel(RankArgs a) {
    const bool one = a.one_term != 0;
    constexpr int KC1 = (NB + 1) / 2, NKQ = (KC1 + 3) / 4;       // 32-wide k-steps over the h + 1 hidden columns; per k quarter
    constexpr int S1 = x3_stride(KC1);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* v3K = reinterpret_cast<unsigned*>(lds);           // [3][32][S1] V3a tile, k = hidden column
    constexpr int kGR = kRankGR2, kMB = kRankMB2;               // (this kernel's row block: 128 rows)
    float* raw = reinterpret_cast<float*>(v3K + 3 * kTI * S1);  // [4][kGR][kRRS] the four k quarters of the logits

    const int nblk = a.nblk > 1 ? a.nblk : 1;
    const int blk = (int)blockIdx.x % nblk, wgi = (int)blockIdx.x / nblk, wgs = (int)gridDim.x / nblk;
    if (wgi >= wgs) return;
    const int erow0 = blk * a.Bb;
    const int B = min(a.Bb, a.B - erow0);
    const float* dh2_blk = a.dh2 + (size_t)erow0 * a.ldh;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldv = a.ldv, N = a.N;
    const int ntiles = (N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;

    // the parameter stream: tensor base in a buffer descriptor, tile offset scalar, slot offset in one vector register;
    // reads beyond the tensor return zero
    const unsigned lane_off = (unsigned)tid * 16u;
    const unsigned tile_bytes = (unsigned)(kTI * ldv) * 4u;
    X3WindowT<WIN> win(N, ldv, tile_bytes);            // (the descriptor's window of the tensor, dec_crit_x3.h: layers beyond 2^31 bytes)
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a);
    auto load_span = [&](int tile, float4* r) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            r[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, tid + kNT * j < tile_f4 ? lane_off + (unsigned)(kNT * 16 * j) : 0x80000000u, win.so(tile), 0));      // (a slot beyond the tile's span: no access, dec_crit_x3.h)
    };
    // epilogue: thread -> row eb, items 4 eq .. 4 eq + 3 of the tile
    const int eb = tid >> 3, eq = tid & 7;
    const bool erow = eb < B;
    const unsigned* kn = a.known ? a.known + (size_t)(erow0 + min(eb, B - 1)) * a.kw : nullptr;
    float4 vreg[NV];
    int tile = wgi;
    const int stride = wgs;
    unsigned kw_next = 0u;
    if (tile < ntiles) {
        load_span(tile, vreg);
        if (kn) kw_next = kn[tile];
    }
    for (int i = tid; i < 3 * kTI * S1 + 4 * kGR * kRRS; i += kNT) v3K[i] = 0u;

    // dh2 -> split A fragments: wave = (row-block pair rp, k quarter kq); lane (fr, fk) of a fragment holds row 16 mb + fr,
    // k = 32 kc + 8 fk + {0..7}, kc = kq + 4 j
    const int rp = wave >> 2, kq = (wave + rp) & 3;
    const int nblk2 = min(2, kMB - 2 * rp);     // row blocks of this pair (the last pair of 7 blocks: one)
    bf16x8 dA[2][NKQ][3];
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
        const int row = 16 * (2 * rp + bq) + fr;
        const float* src = dh2_blk + (size_t)min(row, B - 1) * a.ldh;
#pragma unroll
        for (int j = 0; j < NKQ; ++j) {
            const int k0 = 32 * (kq + 4 * j) + 8 * fk;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
            if (bq < nblk2 && row < B && k0 < a.ldh) x = *reinterpret_cast<const float4*>(src + k0);
            if (bq < nblk2 && row < B && k0 + 4 < a.ldh) y = *reinterpret_cast<const float4*>(src + k0 + 4);
            unsigned p[3][4];
            split3_pair(x.x, x.y, p[0][0], p[1][0], p[2][0], one);
            split3_pair(x.z, x.w, p[0][1], p[1][1], p[2][1], one);
            split3_pair(y.x, y.y, p[0][2], p[1][2], p[2][2], one);
            split3_pair(y.z, y.w, p[0][3], p[1][3], p[2][3], one);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const u32x4_t v = {p[t][0], p[t][1], p[t][2], p[t][3]};
                dA[bq][j][t] = __builtin_bit_cast(bf16x8, v);
            }
        }
    }
    const int nks = kq < KC1 ? (KC1 - kq + 3) / 4 : 0;          // k-steps this wave really has

    int s_rc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int fc = min(tid + kNT * j, tile_f4 - 1), row = fc / f4_per_row;
        s_rc[j] = row * 64 + (fc - row * f4_per_row);
    }
    float tv[K]; int ti[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    float vmin = INFINITY, vmax = -INFINITY;

    // the epilogue of one finished tile: the thread's four cells
    auto epilogue = [&](int i0, unsigned kword) {
        if (!erow || (a.dbg & 2)) return;
        const float4 q0 = *reinterpret_cast<const float4*>(raw + eb * kRRS + 4 * eq);
        const float4 q1 = *reinterpret_cast<const float4*>(raw + kGR * kRRS + eb * kRRS + 4 * eq);
        const float4 q2 = *reinterpret_cast<const float4*>(raw + 2 * kGR * kRRS + eb * kRRS + 4 * eq);
        const float4 q3 = *reinterpret_cast<const float4*>(raw + 3 * kGR * kRRS + eb * kRRS + 4 * eq);
        const float4 lA = make_float4(q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, q0.w + q1.w);
        const float4 lB = make_float4(q2.x + q3.x, q2.y + q3.y, q2.z + q3.z, q2.w + q3.w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * eq + j, item = i0 + n;
            const float v = (&lA.x)[j] + (&lB.x)[j];       // the LOGIT: sigmoid is monotone - applied to the winners only (merge)
            if (item < N) {
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
                if (!((kword >> n) & 1u) && v > tv[K - 1] && !(a.dbg & 1)) {
                    tv[K - 1] = v; ti[K - 1] = item;
#pragma unroll
                    for (int s = K - 1; s > 0; --s) {
                        if (tv[s] > tv[s - 1]) {
                            const float fv = tv[s]; tv[s] = tv[s - 1]; tv[s - 1] = fv;
                            const int iv = ti[s]; ti[s] = ti[s - 1]; ti[s - 1] = iv;
                        }
                    }
                }
            }
        }
    };
    __syncthreads();

    int prev_i0 = -1; unsigned kw_prev = 0u;
    for (; tile < ntiles; tile += stride) {
        if (win.moves(tile)) rP = win.desc(a.V3a);
        const int i0 = tile * kTI;
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int frz = fr + oz, fkz = fk + oz;
        lds_barrier();                          // the previous tile's products are in `raw`, its readers of v3K are done
        // ---- the previous tile's epilogue and this tile's images, one VALU phase
        if (prev_i0 >= 0) epilogue(prev_i0, kw_prev);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (tid + kNT * j < tile_f4 && !(a.dbg & 8)) {
                float4 p = vreg[j];
                if (i0 + (s_rc[j] >> 6) >= N) p = make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned q0[3], q1[3];
                split3_pair(p.x, p.y, q0[0], q0[1], q0[2], one);
                split3_pair(p.z, p.w, q1[0], q1[1], q1[2], one);
                unsigned* d = v3K + (s_rc[j] >> 6) * S1 + 2 * (s_rc[j] & 63);
#pragma unroll
                for (int t = 0; t < 3; ++t) *reinterpret_cast<uint2*>(d + t * (kTI * S1)) = make_uint2(q0[t], q1[t]);
            }
        }
        prev_i0 = i0; kw_prev = kw_next;
        {
            const int nt = min(tile + stride, ntiles - 1);
            load_span(nt, vreg);
            if (kn) kw_next = kn[nt];
        }
        lds_barrier();
        // ---- logits of the wave's row-block pair x both item halves over its k-steps -> its quarter's raw tile
        if (nks > 0 && !(a.dbg & 4)) {
            float* rw = raw + kq * (kGR * kRRS) + (32 * rp + 4 * fkz) * kRRS + frz;
#pragma unroll
            for (int nb2 = 0; nb2 < 2; ++nb2) {
                f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
                for (int j = 0; j < NKQ; ++j)
                    if (j < nks) {
                        bf16x8 bb[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) bb[t] = x3_frag(v3K + t * (kTI * S1), 16 * nb2 + frz, S1, kq + 4 * j, fkz);
                        c0 = mfma_x3(dA[0][j], bb, c0);
                        if (nblk2 > 1) c1 = mfma_x3(dA[1][j], bb, c1);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r * kRRS + 16 * nb2] = c0[r];
                if (nblk2 > 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rw[(16 + r) * kRRS + 16 * nb2] = c1[r];
                }
            }
        }
    }
    lds_barrier();
    if (prev_i0 >= 0) epilogue(prev_i0, kw_prev);

    // ---- the 8 threads of a row merge their lists (K rounds of an 8-lane argmax, ties to the smaller item) -> K candidates
    // of this workgroup for the row; its minimum / maximum
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64)); }
    const size_t slot = (size_t)(erow0 + eb) * wgs + wgi;
    if (erow && eq == 0) { a.mm[2 * slot] = vmin; a.mm[2 * slot + 1] = vmax; }
    for (int r = 0; r < K; ++r) {
        float bv = tv[0]; int bi = ti[0]; int who = eq;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ow = __shfl_xor(who, o, 64);
            if (ov > bv || (ov == bv && (unsigned)oi < (unsigned)bi)) { bv = ov; bi = oi; who = ow; }
        }
        if (erow && eq == 0) { a.cand_v[slot * K + r] = bv; a.cand_i[slot * K + r] = bi; }
        if (eq == who) {
#pragma unroll
            for (int j = 0; j < K - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = -1;
        }
    }
}

// One WAVE per row (4 rows per workgroup): lane t holds the sorted candidate lists of workgroups t, t + 64, ... (one list
// when the call has >= 4 row blocks, its own list as it stands); K rounds of a wave-wide argmax over the lanes' heads - no
// barrier, no LDS.  Candidates and extremes are logits; the scores the reference ranks are their sigmoids: emitted for the
// k winners, scaled by the row's (sigmoid(min), sigmoid(max)) as topk_rows_kernel (kernels.h) scales.
template <int K>
__global__ __launch_bounds__(256) void rank_merge_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                         const float* __restrict__ mm, int rows, int wgs, int k_out,
                                                         int* __restrict__ idx_out, float* __restrict__ val_out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float tv[K]; int ti[K];
    float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < K; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    for (int w = lane; w < wgs; w += 64) {
        const size_t slot = (size_t)row * wgs + w;
        vmin = fminf(vmin, mm[2 * slot]); vmax = fmaxf(vmax, mm[2 * slot + 1]);
        if (w == lane) {
#pragma unroll
            for (int j = 0; j < K; ++j) { tv[j] = cand_v[slot * K + j]; ti[j] = cand_i[slot * K + j]; }
        } else {
            for (int j = 0; j < K; ++j) {               // a further list of this lane: merged in (sorted: stop at the first loser)
                const float v = cand_v[slot * K + j]; const int i = cand_i[slot * K + j];
                if (!(v > tv[K - 1] || (v == tv[K - 1] && (unsigned)i < (unsigned)ti[K - 1]))) break;
                tv[K - 1] = v; ti[K - 1] = i;
#pragma unroll
                for (int s = K - 1; s > 0; --s) {
                    if (tv[s] > tv[s - 1] || (tv[s] == tv[s - 1] && (unsigned)ti[s] < (unsigned)ti[s - 1])) {
                        const float fv = tv[s]; tv[s] = tv[s - 1]; tv[s - 1] = fv;
                        const int iv = ti[s]; ti[s] = ti[s - 1]; ti[s - 1] = iv;
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64)); }
    const float smin = sigmoidf_(vmin), smax = sigmoidf_(vmax);
    const float span = smax - smin;
    const float inv = span > 0.f ? 1.f / span : 1.f;
    for (int r = 0; r < k_out; ++r) {
        float bv_ = tv[0]; int bi = ti[0]; int who = lane;
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv_, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ow = __shfl_xor(who, o, 64);
            if (ov > bv_ || (ov == bv_ && (unsigned)oi < (unsigned)bi)) { bv_ = ov; bi = oi; who = ow; }
        }
        if (lane == 0) {
            idx_out[(size_t)row * k_out + r] = bi;
            val_out[(size_t)row * k_out + r] = bi >= 0 ? (sigmoidf_(bv_) - smin) * inv : 0.f;
        }
        if (lane == who) {
#pragma unroll
            for (int j = 0; j < K - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = -1;
        }
    }
}

}  // namespace aae
