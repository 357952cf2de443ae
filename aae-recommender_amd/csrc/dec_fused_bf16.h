// Fused decoder output layer with bf16 matrix-core inputs (BASELINE config C2: "bf16 MFMA inputs, fp32 accumulate,
// fp32 master parameters and Adam") for reference-sized batches (B <= 112).
//
// Same work per tile of 32 items as dec_fused.h - logits, BCE, dV3 + dec_optim, dA2 (aae.py:176-177, 693-695 and
// their backward) - but the three products run on v_mfma_f32_16x16x32_bf16 (16x the fp32 matrix rate), so the
// matrix phases all but vanish (293 MFMAs of 16 cycles per tile against 2184 of 32) and the kernel is bound by the
// 24 B/parameter of the fused Adam stream: the structure below is built around keeping that stream in flight.
//
//   operands are rounded to bf16 ONCE when they enter LDS / registers, products accumulate in fp32:
//     dh2  (the decoder's last hidden activations)  -> LDS image dhA [b][k]      (GEMM1's A operand)
//                                                   -> registers  dhT [c][b]      (GEMM2's A operand, per wave)
//     V3a tile (fp32 master, read once from HBM)    -> LDS images v3K [n][k], v3T [c][n]; the fp32 values stay in
//                                                      registers for the optimiser
//     G = dL/dlogits (fp32 from the BCE epilogue)   -> LDS images gK [b][n], gT [n][b]
//   every image is "k-contiguous": a lane's 8 k-values of a 16x16x32 fragment are two 8-byte LDS reads; row strides
//   are 4 * odd dwords, which puts the 32 lanes of a ds_read_b64 half-wave on 32 distinct bank pairs.
//
//   per unit of 16 items (half a 32-item tile of the entry buckets), one persistent 1024-thread workgroup per CU, 4
//   LDS-only barriers.  16-item units keep a wave's share of the parameter / moment stream at ONE float4 slot per
//   stream (16 waves x 128 registers: with 32-item units the kernel spilled, and a scratch reload waits for every
//   older global load; 8 waves x 256 registers ran at half the issue rate: the phases are instruction-issue bound):
//     S0     this unit's V3a (registers, requested one unit ahead) -> v3K, v3T; request V3a(u+1), m(u), v(u)
//     GEMM1  logits[b][n] = dhA * v3K^T; epilogue: zero-target BCE -> gK, gT (bf16), raw logits (fp32), loss
//     S2     the tile's CSR entries (non-zero targets) patch their cells
//     GEMM2  dV3a^T[c][n] = dhT * gT^T -> fp32 staging tile os [n][c] (one 16-byte LDS write per lane);
//            GEMM3  dA2[b][c] += gK * v3T^T (fp32 accumulators across the units)
//     S5     dec_optim (Adam) on the unit in the tensors' own row-major order: thread t owns the t-th float4 of the
//            unit's contiguous 13 KB span of V3a / m / v - every global load and store of the kernel is a fully
//            coalesced 1 KB wave access.  (Loading the stream in GEMM2's accumulator layout instead - no staging tile,
//            one barrier less - made every 16-lane group of a load touch 16 different cache lines: the vector-memory
//            unit then needed ~2 us per unit just to take the loads, 4x the coalesced cost.)
#pragma once
#include "dec_fused.h"

namespace aae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// two floats -> packed bf16 pair (round to nearest even: v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ unsigned bf16_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ unsigned short bf16_bits(float a) { return (unsigned short)(bf16_pack(a, 0.f) & 0xFFFFu); }

// adam_update() without the per-element optimiser test
__device__ __forceinline__ void adam_only(float& p, float& m, float& v, float g, const OptScalars& s) {
    m = m + 0.1f * (g - m);
    v = v * 0.999f + (0.001f * g) * g;
    p = p + (s.neg_step_size * m) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * s.inv_bc2_sqrt + 1e-8f);
}

// row stride (dwords) of a k-contiguous bf16 image with kc 32-wide k-steps: 16 * kc + 4 (= 4 * odd)
__host__ __device__ constexpr int bf_stride(int kc) { return 16 * kc + 4; }

// One 16x16x32 fragment: rows `row` of a k-contiguous image, k-step kc.  Lane (fr, fk) takes
// k = 32kc + 4fk + {0..3} and 32kc + 16 + 4fk + {0..3} - the same subset for A and B, which is all an MFMA needs.
__device__ __forceinline__ bf16x8 bf_frag(const unsigned* img, int row, int S, int kc, int fk) {
    const unsigned* p = img + row * S + 16 * kc + 2 * fk;
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 8);
    const u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int kBfSR = 17;      // row stride (floats) of the raw-logit tile [b][n]
// The transposed V3a image v3T [c][n] (k = item) is WRITTEN by threads that hold 4 consecutive columns c of one item:
// consecutive lanes are 4 image rows apart, and with plain rows any even stride puts all 64 lanes of a 16-bit write on
// two banks (4 * 20 dwords = 16 mod 32).  Rows therefore go in blocks of four with a block stride of 66 dwords
// (2 mod 32): the lanes of a write walk the banks in steps of 2.
__host__ __device__ constexpr int vt_off(int c) { return (c >> 2) * 66 + (c & 3) * 16; }   // dword offset of row c (32 halfword slots)
constexpr int kBU = 16;        // items per unit
constexpr int kBT = 1024;      // threads per workgroup
constexpr int kBW = kBT / 64;  // waves

// LDS bytes for NB column blocks
inline size_t dec_fused_bf16_lds_bytes(int NB) {
    const int KC1 = (NB + 1) / 2, S1 = bf_stride(KC1), S2 = bf_stride(4), S3 = bf_stride(1);
    return sizeof(float) * ((size_t)kGR * S1 + (size_t)kBU * S1 + (size_t)vt_off(16 * NB) + (size_t)kGR * S3 +
                            (size_t)kBU * S2 + (size_t)2 * kGR * kBfSR + (size_t)kBU * kSO + 64);
}

// MODE as in dec_fused.h (section 3.2c): kDecCrit = S0, GEMM1, BCE, entries, GEMM3 (what the step waits for) and the
// unit's transposed bf16 dL/dlogits image gT [16][S2] stored to a.Gt; kDecOpt = that image back into LDS, GEMM2, the
// optimiser - on the side stream, beside the rest of the step, with non-temporal streams.
template <int NB, int MODE = kDecFused>   // NB = ceil((h + 1) / 16) column blocks (<= 16)
__global__ __launch_bounds__(kBT) void dec_fused_bf16_kernel(DecFusedArgs a) {
    constexpr bool kFwd = MODE != kDecOpt, kOpt = MODE != kDecCrit, kIsOpt = MODE == kDecOpt;
    constexpr int kAux = kIsOpt ? 2 : 0;        // buffer cache policy of the streams: 2 = nt
    constexpr int KC1 = (NB + 1) / 2;          // 32-wide k-steps over the h + 1 hidden columns
    constexpr int KR = 4;                       // 32-wide k-steps over the (<= 112 -> 128) batch rows
    constexpr int S1 = bf_stride(KC1), S2 = bf_stride(KR), S3 = bf_stride(1);
    static_assert(NB <= kBW, "one column block per wave");
    // Ownership.  GEMM1: wave w < row blocks owns row block w.  GEMM2 / GEMM3: wave w < NB owns column block w.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* dhA = reinterpret_cast<unsigned*>(lds);          // [kGR][S1]   dh2, k = hidden column
    unsigned* v3K = dhA + kGR * S1;                             // [16][S1]    V3a unit, k = hidden column
    unsigned* v3T = v3K + kBU * S1;                             // [16 NB] rows at vt_off(c): V3a unit transposed, k = item (16 of 32 used)
    unsigned* gK = v3T + vt_off(16 * NB);                          // [kGR][S3]   G, k = item (16 of 32 used)
    unsigned* gT = gK + kGR * S3;                               // [16][S2]    G transposed, k = batch row
    float* raw = reinterpret_cast<float*>(gT + kBU * S2);       // [2][kGR][kBfSR] raw logits (fp32): the two k-halves of GEMM1
    float* os = raw + 2 * kGR * kBfSR;                          // [16][kSO]   dV3a of the unit (fp32), row-major like the tensor
    float* red = os + kBU * kSO;                                // [64]

    const int tid = threadIdx.x, lane = tid & 63;
    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[10] = wall_clock64();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int B = a.B, ldv = a.ldv, N = a.N;
    const int nmb = (B + 15) >> 4;
    const int ntiles = (N + kTI - 1) / kTI;     // 32-item tiles of the entry buckets
    const int nunits = (N + kBU - 1) / kBU;
    const OptScalars sc = *a.sc;
    const bool do_adam = a.gradV3 == nullptr;

    // ---- once per workgroup: LDS images of dh2 (zero padded), zeroed G images, zero k-padding of the V3a images
    for (int i = tid; i < kGR * S1 + kBU * S1 + vt_off(16 * NB) + kGR * S3 + kBU * S2; i += kBT) dhA[i] = 0u;
    __syncthreads();
    {
        const int f4 = a.ldh >> 2;
        for (int f = tid; f < B * f4; f += kBT) {
            const int r = f / f4, c4 = f - r * f4;
            const float4 x = *reinterpret_cast<const float4*>(a.dh2 + (size_t)r * a.ldh + c4 * 4);
            *reinterpret_cast<uint2*>(dhA + r * S1 + c4 * 2) = make_uint2(bf16_pack(x.x, x.y), bf16_pack(x.z, x.w));
        }
    }
    __syncthreads();
    const bool own = wave < NB;                 // this wave owns a column block
    const int cb = min(wave, NB - 1);
    // dh2 transposed, in registers: fragment rows c = 16 cb + fr, k = batch row, picked out of the (zero padded,
    // already rounded) dhA image: no masks, no second pass over global memory
    bf16x8 dhT[KR];
    {
        const unsigned short* a16 = reinterpret_cast<const unsigned short*>(dhA);
        const int c = 16 * cb + fr;
#pragma unroll
        for (int kc = 0; kc < KR; ++kc) {
            unsigned h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int b = 32 * kc + 4 * fk + (e & 3) + ((e >> 2) << 4);          // < kGR unless kc == 3 && e >= 4
                h[e] = (32 * kc + ((e >> 2) << 4) + 15 < kGR) ? (unsigned)a16[b * (2 * S1) + c] : 0u;
            }
            const u32x4_t v = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            dhT[kc] = __builtin_bit_cast(bf16x8, v);
        }
    }
    f32x4 acc3[kMB];
#pragma unroll
    for (int q = 0; q < kMB; ++q) acc3[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float loss = 0.f;

    // The parameter stream: thread t owns the t-th float4 of a unit's contiguous span (16 rows x ldv floats) of V3a, m
    // and v.  Addressing: tensor base in a buffer resource descriptor (scalar registers), the unit's byte offset in a
    // scalar register, t * 16 in ONE vector register for every access of the kernel.  Nothing is clamped: the arena
    // keeps 2 * kTI padding rows behind V3a, its moments and its gradient (layout(): pad_rows), so the last unit's
    // rows >= N are readable (never used).  A store without a cell (threads beyond the span, rows beyond the
    // vocabulary, the moments in SGD / export mode) gets a vector offset beyond the descriptor's range and is dropped
    // by the buffer unit's bounds check (raw buffer: offset >= num_records; tensors are kept below 2 GB by the
    // dispatcher): no branch around any load or store, so the compiler's wait counters stay exact and a wait for an
    // older load never has to cover younger stores.
    const int f4_row = ldv >> 2, nf4 = kBU * f4_row;       // float4 per row / per unit (<= 1024: ldv <= 256)
    const int tq = min(tid, nf4 - 1);
    const int t_n = tq / f4_row, t_c0 = 4 * (tq - t_n * f4_row);   // this thread's item (row of the unit) and first column
    const bool t_ok = tid < nf4;
    const unsigned lane_off = (unsigned)tq * 16u;
    constexpr unsigned kOob = 0x80000000u;
    const unsigned tbytes = (unsigned)min((size_t)0x7FFFFFF0u, ((size_t)N + 2 * kTI) * ldv * sizeof(float));
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(a.V3a, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(a.gradV3 ? a.gradV3 : a.V3a, 0, tbytes, 0x00020000);
    auto unit_so = [&](int u) -> unsigned {               // byte offset of unit u (uniform)
        if (a.dbg_skip & 128) u = blockIdx.x;             // timing-only ablation: L2-resident units
        return (unsigned)((size_t)u * kBU * ldv) * 4u;
    };
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int u) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, unit_so(u), kAux));
    };
    // split form: the unit's gT image ([16][S2] dwords, 4 KB, padding included) as stored / re-read: thread t its t-th dword
    // (+ the 64 beyond 1024 by the first wave)
    constexpr int kGtD = kBU * S2;
    unsigned* gtg = reinterpret_cast<unsigned*>(a.Gt);
    unsigned g_nxt0 = 0u, g_nxt1 = 0u;
    auto load_gt = [&](int u) {
        const unsigned* src = gtg + (size_t)u * kGtD;
        g_nxt0 = __builtin_nontemporal_load(src + min(tid, kGtD - 1));
        g_nxt1 = __builtin_nontemporal_load(src + min(kBT + lane, kGtD - 1));
    };
    float4 p_cur, p_nxt, mreg, sreg;
    int unit = blockIdx.x;
    const int stride = gridDim.x;
    const int last_e = max(a.te.start[ntiles] - 1, 0);
    // (VECTOR loads through an index the compiler cannot prove uniform: a scalar load would sit in lgkmcnt, and the
    // LDS-only barriers - s_waitcnt lgkmcnt(0) - would wait out its L2 round trip once per unit)
    int oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    auto load_range = [&](int u, int& lo, int& hi) {
        const int tc = min(u >> 1, ntiles - 1) + oz;
        lo = a.te.start[tc]; hi = a.te.start[tc + 1];
        if (u >= nunits) hi = lo;
    };
    // The entry lists: ONE wave prefetches ranges (two units ahead) and the first 64 entries of the next unit's
    // tile (64 covers all but degenerate vocabularies: C3 has < 1 entry per tile) - as per-thread work of all 16 waves
    // these five small loads per wave were 80 of the 128 vector-memory instructions of a unit.  The range of the
    // running unit reaches the other waves through LDS (red[32..33]); entries beyond the first 64 are read in place.
    const int ewave = (a.dbg_skip & 64) ? kBW - 1 : 0;      // (A/B: 120-128 us with wave 0, 128-140 with the last wave)
    int ce0 = 0, ce1 = 0, ne0 = 0, ne1 = 0, fe0 = 0, fe1 = 0;
    int ent_b = 0, ent_n = 0; float ent_v = 0.f;
    auto load_entry = [&](int lo) {
        const int e = min(lo + lane, last_e);
        ent_b = a.te.eb[e]; ent_n = a.te.en[e]; ent_v = a.te.ev[e];
    };
    if (unit < nunits) {
        p_nxt = ld4(rP, unit);
        if (kIsOpt) load_gt(unit);
        if (kFwd && wave == ewave) {
            load_range(unit, ne0, ne1);
            load_range(unit + stride, fe0, fe1);
            load_entry(ne0);
        }
    }
    __syncthreads();
    // debug (AAE_DEC_TS): 100 MHz phase timestamps of workgroup 0, its eleventh unit
    int iter = 0;
    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[11] = wall_clock64();
    // (debug) arrival of every wave at barrier k of that unit: ts[16 + 16 k + wave]
    auto wstamp = [&](int k) { if (a.ts && blockIdx.x == 0 && lane == 0 && iter == 10) a.ts[16 + 16 * k + wave] = wall_clock64(); };
    auto stamp = [&](int k) { if (a.ts && blockIdx.x == 0 && tid == 0 && iter == 10) { a.ts[k] = wall_clock64(); if (k == 0 || k == 6) a.ts[8 + k / 6] = clock64(); } };

    for (; unit < nunits; unit += stride, ++iter) {
        const int i0 = unit * kBU, half = unit & 1;
        const int par = iter & 1;                         // which k-half of gK / v3T this unit fills (GEMM3 runs on every second unit)
        stamp(0);
        lds_barrier();                                  // the previous unit's readers of v3K / v3T / gK / gT are done
        stamp(14);
        // ---- S0: rotate the pipeline and request the next stage FIRST (the vector-memory unit takes ~0.5 us to accept
        // the 48 KB burst of a unit's loads: the image building below runs beside that), then this unit's V3a (fp32
        // registers) -> bf16 LDS images
        int my_b = 0, my_n = 0; float my_v = 0.f;
        if (kFwd && wave == ewave) {
            ce0 = ne0; ce1 = ne1; ne0 = fe0; ne1 = fe1;
            my_b = ent_b; my_n = ent_n; my_v = ent_v;
            if (lane == 0) { reinterpret_cast<int*>(red)[32] = ce0; reinterpret_cast<int*>(red)[33] = ce1; }
            load_range(unit + 2 * stride, fe0, fe1);
            load_entry(ne0);
        }
        p_cur = p_nxt;
        // (all three unconditional - the moment tensors exist in every mode: a load under a condition is waited for on
        // the spot and its result is carried in duplicate registers down both paths)
        p_nxt = ld4(rP, min(unit + stride, nunits - 1));
        if (kOpt) { mreg = ld4(rM, unit); sreg = ld4(rV, unit); }
        if (kIsOpt) {
            if (tid < kGtD) gT[tid] = g_nxt0;
            if (wave == 0 && kBT + lane < kGtD) gT[kBT + lane] = g_nxt1;
            load_gt(min(unit + stride, nunits - 1));
        }
        if (kFwd && t_ok) {
            float4 p = p_cur;
            if (i0 + t_n >= N) p = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<uint2*>(v3K + t_n * S1 + (t_c0 >> 1)) = make_uint2(bf16_pack(p.x, p.y), bf16_pack(p.z, p.w));
            unsigned short* t16 = reinterpret_cast<unsigned short*>(v3T) + 2 * vt_off(t_c0) + 16 * par + t_n;   // (t_c0 % 4 == 0: one block)
            t16[0] = bf16_bits(p.x);
            t16[32] = bf16_bits(p.y);
            t16[64] = bf16_bits(p.z);
            t16[96] = bf16_bits(p.w);
        }
        wstamp(4);
        stamp(15);
        wstamp(1);
        lds_barrier();
        stamp(1);

        // ---- GEMM1: logits[b][n]: wave w < 14 takes row block w >> 1 and every second k-step (w & 1) -> its half's raw
        // tile (fp32; the BCE phase adds the halves).  C map: row = 4 fk + r -> batch row, col = fr -> item
        if (kFwd && wave < 2 * kMB) {
            const int mb = wave >> 1, kh = wave & 1;
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
            const unsigned* pa = dhA + (16 * mb + fr) * S1 + 2 * fk + 16 * kh;
            const unsigned* pb = v3K + fr * S1 + 2 * fk + 16 * kh;
#pragma unroll
            for (int j = 0; j < (KC1 + 1) / 2; ++j)
                if (2 * j + kh < KC1)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_frag(pa, 0, 0, 2 * j, 0), bf_frag(pb, 0, 0, 2 * j, 0), c, 0, 0, 0);
            float* rw = raw + kh * (kGR * kBfSR) + (16 * mb + 4 * fk) * kBfSR + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[r * kBfSR] = c[r];
        }
        wstamp(2);
        if (kFwd) lds_barrier();
        stamp(2);
        // ---- BCE against a zero target, every thread: cell id = tid + 1024 j -> (b = id >> 4, n = id & 15); dL/dlogit
        // -> gK, gT (bf16), loss.  (As GEMM1's epilogue the 7 waves that own a row block did this alone.)
        if (kFwd) {
            unsigned short* k16 = reinterpret_cast<unsigned short*>(gK);
            unsigned short* t16 = reinterpret_cast<unsigned short*>(gT);
#pragma unroll
            for (int j = 0; j < (kGR * kBU + kBT - 1) / kBT; ++j) {
                const int id = tid + kBT * j, b = id >> 4, n = id & 15;
                if (b < B) {
                    float g, l;
                    bce_elem_t0(raw[b * kBfSR + n] + raw[kGR * kBfSR + b * kBfSR + n], a.gscale, g, l);
                    const bool ok = i0 + n < N;
                    g = ok ? g : 0.f;
                    loss += ok ? l : 0.f;
                    const unsigned short hb = bf16_bits(g);
                    k16[b * (2 * S3) + 16 * par + n] = hb;
                    t16[n * (2 * S2) + b] = hb;
                }
            }
        }
        wstamp(3);
        if (kFwd) lds_barrier();

        // ---- S2: the CSR entries of this half of the 32-item tile (non-zero targets) replace their cell's gradient
        // and loss term
        if (kFwd) {
            unsigned short* k16 = reinterpret_cast<unsigned short*>(gK);
            unsigned short* t16 = reinterpret_cast<unsigned short*>(gT);
            auto patch = [&](int b, int n32, float v) {
                if ((n32 >> 4) != half) return;
                const int n = n32 & 15;
                const float lg = raw[b * kBfSR + n] + raw[kGR * kBfSR + b * kBfSR + n];
                float g0, l0, g1, l1;
                bce_elem_t0(lg, a.gscale, g0, l0);
                bce_elem(lg, v, a.gscale, g1, l1);
                loss += l1 - l0;
                const unsigned short h = bf16_bits(g1);
                k16[b * (2 * S3) + 16 * par + n] = h;
                t16[n * (2 * S2) + b] = h;
            };
            if (wave == ewave && lane < ce1 - ce0) patch(my_b, my_n, my_v);
            const int r0 = reinterpret_cast<const int*>(red)[32], r1 = reinterpret_cast<const int*>(red)[33];
            for (int e = r0 + 64 + tid; e < r1; e += kBT) patch(a.te.eb[e], a.te.en[e], a.te.ev[e]);
        }
        if (kFwd) lds_barrier();
        stamp(3);
        if (MODE == kDecCrit) {      // the finished image -> a.Gt for the deferred launch (the store retires behind GEMM3)
            unsigned* dst = gtg + (size_t)unit * kGtD;
            if (tid < kGtD) dst[tid] = gT[tid];
            if (wave == 0 && kBT + lane < kGtD) dst[kBT + lane] = gT[kBT + lane];
        }

        // ---- GEMM2: dV3a^T[c][n] = sum_b dh2[b][c] G[b][n]; lane holds columns 16 cb + 4 fk ..+3 of item fr -> os
        if (kOpt && own) {
            f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KR; ++kc)
                g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dhT[kc], bf_frag(gT, fr, S2, kc, fk), g, 0, 0, 0);
            if (16 * cb + 4 * fk < kSO)
                *reinterpret_cast<float4*>(os + fr * kSO + 16 * cb + 4 * fk) = make_float4(g[0], g[1], g[2], g[3]);
        }
        stamp(4);
        // ---- GEMM3: dA2[b][c] += sum_n G[b][n] V3a[n][c] for the wave's column block, every row block, on every second
        // unit: two units fill the two halves of the 32-wide k-step of gK / v3T (which items share a k-step is free)
        auto gemm3 = [&]() {
            const bf16x8 vt = bf_frag(v3T + vt_off(16 * cb + fr), 0, 0, 0, fk);
            const unsigned* pg = gK + fr * S3 + 2 * fk;
#pragma unroll
            for (int q = 0; q < kMB; ++q)       // (every row block: rows >= B of gK are zero - no branch between the MFMAs)
                acc3[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_frag(pg, 0, 0, 0, 0) , vt, acc3[q], 0, 0, 0), pg += 16 * S3;
        };
        if (kFwd && own && par) gemm3();
        wstamp(0);
        if (kOpt) lds_barrier();                         // os complete
        // ---- S5: optimiser on the unit in row-major order.  The three stores are issued on every path (see above).
        if (kOpt) {
            const float4 g = *reinterpret_cast<const float4*>(os + t_n * kSO + t_c0);
            const unsigned so = unit_so(unit);
            const bool valid = t_ok && i0 + t_n < N;
            float4 p = p_cur, mm = mreg, vv = sreg;
            if (sc.is_sgd) {                              // (wave-uniform; arithmetic only)
                p.x += sc.neg_step_size * g.x; p.y += sc.neg_step_size * g.y;
                p.z += sc.neg_step_size * g.z; p.w += sc.neg_step_size * g.w;
            } else {
                adam_only(p.x, mm.x, vv.x, g.x, sc); adam_only(p.y, mm.y, vv.y, g.y, sc);
                adam_only(p.z, mm.z, vv.z, g.z, sc); adam_only(p.w, mm.w, vv.w, g.w, sc);
            }
            const float4 out = do_adam ? p : g;
            const unsigned vo = valid ? lane_off : kOob;
            const unsigned vo2 = (valid && do_adam && !sc.is_sgd) ? lane_off : kOob;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, out), do_adam ? rP : rG, vo, so, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, mm), rM, vo2, so, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vv), rV, vo2, so, kAux);
        }
        stamp(6);
    }
    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[7] = wall_clock64();
    if (!kFwd) return;
    if (iter & 1) {                                     // an odd number of units: the last one's half is still to be added
        lds_barrier();
        for (int i = tid; i < kGR * 8; i += kBT) gK[(i >> 3) * S3 + 8 + (i & 7)] = 0u;      // k-half 1 of gK <- 0
        lds_barrier();
        if (own) {
            const bf16x8 vt = bf_frag(v3T + vt_off(16 * cb + fr), 0, 0, 0, fk);
            const unsigned* pg = gK + fr * S3 + 2 * fk;
#pragma unroll
            for (int q = 0; q < kMB; ++q)
                acc3[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_frag(pg, 0, 0, 0, 0) , vt, acc3[q], 0, 0, 0), pg += 16 * S3;
        }
    }

    // ---- dA2 partial of this workgroup -> its slab; loss partial
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    if (own) {
#pragma unroll
        for (int q = 0; q < kMB; ++q) {
            const int rb = q * 16 + fk * 4, cc = cb * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rb + r < B && cc < a.ld_slab) slab[(size_t)(rb + r) * a.ld_slab + cc] = acc3[q][r];
        }
    }
    loss = wave_sum(loss);
    __syncthreads();
    if (lane == 0) red[wave] = loss;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < kBW; ++w) s += red[w];
        a.partials[blockIdx.x] = s;
        if (a.ts && blockIdx.x == 0) { a.ts[12] = wall_clock64(); a.ts[13] = (unsigned long long)iter; }
    }
}

}  // namespace aae
