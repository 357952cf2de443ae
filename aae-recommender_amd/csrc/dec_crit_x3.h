// Critical launch of the fused decoder output layer (dec_fused.h, MODE kDecCrit: logits, BCE, dL/d(hidden), the stored
// dL/dlogits tiles) with its fp32 products EMULATED on the bf16 matrix cores.
//
// gfx950 has no TF32 and its fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate.  Every fp32 operand
// x is therefore split, on its way into LDS / registers, into three bf16 terms x = x1 + x2 + x3 (round to nearest even,
// each residual is exact in fp32: 8 + 8 + 8 significand bits cover fp32's 24), and a product of two operands becomes
// the six leading cross terms
//     a * b  ~  a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1        (dropped: a2 b3 + a3 b2 + a3 b3 < 2^-23 |a b|)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 instructions of 16 cycles per 32 k against 8 of 32 cycles - 2.7x
// the fp32 matrix rate at fp32-level error (every bf16 x bf16 product is exact in fp32; what differs from the fp32 MFMA is
// the summation order and the dropped 2^-24-relative terms - both below the fp32 rounding of the sum itself).
// dtype stays f32: master weights, Adam, the BCE epilogue and all accumulators are fp32, and the reference's fixtures hold
// at unchanged tolerances.
//
// Per tile of 32 items, one persistent 1024-thread workgroup per CU:
//   S0     V3a tile (fp32 registers, requested one tile ahead) -> three bf16 images v3K[t][n][k] (k-contiguous rows);
//          the tile's CSR entries (non-zero BCE targets) -> tgt[b][n]
//   GEMM1  logits[b][n] = dh2 * V3a^T.  dh2 lives in REGISTERS for the whole kernel, already split, as the A fragments of
//          the wave's row block (wave = (row block, k half): 48 VGPRs) - no LDS image of it (three of them would be 156
//          KB) and no LDS traffic for half of GEMM1's operands; B fragments = one conflict-free 16-byte LDS read per term
//   BCE    every thread two adjacent cells: the halves of the k split are added, zero-target form or - where tgt names a
//          target - the reference's exact form; dL/dlogit -> the stored tile in HBM (fp32 [B][32], what the deferred
//          optimiser launch reads) and, split, -> gK[t][b][n]; loss
//   GEMM3  dA2[b][c] += G * V3a.  A fragments from gK; the B operand wants k = item contiguous per column c, i.e. the
//          TRANSPOSE of the v3K rows: read with ds_read_b64_tr_b16 (gfx950's transposing LDS read) from the same images -
//          no second, transposed image and none of the 2-byte scatter writes it would need
// 4 LDS-only barriers per tile.
#pragma once
#include "dec_fused_bf16.h"

namespace aae {

typedef short s16x4_t __attribute__((ext_vector_type(4)));

// the 3-term split of a pair of fp32 values and the six-term product: gemm_f32.h (x3_split_pair, x3_mfma)
// (one: bf16 mode - the operand IS its first term, round-to-nearest-even; the other two images hold zeros)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3, bool one) {
    x3_split_pair(a, b, p1, p2, p3);
    if (one) { p2 = 0u; p3 = 0u; }
}
__device__ __forceinline__ f32x4 mfma_x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) { return x3_mfma(a, b, c); }
template <bool ONE>
__device__ __forceinline__ f32x4 mfma_xt(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
    if constexpr (ONE) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    else return x3_mfma(a, b, c);
}

// Row stride (dwords) of a k-contiguous bf16 image with kc 32-wide k-steps, read with ONE ds_read_b128 per fragment (8
// consecutive k per lane): 16 kc + 8 = a stride of 32 bytes mod 64.  ds_read_b128 is serviced in four groups of 16 lanes
// (MI355X_MICROARCH.md, LDS), each holding every fragment row fr once - half of them with the group's k-eighth fk, half with
// fk ^ 1 -, and with stride / 16 = 2 mod 4 the rows of the first half fall on the even 16-byte slots of the 256-byte bank
// row and those of the second on the odd ones: conflict-free.  (Two ds_read_b64 per fragment get fused by hipcc into
// ds_read2_b64, which runs at half the rate.)
__host__ __device__ constexpr int x3_stride(int kc) { return 16 * kc + 8; }
// one 16x16x32 fragment: row `row`, k = 32 kc + 8 fk + {0..7}
__device__ __forceinline__ bf16x8 x3_frag(const unsigned* img, int row, int S, int kc, int fk) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(img + row * S + 16 * kc + 4 * fk));
}
constexpr int kXRS = 33;       // row stride (floats) of the raw-logit halves [b][n]
constexpr int kXT = 32;        // row stride (floats) of the target tile [b][n]

#ifndef AAE_CRIT_BAL
#define AAE_CRIT_BAL 1
#endif
constexpr bool kCritBal = AAE_CRIT_BAL != 0;   // hidden 200: the products dealt to the waves by SIMD (r6, see the kernel; 0: r5's map, A/B builds)
#ifndef AAE_CRIT_G1PF
#define AAE_CRIT_G1PF 1
#endif
constexpr bool kCritG1Pf = AAE_CRIT_G1PF != 0;  // GEMM1 with the next k-step's first fragments requested ahead (r6, see the kernel)
constexpr int kBalRows = 4;                     // ... the row blocks from here on of column blocks 12 / 10 / 11 go to waves 13 / 14 / 15
constexpr int kXRegSteps = 3;  // k-steps of a wave's dh2 fragments kept in registers (12 VGPRs each); further ones live in LDS
inline size_t dec_crit_x3_lds_bytes(int NB) {
    const int KC1 = (NB + 1) / 2, NKS = (KC1 + 1) / 2, S1 = x3_stride(KC1), S3 = x3_stride(1);
    const int lsteps = NKS > kXRegSteps ? NKS - kXRegSteps : 0;
    return sizeof(float) * ((size_t)3 * kTI * S1 + (size_t)2 * kGR * kXRS + (size_t)3 * kGR * S3 + (size_t)kGR * kXT + 64 +
                            (size_t)lsteps * kMB * 3 * 64 * 4 + (lsteps ? (size_t)2 * kMB * 64 * 4 : 0));
}

// TS: the debug build with in-kernel stamps (AAE_DEC_TS=x3).  The production build carries none: every stamp site is a lane
// test + two exec-mask instructions + a branch in all 16 waves, eight of them per tile - a quarter of the tile loop's ~100 scalar
// instructions per wave, on a CU whose one scalar unit serves all its waves.
// ONE (r4): bf16 mode's instantiation - every operand IS its first term (round-to-nearest-even bf16), so the other two images
// are neither built nor read and a product is ONE matrix instruction instead of six (the run-time form, a.one_term on the
// three-term instantiation, multiplies five zero terms: measured in r3 as 36.7 vs 27.9 us at C2).
template <int NB, bool TS = false, bool ONE = false, bool WIN = false>   // NB = ceil((h + 1) / 16) column blocks; WIN: dec.lin3 beyond 2^31 bytes (dec_fused.h X3WindowT)
__global__ __launch_bounds__(kNT) void dec_crit_x3_kernel(DecFusedArgs a) {
    const bool one = ONE || a.one_term != 0;
    constexpr int NT = ONE ? 1 : 3;            // term images in use
    constexpr int KC1 = (NB + 1) / 2;          // 32-wide k-steps over the h + 1 hidden columns
    constexpr int NKS = (KC1 + 1) / 2;         // ... per k half
    constexpr int NKR = NKS > kXRegSteps ? kXRegSteps : NKS;   // ... of them in registers; the others' fragments in LDS, a private
    constexpr int NKL = NKS - NKR;             // [step][term][lane] block per wave (12 VGPRs per step cost a spill at 128 -
                                               // and a scratch reload waits for the V3a prefetch in flight; an LDS read does not)
    constexpr int S1 = x3_stride(KC1), S3 = x3_stride(1);
    static_assert(NB <= kNW && 2 * kMB <= kNW, "one column block / one (row block, k half) per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* v3K = reinterpret_cast<unsigned*>(lds);          // [3][32][S1]  V3a tile, k = hidden column
    float* raw = reinterpret_cast<float*>(v3K + 3 * kTI * S1); // [2][kGR][kXRS] the two k halves of the logits (fp32)
    unsigned* gK = reinterpret_cast<unsigned*>(raw + 2 * kGR * kXRS);   // [3][kGR][S3] dL/dlogits, k = item
    float* tgt = reinterpret_cast<float*>(gK + 3 * kGR * S3);  // [kGR][kXT]   non-zero BCE targets of the tile (else 0)
    float* red = tgt + kGR * kXT;                              // [64]
    u32x4_t* dAl = reinterpret_cast<u32x4_t*>(red + 64);        // [kMB][NKL][3][64] spilled-by-design dh2 fragments (k half 0 only)
    // (late r4) ... and, where a k-step lives there, the THIRD term of the last register step of both k halves [2][kMB][64]:
    // with the next tile's S0 in the GEMM3 phase the allocator was 4 registers short and chose a dh2 fragment to keep in
    // scratch memory - this is the same choice with an LDS read instead of a scratch reload
    constexpr bool XT = NKL > 0 && !ONE;
    u32x4_t* dAx = dAl + kMB * NKL * 3 * 64;

    // row blocks of one launch (item slices of the data-parallel scheme, dec_fused.h): this workgroup's block and tiles
    const int nblk = a.nblk > 1 ? a.nblk : 1;
    const int blk = nblk > 1 ? (int)blockIdx.x % nblk : 0, wgi = nblk > 1 ? (int)blockIdx.x / nblk : (int)blockIdx.x;
    const int wgs = nblk > 1 ? (int)gridDim.x / nblk : (int)gridDim.x;
    if (nblk > 1 && wgi >= wgs) return;
    const int erow0 = nblk > 1 ? blk * a.Bb : a.erow0;
    const int B = nblk > 1 ? min(a.Bb, a.B - erow0) : a.B;
    const float* dh2_blk = nblk > 1 ? a.dh2 + (size_t)erow0 * a.ldh : a.dh2;
    float* slabs_blk = nblk > 1 ? a.slabs + (size_t)erow0 * a.ld_slab : a.slabs;
    float* Gt_blk = nblk > 1 ? a.Gt + (size_t)blk * ((a.N + kTI - 1) / kTI) * a.Bb * kTI : a.Gt;

    const int tid = threadIdx.x, lane = tid & 63;
    if (TS && a.ts && blockIdx.x == 0 && tid == 0) a.ts[10] = wall_clock64();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldv = a.ldv, N = a.N;
    const int ntiles = (N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;                       // float4 slots per thread of a tile span (tile_f4 <= 2048)

    // ---- the parameter stream, FIRST: the first tile's parameters and entries travel (an HBM round trip, ~2 us) while the
    // workgroup zeroes its images and splits its dh2 fragments below.
    // The stream: tensor base in a buffer descriptor (scalar registers), the tile's byte offset in a scalar
    // register, this thread's slot offset in ONE vector register (+ an immediate); reads beyond the tensor return zero
    // (the dispatcher keeps fused layers below 2 GB)
    // (r5: the descriptor covers a WINDOW of the tensor that starts at tile seg0 - the range check looks at the sum of the scalar
    //  and the vector offset and a descriptor spans < 2^31 bytes, while PubMed's / ACM's dec.lin3 of the reference's dataset table
    //  is 2.4 GB at hidden 200; the window moves up, at a tile's start, once that tile lies 2^30 bytes into it: x3_window)
    const unsigned lane_off = (unsigned)tid * 16u;
    const unsigned tile_bytes = (unsigned)(kTI * ldv) * 4u;
    X3WindowT<WIN> win(N, ldv, tile_bytes);
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a);
    // (a thread's second slot lies beyond the tile's span for most threads - 1632 float4 at hidden 200: slots 1632 .. 2047 are the
    //  NEXT tile's first rows -, so this launch reads 1.25x the layer, profiles/r5_pmc_calibration.txt.  Masking those lanes, as
    //  the deferred launch does since r5, costs this kernel a vector register it does not have: 12 bytes of scratch per lane)
    // (r6) which waves take a SECOND float4 slot of the 1632-slot span (9.5 of the 16 waves): the image build follows GEMM3 in its
    // phase, and with BAL waves 10-15 have three or four of its products where the others have seven, SIMD classes 2 and 3 the
    // fewest - waves 10 .. 15, 2, 3, 6, 7 in place of 0 .. 9.  slot1 = 1024 + 64 x rank + lane; the wave's distance from its natural
    // slot rides in the scalar offset of the load.
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rank1 = NB == 13 && kCritBal ? (int)((0x543210FF98FF76FFull >> (4 * wave_u)) & 15ull) : wave_u;
    const int slot1 = kNT + 64 * rank1 + (tid & 63);                                    // (>= tile_f4: no second slot)
    const unsigned so1 = (unsigned)((kNT + 64 * rank1 - 64 * wave_u) * 16);              // slot1's bytes beyond lane_off
    auto load_span = [&](int tile, float4* r) {
        r[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, lane_off, win.so(tile), 0));
        r[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, lane_off, win.so(tile) + so1, 0));
    };
    float4 vreg[NV];
    int tile = wgi;
    const int stride = wgs;
    const int last_e = max(a.te.start[ntiles] - 1, 0);
    int ozr;
    asm volatile("v_mov_b32 %0, 0" : "=v"(ozr));
    auto load_range = [&](int t, int& lo, int& hi) {       // (vector loads: a scalar load would sit in lgkmcnt, see dec_fused.h)
        const int tc = min(t, ntiles - 1) + ozr;
        lo = a.te.start[tc]; hi = a.te.start[tc + 1];
        if (t >= ntiles) hi = lo;
    };
    int ce0 = 0, ce1 = 0, ne0 = 0, ne1 = 0, fe0 = 0, fe1 = 0;
    int ent_bn = 0; float ent_v = 0.f;          // this thread's entry of the NEXT tile: (doc << 5) | item-in-tile, value
    auto load_entry = [&](int lo) {
        const int e = min(lo + tid, last_e);
        ent_bn = (a.te.eb[e] << 5) | a.te.en[e]; ent_v = a.te.ev[e];
    };
    if (tile < ntiles) {
        load_span(tile, vreg);
        load_range(tile, ne0, ne1);
        load_range(tile + stride, fe0, fe1);
        ne0 = __builtin_amdgcn_readfirstlane(ne0); ne1 = __builtin_amdgcn_readfirstlane(ne1);
        load_entry(ne0);
    }
    // ---- the deferred launch's operands, set aside (late join, abi_output_layer.h): it reads dh2 and the layer's step
    // scalars when it starts, and the NEXT step's forward pass - which it may now overlap - rewrites both.  One row per
    // workgroup (single row block only: the host passes NULL otherwise).
    if (a.dh2_snap && (int)blockIdx.x < a.B) {
        if (tid < a.ldh / 4)
            reinterpret_cast<float4*>(a.dh2_snap + (size_t)blockIdx.x * a.ldh)[tid] = reinterpret_cast<const float4*>(a.dh2 + (size_t)blockIdx.x * a.ldh)[tid];
        if (blockIdx.x == 0 && tid == kNT - 1) *a.sc_snap = *a.sc;
    }
    // ---- once per workgroup: zero the images (k padding of v3K, rows >= B of gK, the target tile)
    for (int i = tid; i < 3 * kTI * S1 + 2 * kGR * kXRS + 3 * kGR * S3 + kGR * kXT; i += kNT) v3K[i] = 0u;

    // ---- dh2 -> split A fragments in registers.  GEMM1: wave w < 14 = (row block w % 7, k half w / 7) takes the k-steps
    // kc = kh, kh + 2, ...; lane (fr, fk) of a fragment holds row 16 mb + fr, k = 32 kc + 8 fk + {0..7}
    // (r6, BAL) Which wave takes which (row block, k half) unit decides what each SIMD's matrix pipe gets: waves w, w + 4, w + 8,
    // w + 12 share a SIMD, and at hidden 200 a unit of k half 0 is 48 matrix instructions per tile, one of k half 1 is 36.
    // wave = row block + 7 x k half (r3-r5) dealt the four SIMDs 168 / 168 / 132 / 120; the table below 144 / 144 / 156 / 144:
    //   SIMD class 0 (waves 0 4 8 12): k half 0 of row blocks 0 1 2, -        class 1 (1 5 9 13): k half 0 of 3 4 5, -
    //   class 2 (2 6 10 14): k half 0 of row block 6, k half 1 of 0 1 2       class 3 (3 7 11 15): k half 1 of 3 4 5 6
    constexpr bool BAL = NB == 13 && kCritBal;
    const int unit = BAL ? (int)((0xD9FFC852B741A630ull >> (4 * wave)) & 15ull) : wave;
    const bool g1 = unit < 2 * kMB;
    const int mb1 = unit % kMB, kh = g1 ? unit / kMB : 1;
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
            for (int t = 0; t < NT; ++t) {
                const u32x4_t v = {p[t][0], p[t][1], p[t][2], p[t][3]};
                if (XT && j == NKR - 1 && t == 2) { if (g1) dAx[(kh * kMB + mb1) * 64 + lane] = v; }
                else if (j < NKR) dA[j < NKR ? j : 0][t] = __builtin_bit_cast(bf16x8, v);
                else if (g1 && kh == 0) dAl[((mb1 * NKL + (j - NKR)) * 3 + t) * 64 + lane] = v;
            }
        }
    }
    const int nks = (KC1 - kh + 1) / 2;         // k-steps this wave really has (the rest of dA is zero and never used)

    // GEMM3: wave w < NB owns column block w for every row block
    // (BAL: 13 column blocks x 7 row blocks = 91 products per tile - one owner per column block dealt them 28 / 21 / 21 / 21 to
    //  the SIMDs; the upper row blocks of column blocks 12, 10, 11 now go to the three waves that had none: 25 / 24 / 21 / 21.
    //  Every accumulator still sums the tiles in order: same bits.)
    const bool own = BAL || wave < NB;
    const int cb = BAL && wave >= NB ? (wave == 13 ? 12 : wave == 14 ? 10 : 11) : min(wave, NB - 1);
    const int q_lo = BAL && wave >= NB ? kBalRows : 0, q_hi = BAL && wave >= 10 && wave < NB ? kBalRows : kMB;
    f32x4 acc3[kMB];
#pragma unroll
    for (int q = 0; q < kMB; ++q) acc3[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float loss = 0.f;

    int s_rc = 0;                               // this thread's slots of a tile span: item row * 64 + float4 column, 16 bits each
    static_assert(NV == 2, "two slots in one register");
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int fc = min(j ? slot1 : tid, tile_f4 - 1), row = fc / f4_per_row;
        s_rc |= (row * 64 + (fc - row * f4_per_row)) << (16 * j);
    }
    const int g_f4 = B * (kTI / 4);             // float4 per stored dL/dlogits tile
    const unsigned gbytes = (unsigned)min((size_t)0x7FFFFFF0u, (size_t)ntiles * g_f4 * 16);
    const __amdgpu_buffer_rsrc_t rGt = __builtin_amdgcn_make_buffer_rsrc(Gt_blk, 0, gbytes, 0x00020000);
    typedef unsigned int fu32x2 __attribute__((ext_vector_type(2)));
    int iter = 0;
    if (TS && a.ts && blockIdx.x == 0 && tid == 0) a.ts[11] = wall_clock64();
    auto stamp = [&](int k) { if (TS && a.ts && blockIdx.x == 0 && tid == 0 && iter == 5) { a.ts[k] = wall_clock64(); if (k == 0 || k == 6) a.ts[8 + k / 6] = clock64(); } };
    __syncthreads();

    // ---- S0 of tile X: its V3a (in vreg, requested a tile ahead) -> the three bf16 images; its CSR entries -> tgt; the
    // requests for the tile after it.  Since late r4 it runs in the phase of the PREVIOUS tile's GEMM3 (below): that product's
    // only reads of v3K are one transposed fragment set per wave, taken in front of BCE - behind the barrier that closes BCE
    // the images, like tgt, belong to the next tile.  One barrier and the S0 phase less per tile.
    auto stage = [&](int X) {
        const int x0 = X * kTI;
        const bool ragged = x0 + kTI > N;       // (uniform) only the vocabulary's last tile has rows beyond N
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if ((j ? slot1 : tid) < tile_f4) {
                float4 p = vreg[j];
                int src;                                // (opaque copy: what is derived from it is recomputed here, not hoisted
                asm volatile("v_bfe_u32 %0, %1, %2, 16" : "=v"(src) : "v"(s_rc), "n"(16 * j));     //  out of the tile loop and spilled)
                if (ragged && x0 + (src >> 6) >= N) p = make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned q0[3], q1[3];
                split3_pair(p.x, p.y, q0[0], q0[1], q0[2], one);
                split3_pair(p.z, p.w, q1[0], q1[1], q1[2], one);
                unsigned* d = v3K + (src >> 6) * S1 + 2 * (src & 63);
#pragma unroll
                for (int t = 0; t < NT; ++t) *reinterpret_cast<uint2*>(d + t * (kTI * S1)) = make_uint2(q0[t], q1[t]);
            }
        }
        // (the ranges are workgroup-uniform values that arrive through vector loads: the pair requested a whole tile ago moves
        //  to scalar registers here - two VGPRs less across the tile loop)
        ce0 = ne0; ce1 = ne1; ne0 = __builtin_amdgcn_readfirstlane(fe0); ne1 = __builtin_amdgcn_readfirstlane(fe1);
        {
            const int rb = (ent_bn >> 5) - erow0;       // (row-blocked launches: entries of other row blocks are not ours)
            if (tid < ce1 - ce0 && (unsigned)rb < (unsigned)B) tgt[rb * kXT + (ent_bn & 31)] = ent_v;
            for (int e = ce0 + kNT + tid; e < ce1; e += kNT) {      // tiles with more than 1024 entries (tiny vocabularies)
                const int r2 = a.te.eb[e] - erow0;
                if ((unsigned)r2 < (unsigned)B) tgt[r2 * kXT + a.te.en[e]] = a.te.ev[e];
            }
        }
        load_range(X + 2 * stride, fe0, fe1);
        load_entry(ne0);
    };
    if (tile < ntiles) stage(tile);
    lds_barrier();

    for (; tile < ntiles; tile += stride, ++iter) {
        const int i0 = tile * kTI;
        if (win.moves(tile)) rP = win.desc(a.V3a);          // (uniform; never for a layer below 2^30 bytes)
        // A zero the compiler cannot see through: the LDS operand addresses of the phases are built from it, so they are
        // recomputed per tile (a few VALU) instead of being hoisted out of the tile loop and held in registers across
        // every phase - at 128 VGPRs that hoisting spilled, and a scratch reload waits for every older global load in
        // flight (the next tile's V3a prefetch)
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int frz = fr + oz, fkz = fk + oz;
        int lanez;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lanez) : "v"(lane));

        // ---- GEMM1: logits of the wave's row block x both item halves over its k-steps -> its half's raw tile
        auto gemm1 = [&]() {
            float* rw = raw + kh * (kGR * kXRS) + (16 * mb1 + 4 * fkz) * kXRS + frz;    // C map: row = 4 fk + r, col = fr
#pragma unroll
            for (int nb2 = 0; nb2 < 2; ++nb2) {         // (one item half at a time: 12 fragment registers live, not 24)
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NKS; ++j)
                    if (j < nks) {
                        bf16x8 bb[3];
#pragma unroll
                        for (int t = 0; t < NT; ++t) bb[t] = x3_frag(v3K + t * (kTI * S1), 16 * nb2 + frz, S1, kh + 2 * j, fkz);
                        if (XT && j == NKR - 1) {
                            bf16x8 aa[3] = {dA[j < NKR ? j : 0][0], dA[j < NKR ? j : 0][1],
                                            __builtin_bit_cast(bf16x8, dAx[(kh * kMB + mb1) * 64 + lanez])};
                            c = mfma_xt<ONE>(aa, bb, c);
                        } else if (j < NKR) c = mfma_xt<ONE>(dA[j < NKR ? j : 0], bb, c);
                        else {
                            bf16x8 al[3];
#pragma unroll
                            for (int t = 0; t < NT; ++t) al[t] = __builtin_bit_cast(bf16x8, dAl[((mb1 * NKL + (j - NKR)) * 3 + t) * 64 + lanez]);
                            c = mfma_xt<ONE>(al, bb, c);
                        }
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r * kXRS + 16 * nb2] = c[r];
            }
        };
        // (r6, kCritG1Pf) the same products with the NEXT k-step's first B fragment requested in front of the current step's six
        // matrix instructions (NK = the wave's k-steps, a compile-time count): a wave's chain was read - wait ~100 clocks -
        // multiply, and with the SIMD's four waves mostly in step the matrix pipe idled a quarter of the phase (in-kernel stamps:
        // 1.68 us for 1.2 us of pipe time; 1.52 with this).  Only the FIRST term's fragment - what the step's first instruction
        // needs - travels ahead: all three (12 registers) cost 20 spilled ones, and the third A term of the steps that keep it in
        // LDS, or GEMM3's fragments a row block ahead, bought nothing (same-box, ms/step: 0.2310 | 0.2313 | 0.2321).
        // Same instruction order per accumulator: same bits.
        auto gemm1_pf = [&](auto NKc) {
            constexpr int NK = decltype(NKc)::value;
            float* rw = raw + kh * (kGR * kXRS) + (16 * mb1 + 4 * fkz) * kXRS + frz;
            // (only the FIRST term's fragment travels a step ahead - the one the step's first matrix instruction needs; the other
            //  two are requested at the step's start and arrive behind that instruction: 4 registers instead of 12)
            bf16x8 b0n = x3_frag(v3K, frz, S1, kh, fkz);
#pragma unroll
            for (int nb2 = 0; nb2 < 2; ++nb2) {
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    const int step = nb2 * NK + j;
                    bf16x8 bb[3];
                    bb[0] = b0n;
#pragma unroll
                    for (int t = 1; t < NT; ++t) bb[t] = x3_frag(v3K + t * (kTI * S1), 16 * nb2 + frz, S1, kh + 2 * j, fkz);
                    if (step + 1 < 2 * NK) b0n = x3_frag(v3K, 16 * ((step + 1) / NK) + frz, S1, kh + 2 * ((step + 1) % NK), fkz);
                    __builtin_amdgcn_sched_barrier(0);
                    if (XT && j == NKR - 1) {
                        bf16x8 aa[3] = {dA[j < NKR ? j : 0][0], dA[j < NKR ? j : 0][1],
                                        __builtin_bit_cast(bf16x8, dAx[(kh * kMB + mb1) * 64 + lanez])};
                        c = mfma_xt<ONE>(aa, bb, c);
                    } else if (j < NKR) c = mfma_xt<ONE>(dA[j < NKR ? j : 0], bb, c);
                    else {
                        bf16x8 al[3];
#pragma unroll
                        for (int t = 0; t < NT; ++t) al[t] = __builtin_bit_cast(bf16x8, dAl[((mb1 * NKL + (j - NKR)) * 3 + t) * 64 + lanez]);
                        c = mfma_xt<ONE>(al, bb, c);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r * kXRS + 16 * nb2] = c[r];
            }
        };
        // ---- BCE: thread -> cells (b, n2), (b, n2 + 1) of pair id tid + 1024 j; the stored tile, the split G images, loss
        float lprod;                            // the product of this thread's 1 + e factors of the tile (BCE below)
        auto bce = [&](int pid0) {              // (the 64 cell pairs from pid0 on: rows pid0 / 16 .. + 3)
            const int pid = pid0 + lanez, b = pid >> 4, n2 = (pid & 15) * 2;
            if (b < B) {
                const float l0 = raw[b * kXRS + n2] + raw[kGR * kXRS + b * kXRS + n2];
                const float l1 = raw[b * kXRS + n2 + 1] + raw[kGR * kXRS + b * kXRS + n2 + 1];
                const float2 tt = *reinterpret_cast<const float2*>(tgt + b * kXT + n2);
                // the zero-target form for both cells; the pair of a target (2 400 of 10 M cells) takes ONE branch afterwards
                // (three lane-dependent branches per pair cost every wave ~10 scalar instructions per tile more)
                // (r6) the loss of a cell in parts (gemm_f32.h bce_elem_t0_parts): only the workgroup's SUM leaves the kernel, so the
                // logarithm of the zero-target form is taken once per thread and tile, of the product of its cells' 1 + e (12 -> 9
                // quarter-rate instructions per thread and tile, the ln 2 product once instead of four times: 73.6 -> 72.7 us per
                // launch over four alternations; every parameter bit unchanged - the gradient does not come through here -, the
                // reported loss moves in its 8th digit)
                float gA, lA, fA, gB, lB, fB;
                bce_elem_t0_parts(l0, a.gscale, gA, lA, fA);
                bce_elem_t0_parts(l1, a.gscale, gB, lB, fB);
                if (tt.x != 0.f || tt.y != 0.f) {
                    if (tt.x != 0.f) { bce_elem(l0, tt.x, a.gscale, gA, lA); fA = 1.f; }
                    if (tt.y != 0.f) { bce_elem(l1, tt.y, a.gscale, gB, lB); fB = 1.f; }
                    *reinterpret_cast<float2*>(tgt + b * kXT + n2) = make_float2(0.f, 0.f);
                }
                if (i0 + kTI > N) {                     // (uniform: the vocabulary's last tile alone has cells beyond N)
                    if (i0 + n2 >= N) { gA = 0.f; lA = 0.f; fA = 1.f; }
                    if (i0 + n2 + 1 >= N) { gB = 0.f; lB = 0.f; fB = 1.f; }
                }
                loss += lA + lB;
                lprod *= fA * fB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fu32x2, make_float2(gA, gB)), rGt, (unsigned)pid * 8u,
                                                      (unsigned)tile * (unsigned)g_f4 * 16u, 0);
                unsigned q[3];
                split3_pair(gA, gB, q[0], q[1], q[2], one);
#pragma unroll
                for (int t = 0; t < NT; ++t) gK[t * (kGR * S3) + b * S3 + (n2 >> 1)] = q[t];
            }
        };
        // ---- GEMM3's B operand (dA2[b][c] += sum_n G[b][n] V3a[n][c]; wave w < NB owns column block w for every row block):
        // the transpose of 2 x (4 items x 16 columns) of the v3K rows per term - ds_read_b64_tr_b16, lane 4 q + p of a 16-lane
        // group names row q, columns 4 p .. 4 p + 3 of its block.  Read in front of the barrier behind which the images are
        // the next tile's.
        bf16x8 vt[3];
        auto read_vt = [&]() {
            const int lz = lane + oz, q = (lz >> 2) & 3, p = lz & 3;
            const unsigned* base = v3K + (8 * fkz + q) * S1 + 8 * cb + 2 * p;        // (16 cb + 4 p) bf16 = 8 cb + 2 p dwords
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4_t __attribute__((address_space(3)))*)(base + t * (kTI * S1)));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4_t __attribute__((address_space(3)))*)(base + t * (kTI * S1) + 4 * S1));
                typedef short s16x8_t __attribute__((ext_vector_type(8)));
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                vt[t] = __builtin_bit_cast(bf16x8, v);
            }
        };
        // ---- GEMM3 over the row blocks [Q0, Q1).  k-order of a fragment (both operands): element j = item 8 fk + j - what
        // x3_frag reads from the gK rows (rows >= B of gK are zero - no branch between the MFMAs)
        auto gemm3 = [&](auto Q0, auto Q1) {
#pragma unroll
            for (int q = decltype(Q0)::value; q < decltype(Q1)::value; ++q) {
                if (BAL && (q < q_lo || q >= q_hi)) continue;     // (uniform: the row blocks of this column block another wave owns)
                bf16x8 ga[3];
#pragma unroll
                for (int t = 0; t < NT; ++t) ga[t] = x3_frag(gK + t * (kGR * S3), 16 * q + frz, S3, 0, fkz);
                acc3[q] = mfma_xt<ONE>(ga, vt, acc3[q]);
            }
        };
        typedef std::integral_constant<int, 0> Q_0;
        typedef std::integral_constant<int, kMB> Q_E;
        constexpr int NJ = (kGR * (kTI / 2) + kNT - 1) / kNT;
        stamp(0);

        stamp(14);
        stamp(1);
        if (kCritG1Pf) {
            if (g1) { if (nks == NKS) gemm1_pf(std::integral_constant<int, NKS>()); else gemm1_pf(std::integral_constant<int, (NKS > 1 ? NKS - 1 : 1)>()); }
        } else if (g1) gemm1();
        // the next tile's V3a: requested HERE, behind the phase with the most live registers (its 8 would be the ones that
        // spill), BCE and GEMM3 ahead of its use in stage() - ~2 us, an HBM round trip
        load_span(min(tile + stride, ntiles - 1), vreg);
        lds_barrier();
        stamp(2);
        lprod = 1.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bce(64 * wave + kNT * j);
        loss += log_pos(lprod);
        if (own) read_vt();
        lds_barrier();
        stamp(3);
        // ---- GEMM3 of this tile and the next tile's S0 (its images, its targets, the requests behind it) in one phase
        if (own) gemm3(Q_0(), Q_E());
        __builtin_amdgcn_sched_barrier(0);
        stage(tile + stride);
        stamp(4);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();                          // gK's readers are done; the next tile's images and targets are in place
        __builtin_amdgcn_sched_barrier(0);
        stamp(5);
        stamp(6);
    }
    if (TS && a.ts && blockIdx.x == 0 && tid == 0) a.ts[7] = wall_clock64();

    // ---- dA2 partial of this workgroup -> its slab; loss partial
    float* slab = slabs_blk + (size_t)wgi * a.slab_stride;
    if (own) {
#pragma unroll
        for (int q = 0; q < kMB; ++q) {
            const int rb = q * 16 + fk * 4, cc = cb * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (q >= q_lo && q < q_hi && rb + r < B && cc < a.ld_slab) slab[(size_t)(rb + r) * a.ld_slab + cc] = acc3[q][r];
        }
    }
    loss = wave_sum(loss);
    __syncthreads();
    if (lane == 0) red[wave] = loss;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < kNW; ++w) s += red[w];
        a.partials[blockIdx.x] = s;
        if (TS && a.ts && blockIdx.x == 0) { a.ts[12] = wall_clock64(); a.ts[13] = (unsigned long long)iter; }
    }
}

}  // namespace aae

namespace aae {

// ---------------------------------------------------------------------------------------------------------------------
// The DEFERRED launch of the split output layer (dec_fused.h MODE kDecOpt: dV3 = G^T dh2 from the stored dL/dlogits tiles,
// then dec_optim on the tile) with the same 3-term bf16 emulation of its fp32 product.  Since r3 the next step waits for
// this launch (the critical launch got 30 % shorter), and a third of a tile's time was the fp32 matrix pipe (26 blocks x
// 28 v_mfma_f32_16x16x4_f32) fed by k-strided 4-byte LDS reads of a 95 KB dh2 image.  Here:
//   dh2 (k = batch row, column c) lives in REGISTERS, split, as the B fragments of the wave's column block for the whole
//        kernel (4 k-steps x 3 terms = 48 VGPRs; no LDS image of it);
//   G   (the tile's stored dL/dlogits, fp32 [B][32]) -> three bf16 images gB[t][b][n]; the A operand G^T[n][k = b] is their
//        transpose: ds_read_b64_tr_b16;
//   dV3 tile -> os [32][kSO] (fp32) -> the optimiser exactly as in dec_fused.h (S5), V3a / m / v streamed non-temporally,
//        V3a kept in registers between its load and its update (no LDS copy).
// Waves 0..9 own a column block for both item halves, waves 10..15 one (column block, item half) each of blocks 10..12.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kXGS = 20;       // row stride (dwords) of a G image row: 32 bf16 + pad

inline size_t dec_opt_x3_lds_bytes() { return sizeof(float) * ((size_t)3 * 128 * kXGS + (size_t)kTI * kSO + 64); }

template <int NB, bool ONE = false, bool WIN = false>      // ONE: bf16 mode (first terms only, one matrix instruction per product - see dec_crit_x3_kernel)
__global__ __launch_bounds__(kNT) void dec_opt_x3_kernel(DecFusedArgs a) {
    const bool one = ONE || a.one_term != 0;
    constexpr int NT = ONE ? 1 : 3;
    constexpr int KR = 4;                       // 32-wide k-steps over the (<= 112 -> 128) batch rows
    static_assert(NB <= 13, "column blocks 10..12 are the ones split over two waves");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* gB = reinterpret_cast<unsigned*>(lds);            // [3][128][kXGS]  G tile, rows = batch rows (>= B: zero)
    float* os = reinterpret_cast<float*>(gB + 3 * 128 * kXGS);  // [32][kSO]       dV3a tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int B = a.B, ldv = a.ldv, N = a.N;
    const int ntiles = (N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;
    const OptScalars sc = *a.sc;
    const bool do_adam = a.gradV3 == nullptr;
    typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
    if (a.dbg_skip & 0x10000) return;           // (TIMING ONLY, AAE_DEC_SKIP=65536: the step without its deferred launch's traffic)
    if (a.dbg_skip & 0x20000) {                 // (TIMING ONLY, 131072: its CUs held for 135 us without any memory traffic)
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 13500ull) __builtin_amdgcn_s_sleep(32);
        return;
    }

    for (int i = tid; i < 3 * 128 * kXGS; i += kNT) gB[i] = 0u;

    // ownership: (column block, item halves)
    int cb, nb_lo, nb_hi;
    if (wave < 10) { cb = wave; nb_lo = 0; nb_hi = 2; }
    else { cb = 10 + ((wave - 10) >> 1); nb_lo = (wave - 10) & 1; nb_hi = nb_lo + 1; }
    const bool live = cb < NB;
    cb = min(cb, NB - 1);
    // dh2 -> split B fragments: lane (fr, fk) holds column 16 cb + fr, k = batch rows 32 kc + 8 fk + {0..7}
    bf16x8 dB[KR][3];
    {
        const int c = min(16 * cb + fr, a.ldh - 1);
#pragma unroll
        for (int kc = 0; kc < KR; ++kc) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = 32 * kc + 8 * fk + j;
                x[j] = (b < B && 16 * cb + fr < a.ldh) ? a.dh2[(size_t)b * a.ldh + c] : 0.f;
            }
            unsigned p[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3_pair(x[2 * q], x[2 * q + 1], p[0][q], p[1][q], p[2][q], one);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4_t v = {p[t][0], p[t][1], p[t][2], p[t][3]};
                dB[kc][t] = __builtin_bit_cast(bf16x8, v);
            }
        }
    }

    // the streams: tensor bases in buffer descriptors, tile offset scalar, slot offset in one vector register
    const unsigned lane_off = (unsigned)tid * 16u;
    const unsigned tile_bytes = (unsigned)(kTI * ldv) * 4u;
    X3WindowT<WIN> win(N, ldv, tile_bytes);            // (the descriptors' window of the tensors, as in the critical launch)
    const float* gbase = a.gradV3 ? a.gradV3 : a.V3a;
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a), rM = win.desc(a.M), rV = win.desc(a.V), rG = win.desc(gbase);
    const int g_f4 = B * (kTI / 4);             // float4 per stored tile (<= 1024: B <= 128)
    const unsigned gbytes = (unsigned)min((size_t)0x7FFFFFF0u, (size_t)ntiles * g_f4 * 16);
    const __amdgpu_buffer_rsrc_t rGt = __builtin_amdgcn_make_buffer_rsrc(a.Gt, 0, gbytes, 0x00020000);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int tile, int j) {      // (beyond the tensor or the tile's span: zeros, no access; aux 2 = non-temporal)
        const unsigned vo = tid + kNT * j < tile_f4 ? lane_off + (unsigned)(kNT * 16 * j) : 0x80000000u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, win.so(tile), 2));
    };
    auto ldg = [&](int tile) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rGt, tid < g_f4 ? lane_off : 0x80000000u, (unsigned)tile * (unsigned)g_f4 * 16u, 2));
    };
    int s_rc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int fc = min(tid + kNT * j, tile_f4 - 1), row = fc / f4_per_row;
        s_rc[j] = row * 64 + (fc - row * f4_per_row);
    }
    float4 p_cur[NV], p_nxt[NV], mreg[NV], sreg[NV], g_nxt;
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    if (tile < ntiles) {
#pragma unroll
        for (int j = 0; j < NV; ++j) p_nxt[j] = ld4(rP, tile, j);
        g_nxt = ldg(tile);
    }
    __syncthreads();

    for (; tile < ntiles; tile += stride) {
        const int i0 = tile * kTI;
        if (win.moves(tile)) { rP = win.desc(a.V3a); rM = win.desc(a.M); rV = win.desc(a.V); rG = win.desc(gbase); }
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int lz = lane + oz;
        lds_barrier();                          // the previous tile's readers of gB / os are done
        // ---- S0: the stored dL/dlogits tile -> split images; rotate the pipeline, request the next stage
        if (tid < g_f4) {
            unsigned q0[3], q1[3];
            split3_pair(g_nxt.x, g_nxt.y, q0[0], q0[1], q0[2], one);
            split3_pair(g_nxt.z, g_nxt.w, q1[0], q1[1], q1[2], one);
            unsigned* d = gB + (tid >> 3) * kXGS + (tid & 7) * 2;
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<uint2*>(d + t * (128 * kXGS)) = make_uint2(q0[t], q1[t]);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) p_cur[j] = p_nxt[j];
        g_nxt = ldg(min(tile + stride, ntiles - 1));
#pragma unroll
        for (int j = 0; j < NV; ++j) p_nxt[j] = ld4(rP, min(tile + stride, ntiles - 1), j);
#pragma unroll
        for (int j = 0; j < NV; ++j) { mreg[j] = ld4(rM, tile, j); sreg[j] = ld4(rV, tile, j); }
        lds_barrier();

        // ---- GEMM2: dV3a[n][c] = sum_b G[b][n] dh2[b][c].  A = G^T: the transpose of 2 x (4 rows b x 16 columns n) of a
        // G image per fragment (ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group names row q, columns 4 p .. 4 p + 3)
        if (live) {
            const int tq = (lz >> 2) & 3, tp = lz & 3, g = lz >> 4;
            for (int nb2 = nb_lo; nb2 < nb_hi; ++nb2) {
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
                const unsigned* base = gB + (8 * g + tq) * kXGS + 8 * nb2 + 2 * tp;      // (16 nb2 + 4 p) bf16 = 8 nb2 + 2 p dwords
#pragma unroll
                for (int kc = 0; kc < KR; ++kc) {
                    bf16x8 ga[3];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned* pb = base + t * (128 * kXGS) + 32 * kc * kXGS;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb));
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 4 * kXGS));
                        typedef short s16x8_t __attribute__((ext_vector_type(8)));
                        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        ga[t] = __builtin_bit_cast(bf16x8, v);
                    }
                    c = mfma_xt<ONE>(ga, dB[kc], c);
                }
                // C map: row = item 16 nb2 + 4 fk + r, column = 16 cb + fr
#pragma unroll
                for (int r = 0; r < 4; ++r) os[(16 * nb2 + 4 * fk + r) * kSO + 16 * cb + fr] = c[r];
            }
        }
        lds_barrier();                          // os complete

        // ---- S5: optimiser on the tile (or gradient export), as dec_fused.h: every store issued on every path, a lane
        // without a cell gets an offset beyond the descriptor
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool valid = tid + kNT * j < tile_f4 && i0 + (s_rc[j] >> 6) < N;
            const float4 g = *reinterpret_cast<const float4*>(os + (s_rc[j] >> 6) * kSO + (s_rc[j] & 63) * 4);
            const unsigned so = win.so(tile);
            const unsigned vo = valid ? lane_off + (unsigned)(kNT * 16 * j) : 0x80000000u;
            float4 p = p_cur[j], mm = mreg[j], vv = sreg[j];
            adam_update(p.x, mm.x, vv.x, g.x, sc); adam_update(p.y, mm.y, vv.y, g.y, sc);
            adam_update(p.z, mm.z, vv.z, g.z, sc); adam_update(p.w, mm.w, vv.w, g.w, sc);
            const float4 out = do_adam ? p : g;
            const unsigned vo2 = (do_adam && !sc.is_sgd) ? vo : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, out), do_adam ? rP : rG, vo, so, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, mm), rM, vo2, so, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, vv), rV, vo2, so, 2);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The deferred half for ALL row blocks of a row-blocked step in one launch (dec_fused.h's dec_opt_blocks_kernel) on the
// emulated product, for any vocabulary: dV3 = G^T dh2 over the WHOLE batch from the tiles the critical launch stored,
// then dec_optim.  dh2 cannot stay resident (512 rows x 208 columns in three bf16 terms: 640 KB against a 512 KB register
// file), so a workgroup takes a GROUP of up to kXBT consecutive tiles (256 items) per pass, keeps their dV3 in registers
// (64 VGPRs of a column-block owner) while it walks the batch in chunks of 64 rows, and loads a chunk's dh2 fragments (24
// VGPRs) once per (pass, chunk): the re-read costs 426 KB / kXBT per tile from L2 beside the tile's own 66 KB of
// dL/dlogits and 154 KB of optimiser traffic.  The row blocks only shape the ADDRESS of a row's stored dL/dlogits (tile-major
// per block, dec_fused.h); the k walk is over global rows, so a batch of 512 rows is 16 full k-steps, no padding per block.
// Wave w < NB owns column block w for both item halves.  One LDS-only barrier per (chunk, tile) step: the step's G image
// was split into the other buffer during the step before, its fp32 values requested two steps ahead.
// a.tpp = number of tile groups G (a multiple of the grid): group g = tiles [g nt / G, (g + 1) nt / G), at most kXBT.
// Summation order: rows in order (32-row k-steps), a function of the batch alone.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kXBT = 6;       // tiles per group: 48 accumulator VGPRs of a column-block owner (7: 17 spilled registers at 128, and a
                               // scratch reload waits for every load in flight - the stream's prefetch)
constexpr int kXCH = 64;       // batch rows per chunk
constexpr int kXRowTab = 768;  // requester threads at most (16 - 4 waves)
constexpr int kXRing = 6;      // steps of dL/dlogits rows in flight (LDS-DMA ring slots of 8 KB)

// 16 bytes per lane global -> LDS without a register destination: lane l's bytes land at lds_dst + 16 l (lds_dst: wave-uniform
// LDS byte address, through M0, which is the compiler's: saved and restored in the same statement).  hipcc does not count
// this load: its completion is waited for by hand (s_waitcnt vmcnt(N), N = younger requests of the wave).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

inline size_t dec_opt_blocks_x3_lds_bytes() { return sizeof(float) * std::max((size_t)2 * 3 * 2 * kXCH * 8 + (size_t)kXRing * kXCH * kTI + 256 + 9 * kXRowTab, (size_t)kXBT * kTI * kSO); }

// dh2 [B][ldh] -> the multipliers' B fragments, split, in the order they are read: [chunk][column block][k-step][term][lane]
// x 16 bytes (lane (fr, fk) = column 16 cb + fr, rows 64 ch + 32 kc + 8 fk + {0..7}; beyond the batch / the row: zeros).
// One launch per step in front of dec_opt_blocks_x3_kernel, on its stream: a multiplier then fetches a chunk's fragments
// with six coalesced 16-byte loads instead of sixteen strided 4-byte ones plus the split, once per (group, chunk).
__global__ __launch_bounds__(128) void dh2_frag_kernel(const float* __restrict__ dh2, int ldh, int B, u32x4_t* __restrict__ out, int one_term) {
    const bool one = one_term != 0;
    const int ch = blockIdx.x, cb = blockIdx.y, nb = gridDim.y;
    const int kc = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    const int c = 16 * cb + fr;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = kXCH * ch + 32 * kc + 8 * fk + j;
        x[j] = (b < B && c < ldh) ? dh2[(size_t)b * ldh + c] : 0.f;
    }
    unsigned p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pair(x[2 * q], x[2 * q + 1], p[0][q], p[1][q], p[2][q], one);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const u32x4_t v = {p[t][0], p[t][1], p[t][2], p[t][3]};
        out[((((size_t)ch * nb + cb) * (kXCH / 32) + kc) * 3 + t) * 64 + lane] = v;
    }
}

template <int NB, bool TS = false, bool WIN = false>   // TS: debug timeline (AAE_DEC_TS=obk) of workgroup 0, steps 8 .. 15 of its first group, waves 0 and 12
__global__ __launch_bounds__(kNT) void dec_opt_blocks_x3_kernel(DecFusedArgs a) {
    const bool one = a.one_term != 0;
    constexpr int KR = kXCH / 32;
    // A term image of a step: [item half][row slot][8 dwords = 16 items], no padding.  ds_read_b64_tr_b16 is serviced in two
    // groups of 32 lanes (MI355X_MICROARCH.md, LDS) which name the rows {8g + q, 8(g+1) + q : q < 4} x 4 column chunks of 8
    // bytes: with rows 0-3 and 8-11 in eight CONSECUTIVE 32-byte slots the group covers the 64 banks exactly once.  Hence
    // slot(row) = row with bits 2 and 3 swapped.  (The [row][20 dwords] images of dec_opt_x3_kernel put rows 3 and 11 of a
    // group on banks that rows 0 and 8 occupy: 2-way conflicts on a quarter of the lanes, and this kernel's product phase is
    // bound by exactly these reads - 312 per step.)
    constexpr int IMG = 2 * kXCH * 8;           // dwords per term image
    auto slot_of = [](int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); };
    static_assert(NB <= kNW, "one column block per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* gB = reinterpret_cast<unsigned*>(lds);            // [2][3][64][kXGS]  a chunk's rows of one G tile (rows >= B: zero)
    float* ring = lds + 2 * 3 * IMG;                            // [kXRing][64 rows][32] fp32 dL/dlogits rows as they arrive (LDS-DMA), + a dump KB
    unsigned* rowtab = reinterpret_cast<unsigned*>(ring + kXRing * kXCH * kTI + 256);     // [3][kXRowTab]: the requesters' row addresses of a chunk
    float* os = lds;                                            // [kXBT][32][kSO]   the group's dV3a tiles (over both, once the products are done)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldv = a.ldv, N = a.N, B = a.B;
    const int ntiles = (N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;
    const OptScalars sc = *a.sc;
    typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
    const int G = max(a.tpp, 1);
    const int nch = (B + kXCH - 1) / kXCH;
    const int cb = min(wave, NB - 1);

    const unsigned lane_off = (unsigned)tid * 16u;
    const unsigned tile_bytes = (unsigned)(kTI * ldv) * 4u;
    X3WindowT<WIN> win(N, ldv, tile_bytes);            // (the descriptors' window of the tensors: it starts at a group's first tile)
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a), rM = win.desc(a.M), rV = win.desc(a.V);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int tile, int j) {      // (beyond the tensor: zeros; aux 2 = non-temporal)
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off + (unsigned)(kNT * 16 * j), win.so(tile), 2));
    };
    // Roles: waves 0 .. NB multiply (wave w = column block w, both item halves - the last block's halves on two waves,
    // below) and go from a step's barrier straight into their products.  The stored dL/dlogits rows of a step arrive by LDS-DMA (global_load_lds_dwordx4: no register
    // destination, so kXRing steps are in flight with no prefetch register and no rotation - rotating prefetch registers
    // made every step wait for the newest request, ~2 us of HBM latency under the optimiser traffic of the other workgroups)
    // into a ring of fp32 slots, requested by the 15 - NB >= 2 spare waves, which do nothing else and know when a slot has
    // landed (their vmcnt); the workgroup's barrier carries that to everybody.  Behind its products EVERY thread then splits
    // one float2 of the next step's slot into the other image (1024 threads x 8 bytes = 64 rows x 32 items): measured
    // alternatives - the split by the spare waves alone (3 float4 per thread: 1.0 us per step against 0.5 us of products),
    // by half the workgroup in front of its products (0.6-0.8 us) - left the multipliers waiting.
    // The LAST column block's two item halves go to two waves (NB - 1 and NB): with one owner per column block a SIMD that
    // holds four of 13 owners (waves w, w + 4, w + 8, w + 12 share a SIMD) has 8 of the 26 (column block, half) units and
    // sets the step's length; this way the SIMDs hold 7 / 7 / 6 / 6.  NM multipliers, the rest request.
    constexpr int NM = NB + 1;
    constexpr int NL = (kNW - NM) * 64, NS = (kXCH * 8 + NL - 1) / NL;
    static_assert(NM < kNW, "no spare wave left");
    static_assert(kXCH * 8 * 2 == kNT, "one float4 of a slot per thread of the lower half");
    const bool mult = wave < NM;
    const int nb_lo = wave == NB ? 1 : 0, nb_hi = wave == NB - 1 ? 1 : 2;       // this multiplier's item halves
    const int lt = tid - NM * 64;               // requesting thread
    // stored dL/dlogits of global row b, tile t: block r = b / Bb holds [tile][Br][32] (dec_fused.h)
    const float inv_bb = 1.0f / (float)a.Bb;
    const size_t blk_floats = (size_t)ntiles * a.Bb * kTI;
    auto g_row = [&](int b, int c4, unsigned& tile_stride) {           // address of (row b, tile 0, float4 c4) + its stride per tile
        const int bc = min(b, B - 1);
        const int r = (int)(((float)bc + 0.5f) * inv_bb);          // bc / Bb (exact: Bb <= 128, bc < 2^16)
        const int Br = min(a.Bb, B - r * a.Bb);
        tile_stride = (unsigned)Br * kTI;
        return a.Gt + (size_t)r * blk_floats + (size_t)(bc - r * a.Bb) * kTI + (size_t)c4 * 4;
    };
    const unsigned ring_lds = (unsigned)reinterpret_cast<size_t>(ring);
    const int lw = wave - NM;                   // requesting wave: its request i of a step covers the float4s 64 (lw + (16 - NM) i) ..

    for (int grp = blockIdx.x; grp < G; grp += gridDim.x) {
        const int t0 = (int)(((long long)grp * ntiles) / G), nt = (int)(((long long)(grp + 1) * ntiles) / G) - t0;
        if (nt <= 0) continue;
        if (win.moves(t0)) { rP = win.desc(a.V3a); rM = win.desc(a.M); rV = win.desc(a.V); }
        const int Q = nch * nt;                         // steps q = ch * nt + j
        const bool stamp_wg = TS && a.ts && blockIdx.x == 0 && grp == (int)blockIdx.x && lane == 0;

        // ---- the optimiser on the group's tiles (all threads, after the multipliers have put every dV3a tile of the group
        // into os: no accumulator is alive beside it, so two tiles' parameters and moments are in flight behind the one
        // being updated, and no barrier sits between the tiles)
        auto optimiser = [&]() {
            const bool on = !(a.dbg_skip & 0x2000);
            int s_rc[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int fc = min(tid + kNT * u, tile_f4 - 1), row = fc / f4_per_row;
                s_rc[u] = row * kSO + (fc - row * f4_per_row) * 4;
            }
            constexpr int D = 2;                        // register sets (the tile in hand + the next one in flight; 3: spills)
            float4 pr[D][NV], mr[D][NV], vr[D][NV];
#pragma unroll
            for (int d = 0; d < D - 1; ++d)
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int t = min(t0 + d, t0 + nt - 1);
                    pr[d][u] = ld4(rP, t, u); mr[d][u] = ld4(rM, t, u); vr[d][u] = ld4(rV, t, u);
                }
#pragma unroll
            for (int j = 0; j < kXBT; ++j) {
                if (j >= nt) break;
                const int tile = t0 + j, i0 = tile * kTI;
                const int cur = j % D, nxt = (j + D - 1) % D;
                if (on) {
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const int t = min(tile + D - 1, t0 + nt - 1);
                        pr[nxt][u] = ld4(rP, t, u); mr[nxt][u] = ld4(rM, t, u); vr[nxt][u] = ld4(rV, t, u);
                    }
                }
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int row = (tid + kNT * u) / f4_per_row;       // (only for the bound check below)
                    const bool valid = tid + kNT * u < tile_f4 && i0 + row < N && on;
                    const float4 g = *reinterpret_cast<const float4*>(os + j * (kTI * kSO) + s_rc[u]);
                    const unsigned so = win.so(tile);
                    const unsigned vo = valid ? lane_off + (unsigned)(kNT * 16 * u) : 0x80000000u;
                    float4 p = pr[cur][u], mm = mr[cur][u], vv = vr[cur][u];
                    adam_update(p.x, mm.x, vv.x, g.x, sc); adam_update(p.y, mm.y, vv.y, g.y, sc);
                    adam_update(p.z, mm.z, vv.z, g.z, sc); adam_update(p.w, mm.w, vv.w, g.w, sc);
                    const unsigned vo2 = !sc.is_sgd ? vo : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, p), rP, vo, so, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, mm), rM, vo2, so, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, vv), rV, vo2, so, 2);
                }
            }
        };

        lds_barrier();                                  // (the previous group's readers of os are done)
        // step q = (chunk q / nt, tile index q % nt); beyond the last step: the last step again (the count of requests in
        // flight is what the waits below rely on)
        // step q = (chunk q / nt, tile index q % nt); beyond the last step: the last step again (the waits count requests)
        // A requester keeps the addresses of its NS rows of the chunk being requested (tile 0) and their strides per tile in
        // LDS words of its own, rewritten when the request stream enters a chunk: a request is then one 64-bit multiply-add
        // (recomputed per request - a row's block, three 32-bit and two 64-bit multiplies - three requests took 0.8 us per
        // step, more than the step's products; as registers they would be alive across the multipliers' accumulators).
        auto request = [&](int q) {
            const int qc = min(q, Q - 1), ch = qc / nt, jj = qc - ch * nt;
            const unsigned slot = (unsigned)(q % kXRing) * (kXCH * kTI * 4u);
            if (jj == 0 && q < Q) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int f = min(lt + NL * i, kXCH * 8 - 1);
                    unsigned ts;
                    const size_t p = reinterpret_cast<size_t>(g_row(kXCH * ch + (f >> 3), f & 7, ts));
                    rowtab[(3 * i + 0) * kXRowTab + lt] = (unsigned)p;
                    rowtab[(3 * i + 1) * kXRowTab + lt] = (unsigned)(p >> 32);
                    rowtab[(3 * i + 2) * kXRowTab + lt] = ts;
                }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int fb = 64 * (lw + (kNW - NM) * i);      // (uniform) beyond the chunk: a request into the dump KB, so that
                const unsigned dst = fb < kXCH * 8 ? slot + (unsigned)fb * 16u : (unsigned)kXRing * (kXCH * kTI * 4u);   // every wave has NS per step
                const size_t p = (size_t)rowtab[(3 * i + 0) * kXRowTab + lt] | ((size_t)rowtab[(3 * i + 1) * kXRowTab + lt] << 32);
                const unsigned ts = rowtab[(3 * i + 2) * kXRowTab + lt];
                if (!(a.dbg_skip & 0x8000)) glds16(reinterpret_cast<const float*>(p) + (size_t)(t0 + jj) * ts, __builtin_amdgcn_readfirstlane(ring_lds + dst));
            }
        };
        // ring slot of step q (chunk ch) -> image q & 1 (rows beyond the batch: zeros), in two halves: the read is issued
        // before the requester's own LDS reads and requests, so that a spare wave queues behind the multipliers' burst of
        // fragment reads once per step, not twice
        // ring slot of step q (chunk ch) -> image q & 1 (rows beyond the batch: zeros): by the LOWER half of the workgroup,
        // one float4 per thread, behind its products.  On a SIMD the matrix pipe serves its four waves roughly in order
        // (arrival at the step's closing barrier, us after wave 0's products: waves 0-3 0.1-0.2, 4-7 0.2-0.4, 8-11 0.4,
        // wave 12 0.55 - SIMD 0 holds four multipliers, 96 MFMAs of 16 cycles per step): waves 0 .. 7 have the time, and
        // nothing then stands between the last multiplier's last product and the barrier.
        auto slot_read = [&](int q) { return *reinterpret_cast<const float4*>(ring + (q % kXRing) * (kXCH * kTI) + 4 * (tid & (kXCH * 8 - 1))); };
        auto split_write = [&](int q, int ch, float4 g) {
            const int f = tid & (kXCH * 8 - 1);
            if ((f >> 3) >= B - kXCH * ch) g = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned q0[3], q1[3];
            split3_pair(g.x, g.y, q0[0], q0[1], q0[2], one);
            split3_pair(g.z, g.w, q1[0], q1[1], q1[2], one);
            unsigned* d = gB + (q & 1) * 3 * IMG + ((f >> 2) & 1) * (kXCH * 8) + slot_of(f >> 3) * 8 + (f & 3) * 2;
#pragma unroll
            for (int t = 0; t < 3; ++t) *reinterpret_cast<uint2*>(d + t * IMG) = make_uint2(q0[t], q1[t]);
        };
        const bool splitter = wave < kNW / 2;
        if (!mult) {
#pragma unroll
            for (int d = 0; d < kXRing; ++d) request(d);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS * (kXRing - 2)) : "memory");      // steps 0 and 1 have landed
        }
        lds_barrier();
        if (splitter) split_write(0, 0, slot_read(0));

        f32x4 acc[kXBT][2];
#pragma unroll
        for (int j = 0; j < kXBT; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        int q = 0;
        bf16x8 dB[KR][3];
        for (int ch = 0; ch < nch; ++ch) {
            // the chunk's dh2 fragments (dh2_frag_kernel's image)
            if (mult && (ch == 0 || !(a.dbg_skip & 0x4000))) {
                const u32x4_t* F = reinterpret_cast<const u32x4_t*>(a.acc) + (((size_t)ch * NB + cb) * KR * 3) * 64 + lane;
#pragma unroll
                for (int kc = 0; kc < KR; ++kc)
#pragma unroll
                    for (int t = 0; t < 3; ++t) dB[kc][t] = __builtin_bit_cast(bf16x8, F[(kc * 3 + t) * 64]);
                // the fragments are waited for HERE: left to the compiler, the wait lands in front of their first use inside
                // every step
#pragma unroll
                for (int kc = 0; kc < KR; ++kc)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        u32x4_t v = __builtin_bit_cast(u32x4_t, dB[kc][t]);
                        asm volatile("" : "+v"(v));
                        dB[kc][t] = __builtin_bit_cast(bf16x8, v);
                    }
            }
#pragma unroll
            for (int j = 0; j < kXBT; ++j) {
                if (j >= nt) break;                     // (uniform)
                int oz;
                asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
                const int lz = lane + oz;
                lds_barrier();                          // image q is complete, slot q + 1 has landed, slot q is free
                const bool stamp = stamp_wg && (wave == 0 || wave == NM) && q >= 8 && q < 16;
                unsigned long long* tsq = a.ts + (wave ? 64 : 0) + (q - 8) * 4;
                if (stamp) tsq[0] = wall_clock64();
                if (mult && !(a.dbg_skip & 0x1000)) {
                    const int tq = (lz >> 2) & 3, tp = lz & 3, g = lz >> 4;
                    const unsigned* img = gB + (q & 1) * 3 * IMG;
#pragma unroll
                    for (int nb2 = 0; nb2 < 2; ++nb2) {
                        if (nb2 < nb_lo || nb2 >= nb_hi) continue;      // (uniform)
                        f32x4 c = acc[j][nb2];
                        // rows 32 kc + 8 g + 4 hi + tq -> slot tq | (g & 1) << 2 | hi << 3 | (g >> 1) << 4 | kc << 5
                        const unsigned* base = img + nb2 * (kXCH * 8) + (tq | ((g & 1) << 2) | ((g >> 1) << 4)) * 8 + 2 * tp;
#pragma unroll
                        for (int kc = 0; kc < KR; ++kc) {
                            bf16x8 ga[3];
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const unsigned* pb = base + t * IMG + 32 * kc * 8;
                                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb));
                                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 8 * 8));
                                typedef short s16x8_t __attribute__((ext_vector_type(8)));
                                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                ga[t] = __builtin_bit_cast(bf16x8, v);
                            }
                            c = mfma_x3(ga, dB[kc], c);
                        }
                        acc[j][nb2] = c;
                    }
                }
                if (stamp) tsq[1] = wall_clock64();
                if (splitter && q + 1 < Q) split_write(q + 1, j + 1 < nt ? ch : ch + 1, slot_read(q + 1));
                if (!mult) request(q + kXRing);         // slot q is requested again
                if (stamp) tsq[2] = wall_clock64();
                if (!mult) {                            // step q + 2 will have landed at the next barrier
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS * (kXRing - 2)) : "memory");
                    if (stamp) tsq[3] = wall_clock64();
                }
                if (TS && stamp_wg && (q == 9 || q == 12)) { a.ts[(q == 9 ? 32 : 96) + wave] = wall_clock64(); if (wave == 0) a.ts[(q == 9 ? 48 : 112)] = tsq[1]; }
                ++q;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the ring's last requests have landed: os takes its place)
        lds_barrier();                                  // every multiplier is done with the images
        if (mult) {
#pragma unroll
            for (int j = 0; j < kXBT; ++j)
#pragma unroll
                for (int nb2 = 0; nb2 < 2; ++nb2)
                    if (nb2 >= nb_lo && nb2 < nb_hi) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) os[j * (kTI * kSO) + (16 * nb2 + 4 * fk + rr) * kSO + 16 * cb + fr] = acc[j][nb2][rr];
                    }
        }
        lds_barrier();
        optimiser();
    }
}

}  // namespace aae
