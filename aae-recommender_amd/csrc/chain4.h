// Row-blocked layer chains, 4 rows per workgroup (the r2 form of chain.h's programs; same ChainProgram, same ops).
//
// chain.h gives a 1024-thread workgroup 16 rows of the batch: a batch of 100 is 7 workgroups on a 256-CU chip, and on
// the one CU that owns them a 201 -> 200 layer costs its weight stream (~1.5 us through the CU's 64 B/clk vector-memory
// path) PLUS 3 us of matrix pipe (13 blocks x 52 v_mfma_f32_16x16x4_f32 on 4 SIMDs) plus epilogue - 6-7 us, 46 layer
// ops per step.  Here a workgroup takes 4 rows (25 workgroups at batch 100, 50 for the discriminator's stacked batch)
// and the layer runs on v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4x4 with K = 1 per instruction - 4 rows x 64 columns
// per wave and k-step at the same 64 flop/clk/SIMD, i.e. a quarter of the matrix time (0.75 us) and a quarter of the
// epilogue per workgroup; every workgroup still streams the whole weight matrix (L2 serves 25 readers of 160 KB).
//
//   operands:  A = x[row = lane % 4][k]  from the LDS slot (one 16-byte read = 4 k-steps; 4 distinct addresses per
//                  wave-instruction, each broadcast to 16 lanes);
//              B = Wkn[k][64 cg + lane]: the weight matrix in its "k-major" form (n contiguous) - the transposed
//                  copy PT of a forward layer (now including the bias row), the matrix itself for a dX layer - one
//                  coalesced 256-byte row segment per wave-instruction;
//              D = 4 registers: rows 0..3 of column 64 cg + lane.
//   waves:     16 = (column group of 64) x (k split): a 200-wide layer has 4 column groups x 4 k-quarters, a 50-wide
//              one 1 x 16.  Partial sums of the k-split meet in LDS ([k-slice][row][column]), the epilogue (dropout +
//              activation, activation' x dropout, ...) runs one output cell per thread.
//   bf16 mode: both operands are rounded to bf16 values on the way in (v_cvt_pk_bf16_f32, ties to even) and multiplied
//              as fp32: the products of bf16 values are exact in fp32, so this IS bf16-input / fp32-accumulate
//              arithmetic (the layer is bound by its weight stream, not by the matrix pipe).
#pragma once
#include "chain.h"

namespace aae {

constexpr int kR4 = 4;          // rows per workgroup
#ifndef C4_WAVES
#define C4_WAVES 16
#endif
constexpr int kC4W = C4_WAVES;              // waves per workgroup: 16, or 8 (every thread then owns two cells of the element-wise work)
constexpr int kC4WS = kC4W == 16 ? 4 : 3;   // log2
constexpr int kC4T = 64 * kC4W; // threads
constexpr int kC4E = 1024 / kC4T;           // cells per thread in element-wise work (4 rows x 256 columns)
constexpr int kC4ER = kC4T / 256;           // rows a pass of the workgroup's threads covers
constexpr int kC4Part = 4096;   // floats of the k-split partial-sum scratch: [16 / cgp][4][64 cgp]

__device__ __forceinline__ float chain4_rb(float x) {       // nearest bf16 value, as fp32
    return __builtin_bit_cast(float, gemm_pack_bf16(x, 0.f) << 16);
}

// the scalars of a linear op, loaded from the op descriptor in one batch by the caller
struct Lin4 { int N, K, ldkn, ns4; const float* W4; const float* Wkn; };

// One layer's matrix work for this wave: MK = k-steps per wave (compile-time bound of the register arrays).
// (-DC4_NO_LOADS / C4_NO_MFMA / C4_NO_A / C4_NO_PART / C4_NO_EPI / C4_NO_ONECOL: timing-only ablations, tools/debug/ubench/README.md)
template <int MK, bool BF, bool TS = false>
__device__ __forceinline__ void chain4_linear(const Lin4& op, const float* src, float* part, int wave, int lane,
                                              int cgs, int kper, unsigned long long* wts_ = nullptr) {
    unsigned long long* wts = TS ? wts_ : nullptr;
    const int N = op.N, K = op.K, ld = op.ldkn;
    const int cgp = 1 << cgs;
    const int cg = wave & (cgp - 1), ks = wave >> cgs;
    const int k0 = ks * kper;
    const int n = min(64 * cg + lane, N - 1);
    // ---- the common case, with as few SCALAR instructions as possible.  The CU has ONE scalar unit for its 16 waves: at
    // ~300 scalar instructions per wave and op (r3 counters: SQ_INSTS_SALU) it alone was ~2 of an op's 3.8 us, every wave
    // computing the same clamps, bounds and offsets.  When this wave's MK k-steps are all its own (kper == MK), lie inside
    // the slot row, and the weight chunks it names exist (the matrix + its kW4Pad zero rows), nothing needs clamping or
    // masking: the A operand beyond K is zero by construction (every writer of a slot zeroes columns >= N; the constant-1
    // column is the last one a layer reads), the chunk offsets advance by one scalar add, the LDS reads take immediate
    // offsets.
    if (op.W4 && kper == MK && k0 + MK <= kCL && (k0 >> 2) + MK / 4 - 1 <= ((K - 1) >> 2) + kW4Pad) {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.W4), 0, 0x7FFFFFF0, 0x00020000);
        const unsigned vo = (unsigned)n * 16u;
        const unsigned st = (unsigned)op.ns4 * 16u;
        unsigned so = (unsigned)(k0 >> 2) * st;
        float4 wq[MK / 4];
#pragma unroll
#ifdef C4_NO_LOADS
        for (int j = 0; j < MK / 4; ++j) { wq[j] = make_float4(1.f + so, 2.f, 3.f, 4.f + vo); so += st; }
#else
        for (int j = 0; j < MK / 4; ++j) { wq[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, so, 0)); so += st; }
#endif
        if (wts && lane == 0) wts[wave] = wall_clock64();
        // Every weight load of the layer is requested before its first product: left to itself hipcc sinks the loads between
        // the products (r4, ISA: two loads, s_waitcnt vmcnt(0), four products, one load, vmcnt(1), ... - 2-3 KB in flight per
        // wave instead of 13).  Same-box A/B of the two builds: 0.2521 -> 0.2502 ms/step, chain launches 28.3 -> 27.4 us.
        __builtin_amdgcn_sched_barrier(0);
        const float4* a4 = reinterpret_cast<const float4*>(src + (lane & 3) * kCL + k0);
        f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int j = 0; j < MK / 4; ++j) {
            float4 x = a4[j];
            float w0 = wq[j].x, w1 = wq[j].y, w2 = wq[j].z, w3 = wq[j].w;
            if (BF) {
                x.x = chain4_rb(x.x); x.y = chain4_rb(x.y); x.z = chain4_rb(x.z); x.w = chain4_rb(x.w);
                w0 = chain4_rb(w0); w1 = chain4_rb(w1); w2 = chain4_rb(w2); w3 = chain4_rb(w3);
            }
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w1, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w2, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, w3, c1, 0, 0, 0);
        }
        c0 += c1;
        float* pq = part + (size_t)ks * (4 * 64 * cgp) + 64 * cg + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) pq[r * 64 * cgp] = c0[r];
        if (wts && lane == 0) wts[32 + wave] = wall_clock64();
        return;
    }
    // ---- the general case: all weight loads of the layer in flight before the first MFMA; row index clamped
    float w[MK];
    if (op.W4) {
        // k4-interleaved copy: four consecutive k of this lane's column in one 16-byte load, a wave-instruction = 1 KB
        // (buffer addressing: the matrix in a scalar descriptor, the k-chunk's byte offset in a scalar register, the lane's
        // column in ONE vector register for all loads of the layer - per-load 64-bit vector addresses spilled the kernel)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.W4), 0, 0x7FFFFFF0, 0x00020000);
        const unsigned vo = (unsigned)n * 16u;
        const int kc0 = k0 >> 2, kcmax = (K - 1) >> 2;
#pragma unroll
        for (int j = 0; j < MK / 4; ++j) {
            const unsigned so = (unsigned)(min(kc0 + j, kcmax) * op.ns4) * 16u;
#ifdef C4_NO_LOADS
            const float4 t = make_float4(1.f + so, 2.f, 3.f, 4.f + vo);
#else
            const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, so, 0));
#endif
            w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w;
        }
    } else {
        const float* wp = op.Wkn + n;
#pragma unroll
        for (int i = 0; i < MK; ++i) w[i] = wp[(size_t)min(k0 + i, K - 1) * ld];
    }
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* a = src + (lane & 3) * kCL;
    if (wts && lane == 0) wts[wave] = wall_clock64();                  // (debug, AAE_CHAIN_TS) every load of the wave is issued
    // The A operand: one 16-byte LDS read per chunk of 4 k, from an always valid (clamped, wave-uniform) address, requested
    // one chunk AHEAD of its products.  This wave's k range is [k0, klim): a chunk inside it needs no masking, the one klim
    // falls into is masked by selects on scalar conditions, chunks beyond it are skipped - three scalar instructions per
    // chunk in the common case.  (r1-r3 masked every element of every chunk against K, the slot row and the k-slice: ~20
    // scalar + 4 vector instructions per chunk beside its 4 MFMAs, and read the chunk under a branch right in front of its
    // first product - the matrix phase of a 201 -> 200 layer took 3.6 us where its loads, products and partial sums take
    // 1.7 us in isolation: tools/debug/ubench/chain_stream.hip, chain_real.hip.)
    const int klim = min(min(K, k0 + kper), kCL);
#ifdef C4_NO_A
    auto ldx = [&](int j) { return make_float4(1.f + j, 2.f, 3.f, 4.f + lane); };
#else
    auto ldx = [&](int j) { return *reinterpret_cast<const float4*>(a + min(k0 + 4 * j, kCL - 4)); };
#endif
    float4 xn = ldx(0);
#pragma unroll
    for (int j = 0; j < MK / 4; ++j) {
        if (wts && j == 1 && lane == 0) { asm volatile("s_nop 0" :: "v"(acc0[0])); wts[16 + wave] = wall_clock64(); }       // ... its first chunk has arrived and is multiplied
        const int kk = k0 + 4 * j;
        float4 x = xn;
        if (j + 1 < MK / 4) xn = ldx(j + 1);
        if (kk >= klim) continue;                                       // (uniform)
        if (kk + 4 > klim) {                                            // (uniform) the chunk klim falls into
            x.y = kk + 1 < klim ? x.y : 0.f;
            x.z = kk + 2 < klim ? x.z : 0.f;
            x.w = 0.f;
        }
        float w0 = w[4 * j], w1 = w[4 * j + 1], w2 = w[4 * j + 2], w3 = w[4 * j + 3];
        if (BF) {
            x.x = chain4_rb(x.x); x.y = chain4_rb(x.y); x.z = chain4_rb(x.z); x.w = chain4_rb(x.w);
            w0 = chain4_rb(w0); w1 = chain4_rb(w1); w2 = chain4_rb(w2); w3 = chain4_rb(w3);
        }
#ifdef C4_NO_MFMA
        acc0[0] += x.x * w0; acc1[0] += x.y * w1; acc0[1] += x.z * w2; acc1[1] += x.w * w3;
#else
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, w3, acc1, 0, 0, 0);
#endif
    }
    acc0 += acc1;
    // partial sums: part[ks][row][64 cg + lane], row stride 64 cgp
    float* pp = part + (size_t)ks * (4 * 64 * cgp) + 64 * cg + lane;
#pragma unroll
#ifndef C4_NO_PART
    for (int r = 0; r < 4; ++r) pp[r * 64 * cgp] = acc0[r];
#else
    if (acc0[0] == 12345.f) pp[0] = acc0[1];
#endif
    if (wts && lane == 0) wts[32 + wave] = wall_clock64();             // ... its partial sums are in LDS
}

// (r6) ONE BARRIER PER OP: the column-owner form of a linear op - bf16 mode, batches of one fused launch (C2).  Above, a wave =
// (64 columns, a slice of K): the k-slices' partial sums meet in LDS behind a workgroup barrier, the epilogue reads them back, a
// second barrier closes the op.  Here the 16 blocks of a 4x4x1 instruction are 4 k-residues x 4 column quartets instead of 16
// column quartets: block b = (kq = b / 4, cq = b % 4), so a wave owns 16 COLUMNS for ALL of K - lane (kq, c) multiplies
// k = 16 J + 4 kq + {0..3} of column 16 wave + c per step J (the same k4-interleaved copy: its float4 of chunk 4 J + kq; the A
// operand x[lane % 4][16 J + 4 kq ..] one 16-byte LDS read as before), the four k-residues' sums of a column - lanes c, c + 16,
// c + 32, c + 48 - are added across the lanes (v_permlane16_swap / v_permlane32_swap: no LDS round trip), and lane (kq, c) runs
// the epilogue of cell (row kq, column 16 wave + c) on its own registers.  No partial sums in LDS, no barrier between products
// and epilogue; 13 of 16 waves carry a 200-wide layer.  Chain launches at C2 21.5 -> 20.5 us, C2 bf16 0.161 -> 0.155 ms/step.
// ANOTHER SUMMATION ORDER than the k-slice form's (fp32 accumulation of the same exact products of bf16 values), which is why
// it is taken where it is: in fp32 mode it would move every bit of every run for nothing at C3 and 0.7 % at C4 (measured,
// HISTORY.md), and the one statistical bound of the bf16 tests that was set with the k-slice order - the fraction of dec.lin1's
// elements more than 1e-4 from the oracle after three 1 000-row steps with a 300-wide condition, 1.2 % there, 2 % allowed -
// reads 2.8 % with this one (near-zero gradients change sign under any other order and Adam moves them by lr): wide batches
// keep the k-slice order.  NJ = 16-deep k-steps (4 | 7 | 13: K <= 64 | 112 | 208); chunks beyond the copy (+ its zero rows)
// read as zero through the descriptor's range check, the A operand beyond K is zero by construction.
template <int NJ>
__device__ __forceinline__ f32x4 chain4_cols(const float* W4, int ns4, int N, int K, const float* src, int wave, int lane) {
    const int kq = lane >> 4, col = min(16 * wave + (lane & 15), N - 1);
    const unsigned st = (unsigned)ns4 * 16u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W4), 0, (unsigned)(((K + 3) >> 2) + kW4Pad) * st, 0x00020000);
    const unsigned vo = (unsigned)col * 16u + (unsigned)kq * st;
    unsigned so = 0;
    float4 wq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { wq[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, so, 0)); so += 4u * st; }
    __builtin_amdgcn_sched_barrier(0);          // (every weight load of the layer requested before its first product: chain4_linear)
    const float4* a4 = reinterpret_cast<const float4*>(src + (lane & 3) * kCL + 4 * kq);
    f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 x = a4[4 * j];             // (both operands rounded to bf16 values, multiplied as fp32: the file's header)
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(chain4_rb(x.x), chain4_rb(wq[j].x), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(chain4_rb(x.y), chain4_rb(wq[j].y), c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(chain4_rb(x.z), chain4_rb(wq[j].z), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(chain4_rb(x.w), chain4_rb(wq[j].w), c1, 0, 0, 0);
    }
    c0 += c1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (a swap of a register with its own copy: one result holds the even 16-lane rows / the lower half twice, the other the odd
        //  rows / the upper half - their sum is x + x[lane ^ 16] resp. x + x[lane ^ 32] in every lane, the same bits in all four)
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[r]), __float_as_uint(c0[r]), false, false);
        const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
        c0[r] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    return c0;
}

// TS: the debug build with in-kernel stamps (AAE_CHAIN_TS).  The production build carries none: a stamp site between a layer's
// weight loads and its products is a (flat) store the compiler orders with s_waitcnt vmcnt(0) - every product then waited for
// ALL of the wave's loads instead of its own chunk's - and each site costs every wave scalar instructions.
// COLS: the instantiation that carries the column-owner form of a linear op (bf16 mode, batches of one fused launch).  The
// k-slice form sits at its 128-register cap: compiled beside the other form it keeps one of its thirteen weight chunks in scratch
// memory (and waits for all of them before its first product: C4's shape in bf16 0.312 -> 0.321 ms/step) - so wide batches are
// launched on the instantiation without it.
template <bool BF, bool TS = false, bool COLS = false>
__global__ __launch_bounds__(kC4T) void chain4_kernel(ChainProgram P) {
    extern __shared__ __attribute__((aligned(16))) float slots[];     // [kCSlots][4][kCL], then the partial-sum scratch
    if (P.bk.enabled && blockIdx.x == gridDim.x - 1) {                // (uniform) the piggy-backed bucket builder
        tile_bucket_body<kBucketMaxDocs, kC4T>(P.bk.bv, P.bk.ntiles, P.bk.tstart, P.bk.eb, P.bk.en, P.bk.ev, reinterpret_cast<int*>(slots));
        return;
    }
    float* part = slots + kCSlots * kR4 * kCL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int erow0 = tid >> 8, ecol = tid & 255;      // element-wise work: cell (erow0 + kC4ER e, ecol), e < kC4E
    const int r0 = blockIdx.x * kR4;
    const int nrows = min(kR4, P.rows - r0);
    const uint64_t key = rng_key(P.seed, (uint64_t)*P.step_ctr, 0);

    // (debug stamps, AAE_CHAIN_TS: workgroup 0 only; one pinned pointer, NULL in production)
    unsigned long long* tsp = (TS && blockIdx.x == 0) ? P.ts : nullptr;
    int nops = P.nops, kslices = COLS ? P.kslices : 1;
    asm volatile("" : "+s"(tsp), "+s"(nops));
    if constexpr (COLS) asm volatile("" : "+s"(kslices));
    for (int oi = 0; oi < nops; ++oi) {
        const ChainOp& op = P.ops[oi];
        if (TS && tsp && tid == 0) tsp[oi] = wall_clock64();
        // The scalars every op needs, and a linear op's, requested together and pinned by ONE statement each: a single
        // s_waitcnt for the batch.  hipcc treats descriptor fields (kernel-argument segment) as free to re-load at their
        // use, each a dependent scalar-load round trip of ~200 clocks (tools/debug/ubench/sload_latency.hip) - ~20 of them
        // per op, serialised, were most of the 4 us an op cost WITHOUT its weight loads and products
        // (tools/debug/ubench/chain_real.hip, r3).
        int kind = op.kind, src_i = op.src, dst_i = op.dst, row_lo = op.row_lo, one_col = op.one_col, opN = op.N, opK = op.K;
        int ldo = op.ldo, ldo2 = op.ldo2, out_row0 = op.out_row0;
        float* outp = op.out; float* out2p = op.out2;
        Lin4 lin; lin.N = opN; lin.K = opK; lin.ldkn = op.ldkn; lin.ns4 = op.ns4; lin.W4 = op.W4; lin.Wkn = op.Wkn;
        int epi_k = op.epi, yslot_k = op.yslot, acc_in = op.acc_in;
        asm volatile("" : "+s"(kind), "+s"(src_i), "+s"(dst_i), "+s"(row_lo), "+s"(one_col), "+s"(opN), "+s"(opK), "+s"(ldo), "+s"(ldo2),
                          "+s"(out_row0), "+s"(outp), "+s"(out2p), "+s"(epi_k), "+s"(yslot_k), "+s"(acc_in));
        const float* ygp = op.y_glb; int yld = op.y_ld;
        asm volatile("" : "+s"(lin.ldkn), "+s"(lin.ns4), "+s"(lin.W4), "+s"(lin.Wkn), "+s"(ygp), "+s"(yld));
        lin.N = opN; lin.K = opK;
        if (r0 < row_lo) continue;                     // (workgroup-uniform: an op of the upper rows' program prefix)
        float* dst = slots + dst_i * kR4 * kCL;
        const float* src = slots + src_i * kR4 * kCL;

        // the other kinds' scalars, the same way (one batch)
        const float* qW = nullptr; float* qaux_ptr = nullptr; size_t qstride = 0;
        int qldw = 0, qdst_col0 = 0, qaux = 0, qaux_ld = 0, qyslot = 0, qgrow0 = 0, qrow_split = 0, qfake_slot = -1;
        float qscale = 0.f;
        if (kind != COP_LINEAR && kind != COP_LINEAR_DX) {
            qW = op.W; qaux_ptr = op.aux_ptr; qstride = op.stride; qldw = op.ldw; qdst_col0 = op.dst_col0; qaux = op.aux;
            qaux_ld = op.aux_ld; qyslot = op.yslot; qgrow0 = op.grow0; qrow_split = op.row_split; qfake_slot = op.fake_slot; qscale = op.scale;
            asm volatile("" : "+s"(qW), "+s"(qaux_ptr), "+s"(qstride), "+s"(qldw), "+s"(qdst_col0), "+s"(qaux), "+s"(qaux_ld), "+s"(qyslot),
                              "+s"(qgrow0), "+s"(qrow_split), "+s"(qfake_slot), "+s"(qscale));
        }
        bool one_done = false;                         // the constant-1 column was written by the op's own cell loop
        if (kind == COP_LINEAR || kind == COP_LINEAR_DX) {
            const int N = opN, K = opK;
            const int CG = (N + 63) >> 6;
            const int cgs = CG <= 1 ? 0 : CG <= 2 ? 1 : 2, cgp = 1 << cgs;  // column groups, rounded to a power of two (shifts: a
            const int KS = kC4W >> cgs;                                     // runtime division is ~150 clocks of its own, three of them per op)
            const int kper = (((K + KS - 1) >> (kC4WS - cgs)) + 3) & ~3;    // k-steps per wave, a multiple of 4
            unsigned long long* wts = (TS && tsp && oi == 2) ? tsp + 64 : nullptr;
            if (wts && tid == 0) { wts[48] = wall_clock64(); wts[49] = clock64(); }
            // ---- the column-owner form (one barrier per op): the common case of every program of the step
            // (the k-slice form below carries fp32 mode, wide batches - the host sets P.kslices -, a layer in place, a matrix without
            //  its k4-interleaved copy and the 8-wave build; AAE_CHAIN_KSLICES forces it for the tests)
            bool cols_ok = false;
            if constexpr (BF && COLS && kC4W == 16) cols_ok = !kslices && lin.W4 && K <= kCWide && N <= 16 * kC4W && src_i != dst_i;
            if (cols_ok) {                              // (uniform; bf16 mode only: the fp32 kernel carries none of this)
                const int crow = lane >> 4, ccol = 16 * wave + (lane & 15);    // this thread's cell of the 4 x 256 block
                float yv1 = 0.f;
                if (ygp) {
                    typedef const __attribute__((address_space(1))) float* gf_t;
                    yv1 = (ccol < N && crow < nrows) ? ((gf_t)ygp)[(size_t)(r0 + crow) * yld + ccol] : 0.f;
                }
                f32x4 cs = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (16 * wave < N) {
                    if (K <= 64) cs = chain4_cols<4>(lin.W4, lin.ns4, N, K, src, wave, lane);
                    else if (K <= 112) cs = chain4_cols<7>(lin.W4, lin.ns4, N, K, src, wave, lane);
                    else cs = chain4_cols<13>(lin.W4, lin.ns4, N, K, src, wave, lane);
                }
                if (wts && lane == 0) wts[32 + wave] = wall_clock64();
                one_done = true;
                if (16 * wave < kCL) {                  // (waves 0 .. 13: the slot row's 212 columns)
                    const EpiCtx ec = chain_epi_ctx(epi_k, op, P, key, slots);
                    float v = 0.f;
                    if (ccol < N && crow < nrows) {
                        v = crow == 0 ? cs[0] : crow == 1 ? cs[1] : crow == 2 ? cs[2] : cs[3];
                        if (acc_in) v += dst[crow * kCL + ccol];
                        if (ec.epi == CEPI_ACTBWD) {
                            v *= act_grad_from_y(ec.act, ygp ? yv1 : (slots + yslot_k * kR4 * kCL)[crow * kCL + ccol]);
                            if (ec.den) v *= chain_keep(ec, r0 + crow, ccol) ? ec.mk : 0.f;
                        } else {
                            v = chain_epi<false>(ec, r0 + crow, crow, ccol, v);
                        }
                    }
                    if (ccol < kCL) dst[crow * kCL + ccol] = ccol == one_col ? (crow < nrows ? 1.f : 0.f) : v;
                }
            } else {
            // y of an ACTBWD epilogue that lives in global memory: this thread's cells, requested in front of the products
            float yv[kC4E];
            if (ygp) {
                typedef const __attribute__((address_space(1))) float* gf_t;
#pragma unroll
                for (int eh = 0; eh < kC4E; ++eh) {
                    const int erow = erow0 + kC4ER * eh;
                    yv[eh] = (ecol < N && erow < nrows) ? ((gf_t)ygp)[(size_t)(r0 + erow) * yld + ecol] : 0.f;
                }
            }
            if ((wave & (cgp - 1)) < CG) {
                if (kper <= 16) chain4_linear<16, BF, TS>(lin, src, part, wave, lane, cgs, kper, wts);
                else if (kper <= 28) chain4_linear<28, BF, TS>(lin, src, part, wave, lane, cgs, kper, wts);
                else if (kper <= 52) chain4_linear<52, BF, TS>(lin, src, part, wave, lane, cgs, kper, wts);
                else if (kC4W == 8) {       // (8 waves: two k-slices of a 200-wide layer - 100 or 104 k-steps per wave)
                    if (kper == 100) chain4_linear<100, BF, TS>(lin, src, part, wave, lane, cgs, kper, wts);
                    else chain4_linear<104, BF, TS>(lin, src, part, wave, lane, cgs, kper, wts);
                }
            }
            one_done = true;
            chain_barrier();
            if (wts && tid == 0) { wts[50] = wall_clock64(); wts[51] = clock64(); }
            // (measured, r4: requesting this context with the op's other descriptor fields, in front of the matrix phase, costs
            //  46 spilled scalar registers and 0.6 % of the step)
            const EpiCtx ec = chain_epi_ctx(epi_k, op, P, key, slots);
            if (wts && tid == 0) wts[52] = wall_clock64();
#ifdef C4_NO_EPI
            if (ecol < 0)
#else
            if (ecol < kCL)
#endif
            for (int eh = 0; eh < kC4E; ++eh) {
                const int erow = erow0 + kC4ER * eh;
                float v = 0.f;
                if (ecol < N && erow < nrows) {
                    // the k-slices' partial sums: every read of the cell in flight at once (a loop over KS waited for each
                    // read before it asked for the next: 4-16 LDS round trips in a row), summed in slice order
                    const float* pp = part + erow * 64 * cgp + ecol;
                    auto sum_slices = [&](auto n) {
                        constexpr int NS = decltype(n)::value;
                        float pv[NS];
#pragma unroll
                        for (int ks = 0; ks < NS; ++ks) pv[ks] = pp[(size_t)ks * (4 * 64 * cgp)];
#pragma unroll
                        for (int ks = 0; ks < NS; ++ks) v += pv[ks];
                    };
                    if (cgs == 2) sum_slices(std::integral_constant<int, kC4W / 4>{});    // (KS = waves >> cgs)
                    else if (cgs == 1) sum_slices(std::integral_constant<int, kC4W / 2>{});
                    else sum_slices(std::integral_constant<int, kC4W>{});
                    if (acc_in) v += dst[erow * kCL + ecol];          // (uniform) the layer's earlier k-part: this thread's own cell
                    // (the epilogue's y slot holds 4-row blocks here: index it with this kernel's row stride)
                    if (ec.epi == CEPI_ACTBWD) {
                        v *= act_grad_from_y(ec.act, ygp ? yv[eh] : (slots + yslot_k * kR4 * kCL)[erow * kCL + ecol]);
                        if (ec.den) v *= chain_keep(ec, r0 + erow, ecol) ? ec.mk : 0.f;
                    } else {
                        v = chain_epi<false>(ec, r0 + erow, erow, ecol, v);
                    }
                }
                // (columns >= N read as zero for the next layer; the constant-1 column of an augmented layer input rides here
                //  instead of in a write + barrier of its own behind every op)
                dst[erow * kCL + ecol] = ecol == one_col ? (erow < nrows ? 1.f : 0.f) : v;
            }
            }
        } else if (kind == COP_LOAD) {
            if (ecol + qdst_col0 < kCL)
                for (int eh = 0; eh < kC4E; ++eh) {
                    const int erow = erow0 + kC4ER * eh;
                    const float lv = (erow < nrows && ecol < opN) ? qW[(size_t)(out_row0 + r0 + erow) * qldw + ecol] * qscale : 0.f;
                    dst[erow * kCL + qdst_col0 + ecol] = qdst_col0 + ecol == one_col ? (erow < nrows ? 1.f : 0.f) : lv;
                }
            one_done = true;
        } else if (kind == COP_SLABSUM) {
            // sum of qaux (<= 16) partial slabs: every slab load of a thread is in flight at once
            if (ecol < kCL)
            for (int eh = 0; eh < kC4E; ++eh) {
                const int erow = erow0 + kC4ER * eh;
                const int rowc = min(erow, max(nrows, 1) - 1), cc = min(ecol, opN - 1);
                float v[16];
#pragma unroll
                for (int z = 0; z < 16; ++z)
                    v[z] = qW[(size_t)min(z, qaux - 1) * qstride + (size_t)(r0 + rowc) * qldw + cc];
                float acc = 0.f;
#pragma unroll
                for (int z = 0; z < 16; ++z)
                    if (z < qaux) acc += v[z];
                if (epi_k == CEPI_ACTBWD) {
                    const EpiCtx sec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, slots);
                    const float y = qaux_ptr[(size_t)(r0 + rowc) * qaux_ld + cc];
                    const bool cell = erow < nrows && ecol < opN;
                    const float kp = (sec.den && cell) ? (chain_keep(sec, r0 + erow, ecol) ? sec.mk : 0.f) : 1.f;
                    acc *= act_grad_from_y(sec.act, y) * kp;
                }
                dst[erow * kCL + ecol] = (erow < nrows && ecol < opN) ? acc : 0.f;
            }
        } else if (kind == COP_DROPACT || kind == COP_ACTBWD) {
            const EpiCtx ec = chain_epi_ctx(kind == COP_DROPACT ? CEPI_DROPACT : CEPI_ACTBWD, op, P, key, slots);
            if (ecol < kCL)
            for (int eh = 0; eh < kC4E; ++eh) {
                const int erow = erow0 + kC4ER * eh;
                float v = 0.f;
                if (erow < nrows && ecol < opN) {
                    v = src[erow * kCL + ecol];
                    if (kind == COP_DROPACT) v = chain_epi<false>(ec, r0 + erow, erow, ecol, v);
                    else {
                        v *= act_grad_from_y(ec.act, (slots + qyslot * kR4 * kCL)[erow * kCL + ecol]);
                        if (ec.den) v *= chain_keep(ec, r0 + erow, ecol) ? ec.mk : 0.f;
                    }
                }
                dst[erow * kCL + ecol] = ecol == one_col ? (erow < nrows ? 1.f : 0.f) : v;
            }
            one_done = true;
        } else if (kind == COP_FINAL_FWD) {
            if (wave < kR4) {                           // one wave per row; softmax / sigmoid / identity, in place on dst
                float* zr = dst + wave * kCL;
                if (qaux == 1) {
                    float mx = -INFINITY;
                    for (int j = lane; j < opN; j += 64) mx = fmaxf(mx, zr[j]);
                    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    float sum = 0.f;
                    for (int j = lane; j < opN; j += 64) sum += expf(zr[j] - mx);
                    sum = wave_sum(sum);
                    for (int j = lane; j < opN; j += 64) zr[j] = expf(zr[j] - mx) / sum;
                } else if (qaux == 2) {
                    for (int j = lane; j < opN; j += 64) zr[j] = sigmoidf_(zr[j]);
                }
            }
        } else if (kind == COP_FINAL_BWD) {
            if (wave < kR4) {
                const float* zr = slots + qyslot * kR4 * kCL + wave * kCL;
                const float* gr = src + wave * kCL;
                float* o = dst + wave * kCL;
                if (qaux == 1) {
                    float dot = 0.f;
                    for (int j = lane; j < opN; j += 64) dot += gr[j] * zr[j];
                    dot = wave_sum(dot);
                    for (int j = lane; j < opN; j += 64) o[j] = zr[j] * (gr[j] - dot);
                } else if (qaux == 2) {
                    for (int j = lane; j < opN; j += 64) o[j] = gr[j] * zr[j] * (1.f - zr[j]);
                } else {
                    for (int j = lane; j < opN; j += 64) o[j] = gr[j];
                }
                for (int j = opN + lane; j < kCL; j += 64) o[j] = 0.f;
            }
        } else if (kind == COP_PRIOR) {
            if (ecol < kCL)
            for (int eh = 0; eh < kC4E; ++eh) {
                const int erow = erow0 + kC4ER * eh;
                const int grow = r0 + erow, n = opN;
                const uint64_t k = key ^ (100ull * 0xA0761D6478BD642Full);
                float v = 0.f;
                if (erow < nrows && ecol < n) {
                    if (grow >= qrow_split) v = qfake_slot >= 0 ? (slots + qfake_slot * kR4 * kCL)[erow * kCL + ecol] : qW[(size_t)grow * qldw + ecol];
                    else if (qaux_ptr) v = qaux_ptr[(size_t)grow * qaux_ld + ecol] * qscale;
                    else if (qaux == 0) {          // gauss: Box-Muller on two words of the counter generator
                        const uint32_t u1 = hash_cell(k, (uint32_t)(grow + qgrow0), (uint32_t)(2 * ecol));
                        const uint32_t u2 = hash_cell(k, (uint32_t)(grow + qgrow0), (uint32_t)(2 * ecol + 1));
                        const float f1 = ((float)(u1 >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
                        const float f2 = (float)(u2 >> 8) * (1.0f / 16777216.0f);
                        v = sqrtf(-2.0f * logf(f1)) * cosf(6.283185307179586f * f2) * qscale;
                    } else if (qaux == 1) {        // categorical: one-hot of a uniform class per row
                        const uint32_t u = hash_cell(k, (uint32_t)(grow + qgrow0), 0xFFFFFFFFu);
                        v = ((int)(u % (uint32_t)n) == ecol) ? qscale : 0.f;
                    }                                // bernoulli: the reference's randint(0, 1) is always 0 (aae.py:86-88)
                }
                dst[erow * kCL + ecol] = v;
            }
        } else if (kind == COP_DISC_HEAD) {
            // one wave per row: the discriminator's 1-unit output layer, its loss and its dX (see chain.h)
            if (wave < kR4) {
                const EpiCtx ec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, slots);
                const int lrow = wave, grow = r0 + lrow;
                const int Kk = opK, Nn = opN;
                float wv[4], dot = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    wv[j] = qW[min(k, Kk - 1)];
                    if (k < Kk) dot += src[lrow * kCL + k] * wv[j];
                }
                const float logit = wave_sum(dot);
                float gv = 0.f;
                if (lrow < nrows) {
                    const float dv = sigmoidf_(logit);
                    const int Bsplit = qrow_split;
                    const float invB = 1.f / (float)Bsplit;
                    float l, gg;
                    if (qaux == 0 && grow >= Bsplit) { l = logf(1.f - dv + kTiny); gg = invB / (1.f - dv + kTiny); }
                    else { l = logf(dv + kTiny); gg = -invB / (dv + kTiny); }
                    gv = gg * dv * (1.f - dv) * qscale;
                    if (lane == 0) {
                        if (P.loss_terms) P.loss_terms[grow] = -l * invB;       // (summed by the weight-gradient launch behind the program)
                        else atomicAdd(P.loss_out + P.loss_slot, -l * invB);
                        if (qaux_ptr) qaux_ptr[(size_t)grow * qaux_ld] = gv;
                    }
                }
                const float* ys = slots + qyslot * kR4 * kCL;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    if (k < kCL) {
                        float v = 0.f;
                        if (lrow < nrows && k < Nn) {
                            v = gv * wv[j] * act_grad_from_y(ec.act, ys[lrow * kCL + k]);
                            if (ec.den) v *= chain_keep(ec, grow, k) ? ec.mk : 0.f;
                        }
                        dst[lrow * kCL + k] = v;
                    }
                }
            }
        }   // COP_STORE: only the stores below.  (COP_ADV / COP_REPARAM*: VAE programs stay on chain.h's kernel)
        if (TS && tsp && oi == 2 && tid == 0) tsp[64 + 53] = wall_clock64();
        chain_barrier();
        if (TS && tsp && oi == 2 && tid == 0) tsp[64 + 54] = wall_clock64();
#ifndef C4_NO_ONECOL
        if (one_col >= 0 && !one_done) {
            if (tid < kR4) dst[tid * kCL + one_col] = tid < nrows ? 1.f : 0.f;
            chain_barrier();
        }
#endif
        for (int eh = 0; eh < kC4E; ++eh) {
            const int erow = erow0 + kC4ER * eh;
            if (erow < nrows && ecol < opN) {
                if (outp) outp[(size_t)(out_row0 + r0 + erow) * ldo + ecol] = dst[erow * kCL + ecol];
                if (out2p) out2p[(size_t)(r0 + erow) * ldo2 + ecol] = dst[erow * kCL + ecol];
            }
        }
    }
    if (TS && tsp && tid == 0) tsp[nops] = wall_clock64();
}

}  // namespace aae
