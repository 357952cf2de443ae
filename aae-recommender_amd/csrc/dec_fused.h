// Fused decoder output layer for reference-sized batches (B <= ~104 at h = 200):
//
//   per tile of 32 items, one persistent 1024-thread workgroup (16 waves, 4 per SIMD) per CU:
//     S0     V3a tile (prefetched registers) -> LDS; request tile t+1's V3a and CSR entries
//     GEMM1  logits[B x 32]   = dh2[B x (h+1)] * V3a[32 x (h+1)]^T          (dec.lin3, aae.py:176)
//            epilogue: BCE of every cell against a zero target -> G = dL/dlogits, loss (aae.py:177, 693-695)
//     S2     the tile's CSR entries (non-zero targets) replace their cell's G and loss term
//     GEMM2  dV3a[32 x (h+1)] = G^T * dh2
//     GEMM3  dA2[B x (h+1)]  += G * V3a[32 x (h+1)]   (accumulated in registers across the tiles
//                                                      of the workgroup, one slab per workgroup)
//     S5     dec_optim (Adam) on this V3a tile, in place (or gradient export)
//
// The three GEMMs run on v_mfma_f32_16x16x4_f32 from LDS-resident operands: dh2 is loaded once
// per workgroup (B x 210 floats), the V3a tile once per tile and serves GEMM1, GEMM3 and the
// optimiser (p), so the layer's weights cross HBM exactly once per step: 4 B read + 8 B (m, v)
// read + 12 B written per parameter = the 24 B/param floor of a fused Adam, and dL/dlogits
// [B x N] never exists in HBM.  Next tile's V3a and this tile's m, v are in flight (registers)
// while the matrix cores work; barriers order LDS only (lds_barrier) so they do not drain them.
// All LDS operand patterns are bank-conflict free: row-major operands (k contiguous) are read two k at a
// time with ds_read_b64 and rely on the strides (kSD, kSG = 4 * odd); k-strided operands (ds_read_b32) on
// walking k in rows 4 apart (GEMM2) resp. on the V3a image keeping items i and i + 2 four rows apart
// (GEMM3, v3_row()).
//
// Debug: AAE_DEC_SKIP (phase ablation mask) and AAE_DEC_TS (100 MHz phase timeline of workgroup 0).
//
// Non-zero BCE targets come as per-tile entry lists (items sorted into 32-item buckets by
// tile_hist/scan/fill below, a counting sort over the batch's CSR entries).
#pragma once
#include <type_traits>
#include "device_common.h"
#include "gemm_f32.h"
#include "kernels.h"
#include "buckets.h"

namespace aae {

constexpr int kSD = 212;       // LDS row stride of the dh2 / V3a images: 4 * odd.  Row-major operand reads are ds_read_b64
                               // (2 k per lane, 256 B/clk instead of the 128 B/clk of ds_read_b32 - the matrix phases
                               // are LDS-bound: doubling GEMM1's reads costs +2.4 us per tile): 16 rows x 2 lanes x
                               // 8 bytes of a half-wave fall on 64 distinct banks
constexpr int kSG = 36;        // LDS row stride of the G tile (4 * odd: the same property for GEMM3's row-major reads)
constexpr int kSO = 212;       // LDS row stride of the dV3a tile (16-byte aligned rows for the float4 epilogue)
constexpr int kGR = 16 * kMB;  // rows of the G tile in LDS (rows >= B are zero)
constexpr int kNT = 1024;      // threads per workgroup: 4 waves per SIMD hide the LDS-operand latency of the
constexpr int kNW = kNT / 64;  // MFMA chains by wave switching (a wave's own chain is load -> wait -> MFMA)

// LDS row of item n of the tile in the V3a image.  GEMM3 walks the items (its k) in chunks of 8, lane fk of a
// k-quad taking items 8c + 2fk + j (j = 0, 1: one ds_read_b64 of G feeds two MFMAs), and reads one V3a row per
// k with ds_read_b32: the rows of fk = 0 / 1 (and 2 / 3), i.e. of items i and i + 2, have to sit 4 (mod 8) rows
// apart for their 16 columns to fall on disjoint banks at stride kSD (4 * 212 = 16 mod 32).  Swapping bits 1 and
// 2 of the item index does that; GEMM1 reads 16 consecutive items = 16 distinct rows (mod 16) either way.
__device__ __forceinline__ int v3_row(int n) { return (n & ~6) | ((n & 2) << 1) | ((n & 4) >> 1); }

struct TileEntries {           // CSR entries of the batch bucketed by item tile
    const int* start;          // [ntiles + 1]
    const int* eb;             // doc (batch row) of the entry
    const int* en;             // item index inside its tile
    const float* ev;           // target value
};

struct DecFusedArgs {
    const float* dh2; int ldh;           // [B][ldh] decoder hidden activations with the constant-1 column at h
    float* V3a; float* M; float* V; int ldv;   // [N][ldv] augmented weights + Adam moments
    float* gradV3;                        // export mode: gradient goes here, no update
    int N, B, h;                          // K of GEMM1 = h + 1
    float gscale;
    TileEntries te;
    float* slabs; size_t slab_stride; int ld_slab;   // dA2 partial per workgroup
    float* partials;                      // loss partial per workgroup
    const OptScalars* sc;
    int dbg_skip;                         // timing-only ablation mask (AAE_DEC_SKIP), 0 in production
    unsigned long long* ts;               // debug (AAE_DEC_TS): 100 MHz phase timestamps of workgroup 0, tile 5; else NULL
    float* Gt;                            // split form: dL/dlogits tile-major [tile][B][32] (written by MODE 1, read by MODE 2)
    // row-blocked form (batches beyond one launch's 112 rows, section 3.2e of DESIGN.md): this launch covers rows
    // [erow0, erow0 + B) of the batch - dh2 / slabs / partials / Gt already point at the block, the tile's entry lists
    // name rows of the WHOLE batch; acc (kDecOptAcc) = the dV3 partial of the earlier row blocks, added before the
    // optimiser (or before the partial is stored again: gradV3 != NULL)
    int erow0;
    const float* acc;
    // ... all row blocks in ONE critical launch (kDecCrit, nblk > 1): workgroup w takes row block w % nblk (Bb rows each,
    // the last one the remainder of B) and the tiles w / nblk, + gridDim.x / nblk, ...; B, dh2, slabs, partials, Gt and
    // erow0 then describe the WHOLE batch and the kernel derives its block's.  Slab (w / nblk) receives the rows of every
    // block once: gridDim.x / nblk slabs in all.
    int nblk, Bb;
    int tpp = 0;
    int one_term = 0;       // dec_crit_x3.h kernels: 1 = every operand keeps the FIRST term of its split only, i.e. is rounded to bf16
                            // (bf16 mode on the same kernels: bf16 operands, fp32 accumulation)                          // dec_opt_blocks_x3_kernel (dec_crit_x3.h): number of tile groups
    // late join (abi_output_layer.h): dec_crit_x3_kernel copies dh2 and *sc here for the deferred launch, which then
    // reads nothing the next step's forward pass rewrites (NULL: no copy)
    float* dh2_snap = nullptr;
    OptScalars* sc_snap = nullptr;
};

// MODE of dec_fused_kernel.  The step's critical path needs only dL/d(dh2) from this layer (the decoder's hidden
// backward waits for it); the weight gradient and the optimiser pass over V3 - all of the layer's HBM traffic - are
// needed by the NEXT step's forward.  The split form runs them as two launches: kDecCrit on the caller's stream,
// kDecOpt on a side stream with fewer workgroups than CUs, behind the latency-bound rest of the step (section 3.2c).
constexpr int kDecFused = 0;   // S0 GEMM1 S2 GEMM2 GEMM3 S5: everything in one launch
constexpr int kDecCrit = 1;    // S0 GEMM1 S2 GEMM3: logits, BCE, dA2 slabs, loss; stores the tile's dL/dlogits to Gt
constexpr int kDecOptAcc = 3;  // kDecOpt with a.acc added to the tile's gradient first
constexpr int kDecOpt = 2;     // S0 GEMM2 S5: dV3 from the stored dL/dlogits + dec_optim.  Its streams (V3, m, v, the stored
                               // tiles) are non-temporal: with plain loads / stores the 0.5 GB pass evicted the hidden layers'
                               // weights from L2 under the layer-chain kernels it runs beside (+25 % on each of them)

// LDS bytes the kernel needs for (B, h)
inline size_t dec_fused_lds_bytes(int B, int h) {
    (void)h;
    return sizeof(float) * ((size_t)B * kSD + (size_t)kTI * kSD + (size_t)kGR * kSG + (size_t)kTI * kSO + 64);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. it would wait for the NEXT tile's V3a, this tile's m / v and the
// previous tile's parameter stores at every phase boundary - exactly the traffic that is meant to
// stay in flight behind the matrix cores.  Global data is never exchanged between waves inside this
// kernel (each tile's rows are read and written by the same lanes), so LDS ordering is all it needs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The window of dec.lin3 (and of its moments: same shape) that the kernels' buffer descriptors cover.  A descriptor spans less
// than 2^31 bytes and its range check sees the scalar + vector offset, but the layer of a 2.9 M-item vocabulary is 2.4 GB
// (r1-r4 sent such layers to the three streaming GEMMs).  A workgroup walks its tiles in ascending order, 0.3-8 MB apart, and
// touches the tile in hand and the next one or two: the window starts at tile seg0 and moves up - at the START of a tile's
// iteration, to that tile - once the tile lies 2^30 bytes into it; what was requested through the old descriptor is unaffected
// (a descriptor is read when the instruction issues).  All scalar: a compare per tile, never taken below 2^30 bytes.
template <bool WIN>
struct X3WindowT {
    int seg0; unsigned seg_tiles, tile_bytes; size_t total;
    __device__ X3WindowT(int N, int ldv, unsigned tb) : seg0(0), seg_tiles((1u << 30) / tb), tile_bytes(tb), total((size_t)N * ldv * sizeof(float)) {}
    __device__ __forceinline__ bool moves(int tile) {
        if ((unsigned)(tile - seg0) <= seg_tiles) return false;
        seg0 = tile;
        return true;
    }
    __device__ __forceinline__ unsigned so(int tile) const { return (unsigned)(tile - seg0) * tile_bytes; }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const float* base) const {      // (beyond the tensor: reads return zero, stores are dropped)
        const size_t off = (size_t)seg0 * tile_bytes;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + off / sizeof(float), 0,
                                                 (unsigned)min((size_t)0x7FFFFFF0u, total - off), 0x00020000);
    }
};
// WIN = false, every layer below 2^31 bytes: ONE descriptor over the whole tensor, the tile's byte offset from its start -
// r1-r4's code, instruction for instruction (the moving window costs the critical launch five scalar registers it does not
// have: with it the 13-block instantiation spilled four vector registers).  The dispatcher picks the instantiation
// (x3_big_span).
template <>
struct X3WindowT<false> {
    unsigned tile_bytes, tbytes;
    __device__ X3WindowT(int N, int ldv, unsigned tb) : tile_bytes(tb), tbytes((unsigned)min((size_t)0x7FFFFFF0u, (size_t)N * ldv * sizeof(float))) {}
    __device__ __forceinline__ bool moves(int) const { return false; }
    __device__ __forceinline__ unsigned so(int tile) const { return (unsigned)tile * tile_bytes; }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const float* base) const {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, tbytes, 0x00020000);
    }
};
// does a layer of n_items x ld floats (+ its padding tiles) need the moving window?
inline bool x3_big_span(int n_items, int ld) { return ((size_t)n_items + 2 * 32) * (size_t)ld * sizeof(float) >= ((size_t)1 << 31) - ((size_t)1 << 24); }


template <int NB, int MODE = kDecFused, bool WIN = false>   // NB = ceil((h + 1) / 16) column blocks; WIN: a layer beyond 2^31 bytes (X3WindowT)
__global__ __launch_bounds__(kNT) void dec_fused_kernel(DecFusedArgs a) {
    constexpr bool kFwd = MODE != kDecOpt && MODE != kDecOptAcc;     // GEMM1, entries, GEMM3, loss, slabs
    constexpr bool kOpt = MODE != kDecCrit;    // GEMM2, optimiser
    constexpr bool kIsOpt = MODE == kDecOpt || MODE == kDecOptAcc;
    constexpr bool kAccIn = MODE == kDecOptAcc;
    constexpr int kAux = kIsOpt ? 2 : 0;       // buffer-store cache policy: 2 = nt
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // row blocks of one launch (the critical launch only): this workgroup's block and its share of the tiles
    constexpr bool kBlk = MODE == kDecCrit;
    const int nblk = kBlk && a.nblk > 1 ? a.nblk : 1;
    const int blk = nblk > 1 ? (int)blockIdx.x % nblk : 0, wgi = nblk > 1 ? (int)blockIdx.x / nblk : (int)blockIdx.x;
    const int wgs = nblk > 1 ? (int)gridDim.x / nblk : (int)gridDim.x;
    if (nblk > 1 && wgi >= wgs) return;                // (a grid that does not divide by the block count: the rest idles)
    const int erow0 = nblk > 1 ? blk * a.Bb : a.erow0;
    const int B = nblk > 1 ? min(a.Bb, a.B - erow0) : a.B;
    const float* dh2_blk = nblk > 1 ? a.dh2 + (size_t)erow0 * a.ldh : a.dh2;
    float* slabs_blk = nblk > 1 ? a.slabs + (size_t)erow0 * a.ld_slab : a.slabs;
    float* Gt_blk = nblk > 1 ? a.Gt + (size_t)blk * ((a.N + kTI - 1) / kTI) * a.Bb * kTI : a.Gt;
    float* dhs = lds;                                  // [B][kSD]
    float* v3s = dhs + (size_t)B * kSD;                // [32][kSD]
    float* gs = v3s + kTI * kSD;                       // [kGR][kSG]  logits, then dL/dlogits
    float* os = gs + kGR * kSG;                        // [32][kSO]   dV3a tile
    float* red = os + kTI * kSO;                       // [64]
    float* raw = os;                                   // [16*kMB][kSG] raw logits of the tile, between GEMM1 and GEMM2

    const int tid = threadIdx.x, lane = tid & 63;
    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[10] = wall_clock64();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: block ids below are scalars
    const int fr = lane & 15, fk = lane >> 4;
    const int K1 = a.h + 1, ldv = a.ldv;
    // k-steps rounded up to the unroll factor 4: the extra ones multiply zero padding (columns >= ldh
    // of both LDS images are zeroed once, G rows >= B are zero)
    const int kch1 = (((K1 + 7) >> 3) + 1) & ~1;       // GEMM1: 8-k chunks over the h+1 hidden columns (even count;
                                                       // columns >= ldh of both LDS images are zero), 8 * kch1 <= kSD
    const int nmb = (B + 15) >> 4;                      // 16-row blocks actually present (<= kMB)
    // A last row block of at most 4 rows (the reference's default batch: 100 = 6 * 16 + 4) does not run as a 7th 16-row
    // block - 12 of its 16 rows would be padding in GEMM1 and GEMM3 - but on v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 x 4
    // per instruction, 8 cycles: the same 64 flop/clk/SIMD with a quarter of the rows), spread over waves that have room:
    // GEMM1 on waves 12..15 (8 items x 8 k-eighths each, reduced over the k-eighths by lane shuffles), GEMM3 on all 16
    // (64 columns x an item quarter each; the four item quarters of a column group meet in LDS at the kernel's end).
    // (Only the critical launch of the split carries it: in the one-launch form the extra accumulators on top of the
    //  optimiser's V/m/v registers push the kernel over 128 VGPRs - 15 spilled, 13 % slower on a 2.2M-item shard.)
    constexpr bool kTail = MODE == kDecCrit;
    const bool tail4 = kTail && nmb > 1 && B - 16 * (nmb - 1) <= 4 && !(a.dbg_skip & 256);     // (256: A/B switch back to the 7th block)
    const int nfb = tail4 ? nmb - 1 : nmb;              // full 16-row blocks of the forward products
    const int kq = (((K1 + 7) >> 3) + 1) & ~1;          // GEMM1 tail: k-steps per eighth (even; 8 * kq <= kSD)
    f32x4 acc3t = (f32x4){0.f, 0.f, 0.f, 0.f};          // GEMM3 tail: rows 16 nfb + 0..3 of column 64 (wave & 3) + lane
    const int ntiles = (a.N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4;                    // ldv % 4 == 0
    const int tile_f4 = kTI * f4_per_row;              // float4 per tile span
    constexpr int NV = 2;                              // float4 slots per thread (tile_f4 <= kNT * 2)
    const OptScalars sc = *a.sc;
    const bool do_adam = a.gradV3 == nullptr;

    // ---- once per workgroup: dh2 -> LDS, zero the G tile (rows >= B stay zero)
    for (int f = tid; f < B * (a.ldh / 4); f += kNT) {
        int r = f / (a.ldh / 4), c4 = f % (a.ldh / 4);
        float4 x = *reinterpret_cast<const float4*>(dh2_blk + (size_t)r * a.ldh + c4 * 4);
        float* d = dhs + r * kSD + c4 * 4;
        *reinterpret_cast<float2*>(d) = make_float2(x.x, x.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(x.z, x.w);
    }
    for (int i = tid; i < kGR * kSG; i += kNT) gs[i] = 0.f;
    for (int i = tid; i < (B + kTI) * (kSD - a.ldh); i += kNT) {      // pad columns of dhs and v3s (contiguous rows)
        const int r = i / (kSD - a.ldh), cidx = a.ldh + i % (kSD - a.ldh);
        dhs[r * kSD + cidx] = 0.f;
    }

    f32x4 acc3[(kMB * NB + kNW - 1) / kNW];
#pragma unroll
    for (int j = 0; j < (kMB * NB + kNW - 1) / kNW; ++j) acc3[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float loss = 0.f;

    // All loads of a span are issued unconditionally from clamped (always valid) addresses and masked
    // afterwards: a load under a lane-dependent `if` makes hipcc branch around it and wait for it
    // (vmcnt(0)) before the next one, which serialises the HBM round trips.
    const size_t last_f4 = ((size_t)a.N * ldv) / 4 - 1;
    auto load_span = [&](const float* base, int tile, float4* r) {
        const size_t f0 = (size_t)((a.dbg_skip & 128) ? (int)blockIdx.x : tile) * kTI * f4_per_row;   // 128: L2-resident ablation
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t f = f0 + (size_t)(tid + kNT * j);
            if (kIsOpt) { const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + (f < last_f4 ? f : last_f4)); r[j] = make_float4(t4[0], t4[1], t4[2], t4[3]); }
            else r[j] = reinterpret_cast<const float4*>(base)[f < last_f4 ? f : last_f4];
        }
        // no masking here: a use right after the load would make the wave wait for it; clamped lanes
        // hold finite values that are either never stored or multiplied by zero gradients
    };

    // Software pipeline over the tiles of this workgroup.  Inside the loop no global load is consumed in
    // the phase that issues it (hipcc would wait for it AND everything older with s_waitcnt vmcnt(0), and
    // loads return in order - one such load exposes the whole prefetch):
    //   V3a(t+1), the CSR entries of tile t+1 and the entry range of tile t+2 are requested in S0(t),
    //   m/v(t) before GEMM2(t); parameter stores of S5(t) retire behind GEMM3(t).
    float4 vreg[NV], mreg[NV], sreg[NV], areg[NV];
    typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
    X3WindowT<WIN> win(a.N, ldv, (unsigned)(kTI * ldv) * 4u);       // (r5: the window of the tensors the store descriptors cover - layers beyond 2^31 bytes)
    const float* gbase = a.gradV3 ? a.gradV3 : a.V3a;
    __amdgpu_buffer_rsrc_t rP = win.desc(a.V3a), rM = win.desc(a.M), rV = win.desc(a.V), rG = win.desc(gbase);
    int tile = wgi;
    const int stride = wgs;
    const int last_e = max(a.te.start[ntiles] - 1, 0);        // clamp for the unconditional entry loads
    // (VECTOR loads through an index the compiler cannot prove uniform: as scalar loads they sit in lgkmcnt, and the
    // LDS-only barrier of the phase that issues them - s_waitcnt lgkmcnt(0) - waits out their L2 round trip once per tile)
    int ozr;
    asm volatile("v_mov_b32 %0, 0" : "=v"(ozr));
    auto load_range = [&](int t, int& lo, int& hi) {
        const int tc = min(t, ntiles - 1) + ozr;
        lo = a.te.start[tc]; hi = a.te.start[tc + 1];
        if (t >= ntiles) hi = lo;
    };
    int ce0 = 0, ce1 = 0, ne0 = 0, ne1 = 0, fe0 = 0, fe1 = 0;  // entry ranges: current, next, the one after
    int ent_b = 0, ent_n = 0; float ent_v = 0.f;               // this thread's entry of the NEXT tile
    auto load_entry = [&](int lo) {
        const int e = min(lo + tid, last_e);
        ent_b = a.te.eb[e]; ent_n = a.te.en[e]; ent_v = a.te.ev[e];
    };
    // split form: this thread's float4 of the tile's stored dL/dlogits ([B][32] floats per tile, contiguous)
    float4 greg = make_float4(0.f, 0.f, 0.f, 0.f);
    const int g_f4 = B * (kTI / 4);                    // float4 per tile (<= kNT: B <= 16 * kMB <= 128)
    const unsigned gbytes = (unsigned)min((size_t)0x7FFFFFF0u, (size_t)ntiles * g_f4 * 16);
    const __amdgpu_buffer_rsrc_t rGt = __builtin_amdgcn_make_buffer_rsrc(Gt_blk ? Gt_blk : a.V3a, 0, gbytes, 0x00020000);
    auto load_g = [&](int t) {
        if (kIsOpt) { const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Gt_blk) + (size_t)t * g_f4 + min(tid, g_f4 - 1)); greg = make_float4(t4[0], t4[1], t4[2], t4[3]); }
        else greg = reinterpret_cast<const float4*>(Gt_blk)[(size_t)t * g_f4 + min(tid, g_f4 - 1)];
    };
    if (tile < ntiles) {
        load_span(a.V3a, tile, vreg);
        if (kFwd) {
            load_range(tile, ne0, ne1);
            load_range(tile + stride, fe0, fe1);
            load_entry(ne0);
        }
        if (kIsOpt) load_g(tile);
    }

    // tile-invariant addressing of this thread's NV float4 slots of a tile span (no division in the loop)
    int s_rc[NV];                                       // row * 64 + float4 column (ldv <= 252)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int fc = min(tid + kNT * j, tile_f4 - 1), row = fc / f4_per_row;
        s_rc[j] = row * 64 + (fc - row * f4_per_row);
    }
    auto slot_v3 = [&](int j) { return v3_row(s_rc[j] >> 6) * kSD + (s_rc[j] & 63) * 4; };
    auto slot_os = [&](int j) { return (s_rc[j] >> 6) * kSO + (s_rc[j] & 63) * 4; };
    int iter = 0;
    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[11] = wall_clock64();
    auto stamp = [&](int k) { if (a.ts && blockIdx.x == 0 && tid == 0 && iter == 5) { a.ts[k] = wall_clock64(); if (k == 0 || k == 6) a.ts[8 + k / 6] = clock64(); } };
    for (; tile < ntiles; tile += stride, ++iter) {
        const int i0 = tile * kTI;
        if (win.moves(tile)) { rP = win.desc(a.V3a); rM = win.desc(a.M); rV = win.desc(a.V); rG = win.desc(gbase); }
        // A zero the compiler cannot see through: the LDS operand pointers of the three GEMMs are built from
        // it, so they are recomputed per tile (~30 VALU) instead of being hoisted out of the tile loop and held
        // in ~20 VGPRs across every phase - at 128 VGPRs (16 waves/CU) that hoisting spilled to scratch, and a
        // scratch reload waits for every older global load in flight.
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int frz = fr + oz;
        stamp(0);
        lds_barrier();                                 // previous tile's readers of v3s / gs are done
        // ---- S0: V3a tile registers -> LDS; rotate the pipeline registers and request the next stage
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (tid + kNT * j < tile_f4) {
                float* d = v3s + slot_v3(j);
                *reinterpret_cast<float2*>(d) = make_float2(vreg[j].x, vreg[j].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(vreg[j].z, vreg[j].w);
            }
        }
        ce0 = ne0; ce1 = ne1; ne0 = fe0; ne1 = fe1;
        const int my_rb = ent_b - erow0;              // (row-blocked launches: entries of other row blocks are not ours)
        const int my_p = my_rb * kSG + ent_n; const float my_v = ent_v;
        if (kIsOpt) {
            if (tid < g_f4) *reinterpret_cast<float4*>(gs + (tid >> 3) * kSG + (tid & 7) * 4) = greg;
            load_g(min(tile + stride, ntiles - 1));
        }
        load_span(a.V3a, min(tile + stride, ntiles - 1), vreg);
        if (kFwd) {
            load_range(tile + 2 * stride, fe0, fe1);
            load_entry(ne0);
        }
        lds_barrier();
        stamp(1);

        if (kFwd && !(a.dbg_skip & 1) && tail4 && wave >= kNW - 4) {
            // ---- S1, tail rows: logits[16 nfb + i][8 w4 + 4g + j] over the k-eighth ke; lane = 8 ke + 4 g + j
            const int w4 = wave - (kNW - 4);
            const int j4 = lane & 3, ke = lane >> 3;
            const int item = 8 * w4 + (lane & 7);
            const float* pa = dhs + min(16 * nfb + j4, B - 1) * kSD + ke * kq;
            const float* pb = v3s + v3_row(item) * kSD + ke * kq + oz;
            f32x4 t0 = (f32x4){0.f, 0.f, 0.f, 0.f}, t1 = t0;
            for (int k = 0; k < kq; k += 2) {
                const float2 xa = *reinterpret_cast<const float2*>(pa + k);
                const float2 xb = *reinterpret_cast<const float2*>(pb + k);
                t0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.x, xb.x, t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.y, xb.y, t1, 0, 0, 0);
            }
            t0 += t1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = t0[r];
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                t0[r] = v;
            }
            if (lane < 8) {
                const int rb = 16 * nfb, cb = item;
                const bool item_ok = i0 + cb < a.N && !(a.dbg_skip & 2);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rb + r < B) {
                        float g = 0.f, l = 0.f;
                        if (item_ok) bce_elem_t0(t0[r], a.gscale, g, l);
                        gs[(rb + r) * kSG + cb] = g;
                        raw[(rb + r) * kSG + cb] = t0[r];
                        loss += l;
                    }
            }
        } else
        if (kFwd && !(a.dbg_skip & 1) && wave < 2 * nfb)
        // ---- S1: GEMM1 logits[b][n] = sum_k dh2[b][k] * V3a[n][k]; blocks (mb, nb2) id = mb*2 + nb2,
        // one block per wave; `wave` is a scalar, so waves past the last block skip with a scalar branch and
        // leave the matrix pipe of their SIMD to the others
        {
            f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int id0 = wave;
            // k-permutation: lane (fr, fk) supplies k = 8c + 2fk + j to MFMA j of chunk c for BOTH operands, so one
            // 8-byte LDS read per operand feeds two MFMAs; two chunks per trip, even / odd accumulators
            const float* pa0 = dhs + min((id0 >> 1) * 16 + frz, B - 1) * kSD + 2 * fk;
            const float* pb0 = v3s + v3_row((id0 & 1) * 16 + frz) * kSD + 2 * fk;
            f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < kch1; c += 2) {
                const float2 xa = *reinterpret_cast<const float2*>(pa0 + 8 * c);
                const float2 ya = *reinterpret_cast<const float2*>(pb0 + 8 * c);
                const float2 xb = *reinterpret_cast<const float2*>(pa0 + 8 * c + 8);
                const float2 yb = *reinterpret_cast<const float2*>(pb0 + 8 * c + 8);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, ya.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, ya.y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.x, yb.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb.y, yb.y, c1, 0, 0, 0);
            }
            c0 += c1;
            // Epilogue = BCE with a zero target, the case of all but a handful of cells: dL/dlogit -> gs[b][n]
            // (C map: row = 4*(lane>>4) + r, col = lane & 15).  The raw logit also goes to the (still unused)
            // dV3a buffer for the cells that do have a target.
            {
                const int rb = (id0 >> 1) * 16 + fk * 4, cb = (id0 & 1) * 16 + fr;
                const bool item_ok = i0 + cb < a.N && !(a.dbg_skip & 2);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rb + r < B) {
                        float g = 0.f, l = 0.f;
                        if (item_ok) bce_elem_t0(c0[r], a.gscale, g, l);
                        gs[(rb + r) * kSG + cb] = g;
                        raw[(rb + r) * kSG + cb] = c0[r];
                        loss += l;
                    }
            }
        }
        if (kFwd) lds_barrier();
        stamp(2);

        // ---- S2: the CSR entries of the tile (non-zero targets) replace their cell's gradient and loss term
        if (kFwd) {
            const int e0 = ce0, e1 = (a.dbg_skip & 2) ? e0 : ce1;
            if (tid < e1 - e0 && (unsigned)my_rb < (unsigned)B) {      // the prefetched entry of this thread
                float g0, l0, g1, l1;
                bce_elem_t0(raw[my_p], a.gscale, g0, l0);
                bce_elem(raw[my_p], my_v, a.gscale, g1, l1);
                loss += l1 - l0;
                gs[my_p] = g1;
            }
            for (int e = e0 + kNT + tid; e < e1; e += kNT) { // tiles with more than 1024 entries (tiny vocabularies)
                const int rb = a.te.eb[e] - erow0;
                if ((unsigned)rb >= (unsigned)B) continue;
                const int p = rb * kSG + a.te.en[e];
                float g0, l0, g1, l1;
                bce_elem_t0(raw[p], a.gscale, g0, l0);
                bce_elem(raw[p], a.te.ev[e], a.gscale, g1, l1);
                loss += l1 - l0;
                gs[p] = g1;
            }
        }
        if (kFwd) lds_barrier();

        stamp(3);
        if (MODE == kDecCrit) {
            // the finished dL/dlogits tile -> Gt for the deferred launch; the store retires behind GEMM3 (a lane without
            // a cell gets an offset beyond the descriptor: no branch around the store, see S5)
            const float4 gq = *reinterpret_cast<const float4*>(gs + (min(tid, g_f4 - 1) >> 3) * kSG + (tid & 7) * 4);
            const unsigned go = tid < g_f4 ? (unsigned)tid * 16u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, gq), rGt, go, (unsigned)tile * (unsigned)g_f4 * 16u, 0);
        }
        // the optimiser moments of this tile travel while GEMM2 and GEMM3 run
        // (unconditional: the moment tensors exist in every mode, and a load under a condition is waited for on the spot)
        if (kOpt) { load_span(a.M, tile, mreg); load_span(a.V, tile, sreg); }
        if (kAccIn) load_span(a.acc, tile, areg);
        // ---- S3: GEMM2 dV3a[item][c] = sum_b G[b][item] * dh2[b][c]; blocks id = nb*2 + ib, a wave owns the ids
        // wave + 16q: they share the item half ib (one G read serves them all) and differ in the column block.
        // k runs over the batch rows in groups of 16: k-step (g, j) multiplies rows 16g + j + 4*fk.  Rows 4
        // apart are 16 banks apart in both operands (4 * kSG = 4 * kSD = 16 mod 32), so the two k of a
        // half-wave never collide; the order of the rows inside the sum is free.
        auto gemm2 = [&](auto NQ) {
            constexpr int nq = decltype(NQ)::value;
            constexpr int NA = nq == 1 ? 2 : 1;                       // one block: even / odd k-steps alternate accumulators
            f32x4 acc2[nq][NA];
            const float* pd[nq];
            const float* pg = gs + 4 * fk * kSG + (wave & 1) * 16 + frz;      // G[b = 16g + j + 4fk][item]
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                for (int u = 0; u < NA; ++u) acc2[q][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                pd[q] = dhs + ((wave + kNW * q) >> 1) * 16 + frz;      // dh2[b][col]
            }
            const int nfull = B >> 4;                                 // groups whose 16 rows all exist: no row clamp,
            for (int g = 0; g < nfull; ++g) {                         // every LDS address = base + constant
                const float* xg = pg + 16 * g * kSG;
                const int yoff = (16 * g + 4 * fk) * kSD;
                float x[4], y[4][nq];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = xg[j * kSG];
#pragma unroll
                    for (int q = 0; q < nq; ++q) y[j][q] = pd[q][yoff + j * kSD];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < nq; ++q)
                        acc2[q][j & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], y[j][q], acc2[q][j & (NA - 1)], 0, 0, 0);
            }
            if (B & 15) {   // the partial last group: rows past B - 1 are clamped (their G rows are zero)
                const int g = nfull;
                float x[4], y[4][nq];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * g + j;
                    const int rowc = min(row + 4 * fk, B - 1) * kSD;
                    x[j] = pg[row * kSG];
#pragma unroll
                    for (int q = 0; q < nq; ++q) y[j][q] = pd[q][rowc];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < nq; ++q)
                        acc2[q][j & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], y[j][q], acc2[q][j & (NA - 1)], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                const int id = wave + kNW * q;
                const int rb = (id & 1) * 16 + fk * 4, cb = (id >> 1) * 16 + fr;
                const f32x4 c = NA == 2 ? acc2[q][0] + acc2[q][NA - 1] : acc2[q][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) os[(rb + r) * kSO + cb] = c[r];
            }
        };
        if (kOpt && !(a.dbg_skip & 4) && wave < 2 * NB) {
            constexpr int Q2 = (2 * NB + kNW - 1) / kNW;
            static_assert(kNW % 2 == 0, "a wave's blocks must share the item half");
            if (wave + kNW * (Q2 - 1) < 2 * NB) gemm2(std::integral_constant<int, Q2>{});     // scalar branch
            else gemm2(std::integral_constant<int, (Q2 > 1 ? Q2 - 1 : 1)>{});
        }

        // ---- S4: GEMM3 dA2[b][c] += sum_n G[b][n] * V3a[n][c]; blocks id = mb*NB + nb, ids wave + 16q.  It runs
        // after S5 so that the parameter stores retire behind these MFMAs, not at the next tile's first wait.
        if (kFwd && !(a.dbg_skip & 8)) {
            constexpr int Q3 = (kMB * NB + kNW - 1) / kNW;
            // scalar: the wave's last block exists, or is a duplicate that only keeps the code branch-free (small
            // batches have several duplicates: they still run, and are never stored)
            const bool full3 = wave + kNW * (Q3 - 1) < nfb * NB || (nmb < kMB && !tail4);
            const float* pg[Q3]; const float* pv[Q3];
#pragma unroll
            for (int q = 0; q < Q3; ++q) {
                const int id = min(wave + kNW * q, nfb * NB - 1);
                const int mb = id / NB, nb = id - mb * NB;
                pg[q] = gs + (mb * 16 + frz) * kSG + 2 * fk;        // G[b][n = 8c + 2fk + j]: float2 per chunk
                pv[q] = v3s + (((fk & 1) << 2) | (fk & 2)) * kSD + nb * 16 + frz;   // V3a row v3_row(8c + 2fk + j) = 8c + this + j
            }
#pragma unroll
            for (int c = 0; c < kTI / 8; ++c) {
                float2 x[Q3]; float y0[Q3], y1[Q3];
#pragma unroll
                for (int q = 0; q < Q3; ++q) {
                    x[q] = *reinterpret_cast<const float2*>(pg[q] + 8 * c);
                    y0[q] = pv[q][(8 * c) * kSD]; y1[q] = pv[q][(8 * c + 1) * kSD];
                }
#pragma unroll
                for (int q = 0; q < Q3; ++q)
                    if (q < Q3 - 1 || full3) {
                        acc3[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].x, y0[q], acc3[q], 0, 0, 0);
                        acc3[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].y, y1[q], acc3[q], 0, 0, 0);
                    }
            }
            if (tail4 && 64 * (wave & 3) < 16 * NB) {
                // tail rows: dA2[16 nfb + i][64 cg + lane] += sum over the item quarter n = 8 kq3 .. + 7 of G[row][n] * V3a[n][col]
                const int cg = wave & 3, kq3 = wave >> 2;
                const float* pgt = gs + (16 * nfb + (lane & 3)) * kSG + 8 * kq3 + oz;
                const float* pvt = v3s + min(64 * cg + lane, kSD - 1);
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc3t = __builtin_amdgcn_mfma_f32_4x4x1f32(pgt[n], pvt[v3_row(8 * kq3 + n) * kSD], acc3t, 0, 0, 0);
            }
        }
        if (kOpt) lds_barrier();                     // os complete
        stamp(4);

        // ---- S5: optimiser on the tile (or gradient export), whole rows, float4 per lane.  The stores are issued on
        // every path with the same count (no branch around them): a lane without a cell (beyond the tile span, rows past
        // the vocabulary, the moments in SGD / export mode) gets a buffer offset beyond the descriptor's range and the
        // bounds check drops it.  With stores under lane- or mode-dependent branches the compiler's wait counters lose
        // track of how many are in flight, and the next tile's first use of its prefetched V3a waited for all of them.
        if (kOpt && !(a.dbg_skip & 16))
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool valid = tid + kNT * j < tile_f4 && i0 + (s_rc[j] >> 6) < a.N && !(a.dbg_skip & 32);
            float4 g = *reinterpret_cast<const float4*>(os + slot_os(j));
            if (kAccIn) { g.x += areg[j].x; g.y += areg[j].y; g.z += areg[j].z; g.w += areg[j].w; }
            // the tile span is contiguous: byte offset = (i0 * ldv + 4 * slot) * 4
            const unsigned so = win.so((a.dbg_skip & 128) ? (int)blockIdx.x : tile);
            const unsigned vo = valid ? (unsigned)(tid + kNT * j) * 16u : 0x80000000u;
            const float* ps = v3s + slot_v3(j);
            float4 p = make_float4(ps[0], ps[1], ps[2], ps[3]);
            float4 mm = mreg[j], vv = sreg[j];
            if (!(a.dbg_skip & 64)) {
                adam_update(p.x, mm.x, vv.x, g.x, sc); adam_update(p.y, mm.y, vv.y, g.y, sc);
                adam_update(p.z, mm.z, vv.z, g.z, sc); adam_update(p.w, mm.w, vv.w, g.w, sc);
            }
            const float4 out = do_adam ? p : g;
            const unsigned vo2 = (do_adam && !sc.is_sgd) ? vo : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, out), do_adam ? rP : rG, vo, so, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, mm), rM, vo2, so, kAux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, vv), rV, vo2, so, kAux);
        }

        stamp(5);
        stamp(6);
    }

    if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[7] = wall_clock64();
    if (!kFwd) return;
    // ---- dA2 partial of this workgroup -> its slab; loss partial
    float* slab = slabs_blk + (size_t)wgi * a.slab_stride;
    if (tail4) {
        // the tail rows' four item-quarter partials of every column meet in LDS (the dV3a / raw-logit buffer is free now)
        __syncthreads();
        float* tl = os;                                    // [4 item quarters][4 rows][256 columns]
        const int cg = wave & 3, kq3 = wave >> 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[((kq3 * 4 + r) << 8) + 64 * cg + lane] = acc3t[r];
        __syncthreads();
        {
            const int r = tid >> 8, cc = tid & 255;         // 1024 threads = 4 rows x 256 columns
            const int row = 16 * nfb + r;
            if (row < B && cc < a.ld_slab)
                slab[(size_t)row * a.ld_slab + cc] = (tl[(r << 8) + cc] + tl[((4 + r) << 8) + cc]) + (tl[((8 + r) << 8) + cc] + tl[((12 + r) << 8) + cc]);
        }
    }
#pragma unroll
    for (int q = 0; q < (kMB * NB + kNW - 1) / kNW; ++q) {
        const int id = wave + kNW * q;
        if (id < nfb * NB) {
            const int mb = id / NB, nb = id - mb * NB;
            const int rb = mb * 16 + fk * 4, cb = nb * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rb + r < B && cb < a.ld_slab) slab[(size_t)(rb + r) * a.ld_slab + cb] = acc3[q][r];
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
        if (a.ts && blockIdx.x == 0) { a.ts[12] = wall_clock64(); a.ts[13] = (unsigned long long)iter; }
    }
}

// ---------------------------------------------------------------------------------------------
// The deferred half for ALL row blocks of a row-blocked step in one launch (fp32): dV3 = sum over the blocks of
// G_r^T * dh2_r from the tiles the critical launch stored, then dec_optim.  A workgroup owns up to kOBT tiles (wg, wg +
// gridDim.x, ...) and keeps their dV3 in registers while it walks the row blocks: a block's dh2 is loaded into LDS once
// and serves every tile of the workgroup.  (One launch per block - kDecOpt / kDecOptAcc with the partial sums going
// through HBM - costs 8 x 30 us on a 12.5 k-item slice, most of it prologue and pipeline fill for 3 tiles per workgroup.)
// Summation order over the rows is the per-block launches' (blocks in order, the same k walk inside a block).
// ---------------------------------------------------------------------------------------------
constexpr int kOBT = 4;        // tiles per workgroup of dec_opt_blocks_kernel
inline size_t dec_opt_blocks_lds_bytes(int Bb) { return sizeof(float) * ((size_t)Bb * kSD + (size_t)kGR * kSG + (size_t)kTI * kSO); }

template <int NB>
__global__ __launch_bounds__(kNT) void dec_opt_blocks_kernel(DecFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dhs = lds;                                  // [Bb][kSD]
    float* gs = dhs + (size_t)a.Bb * kSD;              // [kGR][kSG]
    float* os = gs + kGR * kSG;                        // [32][kSO]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int ldv = a.ldv, ntiles = (a.N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4, tile_f4 = kTI * f4_per_row;
    constexpr int NV = 2;
    constexpr int Q2 = (2 * NB + kNW - 1) / kNW;
    const OptScalars sc = *a.sc;
    typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
    const unsigned tbytes = (unsigned)min((size_t)0x7FFFFFF0u, (size_t)a.N * ldv * sizeof(float));
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(a.V3a, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(a.M, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, tbytes, 0x00020000);

    f32x4 acc[kOBT][Q2];
#pragma unroll
    for (int j = 0; j < kOBT; ++j)
#pragma unroll
        for (int q = 0; q < Q2; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < kGR * kSG; i += kNT) gs[i] = 0.f;          // rows >= the block's stay zero

    const float* pg = gs + 4 * fk * kSG + (wave & 1) * 16 + fr;      // G[b = 16g + j + 4fk][item]
    const float* pd[Q2];
#pragma unroll
    for (int q = 0; q < Q2; ++q) pd[q] = dhs + (min(wave + kNW * q, 2 * NB - 1) >> 1) * 16 + fr;
    const bool live[2] = {wave < 2 * NB, wave + kNW < 2 * NB};

    for (int r = 0; r < a.nblk; ++r) {
        const int r0 = r * a.Bb, Br = min(a.Bb, a.B - r0);
        const int g_f4 = Br * (kTI / 4);
        const f32x4* Gr = reinterpret_cast<const f32x4*>(a.Gt + (size_t)r * ntiles * a.Bb * kTI);
        lds_barrier();                                 // the previous block's readers of dhs / gs are done
        if (Br < a.Bb) for (int i = tid; i < kGR * kSG; i += kNT) gs[i] = 0.f;      // (a shorter last block: stale rows)
        for (int f = tid; f < Br * (a.ldh / 4); f += kNT) {
            const int row = f / (a.ldh / 4), c4 = f % (a.ldh / 4);
            const float4 x = *reinterpret_cast<const float4*>(a.dh2 + (size_t)(r0 + row) * a.ldh + c4 * 4);
            *reinterpret_cast<float4*>(dhs + row * kSD + c4 * 4) = x;
        }
        f32x4 gq = __builtin_nontemporal_load(Gr + (size_t)min((int)blockIdx.x, ntiles - 1) * g_f4 + min(tid, g_f4 - 1));
#pragma unroll
        for (int j = 0; j < kOBT; ++j) {
            const int tile = blockIdx.x + j * gridDim.x;
            if (tile >= ntiles) break;                 // (uniform)
            lds_barrier();                             // readers of the previous tile's G are done (first tile: dhs visible below)
            if (tid < g_f4) *reinterpret_cast<float4*>(gs + (tid >> 3) * kSG + (tid & 7) * 4) = make_float4(gq[0], gq[1], gq[2], gq[3]);
            {   // the next tile's G travels behind this tile's product
                const int nt = min(tile + (int)gridDim.x, ntiles - 1);
                gq = __builtin_nontemporal_load(Gr + (size_t)nt * g_f4 + min(tid, g_f4 - 1));
            }
            lds_barrier();
            // GEMM2 (dec_fused_kernel's walk): k-step (g, jj) multiplies rows 16g + jj + 4 fk
            const int nfull = Br >> 4;
            for (int g = 0; g <= nfull; ++g) {
                if (g == nfull && !(Br & 15)) break;
                float x[4], y[4][Q2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = 16 * g + jj;
                    const int rowc = min(row + 4 * fk, Br - 1) * kSD;      // (rows past the block: clamped, their G rows are zero)
                    x[jj] = pg[row * kSG];
#pragma unroll
                    for (int q = 0; q < Q2; ++q) y[jj][q] = pd[q][rowc];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int q = 0; q < Q2; ++q)
                        if (live[q]) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[jj], y[jj][q], acc[j][q], 0, 0, 0);
            }
        }
    }

    // ---- the optimiser on the workgroup's tiles
#pragma unroll
    for (int j = 0; j < kOBT; ++j) {
        const int tile = blockIdx.x + j * gridDim.x;
        if (tile >= ntiles) break;
        const int i0 = tile * kTI;
        float4 pr[NV], mr[NV], vr[NV];
        const size_t last_f4 = ((size_t)a.N * ldv) / 4 - 1;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const size_t f = (size_t)tile * kTI * f4_per_row + (size_t)(tid + kNT * u);
            const size_t fc = f < last_f4 ? f : last_f4;
            const f32x4 t0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.V3a) + fc);
            const f32x4 t1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.M) + fc);
            const f32x4 t2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.V) + fc);
            pr[u] = make_float4(t0[0], t0[1], t0[2], t0[3]); mr[u] = make_float4(t1[0], t1[1], t1[2], t1[3]);
            vr[u] = make_float4(t2[0], t2[1], t2[2], t2[3]);
        }
        lds_barrier();                                 // the previous tile's readers of os are done
#pragma unroll
        for (int q = 0; q < Q2; ++q)
            if (live[q]) {
                const int id = wave + kNW * q;
                const int rb = (id & 1) * 16 + fk * 4, cb = (id >> 1) * 16 + fr;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) os[(rb + rr) * kSO + cb] = acc[j][q][rr];
            }
        lds_barrier();
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int slot = tid + kNT * u;
            const int fc = min(slot, tile_f4 - 1), row = fc / f4_per_row, c4 = fc - row * f4_per_row;
            const bool valid = slot < tile_f4 && i0 + row < a.N;
            const float4 g = *reinterpret_cast<const float4*>(os + row * kSO + c4 * 4);
            float4 p = pr[u], mm = mr[u], vv = vr[u];
            adam_update(p.x, mm.x, vv.x, g.x, sc); adam_update(p.y, mm.y, vv.y, g.y, sc);
            adam_update(p.z, mm.z, vv.z, g.z, sc); adam_update(p.w, mm.w, vv.w, g.w, sc);
            const unsigned so = (unsigned)((size_t)i0 * ldv) * 4u;
            const unsigned vo = valid ? (unsigned)slot * 16u : 0x80000000u;
            const unsigned vo2 = !sc.is_sgd ? vo : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, p), rP, vo, so, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, mm), rM, vo2, so, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, vv), rV, vo2, so, 2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// counting sort of the batch's CSR entries into 32-item tiles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_hist_kernel(BatchView bv, int* __restrict__ tcount) {
    const int b = blockIdx.x;
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    for (int64_t e = lo + (int64_t)blockIdx.y * 256 + threadIdx.x; e < hi; e += (int64_t)gridDim.y * 256)
        atomicAdd(&tcount[bv.indices[e] / kTI], 1);
}

// exclusive scan of tcount[0..ntiles) -> tstart[0..ntiles]; tcount becomes the fill cursor (zeroed)
__global__ __launch_bounds__(1024) void tile_scan_kernel(int* __restrict__ tcount, int* __restrict__ tstart,
                                                         int ntiles) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (ntiles + 1023) / 1024;
    const int lo = t * per, hi = min(ntiles, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += tcount[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) {
        int c = tcount[i];
        tstart[i] = run;
        tcount[i] = 0;
        run += c;
    }
    if (t == 1023) tstart[ntiles] = part[1023];
}

__global__ __launch_bounds__(256) void tile_fill_kernel(BatchView bv, const int* __restrict__ tstart,
                                                        int* __restrict__ tcursor, int* __restrict__ eb,
                                                        int* __restrict__ en, float* __restrict__ ev) {
    const int b = blockIdx.x;
    const int dc = bv.doc(b);
    const int64_t lo = bv.indptr[dc], hi = bv.indptr[dc + 1];
    for (int64_t e = lo + (int64_t)blockIdx.y * 256 + threadIdx.x; e < hi; e += (int64_t)gridDim.y * 256) {
        const int idx = bv.indices[e], tile = idx / kTI;
        const int pos = tstart[tile] + atomicAdd(&tcursor[tile], 1);
        eb[pos] = b; en[pos] = idx - tile * kTI; ev[pos] = bv.values[e];
    }
}

// after the fused kernel the cursors hold the counts again: reset them for the next step
__global__ void zero_int_kernel(int* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

}  // namespace aae
