// Fused decoder output layer for reference-sized batches (B <= ~104 at h = 200):
//
//   per tile of 32 items, one persistent 1024-thread workgroup (16 waves, 4 per SIMD) per CU:
//     GEMM1  logits[B x 32]   = dh2[B x (h+1)] * V3a[32 x (h+1)]^T          (dec.lin3, aae.py:176)
//     BCE    G = dL/dlogits, loss                                            (aae.py:177, 693-695)
//     GEMM2  dV3a[32 x (h+1)] = G^T * dh2            -> dec_optim (Adam) on this V3a tile, in place
//     GEMM3  dA2[B x (h+1)]  += G * V3a[32 x (h+1)]   (accumulated in registers across the tiles
//                                                      of the workgroup, one slab per workgroup)
//
// The three GEMMs run on v_mfma_f32_16x16x4_f32 from LDS-resident operands: dh2 is loaded once
// per workgroup (B x 210 floats), the V3a tile once per tile and serves GEMM1, GEMM3 and the
// optimiser (p), so the layer's weights cross HBM exactly once per step: 4 B read + 8 B (m, v)
// read + 12 B written per parameter = the 24 B/param floor of a fused Adam, and dL/dlogits
// [B x N] never exists in HBM.  Next tile's V3a and this tile's m, v are in flight (registers)
// while the matrix cores work.
//
// Non-zero BCE targets come as per-tile entry lists (items sorted into 32-item buckets by
// tile_hist/scan/fill below, a counting sort over the batch's CSR entries).
#pragma once
#include "device_common.h"
#include "gemm_f32.h"
#include "kernels.h"

namespace aae {

constexpr int kTI = 32;        // items per tile
constexpr int kSD = 210;       // LDS row stride of dh2 / V3a images: == 2 (mod 4) and /2 odd, so the 16 rows x 2 k
                               // that one ds_read_b32 half-wave touches fall on 32 distinct banks
constexpr int kSG = 34;        // LDS row stride of the G tile (same property for the b-major reads of GEMM3)
constexpr int kSO = 212;       // LDS row stride of the dV3a tile (16-byte aligned rows for the float4 epilogue)
constexpr int kMB = 7;         // 16-row blocks of the batch dimension (B <= 112)
constexpr int kNT = 1024;      // threads per workgroup: 4 waves per SIMD hide the LDS-operand latency of the
constexpr int kNW = kNT / 64;  // MFMA chains by wave switching (a wave's own chain is load -> wait -> MFMA)

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
};

// LDS bytes the kernel needs for (B, h)
inline size_t dec_fused_lds_bytes(int B, int h) {
    (void)h;
    return sizeof(float) * ((size_t)B * kSD + (size_t)kTI * kSD + (size_t)16 * kMB * kSG + (size_t)kTI * kSO + 64);
}

template <int NB>   // NB = ceil((h + 1) / 16) column blocks
__global__ __launch_bounds__(kNT) void dec_fused_kernel(DecFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dhs = lds;                                  // [B][kSD]
    float* v3s = dhs + (size_t)a.B * kSD;              // [32][kSD]
    float* gs = v3s + kTI * kSD;                       // [112][kSG]  logits, then dL/dlogits
    float* os = gs + 16 * kMB * kSG;                   // [32][kSO]   dV3a tile
    float* red = os + kTI * kSO;                       // [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: block ids below are scalars
    const int fr = lane & 15, fk = lane >> 4;
    const int B = a.B, K1 = a.h + 1, ldv = a.ldv;
    // k-steps rounded up to the unroll factor 4: the extra ones multiply zero padding (columns >= ldh
    // of both LDS images are zeroed once, G rows >= B are zero)
    const int ksteps1 = ((K1 + 3) / 4 + 3) & ~3;       // GEMM1: k over the h+1 hidden columns, <= (kSD-2)/4
    const int ksteps2 = min(((B + 3) / 4 + 1) & ~1, 4 * kMB);   // GEMM2: k over docs (2 per loop trip), rows < 16*kMB
    const int nmb = (B + 15) >> 4;                      // 16-row blocks actually present (<= kMB)
    const int ntiles = (a.N + kTI - 1) / kTI;
    const int f4_per_row = ldv / 4;                    // ldv % 4 == 0
    const int tile_f4 = kTI * f4_per_row;              // float4 per tile span
    constexpr int NV = 2;                              // float4 slots per thread (tile_f4 <= kNT * 2)
    const OptScalars sc = *a.sc;
    const bool do_adam = a.gradV3 == nullptr;

    // ---- once per workgroup: dh2 -> LDS, zero the G tile (rows >= B stay zero)
    for (int f = tid; f < B * (a.ldh / 4); f += kNT) {
        int r = f / (a.ldh / 4), c4 = f % (a.ldh / 4);
        float4 x = *reinterpret_cast<const float4*>(a.dh2 + (size_t)r * a.ldh + c4 * 4);
        float* d = dhs + r * kSD + c4 * 4;
        *reinterpret_cast<float2*>(d) = make_float2(x.x, x.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(x.z, x.w);
    }
    for (int i = tid; i < 16 * kMB * kSG; i += kNT) gs[i] = 0.f;
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
        const size_t f0 = (size_t)tile * kTI * f4_per_row;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t f = f0 + (size_t)(tid + kNT * j);
            r[j] = reinterpret_cast<const float4*>(base)[f < last_f4 ? f : last_f4];
        }
        // no masking here: a use right after the load would make the wave wait for it; clamped lanes
        // hold finite values that are either never stored or multiplied by zero gradients
    };

    float4 vreg[NV], mreg[NV], sreg[NV];
    int tile = blockIdx.x;
    if (tile < ntiles) load_span(a.V3a, tile, vreg);

    for (; tile < ntiles; tile += gridDim.x) {
        const int i0 = tile * kTI;
        __syncthreads();                               // previous tile's readers of v3s / os are done
        // ---- S0: V3a tile registers -> LDS; start this tile's m, v and the next tile's V3a
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + kNT * j;
            if (f < tile_f4) {
                float* d = v3s + (f / f4_per_row) * kSD + (f % f4_per_row) * 4;
                *reinterpret_cast<float2*>(d) = make_float2(vreg[j].x, vreg[j].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(vreg[j].z, vreg[j].w);
            }
        }
        if (tile + (int)gridDim.x < ntiles) load_span(a.V3a, tile + gridDim.x, vreg);
        __syncthreads();

        if (!(a.dbg_skip & 1))
        // ---- S1: GEMM1 logits[b][n] = sum_k dh2[b][k] * V3a[n][k]; blocks (mb, nb2) id = mb*2 + nb2,
        // one block per wave (a wave past the last block re-does it and does not store): branch-free
        {
            f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int id0 = min(wave, 2 * nmb - 1);
            const float* pa0 = dhs + min((id0 >> 1) * 16 + fr, B - 1) * kSD + fk;
            const float* pb0 = v3s + ((id0 & 1) * 16 + fr) * kSD + fk;
            // groups of 4 k-steps: all 8 LDS reads of a group are issued before its 4 MFMAs; two
            // accumulators (even / odd k-steps) break the dependent-accumulator latency
            f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < ksteps1; ks += 4) {
                float x[4], y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { x[j] = pa0[(ks + j) * 4]; y[j] = pb0[(ks + j) * 4]; }
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], y[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1], y[1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2], y[2], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[3], y[3], c1, 0, 0, 0);
            }
            c0 += c1;
            // raw logits -> gs[b][n]   (C map: row = 4*(lane>>4) + r, col = lane & 15)
            if (wave < 2 * nmb) {
                const int rb = (id0 >> 1) * 16 + fk * 4, cb = (id0 & 1) * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rb + r < B) gs[(rb + r) * kSG + cb] = c0[r];
            }
        }
        __syncthreads();

        // ---- S2: BCE.  Entries with a non-zero target first (they need the raw logit); their
        // corrected gradient waits in the (still unused) dV3a tile buffer ...
        const int e0 = a.te.start[tile], e1 = (a.dbg_skip & 2) ? e0 : a.te.start[tile + 1];
        int* fix_pos = reinterpret_cast<int*>(os);
        float* fix_g = os + (kTI * kSO) / 2;
        for (int e = e0 + tid; e < e1; e += kNT) {
            const int p = a.te.eb[e] * kSG + a.te.en[e];
            float g0, l0, g1, l1;
            bce_elem_t0(gs[p], a.gscale, g0, l0);
            bce_elem(gs[p], a.te.ev[e], a.gscale, g1, l1);
            loss += l1 - l0;
            fix_pos[e - e0] = p; fix_g[e - e0] = g1;
        }
        __syncthreads();
        // ... then every element with the zero-target form, in place
        for (int i = tid; i < ((a.dbg_skip & 2) ? 0 : B * kTI); i += kNT) {
            const int b = i >> 5, n = i & 31;
            float g = 0.f, l = 0.f;
            if (i0 + n < a.N) bce_elem_t0(gs[b * kSG + n], a.gscale, g, l);
            gs[b * kSG + n] = g;
            loss += l;
        }
        __syncthreads();
        for (int e = tid; e < e1 - e0; e += kNT) gs[fix_pos[e]] = fix_g[e];
        __syncthreads();

        // the optimiser moments of this tile travel while GEMM2 and GEMM3 run
        if (do_adam && !sc.is_sgd) { load_span(a.M, tile, mreg); load_span(a.V, tile, sreg); }
        if (!(a.dbg_skip & 4))
        // ---- S3: GEMM2 dV3a[item][c] = sum_b G[b][item] * dh2[b][c]; blocks id = nb*2 + ib; every wave
        // runs Q2 blocks (ids w + 8q, clamped), branch-free
        {
            constexpr int Q2 = (2 * NB + kNW - 1) / kNW;
            f32x4 acc2[Q2];
            const float* pg[Q2]; const float* pd[Q2];
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                acc2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int id = min(wave + kNW * q, 2 * NB - 1);
                pg[q] = gs + fk * kSG + (id & 1) * 16 + fr;          // G[b = 4ks + fk][item]
                pd[q] = dhs + (id >> 1) * 16 + fr;                   // dh2[b][col], row added below (clamped)
            }
            for (int ks = 0; ks < ksteps2; ks += 2) {        // ksteps2 is even
                float x[2][Q2], y[2][Q2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int rowc = min((ks + j) * 4 + fk, B - 1) * kSD;
#pragma unroll
                    for (int q = 0; q < Q2; ++q) { x[j][q] = pg[q][(ks + j) * 4 * kSG]; y[j][q] = pd[q][rowc]; }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < Q2; ++q)
                        acc2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][q], y[j][q], acc2[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                const int id = wave + kNW * q;
                if (id < 2 * NB) {
                    const int rb = (id & 1) * 16 + fk * 4, cb = (id >> 1) * 16 + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) os[(rb + r) * kSO + cb] = acc2[q][r];
                }
            }
        }

        if (!(a.dbg_skip & 8))
        // ---- S4: GEMM3 dA2[b][c] += sum_n G[b][n] * V3a[n][c]; blocks id = mb*NB + nb, Q3 per wave
        {
            constexpr int Q3 = (kMB * NB + kNW - 1) / kNW;
            const float* pg[Q3]; const float* pv[Q3];
#pragma unroll
            for (int q = 0; q < Q3; ++q) {
                const int id = min(wave + kNW * q, nmb * NB - 1);
                const int mb = id / NB, nb = id - mb * NB;
                pg[q] = gs + (mb * 16 + fr) * kSG + fk;             // G[b][n = 4ks + fk]
                pv[q] = v3s + fk * kSD + nb * 16 + fr;               // V3a[n = 4ks + fk][col]
            }
#pragma unroll
            for (int ks = 0; ks < kTI / 4; ++ks) {
                float x[Q3], y[Q3];
#pragma unroll
                for (int q = 0; q < Q3; ++q) { x[q] = pg[q][ks * 4]; y[q] = pv[q][ks * 4 * kSD]; }
#pragma unroll
                for (int q = 0; q < Q3; ++q) acc3[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[q], acc3[q], 0, 0, 0);
            }
        }
        __syncthreads();                               // os complete

        // ---- S5: optimiser on the tile (or gradient export), whole rows, float4 per lane
        if (!(a.dbg_skip & 16))
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + kNT * j;
            if (f < tile_f4) {
                const int row = f / f4_per_row, c4 = f % f4_per_row;
                if (i0 + row < a.N) {
                    const float4 g = *reinterpret_cast<const float4*>(os + row * kSO + c4 * 4);
                    const size_t off = ((size_t)i0 + row) * ldv + (size_t)c4 * 4;
                    if (!do_adam) {
                        *reinterpret_cast<float4*>(a.gradV3 + off) = g;
                    } else {
                        const float* ps = v3s + row * kSD + c4 * 4;
                        float4 p = make_float4(ps[0], ps[1], ps[2], ps[3]);
                        float4 mm = mreg[j], vv = sreg[j];
                        if (!(a.dbg_skip & 64)) {
                        adam_update(p.x, mm.x, vv.x, g.x, sc); adam_update(p.y, mm.y, vv.y, g.y, sc);
                        adam_update(p.z, mm.z, vv.z, g.z, sc); adam_update(p.w, mm.w, vv.w, g.w, sc);
                        }
                        if (a.dbg_skip & 32) { if (p.x + mm.x + vv.x == 123.f) a.partials[1] = 1.f; continue; }
                        *reinterpret_cast<float4*>(a.V3a + off) = p;
                        if (!sc.is_sgd) {
                            *reinterpret_cast<float4*>(a.M + off) = mm;
                            *reinterpret_cast<float4*>(a.V + off) = vv;
                        }
                    }
                }
            }
        }
    }

    // ---- dA2 partial of this workgroup -> its slab; loss partial
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int q = 0; q < (kMB * NB + kNW - 1) / kNW; ++q) {
        const int id = wave + kNW * q;
        if (id < nmb * NB) {
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
