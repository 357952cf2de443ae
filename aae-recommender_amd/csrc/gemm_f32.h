// Tiled fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered
// fmaf chain), with the layer-specific work fused into the epilogue.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)        64x64 tile per 256-thread workgroup (4 waves, 2x2),
//                                          each wave a 32x32 sub-tile = 2x2 MFMA blocks.
//   AT = 0: A stored [M][K] (k contiguous)      AT = 1: A stored [K][M] (m contiguous)
//   BT = 0: B stored [K][N] (n contiguous)      BT = 1: B stored [N][K] (k contiguous)
//
// Operands are staged global -> registers -> LDS in k-major images As[BK][64+16], Bs[BK][64+16]
// (the +16 pad puts the two k-rows a ds_read_b32 half-wave touches on disjoint banks), the next
// k-slab's global loads are issued before the current slab's MFMAs.  After the k loop the
// accumulators go through LDS once more so that every epilogue reads and writes global memory
// as whole 256-byte row segments (float4 per lane) instead of the MFMA's 64-byte column
// fragments.  blockIdx.z = split-K slice.
#pragma once
#include <cstdlib>
#include "device_common.h"

namespace aae {

struct GemmShape {
    const float* A; const float* B;
    int M, N, K;
    int lda, ldb;
    int k_per_split;   // multiple of 16; gridDim.z slices
    // XCD-aware tile order of the streaming GEMMs (launch_gemm_mode fills these; gx == 0: plain 3-D grid).  The tiles that
    // share the LARGE operand form a group - the m-tiles of one n-tile (forward: the V3 tile), the n-tiles of one m-tile
    // (weight gradient: the dL/dlogits tile), all tiles of one split-K slice (dX: both) - and a group's members get
    // consecutive slots of ONE XCD (workgroup L runs on XCD L % 8): the operand comes from HBM once and from that XCD's L2
    // for the other members.  With the plain grid order a group's members sit on different XCDs, or thousands of
    // workgroups apart, and the vocabulary-wide operand crossed the HBM bus 4-8 times (r3: 1.3 GB instead of 0.69 per
    // dV3 launch at batch 512).
    int gx = 0, gy = 0, gz = 0;
};

// this workgroup's tile: false = a padding workgroup of the remapped grid
__device__ __forceinline__ bool gemm_tile_of_block(const GemmShape& g, int& bx, int& by, int& bz) {
    if (g.gx == 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return true; }
    const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
    int ngroups, gsz;
    if (g.gz > 1) { ngroups = g.gz; gsz = g.gx * g.gy; }
    else if (g.gx >= g.gy) { ngroups = g.gx; gsz = g.gy; }
    else { ngroups = g.gy; gsz = g.gx; }
    const int grp = (s / gsz) * 8 + xcd, mem = s % gsz;
    if (grp >= ngroups) return false;
    if (g.gz > 1) { bz = grp; bx = mem % g.gx; by = mem / g.gx; }
    else if (g.gx >= g.gy) { bx = grp; by = mem; bz = 0; }
    else { by = grp; bx = mem; bz = 0; }
    return true;
}
inline unsigned gemm_remapped_grid(GemmShape& g, int gx, int gy, int gz) {
    g.gx = gx; g.gy = gy; g.gz = gz;
    const int ngroups = gz > 1 ? gz : (gx >= gy ? gx : gy);
    const int gsz = gz > 1 ? gx * gy : (gx >= gy ? gy : gx);
    return (unsigned)(8 * ((ngroups + 7) / 8) * gsz);
}

// TS = square tile edge per 256-thread workgroup (4 waves as 2x2, each wave (TS/2)^2):
//   64 for the streaming GEMMs over the item vocabulary; 32 for the tiny layer GEMMs, where a
//   64-tile grid is <= 16 workgroups and each wave's MFMA chain is the critical path (4x more
//   workgroups, 4x shorter chains).
// BK = k-depth of one staged slab: 16 for the streaming GEMMs (small LDS footprint, 8 workgroups
//   per CU hide the load latency), 64 for the tiny ones (4x fewer load->barrier round trips).
// BF = true (bf16 mode of the build, BASELINE config C2): the same staging, fp32 LDS images and epilogues, but the
// products run on v_mfma_f32_16x16x32_bf16 - a lane gathers its 8 k-values of a 32-deep step from the fp32 image (the
// same conflict-free ds_read_b32 walk, k = kk + fk + 4 j) and rounds them to bf16 (v_cvt_pk_bf16_f32, ties to even) on
// the way into the matrix core; accumulation stays fp32.  One MFMA then does the work of eight.
typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gemm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gemm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gemm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned gemm_pack_bf16(float a, float b) {
    gemm_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, gemm_bf16x2));
}

template <int AT, int BT, int BK, int TS, class Epi, bool BF = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmShape g, Epi epi) {
    static_assert(!BF || BK % 32 == 0, "bf16 products take 32-deep k-steps");
    constexpr int LDT = TS + 16;                    // operand image row stride (floats): the two k-rows a
                                                    // ds_read_b32 half-wave touches land on disjoint banks
    constexpr int LDC = TS + 4;                     // accumulator image row stride
    constexpr int NV = TS * BK / 1024;              // float4 per thread and operand per slab
    constexpr int MI = TS / 32;                     // 16x16 MFMA blocks per wave and dimension
    constexpr int kSmem = (2 * BK * LDT > TS * LDC) ? 2 * BK * LDT : TS * LDC;
    static_assert(NV >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float smem[kSmem];
    float* As = smem;
    float* Bs = smem + BK * LDT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (TS / 2), wn = (wave & 1) * (TS / 2);
    int bx, by, bz;
    if (!gemm_tile_of_block(g, bx, by, bz)) return;
    const int m0 = by * TS, n0 = bx * TS;
    const int kbeg = bz * g.k_per_split;
    const int ntx = g.gx ? g.gx : (int)gridDim.x, nty = g.gx ? g.gy : (int)gridDim.y;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x4 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[NV], rb[NV];
    // k-contiguous operand (tile rows = m or n): float4 index f -> row f / (BK/4), k = 4 * (f % (BK/4))
    // m/n-contiguous operand:                    float4 index f -> k = f / (TS/4), 4 columns at 4 * (f % (TS/4))
    // NOTE: every load is issued unconditionally from a clamped (always valid) address, and the
    // out-of-range lanes are zeroed only when the registers are written to LDS one slab later - a
    // load under a lane-dependent `if`, or any use right behind it, makes the wave wait for it
    // (vmcnt(0)) and serialises the slab's memory round trips with its MFMAs.
    auto load_kc = [&](const float* P, int ld, int r0, int rmax, int k0, float4* r) {
        const int kmax4 = ((kend + 3) & ~3) - 4;          // last float4 that starts inside the padded row
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int row = min(r0 + f / (BK / 4), rmax - 1), k = min(k0 + (f % (BK / 4)) * 4, kmax4);
            r[j] = *reinterpret_cast<const float4*>(P + (size_t)row * ld + k);
        }
    };
    auto load_mc = [&](const float* P, int ld, int c0, int cmax, int k0, float4* r) {
        const int cmax4 = ((cmax + 3) & ~3) - 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int k = min(k0 + f / (TS / 4), kend - 1), cidx = min(c0 + (f % (TS / 4)) * 4, cmax4);
            r[j] = *reinterpret_cast<const float4*>(P + (size_t)k * ld + cidx);
        }
    };
    auto store_kc = [&](float* T, const float4* r, int r0, int rmax, int k0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
            const bool ok = r0 + row < rmax;
            const int k = k0 + kq;
            T[(kq + 0) * LDT + row] = (ok && k < kend) ? r[j].x : 0.f;
            T[(kq + 1) * LDT + row] = (ok && k + 1 < kend) ? r[j].y : 0.f;
            T[(kq + 2) * LDT + row] = (ok && k + 2 < kend) ? r[j].z : 0.f;
            T[(kq + 3) * LDT + row] = (ok && k + 3 < kend) ? r[j].w : 0.f;
        }
    };
    auto store_mc = [&](float* T, const float4* r, int c0, int cmax, int k0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int kr = f / (TS / 4), cq = (f % (TS / 4)) * 4;
            const bool ok = k0 + kr < kend;
            const int cidx = c0 + cq;
            float4 v;
            v.x = (ok && cidx < cmax) ? r[j].x : 0.f;
            v.y = (ok && cidx + 1 < cmax) ? r[j].y : 0.f;
            v.z = (ok && cidx + 2 < cmax) ? r[j].z : 0.f;
            v.w = (ok && cidx + 3 < cmax) ? r[j].w : 0.f;
            *reinterpret_cast<float4*>(&T[kr * LDT + cq]) = v;
        }
    };
    auto load_tiles = [&](int k0) {
        if (AT == 0) load_kc(g.A, g.lda, m0, g.M, k0, ra); else load_mc(g.A, g.lda, m0, g.M, k0, ra);
        if (BT == 1) load_kc(g.B, g.ldb, n0, g.N, k0, rb); else load_mc(g.B, g.ldb, n0, g.N, k0, rb);
    };

    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            if (AT == 0) store_kc(As, ra, m0, g.M, k0); else store_mc(As, ra, m0, g.M, k0);
            if (BT == 1) store_kc(Bs, rb, n0, g.N, k0); else store_mc(Bs, rb, n0, g.N, k0);
            __syncthreads();
            if (k0 + BK < kend) load_tiles(k0 + BK);
            const int fr = lane & 15, fk = lane >> 4;
            if constexpr (BF) {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 32) {
                    gemm_bf16x8 a[MI], b[MI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        float x[8], y[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            x[e] = As[(kk + fk + 4 * e) * LDT + wm + i * 16 + fr];
                            y[e] = Bs[(kk + fk + 4 * e) * LDT + wn + i * 16 + fr];
                        }
                        const gemm_u32x4 xa = {gemm_pack_bf16(x[0], x[1]), gemm_pack_bf16(x[2], x[3]), gemm_pack_bf16(x[4], x[5]), gemm_pack_bf16(x[6], x[7])};
                        const gemm_u32x4 yb = {gemm_pack_bf16(y[0], y[1]), gemm_pack_bf16(y[2], y[3]), gemm_pack_bf16(y[4], y[5]), gemm_pack_bf16(y[6], y[7])};
                        a[i] = __builtin_bit_cast(gemm_bf16x8, xa);
                        b[i] = __builtin_bit_cast(gemm_bf16x8, yb);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < MI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                float a[MI], b[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    a[i] = As[(kk + fk) * LDT + wm + i * 16 + fr];
                    b[i] = Bs[(kk + fk) * LDT + wn + i * 16 + fr];
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            }
            __syncthreads();
        }
    }

    // accumulators -> LDS image (C/D map of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg)
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm + i * 16 + (lane >> 4) * 4 + r) * LDC + wn + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();

    typename Epi::State st;
    epi.begin(st);
    constexpr int RPP = 1024 / TS;                  // tile rows covered per pass of 256 float4
#pragma unroll
    for (int p = 0; p < TS / RPP; ++p) {
        int row = p * RPP + tid / (TS / 4), col = (tid % (TS / 4)) * 4;
        int gm = m0 + row, gn = n0 + col;
        if (gm < g.M && gn < g.N) {
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + col]);
            epi.apply(st, gm, gn, g.N, v, bz);
        }
    }
    __syncthreads();
    epi.finish(st, smem, bx + ntx * (by + nty * bz));
}

// -----------------------------------------------------------------------------------------
// epilogues.  apply() gets 4 consecutive columns gn..gn+3 of row gm; columns >= N must not be
// written (they are zero padding or the next field of the destination).
// -----------------------------------------------------------------------------------------
struct EpiNoState { struct State {}; __device__ void begin(State&) const {}
                    __device__ void finish(State&, float*, int) const {} };

// out = v
struct EpiStore : EpiNoState {
    float* out; int ld;
    __device__ void apply(State&, int gm, int gn, int N, float4 v, int) const {
        float* o = out + (size_t)gm * ld + gn;
        if (gn + 3 < N) { *reinterpret_cast<float4*>(o) = v; return; }
        for (int i = 0; i < 4 && gn + i < N; ++i) o[i] = (&v.x)[i];
    }
};

// out[z][m][n] = v   (split-K partial slabs, reduced by the consumer)
struct EpiSlab : EpiNoState {
    float* out; int ld; size_t slab_stride;
    __device__ void apply(State&, int gm, int gn, int N, float4 v, int z) const {
        float* o = out + (size_t)z * slab_stride + (size_t)gm * ld + gn;
        if (gn + 3 < N) { *reinterpret_cast<float4*>(o) = v; return; }
        for (int i = 0; i < 4 && gn + i < N; ++i) o[i] = (&v.x)[i];
    }
};

// y = act(dropout(v))    (Encoder/Decoder/Discriminator hidden layers, aae.py:135-141)
struct EpiDropAct {
    struct State { uint64_t key; };
    float* out; int ld; int act; DropSpec d; uint64_t seed; const long long* step_ctr;
    __device__ void begin(State& st) const { st.key = d.device_rng ? rng_key(seed, (uint64_t)*step_ctr, 0) : 0; }
    __device__ void finish(State&, float*, int) const {}
    __device__ void apply(State& st, int gm, int gn, int N, float4 v, int) const {
        float* o = out + (size_t)gm * ld + gn;
        for (int i = 0; i < 4 && gn + i < N; ++i) {
            float a = (&v.x)[i];
            if (d.enabled) a = drop_fwd(d, drop_keep(d, st.key, gm, gn + i), a);
            o[i] = act_fwd(act, a);
        }
    }
};

// y = sigmoid(v)      (decoder output in predict, discriminator output)
struct EpiSigmoid : EpiNoState {
    float* out; int ld;
    __device__ void apply(State&, int gm, int gn, int N, float4 v, int) const {
        float* o = out + (size_t)gm * ld + gn;
        float4 y = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
        if (gn + 3 < N) { *reinterpret_cast<float4*>(o) = y; return; }
        for (int i = 0; i < 4 && gn + i < N; ++i) o[i] = (&y.x)[i];
    }
};

// g_pre = v * act'(y) * dropout_scale     (back through activation and dropout of the layer
// whose OUTPUT y is; v is dL/dy)
struct EpiActBwd {
    struct State { uint64_t key; };
    float* out; int ld; const float* y; int ldy; int act; DropSpec d; uint64_t seed; const long long* step_ctr;
    __device__ void begin(State& st) const { st.key = d.device_rng ? rng_key(seed, (uint64_t)*step_ctr, 0) : 0; }
    __device__ void finish(State&, float*, int) const {}
    __device__ void apply(State& st, int gm, int gn, int N, float4 v, int) const {
        float* o = out + (size_t)gm * ld + gn;
        const float* yy = y + (size_t)gm * ldy + gn;
        for (int i = 0; i < 4 && gn + i < N; ++i) {
            float gdy = (&v.x)[i] * act_grad_from_y<true>(act, yy[i]);
            if (d.enabled) gdy *= drop_bwd_mul(d, drop_keep(d, st.key, gm, gn + i));
            o[i] = gdy;
        }
    }
};

// Decoder output + F.binary_cross_entropy(x_hat + TINY, target + TINY) with target = 0
// everywhere (aae.py:176-177, 693-695); the few non-zero targets are patched afterwards by
// bce_fixup_kernel.  Writes dL/dlogit and a per-workgroup loss partial.
//   loss_e = -(t*max(log x,-100) + (1-t)*max(log1p(-x),-100)),  x = sigmoid(l) + TINY, t = TINY
//   dL/dl  = (x - t) / max((1-x)*x, 1e-12) * s*(1-s) * gscale,  gscale = grad_scale/(B*N)
__device__ __forceinline__ void bce_elem(float logit, float t_raw, float gscale, float& g, float& loss) {
    // sigmoid, log(x) and softplus from one exp; hardware rcp/log (1 ulp) instead of libm calls
    const float e = __expf(-fabsf(logit));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float s = logit >= 0.f ? r : e * r;
    const float x = s + kTiny, t = t_raw + kTiny;
    const float lx = fmaxf(__logf(x), -100.f);
    const float lp = e < 0.01f ? e * (1.f - e * (0.5f - e * 0.33333334f)) : __logf(1.f + e);
    // max(log1p(-x), -100) = -softplus(logit) until sigmoid rounds to 1.0f (logit > 17.33), then -100
    const float l1x = logit > 17.32868f ? -100.f : -(fmaxf(logit, 0.f) + lp);
    loss = -(t * lx + (1.f - t) * l1x);
    g = (x - t) * __builtin_amdgcn_rcpf(fmaxf((1.f - x) * x, 1e-12f)) * (s * (1.f - s)) * gscale;
}

// The same for target == 0 (every element the GEMM epilogue sees), simplified analytically:
//   x - t = s and (1-x)x = s(1-s) up to 1e-12, so dL/dl = s * gscale;
//   loss = -log1p(-x) = softplus(l) = max(l,0) + log1p(exp(-|l|))  (the t*log(x) term is <= 1e-10).
// fp32 saturation of the reference is kept: for l > 17.33 sigmoid rounds to exactly 1.0f, the
// reference's log1p(-1) = -inf is clamped to -100 and its s(1-s) factor zeroes the gradient.
// log(x) for 1 <= x <= 2 with the bits of __logf(x): OCML's log_f32 is v_log_f32 (log2) times ln 2 in two-term
// arithmetic, wrapped in a scaling of denormal arguments and a test for infinities - neither can occur here, and the
// wrapper is what made `e < 0.01f ? series : __logf(1 + e)` a BRANCH with 13 instructions + 2 s_nop on its far side,
// taken by every wave (r6: the zero-target form below is 2 x this per cell pair, 1.8 k pairs per tile of the output layer).
__device__ __forceinline__ float log_1_to_2(float x) {
    const float y = __builtin_amdgcn_logf(x);
    const float c = 0x1.62e42ep-1f, cc = 0x1.efa39ep-25f;      // ln 2 = c + cc (0x3f317217, 0x3377d1cf)
    const float t = y * c;
    return __builtin_fmaf(y, c, __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -t)));
}

__device__ __forceinline__ void bce_elem_t0(float l, float gscale, float& g, float& loss) {
    float e = __expf(-fabsf(l));
    float r = __builtin_amdgcn_rcpf(1.f + e);
    float s = l >= 0.f ? r : e * r;
    const float lp_series = e * (1.f - e * (0.5f - e * 0.33333334f)), lp_log = log_1_to_2(1.f + e);
    float lp = e < 0.01f ? lp_series : lp_log;
    g = s * gscale;
    loss = fmaxf(l, 0.f) + lp;
    if (l > 17.32868f) { g = 0.f; loss = 100.f; }
}

// The zero-target form in PARTS, for an epilogue that only needs the SUM of its cells' losses (dec_crit_x3.h): loss = add +
// log(fac) with fac = 1 + e (in [1, 2]) where bce_elem_t0 takes the logarithm and 1 where it takes the series - the caller
// multiplies the factors of several cells and takes ONE logarithm (v_log_f32 is a quarter-rate instruction, and with it go the
// four multiply-adds of the ln 2 product): log(prod (1 + e_i)) for sum log(1 + e_i), the rounding of the few products (< 2^-22
// relative on a value in [1, 16]) below what the rounding of 1 + e already is.  g: the same bits as bce_elem_t0's.
__device__ __forceinline__ void bce_elem_t0_parts(float l, float gscale, float& g, float& add, float& fac) {
    const float e = __expf(-fabsf(l));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float s = l >= 0.f ? r : e * r;
    const float lp_series = e * (1.f - e * (0.5f - e * 0.33333334f));
    const bool small = e < 0.01f;
    g = s * gscale;
    add = fmaxf(l, 0.f) + (small ? lp_series : 0.f);
    fac = small ? 1.f : 1.f + e;
    if (l > 17.32868f) { g = 0.f; add = 100.f; fac = 1.f; }
}
// log(x) for a positive normal x, the arithmetic of log_1_to_2 (v_log_f32 times ln 2 in two terms)
__device__ __forceinline__ float log_pos(float x) { return log_1_to_2(x); }

struct EpiBce {
    struct State { float loss; };
    float* G; int ldg; float gscale; float* partials;
    __device__ void begin(State& st) const { st.loss = 0.f; }
    __device__ void apply(State& st, int gm, int gn, int N, float4 v, int) const {
        float* o = G + (size_t)gm * ldg + gn;
        float4 gv;
        for (int i = 0; i < 4; ++i) {
            float gg = 0.f, ll = 0.f;
            if (gn + i < N) bce_elem_t0((&v.x)[i], gscale, gg, ll);
            (&gv.x)[i] = gg; st.loss += ll;
        }
        if (gn + 3 < N) { *reinterpret_cast<float4*>(o) = gv; return; }
        for (int i = 0; i < 4 && gn + i < N; ++i) o[i] = (&gv.x)[i];
    }
    __device__ void finish(State& st, float* smem, int block_linear) const {
        float s = wave_sum(st.loss);
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partials[block_linear] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    }
};

// weight-gradient tile -> fused torch.optim.Adam / SGD update of the same tile (K10 of
// SURVEY 2.1), or plain store of the gradient (data-parallel export mode).
struct EpiAdam : EpiNoState {
    float* p; float* m; float* v; int ld; const OptScalars* sc;
    // optional k4-interleaved copies of the weight block kept in step with p (chain4.h's layer ops read them, device_common.h)
    W4Copies w4 = {nullptr, nullptr, 0, 0};
    __device__ void apply(State&, int gm, int gn, int N, float4 g, int) const {
        OptScalars s = *sc;
        size_t off = (size_t)gm * ld + gn;
        if (gn + 3 < N) {
            float4 pp = *reinterpret_cast<float4*>(p + off);
            float4 mm = make_float4(0, 0, 0, 0), vv = mm;
            if (!s.is_sgd) { mm = *reinterpret_cast<float4*>(m + off); vv = *reinterpret_cast<float4*>(v + off); }
            adam_update(pp.x, mm.x, vv.x, g.x, s); adam_update(pp.y, mm.y, vv.y, g.y, s);
            adam_update(pp.z, mm.z, vv.z, g.z, s); adam_update(pp.w, mm.w, vv.w, g.w, s);
            *reinterpret_cast<float4*>(p + off) = pp;
            if (!s.is_sgd) { *reinterpret_cast<float4*>(m + off) = mm; *reinterpret_cast<float4*>(v + off) = vv; }
            if (w4.f4) w4_put4(w4, gm, gn, pp);
            return;
        }
        for (int i = 0; i < 4 && gn + i < N; ++i) {
            float pp = p[off + i], mm = s.is_sgd ? 0.f : m[off + i], vv = s.is_sgd ? 0.f : v[off + i];
            adam_update(pp, mm, vv, (&g.x)[i], s);
            p[off + i] = pp;
            if (!s.is_sgd) { m[off + i] = mm; v[off + i] = vv; }
            if (w4.f4) w4_put1(w4, gm, gn + i, pp);
        }
    }
};

// -----------------------------------------------------------------------------------------------------------------
// The streaming GEMMs with their fp32 products EMULATED on the bf16 matrix cores (r3; the idea and its error analysis:
// dec_crit_x3.h).  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate, so every operand element is split ONCE, on its way
// from the staging registers into LDS, into three bf16 terms x = x1 + x2 + x3 (exact), the LDS images hold the terms, and
// a 16x16x32 product is the six leading cross terms on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 2.7x the fp32
// matrix rate at fp32-level error, dtype stays f32.  Same tiling (64x64 per 256-thread workgroup, 2x2 waves x 2x2 blocks),
// same split-K, same epilogues as gemm_f32_kernel; slabs are 32 deep.
//   k-contiguous operand (AT = 0 / BT = 1): image [row][32 k] per term, a fragment = one 16-byte read (8 consecutive k);
//   m/n-contiguous operand (AT = 1 / BT = 0): image [32 k][64 rows] per term as it arrives (no 2-byte scatter), a
//       fragment = the TRANSPOSE of 2 x (4 k x 16 rows): ds_read_b64_tr_b16.
// -----------------------------------------------------------------------------------------------------------------
typedef short gemm_s16x4 __attribute__((ext_vector_type(4)));
typedef short gemm_s16x8 __attribute__((ext_vector_type(8)));
// (a, b) -> three packed bf16 pairs with a = a1 + a2 + a3, b = b1 + b2 + b3 exactly (low half = a)
__device__ __forceinline__ void x3_split_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = gemm_pack_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = gemm_pack_bf16(ra, rb);
    p3 = gemm_pack_bf16(ra - __uint_as_float(p2 << 16), rb - __uint_as_float(p2 & 0xFFFF0000u));
}
// the six leading cross terms of (a1 + a2 + a3) (b1 + b2 + b3), smallest first
__device__ __forceinline__ f32x4 x3_mfma(const gemm_bf16x8 (&a)[3], const gemm_bf16x8 (&b)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

template <int AT, int BT, class Epi>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmShape g, Epi epi) {
    constexpr int TS = 64, BK = 32;
    constexpr int SK = 24;                          // dwords per row of a [row][32 k] image: 64 bytes + 32 (a stride of 32 mod 64
                                                    // bytes keeps the 16-byte fragment reads conflict-free, dec_crit_x3.h)
    constexpr int SM = 36;                          // dwords per row of a [k][64 rows] image: 128 bytes + 16
    constexpr int IA = AT == 0 ? TS * SK : BK * SM, IB = BT == 1 ? TS * SK : BK * SM;      // dwords per term image
    constexpr int LDC = TS + 4;
    constexpr int NV = TS * BK / 1024;              // float4 per thread and operand per slab (2)
    constexpr int kSmem = (3 * (IA + IB) > TS * LDC) ? 3 * (IA + IB) : TS * LDC;
    __shared__ __attribute__((aligned(16))) unsigned smem[kSmem];
    unsigned* As = smem;
    unsigned* Bs = smem + 3 * IA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (TS / 2), wn = (wave & 1) * (TS / 2);
    int bx, by, bz;
    if (!gemm_tile_of_block(g, bx, by, bz)) return;
    const int m0 = by * TS, n0 = bx * TS;
    const int kbeg = bz * g.k_per_split;
    const int ntx = g.gx ? g.gx : (int)gridDim.x, nty = g.gx ? g.gy : (int)gridDim.y;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[NV], rb[NV];
    // (as gemm_f32_kernel: every load unconditional from a clamped address, masked when written to LDS one slab later)
    auto load_kc = [&](const float* P, int ld, int r0, int rmax, int k0, float4* r) {
        const int kmax4 = ((kend + 3) & ~3) - 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int row = min(r0 + f / (BK / 4), rmax - 1), k = min(k0 + (f % (BK / 4)) * 4, kmax4);
            r[j] = *reinterpret_cast<const float4*>(P + (size_t)row * ld + k);
        }
    };
    auto load_mc = [&](const float* P, int ld, int c0, int cmax, int k0, float4* r) {
        const int cmax4 = ((cmax + 3) & ~3) - 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int k = min(k0 + f / (TS / 4), kend - 1), cidx = min(c0 + (f % (TS / 4)) * 4, cmax4);
            r[j] = *reinterpret_cast<const float4*>(P + (size_t)k * ld + cidx);
        }
    };
    auto put3 = [&](unsigned* T, int I, int off, float4 v) {       // 4 consecutive elements -> 8 bytes of each term image
        unsigned q0[3], q1[3];
        x3_split_pair(v.x, v.y, q0[0], q0[1], q0[2]);
        x3_split_pair(v.z, v.w, q1[0], q1[1], q1[2]);
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<uint2*>(T + t * I + off) = make_uint2(q0[t], q1[t]);
    };
    auto store_kc = [&](unsigned* T, int I, const float4* r, int r0, int rmax, int k0) {      // image [row][k]
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
            const bool ok = r0 + row < rmax;
            const int k = k0 + kq;
            float4 v;
            v.x = (ok && k < kend) ? r[j].x : 0.f; v.y = (ok && k + 1 < kend) ? r[j].y : 0.f;
            v.z = (ok && k + 2 < kend) ? r[j].z : 0.f; v.w = (ok && k + 3 < kend) ? r[j].w : 0.f;
            put3(T, I, row * SK + (kq >> 1), v);
        }
    };
    auto store_mc = [&](unsigned* T, int I, const float4* r, int c0, int cmax, int k0) {      // image [k][row]
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = tid + 256 * j;
            const int kr = f / (TS / 4), cq = (f % (TS / 4)) * 4;
            const bool ok = k0 + kr < kend;
            const int cidx = c0 + cq;
            float4 v;
            v.x = (ok && cidx < cmax) ? r[j].x : 0.f; v.y = (ok && cidx + 1 < cmax) ? r[j].y : 0.f;
            v.z = (ok && cidx + 2 < cmax) ? r[j].z : 0.f; v.w = (ok && cidx + 3 < cmax) ? r[j].w : 0.f;
            put3(T, I, kr * SM + (cq >> 1), v);
        }
    };
    auto load_tiles = [&](int k0) {
        if (AT == 0) load_kc(g.A, g.lda, m0, g.M, k0, ra); else load_mc(g.A, g.lda, m0, g.M, k0, ra);
        if (BT == 1) load_kc(g.B, g.ldb, n0, g.N, k0, rb); else load_mc(g.B, g.ldb, n0, g.N, k0, rb);
    };
    // fragment of tile rows [r0, r0 + 16): lane (fr, fk) <- row r0 + fr, k = 8 fk + {0..7}
    const int fr = lane & 15, fk = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    auto frag_kc = [&](const unsigned* T, int r0) {
        return __builtin_bit_cast(gemm_bf16x8, *reinterpret_cast<const gemm_u32x4*>(T + (r0 + fr) * SK + 4 * fk));
    };
    auto frag_mc = [&](const unsigned* T, int r0) {
        const unsigned* pb = T + (8 * fk + tq) * SM + (r0 >> 1) + 2 * tp;
        const gemm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gemm_s16x4 __attribute__((address_space(3)))*)(pb));
        const gemm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gemm_s16x4 __attribute__((address_space(3)))*)(pb + 4 * SM));
        const gemm_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(gemm_bf16x8, v);
    };

    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            if (AT == 0) store_kc(As, IA, ra, m0, g.M, k0); else store_mc(As, IA, ra, m0, g.M, k0);
            if (BT == 1) store_kc(Bs, IB, rb, n0, g.N, k0); else store_mc(Bs, IB, rb, n0, g.N, k0);
            __syncthreads();
            if (k0 + BK < kend) load_tiles(k0 + BK);
            gemm_bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    a[i][t] = AT == 0 ? frag_kc(As + t * IA, wm + 16 * i) : frag_mc(As + t * IA, wm + 16 * i);
                    b[i][t] = BT == 1 ? frag_kc(Bs + t * IB, wn + 16 * i) : frag_mc(Bs + t * IB, wn + 16 * i);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = x3_mfma(a[i], b[j], acc[i][j]);
            __syncthreads();
        }
    }

    // accumulators -> LDS image -> epilogue, exactly as gemm_f32_kernel
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm + i * 16 + (lane >> 4) * 4 + r) * LDC + wn + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    typename Epi::State st;
    epi.begin(st);
    constexpr int RPP = 1024 / TS;
#pragma unroll
    for (int p = 0; p < TS / RPP; ++p) {
        int row = p * RPP + tid / (TS / 4), col = (tid % (TS / 4)) * 4;
        int gm = m0 + row, gn = n0 + col;
        if (gm < g.M && gn < g.N) {
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * LDC + col]);
            epi.apply(st, gm, gn, g.N, v, bz);
        }
    }
    __syncthreads();
    epi.finish(st, reinterpret_cast<float*>(smem), bx + ntx * (by + nty * bz));
}

template <int AT, int BT, int BK, int TS, class Epi, bool BF = false>
inline hipError_t launch_gemm(const GemmShape& g, const Epi& epi, int splits, hipStream_t s, bool remap = false) {
    GemmShape gg = g;
    dim3 grid((g.N + TS - 1) / TS, (g.M + TS - 1) / TS, splits);
    if (remap) grid = dim3(gemm_remapped_grid(gg, (int)grid.x, (int)grid.y, (int)grid.z));
    hipLaunchKernelGGL((gemm_f32_kernel<AT, BT, BK, TS, Epi, BF>), grid, dim3(256), 0, s, gg, epi);
    return hipGetLastError();
}
// the streaming (vocabulary-wide) / small-layer variants in either arithmetic: bf16 = true takes 32-deep slabs
// mode: 0 = fp32 matrix pipe, 1 = bf16 inputs (config C2), 2 = fp32 emulated on the bf16 matrix cores (gemm_x3_kernel; the
// streaming GEMMs only - the small layers are latency-, not matrix-pipe-, bound)
enum { kGemmF32 = 0, kGemmBf16 = 1, kGemmX3 = 2 };
template <int AT, int BT, bool BIG, class Epi>
inline hipError_t launch_gemm_mode(int mode, const GemmShape& g, const Epi& epi, int splits, hipStream_t s) {
    const bool bf16 = mode == kGemmBf16;
    constexpr bool remap = true;
    if (BIG && mode == kGemmX3) {
        GemmShape gg = g;
        dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, splits);
        if (remap) grid = dim3(gemm_remapped_grid(gg, (int)grid.x, (int)grid.y, (int)grid.z));
        hipLaunchKernelGGL((gemm_x3_kernel<AT, BT, Epi>), grid, dim3(256), 0, s, gg, epi);
        return hipGetLastError();
    }
    if (BIG) return bf16 ? launch_gemm<AT, BT, 32, 64, Epi, true>(g, epi, splits, s, remap) : launch_gemm<AT, BT, 16, 64, Epi>(g, epi, splits, s, remap);
    return bf16 ? launch_gemm<AT, BT, 64, 32, Epi, true>(g, epi, splits, s) : launch_gemm<AT, BT, 64, 32, Epi>(g, epi, splits, s);
}

}  // namespace aae
