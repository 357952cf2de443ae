// Tiled fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered
// fmaf chain), with the layer-specific work fused into the epilogue.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)        64x64 tile per 256-thread workgroup (4 waves, 2x2),
//                                          each wave a 32x32 sub-tile = 2x2 MFMA blocks.
//   AT = 0: A stored [M][K] (k contiguous)      AT = 1: A stored [K][M] (m contiguous)
//   BT = 0: B stored [K][N] (n contiguous)      BT = 1: B stored [N][K] (k contiguous)
//
// Operands are staged global -> registers -> LDS in k-major images As[16][64+16], Bs[16][64+16]
// (the +16 pad puts the two k-rows a ds_read_b32 half-wave touches on disjoint banks), the next
// k-slab's global loads are issued before the current slab's MFMAs.  After the k loop the
// accumulators go through LDS once more so that every epilogue reads and writes global memory
// as whole 256-byte row segments (float4 per lane) instead of the MFMA's 64-byte column
// fragments.  blockIdx.z = split-K slice.
#pragma once
#include "device_common.h"

namespace aae {

struct GemmShape {
    const float* A; const float* B;
    int M, N, K;
    int lda, ldb;
    int k_per_split;   // multiple of 16; gridDim.z slices
};

constexpr int kTile = 64;
constexpr int kBK = 16;
constexpr int kLdT = kTile + 16;   // operand image row stride (floats)
constexpr int kLdC = kTile + 4;    // accumulator image row stride (floats)

template <int AT, int BT, class Epi>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmShape g, Epi epi) {
    __shared__ __attribute__((aligned(16))) float smem[kTile * kLdC];
    float* As = smem;
    float* Bs = smem + kBK * kLdT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int m0 = blockIdx.y * kTile, n0 = blockIdx.x * kTile;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra, rb;
    auto load_a = [&](int k0) {
        ra = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AT == 0) {
            int m = m0 + (tid >> 2), k = k0 + (tid & 3) * 4;
            if (m < g.M && k < kend) {
                ra = *reinterpret_cast<const float4*>(g.A + (size_t)m * g.lda + k);
                if (k + 1 >= kend) ra.y = 0.f;
                if (k + 2 >= kend) ra.z = 0.f;
                if (k + 3 >= kend) ra.w = 0.f;
            }
        } else {
            int k = k0 + (tid >> 4), m = m0 + (tid & 15) * 4;
            if (k < kend && m < g.M) {
                ra = *reinterpret_cast<const float4*>(g.A + (size_t)k * g.lda + m);
                if (m + 1 >= g.M) ra.y = 0.f;
                if (m + 2 >= g.M) ra.z = 0.f;
                if (m + 3 >= g.M) ra.w = 0.f;
            }
        }
    };
    auto load_b = [&](int k0) {
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BT == 1) {
            int n = n0 + (tid >> 2), k = k0 + (tid & 3) * 4;
            if (n < g.N && k < kend) {
                rb = *reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + k);
                if (k + 1 >= kend) rb.y = 0.f;
                if (k + 2 >= kend) rb.z = 0.f;
                if (k + 3 >= kend) rb.w = 0.f;
            }
        } else {
            int k = k0 + (tid >> 4), n = n0 + (tid & 15) * 4;
            if (k < kend && n < g.N) {
                rb = *reinterpret_cast<const float4*>(g.B + (size_t)k * g.ldb + n);
                if (n + 1 >= g.N) rb.y = 0.f;
                if (n + 2 >= g.N) rb.z = 0.f;
                if (n + 3 >= g.N) rb.w = 0.f;
            }
        }
    };
    auto store_tiles = [&]() {
        if (AT == 0) {
            int r = tid >> 2, kq = (tid & 3) * 4;
            As[(kq + 0) * kLdT + r] = ra.x; As[(kq + 1) * kLdT + r] = ra.y;
            As[(kq + 2) * kLdT + r] = ra.z; As[(kq + 3) * kLdT + r] = ra.w;
        } else {
            *reinterpret_cast<float4*>(&As[(tid >> 4) * kLdT + (tid & 15) * 4]) = ra;
        }
        if (BT == 1) {
            int r = tid >> 2, kq = (tid & 3) * 4;
            Bs[(kq + 0) * kLdT + r] = rb.x; Bs[(kq + 1) * kLdT + r] = rb.y;
            Bs[(kq + 2) * kLdT + r] = rb.z; Bs[(kq + 3) * kLdT + r] = rb.w;
        } else {
            *reinterpret_cast<float4*>(&Bs[(tid >> 4) * kLdT + (tid & 15) * 4]) = rb;
        }
    };

    if (kbeg < kend) {
        load_a(kbeg); load_b(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += kBK) {
            store_tiles();
            __syncthreads();
            if (k0 + kBK < kend) { load_a(k0 + kBK); load_b(k0 + kBK); }
            const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
            for (int kk = 0; kk < kBK; kk += 4) {
                float a0 = As[(kk + fk) * kLdT + wm + fr];
                float a1 = As[(kk + fk) * kLdT + wm + 16 + fr];
                float b0 = Bs[(kk + fk) * kLdT + wn + fr];
                float b1 = Bs[(kk + fk) * kLdT + wn + 16 + fr];
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // accumulators -> LDS image (C/D map of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg)
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm + i * 16 + (lane >> 4) * 4 + r) * kLdC + wn + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();

    typename Epi::State st;
    epi.begin(st);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int row = p * 16 + (tid >> 4), col = (tid & 15) * 4;
        int gm = m0 + row, gn = n0 + col;
        if (gm < g.M && gn < g.N) {
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * kLdC + col]);
            epi.apply(st, gm, gn, g.N, v, (int)blockIdx.z);
        }
    }
    __syncthreads();
    epi.finish(st, smem, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)));
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
            float gdy = (&v.x)[i] * act_grad_from_y(act, yy[i]);
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
    float s = sigmoidf_(logit);
    float x = s + kTiny, t = t_raw + kTiny;
    float lx = fmaxf(__logf(x), -100.f);
    float l1x = fmaxf(log1pf(-x), -100.f);
    loss = -(t * lx + (1.f - t) * l1x);
    g = (x - t) / fmaxf((1.f - x) * x, 1e-12f) * (s * (1.f - s)) * gscale;
}

struct EpiBce {
    struct State { float loss; };
    float* G; int ldg; float gscale; float* partials;
    __device__ void begin(State& st) const { st.loss = 0.f; }
    __device__ void apply(State& st, int gm, int gn, int N, float4 v, int) const {
        float* o = G + (size_t)gm * ldg + gn;
        float4 gv;
        for (int i = 0; i < 4; ++i) {
            float gg = 0.f, ll = 0.f;
            if (gn + i < N) bce_elem((&v.x)[i], 0.f, gscale, gg, ll);
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
            return;
        }
        for (int i = 0; i < 4 && gn + i < N; ++i) {
            float pp = p[off + i], mm = s.is_sgd ? 0.f : m[off + i], vv = s.is_sgd ? 0.f : v[off + i];
            adam_update(pp, mm, vv, (&g.x)[i], s);
            p[off + i] = pp;
            if (!s.is_sgd) { m[off + i] = mm; v[off + i] = vv; }
        }
    }
};

template <int AT, int BT, class Epi>
inline hipError_t launch_gemm(const GemmShape& g, const Epi& epi, int splits, hipStream_t s) {
    dim3 grid((g.N + kTile - 1) / kTile, (g.M + kTile - 1) / kTile, splits);
    hipLaunchKernelGGL((gemm_f32_kernel<AT, BT, Epi>), grid, dim3(256), 0, s, g, epi);
    return hipGetLastError();
}

}  // namespace aae
