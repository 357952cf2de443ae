// Device-side helpers shared by every kernel of libaaerec_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kTiny = 1e-12f;                       // TINY, reference aaerec/aae.py:28
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;

enum { ACT_RELU = 0, ACT_SELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_ELU = 4, ACT_LEAKY = 5 };

__device__ __forceinline__ float sigmoidf_(float x) {
    // two-sided form keeps full precision for large |x|
    if (x >= 0.f) return 1.f / (1.f + __expf(-x));
    float e = __expf(x);
    return e / (1.f + e);
}

// getattr(nn, activation)() of the reference (aae.py:110).  Applied AFTER dropout.
__device__ __forceinline__ float act_fwd(int act, float x) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_SELU: return x > 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * expm1f(x);
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return sigmoidf_(x);
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        default: return x > 0.f ? x : 0.01f * x;
    }
}

// derivative of the activation expressed through its OUTPUT y (all six are invertible enough
// for that), so the backward pass needs the post-activation tensor only.
__device__ __forceinline__ float act_grad_from_y(int act, float y) {
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
        case ACT_TANH: return 1.f - y * y;
        case ACT_SIGMOID: return y * (1.f - y);
        case ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        default: return y > 0.f ? 1.f : 0.01f;
    }
}

// ---------------------------------------------------------------------------------------
// dropout.  nn.Dropout: u = a * keep/(1-p).  nn.AlphaDropout (SELU): u = a*(keep*A) + B(keep).
// The keep bit comes from an injected uint8 mask or from the counter-based generator.
// ---------------------------------------------------------------------------------------
struct DropSpec {
    const uint8_t* mask_a;   // rows <  split_row   (NULL + !device_rng => keep all)
    const uint8_t* mask_b;   // rows >= split_row   (indexed from row - split_row)
    int split_row;
    int width;               // mask row stride (= layer width)
    int enabled;             // 0: identity (eval mode or p == 0)
    int device_rng;          // 1: hash generator instead of masks
    uint32_t keep_threshold; // device rng: keep iff u32 >= threshold  (threshold = p * 2^32)
    uint32_t stream_id;      // which of the 12 draws of a step
    float mul_keep;          // 1/(1-p)            | a
    float add_keep;          // 0                  | alpha*a*p
    float add_drop;          // 0                  | -alpha*a + alpha*a*p
    // device rng: the generator is keyed by the row of the GLOBAL batch, so a data-parallel rank that holds rows
    // [o, o + B_l) of B_g draws what a single process draws for those rows: global row = row + goff_a for rows below
    // split_row, row + goff_b for the rest (the discriminator's stacked [z_real; z_fake] rows)
    int goff_a, goff_b;
};

__device__ __forceinline__ uint32_t hash_u32(uint64_t key, uint64_t ctr) {
    uint64_t x = key ^ (ctr * 0x9E3779B97F4A7C15ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (uint32_t)x;
}

__device__ __forceinline__ uint64_t rng_key(uint64_t seed, uint64_t step, uint32_t stream) {
    return seed ^ (step * 0xD1B54A32D192ED03ull) ^ ((uint64_t)stream << 56);
}

// One 32-bit word per (row, col) cell of a dropout stream: the two key halves are offset by odd
// multiples of row and col, then two multiply-xorshift rounds (the "lowbias32" constants).  Three
// 32-bit multiplies per cell where the 64-bit mixer above needs sixteen (v_mul_*_u32 are
// quarter-rate) - dropout masks are drawn inside the 16-row layer chain, on a single CU per block.
__device__ __forceinline__ uint32_t hash_cell(uint64_t key, uint32_t row, uint32_t col) {
    uint32_t x = ((uint32_t)key + row * 0x9E3779B1u) ^ ((uint32_t)(key >> 32) + col * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// returns keep in {0,1}
__device__ __forceinline__ int drop_keep(const DropSpec& d, uint64_t key, int row, int col) {
    if (d.device_rng)
        return hash_cell(key ^ ((uint64_t)d.stream_id * 0xA0761D6478BD642Full),
                         (uint32_t)(row + (row < d.split_row ? d.goff_a : d.goff_b)), (uint32_t)col) >= d.keep_threshold;
    const uint8_t* m = row < d.split_row ? d.mask_a : d.mask_b;
    if (!m) return 1;
    int r = row < d.split_row ? row : row - d.split_row;
    return m[(size_t)r * d.width + col] != 0;
}

__device__ __forceinline__ float drop_fwd(const DropSpec& d, int keep, float a) {
    return keep ? a * d.mul_keep + d.add_keep : d.add_drop;
}
__device__ __forceinline__ float drop_bwd_mul(const DropSpec& d, int keep) { return keep ? d.mul_keep : 0.f; }

// ---------------------------------------------------------------------------------------
// optimiser scalars, advanced on the device once per optimiser step so that a captured
// hipGraph replays without host-side arguments.  torch.optim.Adam (single-tensor path):
//   step_size = lr / (1 - b1^t);  bc2_sqrt = sqrt(1 - b2^t)
// ---------------------------------------------------------------------------------------
struct OptScalars {
    long long t;          // step count
    float neg_step_size;  // -(lr / bc1)   (SGD: -lr)
    float bc2_sqrt;
    int is_sgd;
    double lr;
    float inv_bc2_sqrt;   // 1 / bc2_sqrt
    int pad_;
};

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const OptScalars& s) {
    if (s.is_sgd) { p = p + s.neg_step_size * g; return; }
    m = m + 0.1f * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * 0.999f + (0.001f * g) * g;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    // (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps); param.addcdiv_(exp_avg, denom, -step_size) with the
    // 1-ulp hardware sqrt / reciprocal (v_sqrt_f32, v_rcp_f32): |error| < 3e-7 of an O(lr) update
    float denom = __builtin_amdgcn_sqrtf(v) * s.inv_bc2_sqrt + 1e-8f;
    p = p + (s.neg_step_size * m) * __builtin_amdgcn_rcpf(denom);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

}  // namespace aae

namespace aae {
// ---------------------------------------------------------------------------------------------
// k4-interleaved copies of a hidden layer's augmented weights P [M = out][N = in + 1] for chain4.h, whose lanes each own
// ONE output column and walk k: with four consecutive k of a column contiguous a lane takes them in one 16-byte load and
// a wave-instruction is one contiguous 1 KB (a quarter of the vector-memory instructions of the k-major form - the
// layer ops are bound by getting their weights through the CU's memory pipeline):
//   F4 (forward, k = input column i, n = output row o):   F4[((i >> 2) * M + o) * 4 + (i & 3)]
//   D4 (dX,      k = output row o,   n = input column i): D4[((o >> 2) * N + i) * 4 + (o & 3)]
// Both are kept in step with P by the optimiser epilogues that write P (w4_put4 / w4_put1) and re-derived by
// interleave4_kernel after any other writer.
// ---------------------------------------------------------------------------------------------
struct W4Copies { float* f4; float* d4; int M, N; };      // f4 == NULL: the layer has none
constexpr int kW4Pad = 4;      // zero k-chunk rows behind a k4-interleaved copy (what an unclamped run of chunks may read)
// element (o, i) of P just became v
__device__ __forceinline__ void w4_put1(const W4Copies& c, int o, int i, float v) {
    c.f4[((size_t)(i >> 2) * c.M + o) * 4 + (i & 3)] = v;
    c.d4[((size_t)(o >> 2) * c.N + i) * 4 + (o & 3)] = v;
}
// elements (o, i .. i + 3), i % 4 == 0, i + 3 < N
__device__ __forceinline__ void w4_put4(const W4Copies& c, int o, int i, float4 v) {
    *reinterpret_cast<float4*>(c.f4 + ((size_t)(i >> 2) * c.M + o) * 4) = v;
    float* d = c.d4 + ((size_t)(o >> 2) * c.N + i) * 4 + (o & 3);
    d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
}
}  // namespace aae
