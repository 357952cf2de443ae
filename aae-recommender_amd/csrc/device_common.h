// Device-side helpers shared by every kernel of libaaerec_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kTiny = 1e-12f;                       // TINY, reference aaerec/aae.py:28
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;

// (r6) activations 6+: the parameter-free element-wise classes of torch.nn at their default arguments - the reference takes
// ANY class name (getattr(nn, activation)(), aae.py:110); classes with parameters, state or a row-wise definition (PReLU,
// RReLU, Threshold, GLU, Softmax ...) are refused by the host with the list of these names.
enum { ACT_RELU = 0, ACT_SELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_ELU = 4, ACT_LEAKY = 5,
       ACT_SOFTPLUS = 6, ACT_HARDTANH = 7, ACT_RELU6 = 8, ACT_CELU = 9, ACT_SOFTSIGN = 10, ACT_HARDSIGMOID = 11,
       ACT_LOGSIGMOID = 12, ACT_SOFTSHRINK = 13, ACT_HARDSHRINK = 14, ACT_IDENTITY = 15,
       ACT_GELU = 16, ACT_SILU = 17, ACT_MISH = 18, ACT_HARDSWISH = 19, ACT_COUNT = 20 };

__device__ __forceinline__ float sigmoidf_(float x) {
    // two-sided form keeps full precision for large |x|
    if (x >= 0.f) return 1.f / (1.f + __expf(-x));
    float e = __expf(x);
    return e / (1.f + e);
}

// ---- GELU / SiLU / Mish / Hardswish are not monotone: y = f(x) falls from 0 to f(x*) on x < x* and rises beyond, so the
// derivative cannot be written in the layer's OUTPUT alone - and the output is all the backward pass of these kernels keeps
// (an activation's input never reaches memory: dropout and activation run on the accumulators of the layer's product).
// The one missing bit - which side of x* the input was on - rides in the LEAST SIGNIFICANT BIT of the stored output (one unit
// in the last place of y, 6e-8 relative: far inside the tolerances of this path), and the backward pass inverts f on that
// branch (bracketed Newton steps; Hardswish in closed form) to evaluate f'(x).  Near x* the inverse is ill-conditioned but
// f' is ~0 there: the error of f'(x) stays below 1e-4 absolute on the ~1e-3 wide neighbourhood of x*, 1e-6 elsewhere.
__device__ __forceinline__ float nm_xstar(int act) {        // argmin of f
    return act == ACT_GELU ? -0.75179160f : act == ACT_SILU ? -1.27846455f : act == ACT_MISH ? -1.19245934f : -1.5f;
}
__device__ __forceinline__ void nm_f_df(int act, float x, float& f, float& df) {
    if (act == ACT_GELU) {                  // x Phi(x);  Phi + x phi   (torch.nn.GELU(approximate='none'))
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f)), pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        f = x * cdf; df = cdf + x * pdf;
    } else if (act == ACT_SILU) {           // x s(x);  s (1 + x (1 - s))
        const float sg = sigmoidf_(x);
        f = x * sg; df = sg * (1.f + x * (1.f - sg));
    } else {                                // Mish: x tanh(softplus(x));  t + x s (1 - t^2)
        const float sp = x > 20.f ? x : log1pf(__expf(x)), t = tanhf(sp), sg = sigmoidf_(x);
        f = x * t; df = t + x * sg * (1.f - t * t);
    }
}
__device__ __forceinline__ float nm_mark(int act, float x, float y) {      // the branch bit into the stored output
    const unsigned u = (__float_as_uint(y) & ~1u) | (x < nm_xstar(act) ? 1u : 0u);
    return __uint_as_float(u);
}
__device__ float nm_grad_from_y(int act, float ym) {
    const bool left = (__float_as_uint(ym) & 1u) != 0u;
    const float y = __uint_as_float(__float_as_uint(ym) & ~1u);
    if (act == ACT_HARDSWISH) {             // x relu6(x + 3) / 6: 0 | x (x + 3) / 6 | x;  torch: 0 | x / 3 + 0.5 | 1
        if (y >= 3.f) return 1.f;
        if (left && y == 0.f) return 0.f;   // x <= -3 (the one point x = -3 itself has measure zero)
        const float r = sqrtf(fmaxf(9.f + 24.f * y, 0.f)) * (1.f / 6.f);
        return left ? -r : r;
    }
    const float xs = nm_xstar(act);
    // bracket [lo, hi] with (f(lo) - y) and (f(hi) - y) of opposite sign on the branch's monotone piece
    float lo = left ? -40.f : xs, hi = left ? xs : fmaxf(y, 0.f) + 2.f;
    float x = left ? xs - 1.f : fmaxf(y, xs + 0.5f);
    for (int it = 0; it < 24; ++it) {
        float f, df;
        nm_f_df(act, x, f, df);
        const float r = f - y;
        // left branch: f falls with x (r > 0: the root lies to the right... of a falling f: larger x); right: f rises
        const bool root_above = left ? r > 0.f : r < 0.f;
        if (root_above) lo = x; else hi = x;
        float xn = x - r / df;
        if (!(xn > lo && xn < hi)) xn = 0.5f * (lo + hi);      // (also the NaN of df == 0)
        x = xn;
    }
    float f, df;
    nm_f_df(act, x, f, df);
    return df;
}

// getattr(nn, activation)() of the reference (aae.py:110).  Applied AFTER dropout.
// EXT = false (the 4-row and 16-row chain kernels, at their 128-register cap): r1-r5's six classes only - the r6 classes inlined
// there cost the 16-row kernel 74 spilled registers on EVERY activation's path.  A model with one of the r6 classes runs its
// layer programs on chain_kernel<.., true> (16-row blocks on the fp32 pipe), the instantiation that carries them all.
template <bool EXT = true>
__device__ __forceinline__ float act_fwd(int act, float x) {
    if (!EXT && act > ACT_LEAKY) return x;
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_SELU: return x > 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * expm1f(x);
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return sigmoidf_(x);
        case ACT_ELU: case ACT_CELU: return x > 0.f ? x : expm1f(x);          // (CELU(alpha = 1) is ELU(alpha = 1))
        case ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
        case ACT_SOFTPLUS: return x > 20.f ? x : log1pf(__expf(x));           // beta = 1, threshold = 20
        case ACT_HARDTANH: return fminf(fmaxf(x, -1.f), 1.f);
        case ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
        case ACT_SOFTSIGN: return x / (1.f + fabsf(x));
        case ACT_HARDSIGMOID: return fminf(fmaxf(x * (1.f / 6.f) + 0.5f, 0.f), 1.f);
        case ACT_LOGSIGMOID: return fminf(x, 0.f) - log1pf(__expf(-fabsf(x)));
        case ACT_SOFTSHRINK: return x > 0.5f ? x - 0.5f : x < -0.5f ? x + 0.5f : 0.f;
        case ACT_HARDSHRINK: return fabsf(x) > 0.5f ? x : 0.f;
        case ACT_IDENTITY: return x;
        case ACT_GELU: return nm_mark(act, x, x * 0.5f * (1.f + erff(x * 0.70710678118654752f)));
        case ACT_SILU: return nm_mark(act, x, x * sigmoidf_(x));
        case ACT_MISH: return nm_mark(act, x, x * tanhf(x > 20.f ? x : log1pf(__expf(x))));
        case ACT_HARDSWISH: return nm_mark(act, x, x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f));
        default: return x;
    }
}

// derivative of the activation expressed through its OUTPUT y, so the backward pass needs the post-activation tensor only
// (the monotone ones directly; GELU / SiLU / Mish / Hardswish through the branch bit, above).
// EXT = false (the 4-row / 16-row chain kernels): r1-r5's six classes, as act_fwd<false>; the per-layer epilogues and
// chain_kernel<.., true> carry all of them.
template <bool EXT = false>
__device__ __forceinline__ float act_grad_from_y(int act, float y) {
    if (EXT && act >= ACT_GELU) return nm_grad_from_y(act, y);
    if (!EXT) {
        switch (act) {
            case ACT_RELU: return y > 0.f ? 1.f : 0.f;
            case ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
            case ACT_TANH: return 1.f - y * y;
            case ACT_SIGMOID: return y * (1.f - y);
            case ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
            default: return y > 0.f ? 1.f : 0.01f;
        }
    }
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
        case ACT_TANH: return 1.f - y * y;
        case ACT_SIGMOID: return y * (1.f - y);
        case ACT_ELU: case ACT_CELU: return y > 0.f ? 1.f : y + 1.f;
        case ACT_LEAKY: return y > 0.f ? 1.f : 0.01f;
        case ACT_SOFTPLUS: return -expm1f(-y);                               // sigmoid(x) = 1 - exp(-softplus(x)); x > 20: 1
        case ACT_HARDTANH: return (y > -1.f && y < 1.f) ? 1.f : 0.f;
        case ACT_RELU6: return (y > 0.f && y < 6.f) ? 1.f : 0.f;
        case ACT_SOFTSIGN: { const float t = 1.f - fabsf(y); return t * t; }   // 1 / (1 + |x|)^2, 1 - |y| = 1 / (1 + |x|)
        case ACT_HARDSIGMOID: return (y > 0.f && y < 1.f) ? (1.f / 6.f) : 0.f;
        case ACT_LOGSIGMOID: return -expm1f(y);                               // sigmoid(-x) = 1 - exp(logsigmoid(x))
        case ACT_SOFTSHRINK: case ACT_HARDSHRINK: return y != 0.f ? 1.f : 0.f;
        case ACT_IDENTITY: return 1.f;
        default: return 1.f;
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
// r5: wide batches run the layer chains 16 rows per workgroup on the bf16 matrix cores with every fp32 product emulated by
// six bf16 products (chain16x3.h, as the output layer since r3): the B operand of v_mfma_f32_16x16x32_bf16 is a lane's
// 8 consecutive k of one column as bf16, and splitting 40 000 weights into three bf16 terms per workgroup and op would cost more
// vector work than the products.  So the split weights are kept as a third and fourth copy, three bf16 planes each:
//   FX (forward, k = input column i, n = output row o):   FX[(((i >> 5) * 3 + t) * Mp + o) * 32 + (i & 31)]     Mp = M up to 16
//   DX (dX,      k = output row o,   n = input column i): DX[(((o >> 5) * 3 + t) * Np + i) * 32 + (o & 31)]     Np = N up to 16
//   FXB: FX with k counted from input column `split` (the second k-part of a layer whose input is wider than a slot)
// term t of a weight w: w = t0 + t1 + t2 exactly, each a bf16 value (round to nearest even of what is left).  Planes never
// written (k beyond the matrix, columns beyond it) stay zero for the life of the arena: the kernel reads whole 32-deep
// k-steps and 16-column tiles unclamped.  fx == NULL: the model never runs a batch wide enough (aae_create).
struct W4Copies { float* f4; float* d4; int M, N; unsigned short* fx; unsigned short* dx; unsigned short* fxb; int split; };      // f4 == NULL: the layer has none
constexpr int kW4Pad = 4;      // zero k-chunk rows behind a k4-interleaved copy (what an unclamped run of chunks may read)
typedef __bf16 w4_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned w4_rne_pair(float a, float b) {       // (low half = a) v_cvt_pk_bf16_f32: ties to even
    w4_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, w4_bf16x2));
}
// v -> its three bf16 terms (bit patterns)
__device__ __forceinline__ void w4_split3(float v, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
    const unsigned p0 = w4_rne_pair(v, 0.f);
    const float r1 = v - __uint_as_float(p0 << 16);
    const unsigned p1 = w4_rne_pair(r1, 0.f);
    const float r2 = r1 - __uint_as_float(p1 << 16);
    t0 = (unsigned short)p0; t1 = (unsigned short)p1; t2 = (unsigned short)w4_rne_pair(r2, 0.f);
}
__device__ __forceinline__ void w4x_put1(const W4Copies& c, int o, int i, float v) {
    unsigned short t[3];
    w4_split3(v, t[0], t[1], t[2]);
    const int Mp = (c.M + 15) & ~15, Np = (c.N + 15) & ~15;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        c.fx[((size_t)((i >> 5) * 3 + q) * Mp + o) * 32 + (i & 31)] = t[q];
        c.dx[((size_t)((o >> 5) * 3 + q) * Np + i) * 32 + (o & 31)] = t[q];
    }
    if (c.fxb && i >= c.split) {
        const int j = i - c.split;
#pragma unroll
        for (int q = 0; q < 3; ++q) c.fxb[((size_t)((j >> 5) * 3 + q) * Mp + o) * 32 + (j & 31)] = t[q];
    }
}
// element (o, i) of P just became v
__device__ __forceinline__ void w4_put1(const W4Copies& c, int o, int i, float v) {
    c.f4[((size_t)(i >> 2) * c.M + o) * 4 + (i & 3)] = v;
    c.d4[((size_t)(o >> 2) * c.N + i) * 4 + (o & 3)] = v;
    if (c.fx) w4x_put1(c, o, i, v);
}
// elements (o, i .. i + 3), i % 4 == 0, i + 3 < N
__device__ __forceinline__ void w4_put4(const W4Copies& c, int o, int i, float4 v) {
    *reinterpret_cast<float4*>(c.f4 + ((size_t)(i >> 2) * c.M + o) * 4) = v;
    float* d = c.d4 + ((size_t)(o >> 2) * c.N + i) * 4 + (o & 3);
    d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
    if (c.fx) {
        // FX: the four k are neighbours inside one 32-deep step (i % 4 == 0): one 8-byte store per term; DX: four columns
        unsigned short t[4][3];
        w4_split3(v.x, t[0][0], t[0][1], t[0][2]); w4_split3(v.y, t[1][0], t[1][1], t[1][2]);
        w4_split3(v.z, t[2][0], t[2][1], t[2][2]); w4_split3(v.w, t[3][0], t[3][1], t[3][2]);
        const int Mp = (c.M + 15) & ~15, Np = (c.N + 15) & ~15;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint2 pk = make_uint2((unsigned)t[0][q] | ((unsigned)t[1][q] << 16), (unsigned)t[2][q] | ((unsigned)t[3][q] << 16));
            *reinterpret_cast<uint2*>(c.fx + ((size_t)((i >> 5) * 3 + q) * Mp + o) * 32 + (i & 31)) = pk;
            unsigned short* dq = c.dx + ((size_t)((o >> 5) * 3 + q) * Np + i) * 32 + (o & 31);
            dq[0] = t[0][q]; dq[32] = t[1][q]; dq[64] = t[2][q]; dq[96] = t[3][q];
            if (c.fxb && i >= c.split) {     // (split % 4 == 0: the four k stay neighbours)
                const int j = i - c.split;
                *reinterpret_cast<uint2*>(c.fxb + ((size_t)((j >> 5) * 3 + q) * Mp + o) * 32 + (j & 31)) = pk;
            }
        }
    }
}
}  // namespace aae
