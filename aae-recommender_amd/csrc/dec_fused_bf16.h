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
//   per tile, one persistent 512-thread workgroup per CU (8 waves, 2 per SIMD: the 256-register budget holds the
//   wave's share of dh2^T, of the dA2 accumulators and three tiles' worth of parameter / moment stream in flight
//   without spilling - a scratch reload would wait for every older global load), 4 LDS-only barriers:
//     S0     this tile's V3a (registers, requested one tile ahead) -> v3K, v3T; request V3a(t+1), m(t), v(t)
//     GEMM1  logits[b][n] = dhA * v3K^T; epilogue: zero-target BCE -> gK, gT (bf16), raw logits (fp32), loss
//     S2     the tile's CSR entries (non-zero targets) patch their cells
//     GEMM2  dV3a^T[c][n] = dhT * gT^T: the accumulator layout (4 consecutive c of one item per lane) is the layout
//            the wave loaded p / m / v in, so dec_optim (Adam) runs on the accumulators and stores float4 - no
//            staging tile, no barrier;   GEMM3  dA2[b][c] += gK * v3T^T (fp32 accumulators across the tiles)
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

constexpr int kBfSR = 33;      // row stride (floats) of the raw-logit tile [b][n]

// LDS bytes for (B, NB)
inline size_t dec_fused_bf16_lds_bytes(int NB) {
    const int KC1 = (NB + 1) / 2, S1 = bf_stride(KC1), S2 = bf_stride(4), S3 = bf_stride(1);
    return sizeof(float) * ((size_t)kGR * S1 + (size_t)kTI * S1 + (size_t)16 * NB * S3 + (size_t)kGR * S3 +
                            (size_t)kTI * S2 + (size_t)kGR * kBfSR + 64);
}

constexpr int kBT = 512;       // threads per workgroup
constexpr int kBW = kBT / 64;  // waves

template <int NB>   // NB = ceil((h + 1) / 16) column blocks
__global__ __launch_bounds__(kBT) void dec_fused_bf16_kernel(DecFusedArgs a) {
    constexpr int KC1 = (NB + 1) / 2;          // 32-wide k-steps over the h + 1 hidden columns
    constexpr int KR = 4;                       // 32-wide k-steps over the (<= 112 -> 128) batch rows
    constexpr int S1 = bf_stride(KC1), S2 = bf_stride(KR), S3 = bf_stride(1);
    // Ownership.  GEMM1: wave w < row blocks owns row block w (both item halves).  Everything that touches the
    // parameter stream - the V3a / m / v loads, the LDS images of the tile, GEMM2 + optimiser, GEMM3 - is owned by
    // column block: wave w owns column blocks w, w + 8 (QC of them), each as 2 float4 slots per lane (item n = 16 ib +
    // fr, columns 16 cb + 4 fk ..+3): exactly the accumulator layout of GEMM2 (rows = columns c, cols = items).
    constexpr int QC = (NB + kBW - 1) / kBW;
    constexpr int NS = 2 * QC;                  // float4 slots per lane and stream
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* dhA = reinterpret_cast<unsigned*>(lds);          // [kGR][S1]   dh2, k = hidden column
    unsigned* v3K = dhA + kGR * S1;                             // [32][S1]    V3a tile, k = hidden column
    unsigned* v3T = v3K + kTI * S1;                             // [16 NB][S3] V3a tile transposed, k = item
    unsigned* gK = v3T + 16 * NB * S3;                          // [kGR][S3]   G, k = item
    unsigned* gT = gK + kGR * S3;                               // [32][S2]    G transposed, k = batch row
    float* raw = reinterpret_cast<float*>(gT + kTI * S2);       // [kGR][kBfSR] raw logits (fp32) for the entry patch
    float* red = raw + kGR * kBfSR;                             // [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int B = a.B, ldv = a.ldv, N = a.N;
    const int nmb = (B + 15) >> 4;
    const int ntiles = (N + kTI - 1) / kTI;
    const OptScalars sc = *a.sc;
    const bool do_adam = a.gradV3 == nullptr;

    // ---- once per workgroup: LDS images of dh2 (zero padded), zeroed G images, zero k-padding of the V3a images
    for (int i = tid; i < kGR * S1 + kTI * S1 + 16 * NB * S3 + kGR * S3 + kTI * S2; i += kBT) dhA[i] = 0u;
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
    // dh2 transposed, in registers: for each owned column block the fragment rows c = 16 cb + fr, k = batch row,
    // picked out of the (zero padded, already rounded) dhA image: no masks, no second pass over global memory
    bf16x8 dhT[QC][KR];
    {
        const unsigned short* a16 = reinterpret_cast<const unsigned short*>(dhA);
#pragma unroll
        for (int j = 0; j < QC; ++j) {
            const int c = 16 * min(wave + kBW * j, NB - 1) + fr;
#pragma unroll
            for (int kc = 0; kc < KR; ++kc) {
                unsigned h[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int b = 32 * kc + 4 * fk + (e & 3) + ((e >> 2) << 4);          // < kGR unless kc == 3 && e >= 4
                    h[e] = (32 * kc + ((e >> 2) << 4) + 15 < kGR) ? (unsigned)a16[b * (2 * S1) + c] : 0u;
                }
                const u32x4_t v = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
                dhT[j][kc] = __builtin_bit_cast(bf16x8, v);
            }
        }
    }
    f32x4 acc3[QC][kMB];
#pragma unroll
    for (int j = 0; j < QC; ++j)
#pragma unroll
        for (int q = 0; q < kMB; ++q) acc3[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float loss = 0.f;

    // slot s = 2 j + ib: column block cb = wave + 8 j, item half ib.  Addressing: a wave-uniform base (scalar registers:
    // tensor + tile + slot) plus ONE per-lane byte offset shared by every load and store of the kernel.  Nothing is
    // clamped: the arena keeps kTI zero rows behind V3a and its moments (layout(): pad_rows), so the tail tile's rows
    // >= N and the float4 behind a row's last column are readable (and never stored, never used).
    const unsigned lane_off = (unsigned)(fr * ldv + 4 * fk) * 4u;
    auto slot_c0 = [&](int s) { return 16 * (wave + kBW * (s >> 1)) + 4 * fk; };
    auto slot_ok = [&](int s) { return wave + kBW * (s >> 1) < NB && slot_c0(s) < ldv; };   // (ldv % 4 == 0)
    auto slot_ptr = [&](float* P, int tile, int s) -> char* {          // wave-uniform part of a slot's address
        return reinterpret_cast<char*>(P + ((size_t)tile * kTI + 16 * (s & 1)) * ldv + 16 * (wave + kBW * (s >> 1)));
    };
    auto ld4 = [&](float* P, int tile, int s) { return *reinterpret_cast<const float4*>(slot_ptr(P, tile, s) + lane_off); };
    auto st4 = [&](float* P, int tile, int s, float4 v) { *reinterpret_cast<float4*>(slot_ptr(P, tile, s) + lane_off) = v; };
    float4 p_cur[NS], p_nxt[NS], mreg[NS], sreg[NS];
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    const int last_e = max(a.te.start[ntiles] - 1, 0);
    auto load_range = [&](int t, int& lo, int& hi) {
        const int tc = min(t, ntiles - 1);
        lo = a.te.start[tc]; hi = a.te.start[tc + 1];
        if (t >= ntiles) hi = lo;
    };
    int ce0 = 0, ce1 = 0, ne0 = 0, ne1 = 0, fe0 = 0, fe1 = 0;
    int ent_b = 0, ent_n = 0; float ent_v = 0.f;
    auto load_entry = [&](int lo) {
        const int e = min(lo + tid, last_e);
        ent_b = a.te.eb[e]; ent_n = a.te.en[e]; ent_v = a.te.ev[e];
    };
    if (tile < ntiles) {
#pragma unroll
        for (int s = 0; s < NS; ++s) p_nxt[s] = ld4(a.V3a, tile, s);
        load_range(tile, ne0, ne1);
        load_range(tile + stride, fe0, fe1);
        load_entry(ne0);
    }
    __syncthreads();

    for (; tile < ntiles; tile += stride) {
        const int i0 = tile * kTI;
        lds_barrier();                                  // the previous tile's readers of v3K / v3T / gK / gT are done
        // ---- S0: this tile's V3a (fp32 registers) -> bf16 LDS images; rotate the pipeline, request the next stage
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            p_cur[s] = p_nxt[s];
            if (slot_ok(s)) {
                const int n = 16 * (s & 1) + fr, c0 = slot_c0(s);
                float4 p = p_cur[s];
                if (i0 + n >= N) p = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(v3K + n * S1 + (c0 >> 1)) = make_uint2(bf16_pack(p.x, p.y), bf16_pack(p.z, p.w));
                unsigned short* t16 = reinterpret_cast<unsigned short*>(v3T);
                t16[(c0 + 0) * (2 * S3) + n] = bf16_bits(p.x);
                t16[(c0 + 1) * (2 * S3) + n] = bf16_bits(p.y);
                t16[(c0 + 2) * (2 * S3) + n] = bf16_bits(p.z);
                t16[(c0 + 3) * (2 * S3) + n] = bf16_bits(p.w);
            }
        }
        ce0 = ne0; ce1 = ne1; ne0 = fe0; ne1 = fe1;
        const int my_b = ent_b, my_n = ent_n; const float my_v = ent_v;
        {
            const int tn = min(tile + stride, ntiles - 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // (all three unconditional - the moment tensors exist in every mode: a load under a condition is waited
                // for on the spot and its result is carried in duplicate registers down both paths)
                p_nxt[s] = ld4(a.V3a, tn, s);
                mreg[s] = ld4(a.M, tile, s);
                sreg[s] = ld4(a.V, tile, s);
            }
        }
        load_range(tile + 2 * stride, fe0, fe1);
        load_entry(ne0);
        lds_barrier();

        // ---- GEMM1: logits[b][n] for the wave's row block, both item halves; epilogue = BCE against a zero target
        if (wave < nmb) {
            const int mb = wave;
            f32x4 c[2];
            c[0] = c[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC1; ++kc) {
                const bf16x8 x = bf_frag(dhA, 16 * mb + fr, S1, kc, fk);
                c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, bf_frag(v3K, fr, S1, kc, fk), c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, bf_frag(v3K, 16 + fr, S1, kc, fk), c[1], 0, 0, 0);
                // (two k-steps of fragment reads in flight at a time: left alone the scheduler hoists all 21 fragments
                // of the phase above its first MFMA - 84 registers on top of ~150 persistent ones - and spills)
                if (kc & 1) __builtin_amdgcn_sched_barrier(0);
            }
            // C map: row = 4 fk + r -> batch row, col = fr -> item
            const int rb = 16 * mb + 4 * fk;
            unsigned short* k16 = reinterpret_cast<unsigned short*>(gK);
#pragma unroll
            for (int nb2 = 0; nb2 < 2; ++nb2) {
                const int n = 16 * nb2 + fr;
                const bool item_ok = i0 + n < N;
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float l = 0.f;
                    g[r] = 0.f;
                    if (rb + r < B) {
                        if (item_ok) bce_elem_t0(c[nb2][r], a.gscale, g[r], l);
                        raw[(rb + r) * kBfSR + n] = c[nb2][r];
                        k16[(rb + r) * (2 * S3) + n] = bf16_bits(g[r]);
                        loss += l;
                    }
                }
                *reinterpret_cast<uint2*>(gT + n * S2 + (rb >> 1)) = make_uint2(bf16_pack(g[0], g[1]), bf16_pack(g[2], g[3]));
            }
        }
        lds_barrier();

        // ---- S2: the CSR entries of the tile (non-zero targets) replace their cell's gradient and loss term
        {
            unsigned short* k16 = reinterpret_cast<unsigned short*>(gK);
            unsigned short* t16 = reinterpret_cast<unsigned short*>(gT);
            auto patch = [&](int b, int n, float v) {
                const float lg = raw[b * kBfSR + n];
                float g0, l0, g1, l1;
                bce_elem_t0(lg, a.gscale, g0, l0);
                bce_elem(lg, v, a.gscale, g1, l1);
                loss += l1 - l0;
                const unsigned short h = bf16_bits(g1);
                k16[b * (2 * S3) + n] = h;
                t16[n * (2 * S2) + b] = h;
            };
            if (tid < ce1 - ce0) patch(my_b, my_n, my_v);
            for (int e = ce0 + kBT + tid; e < ce1; e += kBT) patch(a.te.eb[e], a.te.en[e], a.te.ev[e]);
        }
        lds_barrier();

        // ---- GEMM2 + optimiser: dV3a^T[c][n] = sum_b dh2[b][c] G[b][n]; lane holds columns c0..c0+3 of item n
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (wave + kBW * (s >> 1) < NB) {           // (wave-uniform)
                const int ib = s & 1;
                f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KR; ++kc)
                    g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dhT[s >> 1][kc], bf_frag(gT, 16 * ib + fr, S2, kc, fk), g, 0, 0, 0);
                if (slot_c0(s) < ldv && i0 + 16 * ib + fr < N) {
                    if (!do_adam) {
                        st4(a.gradV3, tile, s, make_float4(g[0], g[1], g[2], g[3]));
                    } else {
                        float4 p = p_cur[s], mm = mreg[s], vv = sreg[s];
                        adam_update(p.x, mm.x, vv.x, g[0], sc); adam_update(p.y, mm.y, vv.y, g[1], sc);
                        adam_update(p.z, mm.z, vv.z, g[2], sc); adam_update(p.w, mm.w, vv.w, g[3], sc);
                        st4(a.V3a, tile, s, p);
                        if (!sc.is_sgd) {
                            st4(a.M, tile, s, mm);
                            st4(a.V, tile, s, vv);
                        }
                    }
                }
            }
        }
        // ---- GEMM3: dA2[b][c] += sum_n G[b][n] V3a[n][c] for the wave's column blocks, every row block
#pragma unroll
        for (int j = 0; j < QC; ++j) {
            if (wave + kBW * j < NB) {
                const bf16x8 vt = bf_frag(v3T, 16 * (wave + kBW * j) + fr, S3, 0, fk);
#pragma unroll
                for (int q = 0; q < kMB; ++q)
                    if (q < nmb)
                        acc3[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_frag(gK, 16 * q + fr, S3, 0, fk), vt, acc3[j][q], 0, 0, 0);
            }
        }
    }

    // ---- dA2 partial of this workgroup -> its slab; loss partial
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int j = 0; j < QC; ++j) {
        if (wave + kBW * j < NB) {
#pragma unroll
            for (int q = 0; q < kMB; ++q) {
                const int rb = q * 16 + fk * 4, cc = (wave + kBW * j) * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rb + r < B && cc < a.ld_slab) slab[(size_t)(rb + r) * a.ld_slab + cc] = acc3[j][q][r];
            }
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
    }
}

}  // namespace aae
