// Row-blocked layer chains: the hidden stacks of Encoder / Decoder / Discriminator (forward and
// the activation-gradient half of backward) are row-local, so a workgroup can carry 16 rows of
// the batch through a whole sequence of layers without leaving the CU: activations sit in LDS
// "slots" [16][kCL], weights stream from L2 straight into MFMA operand registers, and every layer
// is a handful of v_mfma_f32_16x16x4_f32 per wave.  One launch replaces up to 14 tiny GEMM /
// element-wise launches whose 5-8 us dispatch floor dominated the B=100 step.
//
// A chain is a small program of ChainOps executed in order by every workgroup on its own 16 rows
// (barrier between ops).  The weight-gradient products (sums over the batch) are NOT row-local:
// the chains store the per-layer gradients/activations they need and grouped_dw_kernel
// (below) turns them into optimiser updates.
#pragma once
#include "device_common.h"
#include "gemm_f32.h"
#include "kernels.h"
#include "buckets.h"
#include "w1_update.h"

namespace aae {

constexpr int kCR = 16;        // rows per workgroup
constexpr int kCL = 212;       // slot row stride (floats): widths up to 208, 16-byte aligned rows
constexpr int kCSlots = 10;
constexpr int kCWide = 208;    // columns of a slot row a layer may read (16-byte aligned rows of kCL floats); a wider decoder input runs in two k-parts
constexpr int kCT = 1024;      // threads per workgroup
constexpr int kCW = kCT / 64;  // waves: 16 (one 16-column block of a layer per wave) or 8 (two blocks per wave)
constexpr int kCQ = (13 + kCW - 1) / kCW;   // blocks per wave (N <= 208 -> 13 blocks)
constexpr int kTPR = kCT / 16; // threads per row of the 16-row block in element-wise work
constexpr int kRS = kCT == 1024 ? 6 : 5;    // log2(kTPR)
constexpr int kJ = 64 / kTPR;  // float4 column groups per thread (64 float4 = 256 columns per row)
constexpr int kCMaxOps = 16;

enum { COP_LOAD = 0,        // dst <- global src [rows][lds_] cols [0, N)            (zero padded)
       COP_LINEAR,          // dst[16][N] = epi( src[16][K] * W[N][K]^T )             W k-contiguous rows
       COP_LINEAR_DX,       // dst[16][N] = epi( src[16][K] * W[K][0:N] )             W n-contiguous rows
       COP_FINAL_FWD,       // encoder output activation on dst (in place), width N
       COP_FINAL_BWD,       // dst <- d(final act): src = dL/dz, yslot = z
       COP_ADV,             // adversarial loss + gradient on the discriminator output (slot src col 0 -> dst col 0)
       COP_DROPACT,         // dst <- act(dropout(src)), width N
       COP_SLABSUM,         // dst <- sum_z slabs[z][row][0:N]
       COP_ACTBWD,          // dst <- src * act'(y) * dropout_scale   (y from yslot)
       COP_STORE,           // no compute: only the post-op global store(s) of dst[:, 0:N]
       COP_REPARAM,         // VAE (vae.py:115-118): src = [mu | logvar] (2N cols) -> dst[:, 0:N) = mu + eps * exp(logvar / 2);
                            //   eps: W (global [rows][ldw], injected) or the counter generator; always written to aux_ptr
       COP_REPARAM_BWD,
       COP_DISC_HEAD,       // the discriminator's 1-unit output layer in one op: logit = src[:, 0:K) . W[0:K) (K-1 = bias input), D = sigmoid,
                            //   adversarial loss + dL/dlogit as COP_ADV (aux, row_split, scale; dL/dlogit also -> aux_ptr[row * aux_ld]),
                            //   dst[:, 0:N) = dL/dlogit * W[0:N) through the ACTBWD epilogue (yslot, d)
       COP_PRIOR };         // disc_step input rows: r < row_split: z_real = prior sample (aux = kind; aux_ptr = injected draws
                            //   [row_split][aux_ld] instead) * scale (aae.py:716-718); r >= row_split: W[r] (z_fake)   // src = dL/dz (N/2 cols), yslot = [mu | logvar], eps from aux_ptr -> dst = [dL/dmu | dL/dlogvar] (N cols)
                            //   incl. the KL term's own gradient (vae.py:141-145), whose value goes to the loss slot

enum { CEPI_NONE = 0, CEPI_DROPACT = 1, CEPI_ACTBWD = 2, CEPI_SIGMOID = 3 };

struct ChainOp {
    int kind, src, dst, K, N;
    const float* W; int ldw;            // weights / global source (COP_LOAD, COP_SLABSUM)
    const float* Wkn; int ldkn;         // the same layer's matrix in k-major form [K][N] (n contiguous) for chain4.h
    const float* W4; int ns4;           // ... and k4-interleaved [(K + 3) / 4][ns4][4] (device_common.h), or NULL
    const unsigned short* WX; int xpl;  // ... and split into three bf16 planes for chain16x3.h (device_common.h FX / DX; xpl = rows of a plane), or NULL
    int x16_next_lin[2];                // chain16x3.h: the next linear op behind this one a workgroup runs ([1]: one that runs the upper rows' prefix
                                        // ops too, [0]: one that skips them), or -1 - whose first weights it requests behind its own products
    int epi, yslot;                     // epilogue; slot holding y for ACTBWD / FINAL_BWD
    DropSpec d;
    float* out; int ldo; int out_row0;  // optional global store of dst[:, 0:N] at rows out_row0 + r
    float* out2; int ldo2;              // optional second copy (e.g. z kept for the backward pass)
    int one_col;                        // >= 0: set dst[:, one_col] = 1 (constant-1 column of augmented weights)
    int aux;                            // COP_ADV: mode; COP_SLABSUM: number of slabs; COP_FINAL_*: kind
    size_t stride;                      // COP_SLABSUM: slab stride (floats)
    float scale;                        // COP_ADV: grad scale; COP_LOAD: multiplier
    int row_split;                      // COP_ADV mode 0: rows >= row_split are "fake"
    int dst_col0;                       // COP_LOAD: first destination column (appending a condition block)
    float* aux_ptr; int aux_ld;         // COP_REPARAM*: the eps buffer [rows][aux_ld] in global memory
    int grow0;                          // COP_PRIOR: global-batch row of this rank's row 0 (the prior draw is keyed by global row)
    int row_lo;                         // chain4.h: the op runs only in workgroups whose first row is >= row_lo (a program prefix
                                        // for the upper rows: Enc_eval in front of the discriminator program, aae_abi.hip)
    int fake_slot;                      // COP_PRIOR, chain4.h: >= 0: rows >= row_split come from this slot instead of from W
    const float* y_glb; int y_ld;       // chain4.h, a linear op with the ACTBWD epilogue: y read from global memory [rows][y_ld] (requested
                                        // before the op's products) instead of from a slot that an op of its own loaded
    int acc_in;                         // COP_LINEAR, chain4.h: 1 = the products are added to what dst holds before the epilogue (the
                                        // second k-part of a layer whose input is wider than a slot: the decoder's [z | condition | 1])
};

struct ChainProgram {
    int nops, rows, act;
    uint64_t seed; const long long* step_ctr;
    float* loss_out; int loss_slot;     // COP_ADV accumulates -mean(log ...) here (atomicAdd of per-row terms)
    float* loss_terms;                  // != NULL: the per-row terms go to loss_terms[row] instead (plain stores; the weight-gradient
                                        // launch behind the program sums them in a fixed order, DwGroup::loss).  r5: 1 000-2 000 atomic
                                        // adds on ONE address (a 1 000-row batch) are ~12 ns each at the memory side, and the wave's next
                                        // weight loads return behind its atomic (one in-order vmcnt): 8 us per program, losses not bit-stable
    int dbg;                            // timing-only ablation (AAE_CHAIN_SKIP), 0 in production
    unsigned long long* ts;             // debug (AAE_CHAIN_TS): per-op 100 MHz timestamps of workgroup 0, else NULL
    BucketJob bk;                       // enabled: one extra workgroup builds the fused decoder's tile buckets
    int x16_first_lin[2], x16_row_lo;   // chain16x3.h: the program's first linear op per workgroup class, the prefix ops' row_lo (or 1 << 30)
    ChainOp ops[kCMaxOps];
    int kslices;                        // chain4.h: 1 = every linear op in the k-slice form (wide batches; AAE_CHAIN_KSLICES).  LAST: a field in front
                                        // of ops[] moves every op descriptor's offset in the kernel-argument segment (and with it the width of the
                                        // scalar loads that fetch it: +0.3 us per launch of every chain kernel, measured)
};

// Workgroup barrier that orders LDS traffic only: global stores of finished activations (consumed by
// LATER kernels, never re-read inside this one) and weight loads in flight stay outstanding across
// it, where __syncthreads() would drain them (s_waitcnt vmcnt(0)) once per op.
__device__ __forceinline__ void chain_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Everything an epilogue needs from the op descriptor, read ONCE per op into scalar registers.  The
// descriptor lives in the kernel-argument segment; hipcc treats such loads as free to repeat, and with
// this kernel's scalar-register pressure it re-issued ~8 dependent s_load + s_waitcnt per ELEMENT
// (~0.5 us per 16x16 block).  The empty asm makes each value opaque, i.e. it has to stay in a register.
template <class T> __device__ __forceinline__ T chain_pin(T v) { asm volatile("" : "+s"(v)); return v; }

struct EpiCtx {
    int epi, act, den, drng, split, width, goa, gob;
    uint32_t thr;
    uint64_t dkey;
    float mk, ak, ad;
    const uint8_t* ma; const uint8_t* mb;
    const float* y;                      // slot holding the forward activation (ACTBWD)
};

__device__ __forceinline__ EpiCtx chain_epi_ctx(int epi, const ChainOp& op, const ChainProgram& P, uint64_t key,
                                                const float* slots) {
    // ALL fields requested first, then pinned by two statements: one s_waitcnt for the batch.  (Pinned one by one - r1-r3 -
    // every field was a scalar-load round trip of its own, ~200 clocks each from the kernel-argument segment:
    // tools/debug/ubench/sload_latency.hip.)
    EpiCtx c;
    c.epi = epi; c.act = P.act;
    c.den = op.d.enabled; c.drng = op.d.device_rng;
    c.split = op.d.split_row; c.width = op.d.width;
    c.goa = op.d.goff_a; c.gob = op.d.goff_b;
    c.thr = op.d.keep_threshold;
    uint64_t sid = op.d.stream_id;
    c.mk = op.d.mul_keep; c.ak = op.d.add_keep; c.ad = op.d.add_drop;
    c.ma = op.d.mask_a; c.mb = op.d.mask_b;
    int ys = op.yslot;
    asm volatile("" : "+s"(c.epi), "+s"(c.act), "+s"(c.den), "+s"(c.drng), "+s"(c.split), "+s"(c.width), "+s"(c.goa), "+s"(c.gob), "+s"(c.thr));
    asm volatile("" : "+s"(sid), "+s"(c.mk), "+s"(c.ak), "+s"(c.ad), "+s"(c.ma), "+s"(c.mb), "+s"(ys));
    c.dkey = key ^ (sid * 0xA0761D6478BD642Full);
    c.y = slots + ys * kCR * kCL;
    return c;
}

__device__ __forceinline__ int chain_keep(const EpiCtx& c, int row, int col) {      // drop_keep() on the pinned copy
    if (c.drng) return hash_cell(c.dkey, (uint32_t)(row + (row < c.split ? c.goa : c.gob)), (uint32_t)col) >= c.thr;
    // (the pinned copies lost their address space: say "global" again, or these become flat loads)
    typedef const __attribute__((address_space(1))) uint8_t* gmask_t;
    gmask_t m = (gmask_t)(row < c.split ? c.ma : c.mb);
    if (!m) return 1;
    const int r = row < c.split ? row : row - c.split;
    return m[(size_t)r * c.width + col] != 0;
}

template <bool NM = false>      // (NM: with the non-monotone activations' derivative, device_common.h)
__device__ __forceinline__ float chain_epi(const EpiCtx& c, int grow, int lrow, int col, float v) {
    if (c.epi == CEPI_DROPACT) {
        if (c.den) v = chain_keep(c, grow, col) ? v * c.mk + c.ak : c.ad;
        return act_fwd<NM>(c.act, v);
    }
    if (c.epi == CEPI_ACTBWD) {
        v *= act_grad_from_y<NM>(c.act, c.y[lrow * kCL + col]);
        if (c.den) v *= chain_keep(c, grow, col) ? c.mk : 0.f;
        return v;
    }
    if (c.epi == CEPI_SIGMOID) return sigmoidf_(v);
    return v;
}

// Element-wise work on a 16-row block uses a fixed thread -> (row, float4 column) map:
// row = tid >> kRS, column group c4 = (tid & (kTPR-1)) + kTPR*j (j < kJ): 64 float4 = 256 columns, no
// integer division, and both of a thread's global loads are in flight before the first use (a
// scalar element loop is a chain of dependent L2 round trips, ~1 us each).
__device__ __forceinline__ void chain_load_block(const float* __restrict__ g, int ld, int row0, int nrows, int N,
                                                 float* dst, int dst_col0, float scale) {
    const int tid = threadIdx.x, lrow = tid >> kRS;
    const bool vec = (ld & 3) == 0 && (dst_col0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    const int rowc = min(lrow, max(nrows, 1) - 1);
    const int nf4 = (N + 3) >> 2;
    if (vec) {
        float4 v[2];
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            const int c4 = min((tid & (kTPR - 1)) + kTPR * j, nf4 - 1);
            v[j] = *reinterpret_cast<const float4*>(g + (size_t)(row0 + rowc) * ld + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            const int c4 = (tid & (kTPR - 1)) + kTPR * j;
            if (c4 * 4 + dst_col0 < kCL) {                  // also clears the tail of the slot row
                const bool ok = lrow < nrows;
                float4 o;
                o.x = (ok && c4 * 4 + 0 < N) ? v[j].x * scale : 0.f;
                o.y = (ok && c4 * 4 + 1 < N) ? v[j].y * scale : 0.f;
                o.z = (ok && c4 * 4 + 2 < N) ? v[j].z * scale : 0.f;
                o.w = (ok && c4 * 4 + 3 < N) ? v[j].w * scale : 0.f;
                *reinterpret_cast<float4*>(dst + lrow * kCL + dst_col0 + c4 * 4) = o;
            }
        }
    } else {
        for (int col = tid & (kTPR - 1); col + dst_col0 < kCL; col += kTPR)
            dst[lrow * kCL + dst_col0 + col] =
                (lrow < nrows && col < N) ? g[(size_t)(row0 + lrow) * ld + col] * scale : 0.f;
    }
}

// LDS slot [16][0:N) -> global rows
__device__ __forceinline__ void chain_store_block(const float* src, int nrows, int N, float* __restrict__ g, int ld,
                                                  int row0) {
    const int tid = threadIdx.x, lrow = tid >> kRS;
    if (lrow >= nrows) return;
    if ((ld & 3) == 0 && (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0)) {
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            const int c4 = (tid & (kTPR - 1)) + kTPR * j;
            if (c4 * 4 < N)
                *reinterpret_cast<float4*>(g + (size_t)(row0 + lrow) * ld + c4 * 4) =
                    *reinterpret_cast<const float4*>(src + lrow * kCL + c4 * 4);
        }
    } else {
        for (int col = tid & (kTPR - 1); col < N; col += kTPR) g[(size_t)(row0 + lrow) * ld + col] = src[lrow * kCL + col];
    }
}

// ---- one layer's MFMAs for the (up to) two 16-column blocks of a wave ---------------------------
// All weight loads of the layer are issued before its first MFMA (the arrays are fully unrolled,
// i.e. registers): a layer costs one L2 round trip, not one per k-group.  Loads are unconditional
// from clamped addresses (a lane-conditional load makes hipcc wait for every load separately).
// Forward, W[n][k] k-contiguous.  k-permutation: lane (fr, fk) holds k = kc + 4*fk + j in element j
// of one float4 for BOTH operands, so 16 k need one 16-byte load per operand and feed 4 MFMAs.
// Addressing: every load is  scalar base (op.W + uniform chunk offset)  +  one 32-bit per-lane byte
// offset that is the same for all chunks, i.e. the SGPR-base form of global_load - no 64-bit VALU
// address per load and no VGPR pair to hold it.
template <int MC, bool TWO>
__device__ __forceinline__ void chain_linear_fwd(const ChainOp& op, const float* src, int b0, int b1, int fr, int fk,
                                                 f32x4 (&acc)[2]) {
    const int ldw = op.ldw, N = op.N;
    const int kch = (op.K + 15) >> 4;
    const uint32_t o0 = (uint32_t)(min(b0 * 16 + fr, N - 1) * ldw + 4 * fk) * 4u;
    const uint32_t o1 = (uint32_t)(min(b1 * 16 + fr, N - 1) * ldw + 4 * fk) * 4u;
    const float* a = src + fr * kCL + 4 * fk;
    float4 y0[MC], y1[TWO ? MC : 1];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const char* base = reinterpret_cast<const char*>(op.W + min(c, kch - 1) * 16);     // uniform
        y0[c] = *reinterpret_cast<const float4*>(base + o0);
        if (TWO) y1[c] = *reinterpret_cast<const float4*>(base + o1);
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        if (c < kch) {
            const float4 xa = *reinterpret_cast<const float4*>(a + c * 16);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, y0[c].x, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, y0[c].y, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, y0[c].z, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, y0[c].w, acc[0], 0, 0, 0);
            if (TWO) {
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, y1[c].x, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, y1[c].y, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, y1[c].z, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, y1[c].w, acc[1], 0, 0, 0);
            }
        }
    }
}

// dX, W[k][n] n-contiguous: one dword per lane and k (16 lanes = 64 contiguous bytes).  The same k-permutation
// as forward - lane (fr, fk) takes k = 16c + 4*fk + e - so the A operand comes from LDS as one 16-byte read per
// 4 MFMAs; A is masked to zero for k >= K (the slot may hold stale columns there), the weight rows up to
// 16 * ceil(K / 16) - 1 exist because the arena keeps 16 zero rows behind every weight matrix the chain reads
// (layout(): pad_rows), so the row index needs no clamp.
template <int MC, bool TWO>
__device__ __forceinline__ void chain_linear_dx(const ChainOp& op, const float* src, int b0, int b1, int fr, int fk,
                                                f32x4 (&acc)[2]) {
    const int ldw = op.ldw, N = op.N, K = op.K;
    const int kch = (K + 15) >> 4;
    const uint32_t o0 = (uint32_t)(4 * fk * ldw + min(b0 * 16 + fr, N - 1)) * 4u;
    const uint32_t o1 = (uint32_t)(4 * fk * ldw + min(b1 * 16 + fr, N - 1)) * 4u;
    const float* a = src + fr * kCL + 4 * fk;
    float y0[MC * 4], y1[TWO ? MC * 4 : 1];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const char* base = reinterpret_cast<const char*>(op.W + (size_t)(min(c, kch - 1) * 16 + e) * ldw);   // uniform
            y0[c * 4 + e] = *reinterpret_cast<const float*>(base + o0);
            if (TWO) y1[c * 4 + e] = *reinterpret_cast<const float*>(base + o1);
        }
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        if (c < kch) {
            float4 xa = *reinterpret_cast<const float4*>(a + c * 16);
            const int k0 = c * 16 + 4 * fk;
            if (k0 + 0 >= K) xa.x = 0.f;
            if (k0 + 1 >= K) xa.y = 0.f;
            if (k0 + 2 >= K) xa.z = 0.f;
            if (k0 + 3 >= K) xa.w = 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, y0[c * 4 + 0], acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, y0[c * 4 + 1], acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, y0[c * 4 + 2], acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, y0[c * 4 + 3], acc[0], 0, 0, 0);
            if (TWO) {
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, y1[c * 4 + 0], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, y1[c * 4 + 1], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, y1[c * 4 + 2], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, y1[c * 4 + 3], acc[1], 0, 0, 0);
            }
        }
    }
}

// bf16 mode of the build (BASELINE config C2): the forward-pattern layer on v_mfma_f32_16x16x32_bf16.  Same operands
// (fp32 activations in the LDS slot, fp32 weights streamed from L2), rounded to bf16 on the way into the matrix core
// (v_cvt_pk_bf16_f32, ties to even), fp32 accumulation: 7 MFMAs of 16 cycles instead of 52 of 32 for a 201-wide layer.
// Lane (fr, fk) takes k = 32c + 4 fk + {0..3} and 32c + 16 + 4 fk + {0..3} of chunk c for both operands (two float4
// each).  The last chunk may run past the weight row (the arena keeps 16 zero rows behind every matrix: readable) and
// past the slot row (masked: the slot's columns beyond the layer's K are zero, but the next row's are not).
typedef __bf16 chain_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ chain_bf16x8 chain_pack8(const float4& lo, const float4& hi) {
    const gemm_u32x4 v = {gemm_pack_bf16(lo.x, lo.y), gemm_pack_bf16(lo.z, lo.w), gemm_pack_bf16(hi.x, hi.y), gemm_pack_bf16(hi.z, hi.w)};
    return __builtin_bit_cast(chain_bf16x8, v);
}
template <int MC>
__device__ __forceinline__ void chain_linear_fwd_bf16(const ChainOp& op, const float* src, int b0, int fr, int fk, f32x4& acc) {
    const int ldw = op.ldw, N = op.N;
    const int kch = (op.K + 31) >> 5;
    const uint32_t o0 = (uint32_t)(min(b0 * 16 + fr, N - 1) * ldw + 4 * fk) * 4u;
    const float* a = src + fr * kCL;
    float4 w0[MC], w1[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const char* base = reinterpret_cast<const char*>(op.W + min(c, kch - 1) * 32);     // uniform
        w0[c] = *reinterpret_cast<const float4*>(base + o0);
        w1[c] = *reinterpret_cast<const float4*>(base + o0 + 64);
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        if (c < kch) {
            const int k0 = 32 * c + 4 * fk, k1 = k0 + 16;
            float4 x0 = *reinterpret_cast<const float4*>(a + min(k0, kCL - 4));
            float4 x1 = *reinterpret_cast<const float4*>(a + min(k1, kCL - 4));
            if (k0 > kCL - 4) x0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k1 > kCL - 4) x1 = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(chain_pack8(x0, x1), chain_pack8(w0[c], w1[c]), acc, 0, 0, 0);
        }
    }
}

template <bool BF, bool NM = false>
__global__ __launch_bounds__(kCT) void chain_kernel(ChainProgram P) {
    extern __shared__ __attribute__((aligned(16))) float slots[];     // [kCSlots][16][kCL]
    if (P.bk.enabled && blockIdx.x == gridDim.x - 1) {                // (uniform) the piggy-backed bucket builder
        tile_bucket_body(P.bk.bv, P.bk.ntiles, P.bk.tstart, P.bk.eb, P.bk.en, P.bk.ev, reinterpret_cast<int*>(slots));
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int r0 = blockIdx.x * kCR;
    const int nrows = min(kCR, P.rows - r0);
    const uint64_t key = rng_key(P.seed, (uint64_t)*P.step_ctr, 0);

    for (int oi = 0; oi < P.nops; ++oi) {
        const ChainOp& op = P.ops[oi];
        if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[oi] = wall_clock64();
        float* dst = slots + op.dst * kCR * kCL;
        const float* src = slots + op.src * kCR * kCL;
        const int kind = op.kind;

        if (kind == COP_LINEAR || kind == COP_LINEAR_DX) {
            // output blocks of 16 columns: wave w takes blocks w, w+8 (N <= 208 -> <= 13 blocks)
            const int nblk = (op.N + 15) >> 4;
            f32x4 acc[2];
            acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (acc[1] unused when a wave owns one block)
            const int b0 = min(wave, nblk - 1), b1 = min(wave + kCW, nblk - 1);
            const bool two = kCQ > 1 && nblk > kCW;
            const EpiCtx ec = chain_epi_ctx(op.epi, op, P, key, slots);
            const int epiN = chain_pin(op.N);
            if (!(P.dbg & 1) && wave < nblk) {          // waves without a block (narrow layers) only meet the barriers
                // Load volume follows the layer: short-K layers (K <= 64) take the 4-chunk /
                // chunk variant and layers of <= 128 columns skip the second block - the weight stream of a
                // layer is bound by the CU's 64 B/clk vector-memory path, so unneeded (clamped) loads cost time.
                if (kind == COP_LINEAR && BF) {
                    static_assert(kCQ == 1, "one block per wave");
                    if (((op.K + 31) >> 5) <= 2) chain_linear_fwd_bf16<2>(op, src, b0, fr, fk, acc[0]);
                    else chain_linear_fwd_bf16<7>(op, src, b0, fr, fk, acc[0]);
                } else if (kind == COP_LINEAR) {
                    const int kch = (op.K + 15) >> 4;
                    if (kch <= 4) { if (two) chain_linear_fwd<4, true>(op, src, b0, b1, fr, fk, acc); else chain_linear_fwd<4, false>(op, src, b0, b1, fr, fk, acc); }
                    else          { if (two) chain_linear_fwd<13, true>(op, src, b0, b1, fr, fk, acc); else chain_linear_fwd<13, false>(op, src, b0, b1, fr, fk, acc); }
                } else {
                    const int kch = (op.K + 15) >> 4;
                    if (kch <= 4) { if (two) chain_linear_dx<4, true>(op, src, b0, b1, fr, fk, acc); else chain_linear_dx<4, false>(op, src, b0, b1, fr, fk, acc); }
                    else          { if (two) chain_linear_dx<13, true>(op, src, b0, b1, fr, fk, acc); else chain_linear_dx<13, false>(op, src, b0, b1, fr, fk, acc); }
                }
            }
            chain_barrier();      // dst may alias a slot other waves were still reading (src != dst is required)
#pragma unroll
            for (int q = 0; q < kCQ; ++q) {
                const int blk = wave + kCW * q;
                if (blk < nblk) {
                    const int col = blk * 16 + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lrow = fk * 4 + r;
                        float v = 0.f;
                        if (col < epiN && lrow < nrows) v = chain_epi<NM>(ec, r0 + lrow, lrow, col, acc[q][r]);
                        dst[lrow * kCL + col] = v;
                    }
                }
            }
            // columns [16*nblk, kCL) must read as zero for the next layer's k-padding
            {
                const int c0 = nblk * 16, lrow = tid >> kRS;
                for (int col = c0 + (tid & (kTPR - 1)); col < kCL; col += kTPR) dst[lrow * kCL + col] = 0.f;
            }
        } else if (kind == COP_LOAD) {
            chain_load_block(op.W, op.ldw, op.out_row0 + r0, nrows, op.N, dst, op.dst_col0, op.scale);
        } else if (kind == COP_SLABSUM) {
            // sum of op.aux (<= 16) partial slabs: every slab load of a thread is in flight at once
            const int lrow = tid >> kRS, rowc = min(lrow, max(nrows, 1) - 1), nf4 = (op.N + 3) >> 2;
#pragma unroll
            for (int j = 0; j < kJ; ++j) {
                const int c4 = (tid & (kTPR - 1)) + kTPR * j, c4c = min(c4, nf4 - 1);
                float4 v[16];
#pragma unroll
                for (int z = 0; z < 16; ++z)
                    v[z] = *reinterpret_cast<const float4*>(op.W + (size_t)min(z, op.aux - 1) * op.stride +
                                                            (size_t)(r0 + rowc) * op.ldw + c4c * 4);
                float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int z = 0; z < 16; ++z)
                    if (z < op.aux) { acc4.x += v[z].x; acc4.y += v[z].y; acc4.z += v[z].z; acc4.w += v[z].w; }
                if (op.epi == CEPI_ACTBWD) {
                    // fused: times act'(y) and the dropout scale; y = the forward activation straight from global memory
                    // (aux_ptr [rows][aux_ld]), instead of a LOAD op and an ACTBWD op behind this one
                    const EpiCtx sec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, slots);
                    const float4 y4 = *reinterpret_cast<const float4*>(op.aux_ptr + (size_t)(r0 + rowc) * op.aux_ld + c4c * 4);
                    const int gr = r0 + lrow;
                    // (the mask / generator lookup only for cells of the block that exist: an injected mask tensor
                    // ends with the batch's last row)
                    const bool den = sec.den && lrow < nrows;
                    const float k0 = (den && c4 * 4 + 0 < op.N) ? (chain_keep(sec, gr, c4 * 4 + 0) ? sec.mk : 0.f) : 1.f;
                    const float k1 = (den && c4 * 4 + 1 < op.N) ? (chain_keep(sec, gr, c4 * 4 + 1) ? sec.mk : 0.f) : 1.f;
                    const float k2 = (den && c4 * 4 + 2 < op.N) ? (chain_keep(sec, gr, c4 * 4 + 2) ? sec.mk : 0.f) : 1.f;
                    const float k3 = (den && c4 * 4 + 3 < op.N) ? (chain_keep(sec, gr, c4 * 4 + 3) ? sec.mk : 0.f) : 1.f;
                    acc4.x *= act_grad_from_y<NM>(sec.act, y4.x) * k0; acc4.y *= act_grad_from_y<NM>(sec.act, y4.y) * k1;
                    acc4.z *= act_grad_from_y<NM>(sec.act, y4.z) * k2; acc4.w *= act_grad_from_y<NM>(sec.act, y4.w) * k3;
                }
                const bool ok = lrow < nrows;
                if (!ok || c4 * 4 + 0 >= op.N) acc4.x = 0.f;
                if (!ok || c4 * 4 + 1 >= op.N) acc4.y = 0.f;
                if (!ok || c4 * 4 + 2 >= op.N) acc4.z = 0.f;
                if (!ok || c4 * 4 + 3 >= op.N) acc4.w = 0.f;
                if (c4 * 4 < kCL) *reinterpret_cast<float4*>(dst + lrow * kCL + c4 * 4) = acc4;
            }
        } else if (kind == COP_DROPACT || kind == COP_ACTBWD) {
            const EpiCtx ec = chain_epi_ctx(kind == COP_DROPACT ? CEPI_DROPACT : CEPI_ACTBWD, op, P, key, slots);
            const int epiN = chain_pin(op.N);
            const int lrow = tid >> kRS;
            for (int col = tid & (kTPR - 1); col < kCL; col += kTPR)
                dst[lrow * kCL + col] = (lrow < nrows && col < epiN)
                    ? chain_epi<NM>(ec, r0 + lrow, lrow, col, src[lrow * kCL + col]) : 0.f;
        } else if (kind == COP_FINAL_FWD) {
            // one wave per pair of rows; softmax / sigmoid / identity over N columns, in place on dst
            for (int lrow = wave; lrow < kCR; lrow += kCW) {
                float* zr = dst + lrow * kCL;
                if (op.aux == 1) {
                    float mx = -INFINITY;
                    for (int j = lane; j < op.N; j += 64) mx = fmaxf(mx, zr[j]);
                    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    float sum = 0.f;
                    for (int j = lane; j < op.N; j += 64) sum += expf(zr[j] - mx);
                    sum = wave_sum(sum);
                    for (int j = lane; j < op.N; j += 64) zr[j] = expf(zr[j] - mx) / sum;
                } else if (op.aux == 2) {
                    for (int j = lane; j < op.N; j += 64) zr[j] = sigmoidf_(zr[j]);
                }
            }
        } else if (kind == COP_FINAL_BWD) {
            const float* zs = slots + op.yslot * kCR * kCL;
            for (int lrow = wave; lrow < kCR; lrow += kCW) {
                const float* zr = zs + lrow * kCL;
                const float* gr = src + lrow * kCL;
                float* o = dst + lrow * kCL;
                if (op.aux == 1) {
                    float dot = 0.f;
                    for (int j = lane; j < op.N; j += 64) dot += gr[j] * zr[j];
                    dot = wave_sum(dot);
                    for (int j = lane; j < op.N; j += 64) o[j] = zr[j] * (gr[j] - dot);
                } else if (op.aux == 2) {
                    for (int j = lane; j < op.N; j += 64) o[j] = gr[j] * zr[j] * (1.f - zr[j]);
                } else {
                    for (int j = lane; j < op.N; j += 64) o[j] = gr[j];
                }
                // a following forward-pattern layer (dX on a transposed weight copy) reads whole 16-column k-chunks
                for (int j = op.N + lane; j < kCL; j += 64) o[j] = 0.f;
            }
        } else if (kind == COP_PRIOR) {
            const int lrow = tid >> kRS, grow = r0 + lrow, n = op.N;
            const uint64_t k = key ^ (100ull * 0xA0761D6478BD642Full);
            for (int col = tid & (kTPR - 1); col < kCL; col += kTPR) {
                float v = 0.f;
                if (lrow < nrows && col < n) {
                    if (grow >= op.row_split) v = op.W[(size_t)grow * op.ldw + col];
                    else if (op.aux_ptr) v = op.aux_ptr[(size_t)grow * op.aux_ld + col] * op.scale;
                    else if (op.aux == 0) {          // gauss: Box-Muller on two words of the counter generator
                        const uint32_t u1 = hash_cell(k, (uint32_t)(grow + op.grow0), (uint32_t)(2 * col));
                        const uint32_t u2 = hash_cell(k, (uint32_t)(grow + op.grow0), (uint32_t)(2 * col + 1));
                        const float f1 = ((float)(u1 >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
                        const float f2 = (float)(u2 >> 8) * (1.0f / 16777216.0f);
                        v = sqrtf(-2.0f * logf(f1)) * cosf(6.283185307179586f * f2) * op.scale;
                    } else if (op.aux == 1) {        // categorical: one-hot of a uniform class per row
                        const uint32_t u = hash_cell(k, (uint32_t)(grow + op.grow0), 0xFFFFFFFFu);
                        v = ((int)(u % (uint32_t)n) == col) ? op.scale : 0.f;
                    }                                // bernoulli: the reference's randint(0, 1) is always 0 (aae.py:86-88)
                }
                dst[lrow * kCL + col] = v;
            }
        } else if (kind == COP_REPARAM) {
            const int lrow = tid >> kRS, n = op.N;
            for (int col = tid & (kTPR - 1); col < kCL; col += kTPR) {
                float zv = 0.f;
                if (lrow < nrows && col < n) {
                    float eps;
                    if (op.W) eps = op.W[(size_t)(r0 + lrow) * op.ldw + col];
                    else {      // Box-Muller on two words of the counter generator (stream id = op.aux)
                        const uint64_t k = key ^ ((uint64_t)(uint32_t)op.aux * 0xA0761D6478BD642Full);
                        const uint32_t u1 = hash_cell(k, (uint32_t)(r0 + lrow), (uint32_t)(2 * col));
                        const uint32_t u2 = hash_cell(k, (uint32_t)(r0 + lrow), (uint32_t)(2 * col + 1));
                        const float f1 = ((float)(u1 >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
                        const float f2 = (float)(u2 >> 8) * (1.0f / 16777216.0f);
                        eps = sqrtf(-2.0f * logf(f1)) * cosf(6.283185307179586f * f2);
                    }
                    op.aux_ptr[(size_t)(r0 + lrow) * op.aux_ld + col] = eps;
                    const float mu = src[lrow * kCL + col], lv = src[lrow * kCL + n + col];
                    zv = mu + eps * expf(0.5f * lv);
                }
                dst[lrow * kCL + col] = zv;
            }
        } else if (kind == COP_REPARAM_BWD) {
            const float* ml = slots + op.yslot * kCR * kCL;
            const int lrow = tid >> kRS, n = op.N >> 1;          // op.N = 2n columns are produced (and stored)
            float kl = 0.f;
            for (int col = tid & (kTPR - 1); col < kCL; col += kTPR) {
                float o = 0.f;
                if (lrow < nrows && col < 2 * n) {
                    const int j = col < n ? col : col - n;
                    const float mu = ml[lrow * kCL + j], lv = ml[lrow * kCL + n + j];
                    const float gz = src[lrow * kCL + j];
                    const float ev = expf(lv);
                    if (col < n) {
                        o = gz + op.scale * mu;
                        kl += -0.5f * (1.f + lv - mu * mu - ev);
                    } else {
                        const float eps = op.aux_ptr[(size_t)(r0 + lrow) * op.aux_ld + j];
                        o = gz * eps * 0.5f * expf(0.5f * lv) + op.scale * 0.5f * (ev - 1.f);
                    }
                }
                dst[lrow * kCL + col] = o;
            }
            kl = wave_sum(kl);
            if (lane == 0 && kl != 0.f) atomicAdd(P.loss_out + P.loss_slot, kl);
        } else if (kind == COP_DISC_HEAD) {
            // one wave per row of the block (tid >> 6 == local row): three ops' worth of weight round trips and
            // barriers for a layer whose arithmetic is 16 dot products and 16 scaled copies of one weight row
            static_assert(kCT == 1024 && kTPR == 64, "COP_DISC_HEAD maps one wavefront to one row");
            const EpiCtx ec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, slots);
            const int lrow = tid >> kRS, grow = r0 + lrow, t = tid & (kTPR - 1);
            const int Kk = chain_pin(op.K), Nn = chain_pin(op.N);
            float wv[4], part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = t + 64 * j;
                wv[j] = op.W[min(k, Kk - 1)];
                if (k < Kk) part += src[lrow * kCL + k] * wv[j];
            }
            const float logit = wave_sum(part);
            float gv = 0.f;
            if (lrow < nrows) {
                const float dv = sigmoidf_(logit);
                const int Bsplit = op.row_split;
                const float invB = 1.f / (float)Bsplit;
                float l, gg;
                if (op.aux == 0 && grow >= Bsplit) { l = logf(1.f - dv + kTiny); gg = invB / (1.f - dv + kTiny); }
                else { l = logf(dv + kTiny); gg = -invB / (dv + kTiny); }
                gv = gg * dv * (1.f - dv) * op.scale;
                if (t == 0) {
                    if (P.loss_terms) P.loss_terms[grow] = -l * invB;
                    else atomicAdd(P.loss_out + P.loss_slot, -l * invB);
                    if (op.aux_ptr) op.aux_ptr[(size_t)grow * op.aux_ld] = gv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = t + 64 * j;
                if (k < kCL) dst[lrow * kCL + k] = (lrow < nrows && k < Nn) ? chain_epi<NM>(ec, grow, lrow, k, gv * wv[j]) : 0.f;
            }
        } else if (kind == COP_ADV) {
            // src col 0 = D(x) of each row.  mode 0: rows < row_split real, others fake; mode 1: all fake(gen)
            if (tid < kCR) {
                const int lrow = tid, g = r0 + lrow;
                float gv = 0.f;
                if (lrow < nrows) {
                    const float dv = src[lrow * kCL];
                    const int B = op.row_split;
                    const float invB = 1.f / (float)B;
                    float l, gg;
                    if (op.aux == 0 && g >= B) { l = logf(1.f - dv + kTiny); gg = invB / (1.f - dv + kTiny); }
                    else { l = logf(dv + kTiny); gg = -invB / (dv + kTiny); }
                    gv = gg * dv * (1.f - dv) * op.scale;
                    if (P.loss_terms) P.loss_terms[g] = -l * invB;
                    else atomicAdd(P.loss_out + P.loss_slot, -l * invB);
                }
                dst[lrow * kCL] = gv;
            }
        }
        chain_barrier();
        if (op.one_col >= 0) {      // written by the same lanes for every op kind: rows of the block
            if (tid < kCR) dst[tid * kCL + op.one_col] = tid < nrows ? 1.f : 0.f;
            chain_barrier();
        }
        if (op.out) chain_store_block(dst, nrows, op.N, op.out, op.ldo, op.out_row0 + r0);
        if (op.out2) chain_store_block(dst, nrows, op.N, op.out2, op.ldo2, r0);
    }
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[P.nops] = wall_clock64();
}

// -----------------------------------------------------------------------------------------------
// grouped weight-gradient GEMMs: dW[out][in+1] = G[rows][out]^T * X[rows][in+1] for up to 4 layers
// in one launch, 32x32 tiles, fused optimiser update (or gradient export).
// -----------------------------------------------------------------------------------------------
struct DwJob {
    const float* G; int ldg; const float* X; int ldx; int rows;   // K = rows
    int M, N;                                                      // out, in+1
    float* p; float* m; float* v; float* grad; int ld;             // grad != NULL: export
    W4Copies w4;                                                   // k4-interleaved copies of p kept in step (f4 == NULL: none)
    const OptScalars* sc;
    int tile0;                                                     // first linear tile id of this job
    int tiles_n;                                                   // tiles along N
};
// The first encoder layer shares the launch: its bias' column sums + optimiser (one workgroup per 64 columns) and its
// row-sparse weight gradient + optimiser over the batch's distinct items (w1_update.h: one wavefront per item) depend on
// the same chain outputs as the dense jobs and on nothing else - as launches of their own behind this one they were 8-24 us
// of the step's critical path, twice per step.
struct W1Job {
    int enabled; const float* ga1; int ld, h, rows;
    float* bp; float* bm; float* bv1; float* bgrad; const OptScalars* sc;   // blocks [blk0, blk0 + ncol)
    int blk0, ncol;
    int nitem;            // > 0: blocks [blk0 + ncol, blk0 + ncol + nitem) run the item update (else a launch of its own follows)
    int wave_form;        // 2: batch of one fused launch - four items per block, wave form up to 16 rows, the others by the whole block
                          // (w1_item_hybrid_body); 1: wide batch - the item blocks run the one-wave-per-item form (w1_update.h), items with more rows than
    int* hot; int* hot_count; int* hot_zero;    // kW1WaveRows go to the hot list (worked off by a launch behind this one); hot_zero: the NEXT use's counter
    W1Items items;
    int* cnt_out;         // != NULL: host-visible word that receives the step's distinct-item count (the column sums' first workgroup)
};
// the adversarial loss of the chain program in front of this launch: its per-row terms (ChainProgram::loss_terms) summed in a
// fixed order by one extra workgroup (the launch's last) -> *out
struct LossJob { int enabled; const float* terms; int n; float* out; };
struct DwGroup { int njobs; DwJob jobs[4]; W1Job w1; LossJob loss;
                 int ksplit;           // > 0: jobs of at least this many rows take the k-split form of a tile (grouped_dw_kernel)
                 unsigned long long* ts; int ts_cap; };    // debug: per-workgroup clocks
constexpr int kDwSmemBytes = 2 * 64 * (32 + 16) * 4;       // static LDS of grouped_dw_kernel (2 * BK * LDT floats)

// One slab's products of a wave's 16 x 16 block: BK / 4 matrix instructions on ONE accumulator, k ascending (the order - and
// so every bit of the result - of the plain loop).  Written out as a pipeline because hipcc, left alone, reads two operand
// pairs into the same four registers, waits for the LDS, multiplies, reads the next two ...: 8 LDS round trips in a row, 0.6 of
// the 0.88 us a slab took (tools/debug/ubench/dw_real.hip, in-kernel stamps).  Here the operands of four instructions are
// in flight while the four before them multiply (16 registers).
#ifndef DW_KSPLIT_DEPTH
#define DW_KSPLIT_DEPTH 8          // k-steps a wave of the k-split form requests ahead (6 / 8 / 12: 8 is the fastest, dw_real.hip)
#endif
template <int BK, int LDT>
__device__ __forceinline__ f32x4 dw_slab_products(const float* ap, const float* bp, f32x4 acc) {
#ifdef DW_PLAIN_PRODUCTS
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * LDT], bp[kk * LDT], acc, 0, 0, 0);
    return acc;
#else
    constexpr int Q = 4, NQ = BK / 4 / Q;                   // Q instructions per stage
    static_assert(BK % (4 * Q) == 0 && NQ >= 2, "dw_slab_products: whole stages");
    float a[2][Q], b[2][Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) { a[0][i] = ap[(4 * i) * LDT]; b[0][i] = bp[(4 * i) * LDT]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {
#pragma unroll
            for (int i = 0; i < Q; ++i) { a[(q + 1) & 1][i] = ap[(4 * ((q + 1) * Q + i)) * LDT]; b[(q + 1) & 1][i] = bp[(4 * ((q + 1) * Q + i)) * LDT]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < Q; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 1][i], b[q & 1][i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
#endif
}

#ifdef DW_TS      // (tools/debug/ubench/dw_real.hip: in-kernel stamps of tile 0 and of the first column-sum block)
__device__ unsigned long long dw_ts[96];
#define DW_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) dw_ts[i] = wall_clock64(); } while (0)
#define DW_STAMP_W1(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == grp.w1.blk0) dw_ts[i] = wall_clock64(); } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#define DW_STAMP_W1(i) do { } while (0)
#endif
__device__ __forceinline__ void grouped_dw_body(const DwGroup& grp, float* smem) {
    constexpr int TS = 32, BK = 64, LDT = TS + 16, LDC = TS + 4, NV = TS * BK / 1024;
    static_assert(sizeof(float) * 2 * BK * LDT == kDwSmemBytes, "kDwSmemBytes");
    static_assert(4 * TS * LDC <= 2 * BK * LDT, "the k-split form's four partial tiles");
    DW_STAMP(0);
    if (grp.loss.enabled && blockIdx.x == gridDim.x - 1) {           // (uniform) the loss terms' sum: thread t adds terms t, t + 256, ...
        float acc = 0.f;                                              // in index order, the 256 sums meet in a fixed tree
        for (int i = threadIdx.x; i < grp.loss.n; i += 256) acc += grp.loss.terms[i];
        smem[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) smem[threadIdx.x] += smem[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *grp.loss.out = smem[0];
        return;
    }
    if (grp.w1.enabled && (int)blockIdx.x >= grp.w1.blk0) {          // (uniform) first-layer workgroups
        const W1Job& w = grp.w1;
        const int id = (int)blockIdx.x - w.blk0;
        DW_STAMP_W1(64);
        if (id == 0 && threadIdx.x == 0 && w.cnt_out && w.nitem > 0) *w.cnt_out = *w.items.ucount;
        if (id < w.ncol) { colsum_adam_body(w.ga1, w.rows, w.h, w.ld, w.bp, w.bm, w.bv1, w.bgrad, w.sc, id, smem); DW_STAMP_W1(65); }
        else if (w.wave_form == 2) w1_item_hybrid_body(w.items, reinterpret_cast<unsigned*>(smem), id - w.ncol, w.nitem);
        else if (w.wave_form) {
            if (id == w.ncol && threadIdx.x == 0) *w.hot_zero = 0;
            w1_item_wave_body(w.items, w.hot, w.hot_count, reinterpret_cast<unsigned*>(smem), (id - w.ncol) * 4 + (int)(threadIdx.x >> 6), w.nitem * 4);
        } else w1_item_update_body(w.items, reinterpret_cast<unsigned*>(smem), id - w.ncol, w.nitem);
        return;
    }
    int j = 0;
    for (int q = 1; q < grp.njobs; ++q) if ((int)blockIdx.x >= grp.jobs[q].tile0) j = q;
    DW_STAMP(52);
    const DwJob& J = grp.jobs[j];
    const int t = blockIdx.x - J.tile0;
    const int m0 = (t / J.tiles_n) * TS, n0 = (t % J.tiles_n) * TS;
    DW_STAMP(53);
    float* As = smem; float* Bs = smem + BK * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 16, wn = (wave & 1) * 16;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int K = J.rows;
    const int mmax4 = ((J.M + 3) & ~3) - 4, nmax4 = ((J.N + 3) & ~3) - 4;
    // The slab loop is a chain of dependent round trips (load -> LDS -> products): the NEXT slab's rows are requested before
    // this slab's products, from clamped (always valid) addresses, unconditionally - a batch of 800-1000 rows is 13-16 slabs
    // (r3: one round trip each, 47 us per launch at 800 rows; the products and their order are unchanged).
    float4 ra[NV], rb[NV];
    auto request = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + 256 * q, kr = min(k0 + f / (TS / 4), K - 1), cq = (f % (TS / 4)) * 4;
            ra[q] = *reinterpret_cast<const float4*>(J.G + (size_t)kr * J.ldg + min(m0 + cq, mmax4));
            rb[q] = *reinterpret_cast<const float4*>(J.X + (size_t)kr * J.ldx + min(n0 + cq, nmax4));
        }
    };
    const bool ksplit = grp.ksplit > 0 && K >= grp.ksplit;        // (uniform)
    if (!ksplit) request(0);
    DW_STAMP(54);
    // the optimiser's operands of this thread's four cells (parameter, both moments) depend on nothing this launch
    // computes: requested here, consumed behind the product - one memory round trip less at the kernel's end
    const int erow = tid / (TS / 4), ecol = (tid % (TS / 4)) * 4;
    const int egm = m0 + erow, egn = n0 + ecol;
#ifdef DW_NO_OPT_PREFETCH
    const bool epre = false;
#else
    const bool epre = !J.grad && egm < J.M && egn + 3 < J.N;
#endif
    float4 pp0 = make_float4(0.f, 0.f, 0.f, 0.f), mm0 = pp0, vv0 = pp0;
    const OptScalars osc = *J.sc;
    if (epre) {
        const size_t off = (size_t)egm * J.ld + egn;
        pp0 = *reinterpret_cast<const float4*>(J.p + off);
        // (both moments whatever the optimiser - they are allocated either way: asking the optimiser's scalars first is one
        //  memory round trip in front of every tile's first products)
        mm0 = *reinterpret_cast<const float4*>(J.m + off); vv0 = *reinterpret_cast<const float4*>(J.v + off);
    }
    DW_STAMP(1);
    const int row = tid / (TS / 4), col = (tid % (TS / 4)) * 4;
    const int gm = m0 + row, gn = n0 + col;
    float4 g4;
    if (ksplit) {
        // WIDE batches (r5): the slab loop above is a chain of barrier - wait for the slab's rows - LDS - barrier - products steps,
        // 0.8-0.9 us per 64 rows however the products are scheduled (in-kernel stamps, tools/debug/ubench/dw_real.hip: 17 us per
        // launch at 800 rows, 31 at 2 000).  Here every WAVE multiplies the whole 32 x 32 tile over a quarter of the rows, its
        // operands straight from L2 into the matrix instructions' registers (lane (m, k) of a 16 x 16 x 4 instruction reads G[k][m]),
        // DW_KSPLIT_DEPTH k-steps ahead, no LDS and no barrier in the loop; the four partial tiles meet in LDS and are added in
        // wave order (a fixed order: a function of the row count alone).
        constexpr int D = DW_KSPLIT_DEPTH;
        const int fr = lane & 15, fk = lane >> 4;
        const int S = (K + 3) >> 2, Sw = (S + 3) >> 2, s0 = __builtin_amdgcn_readfirstlane(wave) * Sw, s1 = min(S, s0 + Sw);
        // lane fr of the two 16-row blocks holds tile rows 2 fr and 2 fr + 1 (columns alike): one 8-byte load per operand and
        // k-step, a wave's load = four whole 128-byte lines.  (One 4-byte load per block row - four half lines per load - ran at
        // 11 bytes per clock and CU: 0.15 us per k-step whatever the read-ahead depth.)
        const unsigned ca = (unsigned)min(m0 + 2 * fr, ((J.M + 1) & ~1) - 2), cb = (unsigned)min(n0 + 2 * fr, ((J.N + 1) & ~1) - 2);
        f32x4 c00 = (f32x4){0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
        float2 av[D], bv[D];
        auto ld = [&](int st, int i) {          // (rows beyond the batch: the last row again, multiplied by zeros)
            const unsigned r = (unsigned)min(4 * st + fk, K - 1);
            av[i] = *reinterpret_cast<const float2*>(J.G + (r * (unsigned)J.ldg + ca)); bv[i] = *reinterpret_cast<const float2*>(J.X + (r * (unsigned)J.ldx + cb));
        };
#pragma unroll
        for (int i = 0; i < D; ++i) { ld(s0 + i, i); __builtin_amdgcn_sched_barrier(0); }      // (in this order: they return in order)
        int st = s0;
        // steady passes: every step multiplied and every step requested lies inside the batch and inside this wave's share - no
        // clamps, no selects, one offset add per load.  (A wave issues its matrix instructions ~45 cycles apart and whatever it
        // issues between them comes on top: with the clamped addresses and the selects a k-step took 290 cycles, 180 of them
        // the four products.)
        const int sfull = min(s1, K >> 2);
        unsigned oa = (unsigned)(4 * (s0 + D) + fk) * (unsigned)J.ldg + ca, ob = (unsigned)(4 * (s0 + D) + fk) * (unsigned)J.ldx + cb;
        const unsigned da = 4u * (unsigned)J.ldg, db = 4u * (unsigned)J.ldx;
        for (; st + 2 * D <= sfull; st += D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].y, c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].x, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, c11, 0, 0, 0);
                av[i] = *reinterpret_cast<const float2*>(J.G + oa); bv[i] = *reinterpret_cast<const float2*>(J.X + ob);
                oa += da; ob += db;
                __builtin_amdgcn_sched_barrier(0);      // (left alone, hipcc gathers the loads in clumps and ends every pass with vmcnt(0))
            }
        }
        for (; st < s1; st += D) {              // the last one or two passes: the general form
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const bool ok = st + i < s1 && 4 * (st + i) + fk < K;
                const float a0 = ok ? av[i].x : 0.f, a1 = ok ? av[i].y : 0.f, b0 = bv[i].x, b1 = bv[i].y;
                c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c11, 0, 0, 0);
                ld(st + i + D, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DW_STAMP(44);
        float* Pw = smem + wave * (TS * LDC);                  // this wave's partial tile [32][LDC]
#pragma unroll
        for (int r = 0; r < 4; ++r) {                          // (c00[r]: tile row 2 i, column 2 j with i = 4 (lane >> 4) + r, j = lane & 15)
            const int pi = (lane >> 4) * 4 + r, pj = lane & 15;
            *reinterpret_cast<float2*>(&Pw[(2 * pi) * LDC + 2 * pj]) = make_float2(c00[r], c01[r]);
            *reinterpret_cast<float2*>(&Pw[(2 * pi + 1) * LDC + 2 * pj]) = make_float2(c10[r], c11[r]);
        }
        __syncthreads();
        g4 = *reinterpret_cast<const float4*>(&smem[row * LDC + col]);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(&smem[q * (TS * LDC) + row * LDC + col]);
            g4.x += t4.x; g4.y += t4.y; g4.z += t4.z; g4.w += t4.w;
        }
    } else {
    for (int k0 = 0; k0 < K; k0 += BK) {
        DW_STAMP(2 + min(k0 / BK, 40));
        __syncthreads();
        if (k0 == 4 * BK) DW_STAMP(48);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + 256 * q, kr = f / (TS / 4), cq = (f % (TS / 4)) * 4;
            const bool ok = k0 + kr < K;
            // (selects, not branches: rows beyond the batch and columns beyond the layer become zeros)
            const int na = ok ? J.M - (m0 + cq) : 0, nb = ok ? J.N - (n0 + cq) : 0;
            float4 va, vb;
            va.x = na > 0 ? ra[q].x : 0.f; va.y = na > 1 ? ra[q].y : 0.f; va.z = na > 2 ? ra[q].z : 0.f; va.w = na > 3 ? ra[q].w : 0.f;
            vb.x = nb > 0 ? rb[q].x : 0.f; vb.y = nb > 1 ? rb[q].y : 0.f; vb.z = nb > 2 ? rb[q].z : 0.f; vb.w = nb > 3 ? rb[q].w : 0.f;
            *reinterpret_cast<float4*>(&As[kr * LDT + cq]) = va;
            *reinterpret_cast<float4*>(&Bs[kr * LDT + cq]) = vb;
        }
        if (k0 == 4 * BK) DW_STAMP(49);
        request(k0 + BK);                       // (beyond the batch: the last row again, never stored)
        if (k0 == 4 * BK) DW_STAMP(50);
        __syncthreads();
        if (k0 == 4 * BK) DW_STAMP(51);
        acc = dw_slab_products<BK, LDT>(As + (lane >> 4) * LDT + wm + (lane & 15), Bs + (lane >> 4) * LDT + wn + (lane & 15), acc);
    }
    DW_STAMP(44);
    __syncthreads();
    float* Cs = smem;
#pragma unroll
    for (int r = 0; r < 4; ++r) Cs[(wm + (lane >> 4) * 4 + r) * LDC + wn + (lane & 15)] = acc[r];
    __syncthreads();
    g4 = *reinterpret_cast<const float4*>(&Cs[row * LDC + col]);
    }
    if (gm < J.M && gn < J.N) {
        DW_STAMP(45);
        if (epre) {                             // (EpiAdam::apply's vector path on the operands requested above)
            const size_t off = (size_t)gm * J.ld + gn;
            adam_update(pp0.x, mm0.x, vv0.x, g4.x, osc); adam_update(pp0.y, mm0.y, vv0.y, g4.y, osc);
            adam_update(pp0.z, mm0.z, vv0.z, g4.z, osc); adam_update(pp0.w, mm0.w, vv0.w, g4.w, osc);
            *reinterpret_cast<float4*>(J.p + off) = pp0;
            if (!osc.is_sgd) { *reinterpret_cast<float4*>(J.m + off) = mm0; *reinterpret_cast<float4*>(J.v + off) = vv0; }
            if (J.w4.f4) w4_put4(J.w4, gm, gn, pp0);
        } else
        if (J.grad) {
            EpiStore e; e.out = J.grad; e.ld = J.ld;
            EpiStore::State st; e.apply(st, gm, gn, J.N, g4, 0);
        } else {
            EpiAdam e; e.p = J.p; e.m = J.m; e.v = J.v; e.ld = J.ld; e.sc = J.sc;
            e.w4 = J.w4;
            EpiAdam::State st; e.apply(st, gm, gn, J.N, g4, 0);
        }
    }
    DW_STAMP(46);
}
#ifndef DW_WAVES_PER_EU
#define DW_WAVES_PER_EU 4          // (occupancy the register allocation aims at: 4 = 128 registers, 5 = 96, 6 = 80)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DW_WAVES_PER_EU, 8))) void grouped_dw_kernel(DwGroup grp) {
    __shared__ __attribute__((aligned(16))) float smem[kDwSmemBytes / sizeof(float)];
    // (AAE_DW_TS=1, abi_chains.h: every workgroup's first and last clock - which of the launch's block kinds is its long pole)
    const unsigned long long t0 = grp.ts ? wall_clock64() : 0ull;
    grouped_dw_body(grp, smem);
    if (grp.ts && threadIdx.x == 0 && (int)blockIdx.x < grp.ts_cap) { grp.ts[2 * blockIdx.x] = t0; grp.ts[2 * blockIdx.x + 1] = wall_clock64(); }
}


}  // namespace aae
