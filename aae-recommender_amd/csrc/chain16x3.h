// Layer chains for WIDE batches (r5): the programs of chain.h / chain4.h, 16 rows per workgroup, on the bf16 matrix cores.
//
// chain4.h gives a workgroup 4 rows so that a batch of 100 spreads over 25 CUs; every workgroup streams every layer's whole
// weight matrix and owns a CU (16 waves at 128 registers).  At 800-2 000 rows (C4's batch of 1 000, its discriminator's
// stacked 2 000; an item slice's global batch at world 8; C3 at batch 512 beside the row-blocked deferred launch that holds
// most of the chip) that is 200-500 workgroups for the CUs that are free: the launches ran in two and more rounds, 40-75 us
// where one workgroup's program takes 25 (profiles/r5_chain_wide.txt).  Four times the rows per workgroup is one round - if
// the workgroup's own time does not grow with it.  On the fp32 matrix instructions it does: 16 rows x 201 x 200 are 2.7 us
// of v_mfma_f32_16x16x4_f32 per layer on one CU (chain.h, r1) against 0.75 us for 4 rows.  So the products run as in the
// output layer since r3: every fp32 operand as three bf16 terms, six v_mfma_f32_16x16x32_bf16 per 32-deep k-step
// (x3_mfma, gemm_f32.h: 2^-23 relative per product - the fixtures hold at unchanged tolerances), 1.2 us per layer.
//
//   activations  a slot = the 16 x 224 block as THREE bf16 planes [term][row][240] in LDS (kX16SlotB = 23 040 bytes, 7 slots);
//                hi + mid + lo is the fp32 value exactly, so the planes ARE the fp32 activations (element-wise ops add the
//                terms back up) and a lane's A operand - 8 consecutive k of its row, one term - is one 16-byte read.  Rows are
//                480 bytes apart: the four 16-lane groups a ds_read_b128 is served in then touch every bank once.
//   weights      the split copies FX / DX of device_common.h, kept in step by the optimiser epilogues: a lane's B operand (8
//                consecutive k of its column, one term) is one 16-byte load, a wave-instruction 1 KB contiguous.
//   waves        16 = one 16-column tile of the layer's output each (13 of them busy at 200 columns), every wave over all of
//                K: no partial sums to exchange, the epilogue runs on the accumulators (4 rows of one column per lane), and
//                an op is ONE barrier (chain4.h: two).  Four k-steps of weights in flight per wave.
//   bf16 mode    the same kernel with the leading product only (term 0 = the value rounded to nearest even).
// Global stores come from the registers that hold the value (no pass over the slot behind a barrier).  The programs'
// slot numbers (up to 10) are renamed to the 7 physical slots by live range on the host (x16_remap_slots, abi_chains.h).
#pragma once
#include "chain4.h"

namespace aae {

constexpr int kX16R = 16;                    // rows per workgroup
constexpr int kX16T = 1024;                  // threads
constexpr int kX16Kp = 224;                  // columns of a slot image: 7 k-steps of 32 (layer widths <= 208 + the constant-1 column)
constexpr int kX16RowB = 480;                // bytes between rows of a plane (224 bf16 + 32 bytes: conflict-free 16-byte fragment reads)
constexpr int kX16TermB = kX16R * kX16RowB;      // one plane
constexpr int kX16SlotB = 3 * kX16TermB;       // 23 040
constexpr int kX16Slots = 7;
constexpr int kX16Lds = kX16Slots * kX16SlotB;   // 161 280 of the CU's 163 840 bytes
constexpr int kX16Depth = 4;                 // k-steps of weights a wave keeps in flight

__device__ __forceinline__ void x16_put(char* slot, int row, int col, float v) {
    unsigned short t0, t1, t2;
    w4_split3(v, t0, t1, t2);
    char* p = slot + row * kX16RowB + col * 2;
    *reinterpret_cast<unsigned short*>(p) = t0;
    *reinterpret_cast<unsigned short*>(p + kX16TermB) = t1;
    *reinterpret_cast<unsigned short*>(p + 2 * kX16TermB) = t2;
}
__device__ __forceinline__ float x16_get(const char* slot, int row, int col) {
    const char* p = slot + row * kX16RowB + col * 2;
    const unsigned a = *reinterpret_cast<const unsigned short*>(p), b = *reinterpret_cast<const unsigned short*>(p + kX16TermB),
                   c = *reinterpret_cast<const unsigned short*>(p + 2 * kX16TermB);
    return (__uint_as_float(a << 16) + __uint_as_float(b << 16)) + __uint_as_float(c << 16);      // exact: 8 + 8 + 8 significant bits
}

// One wave's 16 x 16 tile of src[16][K] * W over `ksteps` 32-deep k-steps; n0 = first column of the tile.
//   WX: the split weight copy (FX or DX), xpl = rows of one of its planes (Mp or Np).
//   wb: the wave's weight registers, kX16Depth k-steps deep.  x16_issue requests an op's FIRST k-steps into them - for the op
//   in hand, or (r5) for the NEXT linear op of the program right behind the products of the current one: weights depend on
//   nothing the launch computes, the registers are free during the epilogue and the barrier, and an op then opens with its
//   first products instead of an L2 round trip (~1 us of the 4 us a 201 -> 200 layer's products took).
template <bool BF>
struct X16W { gemm_bf16x8 b[kX16Depth][BF ? 1 : 3]; };

template <bool BF>
__device__ __forceinline__ void x16_issue(X16W<BF>& w, const unsigned short* WX, int xpl, int ksteps, int n0, int lane) {
    constexpr int NT = BF ? 1 : 3;
    const int r = lane & 15, kg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(WX), 0, 0x7FFFFFF0, 0x00020000);
    const unsigned vo = (unsigned)((n0 + r) * 32 + 8 * kg) * 2u;
    const unsigned plane = (unsigned)xpl * 64u;                    // bytes of one (k-step, term) plane
#pragma unroll
    for (int ks = 0; ks < kX16Depth; ++ks)
        if (ks < ksteps)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                w.b[ks][t] = __builtin_bit_cast(gemm_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, (unsigned)(ks * 3 + t) * plane, 0));
}

template <bool BF>
__device__ __forceinline__ f32x4 x16_products(const char* src, X16W<BF>& w, const unsigned short* WX, int xpl, int ksteps, int n0, int lane) {
    constexpr int NT = BF ? 1 : 3;
    const int r = lane & 15, kg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(WX), 0, 0x7FFFFFF0, 0x00020000);
    const unsigned vo = (unsigned)((n0 + r) * 32 + 8 * kg) * 2u;
    const unsigned plane = (unsigned)xpl * 64u;
    const char* a0 = src + r * kX16RowB + kg * 16;
    // Three accumulators, by the size of the terms: the leading products (a0 b0), the 2^-8 ones (a1 b0, a0 b1), the 2^-16 ones
    // (a2 b0, a0 b2, a1 b1).  On ONE accumulator every small-term instruction rounds against the large running sum: 42 roundings
    // per 201-deep layer where the fp32 pipe's chain has its own products' only - measured on the e2e fixture (3 epochs at the
    // C3 shape, tests/test_host_gpu.py::test_c3_scale_ranking_matches_reference): dL/d(a1) of gen_step 5e-5 off the 4-row
    // kernel's, enough to flip the sign of near-zero first-layer gradients and part the Adam trajectories.
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc_m = acc, acc_l = acc;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        if (ks < ksteps) {                       // (uniform)
            gemm_bf16x8 a[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const gemm_bf16x8*>(a0 + t * kX16TermB + ks * 64);
            if (BF) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], w.b[ks % kX16Depth][0], acc, 0, 0, 0);
            else {
                const gemm_bf16x8 (&bb)[NT] = w.b[ks % kX16Depth];
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NT - 1], bb[0], acc_l, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bb[NT - 1], acc_l, 0, 0, 0);
                acc_l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NT > 1 ? 1 : 0], bb[NT > 1 ? 1 : 0], acc_l, 0, 0, 0);
                acc_m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NT > 1 ? 1 : 0], bb[0], acc_m, 0, 0, 0);
                acc_m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bb[NT > 1 ? 1 : 0], acc_m, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bb[0], acc, 0, 0, 0);
            }
            if (ks + kX16Depth < ksteps)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    w.b[ks % kX16Depth][t] = __builtin_bit_cast(gemm_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, (unsigned)((ks + kX16Depth) * 3 + t) * plane, 0));
        }
    }
    if (!BF) acc += acc_m + acc_l;
    return acc;
}

// TS: the debug build with in-kernel stamps of workgroup 0 (AAE_CHAIN_TS): per op its start, the end of wave 0's products, the
// end of wave 0's epilogue (the op's end = the next op's start)
template <bool BF, bool TS = false>
__global__ __launch_bounds__(kX16T) void chain16x3_kernel(ChainProgram P) {
    extern __shared__ __attribute__((aligned(16))) char ximg[];        // [kX16Slots][3][16][480 bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * kX16R;
    const int nrows = min(kX16R, P.rows - r0);
    const uint64_t key = rng_key(P.seed, (uint64_t)*P.step_ctr, 0);
    typedef const __attribute__((address_space(1))) float* gf_t;
    int nops = P.nops;
    asm volatile("" : "+s"(nops));
    // the next linear op's first k-steps of weights (x16_issue): ops of the upper rows' program prefix count only for the
    // workgroups that run them (pfx)
    X16W<BF> wb;
    const int pfx = (r0 + kX16R > P.x16_row_lo) ? 1 : 0;
    int wb_for = -1;
    auto look_ahead = [&](int j) {
        wb_for = j;
        if (j < 0) return;
        const ChainOp& o = P.ops[j];
        int nN = o.N, nK = o.K, nxpl = o.xpl; const unsigned short* nWX = o.WX;
        asm volatile("" : "+s"(nN), "+s"(nK), "+s"(nxpl), "+s"(nWX));
        if (wave * 16 < nN) x16_issue<BF>(wb, nWX, nxpl, (nK + 31) >> 5, wave * 16, lane);
    };
    static const bool kLookAhead = true;
    if (kLookAhead) look_ahead(P.x16_first_lin[pfx]);
    unsigned long long* tsp = (TS && blockIdx.x == 0 && tid == 0) ? P.ts : nullptr;
    for (int oi = 0; oi < nops; ++oi) {
        const ChainOp& op = P.ops[oi];
        if (TS && tsp) tsp[3 * oi] = wall_clock64();
        // (the op's scalars requested together and pinned by one statement each: chain4.h)
        int kind = op.kind, src_i = op.src, dst_i = op.dst, row_lo = op.row_lo, one_col = op.one_col, opN = op.N, opK = op.K;
        int ldo = op.ldo, ldo2 = op.ldo2, out_row0 = op.out_row0;
        float* outp = op.out; float* out2p = op.out2;
        int epi_k = op.epi, yslot_k = op.yslot, acc_in = op.acc_in, xpl = op.xpl;
        const unsigned short* WX = op.WX;
        asm volatile("" : "+s"(kind), "+s"(src_i), "+s"(dst_i), "+s"(row_lo), "+s"(one_col), "+s"(opN), "+s"(opK), "+s"(ldo), "+s"(ldo2),
                          "+s"(out_row0), "+s"(outp), "+s"(out2p), "+s"(epi_k), "+s"(yslot_k), "+s"(acc_in), "+s"(xpl), "+s"(WX));
        const float* ygp = op.y_glb; int yld = op.y_ld;
        asm volatile("" : "+s"(ygp), "+s"(yld));
        // an op of the upper rows' program prefix (Enc_eval in front of the discriminator program) runs in every workgroup that
        // holds one of those rows; the rows below row_lo in it read as zero and store nothing
        if (r0 + kX16R <= row_lo) continue;               // (workgroup-uniform)
        const int rlo = row_lo - r0;                    // local rows >= rlo are the op's
        char* dst = ximg + dst_i * kX16SlotB;
        const char* src = ximg + src_i * kX16SlotB;
        const float* qW = nullptr; float* qaux_ptr = nullptr; size_t qstride = 0;
        int qldw = 0, qdst_col0 = 0, qaux = 0, qaux_ld = 0, qgrow0 = 0, qrow_split = 0, qfake_slot = -1;
        float qscale = 0.f;
        if (kind != COP_LINEAR && kind != COP_LINEAR_DX) {
            qW = op.W; qaux_ptr = op.aux_ptr; qstride = op.stride; qldw = op.ldw; qdst_col0 = op.dst_col0; qaux = op.aux;
            qaux_ld = op.aux_ld; qgrow0 = op.grow0; qrow_split = op.row_split; qfake_slot = op.fake_slot; qscale = op.scale;
            asm volatile("" : "+s"(qW), "+s"(qaux_ptr), "+s"(qstride), "+s"(qldw), "+s"(qdst_col0), "+s"(qaux), "+s"(qaux_ld),
                              "+s"(qgrow0), "+s"(qrow_split), "+s"(qfake_slot), "+s"(qscale));
        }
        // global stores of a cell this thread has just computed (rows of the batch, columns of the op, rows of the op's range)
        auto store = [&](int lrow, int col, float v) {
            if (lrow < nrows && lrow >= rlo && col < opN) {
                if (outp) outp[(size_t)(out_row0 + r0 + lrow) * ldo + col] = v;
                if (out2p) out2p[(size_t)(r0 + lrow) * ldo2 + col] = v;
            }
        };
        if (kind == COP_LINEAR || kind == COP_LINEAR_DX) {
            const int N = opN;
            const int ksteps = (opK + 31) >> 5;
            const int n0 = wave * 16, col = n0 + (lane & 15), lr0 = (lane >> 4) * 4;
            if (n0 < kX16Kp) {                                // (waves 14, 15: no tile of the slot)
                float yv[4] = {0.f, 0.f, 0.f, 0.f};
                if (ygp && col < N)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (lr0 + q < nrows) yv[q] = ((gf_t)ygp)[(size_t)(r0 + lr0 + q) * yld + col];
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (n0 < N) {
                    if (wb_for != oi) x16_issue<BF>(wb, WX, xpl, ksteps, n0, lane);
                    acc = x16_products<BF>(src, wb, WX, xpl, ksteps, n0, lane);
                }
                if (kLookAhead) look_ahead(op.x16_next_lin[pfx]);
                if (TS && tsp) { asm volatile("s_nop 0" :: "v"(acc[0])); tsp[3 * oi + 1] = wall_clock64(); }
                const EpiCtx ec = chain_epi_ctx(epi_k, op, P, key, reinterpret_cast<const float*>(ximg));
                const char* ys = ximg + yslot_k * kX16SlotB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int lr = lr0 + q;
                    float v = 0.f;
                    if (col < N && lr < nrows) {
                        v = acc[q];
                        if (acc_in) v += x16_get(dst, lr, col);           // the layer's earlier k-part: this lane's own cell
                        if (ec.epi == CEPI_ACTBWD) {
                            v *= act_grad_from_y(ec.act, ygp ? yv[q] : x16_get(ys, lr, col));
                            if (ec.den) v *= chain_keep(ec, r0 + lr, col) ? ec.mk : 0.f;
                        } else {
                            v = chain_epi<false>(ec, r0 + lr, lr, col, v);
                        }
                        store(lr, col, v);
                    }
                    // (columns >= N read as zero for the next layer; the constant-1 column of an augmented layer input rides here)
                    x16_put(dst, lr, col, col == one_col ? (lr < nrows ? 1.f : 0.f) : v);
                }
            }
        } else if (kind == COP_LOAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j, c2 = qdst_col0 + ecol;
                if (c2 < kX16Kp) {
                    const bool cell = wave < nrows && wave >= rlo && ecol < opN;
                    const float lv = cell ? qW[(size_t)(out_row0 + r0 + wave) * qldw + ecol] * qscale : 0.f;
                    x16_put(dst, wave, c2, c2 == one_col ? (wave < nrows ? 1.f : 0.f) : lv);
                }
            }
        } else if (kind == COP_SLABSUM) {
            // sum of qaux (<= 16) partial slabs: every slab load of a thread is in flight at once
            EpiCtx sec;
            if (epi_k == CEPI_ACTBWD) sec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, reinterpret_cast<const float*>(ximg));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < kX16Kp) {
                    const int rowc = min(wave, max(nrows, 1) - 1), cc = min(ecol, opN - 1);
                    float v[16];
#pragma unroll
                    for (int z = 0; z < 16; ++z) v[z] = qW[(size_t)min(z, qaux - 1) * qstride + (size_t)(r0 + rowc) * qldw + cc];
                    float acc = 0.f;
#pragma unroll
                    for (int z = 0; z < 16; ++z)
                        if (z < qaux) acc += v[z];
                    const bool cell = wave < nrows && ecol < opN;
                    if (epi_k == CEPI_ACTBWD) {
                        const float y = qaux_ptr[(size_t)(r0 + rowc) * qaux_ld + cc];
                        const float kp = (sec.den && cell) ? (chain_keep(sec, r0 + wave, ecol) ? sec.mk : 0.f) : 1.f;
                        acc *= act_grad_from_y(sec.act, y) * kp;
                    }
                    acc = cell ? acc : 0.f;
                    store(wave, ecol, acc);
                    x16_put(dst, wave, ecol, acc);
                }
            }
        } else if (kind == COP_DROPACT) {
            const EpiCtx ec = chain_epi_ctx(CEPI_DROPACT, op, P, key, reinterpret_cast<const float*>(ximg));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < kX16Kp) {
                    float v = 0.f;
                    if (wave < nrows && ecol < opN) {
                        v = chain_epi<false>(ec, r0 + wave, wave, ecol, x16_get(src, wave, ecol));
                        store(wave, ecol, v);
                    }
                    x16_put(dst, wave, ecol, ecol == one_col ? (wave < nrows ? 1.f : 0.f) : v);
                }
            }
        } else if (kind == COP_STORE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < opN && wave < nrows) store(wave, ecol, x16_get(dst, wave, ecol));
            }
        } else if (kind == COP_FINAL_FWD) {
            // one wave per row: softmax / sigmoid / identity over opN (<= 208) columns, in place on dst
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = lane + 64 * j < opN ? x16_get(dst, wave, lane + 64 * j) : -INFINITY;
            if (qaux == 1) {
                float mx = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (lane + 64 * j < opN) { z[j] = expf(z[j] - mx); sum += z[j]; }
                sum = wave_sum(sum);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = z[j] / sum;
            } else if (qaux == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = sigmoidf_(z[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < opN) {
                    if (qaux) x16_put(dst, wave, ecol, z[j]);
                    store(wave, ecol, z[j]);
                }
            }
            if (one_col >= 0 && lane == 0) x16_put(dst, wave, one_col, wave < nrows ? 1.f : 0.f);
        } else if (kind == COP_FINAL_BWD) {
            const char* zs = ximg + yslot_k * kX16SlotB;
            float g[4], zz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = lane + 64 * j < opN;
                g[j] = in ? x16_get(src, wave, lane + 64 * j) : 0.f;
                zz[j] = (in && qaux) ? x16_get(zs, wave, lane + 64 * j) : 0.f;
            }
            float dot = 0.f;
            if (qaux == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dot += g[j] * zz[j];
                dot = wave_sum(dot);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < kX16Kp) {
                    float o = 0.f;
                    if (ecol < opN) {
                        o = qaux == 1 ? zz[j] * (g[j] - dot) : qaux == 2 ? g[j] * zz[j] * (1.f - zz[j]) : g[j];
                        store(wave, ecol, o);
                    }
                    x16_put(dst, wave, ecol, o);       // (zero beyond opN: a following layer reads whole k-steps)
                }
            }
        } else if (kind == COP_PRIOR) {
            const uint64_t k = key ^ (100ull * 0xA0761D6478BD642Full);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ecol = lane + 64 * j;
                if (ecol < kX16Kp) {
                    const int grow = r0 + wave, n = opN;
                    float v = 0.f;
                    if (wave < nrows && ecol < n) {
                        if (grow >= qrow_split) v = qfake_slot >= 0 ? x16_get(ximg + qfake_slot * kX16SlotB, wave, ecol) : qW[(size_t)grow * qldw + ecol];
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
                        store(wave, ecol, v);
                    }
                    x16_put(dst, wave, ecol, ecol == one_col ? (wave < nrows ? 1.f : 0.f) : v);
                }
            }
        } else if (kind == COP_DISC_HEAD) {
            // one wave per row: the discriminator's 1-unit output layer, its loss and its dX (chain.h)
            const EpiCtx ec = chain_epi_ctx(CEPI_ACTBWD, op, P, key, reinterpret_cast<const float*>(ximg));
            const int lrow = wave, grow = r0 + lrow;
            const int Kk = opK, Nn = opN;
            float wv[4], dot = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * j;
                wv[j] = qW[min(k, Kk - 1)];
                if (k < Kk) dot += x16_get(src, lrow, k) * wv[j];
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
                    if (P.loss_terms) P.loss_terms[grow] = -l * invB;
                    else atomicAdd(P.loss_out + P.loss_slot, -l * invB);
                    if (qaux_ptr) qaux_ptr[(size_t)grow * qaux_ld] = gv;
                }
            }
            const char* ys = ximg + yslot_k * kX16SlotB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lane + 64 * j;
                if (k < kX16Kp) {
                    float v = 0.f;
                    if (lrow < nrows && k < Nn) {
                        v = gv * wv[j] * act_grad_from_y(ec.act, x16_get(ys, lrow, k));
                        if (ec.den) v *= chain_keep(ec, grow, k) ? ec.mk : 0.f;
                        store(lrow, k, v);
                    }
                    x16_put(dst, lrow, k, v);
                }
            }
        }
        if (TS && tsp) tsp[3 * oi + 2] = wall_clock64();
        chain_barrier();
    }
    if (TS && tsp) tsp[3 * nops] = wall_clock64();
}

}  // namespace aae
