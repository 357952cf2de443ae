// predict -> rank fused (rank_x3.h): the host side behind aae_predict_topk / aae_decode_topk / aae_rank_max_rows.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
//
// One call ranks up to aae_rank_max_rows() rows - more than the training batch: nothing of it uses the handle's per-batch
// buffers.  Its workspace is the arena's [max_batch][n_items] scratch (AAE_T_ACT_G: the training step's dL/dlogits tiles,
// free between steps once the deferred optimiser launch is joined) + the dA2 slab area behind it:
//   a1 / eh1 [rows][ldh]    first-layer pre-activations / activations (enc_gather_kernel)
//   dh2      [rows][ldh]    the decoder's last hidden activations (ONE chain program: encoder tail -> code -> condition
//                           block -> decoder hidden layers, 4 rows per workgroup)
//   known    [rows][kw]     bitmap of the rows' input items        cand [rows][wgs][K] x (score, id), mm [rows][wgs][2]
// Launches per call: (deferred-Adam flush of enc.lin1 when rows are behind) gather, chain, known-item mask, rank, merge.
#pragma once

namespace {

constexpr int kRankBb = 16 * kMB;          // (r1-r4: rows per row block of the rank kernels - the training kernels' 7 MFMA row blocks; now kRankGR2, rank_x3.h)
// which of the two rank kernels a call takes (rank_x3.h) and its rows per row block
inline bool rank_v2_nb(int NB, int K) {
    constexpr bool v1 = false;       // (rank_x3_kernel, the r4 wave mapping: kept for K != 10, rank_x3.h)
    return !v1 && (K == 10 || (K == 20 && NB < 13));                 // (the other list sizes spill registers in the v2 mapping: they keep v1)
}
inline int rank_bb(const aae_model*, int) { return kRankGR2; }      // (both kernels: 128-row blocks)
constexpr int kRankMaxRows = 4096;

struct RankPlan {
    int rows, K, nblk, wgs, kw, bb;
    float *a1, *eh1, *dh2, *rscale, *cand_v, *mm; int* cand_i; unsigned* known;
    size_t floats;
};

inline int rank_K(int k) { return k <= 10 ? 10 : k <= 20 ? 20 : 32; }

// lays the workspace out for `rows` rows (base == NULL: measures only)
RankPlan rank_plan(const aae_model* m, int rows, int k, float* base) {
    RankPlan p; memset(&p, 0, sizeof(p));
    p.rows = rows; p.K = rank_K(k);
    const int ntiles = (m->N + kTI - 1) / kTI;
    p.bb = rank_bb(m, p.K);
    p.nblk = (rows + p.bb - 1) / p.bb;
    p.wgs = std::max(1, std::min(m->n_cu / std::max(1, p.nblk), ntiles));
    p.kw = (m->N + 31) / 32;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return base ? base + o : nullptr; };
    p.a1 = take((size_t)rows * m->ldh); p.eh1 = take((size_t)rows * m->ldh); p.dh2 = take((size_t)rows * m->ldh);
    p.rscale = take(rows);
    p.known = reinterpret_cast<unsigned*>(take((size_t)rows * p.kw));
    p.cand_v = take((size_t)rows * p.wgs * p.K);
    p.cand_i = reinterpret_cast<int*>(take((size_t)rows * p.wgs * p.K));
    p.mm = take((size_t)rows * p.wgs * 2);
    p.floats = off;
    return p;
}

// the workspace: the [max_batch][n_items] scratch and, where the layout put them right behind it (it does), the dA2 slabs of
// the output layer - both hold data of a running step only, and a rank call runs between steps behind join_deferred()
size_t rank_ws_floats(const aae_model* m) {
    if (m->slabs.p && m->slabs.p > m->G.p) return (size_t)(m->slabs.p - m->G.p) + m->slabs.floats();
    return m->G.floats();
}

// most rows one fused call can rank (0: the fused path does not apply to this handle)
int rank_rows_cap(const aae_model* m, int k) {
    if (!m->rank_ok || k < 1 || k > 32) return 0;
    const size_t have = rank_ws_floats(m);
    int lo = 0, hi = kRankMaxRows;          // (the plan's size is monotone in rows up to rounding: bisect)
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (rank_plan(m, mid, k, nullptr).floats <= have) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <int NB>
int launch_rank_nb(const RankArgs& a, int K, int grid, hipStream_t s) {
    const bool win = x3_big_span(a.N, a.ldv);      // (dec.lin3 beyond 2^31 bytes: the moving-window instantiations, dec_fused.h)
    if (rank_v2_nb(NB, K)) {
        const uint32_t lds2 = (uint32_t)rank_x3v2_lds_bytes(NB);
        switch (K) {
            case 10: { if (win) hipLaunchKernelGGL((rank_x3v2_kernel<NB, 10, true>), dim3(grid), dim3(kNT), lds2, s, a); else hipLaunchKernelGGL((rank_x3v2_kernel<NB, 10>), dim3(grid), dim3(kNT), lds2, s, a); } break;
            case 20: { if (win) hipLaunchKernelGGL((rank_x3v2_kernel<NB, 20, true>), dim3(grid), dim3(kNT), lds2, s, a); else hipLaunchKernelGGL((rank_x3v2_kernel<NB, 20>), dim3(grid), dim3(kNT), lds2, s, a); } break;
            default: { if (win) hipLaunchKernelGGL((rank_x3v2_kernel<NB, 32, true>), dim3(grid), dim3(kNT), lds2, s, a); else hipLaunchKernelGGL((rank_x3v2_kernel<NB, 32>), dim3(grid), dim3(kNT), lds2, s, a); } break;
        }
        LAUNCHCHK("rank_x3v2");
        return AAE_OK;
    }
    const uint32_t lds = (uint32_t)rank_x3_lds_bytes(NB);
    switch (K) {
        case 10: { if (win) hipLaunchKernelGGL((rank_x3_kernel<NB, 10, true>), dim3(grid), dim3(kNT), lds, s, a); else hipLaunchKernelGGL((rank_x3_kernel<NB, 10>), dim3(grid), dim3(kNT), lds, s, a); } break;
        case 20: { if (win) hipLaunchKernelGGL((rank_x3_kernel<NB, 20, true>), dim3(grid), dim3(kNT), lds, s, a); else hipLaunchKernelGGL((rank_x3_kernel<NB, 20>), dim3(grid), dim3(kNT), lds, s, a); } break;
        default: { if (win) hipLaunchKernelGGL((rank_x3_kernel<NB, 32, true>), dim3(grid), dim3(kNT), lds, s, a); else hipLaunchKernelGGL((rank_x3_kernel<NB, 32>), dim3(grid), dim3(kNT), lds, s, a); } break;
    }
    LAUNCHCHK("rank_x3");
    return AAE_OK;
}

bool rank_set_attributes() {
    bool ok = true;
    auto set = [&](const void* f, int NB) {
        ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_x3_lds_bytes(NB)) == hipSuccess;
    };
    set(reinterpret_cast<const void*>(rank_x3_kernel<4, 10>), 4); set(reinterpret_cast<const void*>(rank_x3_kernel<4, 10, true>), 4); set(reinterpret_cast<const void*>(rank_x3_kernel<4, 20>), 4); set(reinterpret_cast<const void*>(rank_x3_kernel<4, 20, true>), 4);
    set(reinterpret_cast<const void*>(rank_x3_kernel<4, 32>), 4); set(reinterpret_cast<const void*>(rank_x3_kernel<4, 32, true>), 4); set(reinterpret_cast<const void*>(rank_x3_kernel<7, 10>), 7); set(reinterpret_cast<const void*>(rank_x3_kernel<7, 10, true>), 7);
    set(reinterpret_cast<const void*>(rank_x3_kernel<7, 20>), 7); set(reinterpret_cast<const void*>(rank_x3_kernel<7, 20, true>), 7); set(reinterpret_cast<const void*>(rank_x3_kernel<7, 32>), 7); set(reinterpret_cast<const void*>(rank_x3_kernel<7, 32, true>), 7);
    set(reinterpret_cast<const void*>(rank_x3_kernel<13, 10>), 13); set(reinterpret_cast<const void*>(rank_x3_kernel<13, 10, true>), 13); set(reinterpret_cast<const void*>(rank_x3_kernel<13, 20>), 13); set(reinterpret_cast<const void*>(rank_x3_kernel<13, 20, true>), 13);
    set(reinterpret_cast<const void*>(rank_x3_kernel<13, 32>), 13); set(reinterpret_cast<const void*>(rank_x3_kernel<13, 32, true>), 13);
    auto set2 = [&](const void* f, int NB) {
        ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_x3v2_lds_bytes(NB)) == hipSuccess;
    };
    set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 10>), 4); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 10, true>), 4); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 20>), 4); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 20, true>), 4);
    set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 32>), 4); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<4, 32, true>), 4); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 10>), 7); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 10, true>), 7);
    set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 20>), 7); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 20, true>), 7); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 32>), 7); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<7, 32, true>), 7);
    set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 10>), 13); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 10, true>), 13); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 20>), 13); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 20, true>), 13);
    set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 32>), 13); set2(reinterpret_cast<const void*>(rank_x3v2_kernel<13, 32, true>), 13);
    (void)hipGetLastError();
    return ok;
}

// dh2 (workspace) of `rows` rows -> [rows][k] ids and scaled scores
int rank_from_dh2(aae_model* m, const RankPlan& p, const BatchView& bv, int k, int exclude_known, int32_t* idx_out,
                  float* val_out, hipStream_t s) {
    hipLaunchKernelGGL(known_mask_kernel, dim3(p.rows), dim3(256), 0, s, bv, exclude_known ? p.known : (unsigned*)nullptr, p.kw,
                       p.dh2, m->ldh, m->h);
    LAUNCHCHK("known_mask");
    RankArgs a; memset(&a, 0, sizeof(a));
    a.dh2 = p.dh2; a.ldh = m->ldh; a.V3a = m->P[P_V3].p; a.ldv = m->ldh; a.N = m->N; a.B = p.rows;
    a.nblk = p.nblk; a.Bb = p.bb; a.known = exclude_known ? p.known : nullptr; a.kw = p.kw;
    a.cand_v = p.cand_v; a.cand_i = p.cand_i; a.mm = p.mm; a.one_term = m->bf16 ? 1 : 0;
    a.dbg = m->opt.rank_skip;
    const int grid = p.wgs * p.nblk;
    {
        ProfScope ps(m, AAE_K_RANK, s);
        switch (m->fused_nb) {
            case 4: TRY(launch_rank_nb<4>(a, p.K, grid, s)); break;
            case 7: TRY(launch_rank_nb<7>(a, p.K, grid, s)); break;
            default: TRY(launch_rank_nb<13>(a, p.K, grid, s)); break;
        }
    }
    switch (p.K) {
        case 10: hipLaunchKernelGGL(rank_merge_kernel<10>, dim3((p.rows + 3) / 4), dim3(256), 0, s, p.cand_v, p.cand_i, p.mm, p.rows, p.wgs, k, reinterpret_cast<int*>(idx_out), val_out); break;
        case 20: hipLaunchKernelGGL(rank_merge_kernel<20>, dim3((p.rows + 3) / 4), dim3(256), 0, s, p.cand_v, p.cand_i, p.mm, p.rows, p.wgs, k, reinterpret_cast<int*>(idx_out), val_out); break;
        default: hipLaunchKernelGGL(rank_merge_kernel<32>, dim3((p.rows + 3) / 4), dim3(256), 0, s, p.cand_v, p.cand_i, p.mm, p.rows, p.wgs, k, reinterpret_cast<int*>(idx_out), val_out); break;
    }
    LAUNCHCHK("rank_merge");
    return AAE_OK;
}

// the decoder's hidden layers of a chain program whose slot `src` holds [z | condition head] (and slot 5 the rest of a wide
// input): -> p.dh2
void rank_dec_hidden(aae_model* m, ChainBuilder& cb, int srcA, int srcB, int rows, const RankPlan& p, hipStream_t s) {
    const int h = m->h;
    ChainOp& v1 = add_dec_in_fwd(cb, m, srcA, srcB, 3, s);
    v1.d = make_drop(m, 0, false, nullptr, nullptr, rows, h, 2); v1.one_col = h;
    ChainOp& v2 = cb.add(cop_fwd(m, P_V2, 3, 4, h + 1, h, CEPI_DROPACT, s));
    v2.d = make_drop(m, 1, false, nullptr, nullptr, rows, h, 3); v2.one_col = h; cop_out(v2, p.dh2, m->ldh);
}

BatchView rank_view(const aae_batch* b) {
    BatchView bv;
    bv.indptr = b->indptr_dev; bv.indices = b->indices_dev; bv.values = b->values_dev;
    bv.rows = b->rows_dev; bv.row_start = b->row_start; bv.n_rows = b->n_rows;
    return bv;
}

// predict -> rank for batch->n_rows <= rank_rows_cap rows (eval mode: no dropout; aae.py:840-870)
int rank_predict(aae_model* m, const aae_batch* batch, const float* cond_dev, int k, int exclude_known, int32_t* idx_out,
                 float* val_out, hipStream_t s) {
    const int rows = batch->n_rows, h = m->h, c = m->c, cp = m->cp;
    TRY(join_deferred(m, s));
    if (m->flushed_hstep != m->hstep) {     // every row of enc.lin1 through the current step: once after the last training step
        TRY(lazy_flush(m, s));               // (rows fall behind only when a step opens: hstep counts them)
        m->flushed_hstep = m->hstep;
    }
    const RankPlan p = rank_plan(m, rows, k, m->G.p);
    const BatchView bv = rank_view(batch);
    {
        DropSpec d1 = make_drop(m, 0, false, nullptr, nullptr, rows, h, 0);
        ProfScope ps(m, AAE_K_ENC_GATHER, s);
        hipLaunchKernelGGL(enc_gather_kernel, dim3(rows), dim3(1024), (uint32_t)((size_t)16 * r4(h) * sizeof(float)), s, bv,
                           (const float*)m->P[P_W1T].p, m->ldw1, (const float*)m->P[P_B1].p, h, (int)m->cfg.normalize_inputs,
                           p.a1, p.eh1, m->ldh, (int)m->cfg.activation, d1, (uint64_t)m->cfg.seed, (const long long*)m->step_ctr,
                           p.rscale, (const float*)nullptr, AdvanceJob{nullptr, nullptr, nullptr, nullptr, 0}, (long long)-1);
        LAUNCHCHK("enc_gather (rank)");
    }
    ChainBuilder cb(m, rows);
    ChainOp& l = cb.add(cop_load(p.eh1, m->ldh, 0, h)); l.one_col = h;
    ChainOp& w2 = cb.add(cop_fwd(m, P_W2, 0, 1, h + 1, h, CEPI_DROPACT, s));
    w2.d = make_drop(m, 1, false, nullptr, nullptr, rows, h, 1); w2.one_col = h;
    cb.add(cop_fwd(m, P_W3, 1, 2, h + 1, c, CEPI_NONE, s));
    ChainOp& f = m->cfg.enc_final == AAE_FINAL_LINEAR ? cb.P.ops[cb.P.nops - 1] : cb.add(cop(COP_FINAL_FWD, 2, 2, c));
    f.aux = m->cfg.enc_final;
    if (wide_dec_in(m)) {
        const int nA = kCWide - c, ci = m->cfg.cond_inc;
        ChainOp& ca = cb.add(cop_load(cond_dev, ci, 2, nA)); ca.dst_col0 = c;
        ChainOp& cl = cb.add(cop_load(cond_dev + nA, ci, 5, ci - nA)); cl.one_col = cp - kCWide;
    } else if (m->cfg.cond_inc > 0) {
        ChainOp& cl = cb.add(cop_load(cond_dev, m->cfg.cond_inc, 2, m->cfg.cond_inc)); cl.dst_col0 = c; cl.one_col = cp;
    } else {
        f.one_col = cp;
    }
    rank_dec_hidden(m, cb, 2, 5, rows, p, s);
    TRY(launch_chain(m, cb, s));
    return rank_from_dh2(m, p, bv, k, exclude_known, idx_out, val_out, s);
}

// the same from a decoder input the caller built: zc_dev [rows][zc_ld]
int rank_decode(aae_model* m, const float* zc_dev, int64_t zc_ld, const aae_batch* batch, int k, int exclude_known,
                int32_t* idx_out, float* val_out, hipStream_t s) {
    const int rows = batch->n_rows, cp = m->cp;
    TRY(join_deferred(m, s));
    const RankPlan p = rank_plan(m, rows, k, m->G.p);
    ChainBuilder cb(m, rows);
    if (wide_dec_in(m)) {
        cb.add(cop_load(zc_dev, (int)zc_ld, 2, kCWide));
        ChainOp& lb = cb.add(cop_load(zc_dev + kCWide, (int)zc_ld, 5, cp - kCWide)); lb.one_col = cp - kCWide;
    } else {
        ChainOp& l = cb.add(cop_load(zc_dev, (int)zc_ld, 2, cp)); l.one_col = cp;
    }
    rank_dec_hidden(m, cb, 2, 5, rows, p, s);
    TRY(launch_chain(m, cb, s));
    return rank_from_dh2(m, p, rank_view(batch), k, exclude_known, idx_out, val_out, s);
}

int rank_check_batch(const aae_batch* b) {
    if (!b || !b->indptr_dev || !b->indices_dev || !b->values_dev) return fail(AAE_EINVAL, "batch pointers are NULL");
    return AAE_OK;
}

}  // namespace
