// GEMM wrappers and the per-layer (unchained) path of the nets: batch binding, deferred Adam bookkeeping of enc.lin1, encoder /
// discriminator / decoder hidden layers layer by layer (widths beyond the chain kernels, and the A/B path of the tests).
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// GEMM wrappers (see gemm_f32.h for operand forms)
// ------------------------------------------------------------------------------------------
// Y[rows][out] = epi( X[rows][in+1] * Wa[out][in+1]^T )
// bf: arithmetic of the product (gemm_f32.h: kGemmF32 / kGemmBf16 - cfg.dtype, every forward and dX product of a Linear layer
// takes it - / kGemmX3: fp32 emulated on the bf16 matrix cores, the vocabulary-wide streaming GEMMs only)
static inline int gmode(const aae_model* m) { return m->bf16 ? kGemmBf16 : m->x3_gemm ? kGemmX3 : kGemmF32; }
template <class Epi>
int linear_fwd(const float* X, int ldx, int rows, const Ten& Wa, const Epi& epi, hipStream_t s, int bf = 0) {
    GemmShape g{X, Wa.p, rows, (int)Wa.rows, (int)Wa.cols, ldx, (int)Wa.ld, r4((int)Wa.cols) + 16};
    g.k_per_split = ((int)Wa.cols + 63) / 64 * 64;
    if (Wa.rows > 4096) (void)launch_gemm_mode<0, 1, true>(bf, g, epi, 1, s);   // vocabulary-wide: streaming regime
    else (void)launch_gemm_mode<0, 1, false>(bf, g, epi, 1, s);
    LAUNCHCHK("linear_fwd");
    return AAE_OK;
}
// dX[rows][n_in] = epi( Gd[rows][out] * Wa[out][0:n_in] )
template <class Epi>
int linear_dx(const float* Gd, int ldg, int rows, const Ten& Wa, int n_in, const Epi& epi, hipStream_t s, int bf = 0) {
    GemmShape g{Gd, Wa.p, rows, n_in, (int)Wa.rows, ldg, (int)Wa.ld, 0};
    g.k_per_split = ((int)Wa.rows + 63) / 64 * 64;
    (void)launch_gemm_mode<0, 0, false>(bf, g, epi, 1, s);
    LAUNCHCHK("linear_dx");
    return AAE_OK;
}
// dWa[out][in+1] = Gd[rows][out]^T * X[rows][in+1]  -> optimiser update (or gradient export)
int linear_dw(aae_model* m, const float* Gd, int ldg, int rows, const float* X, int ldx, int pid, int which,
              hipStream_t s) {
    const Ten& W = m->P[pid];
    GemmShape g{Gd, X, (int)W.rows, (int)W.cols, rows, ldg, ldx, 0};
    g.k_per_split = (rows + 63) / 64 * 64;
    const bool big = W.rows > 4096;
    // bf16 mode: of the weight gradients only the decoder output layer's is a bf16 product (the hidden layers' and the
    // sparse first layer's stay fp32: they are launch-latency, not matrix-pipe, bound)
    const int bf = pid == P_V3 ? gmode(m) : kGemmF32;
    if (m->cfg.grad_mode == AAE_GRAD_EXPORT) {
        EpiStore e; e.out = m->Gr[pid].p; e.ld = (int)W.ld;
        if (big) (void)launch_gemm_mode<1, 0, true>(bf, g, e, 1, s); else (void)launch_gemm_mode<1, 0, false>(bf, g, e, 1, s);
    } else {
        const int set = (which == O_GEN) ? 1 : 0;
        EpiAdam e; e.p = W.p; e.m = m->M[set][pid].p; e.v = m->V[set][pid].p; e.ld = (int)W.ld; e.sc = m->sc + which;
        m->pt_ok[pid] = false;
        if (big) (void)launch_gemm_mode<1, 0, true>(bf, g, e, 1, s); else (void)launch_gemm_mode<1, 0, false>(bf, g, e, 1, s);
    }
    LAUNCHCHK("linear_dw");
    return AAE_OK;
}

int set_batch(aae_model* m, const aae_batch* b) {
    if (!b || !b->indptr_dev || !b->indices_dev || !b->values_dev) return fail(AAE_EINVAL, "batch pointers are NULL");
    if (b->n_rows < 1 || b->n_rows > m->R) return fail(AAE_EINVAL, "batch n_rows outside [1, max_batch]");
    if (b->nnz_bound > m->cfg.max_nnz) return fail(AAE_EINVAL, "batch nnz_bound > max_nnz");
    m->bv.indptr = b->indptr_dev; m->bv.indices = b->indices_dev; m->bv.values = b->values_dev;
    m->bv.rows = b->rows_dev; m->bv.row_start = b->row_start; m->bv.n_rows = b->n_rows;
    m->rows = b->n_rows; m->have_batch = true; m->buckets_valid = false; m->w1_merged = false;
    {   // 16 entries per workgroup pass; unknown row bound -> 64 strided chunks
        int mr = b->max_row_nnz > 0 ? b->max_row_nnz : 1024;
        m->chunks = std::max(1, std::min(64, (mr + 15) / 16));
    }
    return AAE_OK;
}

// lazy Adam: list the distinct items of the running batch and bring their W1T rows up to date
// (through step t-1 before a training gather, through step t for predict / export)
int lazy_prepare(aae_model* m, int upto_off, bool bump, hipStream_t s) {
    if (bump) hipLaunchKernelGGL(bump_stamp_kernel, dim3(1), dim3(1), 0, s, m->stamp, m->ucount);
    int gy = std::max(1, std::min(16, m->chunks / 16 + 1));
    hipLaunchKernelGGL(uniq_items_kernel, dim3(m->rows, gy), dim3(256), 0, s, m->bv, m->mark, m->stamp, m->ulist,
                       m->ucount);
    LAUNCHCHK("uniq_items");
    if (m->cfg.optimizer == AAE_OPT_ADAM) {
        int grid = std::min(m->cfg.max_nnz, std::max(256, m->rows * 32));
        hipLaunchKernelGGL(w1_catchup_kernel, dim3(grid), dim3(256), 0, s, m->ulist, m->ucount, m->N, m->tsync,
                           m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p, m->M[1][P_W1T].p, m->V[1][P_W1T].p,
                           m->ldw1, m->h, m->tab, m->step_ctr, upto_off);
        LAUNCHCHK("w1_catchup");
    }
    return AAE_OK;
}

// lazy Adam: every row of W1T (and its four moment tensors) through the current step
int lazy_flush(aae_model* m, hipStream_t s) {
    if (!m->lazy || m->cfg.optimizer != AAE_OPT_ADAM) return AAE_OK;
    hipLaunchKernelGGL(w1_catchup_kernel, dim3(std::min(m->N, 8192)), dim3(256), 0, s, (const int*)nullptr,
                       (const int*)nullptr, m->N, m->tsync, m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p,
                       m->M[1][P_W1T].p, m->V[1][P_W1T].p, m->ldw1, m->h, m->tab, m->step_ctr, 0);
    LAUNCHCHK("w1_catchup all");
    return AAE_OK;
}

// Encoder forward (aae.py:129-146) into `z_dst` [rows][ldz_dst] (first n_code columns).
// train=false: eval mode (no dropout).  reuse_a1: skip the gather, start from m->a1.
int encoder_forward(aae_model* m, bool train, const uint8_t* mk1, const uint8_t* mk2, uint32_t sid1, uint32_t sid2,
                    bool reuse_a1, float* z_dst, int ldz_dst, hipStream_t s) {
    const int B = m->rows, h = m->h;
    DropSpec d1 = make_drop(m, 0, train, mk1, nullptr, B, h, sid1);
    DropSpec d2 = make_drop(m, 1, train, mk2, nullptr, B, h, sid2);
    if (!reuse_a1) {
        ProfScope ps(m, AAE_K_ENC_GATHER, s);
        size_t shm = (size_t)16 * r4(h) * sizeof(float);
        hipLaunchKernelGGL(enc_gather_kernel, dim3(B), dim3(1024), shm, s, m->bv, m->P[P_W1T].p, m->ldw1,
                           m->P[P_B1].p, h, m->cfg.normalize_inputs, m->a1.p, m->eh1.p, m->ldh, m->cfg.activation,
                           d1, m->cfg.seed, m->step_ctr, m->rscale, m->doc_l1, AdvanceJob{nullptr, nullptr, nullptr, nullptr, 0},
                           (long long)-1);
        LAUNCHCHK("enc_gather");
    } else {
        hipLaunchKernelGGL(drop_act_kernel, dim3(grid1d((size_t)B * h)), dim3(256), 0, s, m->a1.p, m->eh1.p, B, h,
                           m->ldh, m->cfg.activation, d1, m->cfg.seed, m->step_ctr);
        LAUNCHCHK("drop_act");
    }
    EpiDropAct e2; e2.out = m->eh2.p; e2.ld = m->ldh; e2.act = m->cfg.activation; e2.d = d2; e2.seed = m->cfg.seed;
    e2.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->eh1.p, m->ldh, B, m->P[P_W2], e2, s, gmode(m)));
    EpiStore e3; e3.out = z_dst; e3.ld = ldz_dst;
    TRY(linear_fwd(m->eh2.p, m->ldh, B, m->P[P_W3], e3, s, gmode(m)));
    if (m->cfg.enc_final != AAE_FINAL_LINEAR) {
        hipLaunchKernelGGL(final_act_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, z_dst, B, m->c, ldz_dst,
                           m->cfg.enc_final, (float*)nullptr, 0);
        LAUNCHCHK("final_act_fwd");
    }
    return AAE_OK;
}

int launch_w1_items(aae_model* m, const float* ga1, int rpb, size_t bstride, int which, hipStream_t s);
W1Items w1_items_args(aae_model* m, const float* ga1, int rpb, size_t bstride, int which);
int ensure_buckets(aae_model* m, hipStream_t s);

// Encoder backward from dL/dz (gz [rows][ldgz]) + optimiser `which` (O_ENC or O_GEN) on all
// encoder parameters.  z [rows][ldzz] is the encoder output of the matching forward.
int encoder_backward(aae_model* m, const float* gz, int ldgz, const float* z, int ldzz, const uint8_t* mk1,
                     const uint8_t* mk2, uint32_t sid1, uint32_t sid2, int which, hipStream_t s) {
    const int B = m->rows, h = m->h, cc = m->c;
    DropSpec d1 = make_drop(m, 0, true, mk1, nullptr, B, h, sid1);
    DropSpec d2 = make_drop(m, 1, true, mk2, nullptr, B, h, sid2);
    const float* ga3 = gz; int ldga3 = ldgz;
    if (m->cfg.enc_final != AAE_FINAL_LINEAR) {
        hipLaunchKernelGGL(final_act_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, z, ldzz, gz, ldgz, m->ga3.p,
                           m->ldz, B, cc, m->cfg.enc_final);
        LAUNCHCHK("final_act_bwd");
        ga3 = m->ga3.p; ldga3 = m->ldz;
    }
    // lin3: dX first (needs the old weights), then dW + update
    EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->eh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
    b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
    TRY(linear_dx(ga3, ldga3, B, m->P[P_W3], h, b2, s, gmode(m)));
    TRY(linear_dw(m, ga3, ldga3, B, m->eh2.p, m->ldh, P_W3, which, s));
    // lin2
    EpiActBwd b1; b1.out = m->gb1.p; b1.ld = m->ldh; b1.y = m->eh1.p; b1.ldy = m->ldh; b1.act = m->cfg.activation;
    b1.d = d1; b1.seed = m->cfg.seed; b1.step_ctr = m->step_ctr;
    TRY(linear_dx(m->gb0.p, m->ldh, B, m->P[P_W2], h, b1, s, gmode(m)));
    TRY(linear_dw(m, m->gb0.p, m->ldh, B, m->eh1.p, m->ldh, P_W2, which, s));
    // lin1: bias column sum + its optimiser, then the row-sparse weight gradient + optimiser (w1_update.h)
    const int set = (which == O_GEN) ? 1 : 0;
    const bool exportg = m->cfg.grad_mode == AAE_GRAD_EXPORT;
    hipLaunchKernelGGL(colsum_adam_kernel, dim3((h + 63) / 64), dim3(1024), 0, s, m->gb1.p, B, h, m->ldh,
                       m->P[P_B1].p, m->M[set][P_B1].p, m->V[set][P_B1].p, exportg ? m->Gr[P_B1].p : (float*)nullptr,
                       m->sc + which);
    LAUNCHCHK("colsum_adam");
    return launch_w1_items(m, m->gb1.p, 0, 0, which, s);
}

// Discriminator forward on `rows` rows of m->zin (aae.py:195-213) -> m->dout (sigmoid)
int disc_forward(aae_model* m, int rows, const uint8_t* m1a, const uint8_t* m1b, const uint8_t* m2a,
                 const uint8_t* m2b, int split, uint32_t sid1, uint32_t sid2, hipStream_t s) {
    const int h = m->h;
    DropSpec d1 = make_drop(m, 0, true, m1a, m1b, split, h, sid1);
    DropSpec d2 = make_drop(m, 1, true, m2a, m2b, split, h, sid2);
    EpiDropAct e1; e1.out = m->xh1.p; e1.ld = m->ldh; e1.act = m->cfg.activation; e1.d = d1; e1.seed = m->cfg.seed;
    e1.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->zin.p, m->ldz, rows, m->P[P_D1], e1, s, gmode(m)));
    EpiDropAct e2 = e1; e2.out = m->xh2.p; e2.d = d2;
    TRY(linear_fwd(m->xh1.p, m->ldh, rows, m->P[P_D2], e2, s, gmode(m)));
    EpiSigmoid e3; e3.out = m->dout.p; e3.ld = 4;
    TRY(linear_fwd(m->xh2.p, m->ldh, rows, m->P[P_D3], e3, s));
    return AAE_OK;
}

int finalize_bce_loss(aae_model* m, int nblocks, hipStream_t s) {
    const int B = m->rows;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, m->bce_partials, nblocks, m->fix_partials,
                       B * m->chunks,
                       1.0f / ((float)B * (float)m->N), m->losses, 0);
    LAUNCHCHK("loss_finalize");
    return AAE_OK;
}

int decoder_hidden_forward(aae_model* m, bool train, const uint8_t* mk1, const uint8_t* mk2, int rows,
                           hipStream_t s) {
    DropSpec d1 = make_drop(m, 0, train, mk1, nullptr, rows, m->h, 2);
    DropSpec d2 = make_drop(m, 1, train, mk2, nullptr, rows, m->h, 3);
    EpiDropAct e1; e1.out = m->dh1.p; e1.ld = m->ldh; e1.act = m->cfg.activation; e1.d = d1; e1.seed = m->cfg.seed;
    e1.step_ctr = m->step_ctr;
    TRY(linear_fwd(m->zc.p, m->ldc, rows, m->P[P_V1], e1, s, gmode(m)));
    EpiDropAct e2 = e1; e2.out = m->dh2.p; e2.d = d2;
    TRY(linear_fwd(m->dh1.p, m->ldh, rows, m->P[P_V2], e2, s, gmode(m)));
    return AAE_OK;
}

int stage_zc(aae_model* m, const float* src, int64_t ld, int rows, hipStream_t s) {
    if (src != m->zc.p) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)rows * m->cp)), dim3(256), 0, s, src, (int)ld, m->zc.p,
                           m->ldc, rows, m->cp, 1.0f);
        LAUNCHCHK("copy zc");
    }
    return AAE_OK;
}



}  // namespace
