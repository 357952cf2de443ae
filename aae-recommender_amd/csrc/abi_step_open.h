// The step, part 1: opening a step on a batch (bookkeeping, distinct items, gather, forward programs).
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

// ---- the step ----------------------------------------------------------------------------
// The aae_rng_inject handed to a phase stays in force for the later phases of the same step;
// its device buffers must stay valid until the step's kernels have run.
static void remember_inject(aae_model* m, const aae_rng_inject* inj, bool reset) {
    if (inj) m->inj = *inj;
    else if (reset) memset(&m->inj, 0, sizeof(m->inj));
}

static int ae_encode_impl(aae_handle m, const aae_batch* batch, const aae_rng_inject* inj, float* z_out, bool with_dec,
                          const float* cond_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    TRY(set_batch(m, batch));
    remember_inject(m, inj, true);
    hipStream_t s = S(stream);
    TRY(join_step_open(m, s));      // the previous step's optimiser pass over DEC_V3 reads the step scalars and dh2 (or its
                                    // own copies of them: it then runs on into this step's forward pass - late join)
    m->hstep++;
    // the list of this batch's distinct items and their catch-up were built while the previous step ran
    const bool ahead = m->pf_built && m->pf_step == m->hstep && same_batch(m->pf_built_batch, *batch) && m->lazy;
    m->pf_built = false;
    const bool end_marked = m->end_marked, spec_ok = m->spec_tab_ok;      // (of the step before; this step's advance renews the table entry)
    m->end_marked = false; m->pf_this_step = false; m->spec_tab_ok = m->lazy;
    if (ahead) { std::swap(m->mark, m->mark2); std::swap(m->ulist, m->ulist2); std::swap(m->ucount, m->ucount2); std::swap(m->stamp, m->stamp2); }
    // With the batch's list built ahead nothing sits between the step-opening bookkeeping and the first gather: it rides
    // in that launch (one launch floor, ~4.5 us, less per step)
    constexpr bool fold_ok = true;
    const bool fold_advance = fold_ok && ahead && m->use_chain && !m->ext_first && m->noise_next == nullptr;
    if (!fold_advance)
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       ahead ? (int*)nullptr : m->stamp, ahead ? (int*)nullptr : m->ucount, m->losses);
    m->dense_step = m->noise_next != nullptr;
    const float* noise = m->noise_next; m->noise_next = nullptr;
    if (m->dense_step) {
        if (!m->use_chain) return fail(AAE_ESTATE, "the dense noisy input needs the layer-chain kernels");
        // every row of W1T is read: all of them through the previous step (a scan of tsync after the first such step)
        if (m->lazy && m->cfg.optimizer == AAE_OPT_ADAM) {
            hipLaunchKernelGGL(w1_catchup_kernel, dim3(std::min(m->N, 8192)), dim3(256), 0, s, (const int*)nullptr,
                               (const int*)nullptr, m->N, m->tsync, m->P[P_W1T].p, m->M[0][P_W1T].p, m->V[0][P_W1T].p,
                               m->M[1][P_W1T].p, m->V[1][P_W1T].p, m->ldw1, m->h, m->tab, m->step_ctr, -1);
            LAUNCHCHK("w1_catchup all");
        }
        hipLaunchKernelGGL(uniq_items_kernel, dim3(m->rows, std::max(1, std::min(16, m->chunks / 16 + 1))), dim3(256), 0, s, m->bv,
                           m->mark, m->stamp, m->ulist, m->ucount);       // (the list later phases of the step expect)
        hipLaunchKernelGGL(dense_input_kernel, dim3(m->rows), dim3(1024), 0, s, m->bv, noise, m->noise_ld, m->N,
                           (int)m->cfg.normalize_inputs, m->Xn.p, m->ldn);
        LAUNCHCHK("dense_input");
        const int B = m->rows, h = m->h, N = m->N;
        int tiles = ((B + 63) / 64) * ((h + 63) / 64);
        int splits = std::max(1, std::min(m->max_slabs, 2048 / tiles));
        int kps = ((N + splits - 1) / splits + 63) / 64 * 64;
        splits = (N + kps - 1) / kps;
        GemmShape g{m->Xn.p, m->P[P_W1T].p, B, h, N, m->ldn, m->ldw1, kps};
        EpiSlab e; e.out = m->slabs.p; e.ld = m->ldh; e.slab_stride = (size_t)m->R * m->ldh;
        (void)launch_gemm_mode<0, 0, true>(gmode(m), g, e, splits, s);
        LAUNCHCHK("dense first layer");
        DropSpec d1 = make_drop(m, 0, true, m->inj.masks_dev[0], nullptr, B, h, 0);
        hipLaunchKernelGGL(slab_reduce_fwd_kernel, dim3(grid1d((size_t)B * h)), dim3(256), 0, s, m->slabs.p, splits,
                           e.slab_stride, B, h, m->ldh, m->P[P_B1].p, m->a1.p, m->eh1.p, (int)m->cfg.activation, d1,
                           (uint64_t)m->cfg.seed, m->step_ctr);
        LAUNCHCHK("slab_reduce_fwd");
        m->dec_hidden_done = false; m->enc_bwd_done = false;
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        if (m->pf_armed) m->pf_armed = false;
        return AAE_OK;
    }
    if (m->ext_first) {
        // the first layer is the caller's (aae_set_first_layer_external): AAE_T_ACT_A1 holds this batch's pre-activations
        if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "an external first layer needs the layer-chain kernels (and no VAE mode)");
        m->dec_hidden_done = false; m->enc_bwd_done = false; m->pf_armed = false;
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        return AAE_OK;
    }
    if (m->lazy && !ahead) TRY(lazy_prepare(m, -1, false, s));
    m->dec_hidden_done = false; m->enc_bwd_done = false;
    // Batches beyond one fused launch (112 rows): their tile buckets (the row-blocked output layer's entry lists and the first
    // layer's per-item update read them; one wide launch, 47 us at 512 rows x 100 k items alone, 100 us beside a streaming
    // GEMM) depend on the batch only - built on the side stream beside the list building, the gather and the forward
    // chain, as aae_first_layer_forward does for the item slices, instead of in front of their first reader.  (Up to 112
    // rows the builder rides in the step's first chain launch.)  The side stream is in order behind the previous step's
    // deferred launch, which waited for that step's output layer - the alternate bucket set's last readers are older.
    // (Not on the three-GEMM path: there the side stream holds the previous step's dV3 GEMM for most of this step.)
    {
        constexpr bool bk_ahead = true;
        if (bk_ahead && m->side && m->ev_bk && m->last_out_split && m->rows > 16 * kMB && !m->buckets_valid && fused_decoder_applies(m)) {
            TRY(build_tile_buckets(m, m->side));
            HIPCHK(hipEventRecord(m->ev_bk, m->side));
            m->bk_pending = true;
        }
    }
    const bool pf = m->pf_armed && m->side && m->mark2 && m->lazy && m->use_chain;
    if (m->pf_armed && !pf) m->pf_armed = false;
    if (m->use_chain) {
        // early prefetch (abi_model.h): marked by the step before, this batch's own lists in place (the catch-up skips its rows
        // by them), the table entry of this step written a step early and untouched since
        // (wide batches only: at 100 rows the 3 200 small workgroups of the catch-up, started this early, sit on the CUs the forward
        //  chain's 25 workgroups need whole - C3 0.2386 -> 0.2420 ms/step; C4, 1 000 rows: 0.3932 -> 0.3826; AAE_EARLY_ANY: tests)
        const bool early = pf && ahead && end_marked && spec_ok && m->early_enabled && (m->rows > 16 * kMB || m->early_any);
        if (early) TRY(launch_prefetch(m, true, true));
        TRY(gather_first_layer(m, true, m->inj.masks_dev[0], 0, s, pf && !early, fold_advance));
        if (pf && !early) TRY(launch_prefetch(m));
        TRY(chain_ae_forward(m, with_dec, cond_dev, z_out, s));
        m->phase = 1;
        return AAE_OK;
    }
    TRY(encoder_forward(m, true, m->inj.masks_dev[0], m->inj.masks_dev[1], 0, 1, false, m->zc.p, m->ldc, s));
    // keep a copy of z for the encoder backward (condition plugins replace zc)
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                       m->zsave.p, m->ldz, m->rows, m->c, 1.0f);
    if (z_out)
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                           z_out, m->c, m->rows, m->c, 1.0f);
    LAUNCHCHK("ae_encode copies");
    m->phase = 1;
    return AAE_OK;
}

int aae_ae_encode(aae_handle m, const aae_batch* batch, const aae_rng_inject* inj, float* z_out, void* stream) {
    return ae_encode_impl(m, batch, inj, z_out, false, nullptr, stream);
}


}  // extern "C"
