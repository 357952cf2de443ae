// The step, part 3: decoder-only / VAE entry points, the phases cut at the output layer and at the first layer, disc_step, gen_step, aae_step.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

int aae_decoder_step(aae_handle m, const aae_batch* batch, const float* zin_dev, int64_t zin_ld,
                     const aae_rng_inject* inj, float* dzin_out, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!zin_dev) return fail(AAE_EINVAL, "zin_dev is NULL");
    if (zin_ld < m->cp) return fail(AAE_EINVAL, "zin_ld < n_code + cond_inc");
    TRY(set_batch(m, batch));
    remember_inject(m, inj, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));       // the previous step's optimiser pass over DEC_V3 reads the step scalars and dh2
    m->hstep++; m->pf_armed = false;
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       m->stamp, m->ucount, m->losses);
    LAUNCHCHK("advance_step");
    m->dec_hidden_done = false; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->phase = 1;
    TRY(aae_ae_decode_backward(m, zin_dev, zin_ld, nullptr, dzin_out, stream));
    m->phase = 0;
    return AAE_OK;
}

// VAE.partial_fit (vae.py:147-186): loss = mean BCE + KL sum (vae.py:132-145), one Adam over all five Linears
int aae_vae_step(aae_handle m, const aae_batch* batch, const float* cond_dev, const float* eps_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->vae) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (!m->use_chain) return fail(AAE_ESTATE, "VAE mode needs the layer-chain kernels");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    TRY(set_batch(m, batch));
    remember_inject(m, nullptr, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    m->hstep++; m->pf_armed = false;
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                       m->stamp, m->ucount, m->losses);
    LAUNCHCHK("advance_step");
    if (m->lazy) TRY(lazy_prepare(m, -1, false, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));                 // eh1 = act(fc1(normalize(x))), vae.py:111-113
    TRY(chain_vae_forward(m, cond_dev, eps_dev, m->rows, s));
    m->dec_hidden_done = true; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->vae_bwd = true; m->vae_cut = false; m->phase = 1;
    const int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->vae_bwd = false;
    if (rc != AAE_OK) return rc;
    TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
    m->phase = 0;
    return AAE_OK;
}

// VAE.predict (vae.py:229-266): the same stochastic forward (the reference samples eps in eval mode too)
int aae_vae_predict(aae_handle m, const aae_batch* batch, const float* cond_dev, const float* eps_dev, float* out_dev,
                    int64_t out_ld, void* stream) {
    if (!m || !out_dev) return fail(AAE_EINVAL, "handle/out is NULL");
    if (!m->vae || !m->use_chain) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    if (out_ld < m->N || (out_ld & 3) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return fail(AAE_EINVAL, "out_dev must be 16-byte aligned with out_ld >= n_items and out_ld % 4 == 0");
    TRY(set_batch(m, batch));
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));
    TRY(chain_vae_forward(m, cond_dev, eps_dev, m->rows, s));
    EpiSigmoid e; e.out = out_dev; e.ld = (int)out_ld;
    TRY(linear_fwd(m->dh2.p, m->ldh, m->rows, m->P[P_V3], e, s, gmode(m)));
    m->phase = 0;
    return AAE_OK;
}

// The VAE step cut at the condition boundary (vae.py:120-130: `z = self.conditions.encode_impose(z, condition_data)`
// between reparametrize and decode), for condition plugins that run in the host framework:
//   aae_vae_encode            x -> fc1 -> (mu, logvar) -> z = mu + eps * exp(logvar / 2)     (train != 0 opens a step)
//   [host: zc = conditions.encode_impose(z, c)]
//   aae_vae_decode_backward   fc3 -> fc4 -> BCE, backward to dL/dzc, fc3 / fc4 updates
//   [host: backprop dzc through the conditions -> dz; conditions.step()]
//   aae_vae_encoder_backward  reparametrize' + KL gradient -> [fc21; fc22] -> fc1, their updates
int aae_vae_encode(aae_handle m, const aae_batch* batch, const float* eps_dev, float* z_out_dev, int32_t train, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->vae || !m->use_chain) return fail(AAE_ESTATE, "model was not created in VAE mode (cfg.model_kind = 3)");
    if (m->cfg.rng_mode == AAE_RNG_INJECT && !eps_dev) return fail(AAE_EINVAL, "rng_mode inject needs eps_dev");
    TRY(set_batch(m, batch));
    remember_inject(m, nullptr, true);
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (train) {
        m->hstep++; m->pf_armed = false;
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           m->stamp, m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        if (m->lazy) TRY(lazy_prepare(m, -1, false, s));
    } else if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    TRY(gather_first_layer(m, false, nullptr, 0, s));
    TRY(chain_vae_encode(m, eps_dev, z_out_dev, m->rows, s));
    m->dec_hidden_done = false; m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    m->phase = train ? 1 : 0; m->vae_cut = train != 0;
    return AAE_OK;
}

int aae_vae_decode_backward(aae_handle m, const float* zc_dev, int64_t zc_ld, float* dzc_out_dev, void* stream) {
    if (!m || !zc_dev) return fail(AAE_EINVAL, "NULL argument");
    if (!m->vae || m->phase != 1 || !m->vae_cut) return fail(AAE_ESTATE, "aae_vae_decode_backward without aae_vae_encode(train)");
    if (zc_ld < m->cp) return fail(AAE_EINVAL, "zc_ld < n_code + cond_inc");
    hipStream_t s = S(stream);
    TRY(stage_zc(m, zc_dev, zc_ld, m->rows, s));
    TRY(chain_vae_dec_hidden(m, m->rows, s));
    m->dec_hidden_done = true; m->vae_bwd = true;
    const int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, dzc_out_dev, stream);
    m->vae_bwd = false;
    return rc;
}

int aae_vae_encoder_backward(aae_handle m, const float* dz_dev, int64_t dz_ld, void* stream) {
    if (!m || !dz_dev) return fail(AAE_EINVAL, "NULL argument");
    if (!m->vae || m->phase != 2 || !m->vae_cut) return fail(AAE_ESTATE, "aae_vae_encoder_backward without aae_vae_decode_backward");
    if (dz_ld < m->c) return fail(AAE_EINVAL, "dz_ld < n_code");
    hipStream_t s = S(stream);
    const int B = m->rows;
    TRY(chain_vae_backward_enc(m, dz_dev, (int)dz_ld, s));
    DwBuilder dw;
    dw.add(m, m->gmulv.p, (int)m->gmulv.ld, m->eh1.p, m->ldh, B, P_W3, O_ENC);
    TRY(dw.add_first_layer(m, m->gb3.p, O_ENC, s)); m->w1_merged = true;
    TRY(dw.launch(s));
    TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
    m->phase = 0; m->vae_cut = false;
    return AAE_OK;
}

int aae_ae_encoder_backward(aae_handle m, const float* dz_dev, int64_t dz_ld, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->phase != 2) return fail(AAE_ESTATE, "aae_ae_encoder_backward without aae_ae_decode_backward");
    hipStream_t s = S(stream);
    if (m->use_chain) {
        if (!m->enc_bwd_done) {
            TRY(chain_ae_backward(m, false, true, nullptr, 0, dz_dev, (int)dz_ld, nullptr, O_ENC, s));
            DwBuilder dw;
            dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, m->rows, P_W3, O_ENC);
            dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, m->rows, P_W2, O_ENC);
            TRY(dw.add_first_layer(m, ga1_ptr(m), O_ENC, s)); m->w1_merged = true;
            TRY(dw.launch(s));
        }
        if (!m->ext_first || m->own_first) TRY(encoder_first_layer_update(m, m->gb3.p, O_ENC, s, m->w1_merged));
        m->phase = 3;
        return AAE_OK;
    }
    const float* gz = dz_dev ? dz_dev : m->gzc.p;
    int ld = dz_dev ? (int)dz_ld : m->ldc;
    TRY(encoder_backward(m, gz, ld, m->zsave.p, m->ldz, m->inj.masks_dev[0], m->inj.masks_dev[1], 0, 1, O_ENC, s));
    m->phase = 3;
    return AAE_OK;
}

// ---- the ae phase cut at the decoder's output layer (vocabulary-sharded data parallelism) -----------------------
// aae_ae_forward: encoder + the decoder's hidden layers on this rank's documents; dh2 stays in AAE_T_ACT_DH2.
int aae_ae_forward(aae_handle m, const aae_batch* batch, const float* cond_dev, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "aae_ae_forward needs the layer-chain kernels (and no VAE mode)");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    hipStream_t s = S(stream);
    TRY(ae_encode_impl(m, batch, inj, nullptr, true, cond_dev, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, s, cond_dev,
                           m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    return AAE_OK;
}

// aae_output_layer_step: the decoder's output layer alone over the handle's items - logits from AAE_T_ACT_DH2, BCE
// against the batch, dV3 + dec_optim on V3 (or its gradient in export mode), dL/d(dh2) summed into AAE_T_ACT_DA2.
//   batch != NULL: a step of its own (a handle that owns a shard of the vocabulary: the caller filled ACT_DH2 with
//                  the hidden activations of batch->n_rows documents, e.g. by an all-gather);
//   batch == NULL: continues the step aae_ae_forward started on this handle.
int aae_output_layer_step(aae_handle m, const aae_batch* batch, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae) return fail(AAE_ESTATE, "aae_output_layer_step: not in VAE mode");
    hipStream_t s = S(stream);
    if (batch || m->opt_pending) TRY(join_deferred(m, s));      // (batch = NULL: a prefetch started by this step keeps running)
    if (batch) {
        TRY(set_batch(m, batch));
        remember_inject(m, nullptr, true);
        m->hstep++; m->pf_armed = false;
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           m->stamp, m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        m->enc_bwd_done = false; m->fuse_enc_bwd = false;
    } else if (m->phase != 1 || !m->dec_hidden_done) {
        return fail(AAE_ESTATE, "aae_output_layer_step(batch = NULL) without aae_ae_forward");
    }
    m->dec_hidden_done = true;
    m->phase = 1;
    m->only_output_layer = true;
    int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->only_output_layer = false;
    TRY(rc);
    m->phase = batch ? 0 : 2;
    return AAE_OK;
}

// aae_ae_backward: the rest of the ae phase on this rank's documents from dL/d(dh2) (dA2_dev [rows][ld = ACT_DH2's],
// NULL = AAE_T_ACT_DA2 of this handle): decoder hidden layers backward, encoder backward, their optimiser updates
// (or exported gradients).  Follows aae_ae_forward.
int aae_ae_backward(aae_handle m, const float* dA2_dev, int64_t dA2_ld, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (!m->use_chain || m->vae) return fail(AAE_ESTATE, "aae_ae_backward needs the layer-chain kernels (and no VAE mode)");
    if ((m->phase != 1 && m->phase != 2) || !m->dec_hidden_done) return fail(AAE_ESTATE, "aae_ae_backward without aae_ae_forward");
    if (dA2_dev && dA2_ld != m->ldh) return fail(AAE_EINVAL, "aae_ae_backward: dA2_ld must equal the leading dimension of AAE_T_ACT_DH2");
    hipStream_t s = S(stream);
    const int B = m->rows;
    TRY(chain_ae_backward(m, true, true, dA2_dev ? dA2_dev : m->da2.p, 0, nullptr, 0, nullptr, O_ENC, s, 1));
    DwBuilder dw;
    dw.add(m, m->gb0.p, m->ldh, m->dh1.p, m->ldh, B, P_V2, O_DEC);
    dw.add(m, m->gb1.p, m->ldh, m->zc.p, m->ldc, B, P_V1, O_DEC);
    dw.add(m, m->ga3.p, m->ldz, m->eh2.p, m->ldh, B, P_W3, O_ENC);
    dw.add(m, m->gb2.p, m->ldh, m->eh1.p, m->ldh, B, P_W2, O_ENC);
    TRY(dw.add_first_layer(m, ga1_ptr(m), O_ENC, s)); m->w1_merged = true;      // (external first layer: its bias blocks only)
    TRY(dw.launch(s));
    m->enc_bwd_done = true;
    m->phase = 2;
    return aae_ae_encoder_backward(m, nullptr, 0, stream);
}

// ---- the first encoder layer sharded over the vocabulary (with the decoder's output layer: both [n_items, n_hidden]
// matrices live with the owner of their item slice, the ranks exchange [global rows, n_hidden] activations) ----------
int aae_set_doc_l1(aae_handle m, const float* doc_l1_dev) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    m->doc_l1 = doc_l1_dev;
    return AAE_OK;
}

int aae_set_first_layer_external(aae_handle m, int on) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (on && (!m->use_chain || m->vae)) return fail(AAE_ESTATE, "an external first layer needs the layer-chain kernels (and no VAE mode)");
    m->ext_first = on != 0;
    return AAE_OK;
}

// This handle's share of the first layer's pre-activations for the batch: sum over ITS items of x[b][i] * enc.lin1[:, i]
// (+ bias_dev[n_hidden] when given: the bias stays with the replicas, exactly one share adds it) -> AAE_T_ACT_A1
// [rows][n_hidden].
//   batch != NULL: a new step of this handle (step scalars advance, the batch's rows of the deferred Adam are caught up);
//   batch == NULL: the running batch again with the weights as they are now (disc_step's Enc_eval after enc_optim).
int aae_first_layer_forward(aae_handle m, const aae_batch* batch, const float* bias_dev, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae || m->cfg.grad_mode != AAE_GRAD_FUSED) return fail(AAE_ESTATE, "aae_first_layer_forward: fused optimiser, no VAE mode");
    hipStream_t s = S(stream);
    bool pf = false, fold_advance = false;
    if (batch) {
        TRY(join_deferred(m, s));       // (batch = NULL: a deferred optimiser launch of the output layer keeps running - it
                                        //  touches dec.lin3, its moments, the stored dL/dlogits and dh2, nothing of this layer)
        TRY(set_batch(m, batch));
        remember_inject(m, nullptr, true);
        m->hstep++;
        // (as in aae_step: the list of this batch's distinct items and their catch-up were built while the previous step
        //  ran, if the caller named the batch with aae_prefetch_batch)
        const bool ahead = m->pf_built && m->pf_step == m->hstep && same_batch(m->pf_built_batch, *batch) && m->lazy;
        m->pf_built = false;
        if (ahead) { std::swap(m->mark, m->mark2); std::swap(m->ulist, m->ulist2); std::swap(m->ucount, m->ucount2); std::swap(m->stamp, m->stamp2); }
        constexpr bool fold_ok = true;
        fold_advance = fold_ok && ahead;          // (as in aae_step: nothing between the bookkeeping and the gather)
        if (!fold_advance)
        hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, m->sc, m->step_ctr, m->lazy ? m->tab : nullptr,
                           ahead ? (int*)nullptr : m->stamp, ahead ? (int*)nullptr : m->ucount, m->losses);
        LAUNCHCHK("advance_step");
        if (m->lazy && !ahead) TRY(lazy_prepare(m, -1, false, s));
        m->enc_bwd_done = false; m->fuse_enc_bwd = false; m->dense_step = false;
        // The output layer's tile buckets depend on the batch only: they are built on the side stream beside this
        // handle's list building and gather (and the caller's forward pass) instead of in front of the critical launch.
        // The side stream is in order behind the last deferred launch, which waited for the last critical launch - the
        // last reader of the bucket arrays; without such a launch to order it the build stays where it was.
        constexpr bool bk_ahead = true;
        if (bk_ahead && m->side && m->ev_bk && m->last_out_split && fused_decoder_applies(m)) {
            TRY(build_tile_buckets(m, m->side));
            HIPCHK(hipEventRecord(m->ev_bk, m->side));
            m->bk_pending = true;
        }
        // (a batch named with aae_prefetch_batch stays armed: its list and catch-up are enqueued behind this step's
        //  deferred optimiser launch - aae_output_layer_step - where they need no mark on this stream; a mark riding on
        //  the gather below cost the stream more than the 16 us it moved away)
        if (m->pf_armed && !(m->side && m->mark2 && m->lazy)) m->pf_armed = false;
    } else if (!m->have_batch) {
        return fail(AAE_ESTATE, "aae_first_layer_forward(batch = NULL) without a running batch");
    }
    {
        ProfScope ps(m, AAE_K_ENC_GATHER, s);
        const size_t shm = (size_t)16 * r4(m->h) * sizeof(float);
        DropSpec none; memset(&none, 0, sizeof(none));
        // (four waves per document for wide batches, as gather_first_layer: an item slice holds 1 / world of a document's entries)
        constexpr int g4_rows = 512;
        if (m->rows >= g4_rows)
        hipExtLaunchKernelGGL(enc_gather_kernel_t<4>, dim3(m->rows), dim3(256), (uint32_t)(shm / 4), s, nullptr, pf ? m->ev_head : nullptr, 0,
                              m->bv, (const float*)m->P[P_W1T].p, m->ldw1, bias_dev, m->h, (int)m->cfg.normalize_inputs,
                              m->a1.p, (float*)nullptr, m->ldh, (int)m->cfg.activation, none, (uint64_t)m->cfg.seed,
                              (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                              AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, fold_advance ? 1 : 0},
                              (long long)(fold_advance ? m->hstep : -1));
        else
        hipExtLaunchKernelGGL(enc_gather_kernel, dim3(m->rows), dim3(1024), (uint32_t)shm, s, nullptr, pf ? m->ev_head : nullptr, 0,
                              m->bv, (const float*)m->P[P_W1T].p, m->ldw1, bias_dev, m->h, (int)m->cfg.normalize_inputs,
                              m->a1.p, (float*)nullptr, m->ldh, (int)m->cfg.activation, none, (uint64_t)m->cfg.seed,
                              (const long long*)m->step_ctr, m->rscale, m->doc_l1,
                              AdvanceJob{m->sc, m->step_ctr, m->lazy ? m->tab : nullptr, m->losses, fold_advance ? 1 : 0},
                              (long long)(fold_advance ? m->hstep : -1));
        LAUNCHCHK("enc_gather (partial)");
    }
    if (pf) TRY(launch_prefetch(m));
    if (batch) { m->phase = 1; m->dec_hidden_done = true; }       // aae_output_layer_step(batch = NULL) may follow on this handle
    return AAE_OK;
}

// The first layer's weight gradient from dL/d(a1) of the running batch (ga1_dev [rows][ld], NULL = AAE_T_ACT_GA1 of this
// handle; rows_per_block > 0: blocks of that many rows, block_stride floats apart - the ranks' packets of an all-gather
// read where they landed) restricted to this handle's items, and optimiser `which` (enc_optim 0 / gen_optim 2) on its rows.  (The bias is
// a small replicated parameter: its gradient is a column sum of the replicas' own dL/d(a1), aae_ae_backward / aae_gen_step
// export it with the other small layers'.)
int aae_first_layer_update(aae_handle m, const float* ga1_dev, int64_t ld, int32_t rows_per_block, int64_t block_stride,
                           int which, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->vae || m->cfg.grad_mode != AAE_GRAD_FUSED) return fail(AAE_ESTATE, "aae_first_layer_update: fused optimiser, no VAE mode");
    if (which != O_ENC && which != O_GEN) return fail(AAE_EINVAL, "which must be enc_optim (0) or gen_optim (2)");
    if (!m->have_batch) return fail(AAE_ESTATE, "aae_first_layer_update without a running batch");
    if (ga1_dev && ld != m->ldh) return fail(AAE_EINVAL, "aae_first_layer_update: ld must equal the leading dimension of AAE_T_ACT_GA1");
    if (rows_per_block < 0 || (rows_per_block > 0 && (!ga1_dev || block_stride < (int64_t)rows_per_block * ld)))
        return fail(AAE_EINVAL, "aae_first_layer_update: blocks need ga1_dev and block_stride >= rows_per_block * ld");
    hipStream_t s = S(stream);
    const float* ga1 = ga1_dev ? ga1_dev : m->gb3.p;
    return launch_w1_items(m, ga1, (int)rows_per_block, (size_t)block_stride, which, s);
}

// disc_step (aae.py:713-732)
int aae_disc_step(aae_handle m, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->ae_only) return fail(AAE_ESTATE, "model was created as a plain autoencoder (no discriminator steps)");
    if (m->phase != 3) return fail(AAE_ESTATE, "aae_disc_step before the ae phases of the step");
    remember_inject(m, inj, false);
    hipStream_t s = S(stream);
    const int B = m->rows, h = m->h, cc = m->c;
    const aae_rng_inject& I = m->inj;
    const float pscale = m->cfg.has_prior_scale ? m->cfg.prior_scale : 1.0f;
    // ---- disc_step: z_real rows [0,B), z_fake = Enc_eval(X) rows [B,2B)
    if (m->cfg.rng_mode != AAE_RNG_DEVICE && !I.z_real_dev) return fail(AAE_EINVAL, "rng_mode=inject needs z_real_dev");
    if (!m->use_chain) {                 // (the layer-chain program draws / copies z_real itself: COP_PRIOR)
        if (m->cfg.rng_mode == AAE_RNG_DEVICE) {
            hipLaunchKernelGGL(prior_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, m->zin.p, m->ldz, B, cc,
                               m->cfg.prior, pscale, m->cfg.seed, m->step_ctr, m->rng_row0);
        } else {
            hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, I.z_real_dev, cc, m->zin.p,
                               m->ldz, B, cc, pscale);
        }
        LAUNCHCHK("prior");
    }
    if (m->use_chain) {
        TRY(chain_disc_step(m, s));
        m->phase = 4;
        return AAE_OK;
    }
    TRY(encoder_forward(m, false, nullptr, nullptr, 0, 0, false, m->zin.p + (size_t)B * m->ldz, m->ldz, s));
    TRY(disc_forward(m, 2 * B, I.masks_dev[4], I.masks_dev[6], I.masks_dev[5], I.masks_dev[7], B, 4, 5, s));
    hipLaunchKernelGGL(adv_loss_kernel, dim3(1), dim3(256), 0, s, m->dout.p, 4, B, 0, m->grad_scale, m->ga3.p, 4,
                       m->losses, 1);
    LAUNCHCHK("adv_loss disc");
    {
        DropSpec d1 = make_drop(m, 0, true, I.masks_dev[4], I.masks_dev[6], B, h, 4);
        DropSpec d2 = make_drop(m, 1, true, I.masks_dev[5], I.masks_dev[7], B, h, 5);
        EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->xh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
        b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
        TRY(linear_dx(m->ga3.p, 4, 2 * B, m->P[P_D3], h, b2, s));
        TRY(linear_dw(m, m->ga3.p, 4, 2 * B, m->xh2.p, m->ldh, P_D3, O_DISC, s));
        EpiActBwd b1 = b2; b1.out = m->gb1.p; b1.y = m->xh1.p; b1.d = d1;
        TRY(linear_dx(m->gb0.p, m->ldh, 2 * B, m->P[P_D2], h, b1, s, gmode(m)));
        TRY(linear_dw(m, m->gb0.p, m->ldh, 2 * B, m->xh1.p, m->ldh, P_D2, O_DISC, s));
        TRY(linear_dw(m, m->gb1.p, m->ldh, 2 * B, m->zin.p, m->ldz, P_D1, O_DISC, s));
    }
    m->phase = 4;
    return AAE_OK;
}

// gen_step (aae.py:734-743): Enc_train(X) - layer-1 pre-activations are unchanged since the
// disc_step forward, so m->a1 is re-used - then D on rows [0,B)
int aae_gen_step(aae_handle m, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->phase != 4) return fail(AAE_ESTATE, "aae_gen_step before aae_disc_step");
    remember_inject(m, inj, false);
    hipStream_t s = S(stream);
    const int B = m->rows, h = m->h, cc = m->c;
    const aae_rng_inject& I = m->inj;
    if (m->use_chain) {
        TRY(chain_gen_step(m, s));
        m->phase = 0;
        return AAE_OK;
    }
    TRY(encoder_forward(m, true, I.masks_dev[8], I.masks_dev[9], 8, 9, true, m->zin.p, m->ldz, s));
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)B * cc)), dim3(256), 0, s, m->zin.p, m->ldz, m->zsave.p,
                       m->ldz, B, cc, 1.0f);
    TRY(disc_forward(m, B, I.masks_dev[10], nullptr, I.masks_dev[11], nullptr, B, 10, 11, s));
    hipLaunchKernelGGL(adv_loss_kernel, dim3(1), dim3(256), 0, s, m->dout.p, 4, B, 1, m->grad_scale, m->ga3.p, 4,
                       m->losses, 2);
    LAUNCHCHK("adv_loss gen");
    {
        DropSpec d1 = make_drop(m, 0, true, I.masks_dev[10], nullptr, B, h, 10);
        DropSpec d2 = make_drop(m, 1, true, I.masks_dev[11], nullptr, B, h, 11);
        EpiActBwd b2; b2.out = m->gb0.p; b2.ld = m->ldh; b2.y = m->xh2.p; b2.ldy = m->ldh; b2.act = m->cfg.activation;
        b2.d = d2; b2.seed = m->cfg.seed; b2.step_ctr = m->step_ctr;
        TRY(linear_dx(m->ga3.p, 4, B, m->P[P_D3], h, b2, s));
        EpiActBwd b1 = b2; b1.out = m->gb1.p; b1.y = m->xh1.p; b1.d = d1;
        TRY(linear_dx(m->gb0.p, m->ldh, B, m->P[P_D2], h, b1, s, gmode(m)));
        EpiStore ez; ez.out = m->gzc.p; ez.ld = m->ldc;
        TRY(linear_dx(m->gb1.p, m->ldh, B, m->P[P_D1], cc, ez, s, gmode(m)));
    }
    TRY(encoder_backward(m, m->gzc.p, m->ldc, m->zsave.p, m->ldz, I.masks_dev[8], I.masks_dev[9], 8, 9, O_GEN, s));
    m->phase = 0;
    return AAE_OK;
}

int aae_disc_gen(aae_handle m, const aae_rng_inject* inj, void* stream) {
    TRY(aae_disc_step(m, inj, stream));
    return aae_gen_step(m, nullptr, stream);
}

int aae_step(aae_handle m, const aae_batch* batch, const float* cond_dev, const aae_rng_inject* inj, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->ext_first) return fail(AAE_ESTATE, "aae_step: the first layer is external (aae_set_first_layer_external): drive the phases");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    hipStream_t s = S(stream);
    TRY(ae_encode_impl(m, batch, inj, nullptr, true, cond_dev, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, s, cond_dev,
                           m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    m->fuse_enc_bwd = true;
    int rc = aae_ae_decode_backward(m, nullptr, 0, nullptr, nullptr, stream);
    m->fuse_enc_bwd = false;
    TRY(rc);
    TRY(aae_ae_encoder_backward(m, nullptr, 0, stream));
    if (!m->ae_only) TRY(aae_disc_gen(m, nullptr, stream));
    return AAE_OK;
}

int aae_read_losses(aae_handle m, float out[3], void* stream) {
    if (!m || !out) return fail(AAE_EINVAL, "handle/out is NULL");
    float tmp[4];
    HIPCHK(hipMemcpyAsync(tmp, m->losses, sizeof(tmp), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    out[0] = tmp[0]; out[1] = tmp[1]; out[2] = tmp[2];
    return AAE_OK;
}


}  // extern "C"
