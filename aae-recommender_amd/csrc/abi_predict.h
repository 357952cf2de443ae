// predict (aae.py:840-870): encode, decode, predict, on-device top-k.
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

// ---- predict (aae.py:840-870) -----------------------------------------------------------
int aae_encode(aae_handle m, const aae_batch* batch, float* z_out, void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    TRY(set_batch(m, batch));
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (m->lazy) TRY(lazy_prepare(m, 0, true, s));
    if (m->use_chain) {
        TRY(gather_first_layer(m, false, nullptr, 0, s));
        ChainBuilder cb(m, m->rows);
        chain_encoder_tail(m, cb, false, nullptr, 0, m->rows, nullptr, s);
        ChainOp& f = cb.add(cop(COP_FINAL_FWD, 2, 2, m->c)); f.aux = m->cfg.enc_final; cop_out(f, m->zc.p, m->ldc);
        if (z_out) { f.out2 = z_out; f.ldo2 = m->c; }
        TRY(launch_chain(m, cb, s));
        m->phase = 0;
        return AAE_OK;
    }
    TRY(encoder_forward(m, false, nullptr, nullptr, 0, 0, false, m->zc.p, m->ldc, s));
    if (z_out) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->c)), dim3(256), 0, s, m->zc.p, m->ldc,
                           z_out, m->c, m->rows, m->c, 1.0f);
        LAUNCHCHK("copy z");
    }
    m->phase = 0;
    return AAE_OK;
}

int aae_decode(aae_handle m, const float* zc_dev, int64_t zc_ld, int32_t n_rows, float* out_dev, int64_t out_ld,
               void* stream) {
    if (!m || !out_dev) return fail(AAE_EINVAL, "handle/out is NULL");
    if (n_rows < 1 || n_rows > m->R) return fail(AAE_EINVAL, "n_rows outside [1, max_batch]");
    if (out_ld < m->N || (out_ld & 3) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return fail(AAE_EINVAL, "out_dev must be 16-byte aligned with out_ld >= n_items and out_ld % 4 == 0");
    hipStream_t s = S(stream);
    TRY(join_deferred(m, s));
    if (zc_dev) TRY(stage_zc(m, zc_dev, zc_ld, n_rows, s));
    if (m->vae) TRY(chain_vae_dec_hidden(m, n_rows, s));       // (VAE: one hidden layer, fc3)
    else if (m->use_chain) TRY(chain_dec_hidden(m, false, n_rows, s));
    else TRY(decoder_hidden_forward(m, false, nullptr, nullptr, n_rows, s));
    EpiSigmoid e; e.out = out_dev; e.ld = (int)out_ld;
    TRY(linear_fwd(m->dh2.p, m->ldh, n_rows, m->P[P_V3], e, s, gmode(m)));
    return AAE_OK;
}

int aae_predict(aae_handle m, const aae_batch* batch, const float* cond_dev, float* out_dev, int64_t out_ld,
                void* stream) {
    if (!m) return fail(AAE_EINVAL, "handle is NULL");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    TRY(aae_encode(m, batch, nullptr, stream));
    if (m->cfg.cond_inc > 0) {
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid1d((size_t)m->rows * m->cfg.cond_inc)), dim3(256), 0, S(stream),
                           cond_dev, m->cfg.cond_inc, m->zc.p + m->c, m->ldc, m->rows, m->cfg.cond_inc, 1.0f);
        LAUNCHCHK("copy cond");
    }
    return aae_decode(m, nullptr, 0, m->rows, out_dev, out_ld, stream);
}

// predict + on-device remove_non_missing / argtopk (evaluation.py:183-199, 20-58): only the k best
// items per row (ids and min-max-scaled scores) leave the GPU
int aae_predict_topk(aae_handle m, const aae_batch* batch, const float* cond_dev, int32_t k, int32_t exclude_known,
                     int32_t* idx_out_dev, float* val_out_dev, void* stream) {
    if (!m || !idx_out_dev || !val_out_dev) return fail(AAE_EINVAL, "NULL argument");
    if (k < 1 || k > 32 || k > m->N) return fail(AAE_EINVAL, "k must be in [1, min(32, n_items)]");
    if (m->cfg.cond_inc > 0 && !cond_dev) return fail(AAE_EINVAL, "cond_inc > 0 needs cond_dev");
    TRY(rank_check_batch(batch));
    if (batch->n_rows >= 1 && batch->n_rows <= rank_rows_cap(m, k)) {   // fused: no [rows][N] matrix (abi_rank.h)
        m->phase = 0;
        return rank_predict(m, batch, cond_dev, k, exclude_known, idx_out_dev, val_out_dev, S(stream));
    }
    TRY(aae_predict(m, batch, cond_dev, m->G.p, m->ldn, stream));      // scores into the [rows][N] scratch
    hipStream_t s = S(stream);
    if (k <= 10)
        hipLaunchKernelGGL(topk_rows_kernel<10>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else if (k <= 20)
        hipLaunchKernelGGL(topk_rows_kernel<20>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else
        hipLaunchKernelGGL(topk_rows_kernel<32>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    LAUNCHCHK("topk_rows");
    return AAE_OK;
}

int aae_rank_max_rows(aae_handle m, int32_t k, int32_t* rows_out) {
    if (!m || !rows_out) return fail(AAE_EINVAL, "NULL argument");
    if (k < 1 || k > 32 || k > m->N) return fail(AAE_EINVAL, "k must be in [1, min(32, n_items)]");
    *rows_out = std::max(m->R, rank_rows_cap(m, k));
    return AAE_OK;
}

// the same for a caller-built decoder input (code | imposed conditions of any plugin kind): the second half of predict
// (aae.py:855-866) + remove_non_missing / argtopk; `batch` names the input rows whose items are excluded
int aae_decode_topk(aae_handle m, const float* zc_dev, int64_t zc_ld, const aae_batch* batch, int32_t k,
                    int32_t exclude_known, int32_t* idx_out_dev, float* val_out_dev, void* stream) {
    if (!m || !zc_dev || !idx_out_dev || !val_out_dev) return fail(AAE_EINVAL, "NULL argument");
    if (k < 1 || k > 32 || k > m->N) return fail(AAE_EINVAL, "k must be in [1, min(32, n_items)]");
    if (zc_ld < m->cp) return fail(AAE_EINVAL, "zc_ld < n_code + cond_inc");
    TRY(rank_check_batch(batch));
    if (batch->n_rows >= 1 && batch->n_rows <= rank_rows_cap(m, k)) {
        m->phase = 0;
        return rank_decode(m, zc_dev, zc_ld, batch, k, exclude_known, idx_out_dev, val_out_dev, S(stream));
    }
    TRY(set_batch(m, batch));
    TRY(aae_decode(m, zc_dev, zc_ld, m->rows, m->G.p, m->ldn, stream));   // scores into the [rows][N] scratch
    hipStream_t s = S(stream);
    if (k <= 10)
        hipLaunchKernelGGL(topk_rows_kernel<10>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else if (k <= 20)
        hipLaunchKernelGGL(topk_rows_kernel<20>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    else
        hipLaunchKernelGGL(topk_rows_kernel<32>, dim3(m->rows), dim3(256), 0, s, m->G.p, m->ldn, m->N, m->bv,
                           exclude_known, k, reinterpret_cast<int*>(idx_out_dev), val_out_dev);
    LAUNCHCHK("topk_rows");
    m->phase = 0;
    return AAE_OK;
}


}  // extern "C"
