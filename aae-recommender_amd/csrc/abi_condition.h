// Entry points of the trainable categorical condition and of the CSR helpers (cond_embed.h, kernels.h).
// (one of the parts of aae_abi.hip's translation unit: included there in order, not on its own)
#pragma once

extern "C" {

// ---- CategoricalCondition (cond_embed.h): stateless entry points over caller-owned tables --------------------
static int cat_check(int32_t vocab, int32_t dim, const int32_t* idx_dev, int32_t rows, int32_t width, int32_t reduce) {
    if (vocab < 1 || dim < 1 || dim > kCatMaxDim) return fail(AAE_EINVAL, "categorical condition: need vocab >= 1 and 1 <= dim <= 256");
    if (!idx_dev || rows < 1 || width < 1) return fail(AAE_EINVAL, "categorical condition: empty index block");
    if ((int64_t)rows * width > (1 << 22)) return fail(AAE_EINVAL, "categorical condition: rows * width > 2^22");
    if (reduce != AAE_CAT_SUM && reduce != AAE_CAT_MEAN) return fail(AAE_EINVAL, "categorical condition: reduce must be sum or mean");
    return AAE_OK;
}

int aae_cat_encode(const float* table_dev, int32_t vocab, int32_t dim, const int32_t* idx_dev, int32_t rows,
                   int32_t width, int32_t reduce, float* out_dev, int64_t out_ld, void* stream) {
    TRY(cat_check(vocab, dim, idx_dev, rows, width, reduce));
    if (!table_dev || !out_dev || out_ld < dim) return fail(AAE_EINVAL, "aae_cat_encode: table/out is NULL or out_ld < dim");
    hipLaunchKernelGGL(cat_encode_kernel, dim3(rows, (dim + 63) / 64), dim3(64), 0, S(stream), table_dev, vocab, dim,
                       idx_dev, width, reduce == AAE_CAT_MEAN, out_dev, (long long)out_ld);
    LAUNCHCHK("cat_encode");
    return AAE_OK;
}

int aae_cat_update(float* table_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* grad_scratch_dev, int32_t vocab,
                   int32_t dim, const int32_t* idx_dev, int32_t rows, int32_t width, int32_t reduce,
                   const float* dout_dev, int64_t dout_ld, int32_t optimizer, double lr, int64_t step, void* stream) {
    TRY(cat_check(vocab, dim, idx_dev, rows, width, reduce));
    if (!table_dev || !exp_avg_dev || !exp_avg_sq_dev || !dout_dev || dout_ld < dim)
        return fail(AAE_EINVAL, "aae_cat_update: table/state/dout is NULL or dout_ld < dim");
    if (optimizer != AAE_CAT_SPARSE_ADAM && optimizer != AAE_CAT_ADAM) return fail(AAE_EINVAL, "aae_cat_update: unknown optimizer");
    if (optimizer == AAE_CAT_ADAM && !grad_scratch_dev) return fail(AAE_EINVAL, "aae_cat_update: dense Adam needs grad_scratch_dev");
    if (step < 1) return fail(AAE_EINVAL, "aae_cat_update: step counts from 1");
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    CatUpdate a;
    a.table = table_dev; a.m = exp_avg_dev; a.v = exp_avg_sq_dev;
    a.gdense = optimizer == AAE_CAT_ADAM ? grad_scratch_dev : nullptr;
    a.idx = idx_dev; a.d = dout_dev; a.ldd = dout_ld; a.vocab = vocab; a.dim = dim; a.rows = rows; a.width = width;
    a.mean = reduce == AAE_CAT_MEAN;
    a.neg_step_size = (float)(-(lr * sqrt(bc2) / bc1));
    const int n = rows * width;
    hipLaunchKernelGGL(cat_update_kernel, dim3((n + kCatWaves - 1) / kCatWaves), dim3(64 * kCatWaves), 0, S(stream), a);
    LAUNCHCHK("cat_update");
    if (optimizer == AAE_CAT_ADAM) {
        OptScalars sc; memset(&sc, 0, sizeof(sc));
        sc.t = step; sc.neg_step_size = (float)(-(lr / bc1)); sc.bc2_sqrt = (float)sqrt(bc2); sc.lr = lr;
        sc.inv_bc2_sqrt = 1.0f / sc.bc2_sqrt;
        const size_t total = (size_t)vocab * dim;
        hipLaunchKernelGGL(cat_dense_adam_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), table_dev,
                           exp_avg_dev, exp_avg_sq_dev, grad_scratch_dev, total, sc);
        LAUNCHCHK("cat_dense_adam");
    }
    return AAE_OK;
}

int aae_csr_embed(const int64_t* indptr_dev, const int32_t* indices_dev, const float* values_dev, int32_t n_rows,
                  const float* table_dev, int32_t n_table_rows, int32_t dim, int64_t table_ld, float* out_dev,
                  int64_t out_ld, void* stream) {
    if (!indptr_dev || !table_dev || !out_dev) return fail(AAE_EINVAL, "aae_csr_embed: indptr/table/out is NULL");
    if (n_rows < 0 || n_table_rows < 1 || dim < 1 || table_ld < dim || out_ld < dim)
        return fail(AAE_EINVAL, "aae_csr_embed: bad shape (need n_table_rows >= 1, dim >= 1, leading dimensions >= dim)");
    if (n_rows == 0) return AAE_OK;
    if (!indices_dev || !values_dev) return fail(AAE_EINVAL, "aae_csr_embed: indices/values is NULL");
    hipLaunchKernelGGL(csr_embed_kernel, dim3(n_rows), dim3(256), 0, S(stream), (const long long*)indptr_dev, indices_dev,
                       values_dev, table_dev, n_table_rows, dim, (long long)table_ld, out_dev, (long long)out_ld);
    LAUNCHCHK("csr_embed");
    return AAE_OK;
}

int aae_dense_to_csr(const void* dense_dev, int32_t elem_bytes, int64_t ld, int32_t rows, int32_t n_cols,
                     int64_t* indptr_dev, int32_t* indices_dev, float* values_dev, int64_t capacity,
                     int32_t* scratch_dev, int32_t* stats_out_host, void* stream) {
    if (!dense_dev || !indptr_dev || !indices_dev || !values_dev || !scratch_dev || !stats_out_host)
        return fail(AAE_EINVAL, "aae_dense_to_csr: NULL argument");
    if (rows < 1 || n_cols < 1 || ld < n_cols || capacity < 1) return fail(AAE_EINVAL, "aae_dense_to_csr: bad shape");
    if (elem_bytes != 4 && elem_bytes != 8) return fail(AAE_EINVAL, "aae_dense_to_csr: elem_bytes must be 4 (float32) or 8 (float64)");
    hipStream_t s = S(stream);
    int* stats = scratch_dev;                 // [0..3] statistics, [8..] row counts
    int* rowcnt = scratch_dev + 8;
    HIPCHK(hipMemsetAsync(stats, 0, 8 * sizeof(int), s));
    if (elem_bytes == 4) hipLaunchKernelGGL(dense_count_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)dense_dev, (long long)ld, n_cols, rowcnt, stats);
    else hipLaunchKernelGGL(dense_count_kernel<double>, dim3(rows), dim3(256), 0, s, (const double*)dense_dev, (long long)ld, n_cols, rowcnt, stats);
    hipLaunchKernelGGL(dense_scan_kernel, dim3(1), dim3(1024), 0, s, rowcnt, rows, (long long)capacity, (long long*)indptr_dev, stats);
    if (elem_bytes == 4) hipLaunchKernelGGL(dense_fill_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)dense_dev, (long long)ld, n_cols, (const long long*)indptr_dev, stats, indices_dev, values_dev);
    else hipLaunchKernelGGL(dense_fill_kernel<double>, dim3(rows), dim3(256), 0, s, (const double*)dense_dev, (long long)ld, n_cols, (const long long*)indptr_dev, stats, indices_dev, values_dev);
    LAUNCHCHK("dense_to_csr");
    HIPCHK(hipMemcpyAsync(stats_out_host, stats, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AAE_OK;
}


}  // extern "C"
