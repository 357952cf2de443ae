// CategoricalCondition on the device (reference aaerec/condition.py:397-508): a trainable embedding of a
// categorical attribute (authors, venue ...) whose per-document rows are reduced and concatenated to the code.
//
//   encode   out[r][0:dim] = reduce_w table[idx[r][w]]          nn.Embedding + .sum(1) / .mean(1)   (condition.py:476-489)
//   update   the embedding's backward and its optimiser step    zero_grad / loss.backward / step    (condition.py:491-497)
//
// Index 0 is the padding / out-of-vocabulary token: it reads as the zero row and receives no gradient
// (padding_idx=0).  A batch holds rows * width indices (a few hundred): one wavefront per index slot, no atomics -
// the first slot that names a table row owns it, sums the gradient of every slot naming the same row in slot
// order and applies the optimiser to that row, so results do not depend on scheduling.
#pragma once
#include "device_common.h"

namespace aae {

constexpr int kCatMaxDim = 256;       // 4 columns per lane
constexpr int kCatWaves = 4;          // wavefronts per workgroup

__global__ void __launch_bounds__(64)
cat_encode_kernel(const float* __restrict__ table, int vocab, int dim, const int* __restrict__ idx, int width,
                  int mean, float* __restrict__ out, long long ldo) {
    const int r = blockIdx.x, col = blockIdx.y * 64 + threadIdx.x;
    if (col >= dim) return;
    float acc = 0.f;
    for (int w = 0; w < width; ++w) {
        const int j = idx[(size_t)r * width + w];
        if (j > 0 && j < vocab) acc += table[(size_t)j * dim + col];
    }
    out[(size_t)r * ldo + col] = mean ? acc / (float)width : acc;
}

struct CatUpdate {
    float* table; float* m; float* v;
    float* gdense;              // dense Adam: [vocab][dim] gradient scratch (all zero between steps); NULL = SparseAdam
    const int* idx; const float* d;
    long long ldd;
    int vocab, dim, rows, width, mean;
    float neg_step_size;        // SparseAdam: -(lr * sqrt(1 - b2^t) / (1 - b1^t))
};

__global__ void __launch_bounds__(64 * kCatWaves)
cat_update_kernel(CatUpdate a) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * kCatWaves + (threadIdx.x >> 6);      // this wavefront's index slot
    const int n = a.rows * a.width;
    if (e >= n) return;
    const int j = a.idx[e];
    if (j <= 0 || j >= a.vocab) return;
    // an earlier slot with the same row owns it
    int dup = 0;
    for (int e2 = lane; e2 < e; e2 += 64) dup |= (a.idx[e2] == j);
    if (__any(dup)) return;
    // coalesced gradient of row j: every slot naming it, in slot order (the sparse gradient's coalesce())
    float g[kCatMaxDim / 64];
#pragma unroll
    for (int k = 0; k < kCatMaxDim / 64; ++k) g[k] = 0.f;
    for (int base = e & ~63; base < n; base += 64) {
        const int e2 = base + lane;
        const int hit = (e2 >= e && e2 < n) ? (a.idx[e2] == j) : 0;
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int b = (base + __builtin_ctzll(mask)) / a.width;
            mask &= mask - 1;
#pragma unroll
            for (int k = 0; k < kCatMaxDim / 64; ++k) {
                const int col = lane + 64 * k;
                if (col < a.dim) {
                    const float dv = a.d[(size_t)b * a.ldd + col];
                    g[k] += a.mean ? dv / (float)a.width : dv;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kCatMaxDim / 64; ++k) {
        const int col = lane + 64 * k;
        if (col >= a.dim) continue;
        const size_t o = (size_t)j * a.dim + col;
        if (a.gdense) { a.gdense[o] = g[k]; continue; }       // dense Adam follows over the whole table
        // torch.optim.SparseAdam, in its operation order (torch/optim/_functional.py sparse_adam)
        const float m_old = a.m[o], v_old = a.v[o];
        const float mu = (g[k] - m_old) * 0.1f;
        const float vu = (g[k] * g[k] - v_old) * 0.001f;
        const float m_new = m_old + mu, v_new = v_old + vu;
        a.m[o] = m_new; a.v[o] = v_new;
        a.table[o] += a.neg_step_size * (m_new / (sqrtf(v_new) + 1e-8f));
    }
}

// torch.optim.Adam over the whole table (nn.Embedding(sparse=False)): rows without a gradient still decay their
// moments and move.  Consumes and clears the gradient scratch.
__global__ void __launch_bounds__(256)
cat_dense_adam_kernel(float* __restrict__ table, float* __restrict__ m, float* __restrict__ v, float* __restrict__ g,
                      size_t n, OptScalars sc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    if (gi != 0.f) g[i] = 0.f;
    float p = table[i], mi = m[i], vi = v[i];
    // exact sqrt / division here (adam_update's 1-ulp approximations buy nothing on a table this small)
    mi = mi + 0.1f * (gi - mi);
    vi = vi * 0.999f + (0.001f * gi) * gi;
    const float denom = sqrtf(vi) / sc.bc2_sqrt + 1e-8f;
    p = p + sc.neg_step_size * (mi / denom);
    table[i] = p; m[i] = mi; v[i] = vi;
}

// Weighted bag of embedded words (reference ub.py:52-57, `sparse_scores @ self.embedding`): out[r] = sum over the
// CSR entries e of row r of values[e] * table[indices[e]], entries in CSR order (scipy's order).  One workgroup per
// document, a thread per column (coalesced 1 KB row segments); the TF-IDF rows of a title hold 5-20 words.
__global__ void __launch_bounds__(256)
csr_embed_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ values,
                 const float* __restrict__ table, int n_table_rows, int dim, long long ldt, float* __restrict__ out,
                 long long ldo) {
    const int r = blockIdx.x;
    const long long e0 = indptr[r], e1 = indptr[r + 1];
    for (int col = threadIdx.x; col < dim; col += 256) {
        float acc = 0.f;
        for (long long e = e0; e < e1; ++e) {
            const int j = indices[e];
            if (j >= 0 && j < n_table_rows) acc += values[e] * table[(size_t)j * ldt + col];
        }
        out[(size_t)r * ldo + col] = acc;
    }
}

}  // namespace aae
