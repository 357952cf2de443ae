// Counting sort of the batch's CSR entries into the 32-item tiles of the fused decoder output layer
// (dec_fused.h): per tile the entries (doc, item-in-tile, value) of the batch, in tstart[] / eb / en / ev.
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace aae {

constexpr int kTI = 32;        // items per tile of the fused decoder
constexpr int kMB = 7;         // 16-row blocks of the batch dimension the fused decoder handles (B <= 112)

// The same counting sort in ONE launch for batches whose tile counters fit the LDS of a workgroup (the
// four kernels above are ~5 us of launch latency each and sit on the step's critical path).  Document
// bounds go to LDS first so that the entries form one flat index space: every thread then needs two
// independent global round trips per pass (indices, values) instead of a chain through indptr.
constexpr int kBucketMaxDocs = 16 * kMB;     // the fused decoder's batch limit
constexpr int kBucketMaxTiles = 32 * 1024;   // LDS: (tiles + 1 + docs + 1 + 1024) ints <= 160 KB

// One 1024-thread workgroup; bk_lds: (ntiles + 1 + kBucketMaxDocs + 1 + 1024) ints of LDS.  Runs either as a
// kernel of its own or as an extra workgroup of the step's first layer-chain launch (chain.h), where it costs
// no launch and no time on the critical path (the chain's workgroups occupy a handful of CUs).
// MAXD: documents the instance can take (kBucketMaxDocs, or kBucketWideDocs = 1024 for the global batch an item slice of
// the vocabulary-sharded scheme sees: 8 ranks x 100 documents)
constexpr int kBucketWideDocs = 1024;
template <int MAXD = kBucketMaxDocs, int NT = 1024>     // NT = threads of the calling workgroup (>= MAXD)
__device__ __forceinline__ void tile_bucket_body(const BatchView& bv, int ntiles, int* __restrict__ tstart,
                                                 int* __restrict__ eb, int* __restrict__ en, float* __restrict__ ev,
                                                 int* bk_lds) {
    int* cnt = bk_lds;                               // [ntiles + 1]  histogram, then fill cursor
    int* dbeg = cnt + ntiles + 1;                    // [docs + 1]    first flat entry of each document
    int* part = dbeg + MAXD + 1;                     // [NT]          scan scratch
    __shared__ long long dlo[MAXD];                  // CSR offset of each document's first entry
    const int t = threadIdx.x, docs = bv.n_rows;
    for (int i = t; i <= ntiles; i += NT) cnt[i] = 0;
    int len = 0;
    if (t < docs) {
        const int dc = bv.doc(t);
        const long long lo = bv.indptr[dc];
        dlo[t] = lo;
        len = (int)(bv.indptr[dc + 1] - lo);
    }
    // exclusive scan of the document lengths (docs <= 112 <= 2 waves): plain Hillis-Steele in LDS
    part[t] = len;
    __syncthreads();
    for (int o = 1; o < MAXD; o <<= 1) {
        const int v = (t >= o && t < MAXD) ? part[t - o] : 0;
        __syncthreads();
        if (t < MAXD) part[t] += v;
        __syncthreads();
    }
    if (t < docs) dbeg[t] = part[t] - len;
    if (t == 0) dbeg[docs] = part[MAXD - 1];
    __syncthreads();
    const int total = dbeg[docs];
    auto doc_of = [&](int f) {                       // largest d with dbeg[d] <= f
        int lo = 0, hi = docs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (dbeg[mid] <= f) lo = mid; else hi = mid - 1; }
        return lo;
    };
    // Both passes take U entries per thread and round: their document searches (LDS) and their index / value loads are in
    // flight together - one entry per round was a dependent chain of a search and a global round trip per entry, 84 us for
    // 1000 documents x 4.6 k items in one workgroup (r3, C4's shape).
    constexpr int U = MAXD > 16 * kMB ? 8 : 2;
    // pass 1: histogram over the tiles
    for (int f0 = t; f0 < total; f0 += NT * U) {
        int idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(f0 + u * NT, total - 1);
            const int d = doc_of(f);
            idx[u] = bv.indices[dlo[d] + (f - dbeg[d])];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (f0 + u * NT < total) atomicAdd(&cnt[idx[u] / kTI], 1);
    }
    __syncthreads();
    // exclusive scan of the tile counters -> tstart (global) and the fill cursors (LDS)
    const int per = (ntiles + NT - 1) / NT;
    const int lo = min(t * per, ntiles), hi = min(ntiles, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        tstart[i] = run;
        cnt[i] = run;
        run += c;
    }
    if (t == NT - 1) tstart[ntiles] = part[NT - 1];
    __syncthreads();
    // pass 2: fill
    for (int f0 = t; f0 < total; f0 += NT * U) {
        int idx[U], dd[U];
        float val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(f0 + u * NT, total - 1);
            const int d = doc_of(f);
            const long long e = dlo[d] + (f - dbeg[d]);
            dd[u] = d; idx[u] = bv.indices[e]; val[u] = bv.values[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (f0 + u * NT < total) {
                const int tile = idx[u] / kTI;
                const int pos = atomicAdd(&cnt[tile], 1);
                eb[pos] = dd[u]; en[pos] = idx[u] - tile * kTI; ev[pos] = val[u];
            }
    }
}

__global__ __launch_bounds__(1024) void tile_bucket_kernel(BatchView bv, int ntiles, int* __restrict__ tstart,
                                                           int* __restrict__ eb, int* __restrict__ en,
                                                           float* __restrict__ ev) {
    extern __shared__ int bk_lds_dyn[];
    tile_bucket_body<kBucketMaxDocs>(bv, ntiles, tstart, eb, en, ev, bk_lds_dyn);
}
// ... for up to 1024 documents; LDS: (ntiles + 1 + kBucketWideDocs + 1 + 1024) ints
__global__ __launch_bounds__(1024) void tile_bucket_wide_kernel(BatchView bv, int ntiles, int* __restrict__ tstart,
                                                                int* __restrict__ eb, int* __restrict__ en,
                                                                float* __restrict__ ev) {
    extern __shared__ int bk_lds_dyn[];
    // This ONE workgroup runs 50-110 us on the side stream, and whatever lands beside it on its CU crawls: at C4 the next gather of
    // the main stream took 21-35 us instead of 9 - or, with the CU's LDS claimed (abi_chains.h), torch's index_select of the
    // condition rows, which needs no LDS, 30 instead of 5.  128 registers per wave = the CU's whole register file for its 16
    // waves: nothing else is dealt onto the CU while it runs.
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    tile_bucket_body<kBucketWideDocs>(bv, ntiles, tstart, eb, en, ev, bk_lds_dyn);
}

struct BucketJob {           // piggy-backed on a chain launch when enabled
    BatchView bv; int ntiles; int* tstart; int* eb; int* en; float* ev; int enabled;
};

}  // namespace aae
