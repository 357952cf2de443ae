// The REAL grouped_dw_kernel (csrc/chain.h) on synthetic jobs: `njobs` layers of 200 x 201 over `rows` rows (+ the first
// layer's bias column sums), G and X rewritten by another kernel between launches (as the chain program in front of the real
// launch does), timed per launch by an event pair on the launch and - built with -DDW_TS - by in-kernel stamps of tile 0 and
// of the first column-sum block.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DDW_TS -I../../../aae-recommender_amd/csrc -o dw_real dw_real.hip
//   ./dw_real <rows> <jobs> <colsum 0|1> <0> <k-split threshold, 0 = never> <1: G and X written once (they stay in the L2s)>
// ablations of chain.h: -DDW_NO_OPT_PREFETCH
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm_f32.h"
#include "kernels.h"
#include "chain.h"
using namespace aae;

__global__ void rewrite_kernel(float* a, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v + (float)(i & 15) * 1e-3f;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 800, njobs = argc > 2 ? atoi(argv[2]) : 3, colsum = argc > 3 ? atoi(argv[3]) : 1;
    const int wide = argc > 4 ? atoi(argv[4]) : 0, ksplit = argc > 5 ? atoi(argv[5]) : 0, norewrite = argc > 6 ? atoi(argv[6]) : 0;
    const int M = 200, N = 201, ld = 208, ldp = 208;
    float *G, *X, *p, *mo, *vo; OptScalars* sc;
    const size_t act = (size_t)rows * ld;
    hipMalloc(&G, act * 4 * 5); hipMalloc(&X, act * 4 * 5);
    hipMalloc(&p, (size_t)5 * M * ldp * 4); hipMalloc(&mo, (size_t)5 * M * ldp * 4); hipMalloc(&vo, (size_t)5 * M * ldp * 4);
    hipMemset(p, 0, (size_t)5 * M * ldp * 4); hipMemset(mo, 0, (size_t)5 * M * ldp * 4); hipMemset(vo, 0, (size_t)5 * M * ldp * 4);
    OptScalars hs; memset(&hs, 0, sizeof(hs)); hs.t = 1; hs.neg_step_size = -1e-3f; hs.bc2_sqrt = 0.0316f; hs.inv_bc2_sqrt = 31.6f; hs.lr = 1e-3;
    hipMalloc(&sc, sizeof(hs)); hipMemcpy(sc, &hs, sizeof(hs), hipMemcpyHostToDevice);
    DwGroup g; memset(&g, 0, sizeof(g));
    int tiles = 0;
    for (int j = 0; j < njobs; ++j) {
        DwJob& J = g.jobs[g.njobs++];
        J.G = G + j * act; J.ldg = ld; J.X = X + j * act; J.ldx = ld; J.rows = rows; J.M = M; J.N = N;
        J.p = p + (size_t)j * M * ldp; J.m = mo + (size_t)j * M * ldp; J.v = vo + (size_t)j * M * ldp; J.ld = ldp; J.sc = sc;
        J.tile0 = tiles; J.tiles_n = (N + 31) / 32; tiles += ((M + 31) / 32) * J.tiles_n;
    }
    int blocks = tiles;
    g.ksplit = ksplit;
    if (colsum) {
        W1Job& w = g.w1; w.enabled = 1; w.ga1 = G + 4 * act; w.ld = ld; w.h = M; w.rows = rows;
        w.bp = p + (size_t)4 * M * ldp; w.bm = mo + (size_t)4 * M * ldp; w.bv1 = vo + (size_t)4 * M * ldp; w.sc = sc;
        w.ncol = (M + 63) / 64; w.nitem = 0; w.blk0 = tiles; blocks += w.ncol;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int it = 0; it < 60; ++it) {
        if (!norewrite || it == 0) {
            hipLaunchKernelGGL(rewrite_kernel, dim3(256), dim3(1024), 0, 0, G, act * 5, 0.01f * (it & 3));
            hipLaunchKernelGGL(rewrite_kernel, dim3(256), dim3(1024), 0, 0, X, act * 5, 0.02f * (it & 3));
        }
        hipExtLaunchKernelGGL(grouped_dw_kernel, dim3(blocks), dim3(256), 0, 0, e0, e1, 0, g);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 10) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    printf("rows %d, %d jobs (%d tiles), colsum %d, ksplit %d, %s: median %.2f us per launch (min %.2f)\n", rows, njobs, tiles, colsum, ksplit, wide ? "16-wave form" : "4-wave form", us[us.size() / 2], us[0]);
#ifdef DW_TS
    unsigned long long h[96]; hipMemcpyFromSymbol(h, HIP_SYMBOL(dw_ts), sizeof(h));
    const int nslab = (rows + 63) / 64;
    if (!wide && ksplit) printf("  tile 0: start -> loop %.2f | loop %.2f | partial tiles, optimiser, stores %.2f\n", (h[1] - h[0]) * 0.01, (h[44] - h[1]) * 0.01, (h[46] - h[44]) * 0.01);
    if (!wide && !ksplit) {
        printf("  tile 0: start -> loop %.2f us | slabs:", (h[1] - h[0]) * 0.01);
        for (int i = 0; i < std::min(nslab, 40); ++i) printf(" %.2f", ((i + 1 < nslab ? h[3 + i] : h[44]) - h[2 + i]) * 0.01);
        printf(" | partial tile -> LDS %.2f | optimiser + stores %.2f | total %.2f\n", (h[45] - h[44]) * 0.01, (h[46] - h[45]) * 0.01, (h[46] - h[0]) * 0.01);
        printf("  start: job found %.2f | its tile %.2f | first rows requested %.2f | optimiser's operands requested %.2f\n", (h[52] - h[0]) * 0.01, (h[53] - h[52]) * 0.01, (h[54] - h[53]) * 0.01, (h[1] - h[54]) * 0.01);
        if (nslab > 5) printf("  slab 4: barrier %.2f | operands -> LDS %.2f | next slab's requests %.2f | barrier %.2f | products %.2f\n", (h[48] - h[6]) * 0.01, (h[49] - h[48]) * 0.01,
                              (h[50] - h[49]) * 0.01, (h[51] - h[50]) * 0.01, (h[7] - h[51]) * 0.01);
        if (colsum) printf("  column sums: %.2f us, starting %.2f us after tile 0\n", (h[65] - h[64]) * 0.01, ((double)h[64] - (double)h[0]) * 0.01);
    }
#endif
    return 0;
}
