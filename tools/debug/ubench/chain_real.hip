// The REAL chain4_kernel (csrc/chain4.h) on a synthetic program: LOAD + n x LINEAR(201 -> 200) on one set of weights, 25 workgroups,
// timed per launch and per op (in-kernel stamps) - to find where a layer op's 5 us go when its loads + products + partial sums
// take 1.7 us in isolation (chain_stream.hip).   Build: hipcc -O3 --offload-arch=gfx950 -I../../../aae-recommender_amd/csrc ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "gemm_f32.h"
#include "kernels.h"
#include "chain.h"
#include "chain4.h"
using namespace aae;

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 100, nlin = argc > 2 ? atoi(argv[2]) : 7, epi = argc > 3 ? atoi(argv[3]) : 0;
    const int store = argc > 4 ? atoi(argv[4]) : 0, nsets = argc > 5 ? atoi(argv[5]) : 8, want_ts = argc > 6 ? atoi(argv[6]) : 1;
    const int K = 201, N = 200;
    float *W4, *X, *out; long long* ctr; unsigned long long* ts; float* loss;
    hipMalloc(&W4, (size_t)(52 * N * 4 + 4 * N) * 4 * 8); hipMemset(W4, 0, (size_t)(52 * N * 4 + 4 * N) * 4 * 8);
    hipMalloc(&X, (size_t)rows * 256 * 4); hipMemset(X, 0, (size_t)rows * 256 * 4);
    hipMalloc(&out, (size_t)rows * 256 * 4 * 16);
    hipMalloc(&ctr, 8); hipMemset(ctr, 0, 8); hipMalloc(&ts, 128 * 8); hipMalloc(&loss, 64); hipMemset(loss, 0, 64);
    ChainProgram P; memset(&P, 0, sizeof(P));
    P.rows = rows; P.act = 1; P.seed = 1; P.step_ctr = ctr; P.loss_out = loss; P.loss_slot = 3; P.ts = want_ts ? ts : nullptr;
    auto& o0 = P.ops[P.nops++]; o0.kind = COP_LOAD; o0.dst = 0; o0.N = N; o0.W = X; o0.ldw = 256; o0.scale = 1.f; o0.one_col = N; o0.fake_slot = -1;
    for (int i = 0; i < nlin; ++i) {
        auto& o = P.ops[P.nops++];
        o.kind = COP_LINEAR; o.src = i & 1; o.dst = (i & 1) ^ 1; o.K = K; o.N = N;
        o.W4 = W4 + (size_t)(i % nsets) * (52 * N * 4 + 4 * N); o.ns4 = N; o.epi = epi; o.yslot = 2; o.one_col = N; o.fake_slot = -1;
        if (epi == CEPI_DROPACT) { o.d.enabled = 1; o.d.device_rng = 1; o.d.keep_threshold = 0x33333333u; o.d.mul_keep = 1.25f; o.d.split_row = 1 << 30; o.d.width = N; }
        if (store) { o.out = out + (size_t)i * rows * 256; o.ldo = 256; }
    }
    const int grid = (rows + 3) / 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define C4K(...) do { if (want_ts) hipLaunchKernelGGL((chain4_kernel<false, true>), __VA_ARGS__); else hipLaunchKernelGGL((chain4_kernel<false>), __VA_ARGS__); } while (0)
    const size_t lds = kCSlots * kCR * kCL * sizeof(float);
#ifdef C4_PROG_PTR
    ChainProgram* Pd; hipMalloc(&Pd, sizeof(P)); hipMemcpy(Pd, &P, sizeof(P), hipMemcpyHostToDevice);
#define PARG Pd
#else
#define PARG P
#endif
    for (int i = 0; i < 5; ++i) C4K(dim3(grid), dim3(kC4T), lds, 0, PARG);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) C4K(dim3(grid), dim3(kC4T), lds, 0, PARG);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[128]; hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    printf("rows %d, %d linear ops, epi %d, store %d: %.2f us per launch;  ops:", rows, nlin, epi, store, ms * 10.f);
    for (int i = 0; i < P.nops; ++i) printf(" %.2f", (h[i + 1] - h[i]) * 0.01);
    printf("\n  op 2, us after its start: loads issued");
    for (int w = 0; w < 16; w += 4) printf(" %.2f", ((double)h[64 + w] - (double)h[64 + 48]) * 0.01);
    printf(" | partial sums stored");
    for (int w = 0; w < 16; w += 4) printf(" %.2f", ((double)h[64 + 32 + w] - (double)h[64 + 48]) * 0.01);
    printf(" | matrix phase %.2f | epi ctx %.2f | epilogue %.2f | barrier %.2f | tail (one_col, stores, next op start) %.2f\n", (h[64 + 50] - h[64 + 48]) * 0.01, (h[64 + 52] - h[64 + 50]) * 0.01,
           (h[64 + 53] - h[64 + 52]) * 0.01, (h[64 + 54] - h[64 + 53]) * 0.01, ((double)h[3] - (double)h[64 + 54]) * 0.01);
    printf("  op start -> linear start %.2f\n", ((double)h[64 + 48] - (double)h[2]) * 0.01);
    return 0;
}
