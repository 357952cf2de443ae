// How fast can ONE CU take in a 160 KB matrix that 25-50 other CUs read at the same time (the layer chains' situation)?
//   mode 0: 16 waves x 16-byte register loads, all issued before the first use (what chain4.h does)
//   mode 1: LDS-DMA (global_load_lds_dwordx4) by L loader waves into a ring, consumers only wait
// Build: hipcc -O3 --offload-arch=gfx950 -o cu_ingest cu_ingest.hip ; run: ./cu_ingest [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kBytes = 160 * 1024;     // one 200 x 200 fp32 layer

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(1024) void reg_stream(const float4* W, int reps, float* out, unsigned long long* ts) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        float4 v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = W[(size_t)r * 0 + tid + 1024 * j];      // 10 x 16 KB = 160 KB per pass
#pragma unroll
        for (int j = 0; j < 10; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        __syncthreads();
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ts[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + tid] = acc;
}

template <int L>   // loader waves
__global__ __launch_bounds__(1024) void dma_stream(const float4* W, int reps, float* out, unsigned long long* ts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 160 KB: the whole matrix lands once per pass
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned base = (unsigned)reinterpret_cast<size_t>(lds);
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        if (wave < L) {
            // 160 pieces of 1 KB, dealt round-robin to the loader waves
            for (int p = wave; p < kBytes / 1024; p += L)
                glds16(W + p * 64 + lane, __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // consumers: every thread reads 10 float4 of the image (as the products would)
#pragma unroll
        for (int j = 0; j < 10; ++j) { const float4 v = reinterpret_cast<const float4*>(lds)[tid + 1024 * j]; acc += v.x + v.y + v.z + v.w; }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ts[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + tid] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 50, reps = 200;
    float4* W; float* out; unsigned long long* ts;
    hipMalloc(&W, kBytes); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ts, 256 * 8);
    hipMemset(W, 0, kBytes);
    std::vector<unsigned long long> h(256);
    auto report = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < wgs; ++i) mx = h[i] > mx ? h[i] : mx;
        const double us = mx * 0.01 / reps;
        printf("%-28s %3d workgroups: %.2f us per 160 KB pass = %.0f GB/s per CU\n", name, wgs, us, kBytes / us * 1e-3);
    };
    hipLaunchKernelGGL(reg_stream, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("register loads (16 waves)");
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dma_stream<1>, dim3(wgs), dim3(1024), kBytes, 0, W, reps, out, ts); report("LDS-DMA, 1 loader wave");
    hipLaunchKernelGGL(dma_stream<2>, dim3(wgs), dim3(1024), kBytes, 0, W, reps, out, ts); report("LDS-DMA, 2 loader waves");
    hipLaunchKernelGGL(dma_stream<4>, dim3(wgs), dim3(1024), kBytes, 0, W, reps, out, ts); report("LDS-DMA, 4 loader waves");
    hipLaunchKernelGGL(dma_stream<8>, dim3(wgs), dim3(1024), kBytes, 0, W, reps, out, ts); report("LDS-DMA, 8 loader waves");
    hipLaunchKernelGGL(dma_stream<16>, dim3(wgs), dim3(1024), kBytes, 0, W, reps, out, ts); report("LDS-DMA, 16 loader waves");
    return 0;
}
