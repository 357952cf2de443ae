// LDS-DMA (global_load_lds_dwordx4) to LDS addresses beyond 64 KB: does M0 carry the whole byte address on gfx950?
// hipcc -O3 --offload-arch=gfx950 -o build_ab/glds_hi tools/debug/ubench/glds_hi.hip && build_ab/glds_hi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(1024) void k(const float4* src, float4* dst, unsigned base) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lds);      // (LDS byte address of the dynamic segment)
    const unsigned zone = base + wave * 1024u;
    glds16(src + tid, lds0 + zone);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + zone + lane * 16);   // (behind the asm's memory clobber)
    dst[tid] = v;
}
int main() {
    const int n = 1024;
    std::vector<float4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *d, *e; hipMalloc(&d, n * 16); hipMalloc(&e, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned base : {0u, 32768u, 65536u, 98304u, 131072u, 147456u}) {
        hipMemset(e, 0, n * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), 160 * 1024, 0, d, e, base);
        hipError_t err = hipDeviceSynchronize();
        hipMemcpy(o.data(), e, n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) if (o[i].x != h[i].x || o[i].w != h[i].w) ++bad;
        printf("base %6u: %s, %d mismatches of %d\n", base, hipGetErrorString(err), bad, n);
    }
    return 0;
}
