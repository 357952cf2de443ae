// Cross-stream ordering without a completion signal on the producer's stream: kernel A's last workgroup stores a step number
// into signal memory, the other stream waits for it with hipStreamWaitValue32 - against the usual event (record behind A,
// hipStreamWaitEvent on the other stream).  Measured: the gap between A and the NEXT kernel of A's own stream (what a
// completion signal costs the producer), and how long after A's end the consumer stream's kernel starts.
//   hipcc -O3 --offload-arch=gfx950 -o stream_flag stream_flag.hip ; ./stream_flag
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long wclk() { return wall_clock64(); }

// ~us of work on `blocks` workgroups; the last one to finish publishes `value` (if flag) and stamps its end
__global__ void work_kernel(float* buf, int iters, unsigned* done_cnt, unsigned* flag, unsigned value, unsigned long long* t_start, unsigned long long* t_end) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && t_start) *t_start = wclk();
    float x = buf[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    buf[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned n = atomicAdd(done_cnt, 1u);
        if (n == gridDim.x - 1) {
            *done_cnt = 0u;
            if (t_end) *t_end = wclk();
            if (flag) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int main() {
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t main_s, side_s;
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&side_s, hipStreamNonBlocking, lo);
    float* buf; hipMalloc(&buf, 256 * 1024 * 4 * 4);
    unsigned* cnt; hipMalloc(&cnt, 64); hipMemset(cnt, 0, 64);
    unsigned long long* ts; hipMalloc(&ts, 64 * 8); hipMemset(ts, 0, 64 * 8);
    unsigned* flag = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    *reinterpret_cast<volatile unsigned*>(flag) = 0;
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
    const int iters = 4000;                       // ~10 us
    for (int mode = 0; mode < 3; ++mode) {        // 0: no cross-stream ordering at all (floor), 1: event, 2: flag
        std::vector<double> gapAB, lagC;
        for (int it = 0; it < 60; ++it) {
            const unsigned v = (unsigned)(mode * 1000 + it + 1);
            // main: A then B; side: (wait) then C
            if (mode == 1) {
                hipExtLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, main_s, nullptr, ev, 0, buf, iters, cnt, (unsigned*)nullptr, v, ts + 0, ts + 1);
                hipStreamWaitEvent(side_s, ev, 0);
            } else {
                hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, main_s, buf, iters, cnt, mode == 2 ? flag : (unsigned*)nullptr, v, ts + 0, ts + 1);
                if (mode == 2) {
                    hipError_t w = hipStreamWaitValue32(side_s, flag, v, hipStreamWaitValueGte, 0xFFFFFFFFu);
                    if (w != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(w)); return 0; }
                }
            }
            hipLaunchKernelGGL(work_kernel, dim3(64), dim3(256), 0, main_s, buf + 65536, 400, cnt + 4, (unsigned*)nullptr, 0u, ts + 2, ts + 3);      // B
            hipLaunchKernelGGL(work_kernel, dim3(64), dim3(256), 0, side_s, buf + 131072, 400, cnt + 8, (unsigned*)nullptr, 0u, ts + 4, ts + 5);     // C
            hipStreamSynchronize(main_s); hipStreamSynchronize(side_s);
            unsigned long long h[6]; hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
            if (it >= 10) { gapAB.push_back(((double)h[2] - (double)h[1]) * 0.01); lagC.push_back(((double)h[4] - (double)h[1]) * 0.01); }
        }
        std::sort(gapAB.begin(), gapAB.end()); std::sort(lagC.begin(), lagC.end());
        printf("%-28s A's end -> B's start (same stream) median %6.2f us | A's end -> C's start (other stream) median %6.2f us (min %.2f)\n",
               mode == 0 ? "no ordering:" : mode == 1 ? "event (signal on A):" : "flag + hipStreamWaitValue32:", gapAB[gapAB.size() / 2], lagC[lagC.size() / 2], lagC[0]);
    }
    return 0;
}
