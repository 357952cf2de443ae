// Probe: what does a cross-stream dependency cost on the stream that SIGNALS it?
//   (a) nothing attached            main: K1 K2
//   (b) K1 carries a stop event (hipExtLaunchKernelGGL), side stream waits for it, runs K3
//   (c) K1 writes a word of signal memory, side stream hipStreamWaitValue32 on it, runs K3
// Prints the gap K1 end -> K2 start on the main stream and K1 end -> K3 start (100 MHz wall clock, in us).
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/probe tools/debug/stream_sync_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(unsigned long long* ts, int slot, unsigned* sig, unsigned val, int spin) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[2 * slot] = wall_clock64();
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ts[2 * slot + 1] = wall_clock64();
        if (sig) { __threadfence_system(); __hip_atomic_store(sig, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
int main() {
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* ts; CK(hipMalloc(&ts, 64 * 8));
    unsigned* sig = nullptr;
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(es));
    if (es == hipSuccess) CK(hipMemset(sig, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    unsigned long long h[64];
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && es != hipSuccess) break;
        double g12 = 0, g13 = 0; int n = 0;
        for (int it = 0; it < 30; ++it) {
            CK(hipMemsetAsync(ts, 0, 64 * 8, a)); CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, ts, 3, (unsigned*)nullptr, 0u, 300);       // warm the queue
            if (mode == 0) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, ts, 0, (unsigned*)nullptr, 0u, 1000);
            if (mode == 1) {
                hipExtLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, nullptr, ev, 0, ts, 0, (unsigned*)nullptr, 0u, 1000);
                CK(hipStreamWaitEvent(b, ev, 0));
            }
            if (mode == 2) {
                hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, ts, 0, sig, (unsigned)(it + 1), 1000);
                CK(hipStreamWaitValue32(b, sig, (unsigned)(it + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, ts, 1, (unsigned*)nullptr, 0u, 300);
            if (mode) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, b, ts, 2, (unsigned*)nullptr, 0u, 300);
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
            if (it >= 5) { g12 += (double)(h[2] - h[1]) * 0.01; g13 += mode ? (double)(h[4] - h[1]) * 0.01 : 0; ++n; }
        }
        printf("mode %d: K1 end -> K2 start (main) %.2f us, K1 end -> K3 start (side) %.2f us\n", mode, g12 / n, g13 / n);
    }
    return 0;
}
