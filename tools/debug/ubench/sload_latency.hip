// Latency of a dependent scalar load from (a) the kernel-argument segment, (b) a device buffer (constant address space).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct Big { int next[1200]; };
__global__ void chase_arg(Big b, int start, unsigned long long* out) {
    int i = start;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int k = 0; k < 256; ++k) { i = b.next[i]; asm volatile("" : "+s"(i)); }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void chase_buf(const int* nb, int start, unsigned long long* out) {
    const __attribute__((address_space(4))) int* next = (const __attribute__((address_space(4))) int*)nb;
    int i = start;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int k = 0; k < 256; ++k) { i = next[i]; asm volatile("" : "+s"(i)); }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
int main() {
    Big b; for (int i = 0; i < 1200; ++i) b.next[i] = (i * 37 + 16) % 1200;     // jumps of >= 64 bytes
    int* nb; unsigned long long* out; hipMalloc(&nb, sizeof(b)); hipMalloc(&out, 16);
    hipMemcpy(nb, b.next, sizeof(b), hipMemcpyHostToDevice);
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chase_arg, dim3(1), dim3(64), 0, 0, b, 5, out); hipDeviceSynchronize();
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("kernel-argument segment: %.0f shader clocks per dependent s_load (first touch of each line mostly)\n", h[0] / 256.0);
        hipLaunchKernelGGL(chase_buf, dim3(1), dim3(64), 0, 0, nb, 5, out); hipDeviceSynchronize();
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("device buffer:           %.0f shader clocks per dependent s_load\n", h[0] / 256.0);
    }
    return 0;
}
