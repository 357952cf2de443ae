// Known-size read kernels in the access shapes of dec_opt_x3_kernel (csrc/dec_crit_x3.h), for calibrating rocprofv3's
// FETCH_SIZE / TCC_EA0_RDREQ* on gfx950 (MI355X_MICROARCH.md: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel reads `bytes` once and writes one word per
// workgroup; run under `rocprofv3 --pmc <counter> --kernel-trace` (tools/calib/pmc_calib.sh).
//   stream16_nt   buffer_load_dwordx4, all 1024 lanes, aux = 2 (non-temporal): the p / m / v streams (26 112-byte tile spans)
//   stream16      the same without the non-temporal hint
//   gtile16_nt    buffer_load_dwordx4 aux = 2, 800 of 1024 lanes active, one 12 800-byte tile per step: the stored dL/dlogits
//   stream4       global_load_dword, 4 bytes per lane
//   rows_l2       804-byte rows of ONE 80 KB block re-read by every workgroup (dh2: L2 / MALL hits after the first reader)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int AUX>
__global__ __launch_bounds__(1024) void stream16(const float* __restrict__ src, size_t bytes, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (unsigned)(bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : bytes), 0x00020000);
    const unsigned span = 1024u * 16u;
    unsigned acc = 0;
    for (size_t off = (size_t)blockIdx.x * span; off + span <= bytes; off += (size_t)gridDim.x * span) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16u, (unsigned)off, AUX);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void gtile16_nt(const float* __restrict__ src, size_t bytes, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (unsigned)(bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : bytes), 0x00020000);
    const unsigned tile = 800u * 16u;           // 100 rows x 32 items x 4 bytes
    unsigned acc = 0;
    for (size_t off = (size_t)blockIdx.x * tile; off + tile <= bytes; off += (size_t)gridDim.x * tile) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x < 800 ? threadIdx.x * 16u : 0x80000000u, (unsigned)off, 2);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void stream4(const float* __restrict__ src, size_t bytes, unsigned* out) {
    const size_t n = bytes / 4;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) acc += src[i];
    if (acc == 1.2345f) out[blockIdx.x] = 1;
}
__global__ __launch_bounds__(1024) void rows_l2(const float* __restrict__ src, int rows, int ld, unsigned* out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < rows * ld; i += 1024) acc += src[i];
    if (acc == 1.2345f) out[blockIdx.x] = 1;
}

int main(int argc, char** argv) {
    const size_t MB = (size_t)1 << 20;
    const size_t big = 240 * MB, g = 40 * MB;
    float* buf; unsigned* out;
    CHK(hipMalloc(&buf, big + 4096)); CHK(hipMalloc(&out, 4096 * 4));
    CHK(hipMemset(buf, 1, big)); CHK(hipMemset(out, 0, 4096 * 4));
    const int wgs = 128;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16<2>, dim3(wgs), dim3(1024), 0, 0, buf, big, out);
        hipLaunchKernelGGL(stream16<0>, dim3(wgs), dim3(1024), 0, 0, buf, big, out);
        hipLaunchKernelGGL(gtile16_nt, dim3(wgs), dim3(1024), 0, 0, buf, g, out);
        hipLaunchKernelGGL(stream4, dim3(wgs), dim3(1024), 0, 0, buf, g, out);
        hipLaunchKernelGGL(rows_l2, dim3(wgs), dim3(1024), 0, 0, buf, 100, 204, out);
        CHK(hipDeviceSynchronize());
    }
    printf("bytes read per launch: stream16<2> %zu  stream16<0> %zu  gtile16_nt %zu  stream4 %zu  rows_l2 %d x %d workgroups\n",
           big / (1024 * 16) * (1024 * 16), big / (1024 * 16) * (1024 * 16), g / 12800 * 12800, g, 100 * 204 * 4, wgs);
    return 0;
}
