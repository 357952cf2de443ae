// What slows chain4.h's weight stream?  One layer pass = 16 waves x 13 x 16-byte buffer loads (4 column groups x 4 k-quarters of
// a [51 k4][200 columns][4] matrix), variants of what happens to the data:
//   0 linear addresses, summed            1 chain4's addresses (row stride 3.2 KB, column groups), summed
//   2 = 1 + each float4 through four v_mfma_f32_4x4x1 (A from registers)      3 = 2 + A operand from LDS (ds_read_b128 per chunk)
//   4 = 3 + partial sums to LDS and a second barrier (the whole matrix phase of chain4_linear)
// Build: hipcc -O3 --offload-arch=gfx950 -o chain_stream chain_stream.hip ; run: ./chain_stream [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// COLD: every pass reads another 256 KB window of a 512 MB buffer (the same window in every workgroup): first touch per XCD, as a
// layer's weights are in a fresh kernel (the launch invalidates the L2 of lines other XCDs may have written)
template <int MODE, bool COLD = false>
__global__ __launch_bounds__(1024) void pass_kernel(const float* W, int reps, float* out, unsigned long long* ts) {
    __shared__ __attribute__((aligned(16))) float xs[4 * 256];
    __shared__ float part[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 256; i += 1024) xs[i] = 0.001f * i;
    __syncthreads();
    const int cg = wave & 3, ks = wave >> 2;
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x7FFFFFF0, 0x00020000);
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        float4 w[13];
        const __amdgpu_buffer_rsrc_t rw = COLD ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W) + (size_t)r * 65536, 0, 0x7FFFFFF0, 0x00020000) : rw0;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 13; ++j) w[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)tid * 16u, (unsigned)j * 16384u, 0));
        } else {
            const unsigned vo = (unsigned)min(64 * cg + lane, 199) * 16u;
#pragma unroll
            for (int j = 0; j < 13; ++j) w[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, vo, (unsigned)(min(ks * 13 + j, 50) * 200) * 16u, 0));
        }
        if (MODE <= 1) {
#pragma unroll
            for (int j = 0; j < 13; ++j) acc += w[j].x + w[j].y + w[j].z + w[j].w;
        } else {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            const float* a = xs + (lane & 3) * 256;
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                float4 x = make_float4(1.f, 2.f, 3.f, 4.f);
                if (MODE >= 3) x = *reinterpret_cast<const float4*>(a + 4 * (ks * 13 + j));
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w[j].x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w[j].y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w[j].z, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, w[j].w, a1, 0, 0, 0);
            }
            a0 += a1;
            if (MODE >= 4) {
                float* pp = part + ks * 1024 + 64 * cg + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) pp[q * 256] = a0[q];
                __syncthreads();
                acc += part[tid] + part[1024 + tid] + part[2048 + tid] + part[3072 + tid];
            } else acc += a0[0] + a0[1] + a0[2] + a0[3];
        }
        __syncthreads();
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ts[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + tid] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 25, reps = 200;
    float* W; float* out; unsigned long long* ts;
    hipMalloc(&W, (size_t)512 << 20); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ts, 256 * 8);
    hipMemset(W, 0, (size_t)512 << 20);
    std::vector<unsigned long long> h(256);
    auto report = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < wgs; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-64s %3d workgroups: %.2f us per pass\n", name, wgs, mx * 0.01 / reps);
    };
    hipLaunchKernelGGL(pass_kernel<0>, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("0 linear addresses, summed");
    hipLaunchKernelGGL(pass_kernel<1>, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("1 chain4 addresses, summed");
    hipLaunchKernelGGL(pass_kernel<2>, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("2 chain4 addresses, 4x4x1 MFMAs (A in registers)");
    hipLaunchKernelGGL(pass_kernel<3>, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("3 ... A from LDS");
    hipLaunchKernelGGL(pass_kernel<4>, dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("4 ... partial sums through LDS + barrier");
    hipLaunchKernelGGL((pass_kernel<1, true>), dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("1 COLD: a fresh window per pass");
    hipLaunchKernelGGL((pass_kernel<4, true>), dim3(wgs), dim3(1024), 0, 0, W, reps, out, ts); report("4 COLD");
    return 0;
}
