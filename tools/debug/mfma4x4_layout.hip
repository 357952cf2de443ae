// Checks the operand / result lane maps of v_mfma_f32_4x4x1_16B_f32 that csrc/chain4.h assumes:
//   A: lane l supplies A[i = l % 4] of block l / 4;  B: lane l supplies B[j = l % 4] of block l / 4;
//   D: register v of lane l = D[i = v][j = l % 4] of block l / 4          (16 blocks of 4x4, K = 1)
// i.e. with every block given the same A column x[0..3] and B = w[lane], lane l ends up with x[v] * w[l] in register v.
// build: hipcc --offload-arch=gfx950 -o /tmp/mfma4x4 tools/debug/mfma4x4_layout.hip && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, const float* w, float* out, int K) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < K; ++kk)
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x[(lane & 3) * K + kk], w[kk * 64 + lane], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[v * 64 + lane] = acc[v];
}
int main() {
    const int K = 7;
    float hx[4 * K], hw[K * 64], ho[256], *dx, *dw, *dout;
    for (int i = 0; i < 4 * K; ++i) hx[i] = (float)(1 + (i * 7) % 11);
    for (int i = 0; i < K * 64; ++i) hw[i] = (float)(1 + (i * 13) % 17);
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dw, sizeof(hw)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice); hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dw, dout, K);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r)
        for (int n = 0; n < 64; ++n) {
            float want = 0.f;
            for (int kk = 0; kk < K; ++kk) want += hx[r * K + kk] * hw[kk * 64 + n];
            if (ho[r * 64 + n] != want) { if (bad < 5) printf("mismatch row %d col %d: got %g want %g\n", r, n, ho[r * 64 + n], want); ++bad; }
        }
    printf(bad ? "LAYOUT MISMATCH (%d cells)\n" : "layout ok: D[v][lane] = sum_k x[v][k] * w[k][lane]\n", bad);
    return bad != 0;
}
