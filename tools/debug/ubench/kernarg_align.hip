// Is the kernel-argument segment as aligned as the arguments ask for, and does a 64-byte scalar load from it return what 16
// dword loads return?  (r3: the layer-chain descriptor blocks are read with s_load_dwordx16.)
//   hipcc -O3 --offload-arch=gfx950 kernarg_align.hip -o kernarg_align && ./kernarg_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x16 __attribute__((ext_vector_type(16)));
struct alignas(64) Blk { int v[16]; };
struct Prog { int a, b, c; long long q; Blk ops[4]; };
__global__ void k(Prog P, int oi, unsigned long long* out) {
    const Blk& o = P.ops[oi];
    const i32x16 h = *reinterpret_cast<const i32x16*>(&o);
    int s = 0;
    for (int i = 0; i < 16; ++i) s += (h[i] == oi * 100 + i);
    out[0] = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    out[1] = s;
    out[2] = h[0]; out[3] = h[15];
}
int main() {
    Prog P; P.a = 1; P.b = 2; P.c = 3; P.q = 4;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) P.ops[j].v[i] = j * 100 + i;
    unsigned long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 6; ++rep) {
        int oi = rep % 4;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, P, oi, d);
        unsigned long long h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("launch %d: kernarg %p (mod 64 = %llu), op %d: %llu of 16 match, first %lld last %lld\n", rep, (void*)h[0], h[0] & 63, oi, h[1], (long long)h[2], (long long)h[3]);
    }
    return 0;
}
