// Experiment: how many independent VALU instructions hide behind one v_mfma_f32_32x32x64_f8f6f4 (fp6 x fp6) and
// behind one v_mfma_f32_32x32x16_bf16, one wave per SIMD?  Prints clock ticks per MFMA for 0..10 fillers.
// Build: hipcc -O2 --offload-arch=gfx950 mfma_filler.hip -o mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE, int NF>
__global__ void k(float *out, long long *cyc, int iters) {
    v8i a, b;
    for (int e = 0; e < 8; ++e) a[e] = threadIdx.x * 7 + e, b[e] = threadIdx.x * 3 + e;
    a[6] = a[7] = b[6] = b[7] = 0;
    v16f c = {0};
    int f[10];
    for (int i = 0; i < 10; ++i) f[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 0, 0, 0);
        else {
            v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a4), __builtin_bit_cast(v8bf, b4), c, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(f[(j + 1) % 10]), "v"(f[(j + 2) % 10]));
    }
    long long t1 = clock64();
    float s = 0;
    for (int g = 0; g < 16; ++g) s += c[g];
    for (int j = 0; j < 10; ++j) s += f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int NF>
double run(float *o, long long *c) {
    long long h = 0;
    for (int r = 0; r < 2; ++r) { k<MODE, NF><<<1, 256>>>(o, c, 4096); hipDeviceSynchronize(); }
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    return (double)h / 4096;
}
int main() {
    float *o; long long *c;
    hipMalloc(&o, 4096); hipMalloc(&c, 8);
    printf("fillers:        0      2      4      6      8     10\n");
    printf("fp6 32x32x64: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<0, 0>(o, c), run<0, 2>(o, c), run<0, 4>(o, c), run<0, 6>(o, c), run<0, 8>(o, c), run<0, 10>(o, c));
    printf("bf16 32x32x16:%6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<1, 0>(o, c), run<1, 2>(o, c), run<1, 4>(o, c), run<1, 6>(o, c), run<1, 8>(o, c), run<1, 10>(o, c));
    return 0;
}
