// Experiment: v_mfma_f32_32x32x64_f8f6f4 with A = fp6 (e2m3) and B = fp4 (e2m1) on gfx950.
//   * operand lane / bit maps, checked with exact small-integer data (hypothesis: lane (r, h) holds
//     k = 32h + i as the i-th 6-bit (A) / 4-bit (B) field of a little-endian bit stream);
//   * every integer 0..15 is v/2 in e2m3, a code bit is 1.0 in e2m1: the dot is exact in f32;
//   * issue cost per instruction next to the i8 / bf16 / fp8 forms.
// Build: hipcc -O2 --offload-arch=gfx950 mfma_f6f4_layout.hip -o mfma_f6f4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

static uint32_t e2m3_of_half(uint32_t v) {  // code of v/2, v = 0..15
    if (v < 4) return 4 * v;
    if (v < 8) return 8 + 2 * v;
    return 16 + v;
}

__global__ void k_layout(const uint32_t *A /*64 lanes x 8 dwords*/, const uint32_t *B, float *D) {
    int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v8i a, b;
    for (int e = 0; e < 8; ++e) a[e] = A[lane * 8 + e], b[e] = B[lane * 8 + e];
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2 /*A e2m3*/, 4 /*B e2m1*/, 0, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

template <int MODE>
__global__ void k_time(float *out, long long *cyc, int iters) {
    v8i a, b;
    for (int e = 0; e < 8; ++e) a[e] = threadIdx.x * 7 + e, b[e] = threadIdx.x * 3 + e;
    if (MODE != 2) { a[6] = a[7] = 0; b[4] = b[5] = b[6] = b[7] = 0; }
    v16f c = {0};
    v16i ci = {0};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 4, 0, 0, 0, 0);
        if (MODE == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 0, 0, 0);
        if (MODE == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
        if (MODE == 3) {
            v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
            ci = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ci, 0, 0, 0);
        }
        if (MODE == 4) {
            v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a4), __builtin_bit_cast(v8bf, b4), c, 0, 0, 0);
        }
        if (MODE == 5) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 4, 0, 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int g = 0; g < 16; ++g) s += c[g] + (float)ci[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    // ---- layout ---------------------------------------------------------------------------------
    std::vector<uint32_t> Av(32 * 64), Bv(64 * 32);  // A[row][k] in 0..15, B[k][col] in {0,1}
    srand(7);
    for (auto &v : Av) v = rand() % 16;
    for (auto &v : Bv) v = rand() % 2;
    std::vector<uint32_t> Ap(64 * 8, 0), Bp(64 * 8, 0);
    for (int lane = 0; lane < 64; ++lane) {
        int r = lane & 31, h = lane >> 5;
        for (int i = 0; i < 32; ++i) {
            uint32_t ca = e2m3_of_half(Av[r * 64 + 32 * h + i]);
            int bit = 6 * i;
            Ap[lane * 8 + bit / 32] |= ca << (bit % 32);
            if (bit % 32 > 26) Ap[lane * 8 + bit / 32 + 1] |= ca >> (32 - bit % 32);
            uint32_t cb = Bv[(32 * h + i) * 32 + r] ? 2u : 0u;  // e2m1 1.0
            Bp[lane * 8 + (4 * i) / 32] |= cb << ((4 * i) % 32);
        }
    }
    uint32_t *dA, *dB;
    float *dD;
    hipMalloc(&dA, Ap.size() * 4), hipMalloc(&dB, Bp.size() * 4), hipMalloc(&dD, 4096);
    hipMemcpy(dA, Ap.data(), Ap.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bp.data(), Bp.size() * 4, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int kk = 0; kk < 64; ++kk) s += Av[i * 64 + kk] * Bv[kk * 32 + j];
            bad += (2.0f * D[i * 32 + j] != (float)s);
        }
    printf("f8f6f4 A=e2m3 B=e2m1 layout check: %d mismatches of 1024 (2*D[0]=%g, 2*D[37]=%g)\n", bad, 2 * D[0], 2 * D[37]);

    // ---- issue cost -----------------------------------------------------------------------------
    float *dO;
    long long *dC;
    hipMalloc(&dO, 256 * 4 * 1024), hipMalloc(&dC, 8);
    const int iters = 4096;
    const char *names[6] = {"32x32x64 fp6 x fp4", "32x32x64 fp6 x fp6", "32x32x64 fp8 x fp8", "32x32x32 i8", "32x32x16 bf16", "32x32x64 fp8 x fp4"};
    for (int mode = 0; mode < 6; ++mode) {
        long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {  // 1 wave per SIMD on one CU
            if (mode == 0) k_time<0><<<1, 256>>>(dO, dC, iters);
            if (mode == 1) k_time<1><<<1, 256>>>(dO, dC, iters);
            if (mode == 2) k_time<2><<<1, 256>>>(dO, dC, iters);
            if (mode == 3) k_time<3><<<1, 256>>>(dO, dC, iters);
            if (mode == 4) k_time<4><<<1, 256>>>(dO, dC, iters);
            if (mode == 5) k_time<5><<<1, 256>>>(dO, dC, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
        printf("%-22s %8.2f clock64 ticks per MFMA\n", names[mode], (double)c / iters);
    }
    return bad != 0;
}
