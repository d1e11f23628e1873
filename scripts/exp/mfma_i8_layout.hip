// Experiment: operand / result lane maps of v_mfma_i32_32x32x32_i8 on gfx950, with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// assumed: A[i = lane&31][k = 16*(lane>>5) + e], B[k = 16*(lane>>5) + e][j = lane&31], e = byte index 0..15
__global__ void k(const int8_t* A /*32x32 row-major i,k*/, const int8_t* B /*32x32 row-major k,j*/, int* D /*32x32*/) {
    int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v4i a, b;
    int8_t* ap = (int8_t*)&a; int8_t* bp = (int8_t*)&b;
    for (int e = 0; e < 16; ++e) { ap[e] = A[r * 32 + 16 * h + e]; bp[e] = B[(16 * h + e) * 32 + r]; }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; D[row * 32 + r] = c[g]; }
}
int main() {
    std::vector<int8_t> A(1024), B(1024); std::vector<int> D(1024), R(1024, 0);
    srand(1); for (auto& v : A) v = rand() % 16; for (auto& v : B) v = rand() % 2;
    A[5 * 32 + 7] = 13;  // asymmetric
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += A[i * 32 + kk] * B[kk * 32 + j]; R[i * 32 + j] = s; }
    int8_t *dA, *dB; int* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024 (D[0]=%d ref %d, D[37]=%d ref %d)\n", bad, D[0], R[0], D[37], R[37]);
    return bad != 0;
}
