// Experiment (round 4): can a compute-bound (MFMA) kernel and an HBM-bound kernel launched on two HIP streams run side by side on
// the same CUs?  Kernel A: a matrix-core loop (grid and waves per CU variable); kernel B: a streaming copy.  Prints A alone,
// B alone, both at once -- for A filling every wave slot / leaving half of them / on CU-masked streams.
// Build: hipcc -O3 --offload-arch=gfx950 stream_overlap.hip -o stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    v8i a, b;
    for (int e = 0; e < 8; ++e) a[e] = threadIdx.x * 7 + e, b[e] = (threadIdx.x * 3 + e) & 0x22222222;
    a[6] = a[7] = b[4] = b[5] = b[6] = b[7] = 0;
    v16f c = {0};
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 4, 0, 0, 0, 0);
        asm volatile("" : "+v"(a[0]));
    }
    float s = 0;
    for (int g = 0; g < 16; ++g) s += c[g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ void sleep_kernel(unsigned long long ticks, float *out) {   // idles for `ticks` of the 100 MHz clock: no power, no pipes
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0f;
}
static float run2(hipStream_t sa, hipStream_t sb, bool doA, bool doB, int gridA, int itersA, float *oa, int gridS) {
    hipEvent_t e0, e1, ea, eb;
    hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreate(&ea), hipEventCreate(&eb);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(sa, e0, 0), hipStreamWaitEvent(sb, e0, 0);
    if (doA) mfma_loop<<<gridA, 256, 0, sa>>>(oa, itersA);
    if (doB) sleep_kernel<<<gridS, 64, 0, sb>>>(300000ull, oa + (1 << 20));   // 3 ms
    hipEventRecord(ea, sa), hipEventRecord(eb, sb);
    hipStreamWaitEvent(0, ea, 0), hipStreamWaitEvent(0, eb, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static float run(hipStream_t sa, hipStream_t sb, bool doA, bool doB, int gridA, int itersA, float *oa, const float4 *in, float4 *out, size_t n,
                 int gridB, int repsB) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(sa, e0, 0), hipStreamWaitEvent(sb, e0, 0);
    if (doA) mfma_loop<<<gridA, 256, 0, sa>>>(oa, itersA);
    if (doB)
        for (int r = 0; r < repsB; ++r) copy_kernel<<<gridB, 256, 0, sb>>>(in, out, n);
    hipEvent_t ea, eb;
    hipEventCreate(&ea), hipEventCreate(&eb);
    hipEventRecord(ea, sa), hipEventRecord(eb, sb);
    hipStreamWaitEvent(0, ea, 0), hipStreamWaitEvent(0, eb, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const size_t n = (1ull << 30) / 16;  // 1 GiB in, 1 GiB out per copy
    float4 *in, *out;
    float *oa;
    hipMalloc(&in, n * 16), hipMalloc(&out, n * 16), hipMalloc(&oa, 1 << 24);
    hipMemset(in, 1, n * 16);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking), hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    // CU-masked streams: A on the even CUs of every XCD pair ..., B on the rest (mask words: 256 bits)
    std::vector<uint32_t> ma(8, 0x55555555u), mb(8, 0xAAAAAAAAu);
    hipStream_t ca, cb;
    hipError_t em = hipExtStreamCreateWithCUMask(&ca, 8, ma.data());
    hipError_t em2 = hipExtStreamCreateWithCUMask(&cb, 8, mb.data());
    printf("CU-mask streams: %s / %s\n", hipGetErrorString(em), hipGetErrorString(em2));
    const int reps = 6;
    for (int wavesPerSimdA : {8, 4, 2, 1}) {
        const int gridA = 256 * wavesPerSimdA;  // 256-thread blocks: one wave per SIMD each
        const int itersA = 400000 / wavesPerSimdA;  // same total MFMA work
        for (int warm = 0; warm < 2; ++warm) run(sa, sb, true, true, gridA, itersA, oa, in, out, n, 4096, reps);
        const float a = run(sa, sb, true, false, gridA, itersA, oa, in, out, n, 4096, reps);
        const float b = run(sa, sb, false, true, gridA, itersA, oa, in, out, n, 4096, reps);
        const float ab = run(sa, sb, true, true, gridA, itersA, oa, in, out, n, 4096, reps);
        printf("A = MFMA loop with %d waves/SIMD resident: A alone %.2f ms, B alone (%d x 2 GiB moved) %.2f ms, both %.2f ms  (sum %.2f, max %.2f)\n",
               wavesPerSimdA, a, reps, b, ab, a + b, a > b ? a : b);
    }
    for (int w : {8, 2}) {
        const int gridA = 256 * w, itersA = 400000 / w;
        run2(sa, sb, true, true, gridA, itersA, oa, 256);
        const float a = run2(sa, sb, true, false, gridA, itersA, oa, 256), b = run2(sa, sb, false, true, gridA, itersA, oa, 256);
        const float ab = run2(sa, sb, true, true, gridA, itersA, oa, 256);
        printf("A (%d waves/SIMD) beside a 3 ms SLEEP kernel (256 one-wave blocks): A alone %.2f, sleep alone %.2f, both %.2f\n", w, a, b, ab);
    }
    if (em == hipSuccess && em2 == hipSuccess) {
        const int gridA = 256 * 4, itersA = 100000;
        for (int warm = 0; warm < 2; ++warm) run(ca, cb, true, true, gridA, itersA, oa, in, out, n, 4096, reps);
        const float a = run(ca, cb, true, false, gridA, itersA, oa, in, out, n, 4096, reps);
        const float b = run(ca, cb, false, true, gridA, itersA, oa, in, out, n, 4096, reps);
        const float ab = run(ca, cb, true, true, gridA, itersA, oa, in, out, n, 4096, reps);
        printf("CU-masked halves: A alone %.2f ms, B alone %.2f ms, both %.2f ms\n", a, b, ab);
    }
    return 0;
}
