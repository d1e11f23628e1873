// Experiment (round 4): why does the dim-128 matrix-core scan step (1 x bf16 32x32x16 threshold MFMA + 2 x fp6 32x32x64 on ONE
// accumulator, then a 9-op gate) cost ~135 cycles when its matrix work is 96?  Hypothesis: a dependent MFMA of a different
// opcode cannot start until its predecessor has left the pipe (~44 cycles instead of 32) and that stall is not filled by
// the other waves of the SIMD.  Variants: NACC accumulators interleaved inside one wave's stream (M1a M1b .. M2a M2b ..),
// 1..4 waves per SIMD, with / without the threshold MFMA, B operand as fp6 or fp4, A fragments re-read from LDS per tile.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_interleave.hip -o mfma_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define OPAQUE(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ int imax3(int a, int b, int c) {
    const int m = a > b ? a : b;
    return m > c ? m : c;
}
constexpr int NT = 6;  // sub-tiles per query tile and wave (all variants do the same work)

template <int NACC, int THR, int BFMT, int SB = 0, int GATE = 8, int TREE = 0, int BR = 1>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, int gate_min) {
    extern __shared__ uint32_t lds[];  // 4 slots x 32 rows x 30 dwords: a fake ring of query-tile images
    const uint32_t lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    for (uint32_t i = threadIdx.x; i < 4 * 32 * 30; i += blockDim.x) lds[i] = i * 2654435761u >> 27;  // small fp6 fields
    v8i b[2];
    for (int m = 0; m < 2; ++m) {
        for (int e = 0; e < 8; ++e) b[m][e] = (int)((threadIdx.x * 3 + e + m) & 0x08208208);
        b[m][6] = b[m][7] = 0;
    }
    v4i ub = {(int)threadIdx.x, 3, 5, 7};
    int flagged = 0, keep = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const uint32_t *img = lds + (i & 3) * (32 * 30);
        v8i a[2];
        for (int m = 0; m < 2; ++m) {
            for (int e = 0; e < 6; e += 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&img[j * 30 + 12 * h + 6 * m + e]);
                a[m][e] = (int)v.x, a[m][e + 1] = (int)v.y;
            }
            a[m][6] = a[m][7] = 0;
        }
        int tilemax = (int)0x80000000;
        const v4i ua = *reinterpret_cast<const v4i *>(&img[j * 30 + 24 + 0 * h]);
#pragma unroll
        for (int t0s = 0; t0s < NT; t0s += NACC) {
            v16f c[NACC];
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (THR) {
                    OPAQUE(ub[0]);
                    c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, ua), __builtin_bit_cast(v8bf, ub), z, 0, 0, 0);
                } else
                    c[q] = z;
            }
            if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int q = 0; q < NACC; ++q) {
                    OPAQUE(b[m][0]);
                    if (BFMT == 2) c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[m], c[q], 2, 2, 0, 0, 0, 0);
                    else {
                        c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[m], c[q], 2, 4, 0, 0, 0, 0);
                    }
                }
                if (SB) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const v16i ci = __builtin_bit_cast(v16i, c[q]);
                int mx;
                if (GATE == 0) {
                    keep ^= ci[0];  // keep the accumulator alive (one VALU)
                    continue;
                } else if (TREE) {
                    const int m0 = imax3(ci[0], ci[1], ci[2]), m1 = imax3(ci[3], ci[4], ci[5]), m2 = imax3(ci[6], ci[7], ci[8]);
                    const int m3 = imax3(ci[9], ci[10], ci[11]), m4 = imax3(ci[12], ci[13], ci[14]);
                    mx = imax3(imax3(m0, m1, m2), imax3(m3, m4, ci[15]), ci[15]);
                } else {
                    mx = imax3(ci[0], ci[1], ci[2]);
#pragma unroll
                    for (int g = 3; g < 2 * GATE - 1 && g < 15; g += 2) mx = imax3(mx, ci[g], ci[g + 1]);
                    if (GATE >= 8) mx = mx > ci[15] ? mx : ci[15];
                    
                }
                if (BR) {
                    if (__ballot(mx >= gate_min) != 0ull) {
                        ++flagged;
                        out[threadIdx.x] = c[q][3];
                    }
                } else
                    tilemax = tilemax > mx ? tilemax : mx;
            }
        }
        if (!BR) {
            if (__ballot(tilemax >= gate_min) != 0ull) {
                ++flagged;
                out[threadIdx.x] = (float)tilemax;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(flagged + keep);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = r1 - r0;
}


// Software-pipelined form (additive gate: no threshold MFMA): the two MFMAs of sub-tile t+1 are issued BEFORE the gate of
// sub-tile t, so that the gate's vector instructions run in the shadow of the next sub-tile's matrix work (two accumulators).
// XT = 1: the pipeline also runs across tile boundaries (the next tile's A / C fragments are loaded one tile ahead).
template <int NTT, int XT>
__global__ __launch_bounds__(1024) void kp(float *out, unsigned long long *cyc, int iters, float gate_h) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    for (uint32_t i = threadIdx.x; i < 4 * 32 * 48; i += blockDim.x) lds[i] = i * 2654435761u >> 27;
    v8i b[2];
    for (int m = 0; m < 2; ++m) {
        for (int e = 0; e < 8; ++e) b[m][e] = (int)((threadIdx.x * 3 + e + m) & 0x22222222);
        b[m][4] = b[m][5] = b[m][6] = b[m][7] = 0;
    }
    int flagged = 0;
    const float hc = gate_h;
    __syncthreads();
    auto load_tile = [&](int i, v8i (&a)[2], v16f &c) {
        const uint32_t *img = lds + (i & 3) * (32 * 48);
        for (int e = 0; e < 12; e += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&img[j * 28 + 12 * h + e]);
            if (e == 0) a[0][0] = v.x, a[0][1] = v.y, a[0][2] = v.z, a[0][3] = v.w;
            if (e == 4) a[0][4] = v.x, a[0][5] = v.y, a[1][0] = v.z, a[1][1] = v.w;
            if (e == 8) a[1][2] = v.x, a[1][3] = v.y, a[1][4] = v.z, a[1][5] = v.w;
        }
        a[0][6] = a[0][7] = a[1][6] = a[1][7] = 0;
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 cv = *reinterpret_cast<const float4 *>(&img[32 * 28 + 8 * g4 + 4 * h]);
            c[4 * g4] = cv.x, c[4 * g4 + 1] = cv.y, c[4 * g4 + 2] = cv.z, c[4 * g4 + 3] = cv.w;
        }
    };
    auto mm2 = [&](const v8i (&a)[2], const v16f &c0) {
        OPAQUE(b[0][0]);
        v16f c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[0], c0, 2, 4, 0, 0, 0, 0);
        OPAQUE(b[1][0]);
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[1], c, 2, 4, 0, 0, 0, 0);
    };
    auto gate = [&](const v16f &c) {
        float mx = hc;
#pragma unroll
        for (int g = 0; g < 16; g += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, c[g]), c[g + 1]);
        if (__builtin_expect(__ballot(mx > hc) != 0ull, 0)) {
            ++flagged;
            out[threadIdx.x] = c[3];
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    v8i a[2], an[2];
    v16f ci, cn;
    load_tile(0, a, ci);
    v16f pend;  // the accumulator whose gate is still owed
    bool have = false;
    for (int i = 0; i < iters; ++i) {
        if (XT) load_tile(i + 1, an, cn);  // one tile ahead
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const v16f cur = mm2(a, ci);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0 || (XT && have)) gate(pend);
            __builtin_amdgcn_sched_barrier(0);
            pend = cur;
        }
        if (!XT) {
            gate(pend);
            load_tile(i + 1, a, ci);
        } else {
            have = true;
            a[0] = an[0], a[1] = an[1], ci = cn;
        }
    }
    if (XT) gate(pend);
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)flagged;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = r1 - r0;
}
template <int NTT, int XT>
void runp(float *o, unsigned long long *c, int wps) {
    unsigned long long h[2] = {0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    const int iters = 4096;
    hipFuncSetAttribute((const void *)kp<NTT, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        kp<NTT, XT><<<256, 256 * wps, 100 * 1024>>>(o, c, iters, 1.0e30f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double steps_per_simd = (double)wps * iters * NTT;
    printf("PIPELINED nt %d across-tiles %d waves/SIMD %d: %7.1f ns/step/SIMD  memtime/realtime = %.3f GHz  (%.2f ms)\n", NTT, XT, wps,
           ms * 1e6 / steps_per_simd, (double)h[0] / ((double)h[1] * 10.0), ms);
}
template <int NACC, int THR, int BFMT, int SB = 0, int GATE = 8, int TREE = 0, int BR = 1>
void run(float *o, unsigned long long *c, int wps) {
    unsigned long long h[2] = {0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    const int iters = 4096;
    hipFuncSetAttribute((const void *)k<NACC, THR, BFMT, SB, GATE, TREE, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        k<NACC, THR, BFMT, SB, GATE, TREE, BR><<<256, 256 * wps, 100 * 1024>>>(o, c, iters, 0x7F000000);  // > 80 KB of LDS: one block per CU
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double steps_per_simd = (double)wps * iters * NT;
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // s_memrealtime ticks at 100 MHz
    printf("acc %d thr %d bfmt %d sb %d gate %d tree %d br %d waves/SIMD %d: %7.1f ns/step/SIMD  %7.1f cyc(memtime)/step/SIMD  memtime/realtime = %.3f GHz  (%.2f ms)\n",
           NACC, THR, BFMT, SB, GATE, TREE, BR, wps, ms * 1e6 / steps_per_simd, (double)h[0] / steps_per_simd * 1.0, ghz, ms);
}
int main() {
    float *o;
    unsigned long long *c;
    hipMalloc(&o, 256 * 1024 * 4);
    hipMalloc(&c, 16);
    for (int wps = 1; wps <= 4; ++wps) {
        run<1, 0, 4, 0, 8, 0, 1>(o, c, wps);
        runp<4, 0>(o, c, wps);
        runp<4, 1>(o, c, wps);
        runp<6, 0>(o, c, wps);
        runp<6, 1>(o, c, wps);
    }
    return 0;
}
