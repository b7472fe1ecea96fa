// What does the fp32 MFMA pipe deliver under pure matrix load?  Independent / dependent accumulator chains, 1-4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate/mfma_peak.hip -o tools/ablate/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool M32>
__global__ __launch_bounds__(256) void mfma_k(float* out, int iters, float a0, float b0)
{
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if (a0 < 0.0f) {   // random-looking operands: 8 values per lane from a hash (data-dependent power draw)
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float av[8], bv[8];
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            av[k] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
            h = h * 1664525u + 1013904223u;
            bv[k] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(k + i) & 7], bv[(k + 3 * i) & 7], acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].w;
        if (s == 12345.0f) out[0] = s;
        return;
    }
    if (M32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
        if (s == 12345.0f) out[0] = s;
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].w;
        if (s == 12345.0f) out[0] = s;
    }
}

template <typename K>
void run(const char* name, K k, int wgs, int nacc, bool m32, float* d, float a0 = 1.0f)
{
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters, a0, 1.0f);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters, a0, 1.0f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double flop = (double)wgs * 4 * iters * 8 * nacc * (m32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2);
    printf("%-46s %8.3f ms  %7.1f TFLOP/s  = %.3f of 157.3\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main()
{
    float* d;
    hipMalloc(&d, 4096);
    // waves per SIMD = wgs*4 / 1024
    run("16x16x4, 8 independent acc, 1 wave/SIMD", mfma_k<8, false>, 256, 8, false, d);
    run("16x16x4, 8 independent acc, 2 waves/SIMD", mfma_k<8, false>, 512, 8, false, d);
    run("16x16x4, 8 independent acc, 4 waves/SIMD", mfma_k<8, false>, 1024, 8, false, d);
    run("16x16x4, 1 dependent chain, 1 wave/SIMD", mfma_k<1, false>, 256, 1, false, d);
    run("16x16x4, 1 dependent chain, 2 waves/SIMD", mfma_k<1, false>, 512, 1, false, d);
    run("16x16x4, 2 chains, 1 wave/SIMD", mfma_k<2, false>, 256, 2, false, d);
    run("16x16x4, 3 chains, 1 wave/SIMD", mfma_k<3, false>, 256, 3, false, d);
    run("16x16x4, 4 chains, 1 wave/SIMD", mfma_k<4, false>, 256, 4, false, d);
    // the kernels of this library interleave 2-4 accumulators at 2-4 waves per SIMD: does the second wave fill the dependent-issue gap?
    run("16x16x4, 2 chains, 2 waves/SIMD", mfma_k<2, false>, 512, 2, false, d);
    run("16x16x4, 2 chains, 4 waves/SIMD", mfma_k<2, false>, 1024, 2, false, d);
    run("16x16x4, 3 chains, 2 waves/SIMD", mfma_k<3, false>, 512, 3, false, d);
    run("16x16x4, 3 chains, 4 waves/SIMD", mfma_k<3, false>, 1024, 3, false, d);
    run("16x16x4, 4 chains, 2 waves/SIMD", mfma_k<4, false>, 512, 4, false, d);
    run("16x16x4, 4 chains, 4 waves/SIMD", mfma_k<4, false>, 1024, 4, false, d);
    run("16x16x4, 6 chains, 2 waves/SIMD", mfma_k<6, false>, 512, 6, false, d);
    run("16x16x4, 16 chains, 2 waves/SIMD", mfma_k<16, false>, 512, 16, false, d);
    run("32x32x2, 4 independent acc, 1 wave/SIMD", mfma_k<4, true>, 256, 4, true, d);
    run("32x32x2, 4 independent acc, 2 waves/SIMD", mfma_k<4, true>, 512, 4, true, d);
    run("32x32x2, 1 dependent chain, 1 wave/SIMD", mfma_k<1, true>, 256, 1, true, d);
    run("32x32x2, 1 dependent chain, 2 waves/SIMD", mfma_k<1, true>, 512, 1, true, d);
    run("16x16x4, 8 acc, RANDOM operands, 1 wave/SIMD", mfma_k<8, false>, 256, 8, false, d, -1.0f);
    run("16x16x4, 8 acc, RANDOM operands, 2 waves/SIMD", mfma_k<8, false>, 512, 8, false, d, -1.0f);
    run("16x16x4, 8 acc, RANDOM operands, 2 waves/SIMD (longer)", mfma_k<8, false>, 2048, 8, false, d, -1.0f);
    return 0;
}
