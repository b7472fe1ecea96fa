// Where does conv8_lds_k's time go?  Timing-only variants (results are garbage by design).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv8_lds_ablate.hip -o tools/ablate/bin/ablate_c8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_conv8_lds.h"

template <typename K>
static float run(const char* name, K k, ConvArgs A, int grid, int threads)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CONV8);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), LDS_CONV8, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), LDS_CONV8, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.4f ms  (%s)\n", name, ms / 5, hipGetErrorString(hipGetLastError()));
    return ms / 5;
}

// realistic operands (zero-filled buffers draw less power and clock higher: 2.40 vs ~2.33 GHz): values in [-1, 1)
__global__ void fill_k(float* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
    }
}
static void fill(float* p, size_t n, unsigned seed) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed); }

int main()
{
    const int nt = 2048;
    const size_t act = (size_t)nt * 512 * 16 * 32 * 4;
    float *in, *out, *skip, *mean, *rstd, *w, *bias, *gam, *bet;
    double *ps, *pq;
    hipMalloc(&in, act), hipMalloc(&out, act), hipMalloc(&skip, act);
    hipMalloc(&mean, (size_t)nt * 8 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 8 * 32 * 4);
    hipMalloc(&ps, (size_t)nt * 64 * 8 * 32 * 8), hipMalloc(&pq, (size_t)nt * 64 * 8 * 32 * 8);
    hipMalloc(&w, 27 * 64 * 16), hipMalloc(&bias, 64), hipMalloc(&gam, 64), hipMalloc(&bet, 64);
    hipMemset(in, 0, act), hipMemset(skip, 0, act);
    hipMemset(mean, 0, (size_t)nt * 8 * 32 * 4), hipMemset(rstd, 0, (size_t)nt * 8 * 32 * 4);
    hipMemset(w, 0, 27 * 64 * 16), hipMemset(bias, 0, 64), hipMemset(gam, 0, 64), hipMemset(bet, 0, 64);
    fill(in, act / 4, 1), fill(skip, act / 4, 2), fill(w, 27 * 64 * 4, 3), fill(bias, 16, 4), fill(gam, 16, 5), fill(bet, 16, 6);
    fill(mean, (size_t)nt * 8 * 32, 7), fill(rstd, (size_t)nt * 8 * 32, 8);
    hipDeviceSynchronize();
    ConvArgs A{};
    A.in = in, A.out = out, A.skip = skip, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.part_s = ps, A.part_q = pq, A.n_tiles = nt;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    printf("CUs %d\n", cus);
#define R(RESID, STATS, NW, ABL) run("RESID " #RESID " STATS " #STATS " NW " #NW " ABL " #ABL, conv8_lds_k<RESID, STATS, NW, ABL>, A, cus, NW * 64)
    // ABL bits: 1 no barriers, 2 no epilogue stores/stats, 4 no plane write + prefetch, 8 no LDS B reads (stale registers), 16 no MFMAs
    R(false, true, 8, 0); R(false, true, 8, 1); R(false, true, 8, 4);
    R(true, false, 8, 0); R(true, false, 16, 0); R(true, false, 8, 0); R(true, false, 16, 0);
    run("STATS, 8 waves, border-row loaders", conv8_lds_k<false, true, 8, 0, true>, A, cus, 512);
    run("RESID, 8 waves, border-row loaders", conv8_lds_k<true, false, 8, 0, true>, A, cus, 512);
#define RL(ABL) run("STATS, border-row loaders, ABL " #ABL, conv8_lds_k<false, true, 8, ABL, true>, A, cus, 512)
    RL(1); RL(4);
    R(true, false, 16, 0); R(true, false, 16, 1); R(true, false, 16, 4); R(true, false, 16, 5); R(false, true, 16, 0); R(true, false, 16, 0);
    return 0;
}
