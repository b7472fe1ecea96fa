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
    ConvArgs A{};
    A.in = in, A.out = out, A.skip = skip, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.part_s = ps, A.part_q = pq, A.n_tiles = nt;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    printf("CUs %d\n", cus);
#define R(RESID, STATS, NW, ABL) run("RESID " #RESID " STATS " #STATS " NW " #NW " ABL " #ABL, conv8_lds_k<RESID, STATS, NW, ABL>, A, cus, NW * 64)
    // ABL bits: 1 no barriers, 2 no epilogue stores/stats, 4 no plane write + prefetch, 8 no LDS B reads (stale registers), 16 no MFMAs
    R(false, true, 8, 0); R(false, true, 8, 1); R(false, true, 8, 4); R(false, true, 8, 8); R(false, true, 8, 16);
    R(true, false, 8, 0); R(true, false, 16, 0);
    run("STATS, 8 waves, border-row loaders", conv8_lds_k<false, true, 8, 0, true>, A, cus, 512);
    run("RESID, 8 waves, border-row loaders", conv8_lds_k<true, false, 8, 0, true>, A, cus, 512);
    return 0;
}
