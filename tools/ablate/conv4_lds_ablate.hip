// Where does conv4_lds_k's time go?  Timing-only variants (results are garbage by design) next to the row kernel it replaces.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv4_lds_ablate.hip -o tools/ablate/bin/ablate_c4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_conv4_lds.h"
#include "vq_convdown_lds.h"

template <typename K>
static float run(const char* name, K k, ConvArgs A, int grid, int threads, size_t lds)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.4f ms  (%s)\n", name, ms / 6, hipGetErrorString(hipGetLastError()));
    return ms / 6;
}

__global__ void fill_k(float* p, size_t n, unsigned seed, float lo, float hi)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = lo + (float)(h & 0xffffff) * ((hi - lo) / 16777216.0f);
    }
}
static void fill(float* p, size_t n, unsigned seed, float lo = -1.0f, float hi = 1.0f) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed, lo, hi); }

int main()
{
    const int nt = 2048;
    const size_t act = (size_t)nt * 64 * 32 * 32 * 4;
    float *in, *out, *skip, *mean, *rstd, *w, *bias, *gam, *bet, *csum;
    hipMalloc(&in, act), hipMalloc(&out, act), hipMalloc(&skip, act);
    hipMalloc(&mean, (size_t)nt * 8 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 8 * 32 * 4), hipMalloc(&csum, (size_t)nt * 32 * 32 * 4);
    hipMalloc(&w, 27 * 4 * 64 * 16), hipMalloc(&bias, 128), hipMalloc(&gam, 128), hipMalloc(&bet, 128);
    fill(in, act / 4, 1), fill(skip, act / 4, 2), fill(w, 27 * 4 * 64 * 4, 3, -0.1f, 0.1f), fill(bias, 32, 4), fill(gam, 32, 5, 0.5f, 1.5f), fill(bet, 32, 6);
    fill(mean, (size_t)nt * 8 * 32, 7, -0.2f, 0.2f), fill(rstd, (size_t)nt * 8 * 32, 8, 0.8f, 1.6f);
    hipDeviceSynchronize();
    ConvArgs A{};
    A.in = in, A.out = out, A.skip = skip, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.out_mean = mean, A.out_rstd = rstd, A.out_csum = csum, A.n_tiles = nt;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    printf("CUs %d\n", cus);
    // (mean / rstd double as output buffers of the STATS variant: timing only)
#define R(RESID, STATS, CSUM, ABL) run("conv4_lds_k RESID " #RESID " STATS " #STATS " CSUM " #CSUM " ABL " #ABL, conv4_lds_k<RESID, STATS, CSUM, ABL>, A, cus, 512, LDS_CONV4)
    // ABL bits: 1 no barriers, 2 no epilogue, 4 no plane write / prefetch, 8 no LDS B reads, 16 no A-fragment loads, 32 no MFMAs
    R(false, true, false, 0); R(true, false, true, 0); R(false, true, false, 0); R(true, false, true, 0);
#define RS(RESID, STATS, CSUM, STG) run("conv4_lds_k RESID " #RESID " STATS " #STATS " CSUM " #CSUM " STG " #STG, conv4_lds_k<RESID, STATS, CSUM, 0, STG>, A, cus, 512, LDS_CONV4)
    RS(false, true, false, 0); RS(true, false, true, 0); RS(false, true, false, 1); RS(true, false, true, 1); RS(false, true, false, 0); RS(true, false, true, 0);
    R(false, true, false, 1); R(false, true, false, 2); R(false, true, false, 4); R(false, true, false, 8); R(false, true, false, 16); R(false, true, false, 24);
    R(false, true, false, 31); R(false, true, false, 32);
    {   // the down conv: 16 -> 32 k4 s2, input 8^3 x 16 channels (its own input buffer; outputs and statistics land in the 4^3 buffers: timing only)
        float* wd;
        hipMalloc(&wd, 64 * 2 * 64 * 16);
        fill(wd, 64 * 2 * 64 * 4, 9, -0.1f, 0.1f);
        hipDeviceSynchronize();
        float* din;
        const size_t dact = (size_t)nt * 512 * 16 * 32 * 4;     // 8^3 x 16 channels per leaf
        hipMalloc(&din, dact);
        fill(din, dact / 4, 10);
        hipDeviceSynchronize();
        ConvArgs D = A;
        D.wfrag = wd, D.in = din;
#define RD(ABL, TWOB) run("conv_down_lds_k 16 waves ABL " #ABL " TWOB " #TWOB, conv_down_lds_k<ABL, TWOB>, D, cus, 1024, LDS_CONVDOWN)
#define RD8(ABL) run("conv_down_lds_k 8 waves ABL " #ABL, conv_down_lds_k<ABL, false, 1, 8>, D, cus, 512, LDS_CONVDOWN)
        RD8(0); RD(0, false); RD8(0); RD(0, false); RD(0, true); RD(1, false); RD(2, false); RD(4, false); RD(8, false); RD(16, false);
    }
    return 0;
}
