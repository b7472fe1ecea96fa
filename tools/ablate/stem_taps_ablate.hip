// Where does stem_taps_k's time go, and which gather pipelining / wave mapping is fastest?  Variants with ABL != 0 are timing-only
// (results garbage by design); the ABL == 0 variants must produce the SAME d2 / statistics bits (checksums printed).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/stem_taps_ablate.hip -o tools/ablate/bin/ablate_stem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_stem_taps.h"

__global__ void fill_k(float* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
    }
}
__global__ void fill_idx_k(uint8_t* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = (uint8_t)(h >> 8);
    }
}
__global__ void checksum_k(const unsigned* p, size_t n, unsigned long long* out)
{
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)p[i] * (unsigned)(i % 1000003u + 1);
    atomicAdd(out, s);
}
static unsigned long long checksum(const void* p, size_t n_words)
{
    unsigned long long* d;
    hipMalloc(&d, 8), hipMemset(d, 0, 8);
    hipLaunchKernelGGL(checksum_k, dim3(1024), dim3(256), 0, 0, (const unsigned*)p, n_words, d);
    unsigned long long h;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return h;
}

static StemFusedArgs G;
static size_t d2_words, st_words;
template <typename K>
static float run(const char* name, K k, int grid, bool check)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_STEM_TAPS);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipMemset(G.d2, 0, d2_words * 4);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_STEM_TAPS, 0, G);
    hipEventRecord(a, 0);
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_STEM_TAPS, 0, G);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (check) printf("%-56s %8.4f ms  d2 %016llx mean %016llx rstd %016llx (%s)\n", name, ms / 6, checksum(G.d2, d2_words), checksum(G.out_mean, st_words),
                      checksum(G.out_rstd, st_words), hipGetErrorString(hipGetLastError()));
    else printf("%-56s %8.4f ms  (%s)\n", name, ms / 6, hipGetErrorString(hipGetLastError()));
    return ms / 6;
}

int main()
{
    const int nt = 2048;
    const int64_t n = (int64_t)nt * 32;
    float *T, *bias, *gam, *bet, *d2, *mean, *rstd;
    uint8_t* idx;
    d2_words = (size_t)nt * 64 * 16 * 32 * 4, st_words = (size_t)nt * 8 * 32;
    hipMalloc(&T, 27 * 256 * 64 * 4), hipMalloc(&bias, 256), hipMalloc(&gam, 256), hipMalloc(&bet, 256);
    hipMalloc(&d2, d2_words * 4), hipMalloc(&mean, st_words * 4), hipMalloc(&rstd, st_words * 4), hipMalloc(&idx, n * 64);
    hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, T, (size_t)27 * 256 * 64, 1u);
    hipLaunchKernelGGL(fill_k, dim3(1), dim3(64), 0, 0, bias, (size_t)64, 2u);
    hipLaunchKernelGGL(fill_k, dim3(1), dim3(64), 0, 0, gam, (size_t)64, 3u);
    hipLaunchKernelGGL(fill_k, dim3(1), dim3(64), 0, 0, bet, (size_t)64, 4u);
    hipLaunchKernelGGL(fill_idx_k, dim3(1024), dim3(256), 0, 0, idx, (size_t)n * 64, 5u);
    hipDeviceSynchronize();
    G.idx = idx, G.T = T, G.bias = bias, G.gamma = gam, G.beta = bet, G.d2 = d2, G.out_mean = mean, G.out_rstd = rstd, G.n_leaves = n, G.n_tiles = nt;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    printf("CUs %d, %d tiles\n", cus, nt);
#define V(PIPE, HMAP, RELAX, STAG) run("PIPE " #PIPE " HMAP " #HMAP " RELAX " #RELAX " STAG " #STAG, stem_taps_k<PIPE, HMAP, 0, RELAX, STAG>, cus, true)
#define AB(PIPE, HMAP, ABL) run("PIPE " #PIPE " HMAP " #HMAP " ABL " #ABL, stem_taps_k<PIPE, HMAP, ABL>, cus, false)
#define VC(PIPE, HMAP, RELAX, CHK) run("PIPE " #PIPE " HMAP " #HMAP " RELAX " #RELAX " CHK " #CHK, stem_taps_k<PIPE, HMAP, 0, RELAX, 0, CHK>, cus, true)
    V(0, 0, false, 0); V(0, 1, true, 0); VC(0, 1, true, true); VC(0, 0, true, true); VC(1, 1, true, true); V(0, 1, true, 0); VC(0, 1, true, true);
    // ABL bits: 1 no gather reads, 2 no adds, 4 no table DMA, 8 no epilogue, 16 no per-tap barrier, 32 no output stores
    AB(0, 1, 1); AB(0, 1, 4); AB(0, 1, 8); AB(0, 1, 16); AB(0, 1, 32);
    return 0;
}
