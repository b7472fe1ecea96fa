// Where does vq_folded_k's time go, and how should the winning tile's scores be kept?  ABL != 0 variants are timing-only (garbage
// results by design); the SCAN variants must produce the SAME index bytes (checksum printed).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/vq_folded_ablate.hip -o tools/ablate/bin/ablate_vq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_kernels.h"

__global__ void fill_k(float* p, size_t n, unsigned seed, float lo, float hi)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = lo + (float)(h & 0xffffff) * ((hi - lo) / 16777216.0f);
    }
}
static void fill(float* p, size_t n, unsigned seed, float lo = -1.0f, float hi = 1.0f) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed, lo, hi); }
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

static VqArgs G;
template <typename K>
static float run(const char* name, K k, int nt, int ysplit, bool check)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipMemset(G.idx, 0, (size_t)nt * 32 * 64);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((nt + 7) / 8, ysplit), dim3(512), 0, 0, G);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3((nt + 7) / 8, ysplit), dim3(512), 0, 0, G);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (check) printf("%-44s %8.4f ms  idx %016llx (%s)\n", name, ms / 20, checksum(G.idx, (size_t)nt * 32 * 64 / 4), hipGetErrorString(hipGetLastError()));
    else printf("%-44s %8.4f ms  (%s)\n", name, ms / 20, hipGetErrorString(hipGetLastError()));
    return ms / 20;
}

int main()
{
    const int nt = 2048;
    float *in, *gate, *ep, *ck;
    uint8_t* idx;
    hipMalloc(&in, (size_t)nt * 64 * 8 * 32 * 16), hipMalloc(&gate, (size_t)nt * 32 * 32 * 4), hipMalloc(&ep, 4 * 8 * 64 * 16), hipMalloc(&ck, 8 * 2 * 4 * 16);
    hipMalloc(&idx, (size_t)nt * 32 * 64);
    fill(in, (size_t)nt * 64 * 8 * 32 * 4, 1), fill(gate, (size_t)nt * 32 * 32, 2, 0.2f, 0.9f), fill(ep, 4 * 8 * 64 * 4, 3), fill(ck, 8 * 2 * 4 * 4, 4, -2.0f, 0.0f);
    hipDeviceSynchronize();
    G.in = in, G.se_gate = gate, G.epfrag = ep, G.ck_frag = ck, G.idx = idx, G.n_leaves = (int64_t)nt * 32, G.n_tiles = nt;
    for (int rep = 0; rep < 3; ++rep) {
        run("vq_folded_k<8> SCAN 0 (selects)", vq_folded_k<8, 0>, nt, 2, true);
        run("vq_folded_k<8> SCAN 1 (masked moves)", vq_folded_k<8, 1>, nt, 2, true);
        run("vq_folded_k<8> SCAN 2 (scores kept in LDS)", vq_folded_k<8, 2>, nt, 2, true);
    }
    // ABL bits: 1 no scan at all, 2 no score keeping (tree + compare + tile number only), 4 no MFMAs
    run("SCAN 0, no score keeping", vq_folded_k<8, 0, 2>, nt, 2, false);
    run("SCAN 0, no scan", vq_folded_k<8, 0, 1>, nt, 2, false);
    run("SCAN 0, no MFMAs", vq_folded_k<8, 0, 4>, nt, 2, false);
    run("no scan, no MFMAs (loads, gates, stores)", vq_folded_k<8, 0, 5>, nt, 2, false);
    return 0;
}
