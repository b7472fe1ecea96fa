// Where does the time of the folded decoder tail with D x H zero skipping go (tail_rows16_k, vq_tail_rows.h)?  Timing-only variants
// (ABL != 0: garbage results by design), 2048 tiles = 65 536 leaves, random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/tail_rows_ablate.hip -o tools/ablate/bin/ablate_tail_rows
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_tail_groups.h"

__global__ void fill_k(float* p, size_t n, unsigned seed, float lo, float hi)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = lo + (float)(h & 0xffffff) * ((hi - lo) / 16777216.0f);
    }
}
static void fill(float* p, size_t n, unsigned seed, float lo = -1.0f, float hi = 1.0f) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed, lo, hi); }

template <typename K>
static float rung(const char* name, K k, ConvArgs A)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL_GROUPS);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int g = (2 * A.n_tiles + 7) / 8;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), LDS_TAIL_GROUPS, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), LDS_TAIL_GROUPS, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.4f ms  (%s)\n", name, ms / 10, hipGetErrorString(hipGetLastError()));
    return ms / 10;
}
template <typename K>
static float run32(const char* name, K k, ConvArgs A)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL_ROWS);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int g = (A.n_tiles + 3) / 4;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(256), LDS_TAIL_ROWS, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(256), LDS_TAIL_ROWS, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.4f ms  (%s)\n", name, ms / 10, hipGetErrorString(hipGetLastError()));
    return ms / 10;
}
template <typename K>
static float run(const char* name, K k, ConvArgs A)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL_ROWS);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int g = (2 * A.n_tiles + 7) / 8;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), LDS_TAIL_ROWS, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), LDS_TAIL_ROWS, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.4f ms  (%s)\n", name, ms / 10, hipGetErrorString(hipGetLastError()));
    return ms / 10;
}

int main()
{
    const int nt = 2048;
    float *in, *out, *w, *bias, *csum, *fc0, *fc2;
    const size_t wn = (size_t)TG_STREAM_SLICES * (TG_SLICE / 4);   // (the larger of the two streams)
    hipMalloc(&in, (size_t)nt * 64 * 16 * 32 * 16), hipMalloc(&out, (size_t)nt * 32 * 512 * 4), hipMalloc(&w, wn * 4), hipMalloc(&bias, 512 * 4);
    hipMalloc(&csum, (size_t)nt * 64 * 32 * 4), hipMalloc(&fc0, 16 * 64 * 4), hipMalloc(&fc2, 64 * 16 * 4);
    fill(in, (size_t)nt * 64 * 16 * 32 * 4, 1), fill(w, wn, 2, -0.05f, 0.05f), fill(bias, 512, 3), fill(csum, (size_t)nt * 64 * 32, 4, -8.0f, 8.0f);
    fill(fc0, 16 * 64, 5, -0.2f, 0.2f), fill(fc2, 64 * 16, 6, -0.2f, 0.2f);
    hipDeviceSynchronize();
    ConvArgs A{};
    A.in = in, A.out = out, A.wfrag = w, A.bias_frag = bias, A.se_csum = csum, A.se_fc0 = fc0, A.se_fc2 = fc2, A.n_tiles = nt, A.n_leaves = (int64_t)nt * 32;
#define T(ABL) run("folded tail (rows16), ABL " #ABL, tail_rows16_k<ABL>, A)
    // ABL bits: 1 no barriers, 2 no weight streaming, 4 no LDS fragment reads, 8 no activation re-loads, 16 no gate multiply, 32 no epilogue, 64 no lane swap, 128 no MFMAs
#define G(ABL) rung("folded tail (groups16), ABL " #ABL, tail_groups16_k<ABL>, A)
    G(0); G(0); G(1); G(2); G(4); G(8); G(256); G(16); G(80); G(32); G(128); G(0);
    T(0); T(0); T(1); T(2); T(4); T(8); T(16); T(80); T(32); T(128); T(0);
#define U(ABL) run32("folded tail (rows32, 1 wave/SIMD), ABL " #ABL, tail_rows32_k<ABL>, A)
    U(0); U(0); U(1); U(2); U(4); U(8); U(16); U(80); U(32); U(128); U(0);
    return 0;
}
