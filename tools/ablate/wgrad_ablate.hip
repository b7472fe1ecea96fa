// What bounds the weight-gradient kernel wgrad32_k (training, 43-51 % of the fp32 MFMA peak)?  Timing-only variants, one binary per -DWGRAD_ABL=n:
//   for a in 0 1 2 4 5; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWGRAD_ABL=$a -I vqvdb_amd/csrc tools/ablate/wgrad_ablate.hip -o tools/ablate/bin/ablate_wgrad_$a; done
// WGRAD_ABL bits: 1 one global fetch only, 2 no MFMAs, 4 no LDS staging / barriers.  -DROWS4_PIPE=0|1|2: operand read-ahead of wgrad_rows4_k
// (0: 109.6 / 101.1 TFLOP/s for the stem / 64->64 layer at 8192 leaves, 1: 116.5 / 106.4, 2: 116.1 / 106.7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_device.h"
#include "vq_grad_kernels.h"

__global__ void fill_k(float* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
    }
}
static void fill(float* p, size_t n, unsigned seed) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed); }

static void wsteps_k3_4(std::vector<int>& pairs, std::vector<int>& start)
{
    for (int t = 0; t < 27; ++t) {
        start.push_back((int)pairs.size() / 2);
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        for (int p = 0; p < 64; ++p) {
            const int id = (p >> 4) + kd - 1, ih = ((p >> 2) & 3) + kh - 1, iw = (p & 3) + kw - 1;
            if (id < 0 || id > 3 || ih < 0 || ih > 3 || iw < 0 || iw > 3) continue;
            pairs.push_back((id * 4 + ih) * 4 + iw), pairs.push_back(p);
        }
    }
    start.push_back((int)pairs.size() / 2);
}

template <int CIN, int COUT>
static void run(const char* name, int nt, int ng, int chunks, WgradArgs A)
{
    constexpr int NT = ((COUT + 31) / 32) * ((CIN + 31) / 32) * 64;
    A.n_tiles = nt, A.tiles_per_group = (nt + ng - 1) / ng, A.KT = 27;
    auto k = wgrad32_k<CIN, COUT, 64, 64, 0, 0>;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(27, ng, chunks), dim3(NT), 0, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(27, ng, chunks), dim3(NT), 0, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    const double flop = 2.0 * 1000.0 * CIN * COUT * 32.0 * nt;   // 1000 valid (ip, po) pairs of the 27 taps
    printf("ABL %d  %-22s nt %4d groups %3d chunks %d : %8.4f ms  %6.1f TFLOP/s  (%s)\n", WGRAD_ABL, name, nt, ng, chunks, ms, flop / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

template <int CIN, int COUT>
static void run_rows(const char* name, int nt, int Q, WgradArgs A)
{
    constexpr int NT = rows4_threads<CIN, COUT>();
    A.n_tiles = nt, A.KT = 27;
    auto k = wgrad_rows4_k<CIN, COUT, 0, 0>;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(Q), dim3(NT), 0, 0, A);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(Q), dim3(NT), 0, 0, A);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    const double flop = 2.0 * 1000.0 * CIN * COUT * 32.0 * nt;
    printf("rows4 ABL %d %-22s nt %4d slices %4d          : %8.4f ms  %6.1f TFLOP/s  (%s)\n", WGRAD_ABL, name, nt, Q, ms, flop / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int NTMAX = 256;
    float *dy, *x, *part;
    const size_t nx = (size_t)NTMAX * 64 * 128 * 32, npart = (size_t)128 * 27 * 128 * 64;   // >= (1024 + 8) slots x 3 taps of rows4
    hipMalloc(&dy, nx * 4), hipMalloc(&x, nx * 4), hipMalloc(&part, npart * 4);
    fill(dy, nx, 1), fill(x, nx, 2);
    std::vector<int> pairs, start;
    wsteps_k3_4(pairs, start);
    int *dp, *ds;
    hipMalloc(&dp, pairs.size() * 4), hipMalloc(&ds, start.size() * 4);
    hipMemcpy(dp, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice), hipMemcpy(ds, start.data(), start.size() * 4, hipMemcpyHostToDevice);
    WgradArgs A{};
    A.dy = dy, A.x = x, A.wsteps = (const int2*)dp, A.tap_start = ds, A.part = part;
    for (int nt : {64, 256}) {
        run<128, 64>("stem 128->64", nt, 32, 1, A);
        run<128, 64>("stem 128->64", nt, std::min(nt, 128), 1, A);
        run<64, 64>("res64 64->64", nt, 32, 2, A);
        run<64, 64>("res64 64->64", nt, std::min(nt, 128), 2, A);
        run<32, 32>("res32 32->32", nt, 32, 4, A);
        run<32, 32>("res32 32->32", nt, std::min(nt, 128), 4, A);
#if 1
        for (int Q : {256, 512, 768}) run_rows<128, 64>("stem 128->64", nt, Q, A);
        for (int Q : {256, 512, 768, 1024}) run_rows<64, 64>("res64 64->64", nt, Q, A);
        for (int Q : {256, 512, 1024}) run_rows<32, 32>("res32 32->32", nt, Q, A);
#endif
    }
    return 0;
}
