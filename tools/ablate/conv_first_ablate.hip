// Where does conv_first_k's time go?  Timing-only variants of the library's kernel (ABL != 0: results are garbage by design), 2048 tiles.
// (Round 2's version of this file carried its own copies of the kernel — dword loads against the row layout, statistics per row — and
// is in the history; this one instantiates vq_kernels.h's conv_first_k itself.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv_first_ablate.hip -o tools/ablate/bin/ablate_cf
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

// the library's schedule (vq_runtime.hip steps_rows8_kd): one step per (output row, valid kd)
static std::vector<int> steps_rows8_kd()
{
    std::vector<int> t;
    for (int od = 0; od < 8; ++od)
        for (int oh = 0; oh < 8; ++oh) {
            const size_t first = t.size();
            for (int kd = 0; kd < 3; ++kd) {
                const int id = od + kd - 1;
                if (id < 0 || id > 7) continue;
                int mask = 0;
                for (int kh = 0; kh < 3; ++kh)
                    if (oh + kh - 1 >= 0 && oh + kh - 1 <= 7) mask |= 1 << kh;
                t.insert(t.end(), {(id * 8 + oh) * 8, kd, (od * 8 + oh) * 8, mask << 8});
            }
            t[first + 3] |= 1;
            t[t.size() - 1] |= 2;
        }
    return t;
}

template <typename K>
static float run(const char* name, K k, ConvArgs A, const int4* steps)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int g4 = (A.n_tiles + 3) / 4;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(g4), dim3(256), 0, 0, A, steps);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(g4), dim3(256), 0, 0, A, steps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-72s %8.4f ms  (%s)\n", name, ms / 20, hipGetErrorString(hipGetLastError()));
    return ms / 20;
}

int main()
{
    const int nt = 2048;
    float *xr, *out, *w, *bias, *gam, *bet, *mean, *rstd, *om, *orr;
    hipMalloc(&xr, (size_t)nt * VQ_XR_TILE * 4), hipMalloc(&out, (size_t)nt * 512 * 16 * 32 * 4);
    hipMalloc(&w, 9 * 64 * 4), hipMalloc(&bias, 64), hipMalloc(&gam, 64), hipMalloc(&bet, 64);
    hipMalloc(&mean, (size_t)nt * 4 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 4 * 32 * 4), hipMalloc(&om, (size_t)nt * 8 * 32 * 4), hipMalloc(&orr, (size_t)nt * 8 * 32 * 4);
    fill(xr, (size_t)nt * VQ_XR_TILE, 1, 0.0f, 1.0f), fill(w, 9 * 64, 2, -0.2f, 0.2f), fill(bias, 16, 3), fill(gam, 16, 4, 0.5f, 1.5f), fill(bet, 16, 5);
    fill(mean, (size_t)nt * 4 * 32, 6, -0.2f, 0.2f), fill(rstd, (size_t)nt * 4 * 32, 7, 0.8f, 1.6f);
    std::vector<int> t = steps_rows8_kd();
    int4* steps;
    hipMalloc(&steps, t.size() * 4);
    hipMemcpy(steps, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    ConvArgs S{};   // statistics pass: no activation store
    S.in = xr, S.wfrag = w, S.bias_frag = bias, S.out_mean = mean, S.out_rstd = rstd, S.n_tiles = nt, S.n_steps = (int)t.size() / 4;
    ConvArgs N = S;  // normalising pass: recompute, GroupNorm + ReLU, store, statistics of the result
    N.out = out, N.in_mean = mean, N.in_rstd = rstd, N.in_gamma = gam, N.in_beta = bet, N.out_mean = om, N.out_rstd = orr;
    // ABL bits: 1 no input loads inside the loop, 2 no MFMAs, 4 no statistics, 8 no stores
    for (int rep = 0; rep < 2; ++rep) {
        run("statistics pass  conv_first_k<0>", conv_first_k<0, 0>, S, steps);
        run("normalising pass conv_first_k<1>", conv_first_k<1, 0>, N, steps);
    }
    run("statistics pass,  no input loads in the loop", conv_first_k<0, 1>, S, steps);
    run("statistics pass,  no MFMAs", conv_first_k<0, 2>, S, steps);
    run("statistics pass,  no statistics", conv_first_k<0, 4>, S, steps);
    run("statistics pass,  MFMAs only (no loads, no statistics)", conv_first_k<0, 5>, S, steps);
    run("normalising pass, no input loads in the loop", conv_first_k<1, 1>, N, steps);
    run("normalising pass, no MFMAs", conv_first_k<1, 2>, N, steps);
    run("normalising pass, no statistics", conv_first_k<1, 4>, N, steps);
    run("normalising pass, no stores", conv_first_k<1, 8>, N, steps);
    run("normalising pass, no stores, no statistics", conv_first_k<1, 12>, N, steps);
    run("normalising pass, MFMAs only", conv_first_k<1, 13>, N, steps);
    return 0;
}
