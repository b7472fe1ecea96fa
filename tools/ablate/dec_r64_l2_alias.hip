// Upper bound for staging the decoder's 64 -> 64 convs' input in LDS: the library's instantiations of conv_rows16_k for
// dec_res64_conv1 / conv2 as they are (input re-fetched 6.4x / 7.4x from HBM) against the same kernels with every tile reading tile 0's input
// (ABL 64: every activation re-load is an L2 hit — what a perfect staging scheme could at most give back).  Alternating, 10 launches each.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/dec_r64_l2_alias.hip -o tools/ablate/bin/dec_r64_l2_alias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define VQ_ABLATE 1
#include "vq_kernels.h"

// realistic operands (zero-filled buffers draw less power and clock higher): values in [-1, 1)
__global__ void fill_k(float* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
    }
}
static void fill(float* p, size_t n, unsigned seed) { hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed); }

static std::vector<int> steps_rows(int SI, int SO, int KS, int STRIDE, int PAD)
{
    std::vector<int> t;
    for (int od = 0; od < SO; ++od)
        for (int oh = 0; oh < SO; ++oh) {
            const size_t first = t.size();
            for (int kd = 0; kd < KS; ++kd)
                for (int kh = 0; kh < KS; ++kh) {
                    const int id = od * STRIDE - PAD + kd, ih = oh * STRIDE - PAD + kh;
                    if (id < 0 || id >= SI || ih < 0 || ih >= SI) continue;
                    t.insert(t.end(), {(id * SI + ih) * SI, (kd * KS + kh) * KS, (od * SO + oh) * SO, 1 << 8});
                }
            t[first + 3] |= 1;
            t[t.size() - 1] |= 2;
        }
    return t;
}

constexpr size_t LDS = (size_t)2 * (3 * 16 * 64) * 16;
template <typename K>
static void run(const char* name, K k, ConvArgs A, const int4* steps, int nt)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDS, 0, A, steps);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDS, 0, A, steps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-56s %8.4f ms  (%s)\n", name, ms / 10, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int nt = 2048;
    const size_t act = (size_t)nt * 64 * 16 * 32 * 16;
    float *in, *out, *skip, *mean, *rstd, *om, *orr, *w, *bias, *gam, *bet, *csum;
    hipMalloc(&in, act), hipMalloc(&out, act), hipMalloc(&skip, act);
    hipMalloc(&mean, (size_t)nt * 8 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 8 * 32 * 4), hipMalloc(&om, (size_t)nt * 8 * 32 * 4), hipMalloc(&orr, (size_t)nt * 8 * 32 * 4);
    hipMalloc(&csum, (size_t)nt * 64 * 32 * 4);
    hipMalloc(&w, (size_t)27 * 16 * 64 * 16), hipMalloc(&bias, 256), hipMalloc(&gam, 256), hipMalloc(&bet, 256);
    fill(in, act / 4, 1), fill(skip, act / 4, 8), fill(w, (size_t)27 * 16 * 64 * 4, 2), fill(bias, 64, 3), fill(gam, 64, 4), fill(bet, 64, 5);
    fill(mean, (size_t)nt * 8 * 32, 6), fill(rstd, (size_t)nt * 8 * 32, 7);
    hipDeviceSynchronize();
    std::vector<int> t = steps_rows(4, 4, 3, 1, 1);
    int4* steps;
    hipMalloc(&steps, t.size() * 4);
    hipMemcpy(steps, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    ConvArgs A{};
    A.in = in, A.out = out, A.skip = skip, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.out_mean = om, A.out_rstd = orr, A.out_csum = csum, A.n_tiles = nt, A.n_steps = (int)t.size() / 4, A.n_taps = 27;
#define C1(ABL) run("dec_res64_conv1 (library instantiation), ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, ABL, true>, A, steps, nt)
#define C2(ABL) run("dec_res64_conv2 (library instantiation), ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, false, 1, false, 8, false, ABL, true>, A, steps, nt)
    for (int r = 0; r < 3; ++r) { C1(0); C1(64); C2(0); C2(64); }
    C1(63);
    return 0;
}
