// Where does the decoder's 64 -> 64 conv (conv_rows16_k, streamed weights) lose its 16 % to the MFMA peak?  Timing-only variants.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv_rows16_ablate.hip -o tools/ablate/bin/ablate_r16
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
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDS, 0, A, steps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-56s %8.4f ms  (%s)\n", name, ms / 5, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int nt = 2048;
    const size_t act = (size_t)nt * 512 * 16 * 32 * 4 + 4096;   // large enough for every layer tested here (16 ch @ 8^3)
    float *in, *out, *mean, *rstd, *om, *orr, *w, *bias, *gam, *bet;
    hipMalloc(&in, act), hipMalloc(&out, act);
    hipMalloc(&mean, (size_t)nt * 8 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 8 * 32 * 4), hipMalloc(&om, (size_t)nt * 8 * 32 * 4), hipMalloc(&orr, (size_t)nt * 8 * 32 * 4);
    hipMalloc(&w, (size_t)27 * 16 * 64 * 16), hipMalloc(&bias, 256), hipMalloc(&gam, 256), hipMalloc(&bet, 256);
    hipMemset(in, 0, act), hipMemset(mean, 0, (size_t)nt * 8 * 32 * 4), hipMemset(rstd, 0, (size_t)nt * 8 * 32 * 4);
    hipMemset(w, 0, (size_t)27 * 16 * 64 * 16), hipMemset(bias, 0, 256), hipMemset(gam, 0, 256), hipMemset(bet, 0, 256);
    fill(in, act / 4, 1), fill(w, (size_t)27 * 16 * 64 * 4, 2), fill(bias, 64, 3), fill(gam, 64, 4), fill(bet, 64, 5);
    fill(mean, (size_t)nt * 8 * 32, 6), fill(rstd, (size_t)nt * 8 * 32, 7);
    hipDeviceSynchronize();
    std::vector<int> t = steps_rows(4, 4, 3, 1, 1);
    int4* steps;
    hipMalloc(&steps, t.size() * 4);
    hipMemcpy(steps, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    ConvArgs A{};
    A.in = in, A.out = out, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.out_mean = om, A.out_rstd = orr, A.n_tiles = nt, A.n_steps = (int)t.size() / 4, A.n_taps = 27;
#define R(ABL) run("dec res64 conv1, ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, ABL>, A, steps, nt)
    // ABL bits: 1 no barriers, 2 no weight streaming, 4 no LDS A reads, 8 no activation re-loads, 16 no GN transform, 32 no epilogue
    R(0); R(1); R(2); R(8); R(32); R(63);
#define RK2(ABL) run("dec res64 conv1, kw-outer, 2 outputs per run, ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, ABL, true, 2>, A, steps, nt)
#define RK4(ABL) run("dec res64 conv1, kw-outer, 4 outputs per run, ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, ABL, true, 4>, A, steps, nt)
#define RK(ABL) run("dec res64 conv1, kw-outer, ABL " #ABL, conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, ABL, true>, A, steps, nt)
    RK(0); RK(128); RK(8); RK(16); RK(32); RK(64); RK(63);
    RK(0); RK2(0); RK4(0); RK(0); RK2(0); RK4(0); RK2(63); RK4(63);
    {   // encoder res32 conv1: 32 -> 32 k3 @4^3, weights LDS-resident (108 KB), 16 waves
        std::vector<int> t2 = steps_rows(4, 4, 3, 1, 1);
        ConvArgs B = A;
        B.n_taps = 27;
        constexpr size_t LDS32 = (size_t)27 * (2 * 2 * 64) * 16;
#define R32(ABL) { auto k = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 1, false, 16, false, ABL>; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32); \
        hipEvent_t a, b; hipEventCreate(&a), hipEventCreate(&b); \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 15) / 16), dim3(1024), LDS32, 0, B, steps); \
        hipEventRecord(a, 0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 15) / 16), dim3(1024), LDS32, 0, B, steps); \
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
        printf("enc res32 conv1, ABL %-3d                                  %8.4f ms  (%s)\n", ABL, ms / 5, hipGetErrorString(hipGetLastError())); }
        R32(0) R32(8) R32(32) R32(60)
#define R32K(ABL) { auto k = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 1, false, 16, false, ABL, true>; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32); \
        hipEvent_t a, b; hipEventCreate(&a), hipEventCreate(&b); \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 15) / 16), dim3(1024), LDS32, 0, B, steps); \
        hipEventRecord(a, 0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 15) / 16), dim3(1024), LDS32, 0, B, steps); \
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
        printf("enc res32 conv1, kw-outer, ABL %-3d                        %8.4f ms  (%s)\n", ABL, ms / 5, hipGetErrorString(hipGetLastError())); }
        R32K(0) R32K(128)
#define R32K8(ABL) { auto k = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 1, false, 8, false, ABL, true>; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32); \
        hipEvent_t a, b; hipEventCreate(&a), hipEventCreate(&b); \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDS32, 0, B, steps); \
        hipEventRecord(a, 0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDS32, 0, B, steps); \
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
        printf("enc res32 conv1, kw-outer, 8 waves, ABL %-3d               %8.4f ms  (%s)\n", ABL, ms / 5, hipGetErrorString(hipGetLastError())); }
        R32K8(0) R32K(0) R32K8(0)
    }
    {   // encoder down conv: 16 -> 32 k4 s2 @8^3 -> 4^3, weights LDS-resident (128 KB), 8 waves, two-step prefetch
        std::vector<int> t3 = steps_rows(8, 4, 4, 2, 1);
        int4* steps3;
        hipMalloc(&steps3, t3.size() * 4);
        hipMemcpy(steps3, t3.data(), t3.size() * 4, hipMemcpyHostToDevice);
        ConvArgs B = A;
        B.n_taps = 64, B.n_steps = (int)t3.size() / 4;
        constexpr size_t LDSD = (size_t)64 * (1 * 2 * 64) * 16;
#define RD(ABL) { auto k = conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true, 1, true, 8, false, ABL>; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSD); \
        hipEvent_t a, b; hipEventCreate(&a), hipEventCreate(&b); \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDSD, 0, B, steps3); \
        hipEventRecord(a, 0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDSD, 0, B, steps3); \
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
        printf("enc down, ABL %-3d                                         %8.4f ms  (%s)\n", ABL, ms / 5, hipGetErrorString(hipGetLastError())); }
        RD(0) RD(8) RD(32) RD(44)
#define RDK(ABL) { auto k = conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true, 1, false, 8, false, ABL, true>; \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSD); \
        hipEvent_t a, b; hipEventCreate(&a), hipEventCreate(&b); \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDSD, 0, B, steps3); \
        hipEventRecord(a, 0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3((2 * nt + 7) / 8), dim3(512), LDSD, 0, B, steps3); \
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); \
        printf("enc down, kw-outer (no PF2), ABL %-3d                      %8.4f ms  (%s)\n", ABL, ms / 5, hipGetErrorString(hipGetLastError())); }
        RDK(0) RDK(128)
    }
    return 0;
}
