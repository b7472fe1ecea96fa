// Where does the folded decoder tail's time go (conv_mfma32_k OUTMODE 2)?  Timing-only variants (ABL != 0: garbage results
// by design) of the library's instantiation, 2048 tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv_mfma32_ablate.hip -o tools/ablate/bin/ablate_tail
#include <hip/hip_runtime.h>
#include <algorithm>
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

constexpr size_t LDS_TAIL = (size_t)2 * (8 * 4 * 64) * 16;   // 2 x 32 KB weight window (vq_runtime.hip LDS_DEC_TAIL)
template <typename K>
static float run(const char* name, K k, ConvArgs A, const int4* steps)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int g8 = (A.n_tiles + 7) / 8;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(g8), dim3(512), LDS_TAIL, 0, A, steps);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(g8), dim3(512), LDS_TAIL, 0, A, steps);
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
    // the library's schedule (vq_runtime.hip build_folded_tail): slab d visits the input positions of planes max(0,d-2) .. min(3,d+2)
    std::vector<int> st;
    int nfrag = 0;
    for (int d = 0; d < 4; ++d) {
        const size_t first = st.size();
        for (int p = std::max(0, d - 2) * 16; p < (std::min(3, d + 2) + 1) * 16; ++p) st.insert(st.end(), {p, nfrag++, d, 1 << 8});
        st[first + 3] |= 1;
        st[st.size() - 1] |= 2;
    }
    float *in, *out, *w, *bias, *csum, *fc0, *fc2;
    hipMalloc(&in, (size_t)nt * 64 * 16 * 32 * 16), hipMalloc(&out, (size_t)nt * 32 * 512 * 4), hipMalloc(&w, (size_t)nfrag * 8 * 4 * 64 * 16), hipMalloc(&bias, 4 * 4 * 8 * 16);
    hipMalloc(&csum, (size_t)nt * 64 * 32 * 4), hipMalloc(&fc0, 16 * 64 * 4), hipMalloc(&fc2, 64 * 16 * 4);
    fill(in, (size_t)nt * 64 * 16 * 32 * 4, 1), fill(w, (size_t)nfrag * 8 * 4 * 64 * 4, 2, -0.05f, 0.05f), fill(bias, 4 * 4 * 8 * 4, 3), fill(csum, (size_t)nt * 64 * 32, 4, -8.0f, 8.0f);
    fill(fc0, 16 * 64, 5, -0.2f, 0.2f), fill(fc2, 64 * 16, 6, -0.2f, 0.2f);
    int4* steps;
    hipMalloc(&steps, st.size() * 4);
    hipMemcpy(steps, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    ConvArgs A{};
    A.in = in, A.out = out, A.wfrag = w, A.bias_frag = bias, A.se_csum = csum, A.se_fc0 = fc0, A.se_fc2 = fc2, A.n_tiles = nt, A.n_leaves = (int64_t)nt * 32;
    A.n_steps = (int)st.size() / 4;
#define T(ABL) run("folded tail, ABL " #ABL, conv_mfma32_k<64, 128, 64, 4, 8, true, 1, 2, 0, false, 0, false, 2, ABL>, A, steps)
    // ABL bits: 1 no barriers, 2 no weight streaming, 4 no LDS A-fragment reads, 8 no activation re-loads, 16 no gate multiply, 32 no epilogue,
    // 64 no row-blocked totals, 128 no MFMAs
#define TW(ABL, WM) run("folded tail, ABL " #ABL " WMODE " #WM, conv_mfma32_k<64, 128, 64, 4, 8, true, 1, 2, 0, false, 0, false, 2, ABL, WM>, A, steps)
    T(0); TW(0, 0); TW(0, 1); TW(0, 0); TW(0, 1);   // (T = the library's WMODE 0)
    T(1); T(2); T(3); T(4); T(8); T(16); T(32); T(64); T(127); T(128); TW(8, 0); TW(2, 0); T(0);
    return 0;
}
