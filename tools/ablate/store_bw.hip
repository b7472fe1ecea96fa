// How fast can 2 GiB of fp32 activations be written, and with which cache policy?  (conv_first_k<1> writes 2.15 GB in 0.45 ms = 4.8 TB/s.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate/store_bw.hip -o tools/ablate/bin/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void st_k(float* out, size_t n16_per_block)
{
    // one workgroup = one contiguous run (like a tile of the activation layout), 1 KiB per wave-instruction
    float* base = out + (size_t)blockIdx.x * n16_per_block * 4;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = threadIdx.x; i < n16_per_block; i += 256)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(i * 16), 0, AUX);
}

template <typename K>
static void run(const char* name, K k, float* out, size_t bytes)
{
    const int blocks = 2048 * 4;
    const size_t per = bytes / 16 / blocks;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, per);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, per);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s %7.4f ms  %6.2f TB/s  (%s)\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const size_t bytes = (size_t)2048 * 512 * 16 * 32 * 4;   // 2 GiB: one 16-channel 8^3 activation of a 65 536-leaf chunk
    float* out;
    hipMalloc(&out, bytes);
    run("aux 0 (default)", st_k<0>, out, bytes);
    run("aux 1 (sc0)", st_k<1>, out, bytes);
    run("aux 2 (nt)", st_k<2>, out, bytes);
    run("aux 3 (sc0 nt)", st_k<3>, out, bytes);
    run("aux 16 (sc1)", st_k<16>, out, bytes);
    run("aux 17 (sc0 sc1)", st_k<17>, out, bytes);
    run("aux 18 (sc1 nt)", st_k<18>, out, bytes);
    run("aux 19 (sc0 sc1 nt)", st_k<19>, out, bytes);
    return 0;
}
