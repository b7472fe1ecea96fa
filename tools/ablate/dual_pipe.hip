// Do the fp32 matrix pipe and the fp32 vector pipe of a SIMD run side by side?  MFMA-only waves (16x16x4 f32, independent accumulators)
// and VALU-only waves (v_fma_f32 with one SGPR operand: the shape of a conv whose weights are wave-uniform) share a workgroup so
// that every SIMD hosts both kinds; reported: each pipe alone, both together, and the clock (s_memtime ticks / wall time).
// Random operands (power draw depends on the data).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate/dual_pipe.hip -o tools/ablate/bin/dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MW = MFMA waves per SIMD, VW = VALU waves per SIMD; wave w of the workgroup sits on SIMD w % 4, (w / 4) < MW -> MFMA wave
template <int MW, int VW, bool PK>
__global__ __launch_bounds__((MW + VW) * 256) void dual_k(float* out, const float* __restrict__ wts, int it_m, int it_v, unsigned long long* clk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() {
        h = h * 1664525u + 1013904223u;
        return ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((wave >> 2) < MW) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float av[8], bv[8];
        for (int k = 0; k < 8; ++k) av[k] = rnd(), bv[k] = rnd();
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(k + i) & 7], bv[(k + 3 * i) & 7], acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].w;
        if (s == 12345.0f) out[0] = s;
    } else {
        // 32 accumulators, 4 x values; per iteration 32 weights arrive by scalar loads and feed 4 x 32 fmas
        float acc[32], x[4];
        for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
        for (int k = 0; k < 4; ++k) x[k] = rnd();
        for (int it = 0; it < it_v; ++it) {
            const float* w = wts + (it & 63) * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (PK) {
#pragma unroll
                    for (int i = 0; i < 32; i += 2) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        f32x2 a = {acc[i], acc[i + 1]}, ww = {w[(i + k) & 31], w[(i + 1 + k) & 31]}, xx = {x[k], x[k]};
                        a = __builtin_elementwise_fma(ww, xx, a);
                        acc[i] = a.x, acc[i + 1] = a.y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(w[(i + k) & 31]), "v"(x[k]));   // (plain fmaf gets SLP-packed into v_pk_fma_f32)
                }
            }
        }
        float s = 0;
        for (int i = 0; i < 32; ++i) s += acc[i];
        if (s == 12345.0f) out[1] = s;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename K>
void run(const char* name, K k, int mw, int vw, int it_m, int it_v, float* d, float* w, unsigned long long* clk)
{
    const int wgs = 256 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(wgs), dim3((mw + vw) * 256), 0, 0, d, w, it_m, it_v, clk);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3((mw + vw) * 256), 0, 0, d, w, it_m, it_v, clk);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double fm = (double)wgs * mw * 4 * it_m * 64 * (16.0 * 16 * 4 * 2), fv = (double)wgs * vw * 4 * it_v * 128 * 64 * 2.0;
    printf("%-44s %8.3f ms  mfma %6.1f TF  valu %6.1f TF  sum %6.1f TF = %.3f of 157.3   (wg0 ticks %llu)\n", name, ms, fm / ms / 1e9, fv / ms / 1e9,
           (fm + fv) / ms / 1e9, (fm + fv) / ms / 1e9 / 157.3, c);
}

int main()
{
    float *d, *w;
    unsigned long long* clk;
    hipMalloc(&d, 4096);
    hipMalloc(&clk, 64);
    std::vector<float> hw(64 * 32);
    unsigned h = 777;
    for (auto& v : hw) h = h * 1664525u + 1013904223u, v = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    hipMalloc(&w, hw.size() * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int IM = 600, IV = 600 * 16;   // one MFMA iteration = 64 x 32 cyc = 2048 cyc; one VALU iteration = 128 fma x 2 cyc = 256 cyc
    run("mfma only, 2 waves/SIMD", dual_k<2, 0, false>, 2, 0, IM, IV, d, w, clk);
    run("mfma only, 1 wave/SIMD", dual_k<1, 0, false>, 1, 0, IM, IV, d, w, clk);
    run("valu only (v_fma, sgpr w), 2 waves/SIMD", dual_k<0, 2, false>, 0, 2, IM, IV / 2, d, w, clk);
    run("valu only (v_fma, sgpr w), 1 wave/SIMD", dual_k<0, 1, false>, 0, 1, IM, IV, d, w, clk);
    run("valu only (v_pk_fma), 2 waves/SIMD", dual_k<0, 2, true>, 0, 2, IM, IV / 2, d, w, clk);
    run("1 mfma + 1 valu wave per SIMD", dual_k<1, 1, false>, 1, 1, IM, IV, d, w, clk);
    run("2 mfma + 2 valu waves per SIMD", dual_k<2, 2, false>, 2, 2, IM / 2, IV / 2, d, w, clk);
    run("2 mfma + 1 valu wave per SIMD", dual_k<2, 1, false>, 2, 1, IM / 2, IV / 2, d, w, clk);
    run("1 mfma + 1 valu (half the valu work)", dual_k<1, 1, false>, 1, 1, IM, IV / 2, d, w, clk);
    run("1 mfma + 1 valu (quarter valu work)", dual_k<1, 1, false>, 1, 1, IM, IV / 4, d, w, clk);
    run("1 mfma + 1 pk valu wave per SIMD", dual_k<1, 1, true>, 1, 1, IM, IV, d, w, clk);
    return 0;
}
