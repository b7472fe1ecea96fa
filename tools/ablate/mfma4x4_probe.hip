// v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) operand / result lane layout, (2) is it an exact fmaf, (3) issue rate against
// v_mfma_f32_16x16x4_f32, alone and with fp64 / packed-fp32 statistics-style VALU work interleaved (the first conv's mix).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ablate/mfma4x4_probe.hip -o tools/ablate/bin/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- layout: every lane supplies its own a and b; result = 4 regs per lane ----
__global__ void layout_k(const float* a, const float* b, const float* c, float* d)
{
    const int l = threadIdx.x;
    f32x4 acc = ((const f32x4*)c)[l];
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    ((f32x4*)d)[l] = acc;
}

// ---- rate ----
// MODE 0: 4x4x1 only; 1: 16x16x4 only; 2: 4x4x1 + NV fp64 statistics triples (cvt, add, fma) per 24 MFMAs; 3: 16x16x4 + the same per 8 MFMAs
// (24 4x4x1 = 8 16x16x4 in useful MACs when the 16x16x4 K slot 3 is padding); 4 / 5: the same with packed-fp32 partial sums (pk_add, pk_fma per 2 values)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void rate_k(float* out, int iters)
{
    float av[8], bv[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 8; ++k) {
        h = h * 1664525u + 1013904223u;
        av[k] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
        h = h * 1664525u + 1013904223u;
        bv[k] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    double s = 0.0, q = 0.0;
    f32x2 ps = {0, 0}, pq = {0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[(k + i) & 7], bv[(k + 3 * i) & 7], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 7], bv[(3 * i) & 7], acc[i], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double d = (double)acc[v & 7][v >> 3 & 3];
                s += d;
                q = fma(d, d, q);
            }
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int v = 0; v < NV; v += 2) {
                const f32x2 y = {acc[v & 7][v >> 3 & 3], acc[(v + 1) & 7][(v + 1) >> 3 & 3]};
                ps = ps + y;
                pq = __builtin_elementwise_fma(y, y, pq);
            }
        }
    }
    float t = (float)s + (float)q + ps.x + ps.y + pq.x + pq.y;
    for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].w;
    if (t == 12345.0f) out[0] = t;
}

template <typename K>
static void run(const char* name, K k, int wgs, double macs_per_iter, float* d)
{
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(a, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double cyc = ms * 1e-3 * 2.4e9 / iters / (wgs * 4 / 1024.0);   // cycles per iteration per wave at 2.4 GHz, per SIMD time share
    printf("%-72s %8.3f ms  %7.1f cycles / iteration / wave  (useful %.1f TFLOP/s)\n", name, ms, cyc, 2 * macs_per_iter * wgs * 4 * iters / ms / 1e9);
}

int main()
{
    std::vector<float> a(64), b(64), c(256), d(256);
    for (int l = 0; l < 64; ++l) a[l] = 1.0f + l, b[l] = 100.0f * (1 + l);
    for (int i = 0; i < 256; ++i) c[i] = 0.0f;
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 256), hipMalloc(&db, 256), hipMalloc(&dc, 1024), hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // expected if lane l = 4 * block + i supplies A[i] and B[j = i] of its block and holds D[r][j]: d[l][r] = a[4 * (l / 4) + r] * b[l]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
            if (d[l * 4 + r] != a[4 * (l / 4) + r] * b[l]) ++bad;
    printf("layout: reg r of lane l = A(lane 4*(l/4)+r) * B(lane l): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) {
        for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
    }
    // exact fmaf: one product that needs the unrounded a*b
    {
        for (int l = 0; l < 64; ++l) a[l] = 1.0f + ldexpf(1.0f, -12) * (l + 1), b[l] = 1.0f - ldexpf(1.0f, -13) * (l + 3);
        for (int i = 0; i < 256; ++i) c[i] = -1.0f;
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        int badf = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (d[l * 4 + r] != fmaf(a[4 * (l / 4) + r], b[l], -1.0f)) ++badf;
        printf("exact fmaf: %s (%d mismatches)\n", badf ? "NO" : "yes", badf);
    }
    float* o;
    hipMalloc(&o, 4096);
    const double m4 = 24 * 16.0 * 16, m16 = 8 * 16.0 * 16 * 3;   // useful MACs per iteration per wave (K slot 3 of 16x16x4 = padding)
    run("4x4x1, 24 per iteration, 1 wave/SIMD", rate_k<0, 0>, 256, m4, o);
    run("4x4x1, 24 per iteration, 2 waves/SIMD", rate_k<0, 0>, 512, m4, o);
    run("16x16x4, 8 per iteration (3 of 4 K slots useful), 1 wave/SIMD", rate_k<1, 0>, 256, m16, o);
    run("16x16x4, 8 per iteration, 2 waves/SIMD", rate_k<1, 0>, 512, m16, o);
    run("4x4x1 x24 + 32 fp64 statistic triples, 2 waves/SIMD", rate_k<2, 32>, 512, m4, o);
    run("16x16x4 x8 + 32 fp64 statistic triples, 2 waves/SIMD", rate_k<3, 32>, 512, m16, o);
    run("4x4x1 x24 + 32 values as packed fp32 partials, 2 waves/SIMD", rate_k<4, 32>, 512, m4, o);
    run("16x16x4 x8 + 32 values as packed fp32 partials, 2 waves/SIMD", rate_k<5, 32>, 512, m16, o);
    run("4x4x1 x24 + 16 fp64 statistic triples, 2 waves/SIMD", rate_k<2, 16>, 512, m4, o);
    // the first conv's real ratio: 32 output values per 7.56 (kd,kh) groups = 4 values per group of 24 / 8 MFMAs
    run("4x4x1 x24 + 4 fp64 statistic triples, 2 waves/SIMD", rate_k<2, 4>, 512, m4, o);
    run("16x16x4 x8 + 4 fp64 statistic triples, 2 waves/SIMD", rate_k<3, 4>, 512, m16, o);
    run("4x4x1 x24 + 4 values as packed fp32 partials, 2 waves/SIMD", rate_k<4, 4>, 512, m4, o);
    run("16x16x4 x8 + 4 values as packed fp32 partials, 2 waves/SIMD", rate_k<5, 4>, 512, m16, o);
    return 0;
}
