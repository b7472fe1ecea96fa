// vq_train_tail.h — training THROUGH the folded decoder tail.
//
// up_conv (64->256, k3 @4^3) -> PixelShuffle3D(2) -> final (32->1, k3 @8^3) has no nonlinearity in between (VQVAE_v2.py:274-275): it is
// linear in its input and bilinear in (up_conv, final) weights.  Inference folds it into one operator Wc[ov][p][ci] (output voxel,
// input position, input channel; 884 736 structurally non-zero entries) — build_folded_tail in vq_runtime.hip.  The training step does
// the same instead of materialising the 256-channel tensor (3.1 of 10.1 ms at 2048 leaves):
//
//   forward       pre = Wc x' + bc            the inference tail kernels on fragments rebuilt ON THE DEVICE from the live parameters
//   data gradient dx' = Wc^T dpre             conv_mfma32_k with the output slabs as "input positions" (128 voxels = channels)
//   folded wgrad  dWc = sum_leaves dpre x'^T  wgrad32_k with the 224 (slab, position) pairs as "taps"
//   chain rule    G = Wf.Wu,  Wc = scatter(G):   dG = gather(dWc),  dWu = Wf^T dG,  dWf = dG.Wu + dBg.bu,  dbu = Wf^T dBg,  dbf = sum dbc
//
// The fold arithmetic (fp64, same summation order) is build_folded_tail's, so the training-mode reconstruction equals the inference
// tail's on the same parameters.
#pragma once
#include "vq_kernels.h"

// forward step (= weight fragment) index of (slab d, input position p); slab d visits pd in [max(0,d-2), min(3,d+2)]
__device__ __host__ inline int tail_step_base(int d) { return d == 0 ? 0 : d == 1 ? 48 : d == 2 ? 112 : 176; }
__device__ __host__ inline int tail_step_p0(int d) { return (d > 2 ? d - 2 : 0) * 16; }
__device__ __host__ inline int tail_step_count(int d) { return ((d + 2 < 3 ? d + 2 : 3) + 1) * 16 - tail_step_p0(d); }
__device__ inline void tail_step_decode(int step, int& d, int& p)
{
    d = step < 48 ? 0 : step < 112 ? 1 : step < 176 ? 2 : 3;
    p = tail_step_p0(d) + step - tail_step_base(d);
}

// G[dl][sb][t][ci] = sum_oc Wf[oc][dl] * Wu[oc*8+sb][ci][t]   (fp64, oc ascending);  Bg[dl][sb] = sum_oc Wf[oc][dl] * bu[oc*8+sb]
__global__ __launch_bounds__(256) void tail_fold_g_k(const float* __restrict__ Wu, const float* __restrict__ bu, const float* __restrict__ Wf,
                                                     double* __restrict__ G, double* __restrict__ Bg)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < 27 * 8) {
        const int dl = e / 8, sb = e % 8;
        double b = 0.0;
        for (int oc = 0; oc < 32; ++oc) b = fma((double)Wf[oc * 27 + dl], (double)bu[oc * 8 + sb], b);
        Bg[e] = b;
    }
    if (e >= 27 * 8 * 27 * 64) return;
    const int ci = e & 63, t = (e >> 6) % 27, sb = (e / (64 * 27)) & 7, dl = e / (64 * 27 * 8);
    double a = 0.0;
    for (int oc = 0; oc < 32; ++oc) a = fma((double)Wf[oc * 27 + dl], (double)Wu[((size_t)(oc * 8 + sb) * 64 + ci) * 27 + t], a);
    G[e] = a;
}

// one entry of the folded operator: sum over the final-conv taps dl (ascending) that reach output voxel ov of G[dl][sub-voxel][up tap][ci],
// the up tap being the one that connects the coarse voxel under that tap to input position p (at most one per dl)
__device__ __forceinline__ double tail_wc(const double* __restrict__ G, int ov, int p, int ci)
{
    const int od = ov >> 6, oh = (ov >> 3) & 7, ow = ov & 7, pd = p >> 4, ph = (p >> 2) & 3, pw = p & 3;
    double a = 0.0;
    for (int dd = 0; dd < 3; ++dd) {
        const int zd = od + dd - 1, td = pd - (zd >> 1) + 1;
        if (zd < 0 || zd > 7 || td < 0 || td > 2) continue;
        for (int dh = 0; dh < 3; ++dh) {
            const int zh = oh + dh - 1, th = ph - (zh >> 1) + 1;
            if (zh < 0 || zh > 7 || th < 0 || th > 2) continue;
            for (int dw = 0; dw < 3; ++dw) {
                const int zw = ow + dw - 1, tw = pw - (zw >> 1) + 1;
                if (zw < 0 || zw > 7 || tw < 0 || tw > 2) continue;
                const int dl = (dd * 3 + dh) * 3 + dw, sb = (zd & 1) * 4 + (zh & 1) * 2 + (zw & 1), t = (td * 3 + th) * 3 + tw;
                a = a + G[(((size_t)dl * 8 + sb) * 27 + t) * 64 + ci];
            }
        }
    }
    return a;
}

// MFMA fragments of the folded operator from G:
//   fw  forward  [step 224][u 8][mt 4][lane][4]:  rows = the slab's 128 voxels, k = input channel          (conv_mfma32_k OUTMODE 2/3, tail_small_k)
//   tw  transposed [step 224][u 16][mt 2][lane][4]: rows = input channels, k = the slab's 128 voxels       (data gradient)
//   fb  bias in D-fragment order per slab: bc[ov] = bf + sum over valid dl of Bg[dl][sub-voxel]
__global__ __launch_bounds__(256) void tail_fold_frags_k(const double* __restrict__ G, const double* __restrict__ Bg, const float* __restrict__ bf,
                                                         float* __restrict__ fw, float* __restrict__ tw, float* __restrict__ fb)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int64_t NF = (int64_t)224 * 8192;
    if (e < NF) {
        const int step = (int)(e >> 13), r = (int)(e & 8191);
        const int u = r >> 10, mt = (r >> 8) & 3, lane = (r >> 2) & 63, i = r & 3;
        int d, p;
        tail_step_decode(step, d, p);
        fw[e] = (float)tail_wc(G, d * 128 + 32 * mt + (lane & 31), p, 8 * u + 4 * (lane >> 5) + i);
    } else if (e < 2 * NF) {
        const int64_t e2 = e - NF;
        const int step = (int)(e2 >> 13), r = (int)(e2 & 8191);
        const int u = r >> 9, mt = (r >> 8) & 1, lane = (r >> 2) & 63, i = r & 3;
        int d, p;
        tail_step_decode(step, d, p);
        tw[e2] = (float)tail_wc(G, d * 128 + 8 * u + 4 * (lane >> 5) + i, p, 32 * mt + (lane & 31));
    } else if (e < 2 * NF + 512) {
        const int f = (int)(e - 2 * NF), d = f >> 7, mt = (f >> 5) & 3, q = (f >> 4) & 1, r = f & 15;
        const int ov = d * 128 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * q;
        const int od = ov >> 6, oh = (ov >> 3) & 7, ow = ov & 7;
        double b = (double)bf[0];
        for (int dd = 0; dd < 3; ++dd)
            for (int dh = 0; dh < 3; ++dh)
                for (int dw = 0; dw < 3; ++dw) {
                    const int zd = od + dd - 1, zh = oh + dh - 1, zw = ow + dw - 1;
                    if (zd < 0 || zd > 7 || zh < 0 || zh > 7 || zw < 0 || zw > 7) continue;
                    b = b + Bg[((dd * 3 + dh) * 3 + dw) * 8 + (zd & 1) * 4 + (zh & 1) * 2 + (zw & 1)];
                }
        fb[f] = (float)b;
    }
}

// dbc[ov] = sum over leaves of dpre (tile layout [tile][512][32]; padded leaves carry 0)
__global__ __launch_bounds__(256) void tail_dbc_k(const float* __restrict__ dpre, int n_tiles, float* __restrict__ dbc)
{
    __shared__ double red[256];
    const int ov = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_tiles * 32; i += 256) s += (double)dpre[((size_t)(i >> 5) * 512 + ov) * 32 + (i & 31)];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbc[ov] = (float)red[0];
}

// dG[dl][sb][t][ci] = sum over the output voxels ov whose tap dl lands on a fine voxel with sub-voxel sb, of dWc[ov][p = coarse + t - 1][ci];
// dBg[dl][sb] likewise from dbc.  dWc comes from wgrad_reduce_k: dWc[((voxel in slab)*64 + ci)*224 + step(d, p)].
__global__ __launch_bounds__(256) void tail_chain_dg_k(const float* __restrict__ dWc, double* __restrict__ dG)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 27 * 8 * 27 * 64) return;
    const int ci = e & 63, t = (e >> 6) % 27, sb = (e / (64 * 27)) & 7, dl = e / (64 * 27 * 8);
    const int dd = dl / 9, dh = (dl / 3) % 3, dw = dl % 3, td = t / 9, th = (t / 3) % 3, tw = t % 3;
    double a = 0.0;
    for (int c3 = 0; c3 < 64; ++c3) {   // coarse voxel (cd, ch, cw) under the tap: fine voxel z = 2c + sub-voxel bits, output voxel o = z - (tap - 1)
        const int cd = c3 >> 4, ch = (c3 >> 2) & 3, cw = c3 & 3;
        const int od = 2 * cd + (sb >> 2) - (dd - 1), oh = 2 * ch + ((sb >> 1) & 1) - (dh - 1), ow = 2 * cw + (sb & 1) - (dw - 1);
        if (od < 0 || od > 7 || oh < 0 || oh > 7 || ow < 0 || ow > 7) continue;
        const int ov = (od * 8 + oh) * 8 + ow;
        const int pd = cd + td - 1, ph = ch + th - 1, pw = cw + tw - 1;
        if (pd < 0 || pd > 3 || ph < 0 || ph > 3 || pw < 0 || pw > 3) continue;
        const int d = od >> 1, p = (pd * 4 + ph) * 4 + pw;
        a += (double)dWc[((size_t)(ov & 127) * 64 + ci) * 224 + tail_step_base(d) + p - tail_step_p0(d)];
    }
    dG[e] = a;
}
// (dBg has its own tiny kernel: 216 outputs)
__global__ __launch_bounds__(256) void tail_chain_dbg_k(const float* __restrict__ dbc, double* __restrict__ dBg)
{
    const int e = threadIdx.x;
    if (e >= 27 * 8) return;
    const int dl = e / 8, sb = e % 8, dd = dl / 9, dh = (dl / 3) % 3, dw = dl % 3;
    double b = 0.0;
    for (int c3 = 0; c3 < 64; ++c3) {
        const int cd = c3 >> 4, ch = (c3 >> 2) & 3, cw = c3 & 3;
        const int od = 2 * cd + (sb >> 2) - (dd - 1), oh = 2 * ch + ((sb >> 1) & 1) - (dh - 1), ow = 2 * cw + (sb & 1) - (dw - 1);
        if (od < 0 || od > 7 || oh < 0 || oh > 7 || ow < 0 || ow > 7) continue;
        b += (double)dbc[(od * 8 + oh) * 8 + ow];
    }
    dBg[e] = b;
}

// dWu[oc*8+sb][ci][t] = sum_dl Wf[oc][dl] dG[dl][sb][t][ci];  dbu[oc*8+sb] = sum_dl Wf[oc][dl] dBg[dl][sb]
__global__ __launch_bounds__(256) void tail_chain_wu_k(const double* __restrict__ dG, const double* __restrict__ dBg, const float* __restrict__ Wf,
                                                       float* __restrict__ gWu, float* __restrict__ gbu)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < 256) {
        const int oc = e >> 3, sb = e & 7;
        double b = 0.0;
        for (int dl = 0; dl < 27; ++dl) b = fma((double)Wf[oc * 27 + dl], dBg[dl * 8 + sb], b);
        gbu[e] = (float)b;
    }
    if (e >= 256 * 64 * 27) return;
    const int t = e % 27, ci = (e / 27) & 63, co = e / (27 * 64), oc = co >> 3, sb = co & 7;
    double a = 0.0;
    for (int dl = 0; dl < 27; ++dl) a = fma((double)Wf[oc * 27 + dl], dG[(((size_t)dl * 8 + sb) * 27 + t) * 64 + ci], a);
    gWu[e] = (float)a;
}

// dWf[oc][dl] = sum_{sb,t,ci} dG[dl][sb][t][ci] Wu[oc*8+sb][ci][t] + sum_sb dBg[dl][sb] bu[oc*8+sb];  dbf = sum_ov dbc[ov]  (block 864)
__global__ __launch_bounds__(256) void tail_chain_wf_k(const double* __restrict__ dG, const double* __restrict__ dBg, const float* __restrict__ Wu,
                                                       const float* __restrict__ bu, const float* __restrict__ dbc, float* __restrict__ gWf,
                                                       float* __restrict__ gbf)
{
    __shared__ double red[256];
    const int o = blockIdx.x;
    double s = 0.0;
    if (o < 864) {
        const int oc = o / 27, dl = o % 27;
        for (int i = threadIdx.x; i < 8 * 27 * 64; i += 256) {
            const int ci = i & 63, t = (i >> 6) % 27, sb = i / (64 * 27);
            s = fma(dG[(((size_t)dl * 8 + sb) * 27 + t) * 64 + ci], (double)Wu[((size_t)(oc * 8 + sb) * 64 + ci) * 27 + t], s);
        }
        if (threadIdx.x < 8) s = fma(dBg[dl * 8 + threadIdx.x], (double)bu[oc * 8 + threadIdx.x], s);
    } else {
        for (int i = threadIdx.x; i < 512; i += 256) s += (double)dbc[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) (o < 864 ? gWf[o] : gbf[0]) = (float)red[0];
}
