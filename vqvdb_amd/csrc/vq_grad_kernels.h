// vq_grad_kernels.h — kernels of the full training step (stage 2): training-mode forward pieces that the
// inference path folds away (real stem conv input, unfolded up_conv / PixelShuffle3D / final, python/VQVAE_v2.py:253-275),
// the loss of python/training.py:147-155, and the backward of every layer.  fp32 throughout; activations and gradients in
// the leaf-tile ("L4") layout of vq_device.h, single-channel tensors as [tile][512][32].
// Gradients are validated against PyTorch autograd of a plain fp32 restatement (tests/torch_ref.py), which is pinned to the
// imported reference.  Reductions over leaves are done as per-tile (or per-group) partials + an ordered reduce: deterministic.
#pragma once
#include "vq_device.h"

// ------------------------------------------------------------------------------------------
// Weight fragments rebuilt on the device after every optimizer step (same index maps as the host builders in vq_runtime.hip).
// W is the raw PyTorch tensor [OC][IC][KT].  transpose = 0: forward fragments of rows [row0, row0+COUT) x CIN=IC.
// transpose = 1: fragments of the data-gradient conv (stride 1): COUT' = IC, CIN' = sub-range [row0,row0+CIN) of OC, taps flipped.
// ------------------------------------------------------------------------------------------
// ONE launch for all of them: the optimizer step rebuilds ~60 fragment tables and copies ~30 small vectors; as ~90 launches /
// device-to-device copies of 3-4 us each (one kernel per table kind, rounds 1-3) that was 0.15-0.2 ms of a 4.8 ms step.  A job = one
// table (kind, source, destination, integer parameters, element count, first workgroup); a workgroup looks its job up (the list is
// sorted by first workgroup) and writes 256 elements.  Kinds:
//   1  32x32x2 fragments [tap][u][mt][lane][i] = W(co = 32mt + (lane&31), ci = 8u + 4(lane>>5) + i, tap) of rows [row0, row0+COUT) x CIN = IC
//   2  16x16x4 fragments of a 16 -> 16 layer [tap][lane][i] = W(co = lane&15, ci = 4(lane>>4)+i, tap)
//   3  16x16x4 fragments of conv_rows16_k [tap][cb][mt][lane][i] = W(co = 16mt + (lane&15), ci = 16cb + 4(lane>>4) + i, tap)
//   4  the first conv's K = {kw0, kw1, kw2, pad} fragments;  5  a per-cout vector in 32x32 D-fragment order;
//   6  the down conv's transposed fragments [tap][blk][lane][i] = W[16 blk + 4(lane>>4) + i][lane & 15][tap];  0 copy;  7 zeros
struct RefragJob {
    const float* src;
    float* dst;
    int kind;      // 0 copy, 1 refrag32, 2 refrag16, 3 refrag16g, 4 refrag_first, 5 dfrag32, 6 refrag_down_t, 7 zero
    int a, b, c, d, e, f;   // integer parameters of the kind (below)
    float scale;
    int total;     // elements
    int wg0;       // first workgroup of this job
};
__global__ __launch_bounds__(256) void refrag_multi_k(const RefragJob* __restrict__ jobs, int n_jobs)
{
    int lo = 0, hi = n_jobs - 1;   // last job whose first workgroup is <= this one
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].wg0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const RefragJob J = jobs[lo];
    const int t = ((int)blockIdx.x - J.wg0) * 256 + (int)threadIdx.x;
    if (t >= J.total) return;
    const float* __restrict__ W = J.src;
    float v = 0.0f;
    switch (J.kind) {
    case 0: v = W[t]; break;
    case 1: {   // COUT = a, CIN = b, KT = c, IC = d, row0 = e, transpose = f
        const int NU = J.b / 8, NMT = J.a / 32, KT = J.c;
        const int i = t & 3, lane = (t >> 2) & 63;
        int r = t >> 8;
        const int mt = r % NMT;
        r /= NMT;
        const int u = r % NU, tap = r / NU;
        const int co = 32 * mt + (lane & 31), ci = 8 * u + 4 * (lane >> 5) + i;
        v = J.scale * (J.f ? W[((int64_t)(J.e + ci) * J.d + co) * KT + (KT - 1 - tap)] : W[((int64_t)(J.e + co) * J.d + ci) * KT + tap]);
        break;
    }
    case 2: {   // KT = a, transpose = b
        const int KT = J.a, i = t & 3, lane = (t >> 2) & 63, tap = t >> 8;
        const int co = lane & 15, ci = 4 * (lane >> 4) + i;
        v = J.scale * (J.b ? W[(ci * 16 + co) * KT + (KT - 1 - tap)] : W[(co * 16 + ci) * KT + tap]);
        break;
    }
    case 3: {   // COUT = a, CIN = b, KT = c, transpose = d
        const int COUT = J.a, CIN = J.b, KT = J.c, CBN = CIN / 16, MTN = COUT / 16;
        const int i = t & 3, lane = (t >> 2) & 63;
        int r = t >> 8;
        const int mt = r % MTN;
        r /= MTN;
        const int cb = r % CBN, tap = r / CBN;
        const int co = 16 * mt + (lane & 15), ci = 16 * cb + 4 * (lane >> 4) + i;
        v = J.scale * (J.d ? W[((size_t)ci * COUT + co) * KT + (KT - 1 - tap)] : W[((size_t)co * CIN + ci) * KT + tap]);
        break;
    }
    case 4: {   // dst[(kd*3+kh) * 64 + lane] = kw < 3 ? W[(lane & 15) * 27 + (kd*3+kh) * 3 + kw] : 0, kw = lane >> 4
        const int lane = t & 63, t9 = t >> 6, kw = lane >> 4;
        v = kw < 3 ? W[(lane & 15) * 27 + t9 * 3 + kw] : 0.0f;
        break;
    }
    case 5: {   // [(mt*2+q)*16 + r] = v[row0 + 32mt + (r&3) + 8(r>>2) + 4q], row0 = a
        const int mt = t / 32, q = (t / 16) % 2, r = t % 16;
        v = W[J.a + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * q];
        break;
    }
    case 6: {
        const int i = t & 3, lane = (t >> 2) & 63, tb = t >> 8, tap = tb >> 1, blk = tb & 1;
        v = W[((size_t)(16 * blk + 4 * (lane >> 4) + i) * 16 + (lane & 15)) * 64 + tap];
        break;
    }
    default: break;   // 7: zero
    }
    J.dst[t] = v;
}

// quantized latent as the decoder's input (F.embedding + permute, VQVAE_v2.py:127-131): q[tile][pos][32 quads][32][4] = E[idx[leaf][pos]]
__global__ __launch_bounds__(256) void gather_codes_k(const uint8_t* __restrict__ idx, const float* __restrict__ E, float* __restrict__ q,
                                                      int64_t n_leaves, int n_tiles)
{
    const int64_t total = (int64_t)n_tiles * 64 * 32 * 32;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int j = t & 31, quad = (t >> 5) & 31, pos = (t >> 10) & 63;
        const int64_t tile = t >> 16, leaf = tile * 32 + j;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (leaf < n_leaves) v = ((const f32x4*)E)[(int)idx[leaf * 64 + pos] * 32 + quad];
        ((f32x4*)q)[t] = v;
    }
}

// ChannelAttention gates per (tile, channel, leaf) from the channel sums (VQVAE_v2.py:224-227): one wave per tile
template <int C>
__global__ __launch_bounds__(64) void se_gate_k(const float* __restrict__ csum, const float* __restrict__ fc0, const float* __restrict__ fc2,
                                                float* __restrict__ gate)
{
    const int tile = blockIdx.x, lane = threadIdx.x, j = lane & 31;
    float hid[C / 4], g[C];
    se_hidden<C>(csum + (size_t)tile * C * 32 + j, fc0, hid);
    se_gates<C>(hid, fc2, g);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < C; ++c) gate[((size_t)tile * C + c) * 32 + j] = g[c];
    }
}

// PixelShuffle3D(2) (VQVAE_v2.py:172-187) as a copy between the two 128-channel up_conv tensors (L4, 4^3) and a 32-channel L4 tensor
// at 8^3: channel c at voxel (D,H,W) <- channel c*8 + (D&1)*4 + (H&1)*2 + (W&1) at (D>>1,H>>1,W>>1).  One thread per destination
// float4 (4 channels of one leaf and voxel); REVERSE = the backward direction (gradient of the shuffle = inverse copy).
template <bool REVERSE>
__global__ __launch_bounds__(256) void pixshuf_k(float* __restrict__ upA, float* __restrict__ upB, float* __restrict__ ps, int n_tiles)
{
    const int64_t total = (int64_t)n_tiles * 512 * 8 * 32;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int j = t & 31, quad = (t >> 5) & 7, P = (t >> 8) & 511;
        const int64_t tile = t >> 17;
        const int D = P >> 6, H = (P >> 3) & 7, W = P & 7;
        const int sub = (D & 1) * 4 + (H & 1) * 2 + (W & 1), pos = ((D >> 1) * 4 + (H >> 1)) * 4 + (W >> 1);
        float v[4];
        if (REVERSE) {
            const f32x4 s4 = ((const f32x4*)ps)[t];
            v[0] = s4.x, v[1] = s4.y, v[2] = s4.z, v[3] = s4.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = (4 * quad + k) * 8 + sub;
            float* src = (ch < 128 ? upA : upB) + ((((size_t)tile * 64 + pos) * 32 + ((ch & 127) >> 2)) * 32 + j) * 4 + (ch & 3);
            if (REVERSE) *src = v[k];
            else v[k] = *src;
        }
        if (!REVERSE) ((f32x4*)ps)[t] = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// final conv 32 -> 1, k3 p1 @8^3 on the pixel-shuffled tensor (VQVAE_v2.py:269,275) and sigmoid.  VALU: 864 MAC per voxel,
// one float4 load per 4 MACs.  block = 32 leaves x 8 voxels of one tile; grid (tile, 64).
__global__ __launch_bounds__(256) void final_fwd_k(const float* __restrict__ ps, const float* __restrict__ Wf, const float* __restrict__ bf,
                                                   float* __restrict__ pre, float* __restrict__ recon)
{
    __shared__ float w[864];
    for (int i = threadIdx.x; i < 864; i += 256) w[i] = Wf[i];
    __syncthreads();
    const int tile = blockIdx.x, j = threadIdx.x & 31;
    const int P = blockIdx.y * 8 + (threadIdx.x >> 5), D = P >> 6, H = (P >> 3) & 7, Wd = P & 7;
    const f32x4* src = (const f32x4*)ps + (size_t)tile * 512 * 8 * 32 + j;
    float acc = 0.0f;
    for (int kd = 0; kd < 3; ++kd) {
        const int d = D + kd - 1;
        if (d < 0 || d > 7) continue;
        for (int kh = 0; kh < 3; ++kh) {
            const int h = H + kh - 1;
            if (h < 0 || h > 7) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int x = Wd + kw - 1;
                if (x < 0 || x > 7) continue;
                const int tap = (kd * 3 + kh) * 3 + kw, Q = (d * 8 + h) * 8 + x;
#pragma unroll
                for (int quad = 0; quad < 8; ++quad) {
                    const f32x4 v = src[((size_t)Q * 8 + quad) * 32];
                    acc = __builtin_fmaf(w[(4 * quad + 0) * 27 + tap], v.x, acc);
                    acc = __builtin_fmaf(w[(4 * quad + 1) * 27 + tap], v.y, acc);
                    acc = __builtin_fmaf(w[(4 * quad + 2) * 27 + tap], v.z, acc);
                    acc = __builtin_fmaf(w[(4 * quad + 3) * 27 + tap], v.w, acc);
                }
            }
        }
    }
    acc = acc + bf[0];
    pre[((size_t)tile * 512 + P) * 32 + j] = acc;
    recon[((size_t)tile * 512 + P) * 32 + j] = vq_sigmoid(acc);
}

// d(loss)/d(pre) for loss = 0.8*mse + 0.2*l1 over N voxels (python/training.py:147-155): c_mse = 1.6/N, c_l1 = 0.2/N
// dpre4 (optional): the same gradient in the L4 layout with the 128 voxels of an output slab as "channels" ([tile][4 slabs][32 quads][32][4]):
// the operand layout of the folded tail's data and weight gradients (vq_train_tail.h)
__global__ __launch_bounds__(256) void loss_grad_k(const float* __restrict__ x, const float* __restrict__ recon, float* __restrict__ dpre, float c_mse,
                                                   float c_l1, int64_t n_leaves, int n_tiles, float* __restrict__ dpre4 = nullptr)
{
    const int64_t total = (int64_t)n_tiles * 512 * 32;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t leaf = (t >> 14) * 32 + (t & 31);
        float g = 0.0f;
        if (leaf < n_leaves) {
            const float r = recon[t], d = r - x[t];
            g = (c_mse * d + (d > 0.0f ? c_l1 : (d < 0.0f ? -c_l1 : 0.0f))) * (r * (1.0f - r));
        }
        dpre[t] = g;
        if (dpre4) {
            const int v = (int)((t >> 5) & 511);
            dpre4[((((t >> 14) * 128) + (v >> 2)) * 32 + (t & 31)) * 4 + (v & 3)] = g;
        }
    }
}
// sums of (recon-x)^2 and |recon-x| over the real leaves in the tile layout: partials per block, ordered reduce by the caller
__global__ __launch_bounds__(256) void loss_sums_k(const float* __restrict__ x, const float* __restrict__ recon, int64_t n_leaves, int n_tiles,
                                                   double* __restrict__ part /*[gridDim.x][2]*/)
{
    __shared__ double s2[256], s1[256];
    double a2 = 0.0, a1 = 0.0;
    const int64_t total = (int64_t)n_tiles * 512 * 32;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t leaf = (t >> 14) * 32 + (t & 31);
        if (leaf < n_leaves) {
            const double d = (double)recon[t] - (double)x[t];
            a2 = fma(d, d, a2);
            a1 += d < 0.0 ? -d : d;
        }
    }
    s2[threadIdx.x] = a2;
    s1[threadIdx.x] = a1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s2[threadIdx.x] += s2[threadIdx.x + w];
            s1[threadIdx.x] += s1[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s2[0];
        part[2 * blockIdx.x + 1] = s1[0];
    }
}

// ------------------------------------------------------------------------------------------
// final conv backward (32 -> 1, k3 p1 @8^3 over the pixel-shuffled tensor)
// ------------------------------------------------------------------------------------------
// data gradient wrt the pixel-shuffled tensor (L4, 32 channels @8^3)
__global__ __launch_bounds__(256) void final_bwd_data_k(const float* __restrict__ dpre, const float* __restrict__ Wf, float* __restrict__ dps)
{
    __shared__ float w[864];
    for (int i = threadIdx.x; i < 864; i += 256) w[i] = Wf[i];
    __syncthreads();
    const int tile = blockIdx.x, j = threadIdx.x & 31;
    const int P = blockIdx.y * 8 + (threadIdx.x >> 5), D = P >> 6, H = (P >> 3) & 7, Wd = P & 7;
    float g[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        // ps[c][P] fed pre[P - off(t)] with weight W[c][t]
        const int d = D - (t / 9 - 1), h = H - ((t / 3) % 3 - 1), x = Wd - (t % 3 - 1);
        g[t] = (d >= 0 && d < 8 && h >= 0 && h < 8 && x >= 0 && x < 8) ? dpre[((size_t)tile * 512 + (d * 8 + h) * 8 + x) * 32 + j] : 0.0f;
    }
    f32x4* dst = (f32x4*)dps + ((size_t)tile * 512 + P) * 8 * 32 + j;
    for (int quad = 0; quad < 8; ++quad) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 27; ++t) s[k] = __builtin_fmaf(w[(4 * quad + k) * 27 + t], g[t], s[k]);
        dst[(size_t)quad * 32] = (f32x4){s[0], s[1], s[2], s[3]};
    }
}
// weight + bias gradient partials: part[(tile*gridDim.y + y)][865] ([c*27+tap], last = bias).  8 waves, wave w owns channel quad w;
// gridDim.y cuts each lane half's 256 voxels into ranges.
__global__ __launch_bounds__(512) void final_wgrad_k(const float* __restrict__ dpre, const float* __restrict__ ps, float* __restrict__ part)
{
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int np = 256 / gridDim.y;
    part += (size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 865;
    const f32x4* src = (const f32x4*)ps + (size_t)tile * 512 * 8 * 32 + wave * 32 + j;
    float acc[27][4];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] = 0.0f;
    float bsum = 0.0f;
    // input-voxel-major: one float4 of the pixel-shuffled tensor per voxel Q, the 27 output gradients that used it as scalars
    for (int Q = 256 * h + np * blockIdx.y; Q < 256 * h + np * (blockIdx.y + 1); ++Q) {
        const f32x4 v = src[(size_t)Q * 8 * 32];
        const int D = Q >> 6, H = (Q >> 3) & 7, Wd = Q & 7;
        const float* dp0 = dpre + (size_t)tile * 512 * 32 + j;
        bsum += dp0[(size_t)Q * 32];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int d = D - (t / 9 - 1), hh = H - ((t / 3) % 3 - 1), x = Wd - (t % 3 - 1);   // output voxel P = Q - off(t)
            if (d < 0 || d > 7 || hh < 0 || hh > 7 || x < 0 || x > 7) continue;
            const float dp = dp0[(size_t)((d * 8 + hh) * 8 + x) * 32];
            acc[t][0] = __builtin_fmaf(dp, v.x, acc[t][0]);
            acc[t][1] = __builtin_fmaf(dp, v.y, acc[t][1]);
            acc[t][2] = __builtin_fmaf(dp, v.z, acc[t][2]);
            acc[t][3] = __builtin_fmaf(dp, v.w, acc[t][3]);
        }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = acc[t][k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) part[(4 * wave + k) * 27 + t] = v;
        }
    if (wave == 0) {
        for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o, 64);
        if (lane == 0) part[864] = bsum;
    }
}
// dst[i] = scale * sum over the parts of part[p][i].  64 outputs per workgroup (grid = ceil(n / 64)); wave w adds the parts w, w+4, ...
// ascending, the four wave sums are added in wave order (wgrad_reduce_k's scheme).
__global__ __launch_bounds__(256) void parts_reduce_k(const float* __restrict__ part, int n_parts, int n, float* __restrict__ dst, float scale)
{
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + o;
    float s = 0.0f;
    if (i < n) {
#pragma unroll 4
        for (int p = w; p < n_parts; p += 4) s += part[(size_t)p * n + i];
    }
    red[w][o] = s;
    __syncthreads();
    if (w == 0 && i < n) dst[i] = scale * (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
}
// dst[c] = scale * sum over tiles and leaves of part[tile][c][32]  (bias / GroupNorm-affine gradients from per-(tile,channel,leaf) sums).
// One 256-thread block per channel: thread t walks tiles t>>5, t>>5 + 8, ... for leaf t&31, then a fixed LDS tree (deterministic).
__global__ __launch_bounds__(256) void chan_reduce_k(const float* __restrict__ part, int n_tiles, int C, float* __restrict__ dst, float scale)
{
    __shared__ float red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    float s = 0.0f;
    for (int tile = t >> 5; tile < n_tiles; tile += 8) s += part[((size_t)tile * C + c) * 32 + (t & 31)];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) dst[c] = scale * red[0];
}

// ------------------------------------------------------------------------------------------
// Round 6: the small reductions of a training step that nothing on the data-gradient chain waits for — per-channel sums of every
// layer's dY (bias gradients), the channel reductions of the GroupNorm-affine partials, the attention-weight partials — as TWO launches
// at the end of the chain instead of 40 between its kernels (vq_train_full.inc, Bwd::join).  A job list travels as a kernel argument;
// a workgroup looks its job up by block index and runs exactly the arithmetic of csum_seq_k / chan_reduce_k / parts_reduce_k.
// ------------------------------------------------------------------------------------------
struct CsumJob {
    const float* x;     // L4 [tile][NP][C/4][32][4]
    float* part;        // [tile][C][32]
    int C, NP, first_block;
};
constexpr int RED_MAX_CSUM = 16, RED_MAX_JOBS = 48;
struct CsumJobs {
    CsumJob j[RED_MAX_CSUM];
    int n;
};
// grid = sum over the jobs of n_tiles * C / 16 workgroups of 128 threads (a tile's 64 C / 8 threads = C / 16 workgroups)
__global__ __launch_bounds__(128) void csum_multi_k(CsumJobs J)
{
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.j[k + 1].first_block) ++k;   // (uniform)
    const CsumJob job = J.j[k];
    const int local = (int)blockIdx.x - job.first_block, upt = job.C / 16;
    const int tile = local / upt, tid = (local % upt) * 128 + (int)threadIdx.x;
    if (job.C == 64 && job.NP == 64) csum_seq_thread<64, 64>(job.x, job.part, tile, tid);
    else if (job.C == 128 && job.NP == 64) csum_seq_thread<128, 64>(job.x, job.part, tile, tid);
    else if (job.C == 32 && job.NP == 64) csum_seq_thread<32, 64>(job.x, job.part, tile, tid);
    else if (job.C == 16 && job.NP == 512) csum_seq_thread<16, 512>(job.x, job.part, tile, tid);
}
struct RedJob {
    const float* part;
    float* dst;
    int kind;           // 0: chan_reduce_k (part[tile][C][32], one workgroup per channel); 1: parts_reduce_k (part[p][n], 64 outputs per workgroup)
    int n_parts, n, first_block;
    float scale;
};
struct RedJobs {
    RedJob j[RED_MAX_JOBS];
    int n;
};
__global__ __launch_bounds__(256) void reduce_multi_k(RedJobs J)
{
    __shared__ float red[256];
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.j[k + 1].first_block) ++k;   // (uniform)
    const RedJob job = J.j[k];
    const int b = (int)blockIdx.x - job.first_block, t = threadIdx.x;
    if (job.kind == 0) {   // chan_reduce_k: thread t walks tiles t>>5, t>>5 + 8, ... for leaf t&31, then a fixed LDS tree
        const int C = job.n, c = b;
        float s = 0.0f;
        for (int tile = t >> 5; tile < job.n_parts; tile += 8) s += job.part[((size_t)tile * C + c) * 32 + (t & 31)];
        red[t] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (t < w) red[t] += red[t + w];
            __syncthreads();
        }
        if (t == 0) job.dst[c] = job.scale * red[0];
    } else {               // parts_reduce_k: wave w adds the parts w, w+4, ... ascending, the four wave sums in wave order
        const int o = t & 63, w = t >> 6, i = b * 64 + o;
        float s = 0.0f;
        if (i < job.n) {
#pragma unroll 4
            for (int p = w; p < job.n_parts; p += 4) s += job.part[(size_t)p * job.n + i];
        }
        red[w * 64 + o] = s;
        __syncthreads();
        if (w == 0 && i < job.n) job.dst[i] = job.scale * (((red[o] + red[64 + o]) + red[128 + o]) + red[192 + o]);
    }
}

// ------------------------------------------------------------------------------------------
// Generic weight gradient of a leaf-tile convolution: dW[co][ci][tap] = sum over leaves, valid (ip, po) pairs of the tap of
// dY[po][co][leaf] * X'[ip][ci][leaf], X' = X, relu(GroupNorm(X)) or gate*X exactly as the forward kernel formed it on load.
// The leaves are the MFMA K axis: the dY and X' blocks of one position pair are transposed through LDS ([channel][leaf], row
// stride 33 floats -> conflict-free) and each wave accumulates one 32 x 32 (co, ci) block over ALL steps of its workgroup.
// grid (taps, tile groups); wsteps = (ip, po) pairs sorted by tap, tap_start[t] their offsets.  Output: per-group partials
// part[((grp*KT + tap)*COUT + co)*CIN + ci]; wgrad_reduce_k adds the groups in order and writes PyTorch's [OC][IC][KT] layout.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy;      // L4 [tile][NPO][COUT/4][32][4]
    const float* x;       // L4 [tile][NPI][CIN/4][32][4]
    const float* mean;    // INMODE 1: [tile][GIN][32]
    const float* rstd;
    const float* gamma;   // [CIN]
    const float* beta;
    const float* gate;    // INMODE 2: [tile][CIN][32]
    const int2* wsteps;
    const int* tap_start;
    float* part;
    int n_tiles, tiles_per_group, KT;
};

#ifndef WGRAD_ABL   // timing-only switches of tools/ablate/wgrad_ablate.hip: 1 one global fetch only, 2 no MFMAs, 4 no LDS staging / barriers
#define WGRAD_ABL 0
#endif
template <int CIN, int COUT, int NPI, int NPO, int INMODE, int GIN>
__global__ __launch_bounds__(((COUT + 31) / 32) * ((CIN + 31) / 32) * 64) void wgrad32_k(WgradArgs A)
{
    // CIN == 16 (the 8^3 layers): one wave, 16x16x4 MFMAs (K = 4 leaves), COUT/16 accumulators; otherwise one wave per
    // 32 x 32 (co, ci) block on the 32x32x2 MFMA (K = 2 leaves)
    constexpr bool SMALL = CIN == 16;
    constexpr int CB = SMALL ? 1 : (COUT + 31) / 32, IB = SMALL ? 1 : (CIN + 31) / 32, NT = CB * IB * 64;
    constexpr int ROWS_DY = SMALL ? COUT : CB * 32, ROWS_X = SMALL ? 16 : IB * 32;
    // blocks in LDS as [leaf][channel] (round 2; [channel][leaf] before): a thread's float4 = 4 channels of one leaf goes in with ONE
    // 16-byte write instead of four scalar ones; the MFMA operands A[row = channel][k = leaf] are read row-wise over the channels
    // (consecutive banks).  Row stride +4 floats: the 32 leaves of a write land on 8 bank groups instead of one.
    // Two buffers (round 3): the blocks of pair i+1 are written while the MFMAs of pair i read the other buffer — ONE barrier per pair
    // (two before, with the staging and the MFMA phase of a workgroup strictly alternating: 20 % of the launch by ablation,
    // tools/ablate/wgrad_ablate.hip).
    constexpr int SDY = ROWS_DY + 4, SX = ROWS_X + 4;
    __shared__ __attribute__((aligned(16))) float sdy[2][32][SDY];
    __shared__ __attribute__((aligned(16))) float sx[2][32][SX];
    for (int i = threadIdx.x; i < 2 * 32 * SDY; i += NT) (&sdy[0][0][0])[i] = 0.0f;
    for (int i = threadIdx.x; i < 2 * 32 * SX; i += NT) (&sx[0][0][0])[i] = 0.0f;
    // Workgroup -> (tap, chunk, tile group), XCD-aware: the dispatcher is observed to place workgroup b on XCD b % 8 (a speed assumption only), so
    // the linear id is remapped to give every XCD a contiguous range of (group, chunk, tap) with the tap fastest: all taps of a tile group run
    // on one XCD at the same time and re-read the group's dY / X blocks from that XCD's L2 instead of 27 x from the fabric.
    int tap, grp, chunk;
    {
        const int n = gridDim.x * gridDim.y * gridDim.z, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int xcd = lin & 7, q = n >> 3, r = n & 7;
        const int virt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        tap = virt % (int)gridDim.x;
        chunk = (virt / (int)gridDim.x) % (int)gridDim.z;
        grp = virt / (int)(gridDim.x * gridDim.z);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cb = wave / IB, ib = wave % IB;
    // the tap's (ip, po) pairs are cut into gridDim.z chunks (more workgroups for the small layers)
    const int t_s0 = A.tap_start[tap], t_s1 = A.tap_start[tap + 1];
    const int s0 = t_s0 + (int)((int64_t)(t_s1 - t_s0) * chunk / gridDim.z), s1 = t_s0 + (int)((int64_t)(t_s1 - t_s0) * (chunk + 1) / gridDim.z);
    const int t0 = grp * A.tiles_per_group, t1 = min(A.n_tiles, t0 + A.tiles_per_group);
    f32x16 acc;
    f32x4 acc16[SMALL ? COUT / 16 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int b = 0; b < (SMALL ? COUT / 16 : 1); ++b) acc16[b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int CPG = GIN > 0 ? CIN / GIN : 1;
    // The dY / X blocks of one position pair of one tile are requested TWO pairs ahead into registers (in flight during a whole MFMA phase),
    // written to LDS one pair ahead.  Tile / pair counters advance incrementally (no division by the run-time pair count in the loop).
    constexpr int ND = ((COUT / 4) * 32 + NT - 1) / NT, NX = ((CIN / 4) * 32 + NT - 1) / NT;
    const int npairs = s1 - s0;
    const int total = npairs > 0 && t1 > t0 ? npairs * (t1 - t0) : 0;
    f32x4 rdy[ND], rx[NX];
    int ft = t0, fg = 0;   // next pair to request
    bool fetched = false;
    // (buffer addressing, see buf_ld16: the block of a position is one contiguous run, thread i takes float4 i)
    auto fetch_pair = [&]() {
        if (ft >= t1 || npairs <= 0) return;   // (uniform)
        if ((WGRAD_ABL & 1) && fetched) {
            if (++fg == npairs) fg = 0, ++ft;
            return;
        }
        fetched = true;
        const vq_buf dyb = buf_of((const f32x4*)A.dy + (size_t)ft * NPO * (COUT / 4) * 32);
        const vq_buf xb = buf_of((const f32x4*)A.x + (size_t)ft * NPI * (CIN / 4) * 32);
        const int2 e = A.wsteps[s0 + fg];   // x = input position, y = output position
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int i = threadIdx.x + d * NT;
            if (i < (COUT / 4) * 32) rdy[d] = buf_ld16(dyb, (unsigned)i * 16u, (unsigned)e.y * (COUT / 4) * 512u);
        }
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            const int i = threadIdx.x + d * NT;
            if (i < (CIN / 4) * 32) rx[d] = buf_ld16(xb, (unsigned)i * 16u, (unsigned)e.x * (CIN / 4) * 512u);
        }
        if (++fg == npairs) fg = 0, ++ft;
    };
    // the input transform of this thread's channels (GroupNorm scale / shift or attention gate): per (tile, channel, leaf), so it
    // changes only when the tile does — it used to be re-loaded in every step, 8-16 scalar loads per 16 MFMAs
    float tia[NX][4], tib[NX][4];
    auto load_transform = [&](int tile) {
#pragma unroll
        for (int dd = 0; dd < NX; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (CIN / 4) * 32) {
                const int quad = i >> 5, leaf = i & 31;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int ch = 4 * quad + kk;
                    if (INMODE == 1) {
                        const int g = ch / CPG;
                        tia[dd][kk] = A.rstd[((size_t)tile * GIN + g) * 32 + leaf] * A.gamma[ch];
                        tib[dd][kk] = __builtin_fmaf(-A.mean[((size_t)tile * GIN + g) * 32 + leaf], tia[dd][kk], A.beta[ch]);
                    } else if (INMODE == 2) {
                        tia[dd][kk] = A.gate[((size_t)tile * CIN + ch) * 32 + leaf];
                    }
                }
            }
        }
    };
    // the requested pair (in registers) -> LDS buffer `buf` ([leaf][channel], GroupNorm+ReLU / gate applied to X on the way)
    int wt = t0, wp = 0;   // tile / pair being written
    auto stage = [&](int buf) {
        if (INMODE != 0 && wp == 0) load_transform(wt);   // (uniform)
        if (++wp == npairs) wp = 0, ++wt;
        if (WGRAD_ABL & 4) return;
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (COUT / 4) * 32) *(f32x4*)&sdy[buf][i & 31][4 * (i >> 5)] = rdy[dd];
        }
#pragma unroll
        for (int dd = 0; dd < NX; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (CIN / 4) * 32) {
                const int quad = i >> 5, leaf = i & 31;
                const f32x4 v = rx[dd];
                float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (INMODE == 1) o[kk] = fmaxf(__builtin_fmaf(o[kk], tia[dd][kk], tib[dd][kk]), 0.0f);
                    else if (INMODE == 2) o[kk] = o[kk] * tia[dd][kk];
                }
                *(f32x4*)&sx[buf][leaf][4 * quad] = (f32x4){o[0], o[1], o[2], o[3]};
            }
        }
    };
    fetch_pair();
    __syncthreads();   // the zero fill
    if (total > 0) {
        stage(0);
        fetch_pair();
    }
    __syncthreads();
    for (int i = 0; i < total; ++i) {
        const int buf = i & 1;
        if (i + 1 < total) {   // (uniform)
            stage(buf ^ 1);    // every wave has left the MFMAs of pair i-1, which read that buffer (barrier below)
            fetch_pair();      // pair i+2: in flight during this pair's MFMAs and the next one's
        }
        if (!(WGRAD_ABL & 2)) {
            if (SMALL) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float bx = sx[buf][4 * m + (lane >> 4)][lane & 15];
#pragma unroll
                    for (int b = 0; b < COUT / 16; ++b) acc16[b] = mfma16(sdy[buf][4 * m + (lane >> 4)][16 * b + (lane & 15)], bx, acc16[b]);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    acc = mfma32(sdy[buf][2 * m + (lane >> 5)][32 * cb + (lane & 31)], sx[buf][2 * m + (lane >> 5)][32 * ib + (lane & 31)], acc);
            }
        }
        if (!(WGRAD_ABL & 4)) __syncthreads();
    }
    float* dst = A.part + (((size_t)grp * gridDim.z + chunk) * A.KT + tap) * COUT * CIN;
    if (SMALL) {
#pragma unroll
        for (int b = 0; b < COUT / 16; ++b) {
            const float v[4] = {acc16[b].x, acc16[b].y, acc16[b].z, acc16[b].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(16 * b + 4 * (lane >> 4) + r) * CIN + (lane & 15)] = v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = 32 * ib + (lane & 31);
            if (co < COUT && ci < CIN) dst[(size_t)co * CIN + ci] = acc[r];
        }
    }
}
// Weight gradient of the 16 -> 16, k3 p1 layers at 8^3 with row reuse: one wave per (kd, kh, tile group, row chunk).  For an output row
// (od, oh) the 8 dY blocks and the 8 X' blocks of the input row (od+kd-1, oh+kh-1) are staged once in LDS and serve the three kw taps
// (22 block products from 16 block loads instead of 44: the tap-major kernel is bound by re-reading blocks from L2).
// part layout as wgrad32_k: part[((grp*n_chunks + chunk)*27 + tap)*256 + co*16 + ci].
__global__ __launch_bounds__(192) void wgrad16_rows_k(WgradArgs A, int n_chunks)
{
    // blocks as [position][leaf][16 channels]: a thread's float4 (4 channels of one leaf) is ONE 16-byte write, and the 64 operand reads of an
    // MFMA (lane = (channel, leaf 4m + k)) cover 64 consecutive floats
    __shared__ __attribute__((aligned(16))) float sdy[8][32][16];
    __shared__ __attribute__((aligned(16))) float sx[8][32][16];
    const int kdkh = blockIdx.x, kd = kdkh / 3, kh = kdkh % 3, grp = blockIdx.y, chunk = blockIdx.z;
    const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6;   // three waves: one kw tap each, staging shared
    const int t0 = grp * A.tiles_per_group, t1 = min(A.n_tiles, t0 + A.tiles_per_group);
    const int r0 = chunk * 64 / n_chunks, r1 = (chunk + 1) * 64 / n_chunks;   // output rows (od*8 + oh) of this chunk
    // staging: element i = tid + 192 k (k = 0..5, i < 1024) is float4 (position i >> 7, quad (i >> 5) & 3, leaf i & 31): the leaf of a thread
    // never changes and its quad alternates between q0 and q0 ^ 2, so the GroupNorm scale / shift of its 8 channels are loaded once per tile
    // (they used to be re-loaded for every element: 16 scalar loads per float4)
    const int leaf = threadIdx.x & 31, q0 = (threadIdx.x >> 5) & 3;
    float tia[2][4], tib[2][4];
    f32x4 rdy[6], rx[6];
    auto row_valid = [&](int row) { const int id = (row >> 3) + kd - 1, ih = (row & 7) + kh - 1; return id >= 0 && id <= 7 && ih >= 0 && ih <= 7; };
    auto fetch = [&](int tile, int row) {   // the 8 dY blocks of output row `row` and the 8 X blocks of its input row -> registers
        const int irow = ((row >> 3) + kd - 1) * 8 + (row & 7) + kh - 1;
        const vq_buf dyb = buf_of((const f32x4*)A.dy + ((size_t)tile * 512 + row * 8) * 128);
        const vq_buf xb = buf_of((const f32x4*)A.x + ((size_t)tile * 512 + irow * 8) * 128);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = threadIdx.x + 192 * k;
            if (i < 1024) rdy[k] = buf_ld16(dyb, (unsigned)i * 16u, 0u), rx[k] = buf_ld16(xb, (unsigned)i * 16u, 0u);
        }
    };
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tile = t0; tile < t1; ++tile) {
        int row = r0;
        while (row < r1 && !row_valid(row)) ++row;
        if (row >= r1) break;   // (the valid rows do not depend on the tile)
        fetch(tile, row);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // X' = relu(GroupNorm(8,16)(X)) as the forward conv formed it
                const int ch = 4 * (q0 ^ (2 * h)) + k, gidx = ch >> 1;
                tia[h][k] = A.rstd[((size_t)tile * 8 + gidx) * 32 + leaf] * A.gamma[ch];
                tib[h][k] = __builtin_fmaf(-A.mean[((size_t)tile * 8 + gidx) * 32 + leaf], tia[h][k], A.beta[ch]);
            }
        while (row < r1) {
            __syncthreads();   // the previous row's MFMAs have read the blocks
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = threadIdx.x + 192 * k;
                if (i < 1024) {
                    const int w = i >> 7, quad = (i >> 5) & 3, h = k & 1;   // quad = q0 ^ (2 h): 192 k / 32 = 6 k = 2 k (mod 4)
                    *(f32x4*)&sdy[w][leaf][4 * quad] = rdy[k];
                    const f32x4 v = rx[k];
                    *(f32x4*)&sx[w][leaf][4 * quad] = (f32x4){fmaxf(__builtin_fmaf(v.x, tia[h][0], tib[h][0]), 0.0f), fmaxf(__builtin_fmaf(v.y, tia[h][1], tib[h][1]), 0.0f),
                                                              fmaxf(__builtin_fmaf(v.z, tia[h][2], tib[h][2]), 0.0f), fmaxf(__builtin_fmaf(v.w, tia[h][3], tib[h][3]), 0.0f)};
                }
            }
            int nrow = row + 1;
            while (nrow < r1 && !row_valid(nrow)) ++nrow;
            if (nrow < r1) fetch(tile, nrow);   // in flight during this row's MFMAs
            __syncthreads();
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                const int iw = ow + kw - 1;
                if (iw < 0 || iw > 7) continue;   // wave-uniform
#pragma unroll
                for (int m = 0; m < 8; ++m) acc = mfma16(sdy[ow][4 * m + (lane >> 4)][lane & 15], sx[iw][4 * m + (lane >> 4)][lane & 15], acc);
            }
            row = nrow;
        }
    }
    float* dst = A.part + (((size_t)grp * n_chunks + chunk) * 27 + kdkh * 3 + kw) * 256;
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = v[r];
}
// Weight gradient of the 16 -> 16, k3 p1 layers at 8^3, a plane at a time (round 3).  wgrad16_rows_k above re-reads every dY row and every X'
// row once per (kd, kh) class: 9 x 0.77 x 134 MB = 930 MB at 2048 leaves, which is what its 0.17 ms are.  Here a workgroup keeps kd fixed and
// walks the output rows oh = 0..7 of a plane (tile, od): the dY row goes to LDS, the input rows ih = oh-1, oh, oh+1 of plane id = od+kd-1
// live in a ring of three row slots (each row is staged once per plane and serves three output rows; row oh+1 replaces row oh-2), so all nine (kh, kw) taps are fed
// from 32 KB per output row instead of 96.  Eight waves = the eight output positions ow of the row: wave ow reads its dY block's operands
// once and multiplies them with up to nine X' blocks (ih, iw = ow+kw-1); its nine partial accumulators are added over the waves in wave
// order through LDS at the end of the workgroup.  grid (ceil(8 n_tiles / P), 3 = kd), P planes per workgroup; the rows of the next step
// (dY row oh+1, X' row oh+2; three rows when a new plane starts) are in flight in registers during the MFMAs.
// part[(kd * gridDim.x + block) * 9 + kh * 3 + kw][co * 16 + ci]; wgrad16_planes_reduce_k adds the workgroups of a kd in order.
__global__ __launch_bounds__(512) void wgrad16_planes_k(WgradArgs A, int P)
{
    __shared__ __attribute__((aligned(16))) float sdy[8][32][16];
    __shared__ __attribute__((aligned(16))) float sxr[3][8][32][16];
    const int kd = blockIdx.y, n_od = kd == 1 ? 8 : 7, od0 = kd == 0 ? 1 : 0;
    const int n_planes = n_od * A.n_tiles, p0 = blockIdx.x * P, p1 = min(p0 + P, n_planes);
    float* dst = A.part + (size_t)(kd * gridDim.x + blockIdx.x) * 9 * 256;
    if (p0 >= n_planes) return;   // (never read by the reduction)
    const int lane = threadIdx.x & 63, ow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lc = lane & 15, lk = lane >> 4;
    // staging: thread -> float4 (position w0 and w0 + 4, quad, leaf) of a row: quad and leaf never change, so the GroupNorm scale / shift of
    // its four channels are loaded once per tile
    const int leaf = threadIdx.x & 31, quad = (threadIdx.x >> 5) & 3, w0 = threadIdx.x >> 7;
    float tia[4], tib[4];
    int t_tile = -1;
    f32x4 rdy[2], rxa[2], rxb[2];
    const int total = (p1 - p0) * 8;
    auto where = [&](int s, int& tile, int& od, int& oh) {
        const int p = p0 + (s >> 3);
        tile = p / n_od, od = od0 + p % n_od, oh = s & 7;
    };
    auto fetch = [&](int s) {
        if (s >= total) return;   // (uniform)
        int tile, od, oh;
        where(s, tile, od, oh);
        const int id = od + kd - 1;
        const vq_buf dyb = buf_of((const f32x4*)A.dy + ((size_t)tile * 512 + (od * 8 + oh) * 8) * 128);
        const unsigned lo = (unsigned)(w0 * 128 + quad * 32 + leaf) * 16u;
        rdy[0] = buf_ld16(dyb, lo, 0u), rdy[1] = buf_ld16(dyb, lo, 4u * 128u * 16u);
        if (oh < 7) {   // (uniform) input row oh + 1
            const vq_buf xb = buf_of((const f32x4*)A.x + ((size_t)tile * 512 + (id * 8 + oh + 1) * 8) * 128);
            rxa[0] = buf_ld16(xb, lo, 0u), rxa[1] = buf_ld16(xb, lo, 4u * 128u * 16u);
        }
        if (oh == 0) {  // (uniform) a new plane: its input row 0 as well
            const vq_buf xb = buf_of((const f32x4*)A.x + ((size_t)tile * 512 + (id * 8) * 8) * 128);
            rxb[0] = buf_ld16(xb, lo, 0u), rxb[1] = buf_ld16(xb, lo, 4u * 128u * 16u);
        }
    };
    auto xform = [&](f32x4 v) {   // X' = relu(GroupNorm(8,16)(X)) as the forward conv formed it
        return (f32x4){fmaxf(__builtin_fmaf(v.x, tia[0], tib[0]), 0.0f), fmaxf(__builtin_fmaf(v.y, tia[1], tib[1]), 0.0f),
                       fmaxf(__builtin_fmaf(v.z, tia[2], tib[2]), 0.0f), fmaxf(__builtin_fmaf(v.w, tia[3], tib[3]), 0.0f)};
    };
    auto stage = [&](int s) {
        int tile, od, oh;
        where(s, tile, od, oh);
        if (tile != t_tile) {   // (uniform)
            t_tile = tile;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = 4 * quad + k, g = ch >> 1;
                tia[k] = A.rstd[((size_t)tile * 8 + g) * 32 + leaf] * A.gamma[ch];
                tib[k] = __builtin_fmaf(-A.mean[((size_t)tile * 8 + g) * 32 + leaf], tia[k], A.beta[ch]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int w = w0 + 4 * h;
            *(f32x4*)&sdy[w][leaf][4 * quad] = rdy[h];
            if (oh < 7) *(f32x4*)&sxr[(oh + 1) % 3][w][leaf][4 * quad] = xform(rxa[h]);
            if (oh == 0) *(f32x4*)&sxr[0][w][leaf][4 * quad] = xform(rxb[h]);
        }
    };
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    fetch(0);
    for (int s = 0; s < total; ++s) {
        __syncthreads();   // the previous step's MFMAs have read the row buffers
        stage(s);
        fetch(s + 1);      // in flight during this step's MFMAs
        __syncthreads();
        const int oh = s & 7;
        float a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = sdy[ow][4 * m + lk][lc];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh + kh - 1;
            if (ih < 0 || ih > 7) continue;   // (uniform)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow + kw - 1;
                if (iw < 0 || iw > 7) continue;   // (wave-uniform)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[kh * 3 + kw] = mfma16(a[m], sxr[ih % 3][iw][4 * m + lk][lc], acc[kh * 3 + kw]);
            }
        }
    }
    // the nine taps of the eight waves, added in wave order (scratch = the dY row buffer and the first ring slot: 9 x 256 floats)
    float* red = &sdy[0][0][0];
    static_assert(sizeof(sdy) >= 9 * 256 * sizeof(float), "reduction scratch");
    for (int wv = 0; wv < 8; ++wv) {
        __syncthreads();
        if (ow == wv) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* q = red + t * 256 + (4 * lk + r) * 16 + lc;
                    *q = wv == 0 ? acc[t][r] : *q + acc[t][r];
                }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 256; i += 512) dst[i] = red[i];
}
// dW[co][ci][tap = (kd*3+kh)*3+kw] = scale * sum over the workgroups of kd (ascending) of part[(kd * gx + b) * 9 + kh*3+kw][co*16+ci];
// 64 outputs per workgroup, four waves each adding every fourth partial, wave sums in wave order (wgrad_reduce_k's scheme)
__global__ __launch_bounds__(256) void wgrad16_planes_reduce_k(const float* __restrict__ part, int gx, int n_tiles, int P, float* __restrict__ dW, float scale)
{
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6, t = blockIdx.x * 64 + o;   // t < 27 * 256
    const int tap = t >> 8, e = t & 255, kd = tap / 9, nb = ((kd == 1 ? 8 : 7) * n_tiles + P - 1) / P;
    float s = 0.0f;
#pragma unroll 4
    for (int b = w; b < nb; b += 4) s += part[((size_t)(kd * gx + b) * 9 + tap % 9) * 256 + e];
    red[w][o] = s;
    __syncthreads();
    if (w == 0) dW[(size_t)e * 27 + tap] = scale * (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
}

// dW[(row0+co)*IC + ci][tap] = scale * sum_grp part[grp][tap][co][ci].  A workgroup takes 64 consecutive outputs; wave w adds the groups
// w, w+4, w+8, ... (ascending, several loads in flight), the four wave sums are added in wave order: a fixed order, and a quarter of the
// dependent-load chain of one thread per output (that kernel took 60-120 us per layer for 7-30 MB of partials).
__global__ __launch_bounds__(256) void wgrad_reduce_k(const float* __restrict__ part, int n_groups, int KT, int COUT, int CIN, float* __restrict__ dW, int row0,
                                                      int IC, float scale)
{
    __shared__ float red[4][64];
    const int64_t total = (int64_t)KT * COUT * CIN;
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {   // (uniform trip count)
        const int64_t t = base + o;
        float s = 0.0f;
        if (t < total) {
#pragma unroll 4
            for (int g = w; g < n_groups; g += 4) s += part[(size_t)g * total + t];
        }
        red[w][o] = s;
        __syncthreads();
        if (w == 0 && t < total) {
            const int ci = t % CIN, co = (t / CIN) % COUT, tap = (int)(t / ((int64_t)CIN * COUT));
            dW[((size_t)(row0 + co) * IC + ci) * KT + tap] = scale * (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the k3 p1 layers at 4^3 with a rolling window along W (round 3).  wgrad32_k stages one dY block and one X' block per
// position pair and spends 16 MFMAs per wave on them: 192-512 bytes through LDS per MFMA, and by ablation (tools/ablate/wgrad_ablate.hip)
// 25 % of its time goes to fetching and staging, another 20 % to the unequal pair counts of the taps (27 ... 64) with every workgroup
// resident at once.  Here a workgroup keeps (kd, kh) fixed ("class", 9 of them) and walks row pairs (od, oh) -> (id, ih) = (od+kd-1, oh+kh-1);
// step j = 0..3 of a row pair stages dY[ow = j] and X'[iw = j] into a ring of three LDS slots and multiplies
//     dY[j]^T X'[j] -> kw = 1,     dY[j]^T X'[j-1] -> kw = 0,     dY[j-1]^T X'[j] -> kw = 2        (iw = ow + kw - 1)
// i.e. 10 block products per 8 staged blocks instead of 10 per 20, into three accumulators (independent MFMA chains); the blocks of step
// s+2 are in flight in registers, those of step s+1 are written to the ring while step s multiplies: one barrier per step.
// Work split: the (class, tile, row pair) triples in class-major order are M = 100 n_tiles "macro steps" of equal cost; workgroup q of Q
// takes [M q / Q, M (q+1) / Q) and flushes its accumulators whenever the class changes.  The segment of slice q in class c goes to partial
// slot q + c (both only grow along the order, so the slot is unique): part[(q + c) * 3 + kw][co][ci]; wgrad_rows4_reduce_k adds the
// slots of a class in slice order.  Equal work for every workgroup whatever the batch size; no atomics, fixed order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int rows4_nd(int k) { return k == 1 ? 4 : 3; }                 // valid output rows (or planes) of a tap offset
__device__ __forceinline__ int rows4_class_rows(int c) { return rows4_nd(c / 3) * rows4_nd(c % 3); }
__device__ __forceinline__ int rows4_class_start(int c)   // row pairs of the classes before c: 9, 12, 9, 12, 16, 12, 9, 12, 9
{
    int s = 0;
    for (int i = 0; i < c; ++i) s += rows4_class_rows(i);
    return s;
}
__device__ __forceinline__ int64_t rows4_slice_begin(int64_t M, int q, int Q) { return M * q / Q; }
struct Rows4Iter {   // position in the class-major macro-step order, and the step inside the row pair
    int c, tile, r, j, nc;
    __device__ __forceinline__ void seek(int64_t m, int n_tiles)
    {
        c = 0;
        while (c < 8 && (int64_t)rows4_class_start(c + 1) * n_tiles <= m) ++c;
        nc = rows4_class_rows(c);
        const int64_t rem = m - (int64_t)rows4_class_start(c) * n_tiles;
        tile = (int)(rem / nc), r = (int)(rem % nc), j = 0;
    }
    __device__ __forceinline__ void next(int n_tiles)
    {
        if (++j < 4) return;
        j = 0;
        if (++r < nc) return;
        r = 0;
        if (++tile < n_tiles) return;
        tile = 0, ++c, nc = rows4_class_rows(c < 9 ? c : 8);
    }
    __device__ __forceinline__ void positions(int& pi, int& po) const
    {
        const int kd = c / 3, kh = c % 3, nh = rows4_nd(kh), a = r / nh, b = r % nh;
        const int od = (kd == 0) + a, oh = (kh == 0) + b;
        po = od * 16 + oh * 4 + j;
        pi = (od + kd - 1) * 16 + (oh + kh - 1) * 4 + j;
    }
};
// A 32 x 32 output (the encoder's residual block) would be ONE wave doing the staging and the MFMAs in turn; it is cut into four 16 x 16
// quadrants on the 16x16x4 MFMA instead (Q16): four waves share the staged blocks, same flops per wave-cycle.
#ifndef ROWS4_PIPE   // leaf pairs whose MFMA operands are requested ahead (tools/ablate/wgrad_ablate.hip: 0 = the compiler's schedule, 1, 2)
#define ROWS4_PIPE 1
#endif
template <int CIN, int COUT>
constexpr int rows4_threads() { return CIN == 32 && COUT == 32 ? 256 : (COUT / 32) * (CIN / 32) * 64; }
template <int CIN, int COUT, int INMODE, int GIN>
__global__ __launch_bounds__((rows4_threads<CIN, COUT>())) void wgrad_rows4_k(WgradArgs A)
{
    static_assert(CIN % 32 == 0 && COUT % 32 == 0, "32 x 32 output blocks");
    constexpr bool Q16 = CIN == 32 && COUT == 32;
    constexpr int CB = Q16 ? 2 : COUT / 32, IB = Q16 ? 2 : CIN / 32, NT = CB * IB * 64;
    // [leaf][channel] blocks, see wgrad32_k.  Q16: row stride 48 floats — the 64 operand reads of a 16x16x4 MFMA (16 channels x 4 leaves) then
    // fall on every bank exactly twice
    constexpr int SDY = Q16 ? 48 : COUT + 4, SX = Q16 ? 48 : CIN + 4;
    // up to 76.8 KB of static LDS (the 128 <-> 64 channel layers): within gfx950's 160 KB per workgroup, above the 64 KB of every earlier CDNA part
    static_assert(sizeof(float) * 3 * 32 * (SDY + SX) <= 160 * 1024, "wgrad_rows4_k: LDS ring larger than a gfx950 workgroup may hold");
    __shared__ __attribute__((aligned(16))) float sdy[3][32][SDY];
    __shared__ __attribute__((aligned(16))) float sx[3][32][SX];
    // XCD-aware slice number: workgroup b is observed to run on XCD b % 8 (speed only); every XCD gets a contiguous range of slices, i.e. of
    // tiles of a class, so the ten row pairs that share a dY / X' row meet in one L2
    int q;
    const int Q = gridDim.x;
    {
        const int lin = blockIdx.x, xcd = lin & 7, qq = Q >> 3, r = Q & 7;
        q = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (lin >> 3);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cb = wave / IB, ib = wave % IB;
    const int64_t M = (int64_t)100 * A.n_tiles, m0 = rows4_slice_begin(M, q, Q), m1 = rows4_slice_begin(M, q + 1, Q);
    const int total = (int)(m1 - m0) * 4;
    constexpr int CPG = GIN > 0 ? CIN / GIN : 1;
    constexpr int ND = ((COUT / 4) * 32 + NT - 1) / NT, NX = ((CIN / 4) * 32 + NT - 1) / NT;
    f32x4 rdy[ND], rx[NX];
    Rows4Iter itf, itw, itc;   // the step being fetched / written to the ring / multiplied
    itf.seek(m0, A.n_tiles), itw = itf, itc = itf;
    int nf = 0;
    auto fetch = [&]() {
        if (nf >= total) return;   // (uniform)
        ++nf;
        if ((WGRAD_ABL & 1) && nf > 1) {
            itf.next(A.n_tiles);
            return;
        }
        int pi, po;
        itf.positions(pi, po);
        const vq_buf dyb = buf_of((const f32x4*)A.dy + (size_t)itf.tile * 64 * (COUT / 4) * 32);
        const vq_buf xb = buf_of((const f32x4*)A.x + (size_t)itf.tile * 64 * (CIN / 4) * 32);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int i = threadIdx.x + d * NT;
            if (i < (COUT / 4) * 32) rdy[d] = buf_ld16(dyb, (unsigned)i * 16u, (unsigned)po * (COUT / 4) * 512u);
        }
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            const int i = threadIdx.x + d * NT;
            if (i < (CIN / 4) * 32) rx[d] = buf_ld16(xb, (unsigned)i * 16u, (unsigned)pi * (CIN / 4) * 512u);
        }
        itf.next(A.n_tiles);
    };
    float tia[NX][4], tib[NX][4];
    int t_tile = -1;
    auto load_transform = [&](int tile) {
#pragma unroll
        for (int dd = 0; dd < NX; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (CIN / 4) * 32) {
                const int quad = i >> 5, leaf = i & 31;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int ch = 4 * quad + kk;
                    if (INMODE == 1) {
                        const int g = ch / CPG;
                        tia[dd][kk] = A.rstd[((size_t)tile * GIN + g) * 32 + leaf] * A.gamma[ch];
                        tib[dd][kk] = __builtin_fmaf(-A.mean[((size_t)tile * GIN + g) * 32 + leaf], tia[dd][kk], A.beta[ch]);
                    } else if (INMODE == 2) {
                        tia[dd][kk] = A.gate[((size_t)tile * CIN + ch) * 32 + leaf];
                    }
                }
            }
        }
    };
    auto stage = [&](int slot) {
        if (INMODE != 0 && itw.tile != t_tile) load_transform(itw.tile), t_tile = itw.tile;   // (uniform)
        itw.next(A.n_tiles);
        if (WGRAD_ABL & 4) return;
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (COUT / 4) * 32) *(f32x4*)&sdy[slot][i & 31][4 * (i >> 5)] = rdy[dd];
        }
#pragma unroll
        for (int dd = 0; dd < NX; ++dd) {
            const int i = threadIdx.x + dd * NT;
            if (i < (CIN / 4) * 32) {
                const int quad = i >> 5, leaf = i & 31;
                const f32x4 v = rx[dd];
                float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (INMODE == 1) o[kk] = fmaxf(__builtin_fmaf(o[kk], tia[dd][kk], tib[dd][kk]), 0.0f);
                    else if (INMODE == 2) o[kk] = o[kk] * tia[dd][kk];
                }
                *(f32x4*)&sx[slot][leaf][4 * quad] = (f32x4){o[0], o[1], o[2], o[3]};
            }
        }
    };
    f32x16 acc[3];
    f32x4 acc4[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        acc4[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    fetch();
    if (total > 0) {
        stage(0);
        fetch();
    }
    __syncthreads();
    int cur = 0, prev = 2;   // ring slots of step s and s-1
    for (int s = 0; s < total; ++s) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        if (s + 1 < total) {   // (uniform)
            stage(nxt);        // slot of step s-2: its last readers were the MFMAs of step s-1 (barrier below)
            fetch();
        }
        const int rA = lane >> 5;
        if (WGRAD_ABL & 2) {
        } else if (Q16) {
            const int lk = lane >> 4, lc = lane & 15;
            if (itc.j == 0) {   // (uniform)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc4[1] = mfma16(sdy[cur][4 * m + lk][16 * cb + lc], sx[cur][4 * m + lk][16 * ib + lc], acc4[1]);
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float ac = sdy[cur][4 * m + lk][16 * cb + lc], bc = sx[cur][4 * m + lk][16 * ib + lc];
                    const float ap = sdy[prev][4 * m + lk][16 * cb + lc], bp = sx[prev][4 * m + lk][16 * ib + lc];
                    acc4[1] = mfma16(ac, bc, acc4[1]);
                    acc4[0] = mfma16(ac, bp, acc4[0]);
                    acc4[2] = mfma16(ap, bc, acc4[2]);
                }
            }
        } else if (itc.j == 0) {   // (uniform) first position of a row: the centre tap only
#pragma unroll 4
            for (int m = 0; m < 16; ++m) acc[1] = mfma32(sdy[cur][2 * m + rA][32 * cb + (lane & 31)], sx[cur][2 * m + rA][32 * ib + (lane & 31)], acc[1]);
        } else {
#if ROWS4_PIPE
            // operands of leaf pair m+1 requested before the MFMAs of pair m (hand-ordered LDS reads: the compiler issues a trip's reads only
            // after the previous trip's MFMAs and then waits for them)
            const unsigned ya = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)&sdy[cur][rA][32 * cb + (lane & 31)];
            const unsigned xa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)&sx[cur][rA][32 * ib + (lane & 31)];
            const unsigned yp = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)&sdy[prev][rA][32 * cb + (lane & 31)];
            const unsigned xp = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)&sx[prev][rA][32 * ib + (lane & 31)];
            constexpr int PD = ROWS4_PIPE;   // leaf pairs requested ahead
            float o[PD + 1][4];
#define R4(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#pragma unroll
            for (int m = 0; m < PD; ++m) {
                R4(o[m][0], ya, m * 2 * SDY * 4); R4(o[m][1], xa, m * 2 * SX * 4);
                R4(o[m][2], yp, m * 2 * SDY * 4); R4(o[m][3], xp, m * 2 * SX * 4);
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int c_ = m % (PD + 1), n_ = (m + PD) % (PD + 1);
                if (m + PD < 16) {
                    R4(o[n_][0], ya, (m + PD) * 2 * SDY * 4); R4(o[n_][1], xa, (m + PD) * 2 * SX * 4);
                    R4(o[n_][2], yp, (m + PD) * 2 * SDY * 4); R4(o[n_][3], xp, (m + PD) * 2 * SX * 4);
                }
                // (LDS returns in order: everything but the reads of the pairs after m must have arrived)
                const int newer = (16 - 1 - m < PD ? 16 - 1 - m : PD) * 4;
                if (newer == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(o[c_][0]), "+v"(o[c_][1]), "+v"(o[c_][2]), "+v"(o[c_][3]));
                acc[1] = mfma32(o[c_][0], o[c_][1], acc[1]);
                acc[0] = mfma32(o[c_][0], o[c_][3], acc[0]);
                acc[2] = mfma32(o[c_][2], o[c_][1], acc[2]);
            }
#undef R4
#else
#pragma unroll 4
            for (int m = 0; m < 16; ++m) {   // (four leaf pairs per trip: sixteen operand registers live, not sixty-four)
                const float ac = sdy[cur][2 * m + rA][32 * cb + (lane & 31)], bc = sx[cur][2 * m + rA][32 * ib + (lane & 31)];
                const float ap = sdy[prev][2 * m + rA][32 * cb + (lane & 31)], bp = sx[prev][2 * m + rA][32 * ib + (lane & 31)];
                acc[1] = mfma32(ac, bc, acc[1]);
                acc[0] = mfma32(ac, bp, acc[0]);
                acc[2] = mfma32(ap, bc, acc[2]);
            }
#endif
        }
        // end of the class segment (or of the slice): the three taps go to partial slot q + class
        const int c_now = itc.c;
        itc.next(A.n_tiles);
        if (s + 1 == total || itc.c != c_now) {   // (uniform)
            float* dst = A.part + (size_t)(q + c_now) * 3 * COUT * CIN;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (Q16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((size_t)t * COUT + 16 * cb + 4 * (lane >> 4) + r) * CIN + 16 * ib + (lane & 15)] = acc4[t][r];
                    acc4[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = 32 * ib + (lane & 31);
                    dst[((size_t)t * COUT + co) * CIN + ci] = acc[t][r];
                    acc[t][r] = 0.0f;
                }
            }
        }
        prev = cur, cur = nxt;
        if (!(WGRAD_ABL & 4)) __syncthreads();
    }
}
// dW[(row0+co)*IC + ci][tap = 3 c + kw] = scale * sum over the slices q that meet class c of part[(q + c) * 3 + kw][co][ci]; the order of
// wgrad_reduce_k: 64 outputs per workgroup, wave w adds the slices q0 + w, q0 + w + 4, ... ascending, then the four wave sums in wave order
__global__ __launch_bounds__(256) void wgrad_rows4_reduce_k(const float* __restrict__ part, int Q, int n_tiles, int COUT, int CIN, float* __restrict__ dW, int row0,
                                                            int IC, float scale)
{
    __shared__ float red[4][64];
    const int64_t M = (int64_t)100 * n_tiles, total = (int64_t)27 * COUT * CIN;
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {   // (COUT * CIN is a multiple of 64: one tap per trip)
        const int64_t t = base + o;
        const int ci = t % CIN, co = (t / CIN) % COUT, tap = (int)(t / ((int64_t)CIN * COUT)), c = tap / 3, kw = tap % 3;
        const int64_t lo = (int64_t)rows4_class_start(c) * n_tiles, hi = (int64_t)rows4_class_start(c + 1) * n_tiles - 1;   // first / last macro step
        int q0 = (int)(lo * Q / M), q1 = (int)(hi * Q / M);
        while (q0 > 0 && rows4_slice_begin(M, q0, Q) > lo) --q0;
        while (q0 + 1 < Q && rows4_slice_begin(M, q0 + 1, Q) <= lo) ++q0;
        while (q1 > 0 && rows4_slice_begin(M, q1, Q) > hi) --q1;
        while (q1 + 1 < Q && rows4_slice_begin(M, q1 + 1, Q) <= hi) ++q1;
        float s = 0.0f;
        const float* p = part + ((size_t)kw * COUT + co) * CIN + ci;
#pragma unroll 4
        for (int q = q0 + w; q <= q1; q += 4) {
            // (an empty slice, Q > M, wrote nothing: its slot may hold anything)
            const bool live = rows4_slice_begin(M, q + 1, Q) != rows4_slice_begin(M, q, Q);
            const float v = p[(size_t)(q + c) * 3 * COUT * CIN];
            s += live ? v : 0.0f;
        }
        red[w][o] = s;
        __syncthreads();
        if (w == 0) dW[((size_t)(row0 + co) * IC + ci) * 27 + tap] = scale * (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm + ReLU backward:  a = relu(gamma * xh + beta), xh = (x - mean) * rstd.  Given da:
//   gi = [pre > 0] da gamma;  dx = rstd (gi - mean_grp(gi) - xh mean_grp(gi xh));  dgamma = sum [pre>0] da xh;  dbeta = sum [pre>0] da
// pass 1 (gn_bwd_sums_k): per (leaf, group) S1 = sum gi, S2 = sum gi xh, and per (tile, channel, leaf) dgamma / dbeta partials;
// pass 2 (gn_bwd_apply_k): elementwise, optionally adding the gradient of the skip path.  Thread mapping of gn_stats_seq_k.
// ------------------------------------------------------------------------------------------
template <int C, int NP, int CPG>
__global__ __launch_bounds__(64 * C / 8) void gn_bwd_sums_k(const float* __restrict__ x, const float* __restrict__ da, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ s1, float* __restrict__ s2, float* __restrict__ dgam, float* __restrict__ dbet)
{
    constexpr int G = C / CPG;
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int quad = 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
    const int tile = blockIdx.x;
    float mu[4], rs[4], ia[4], ib[4], gm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = 4 * quad + k, g = ch / CPG;
        mu[k] = mean[((size_t)tile * G + g) * 32 + j];
        rs[k] = rstd[((size_t)tile * G + g) * 32 + j];
        gm[k] = gamma[ch];
        ia[k] = rs[k] * gm[k];
        ib[k] = __builtin_fmaf(-mu[k], ia[k], beta[ch]);
    }
    const size_t base = (size_t)tile * NP * (C / 4) * 32 + quad * 32 + j;
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    // gridDim.y position ranges per tile; partial sums are indexed by (tile * gridDim.y + blockIdx.y) and added up by the consumers
    const int p0 = (int)(blockIdx.y * NP / gridDim.y), p1 = (int)((blockIdx.y + 1) * NP / gridDim.y);
    const size_t pt = (size_t)tile * gridDim.y + blockIdx.y;
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const f32x4 xv = ((const f32x4*)x)[base + (size_t)p * (C / 4) * 32];
        const f32x4 dv = ((const f32x4*)da)[base + (size_t)p * (C / 4) * 32];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xs[k] - mu[k]) * rs[k];
            const float d = __builtin_fmaf(xs[k], ia[k], ib[k]) > 0.0f ? ds[k] : 0.0f;
            const float gi = d * gm[k];
            a1[k] += gi;
            a2[k] = __builtin_fmaf(gi, xh, a2[k]);
            dg[k] = __builtin_fmaf(d, xh, dg[k]);
            db[k] += d;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dgam[(pt * C + 4 * quad + k) * 32 + j] = dg[k];
        dbet[(pt * C + 4 * quad + k) * 32 + j] = db[k];
    }
    if (CPG == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            s1[(pt * G + 2 * quad + k) * 32 + j] = a1[2 * k] + a1[2 * k + 1];
            s2[(pt * G + 2 * quad + k) * 32 + j] = a2[2 * k] + a2[2 * k + 1];
        }
    } else {
        float t1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), t2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
        if (CPG == 8) {
            t1 += __shfl_xor(t1, 32, 64);
            t2 += __shfl_xor(t2, 32, 64);
        }
        if (CPG == 4 || (lane >> 5) == 0) {
            const int g = CPG == 4 ? quad : quad >> 1;
            s1[(pt * G + g) * 32 + j] = t1;
            s2[(pt * G + g) * 32 + j] = t2;
        }
    }
}
// out[tile][i] = sum over the n_split position ranges of part[tile][range][i]   (i over G*32 values per tile)
// ... and, in the same launch (the workgroups behind the combine's), the two per-channel reductions of the GroupNorm-affine gradients
// (chan_reduce_k's arithmetic for dgamma and dbeta): three launches of 5-15 us on the data-gradient chain became one
__global__ __launch_bounds__(256) void gn_bwd_finish_k(const float* __restrict__ p1, const float* __restrict__ p2, int n_split, int per_tile, int n_tiles,
                                                       float* __restrict__ o1, float* __restrict__ o2, const float* __restrict__ part_g,
                                                       const float* __restrict__ part_b, int C, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int nb = (n_tiles * per_tile + 255) / 256;
    if ((int)blockIdx.x < nb) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        if (t >= n_tiles * per_tile) return;
        const int tile = t / per_tile, i = t % per_tile;
        float a = 0.0f, b = 0.0f;
        for (int sp = 0; sp < n_split; ++sp) {
            a += p1[((size_t)tile * n_split + sp) * per_tile + i];
            b += p2[((size_t)tile * n_split + sp) * per_tile + i];
        }
        o1[t] = a;
        o2[t] = b;
        return;
    }
    __shared__ float red[256];
    const int cb = (int)blockIdx.x - nb, c = cb % C, t = threadIdx.x;
    const float* __restrict__ part = cb < C ? part_g : part_b;
    const int nt2 = n_tiles * n_split;
    float s = 0.0f;
    for (int tile = t >> 5; tile < nt2; tile += 8) s += part[((size_t)tile * C + c) * 32 + (t & 31)];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) (cb < C ? dgamma : dbeta)[c] = red[0];
}
template <int C, int NP, int CPG>
__global__ __launch_bounds__(256) void gn_bwd_apply_k(const float* __restrict__ x, const float* __restrict__ da, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ s1, const float* __restrict__ s2, int n_split, const float* __restrict__ add,
                                                      float* __restrict__ dx)
{
    constexpr int G = C / CPG;
    constexpr float inv_n = 1.0f / (float)(CPG * NP);
    const int tile = blockIdx.x;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < NP * (C / 4) * 32; i += gridDim.y * 256) {
        const int j = i & 31, quad = (i >> 5) % (C / 4);
        const size_t o = (size_t)tile * NP * (C / 4) * 32 + i;
        const f32x4 xv = ((const f32x4*)x)[o], dv = ((const f32x4*)da)[o];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = 4 * quad + k, g = ch / CPG;
            const float mu = mean[((size_t)tile * G + g) * 32 + j], rs = rstd[((size_t)tile * G + g) * 32 + j];
            const float ia = rs * gamma[ch], ib = __builtin_fmaf(-mu, ia, beta[ch]);
            const float xh = (xs[k] - mu) * rs;
            const float gi = __builtin_fmaf(xs[k], ia, ib) > 0.0f ? ds[k] * gamma[ch] : 0.0f;
            float t1 = 0.0f, t2 = 0.0f;
            for (int sp = 0; sp < n_split; ++sp) {
                t1 += s1[(((size_t)tile * n_split + sp) * G + g) * 32 + j];
                t2 += s2[(((size_t)tile * n_split + sp) * G + g) * 32 + j];
            }
            r[k] = rs * (gi - t1 * inv_n - xh * (t2 * inv_n));
        }
        f32x4 out = {r[0], r[1], r[2], r[3]};
        if (add) out = out + ((const f32x4*)add)[o];
        ((f32x4*)dx)[o] = out;
    }
}

// The same in ONE launch (round 4).  On the data-gradient chain the three launches above cost 1.06 ms of a 4.6 ms step (2048 leaves per
// rank, profiles/r04_training_step_timeline.txt): each reads x and da from HBM again, next to the weight gradients of the other stream.
// Here a workgroup owns whole GroupNorm groups — (tile, QW channel quads, LW leaves) for ALL positions — and every thread keeps its PT
// positions of (xh, gi) in registers between the reduction and the elementwise phase: x and da are read once.  The sums meet in LDS in a
// fixed order (position ranges ascending), so the result does not depend on scheduling.  Per-(tile, channel, leaf) dgamma / dbeta
// partials leave in gn_bwd_sums_k's layout (one position range) for gn_bwd_finish_k's channel reduction.
//   C = 64: group = 2 quads, 512 threads;  C = 32: group = 1 quad, 256 threads;  C = 16: one quad (two groups of 2 or one of 4) and
//   16 leaves per workgroup, 1024 threads.  Eight positions per thread everywhere; grid (tiles, 8).
template <int C, int NP, int CPG, int QW, int LW, int NT>
__global__ __launch_bounds__(NT) void gn_bwd_fused_k(const float* __restrict__ x, const float* __restrict__ da, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ add, float* __restrict__ dx, float* __restrict__ dgam, float* __restrict__ dbet)
{
    constexpr int G = C / CPG, COLS = QW * LW, PS = NT / COLS, PT = NP / PS, NWV = NT / 64, RPW = 64 / COLS;
    constexpr int NGL = CPG == 2 ? 2 : 1;            // groups a thread's four channels belong to
    constexpr int NV = 2 * NGL + 8;                  // values a thread contributes: (S1, S2) per group, dgamma x4, dbeta x4
    static_assert(PT * PS == NP && COLS * PS == NT && (COLS == 16 || COLS == 32 || COLS == 64), "thread mapping");
    static_assert((C / 4 / QW) * (32 / LW) == 8, "gridDim.y = 8 units per tile");
    static_assert(CPG == 2 || CPG == 4 || (CPG == 8 && QW == 2), "a workgroup owns whole groups");
    constexpr float inv_n = 1.0f / (float)(CPG * NP);
    __shared__ float red[NWV][COLS][NV];
    __shared__ float gsum[COLS][2 * NGL];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int col = tid % COLS, ps = tid / COLS;
    const int tile = blockIdx.x, unit = blockIdx.y;
    const int q0 = LW == 32 ? unit * QW : unit >> 1, leaf0 = LW == 32 ? 0 : 16 * (unit & 1);
    const int quad = q0 + col / LW, j = leaf0 + col % LW;
    float mu[4], rs[4], ia[4], ib[4], gm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = 4 * quad + k, g = ch / CPG;
        mu[k] = mean[((size_t)tile * G + g) * 32 + j];
        rs[k] = rstd[((size_t)tile * G + g) * 32 + j];
        gm[k] = gamma[ch];
        ia[k] = rs[k] * gm[k];
        ib[k] = __builtin_fmaf(-mu[k], ia[k], beta[ch]);
    }
    const size_t base = (size_t)tile * NP * (C / 4) * 32 + quad * 32 + j + (size_t)(ps * PT) * (C / 4) * 32;
    f32x4 xv[PT], dv[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) xv[i] = ((const f32x4*)x)[base + (size_t)i * (C / 4) * 32], dv[i] = ((const f32x4*)da)[base + (size_t)i * (C / 4) * 32];
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        float xs[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, ds[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xs[k] - mu[k]) * rs[k];
            const float d = __builtin_fmaf(xs[k], ia[k], ib[k]) > 0.0f ? ds[k] : 0.0f;
            const float gi = d * gm[k];
            a1[k] += gi;
            a2[k] = __builtin_fmaf(gi, xh, a2[k]);
            dg[k] = __builtin_fmaf(d, xh, dg[k]);
            db[k] += d;
            xs[k] = xh, ds[k] = gi;   // kept for the elementwise phase
        }
        xv[i] = (f32x4){xs[0], xs[1], xs[2], xs[3]}, dv[i] = (f32x4){ds[0], ds[1], ds[2], ds[3]};
    }
    float v[NV];
    if (CPG == 2) v[0] = a1[0] + a1[1], v[1] = a2[0] + a2[1], v[2] = a1[2] + a1[3], v[3] = a2[2] + a2[3];
    else v[0] = (a1[0] + a1[1]) + (a1[2] + a1[3]), v[1] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[2 * NGL + k] = dg[k], v[2 * NGL + 4 + k] = db[k];
    // the RPW position ranges of a wave that share a column: lanes COLS apart
#pragma unroll
    for (int m = COLS; m < 64; m <<= 1)
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] += __shfl_xor(v[e], m, 64);
    if ((tid & 63) < COLS) {
#pragma unroll
        for (int e = 0; e < NV; ++e) red[wave][col][e] = v[e];
    }
    __syncthreads();
    // the column's totals over the position ranges (waves ascending), by one thread per column; dgamma / dbeta leave from here
    if (tid < COLS) {
        float tot[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) tot[e] = 0.0f;
#pragma unroll 2
        for (int w = 0; w < NWV; ++w)
#pragma unroll
            for (int e = 0; e < NV; ++e) tot[e] += red[w][col][e];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dgam[((size_t)tile * C + 4 * quad + k) * 32 + j] = tot[2 * NGL + k];
            dbet[((size_t)tile * C + 4 * quad + k) * 32 + j] = tot[2 * NGL + 4 + k];
        }
#pragma unroll
        for (int e = 0; e < 2 * NGL; ++e) gsum[col][e] = tot[e];
    }
    __syncthreads();
    float t1[NGL], t2[NGL];
#pragma unroll
    for (int g = 0; g < NGL; ++g) {
        float s1 = gsum[col][2 * g], s2 = gsum[col][2 * g + 1];
        if (CPG == 8) s1 += gsum[col ^ 32][2 * g], s2 += gsum[col ^ 32][2 * g + 1];   // the group's other quad (commutative: both quads get the same bits)
        t1[g] = s1 * inv_n, t2[g] = s2 * inv_n;
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const float xh[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, gi[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = CPG == 2 ? k >> 1 : 0;
            r[k] = rs[k] * (gi[k] - t1[g] - xh[k] * t2[g]);
        }
        f32x4 out = {r[0], r[1], r[2], r[3]};
        if (add) out = out + ((const f32x4*)add)[base + (size_t)i * (C / 4) * 32];
        ((f32x4*)dx)[base + (size_t)i * (C / 4) * 32] = out;
    }
}

// ------------------------------------------------------------------------------------------
// ChannelAttention backward (y = x * g, g = sigmoid(W2 relu(W0 mean_pos(x))), VQVAE_v2.py:213-228), one wave per tile.
// dyA (+ dyB): gradient(s) wrt y.  dx = dy*g + dm/NP;  partial weight gradients summed over the tile's leaves:
// pfc0[tile][R][C], pfc2[tile][C][R].
// ------------------------------------------------------------------------------------------
template <int C, int NP>
__global__ __launch_bounds__(64 * C / 8) void se_bwd_k(const float* __restrict__ x, const float* __restrict__ dyA, const float* __restrict__ dyB,
                                                       const float* __restrict__ csum, const float* __restrict__ fc0, const float* __restrict__ fc2,
                                                       float* __restrict__ dx, float* __restrict__ pfc0, float* __restrict__ pfc2)
{
    // C/8 waves; wave w owns channel quads 2w, 2w+1 (lane half h -> quad 2w+h) in the position sweeps; wave 0 does the per-leaf MLP
    constexpr int R = C / 4;
    __shared__ float dgl[C][32], gl[C][32], dml[C][32], hl[R][32], dhl[R][32], ml[C][32];
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int quad = 2 * wave + h;
    const size_t base = (size_t)tile * NP * (C / 4) * 32 + quad * 32 + j;
    {
        f32x4 s = {0, 0, 0, 0};
        // (sixteen positions in flight per thread: the sweep is a chain of HBM round trips, four at a time it took a third of the kernel)
#pragma unroll 16
        for (int p = 0; p < NP; ++p) {
            const size_t o = base + (size_t)p * (C / 4) * 32;
            f32x4 d = ((const f32x4*)dyA)[o];
            if (dyB) d = d + ((const f32x4*)dyB)[o];
            s = s + d * ((const f32x4*)x)[o];
        }
        dgl[4 * quad + 0][j] = s.x, dgl[4 * quad + 1][j] = s.y, dgl[4 * quad + 2][j] = s.z, dgl[4 * quad + 3][j] = s.w;
    }
    // the tile's channel means, once (the weight-gradient loop below read csum from global memory inside its leaf loop: 32 dependent
    // round trips per entry)
    for (int e = threadIdx.x; e < C * 32; e += 64 * C / 8) ml[e >> 5][e & 31] = csum[(size_t)tile * C * 32 + e] * (1.0f / (float)NP);
    __syncthreads();
    const float* cs = csum + (size_t)tile * C * 32 + j;
    if (wave == 0) {
        float hid[R], dh[R];
        se_hidden<C>(cs, fc0, hid);
#pragma unroll
        for (int r = 0; r < R; ++r) dh[r] = 0.0f;
        for (int c = 0; c < C; ++c) {   // gate, da = dg * g (1-g), dh += W2^T da
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) a = __builtin_fmaf(fc2[c * R + r], hid[r], a);
            const float gg = vq_sigmoid(a);
            const float da = dgl[c][j] * (gg * (1.0f - gg));
#pragma unroll
            for (int r = 0; r < R; ++r) dh[r] = __builtin_fmaf(fc2[c * R + r], da, dh[r]);
            if (h == 0) {
                gl[c][j] = gg;
                dgl[c][j] = da;   // reused: da
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dh[r] = hid[r] > 0.0f ? dh[r] : 0.0f;
            if (h == 0) hl[r][j] = hid[r], dhl[r][j] = dh[r];
        }
        for (int c = h; c < C; c += 2) {   // dm = W0^T dh, two channels per pass (one per lane half)
            float dm = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) dm = __builtin_fmaf(fc0[r * C + c], dh[r], dm);
            dml[c][j] = dm * (1.0f / (float)NP);
        }
    }
    __syncthreads();
    // FC weight-gradient partials, summed over the tile's 32 leaves: thread t owns entries t, t + blockDim, ...
    for (int e = threadIdx.x; e < C * R; e += 64 * C / 8) {
        const int c2 = e / R, r2 = e % R;      // pfc2[c][r] = sum_j da[c][j] h[r][j]
        const int r0 = e / C, c0 = e % C;      // pfc0[r][c] = sum_j dh[r][j] m[c][j]
        float v2 = 0.0f, v0 = 0.0f;
        for (int jj = 0; jj < 32; ++jj) {
            v2 = __builtin_fmaf(dgl[c2][jj], hl[r2][jj], v2);
            v0 = __builtin_fmaf(dhl[r0][jj], ml[c0][jj], v0);
        }
        pfc2[(size_t)tile * C * R + e] = v2;
        pfc0[(size_t)tile * C * R + e] = v0;
    }
    {
        const f32x4 g4 = {gl[4 * quad][j], gl[4 * quad + 1][j], gl[4 * quad + 2][j], gl[4 * quad + 3][j]};
        const f32x4 m4 = {dml[4 * quad][j], dml[4 * quad + 1][j], dml[4 * quad + 2][j], dml[4 * quad + 3][j]};
#pragma unroll 16
        for (int p = 0; p < NP; ++p) {
            const size_t o = base + (size_t)p * (C / 4) * 32;
            f32x4 d = ((const f32x4*)dyA)[o];
            if (dyB) d = d + ((const f32x4*)dyB)[o];
            ((f32x4*)dx)[o] = d * g4 + m4;
        }
    }
}

// straight-through estimator + commitment loss (VQVAE_v2.py:146-150): dz = dq + cst * (z - q), cst = 2*commitment/(rows*128) over the
// GLOBAL batch; zero for the padded leaves of the last tile.  All three tensors in the L4 layout [tile][64][32][32][4].
__global__ __launch_bounds__(256) void st_grad_k(const float* __restrict__ dq, const float* __restrict__ z, const float* __restrict__ q, float* __restrict__ dz,
                                                 float cst, int64_t n_leaves, int n_tiles)
{
    const int64_t total = (int64_t)n_tiles * 64 * 32 * 32;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t leaf = (t >> 16) * 32 + (t & 31);
        f32x4 r = {0.0f, 0.0f, 0.0f, 0.0f};
        if (leaf < n_leaves) r = ((const f32x4*)dq)[t] + (((const f32x4*)z)[t] - ((const f32x4*)q)[t]) * cst;
        ((f32x4*)dz)[t] = r;
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient of the down conv (Conv3d 16->32, k4 s2 p1, VQVAE_v2.py:239): a transposed convolution 32ch@4^3 -> 16ch@8^3.
// 16x16x4 MFMA: rows = 16 input channels of the forward conv, cols = 16 leaves, K = the 32 output channels (two float4 per
// lane).  Input voxel P receives from tap k (per axis) iff (P + 1 - k) is even and o = (P + 1 - k)/2 lies in [0,4): at most two
// taps per axis.  One wave per 16-leaf half tile walks the 512 voxels.
// wfrag[((tap*2 + blk)*64 + lane)*4 + i] = W[16 blk + 4 (lane>>4) + i][lane & 15][tap]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv_down_k(const float* __restrict__ dy /*L4 32ch@4^3*/, const float* __restrict__ wfrag, float* __restrict__ dx /*L4 16ch@8^3*/,
                                                     int n_tiles)
{
    const int lane = threadIdx.x & 63;
    const int half = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tile = half >> 1;
    if (tile >= n_tiles) return;
    const int jj = (lane & 15) + 16 * (half & 1), q4 = lane >> 4;
    const f32x4* in4 = (const f32x4*)dy + (size_t)tile * 64 * 8 * 32 + jj;
    f32x4* out4 = (f32x4*)dx + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj;
    const f32x4* wf = (const f32x4*)wfrag + lane;
    const int P0 = (int)(blockIdx.y * 512 / gridDim.y), P1 = (int)((blockIdx.y + 1) * 512 / gridDim.y);
    for (int P = P0; P < P1; ++P) {
        const int D = P >> 6, H = (P >> 3) & 7, Wd = P & 7;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int a = 0; a < 2; ++a) {
            const int kd = ((D + 1) & 1) + 2 * a, od = (D + 1 - kd) >> 1;
            if (od < 0 || od > 3) continue;
            for (int b = 0; b < 2; ++b) {
                const int kh = ((H + 1) & 1) + 2 * b, oh = (H + 1 - kh) >> 1;
                if (oh < 0 || oh > 3) continue;
                for (int e = 0; e < 2; ++e) {
                    const int kw = ((Wd + 1) & 1) + 2 * e, ow = (Wd + 1 - kw) >> 1;
                    if (ow < 0 || ow > 3) continue;
                    const int tap = (kd * 4 + kh) * 4 + kw, o = (od * 4 + oh) * 4 + ow;
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const f32x4 w = wf[(size_t)(tap * 2 + blk) * 64];
                        const f32x4 v = in4[((size_t)o * 8 + 4 * blk + q4) * 32];
                        acc = mfma16(w.x, v.x, acc);
                        acc = mfma16(w.y, v.y, acc);
                        acc = mfma16(w.z, v.z, acc);
                        acc = mfma16(w.w, v.w, acc);
                    }
                }
            }
        }
        out4[(size_t)P * 4 * 32] = acc;
    }
}

// weight gradient of the first conv (1 -> 16, k3 p1 @8^3): part[tile * 8 + od][16*27] ([co*27 + tap]); grid (tiles, 8 output planes);
// 9 waves = (kd, kh), 3 kw each; lane = (leaf, half of the plane's 64 positions).  (One workgroup per tile walked all 512 positions: 64
// workgroups at 2048 leaves, a quarter of the CUs, 0.18 ms for 67 MB of dY.)
__global__ __launch_bounds__(576) void wgrad_first_k(const float* __restrict__ dy /*L4 16ch@8^3*/, const float* __restrict__ xt /*[tile][512][32]*/,
                                                     float* __restrict__ part)
{
    const int tile = blockIdx.x, D = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int kd = wave / 3, kh = wave % 3;
    float acc[3][16];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[k][c] = 0.0f;
    const f32x4* dy4 = (const f32x4*)dy + (size_t)tile * 512 * 4 * 32 + j;
    const float* x = xt + (size_t)tile * 512 * 32 + j;
    const int d = D + kd - 1;
    if (d >= 0 && d <= 7) {   // (wave-uniform)
#pragma unroll 2
        for (int P = 64 * D + 32 * h; P < 64 * D + 32 * h + 32; ++P) {
            const int H = (P >> 3) & 7, Wd = P & 7;
            const int hh = H + kh - 1;
            if (hh < 0 || hh > 7) continue;
            f32x4 g[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) g[qd] = dy4[((size_t)P * 4 + qd) * 32];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xw = Wd + kw - 1;
                if (xw < 0 || xw > 7) continue;
                const float xv = x[(size_t)((d * 8 + hh) * 8 + xw) * 32];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    acc[kw][4 * qd + 0] = __builtin_fmaf(g[qd].x, xv, acc[kw][4 * qd + 0]);
                    acc[kw][4 * qd + 1] = __builtin_fmaf(g[qd].y, xv, acc[kw][4 * qd + 1]);
                    acc[kw][4 * qd + 2] = __builtin_fmaf(g[qd].z, xv, acc[kw][4 * qd + 2]);
                    acc[kw][4 * qd + 3] = __builtin_fmaf(g[qd].w, xv, acc[kw][4 * qd + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = acc[kw][c];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) part[((size_t)tile * 8 + D) * 432 + c * 27 + (kd * 3 + kh) * 3 + kw] = v;
        }
}

// torch.optim.AdamW (decoupled weight decay) on the flat parameter vector; bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t)
__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                               float lr, float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2s)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * weight_decay);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
        p[i] = pi;
    }
}
