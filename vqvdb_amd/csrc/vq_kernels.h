// vq_kernels.h — gfx950 kernels of the VQ-VAE leaf codec (encode+quantize, decode).
//
// Common design: one WAVEFRONT owns one leaf tile (32 leaves) for a whole
// layer and walks the output positions; the GEMM is  D[cout][leaf] += W[cout][k] * X[k][leaf]
// with the weights as the MFMA A operand (pre-shuffled on the host into fragment order, read
// from LDS with one ds_read_b128 per 4 MFMAs) and the activations as the B operand, read
// straight from HBM/L2 in the leaf-tile layout (one coalesced float4 per lane per 4 MFMAs).
// Because a whole MFMA tile sits at ONE spatial position, zero-padding taps are skipped
// exactly (wave-uniform loop bounds) instead of being multiplied by zero.
// Every accumulation order here is restated by oracle/vqvae_oracle.c; encode is bit-exact.
#pragma once
#include "vq_device.h"

struct ConvArgs {
    const float* in;         // L4 activations [tile][NPI][CIN/4][32][4]
    float* out;              // L4 activations [tile][NPO][COUT/4][32][4] (or pixel-shuffled)
    const float* wfrag;      // fragment-ordered weights
    const float* bias_frag;  // bias in D-fragment order (mfma32) or plain (mfma16)
    const float* skip;       // residual input (same layout as out)
    const float* in_mean;    // [tile][GIN][32]
    const float* in_rstd;
    const float* in_gamma;   // [CIN]
    const float* in_beta;
    const float* se_csum;    // [tile][CIN][32] channel sums over the 64 positions
    const float* se_fc0;     // [CIN/4][CIN]
    const float* se_fc2;     // [CIN][CIN/4]
    const float* se_gate;    // tail_small_k: gates precomputed per tile by csum_seq_k, [tile][CIN][32]
    float* out_mean;         // [tile][GOUT][32]
    float* out_rstd;
    float* out_csum;         // [tile][COUT][32]
    int n_steps;             // length of the flattened (output, valid taps) schedule passed beside ConvArgs
    int n_taps;
    int64_t n_leaves;        // OUTMODE 2 only: leaves that really exist (the last tile may be padded)
    int n_tiles;
    // Position-split launches (small batches: too few leaf tiles to fill 1024 SIMDs): gridDim.y > 1 cuts the kernel's
    // output groups (the units its outer loop walks: rows, row groups or positions) into gridDim.y ranges;
    // grp_start[g] = index of the first schedule step of group g.
    const int* grp_start;
    // ... with fused statistics: a split launch covers whole statistics blocks (16 per leaf) and stores each block's sums
    // here instead of finishing; gn_combine_k / csum_combine_k add the blocks in order (same result as the one-wave-per-tile launch)
    double* part_s;          // [tile][16 blocks][16 slots][32]
    double* part_q;
    float* part_c;           // channel sums: [tile][16 blocks][64 channels][32]
};

// group range [g0, g1) of this workgroup for a kernel whose outer loop walks NG output groups
template <int NG>
__device__ __forceinline__ void split_range(int& g0, int& g1)
{
    g0 = 0, g1 = NG;
    if (gridDim.y > 1) {
        g0 = (int)(blockIdx.y * NG / gridDim.y);
        g1 = (int)((blockIdx.y + 1) * NG / gridDim.y);
    }
}

// One step = one output position/row and a run of 1..KWG valid taps along kw.  x: first input
// position (or input row base), y: first tap index (weight fragment), z: output position (or
// output row base), w: bit0 = first step of this output, bit1 = last step, bits 8.. = number of
// taps in the run.  Zero-padding taps simply do not appear in the table.
struct StepEnt {
    int ip, tap, po, flags;
};

// async global -> LDS copy of 64 float4 (1 KiB) per wave-instruction; LDS destination is the
// wave-uniform base + lane*16 (cdna guide §5), tracked by vmcnt.
__device__ __forceinline__ void glds16(const f32x4* gsrc_lane, f32x4* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS reads the compiler does not see as LDS reads.  Behind a global_load_lds the compiler puts "s_waitcnt vmcnt(<= that copy)" in
// front of EVERY ordinary LDS load that follows (it cannot tell the window being filled from the one being read), so a kernel that
// streams the next step's weights while it reads this step's would begin each step by waiting for the copy it has just started.
// The kernels below order the two by hand (vmcnt + barrier at the step boundary), so the fragment reads go through these:
// lds_frag_read issues, lds_frags_wait() + lds_frag_use() in front of the first use make the data dependence visible again.
__device__ __forceinline__ unsigned lds_offset_of(const void* p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ f32x4 lds_frag_read(unsigned byte_addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ void lds_frags_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_frag_use(f32x4& v) { asm volatile("" : "+v"(v)); }

// ------------------------------------------------------------------------------------------
// pack: host/leaf-major leaves [n][512] (VQVAECodec.cpp:36-59 layout in) ->
//   xr[tile][row 64][leaf 32][12]   the first conv's input: one W-row of a leaf = (0, x0 .. x7, 0, -, -), halo zeros included, so
//                                   that a lane reads its eight B operands x[ow + kw - 1] (ow = 0..7) as two 16-byte loads
//   xt[tile][512][32]               position-major copy, only when `xt` != nullptr (training: loss and first-conv weight gradients)
// ------------------------------------------------------------------------------------------
#define VQ_XR_REC 12              // floats per (row, leaf) record of xr
#define VQ_XR_TILE (64 * 32 * VQ_XR_REC)
__global__ __launch_bounds__(256) void pack_leaves_k(const float* __restrict__ leaves, float* __restrict__ xr, float* __restrict__ xt, int64_t n_leaves)
{
    __shared__ float tile[32][65];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    // gridDim.y = 1: one workgroup walks the 8 position chunks of its tile; gridDim.y = 8 (small batches): one chunk each
    const int pc0 = gridDim.y > 1 ? blockIdx.y * 64 : 0, pc1 = gridDim.y > 1 ? pc0 + 64 : 512;
    for (int p0 = pc0; p0 < pc1; p0 += 64) {
        // read: 32 leaves x 64 positions, position fastest (256-B runs per leaf)
        for (int i = tid; i < 32 * 64; i += 256) {
            const int l = i >> 6, p = i & 63;
            const int64_t leaf = (int64_t)t * 32 + l;
            tile[l][p] = leaf < n_leaves ? leaves[leaf * 512 + p0 + p] : 0.0f;
        }
        __syncthreads();
        if (xt) {
            for (int i = tid; i < 32 * 64; i += 256) {
                const int p = i >> 5, l = i & 31;
                xt[((int64_t)t * 512 + p0 + p) * 32 + l] = tile[l][p];
            }
        }
        // the chunk's 8 rows: 8 x 32 records of 12 floats, written as one contiguous run
        float* dst = xr + (size_t)t * VQ_XR_TILE + (size_t)(p0 >> 3) * 32 * VQ_XR_REC;
        for (int i = tid; i < 8 * 32 * VQ_XR_REC; i += 256) {
            const int r = i / (32 * VQ_XR_REC), rem = i % (32 * VQ_XR_REC), l = rem / VQ_XR_REC, e = rem % VQ_XR_REC;
            dst[i] = (e >= 1 && e <= 8) ? tile[l][r * 8 + e - 1] : 0.0f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// E1-E2: Conv3d(1->16,k3,p1) @8^3 (VQVAE_v2.py:235) + GroupNorm(4,16) + ReLU (:236-237) and the
// statistics of the result for ResidualBlock.gn1 (:205).
// 16x16x4 MFMA: rows = 16 couts, cols = 16 leaves, K = (kd,kh) x {kw0,kw1,kw2,pad}.  A wave owns a
// 32-leaf tile (two 16-leaf sub-tiles) and a full output row of 8 positions (16 independent
// accumulators); one step = (output row, valid kd) = up to 3 kh x 16 MFMAs, the next step's input (3 rows x 2 sub-tiles x
// two 16-byte loads from the row layout xr, see pack_leaves_k) is prefetched meanwhile.  Twelve wide loads per step instead of
// 48 dword loads: the kernel used to be bound by the rate at which a CU accepts vector-memory instructions (ablations in
// tools/ablate/conv_first_ablate.hip: 0.45 -> 0.32 ms for the statistics pass), not by its fp64 statistics.  The conv is so cheap (221 k MAC/leaf) that it is run TWICE instead
// of materialising its 32 KiB/leaf output:
//   MODE 0: statistics of y1 = conv(x)+bias for GroupNorm(4,16) (no activation store unless A.out != 0)
//   MODE 1: recompute y1, a1 = relu(gn(y1)) -> store, statistics of a1 for GroupNorm(8,16).
// ------------------------------------------------------------------------------------------
// ABL (tools/ablate/conv_first_ablate.hip only): 1 no input loads inside the loop, 2 no MFMAs, 4 no statistics, 8 no stores
// RAW (round 6): the loads read the CALLER'S layout float[leaf][512] directly instead of the row layout pack_leaves_k writes (one launch and
// 5 KiB per leaf of traffic less): a lane's 8 floats of (row, kw) start at element row * 8 + kw - 1 of its leaf; the two halo elements that
// fall into the neighbouring rows (x[-1] of the kw = 0 slot, x[8] of the kw = 2 slot) are replaced by zeros with a lane select, leaves past
// n_leaves lie beyond the tile's buffer range and read as zeros.  Same operands, same MFMAs.  Measured SLOWER than the row layout + pack_leaves_k (a load touches 16 cache lines instead of 6): kept behind VQHIP_FIRST_SRC=raw.
template <int MODE, int ABL = 0, bool RAW = false>
__global__ __launch_bounds__(256, 2) void conv_first_k(ConvArgs A, const int4* __restrict__ steps)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    // K slot q4 = kw (3 = pad).  A lane's eight B operands of a row, x[ow + kw - 1] for ow = 0..7, are the 8 floats starting at
    // element kw of the row's record (0, x0..x7, 0): two 16-byte loads at 4-byte alignment.  The pad slot reads a record of zeros
    // (stride 0), so it contributes fmaf(0, 0, acc) exactly like the contract says.
    // Buffer addressing (see buf_ld16): row r, sub-tile sb, half row hf of this lane's K slot.  The pad slot's lane offset lies beyond
    // the descriptor's range, and an out-of-range buffer load returns 0.
    const vq_buf xb = RAW ? buf_of_n(A.in + (size_t)tile * 32 * 512, (unsigned)min((int64_t)32, A.n_leaves - (int64_t)tile * 32) * 2048u)
                          : buf_of(A.in + (size_t)tile * VQ_XR_TILE);
    const unsigned lane_x = q4 < 3 ? (RAW ? (unsigned)(jj * 512) * 4u : (unsigned)(jj * VQ_XR_REC + q4) * 4u) : 0x80000000u;
    // RAW: a load whose FIRST byte lies outside the buffer range is dropped whole, so no lane may start below its leaf or end above it:
    // the kw = 0 slot starts at x[0] and shifts right by one (x[-1] = 0), the kw = 2 slot of the upper half starts at x[4] and shifts left
    // (x[8] = 0).  The leaf sits in the LANE offset for both sub-tiles (the range check of the ragged last tile must see it).
    const bool halo_lo = q4 == 0, halo_hi = q4 == 2;
    const unsigned raw_lo = lane_x + (q4 < 3 ? (unsigned)(q4 > 0 ? q4 - 1 : 0) * 4u : 0u);            // hf = 0: first element max(kw - 1, 0)
    const unsigned raw_hi = lane_x + (q4 < 3 ? (unsigned)(4 + (q4 == 2 ? 0 : q4 - 1)) * 4u : 0u);     // hf = 1: first element 4 + kw - 1, kw = 2: 4
    auto ldrow = [&](int r, int sb, int hf) -> f32x4 {
        if (!RAW) return buf_ld16(xb, lane_x, (unsigned)((r * 32 + sb * 16) * VQ_XR_REC + hf * 4) * 4u);
        f32x4 v = buf_ld16(xb, (hf ? raw_hi : raw_lo) + (q4 < 3 && sb ? 16u * 2048u : 0u), (unsigned)(r * 8) * 4u);
        if (hf == 0) v = halo_lo ? (f32x4){0.0f, v.x, v.y, v.z} : v;
        else v = halo_hi ? (f32x4){v.y, v.z, v.w, 0.0f} : v;
        return v;
    };
    const bool has_out = A.out != nullptr;
    const vq_buf outb = buf_of(has_out ? (const f32x4*)A.out + (size_t)tile * 512 * 4 * 32 : (const f32x4*)A.in);
    const unsigned lane_o = (unsigned)(q4 * 32 + jj) * 16u;
    f32x4 ia[2], ib[2];  // MODE 1: GroupNorm(4,16) of y1, group = q4 (this lane's 4 couts)
    if (MODE == 1) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const float mean = A.in_mean[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
            const float rstd = A.in_rstd[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ia[sb][i] = rstd * A.in_gamma[4 * q4 + i];
                ib[sb][i] = __builtin_fmaf(-mean, ia[sb][i], A.in_beta[4 * q4 + i]);
            }
        }
    }
    GnAcc st[2][MODE == 1 ? 2 : 1];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) st[sb][k].init();

    const int NS = A.n_steps;
    int g0, g1;
    split_range<64>(g0, g1);
    int si = gridDim.y > 1 ? A.grp_start[g0] : 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    // step entry: x = centre input row base (id*8+oh)*8, y = kd, w bits 8..10 = valid kh mask
    f32x4 xn[3][2][2];   // [kh][sub-tile][half row]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int r = max(0, min((e.x >> 3) + (kh - 1), 63));
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            xn[kh][sb][0] = ldrow(r, sb, 0);
            xn[kh][sb][1] = ldrow(r, sb, 1);
        }
    }
    for (int row = g0; row < g1; ++row) {
        f32x4 acc[8][2];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow][0] = acc[ow][1] = (f32x4){0, 0, 0, 0};
        bool last;
        do {
            f32x4 xc[3][2][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) xc[kh][sb][0] = xn[kh][sb][0], xc[kh][sb][1] = xn[kh][sb][1];   // first use: waits for this step's prefetch
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {  // unconditional, clamped: rows outside the leaf are never used (mask)
                const int r = max(0, min((en.x >> 3) + (kh - 1), 63));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    if (ABL & 1) continue;
                    xn[kh][sb][0] = ldrow(r, sb, 0);
                    xn[kh][sb][1] = ldrow(r, sb, 1);
                }
            }
            // e.y = kd selects the weight registers; a 3-way uniform select keeps the register index static
            const float w0 = e.y == 0 ? w[0] : (e.y == 1 ? w[3] : w[6]);
            const float w1 = e.y == 0 ? w[1] : (e.y == 1 ? w[4] : w[7]);
            const float w2 = e.y == 0 ? w[2] : (e.y == 1 ? w[5] : w[8]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if ((e.w >> (8 + kh)) & 1) {
                    const float wv = kh == 0 ? w0 : (kh == 1 ? w1 : w2);
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) {
                        if (ABL & 2) {
                            acc[ow][0].x += wv + xc[kh][0][ow >> 2][ow & 3], acc[ow][1].x += wv + xc[kh][1][ow >> 2][ow & 3];
                            continue;
                        }
                        acc[ow][0] = mfma16(wv, xc[kh][0][ow >> 2][ow & 3], acc[ow][0]);
                        acc[ow][1] = mfma16(wv, xc[kh][1][ow >> 2][ow & 3], acc[ow][1]);
                    }
                }
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
#pragma unroll
        for (int ow = 0; ow < 8; ++ow)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                f32x4 v = acc[ow][sb] + bias4;
                if (MODE == 2) {  // plain conv output + statistics partials (position-split path)
                    buf_st16(v, outb, lane_o, (unsigned)((row * 8 + ow) * 128 + 16 * sb) * 16u);
                    st[sb][0].add(v.x);
                    st[sb][0].add(v.y);
                    st[sb][0].add(v.z);
                    st[sb][0].add(v.w);
                } else if (MODE == 0) {
                    if (has_out) buf_st16(v, outb, lane_o, (unsigned)((row * 8 + ow) * 128 + 16 * sb) * 16u);   // debug only
                    if (ABL & 4) {
                        st[sb][0].bs += (double)(v.x + v.y + v.z + v.w);
                        continue;
                    }
                    st[sb][0].add(v.x);
                    st[sb][0].add(v.y);
                    st[sb][0].add(v.z);
                    st[sb][0].add(v.w);
                } else {
                    v = gn_relu4(v, ia[sb], ib[sb]);   // (packed fmas: vq_device.h)
                    // streaming (nontemporal) store: the 2.1 GB of a1 are read by the next kernel from HBM anyway, and a plain store's
                    // write-allocate traffic through L2 made this pass store-bound (0.65 -> 0.45 ms)
                    if (!(ABL & 8)) buf_st16_nt(v, outb, lane_o, (unsigned)((row * 8 + ow) * 128 + 16 * sb) * 16u);
                    if (ABL & 4) {
                        st[sb][0].bs += (double)(v.x + v.y), st[sb][1].bs += (double)(v.z + v.w);
                        continue;
                    }
                    st[sb][0].add(v.x);
                    st[sb][0].add(v.y);
                    st[sb][1].add(v.z);
                    st[sb][1].add(v.w);
                }
            }
        if ((row & 3) == 3) {   // 4 rows = 32 positions = one statistics block; slot = GroupNorm group (4 of 4 channels / 8 of 2)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k)
                    st[sb][k].fold_store(A.part_s, A.part_q, part_index(tile, row >> 2, MODE == 1 ? 2 * q4 + k : q4, 16 * sb + jj));
        }
    }
    if (MODE == 2 || A.part_s) return;   // split launch: gn_combine_k finishes
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        if (MODE == 0) {
            float m, r;
            gn_finish(st[sb][0].s, st[sb][0].q, 1.0 / 2048.0, m, r);
            A.out_mean[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj] = r;
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float m, r;
                gn_finish(st[sb][k].s, st[sb][k].q, 1.0 / 1024.0, m, r);
                A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
                A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Elementwise y = relu(GroupNorm_GIN(x)) + statistics of y for the next GroupNorm(8, C).
// Used after both stems (VQVAE_v2.py:236-237 -> ResidualBlock.gn1 :205; :258-259 -> :205).
// Lane (leaf j, half h) owns channels [h*C/2, (h+1)*C/2).
// ------------------------------------------------------------------------------------------
template <int C, int NP, int GIN>
__global__ __launch_bounds__(256) void gn_relu_stats_k(ConvArgs A)
{
    constexpr int NG = C / 4, NGL = NG / 2, CPGI = C / GIN, CPGO = C / 8;
    static_assert(CPGO == 2 || CPGO == 8, "output groups of 2 or 8 channels");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= A.n_tiles) return;
    const int j = lane & 31, h = lane >> 5;
    float ia[NGL][4], ib[NGL][4];
#pragma unroll
    for (int gl = 0; gl < NGL; ++gl)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (h * NGL + gl) * 4 + i;
            const int g = c / CPGI;
            const float mean = A.in_mean[((size_t)tile * GIN + g) * 32 + j];
            const float rstd = A.in_rstd[((size_t)tile * GIN + g) * 32 + j];
            ia[gl][i] = rstd * A.in_gamma[c];
            ib[gl][i] = __builtin_fmaf(-mean, ia[gl][i], A.in_beta[c]);
        }
    constexpr int NACC = (CPGO == 2) ? NGL * 2 : NGL;
    GnAcc st[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) st[k].init();
    const f32x4* in4 = (const f32x4*)A.in + (size_t)tile * NP * NG * 32 + (size_t)h * NGL * 32 + j;
    f32x4* out4 = (f32x4*)A.out + (size_t)tile * NP * NG * 32 + (size_t)h * NGL * 32 + j;
    int b0, b1;   // statistics blocks of this workgroup (all 16 unless the launch is position-split)
    split_range<16>(b0, b1);
    auto close_block = [&](int blk) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const int slot = CPGO == 2 ? h * NGL * 2 + k : h * NGL + k;   // group of 2 channels, or 4-channel partial of a group of 8
            st[k].fold_store(A.part_s, A.part_q, part_index(tile, blk, slot, j));
        }
    };
#pragma unroll 2
    for (int p = b0 * (NP / 16); p < b1 * (NP / 16); ++p) {
        if (p > b0 * (NP / 16) && p % (NP / 16) == 0) close_block(p / (NP / 16) - 1);   // statistics block boundary
#pragma unroll
        for (int gl = 0; gl < NGL; ++gl) {
            f32x4 v = in4[((size_t)p * NG + gl) * 32];
            f32x4 y;
            y.x = fmaxf(__builtin_fmaf(v.x, ia[gl][0], ib[gl][0]), 0.0f);
            y.y = fmaxf(__builtin_fmaf(v.y, ia[gl][1], ib[gl][1]), 0.0f);
            y.z = fmaxf(__builtin_fmaf(v.z, ia[gl][2], ib[gl][2]), 0.0f);
            y.w = fmaxf(__builtin_fmaf(v.w, ia[gl][3], ib[gl][3]), 0.0f);
            out4[((size_t)p * NG + gl) * 32] = y;
            if (CPGO == 2) {
                st[gl * 2 + 0].add(y.x);
                st[gl * 2 + 0].add(y.y);
                st[gl * 2 + 1].add(y.z);
                st[gl * 2 + 1].add(y.w);
            } else {
                st[gl].add(y.x);
                st[gl].add(y.y);
                st[gl].add(y.z);
                st[gl].add(y.w);
            }
        }
    }
    close_block(b1 - 1);
    if (A.part_s) return;   // split launch: gn_combine_k finishes
    const double inv_n = 1.0 / (double)(CPGO * NP);
    if (CPGO == 2) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            float m, r;
            gn_finish(st[k].s, st[k].q, inv_n, m, r);
            const int g = (h * NGL) * 2 + k;
            A.out_mean[((size_t)tile * 8 + g) * 32 + j] = m;
            A.out_rstd[((size_t)tile * 8 + g) * 32 + j] = r;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NACC / 2; ++k) {
            float m, r;
            gn_finish(st[2 * k].s + st[2 * k + 1].s, st[2 * k].q + st[2 * k + 1].q, inv_n, m, r);
            const int g = (h * NGL) / 2 + k;
            A.out_mean[((size_t)tile * 8 + g) * 32 + j] = m;
            A.out_rstd[((size_t)tile * 8 + g) * 32 + j] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// E4/E6: Conv3d(16->16,k3,p1) @8^3 inside ResidualBlock(16) (VQVAE_v2.py:190-210, :238).
// 16x16x4 MFMA (rows 16 couts, cols 16 leaves, K = 4 channels/step, order P16), weights (27 KB) resident in LDS.
// One wave owns a 16-leaf half tile and NR (2 or 4) adjacent output rows of 8 positions each;
// one step = (row group, kd, input row ih in [oh0-1, oh0+NR]).  An input row feeds output row A = oh0 with
// kh = ih-oh0+1 and row B = oh0+1 with kh = ih-oh0 (whichever are valid), so 12 row loads serve two
// output rows instead of 18, and the table has 1/3 fewer steps.  The input row lives in a single rolling
// register buffer: transformed in place at the start of a step, each position re-loaded for the next step
// right after its last use.  Per output, taps still arrive in ascending (kd,kh,kw) order -> same
// arithmetic as a plain tap-by-tap evaluation.  conv2 fuses the residual `x + 0.1*y`; conv1 emits the
// GroupNorm(8,16) statistics of its output.
// ------------------------------------------------------------------------------------------
template <int NR, bool RESID, bool STATS, bool RAWIN = false>
__global__ __launch_bounds__(256, 2) void conv8_c16_k(ConvArgs A, const int4* __restrict__ steps)
{
    static_assert(NR == 2 || NR == 4, "output rows per wave");
    __shared__ f32x4 wl[27 * 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) wl[i] = ((const f32x4*)A.wfrag)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = blockIdx.x * 4 + wave;
    const int tile = half >> 1;
    if (tile >= A.n_tiles) return;
    const int jj = (lane & 15) + 16 * (half & 1), q4 = lane >> 4;
    float ia[4], ib[4];
    if (!RAWIN) {  // RAWIN: plain convolution of the input (data-gradient launches)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * q4 + i, g = c >> 1;
            const float mean = A.in_mean[((size_t)tile * 8 + g) * 32 + jj];
            const float rstd = A.in_rstd[((size_t)tile * 8 + g) * 32 + jj];
            ia[i] = rstd * A.in_gamma[c];
            ib[i] = __builtin_fmaf(-mean, ia[i], A.in_beta[c]);
        }
    }
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    // buffer addressing (see buf_ld16): position p of this lane's channel quad
    const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * 512 * 4 * 32);
    const vq_buf outb = buf_of((const f32x4*)A.out + (size_t)tile * 512 * 4 * 32);
    const vq_buf skb = buf_of(RESID ? (const f32x4*)A.skip + (size_t)tile * 512 * 4 * 32 : (const f32x4*)A.out);
    const unsigned lane_b = (unsigned)(q4 * 32 + jj) * 16u;
    GnAcc st[2];
    st[0].init();
    st[1].init();
    const int NS = A.n_steps;
    int g0, g1;
    split_range<64 / NR>(g0, g1);
    int si = gridDim.y > 1 ? A.grp_start[g0] : 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    f32x4 xr[8];
#pragma unroll
    for (int iw = 0; iw < 8; ++iw) xr[iw] = buf_ld16(inb, lane_b, (unsigned)(e.x + iw) * 2048u);
    for (int grp = g0; grp < g1; ++grp) {  // 8 od x (8/NR) row groups
        f32x4 acc[NR][8];
#pragma unroll
        for (int rw = 0; rw < NR; ++rw)
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) acc[rw][ow] = (f32x4){0, 0, 0, 0};
        const int obase = e.z;  // (od*8 + oh0)*8 of this group, from the schedule (the table decides the order of the groups)
        bool last;
        do {
#pragma unroll
            for (int iw = 0; iw < 8 && !RAWIN; ++iw) {  // first use of the rolling buffer: waits for the loads of the previous step
                f32x4 v = xr[iw];
                v.x = fmaxf(__builtin_fmaf(v.x, ia[0], ib[0]), 0.0f);
                v.y = fmaxf(__builtin_fmaf(v.y, ia[1], ib[1]), 0.0f);
                v.z = fmaxf(__builtin_fmaf(v.z, ia[2], ib[2]), 0.0f);
                v.w = fmaxf(__builtin_fmaf(v.w, ia[3], ib[3]), 0.0f);
                xr[iw] = v;
            }
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
            // e.y = kd*3; bits 8+4*rw .. : kh+1 of output row rw of the group (0 = this input row does not feed it)
            int kh[NR];
            f32x4 w[NR][3];
#pragma unroll
            for (int rw = 0; rw < NR; ++rw) {
                kh[rw] = ((e.w >> (8 + 4 * rw)) & 15) - 1;
                const f32x4* wp = wl + (e.y + (kh[rw] < 0 ? 0 : kh[rw])) * 3 * 64 + lane;
                w[rw][0] = wp[0], w[rw][1] = wp[64], w[rw][2] = wp[128];
            }
#pragma unroll
            for (int iw = 0; iw < 8; ++iw) {
#pragma unroll
                for (int rw = 0; rw < NR; ++rw) {
                    if (kh[rw] >= 0) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int ow = iw - kw + 1;
                            if (ow < 0 || ow > 7) continue;
                            acc[rw][ow] = mfma16(w[rw][kw].x, xr[iw].x, acc[rw][ow]);
                            acc[rw][ow] = mfma16(w[rw][kw].y, xr[iw].y, acc[rw][ow]);
                            acc[rw][ow] = mfma16(w[rw][kw].z, xr[iw].z, acc[rw][ow]);
                            acc[rw][ow] = mfma16(w[rw][kw].w, xr[iw].w, acc[rw][ow]);
                        }
                    }
                }
                xr[iw] = buf_ld16(inb, lane_b, (unsigned)(en.x + iw) * 2048u);  // next step's row (table index clamped)
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
        // epilogue: the NR rows in order (positions ascending)
#pragma unroll
        for (int rw = 0; rw < NR; ++rw) {
            f32x4 sk[RESID ? 8 : 1];
            if (RESID) {
#pragma unroll
                for (int ow = 0; ow < 8; ++ow) sk[ow] = buf_ld16(skb, lane_b, (unsigned)(obase + rw * 8 + ow) * 2048u);
            }
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                f32x4 v = acc[rw][ow] + bias4;
                if (RESID) {
                    const f32x4 u = v * 0.1f;
                    v = sk[ow] + u;
                }
                buf_st16(v, outb, lane_b, (unsigned)(obase + rw * 8 + ow) * 2048u);
                if (STATS) {
                    st[0].add(v.x);
                    st[0].add(v.y);
                    st[1].add(v.z);
                    st[1].add(v.w);
                    if ((ow & 3) == 3) {   // half an output row closed = one statistics block of this tensor (128 half-row blocks per leaf)
                        const int hrow = 2 * ((obase >> 3) + rw) + (ow >> 2);
                        st[0].fold_store(A.part_s, A.part_q, part_index_rows(tile, hrow, 2 * q4 + 0, jj));   // slot = GroupNorm(8,16) group
                        st[1].fold_store(A.part_s, A.part_q, part_index_rows(tile, hrow, 2 * q4 + 1, jj));
                    }
                }
            }
        }
    }
    // (the half-row blocks of this tensor are added row-major, not in the order a wave produces them: gn_combine_k<false, true> finishes;
    // every STATS launch passes the partial buffers)
}

// ------------------------------------------------------------------------------------------
// Generic leaf-tile conv on the 32x32x2 fp32 MFMA: rows = 32 couts, cols = 32 leaves, K order P8.
// One wave owns one tile and walks a host-built, flattened schedule of (output position, valid
// tap) steps.  Per step it consumes CIN/8 float4 of activations per lane and, per 32-cout tile,
// one ds_read_b128 of weights per 4 MFMAs.  Software pipeline per step:
//     wait(prefetched B(s), W(s)) -> [barrier] -> issue B(s+1) loads (+ async global->LDS copy of
//     W(s+1) into the other LDS buffer) -> NU*NMT*4 MFMAs on B(s) with the LDS A-fragment read
//     one group ahead -> epilogue on the last tap of a position.
// Weights are either fully LDS-resident (STREAM=false: encoder 4^3 layers, <=128 KB) or streamed
// one tap per step through a double-buffered LDS window, all waves of the workgroup in lock-step
// (STREAM=true: decoder layers, 0.4-1.8 MB of weights).
//   INMODE 0: raw input      1: relu(GroupNorm(GIN)) on load      2: squeeze-excite gate on load
//   RESID  : out = skip + 0.1*(acc+bias)        GOUT: GroupNorm statistics of the output
//   CSUM   : per-channel sums of the output (feeds ChannelAttention of the next kernel)
//   OUTMODE 0: store the leaf-tile activation.  2: folded decoder tail — the "positions" are the four
//            128-voxel output slabs, the "taps" the input positions of the slab's receptive field, the
//            rows of the weight fragments output VOXELS; epilogue = per-voxel bias, sigmoid, store into
//            the caller's leaf-major [n][512] buffer (VQVAECodec.cpp:182-192).
// ------------------------------------------------------------------------------------------
// Folded decoder tail: does output depth plane od (0..7 at 8^3) depend on input depth plane pd (0..3 at 4^3)?  final conv taps reach
// z = od-1 .. od+1 (inside 0..7), z lies in coarse cell z/2, the up conv's taps reach cell-1 .. cell+1 (inside 0..3).
__device__ __forceinline__ int vq_tail_pd_lo(int od) { const int c = (od > 0 ? od - 1 : 0) >> 1; return c > 0 ? c - 1 : 0; }
__device__ __forceinline__ int vq_tail_pd_hi(int od) { const int c = (od < 7 ? od + 1 : 7) >> 1; return c < 3 ? c + 1 : 3; }
__device__ __forceinline__ bool vq_tail_plane_needs(int od, int pd) { return pd >= vq_tail_pd_lo(od) && pd <= vq_tail_pd_hi(od); }

// WMODE: how the next step's weight pieces travel global -> register -> LDS.  0 (the library): one piece per few groups, requested at
// the group's start and stored after its MFMAs.  1: all of the step's pieces requested at the step's start, before the step's
// activation re-loads, and stored at its end (NPW x 4 registers more) — built on the suspicion that the in-order return of
// vector-memory operations made every store wait for an activation re-load behind an L2 miss; the waits turned out to be exact
// counts that leave the younger re-loads in flight either way, and the two modes time the same (profiles/r04_ablate_folded_tail.txt).
// ABL (tools/ablate/conv_mfma32_ablate.hip only): 1 no barriers, 2 no weight streaming, 4 no LDS A-fragment reads, 8 no activation
// re-loads, 16 no input transform, 32 no epilogue, 64 no row-blocked totals, 128 no MFMAs
template <int CIN, int COUT, int NPI, int NPO, int NW, bool STREAM, int KWG, int INMODE, int GIN, bool RESID, int GOUT,
          bool CSUM, int OUTMODE, int ABL = 0, int WMODE = 0>
__global__ __launch_bounds__(NW * 64, 2) void conv_mfma32_k(ConvArgs A, const int4* __restrict__ steps)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* lds = (f32x4*)smem_raw;
    constexpr int NU = CIN / 8, NMT = COUT / 32, NK = NU * NMT;
    static_assert(STREAM, "every instantiation streams its weights (the LDS-resident variant went with the encoder's 4^3 layers)");
    static_assert(GOUT == 0 && !CSUM, "the fused statistics of this kernel predate the 16-block contract (no instantiation uses them)");
    // One step = KWG consecutive taps (input positions e.x .. e.x + KWG - 1, fragments e.y .. e.y + KWG - 1) = NGR groups of one
    // channel octet x NMT cout tiles.  The next step's WSTEP float4 of weights travel global -> register -> LDS in 1 KiB pieces, one
    // piece per wave every few groups (requested at the group's start, stored after its MFMAs: see REGW in conv_rows16_k for why not
    // global_load_lds), into the other half of a double-buffered window; one barrier per step.
    constexpr int WTAP = NK * 64, WSTEP = KWG * WTAP;   // float4 per tap / per step
    constexpr int NGR = KWG * NU;                       // MFMA groups per step
    constexpr int NPW = WSTEP / 64 / NW;                // weight pieces per wave and step
    static_assert(WSTEP % (NW * 64) == 0, "streamed taps split evenly over the waves");
    constexpr int PPG = (NPW + NGR - 1) / NGR;          // most pieces any group carries
    auto piece_first = [](int g) { return (g * NPW + NGR - 1) / NGR; };   // piece k travels with group floor(k * NGR / NPW)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, q = lane >> 5;
    int tile = blockIdx.x * NW + wave;
    const bool active = tile < A.n_tiles;
    if (!active) tile = A.n_tiles - 1;   // a wave without a tile re-computes the last one (it carries its share of the weights)

    // ---- input transform, per lane: channels cin = 8u + 4q + i ----
    float ta[INMODE == 0 ? 1 : NU][4], tb[INMODE == 1 ? NU : 1][4];
    if (INMODE == 1) {
        constexpr int CPGI = CIN / (GIN > 0 ? GIN : 1);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 8 * u + 4 * q + i, g = c / CPGI;
                const float mean = A.in_mean[((size_t)tile * GIN + g) * 32 + j];
                const float rstd = A.in_rstd[((size_t)tile * GIN + g) * 32 + j];
                ta[u][i] = rstd * A.in_gamma[c];
                tb[u][i] = __builtin_fmaf(-mean, ta[u][i], A.in_beta[c]);
            }
    } else if (INMODE == 2) {
        float hid[CIN / 4], gall[CIN];
        se_hidden<CIN>(A.se_csum + (size_t)tile * CIN * 32 + j, A.se_fc0, hid);
        se_gates<CIN>(hid, A.se_fc2, gall);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) ta[u][i] = q ? gall[8 * u + 4 + i] : gall[8 * u + i];
    }

    // activations and weights through buffer addressing (see buf_ld16): position ip, channel octet u of this lane; weight piece t of tap0
    const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * NPI * (CIN / 4) * 32);
    const unsigned lane_b = (unsigned)(q * 32 + j) * 16u;
    auto ldx = [&](int ip, int u) -> f32x4 { return buf_ld16(inb, lane_b + u * 1024, (unsigned)(ip < NPI ? ip : NPI - 1) * (CIN / 4) * 512u); };
    const vq_buf wb = buf_of(A.wfrag);
    auto ldw = [&](int tap0, int k) -> f32x4 { return buf_ld16(wb, (unsigned)lane * 16u, ((unsigned)tap0 * WTAP + (unsigned)(k * NW + wave) * 64u) * 16u); };
    const f32x4* bf4 = (const f32x4*)A.bias_frag;
    const int NS = A.n_steps;
    // outputs: the leaf-tile activation (OUTMODE 0) or the tile's 32 leaves of the caller's leaf-major [n][512] buffer (OUTMODE 2)
    const vq_buf outb = buf_of(OUTMODE == 2 ? (const void*)(A.out + (size_t)tile * 32 * 512) : (const void*)((const f32x4*)A.out + (size_t)tile * NPO * (COUT / 4) * 32));
    const vq_buf skb = buf_of(RESID ? (const void*)((const f32x4*)A.skip + (size_t)tile * NPO * (COUT / 4) * 32) : (const void*)A.out);

    // ---- prologue: the first step's weights into window si & 1, its positions into registers ----
    int g0, g1;
    split_range<NPO>(g0, g1);
    int si = gridDim.y > 1 ? A.grp_start[g0] : 0;
    int4 e = steps[si];
    int4 en = steps[si + 1 < NS ? si + 1 : NS - 1];
#pragma unroll
    for (int k = 0; k < NPW; ++k) lds[(si & 1) * WSTEP + (k * NW + wave) * 64 + lane] = ldw(e.y, k);
    f32x4 bn[KWG][NU];
#pragma unroll
    for (int kg = 0; kg < KWG; ++kg)
#pragma unroll
        for (int u = 0; u < NU; ++u) bn[kg][u] = ldx(e.x + kg, u);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // enter the loop with nothing in flight (see conv_rows16_k)
    for (int po = g0; po < g1; ++po) {
        f32x16 acc[NMT];
        // OUTMODE 2 (folded tail): ROW-BLOCKED accumulation — the four input positions of a W-row (e.x = 4r .. 4r+3) are one fmaf
        // chain from zero, the row sums are added in row order (oracle tail_apply_ex).  One chain over all 3072-4096 terms was 12x less
        // accurate on a trained checkpoint (pre-activations of +-20), tests/test_golden_regimes.py.
        f32x16 tot[OUTMODE == 2 ? NMT : 1];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        if (OUTMODE == 2) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[mt][r] = 0.0f;
        }
        bool last;
        do {
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];   // consumed one step later
            if (!(ABL & 1)) __syncthreads();   // every wave's pieces of W(s) are in LDS; every wave done reading W(s-1)
            const f32x4* wl = lds + (si & 1) * WSTEP + lane;
            // OUTMODE 2 (folded tail): cout tiles 0,1 are the voxels of depth od = 2 po, tiles 2,3 those of od = 2 po + 1, and a voxel plane
            // only depends on the input planes pd in [pd_lo(od), pd_hi(od)] (two levels of 3-tap receptive fields: od +-1 -> coarse cell
            // -> +-1): 64 of the 224 steps feed one of the two planes only; the other plane's composite weights are structurally zero
            // there and its MFMAs are skipped (a seventh of the kernel's matrix work).
            // input plane pd = e.x / 16 against the two output planes of slab po (wave-uniform): which tile pairs this step feeds
            bool act_lo = true, act_hi = true;
            if (OUTMODE == 2 && NMT == 4) {
                const int pd = e.x >> 4;
                act_lo = vq_tail_plane_needs(2 * po, pd), act_hi = vq_tail_plane_needs(2 * po + 1, pd);
            }
            f32x4 a_nx = wl[0];
            f32x4 wstep[WMODE == 1 ? NPW : 1];
            if (WMODE == 1 && !(ABL & 2)) {
#pragma unroll
                for (int k = 0; k < NPW; ++k) wstep[k] = ldw(en.y, k);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < NGR; ++g) {
                const int kg = g / NU, u = g % NU;
                f32x4 wnext[PPG];
#pragma unroll
                for (int jp = 0; jp < PPG; ++jp)
                    if (WMODE == 0 && !(ABL & 2) && piece_first(g) + jp < piece_first(g + 1)) wnext[jp] = ldw(en.y, piece_first(g) + jp);
                __builtin_amdgcn_sched_barrier(0);   // requested here, a whole group ahead of the ds_write
                f32x4 b = bn[kg][u];
                if (ABL & 16) {
                } else if (INMODE == 1) {
                    b.x = fmaxf(__builtin_fmaf(b.x, ta[u][0], tb[u][0]), 0.0f);
                    b.y = fmaxf(__builtin_fmaf(b.y, ta[u][1], tb[u][1]), 0.0f);
                    b.z = fmaxf(__builtin_fmaf(b.z, ta[u][2], tb[u][2]), 0.0f);
                    b.w = fmaxf(__builtin_fmaf(b.w, ta[u][3], tb[u][3]), 0.0f);
                } else if (INMODE == 2) {
                    b = b * (f32x4){ta[u][0], ta[u][1], ta[u][2], ta[u][3]};   // (two packed multiplies where the gates sit in register pairs)
                }
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const f32x4 a = a_nx;
                    if (!(ABL & 4) && g * NMT + mt + 1 < NGR * NMT) a_nx = wl[(g * NMT + mt + 1) * 64];   // LDS A fragment one group ahead of its MFMAs
                    if (OUTMODE == 2 && NMT == 4 && !(mt < NMT / 2 ? act_lo : act_hi)) continue;   // (wave-uniform) structurally zero weights
                    if (ABL & 128) {
                        acc[mt][0] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
                        continue;
                    }
                    acc[mt] = mfma32(a.x, b.x, acc[mt]);
                    acc[mt] = mfma32(a.y, b.y, acc[mt]);
                    acc[mt] = mfma32(a.z, b.z, acc[mt]);
                    acc[mt] = mfma32(a.w, b.w, acc[mt]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 8)) bn[kg][u] = ldx(en.x + kg, u);   // the next step's position, right after this one's last use (index clamped)
#pragma unroll
                for (int jp = 0; jp < PPG; ++jp)
                    if (WMODE == 0 && !(ABL & 2) && piece_first(g) + jp < piece_first(g + 1))
                        lds[((si + 1) & 1) * WSTEP + ((piece_first(g) + jp) * NW + wave) * 64 + lane] = wnext[jp];
            }
            if (WMODE == 1 && !(ABL & 2)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NPW; ++k) lds[((si + 1) & 1) * WSTEP + (k * NW + wave) * 64 + lane] = wstep[k];
            }
            if (OUTMODE == 2 && !(ABL & 64) && (e.x & 3) == 3) {   // row complete (a slab's positions start and end on row boundaries)
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[mt][r] = tot[mt][r] + acc[mt][r], acc[mt][r] = 0.0f;
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
        if (OUTMODE == 2 && !(ABL & 64)) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = tot[mt];
        }
        if (ABL & 32) {
            float t = 0.0f;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) t += acc[mt][0] + acc[mt][15];
            if (t == 12345.678f) A.out[threadIdx.x] = t;
            continue;
        }

        // ---- epilogue for output position po ----
        f32x4 skv[RESID ? NMT : 1][4];
        if (RESID) {  // issue all residual loads before any of the epilogue math
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) skv[mt][g] = buf_ld16(skb, lane_b + (8 * mt + 2 * g) * 512, (unsigned)po * (COUT / 4) * 512u);
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = bf4[((OUTMODE == 2 ? po * NMT : 0) + mt) * 8 + q * 4 + g];
                f32x4 v;
                v.x = acc[mt][4 * g + 0] + bias.x;
                v.y = acc[mt][4 * g + 1] + bias.y;
                v.z = acc[mt][4 * g + 2] + bias.z;
                v.w = acc[mt][4 * g + 3] + bias.w;
                // regs 4g..4g+3 of half q hold couts 32mt + 8g + 4q + {0..3}: L4 group 8mt+2g+q
                if (RESID) {
                    const f32x4 u = v * 0.1f;
                    v = skv[mt][g] + u;
                }
                if (OUTMODE == 0 && active) buf_st16(v, outb, lane_b + (8 * mt + 2 * g) * 512, (unsigned)po * (COUT / 4) * 512u);
                if (OUTMODE == 2) {
                    // rows 32mt + 8g + 4q + {0..3} of slab po are 4 consecutive voxels of this lane's leaf
                    const int64_t leaf = (int64_t)tile * 32 + j;
                    if (active && leaf < A.n_leaves) {
                        f32x4 sg;
                        sg.x = vq_sigmoid(v.x), sg.y = vq_sigmoid(v.y), sg.z = vq_sigmoid(v.z), sg.w = vq_sigmoid(v.w);
                        buf_st16(sg, outb, (unsigned)(j * 512 + 4 * q) * 4u, (unsigned)(po * 128 + 32 * mt + 8 * g) * 4u);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Folded decoder tail for small batches (position-split path): one wave = (tile, output slab d, block mb of 32 voxels), the
// four blocks of a slab in one workgroup (their activation loads hit the same lines).  Same fragments (tail.w, tail.b) and the
// same accumulation order per output as conv_mfma32_k<OUTMODE 2>: input positions of the slab's receptive field ascending,
// channels in P8 order.  A lone wave has nobody to hide its load latency behind, so there is no LDS window and no barrier:
// weights and activations of two consecutive positions live in two register sets, each refilled for position p+2 right after
// its last MFMA has issued (64-leaf decode: the LDS-window variant spent 3 us per 0.85 us of MFMAs).
// ------------------------------------------------------------------------------------------
// TILEOUT (training forward): the sigmoid output goes to the position-major tile layout [tile][512][32] instead of the caller's [n][512].
template <bool TILEOUT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void tail_small_k(ConvArgs A)   // 2 waves/SIMD: room for both register sets
{
    constexpr int CIN = 64, NU = 8, NMT = 4, NPI = 64;
    const int lane = threadIdx.x & 63;
    const int mb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, q = lane >> 5;
    const int tile = blockIdx.x, d = blockIdx.y;
    // the slab's fragments cover input planes [d-2, d+2] (first one: ps0); this wave's 32 voxels lie in depth plane od = 2d + mb/2 and
    // depend on planes pd_lo(od) .. pd_hi(od) only (the rest of the slab's composite weights are structurally zero for them)
    const int od = 2 * d + (mb >> 1);
    const int ps0 = (d > 2 ? d - 2 : 0) * 16;
    const int p0 = vq_tail_pd_lo(od) * 16, p1 = (vq_tail_pd_hi(od) + 1) * 16;             // an even number of positions (32, 48 or 64)
    const int sbase = (d == 0 ? 0 : d == 1 ? 48 : d == 2 ? 112 : 176) - ps0;              // step of position 0 in tail.w for slab d
    float ta[NU][4];   // attention gates of this lane's channels 8u + 4q + i (precomputed once per tile by csum_seq_k)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) ta[u][i] = A.se_gate[((size_t)tile * CIN + 8 * u + 4 * q + i) * 32 + j];
    const f32x4* in4 = (const f32x4*)A.in + (size_t)tile * NPI * (CIN / 4) * 32 + q * 32 + j;       // + p*(CIN/4)*32 + u*64
    const f32x4* w4 = (const f32x4*)A.wfrag + ((ptrdiff_t)sbase * NU * NMT + mb) * 64 + lane;   // + (p*NU*NMT + u*NMT)*64
    f32x4 w[2][NU], b[2][NU];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            w[h][u] = w4[((size_t)(p0 + h) * NU + u) * NMT * 64];
            b[h][u] = in4[(size_t)(p0 + h) * (CIN / 4) * 32 + u * 64];
        }
    // row-blocked accumulation like conv_mfma32_k<OUTMODE 2>: the four positions of a W-row are one chain from zero (acc), the row
    // sums are added in row order (tot)
    f32x16 acc, tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f, tot[r] = 0.0f;
    for (int p = p0; p < p1; p += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int pn = p + h + 2 < p1 ? p + h + 2 : p + h;   // the last pair re-requests valid data
            asm volatile("" : "+s"(pn));   // opaque: otherwise the optimiser replaces the loop-carried prefetch by a load at the point of use
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const f32x4 a = w[h][u];
                f32x4 x = b[h][u];
                x.x = x.x * ta[u][0];
                x.y = x.y * ta[u][1];
                x.z = x.z * ta[u][2];
                x.w = x.w * ta[u][3];
                acc = mfma32(a.x, x.x, acc);
                acc = mfma32(a.y, x.y, acc);
                acc = mfma32(a.z, x.z, acc);
                acc = mfma32(a.w, x.w, acc);
                w[h][u] = w4[((size_t)pn * NU + u) * NMT * 64];
                b[h][u] = in4[(size_t)pn * (CIN / 4) * 32 + u * 64];
                __builtin_amdgcn_sched_barrier(0);   // issue the refill here, not where the scheduler would like it (next to its use)
            }
        }
        if (p & 2) {   // second pair of a row (p0 is a multiple of 16)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[r] = tot[r] + acc[r], acc[r] = 0.0f;
        }
    }
    acc = tot;
    const f32x4* bf4 = (const f32x4*)A.bias_frag;
    const int64_t leaf = (int64_t)tile * 32 + j;
    if (!TILEOUT && leaf >= A.n_leaves) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bias = bf4[(d * NMT + mb) * 8 + q * 4 + g];
        f32x4 sg;
        sg.x = vq_sigmoid(acc[4 * g + 0] + bias.x);
        sg.y = vq_sigmoid(acc[4 * g + 1] + bias.y);
        sg.z = vq_sigmoid(acc[4 * g + 2] + bias.z);
        sg.w = vq_sigmoid(acc[4 * g + 3] + bias.w);
        if (TILEOUT) {
            float* o = A.out + ((size_t)tile * 512 + d * 128 + 32 * mb + 8 * g + 4 * q) * 32 + j;
            o[0] = sg.x, o[32] = sg.y, o[64] = sg.z, o[96] = sg.w;
        } else {
            *(f32x4*)(A.out + leaf * 512 + d * 128 + 32 * mb + 8 * g + 4 * q) = sg;   // 4 consecutive voxels of this lane's leaf
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same folded tail on the 16x16x4 MFMA for the SMALLEST batches (a few tiles: SOP-sized calls): one wave = (tile, output slab d,
// block of 16 voxels, 16-leaf half tile), sixteen waves per (tile, slab) instead of four.  Per output the arithmetic is the
// 32x32x2 kernels' exactly — input positions ascending, channels in P8 order (8u, 8u+4, 8u+1, 8u+5 | 8u+2, 8u+6, 8u+3, 8u+7: two
// MFMAs of four k steps per octet instead of four of two) — but a wave's serial chain is a quarter as long (K = 4 per 32-cycle
// MFMA instead of K = 2 per 64), which is all that counts when a launch has fewer waves than the chip has SIMDs
// (64-leaf decode: 90 -> ~30 us for this kernel).  Fragments: tail.w16[((step*8 + mb)*4 + uu)*64 + lane][e], octet u = 2uu + (e>>1),
// MFMA mf = e&1, = Wc[voxel 16mb + (lane&15)][pos][8u + 4(k&1) + (k>>1) + 2mf], k = lane>>4; bias: plain per voxel.
// Lane (leaf n = lane&15, k slot q4 = lane>>4) loads the channel quad 8u + 4(q4&1) of its leaf and feeds elements (q4>>1) and
// 2 + (q4>>1) of it.  DEPTH positions of weights and activations in flight per wave (a position is only 16 MFMAs long).
// ------------------------------------------------------------------------------------------
template <int DEPTH = 4>
__global__ __launch_bounds__(256) void tail_small16_k(ConvArgs A)
{
    constexpr int CIN = 64, NPI = 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x, d = blockIdx.y, unit = blockIdx.z * 4 + wave;   // 16 units: 8 voxel blocks x 2 leaf halves
    const int mb = unit >> 1, half = unit & 1;
    const int n = lane & 15, q4 = lane >> 4, jj = 16 * half + n;
    // this wave's 16 voxels lie in depth plane od = 2d + mb/4 and depend on input planes pd_lo(od) .. pd_hi(od) only (see tail_small_k)
    const int od = 2 * d + (mb >> 2);
    const int ps0 = (d > 2 ? d - 2 : 0) * 16;
    const int p0 = vq_tail_pd_lo(od) * 16, p1 = (vq_tail_pd_hi(od) + 1) * 16;             // 32, 48 or 64 positions: multiples of DEPTH
    const int sbase = (d == 0 ? 0 : d == 1 ? 48 : d == 2 ? 112 : 176) - ps0;              // step of position 0 in tail.w16 for slab d
    static_assert(32 % DEPTH == 0 && 48 % DEPTH == 0 && 64 % DEPTH == 0, "whole rings");
    float ta[8][2];   // attention gates of this lane's channels (computed once per tile by csum_combine_k)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) ta[u][mf] = A.se_gate[((size_t)tile * CIN + 8 * u + 4 * (q4 & 1) + (q4 >> 1) + 2 * mf) * 32 + jj];
    const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * NPI * (CIN / 4) * 32);
    const unsigned lane_x = (unsigned)((q4 & 1) * 32 + jj) * 16u;
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;
    f32x4 w[DEPTH][4], b[DEPTH][8];
    auto request = [&](int h, int p) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) w[h][uu] = buf_ld16(wb, lane_w, (unsigned)(((sbase + p) * 8 + mb) * 4 + uu) * 1024u);
#pragma unroll
        for (int u = 0; u < 8; ++u) b[h][u] = buf_ld16(inb, lane_x, (unsigned)(p * (CIN / 4) + 2 * u) * 512u);
    };
#pragma unroll
    for (int h = 0; h < DEPTH; ++h) request(h, p0 + h);
    static_assert(DEPTH == 4, "one ring = one W-row of input positions = one accumulation block");
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, tot = {0.0f, 0.0f, 0.0f, 0.0f};   // row-blocked accumulation (see conv_mfma32_k<OUTMODE 2>)
    const bool hi = (q4 >> 1) != 0;
    for (int p = p0; p < p1; p += DEPTH) {
#pragma unroll
        for (int h = 0; h < DEPTH; ++h) {
            int pn = p + h + DEPTH < p1 ? p + h + DEPTH : p + h;   // the last ring re-requests valid data
            asm volatile("" : "+s"(pn));   // opaque: keeps the loop-carried prefetch a prefetch
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 x = b[h][u], a = w[h][u >> 1];
                const float x0 = (hi ? x.y : x.x) * ta[u][0], x1 = (hi ? x.w : x.z) * ta[u][1];
                acc = mfma16((u & 1) ? a.z : a.x, x0, acc);
                acc = mfma16((u & 1) ? a.w : a.y, x1, acc);
            }
            request(h, pn);
            __builtin_amdgcn_sched_barrier(0);   // issue the refill here, not next to its use
        }
        tot = tot + acc;
        acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    acc = tot;
    const int64_t leaf = (int64_t)tile * 32 + jj;
    if (leaf >= A.n_leaves) return;
    const f32x4 bias = *(const f32x4*)(A.bias_frag + d * 128 + 16 * mb + 4 * q4);
    f32x4 sg;
    sg.x = vq_sigmoid(acc.x + bias.x);
    sg.y = vq_sigmoid(acc.y + bias.y);
    sg.z = vq_sigmoid(acc.z + bias.z);
    sg.w = vq_sigmoid(acc.w + bias.w);
    *(f32x4*)(A.out + leaf * 512 + d * 128 + 16 * mb + 4 * q4) = sg;   // 4 consecutive voxels of this lane's leaf
}

// ------------------------------------------------------------------------------------------
// Row-blocked leaf-tile conv on the 16x16x4 MFMA for every layer with a 4^3 output: the decoder's ResidualBlock(64) convs
// (VQVAE_v2.py:190-210, :260), the encoder's ResidualBlock(32) convs (:240) and the down conv 16->32 k4 s2 (:239).
// A wave owns a 16-leaf HALF tile and one output row of SO = 4 positions (SO x COUT/16 accumulators of 16 couts x 16 leaves);
// one step = (output row, valid (kd,kh)): the SI input positions of the row sit in a rolling register buffer (GroupNorm+ReLU
// applied on arrival), each feeds its (ow,kw) pairs and is re-loaded for the next step right after its last use (k3: 6.25 row
// loads per output row instead of 15.6 position loads per output position).  Weights: LDS-resident where the layer's fragments
// fit (RESIDENT), else the KS kw-taps of the step stream through a double-buffered LDS window shared by the 8 waves (4 tiles) of
// the workgroup, one barrier per step (64->64: 640 MFMAs) — global -> register -> ds_write, one piece per MFMA group (REGW below),
// in the kw-outer instantiations; by global_load_lds in the others (channel-split small-batch variants, training forwards).
// K order inside a tap: 16-channel blocks ascending, "P16" inside a block (0,4,8,12,1,5,...), restated by the oracle for these
// layers.  KWO: kw-outer MFMA order with the LDS fragments read a group ahead.  History of the 64->64 layers: 77 % pipe
// utilisation at 2.13 GHz (one tap per step on the 32x32x2 MFMA, 17 GB fetched per launch) -> 83 % at 2.33 GHz (this kernel, 7 GB)
// -> 91-93 % at 2.38 GHz (buffer addressing, REGW, lazy arrival).
// wfrag[((tap*CBN + cb)*MTN + mt)*64 + lane][i] = W[16mt + (lane&15)][16cb + 4(lane>>4) + i][tap];  bias: plain [COUT].
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int SI, int SO, int KS, int STRIDE, int PAD, int INMODE, bool RESID, int GOUT, bool CSUM, bool RESIDENT = false, int MSPLIT = 1, bool PF2 = false, int NWV = 8, bool PARTS = false, int ABL = 0, bool KWO = false, int OWI = 1>
__global__ __launch_bounds__(NWV * 64, 1) void conv_rows16_k(ConvArgs A, const int4* __restrict__ steps)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    // ABL: timing-only ablations for tools/ablate/conv_rows16_ablate.hip (0 in the library): 1 no barriers, 2 no weight streaming,
    // 4 no LDS A-fragment reads, 8 no activation re-loads, 16 no GroupNorm transform, 32 no epilogue, 64 every tile reads tile 0 (L2 hits)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* lds = (f32x4*)smem_raw;
    constexpr int CBN = CIN / 16, MTN = COUT / 16, NPI = SI * SI * SI, NPO = SO * SO * SO;
    constexpr int WTAP = CBN * MTN * 64, WSTEP = KS * WTAP;   // float4 per tap / per step (KS kw taps)
    // MSPLIT > 1 (small batches): blockIdx.z selects MTL of the MTN 16-cout blocks, so a wave's serial MFMA chain is MSPLIT times
    // shorter; the K order of every output is unchanged.  Only the needed weight pieces are staged, packed densely in LDS.
    constexpr int MTL = MTN / MSPLIT, WTAPL = WTAP / MSPLIT, WSTEPL = WSTEP / MSPLIT;
    // REGW: the next step's weight pieces travel global -> register -> LDS, one piece per MFMA group, instead of by global_load_lds.
    // The async copy looks cheaper (no registers, no ds_write) but costs a stall per step: while one is in flight the compiler answers
    // EVERY vector-memory dependence with s_waitcnt vmcnt(0) (a pending LDS-DMA counts as a flat access that may complete out of order),
    // so the first use of a re-loaded input position waited for the weight copy requested a few instructions earlier — one L2 round
    // trip per step with the matrix pipe idle, for both waves of a SIMD at once.  With plain loads every wait is an exact count.
    constexpr int NPCW = (WSTEPL / 64) / NWV;                    // weight pieces (1 KiB) per wave and step
    constexpr bool REGW = KWO && !RESIDENT && (WSTEPL / 64) % NWV == 0 && !(ABL & 2);
    constexpr bool APF = KWO && (REGW || RESIDENT) && !(ABL & 128);   // A fragments read one group ahead (ABL 128: at the group's start)
    constexpr int CPW = APF ? 1 : (CBN % 2 == 0 ? 2 : 1);        // channel blocks per kw-outer MFMA group
    constexpr int NGRP = KS * (CBN / CPW);                       // MFMA groups per step
    constexpr int PPG = REGW ? (NPCW + NGRP - 1) / NGRP : 1;     // most pieces any group carries
    auto regw_first = [](int g) { return (g * NPCW + NGRP - 1) / NGRP; };   // first piece of group g: piece k travels with group floor(k*NGRP/NPCW)
    static_assert(MTN % MSPLIT == 0 && (MSPLIT == 1 || PARTS || (GOUT == 0 && !CSUM)), "cout split: statistics only as per-block partials");
    const int mz = MSPLIT > 1 ? blockIdx.z * MTL : 0;
    static_assert(INMODE == 0 || INMODE == 1, "raw or GroupNorm(8)+ReLU input");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* wg4 = (const f32x4*)A.wfrag;
    if (RESIDENT) {   // all taps of the layer stay in LDS (A.n_taps x WTAP float4): no per-step barrier
        for (int i = threadIdx.x; i < A.n_taps * WTAPL; i += NWV * 64) {
            const int t = i >> 6;
            lds[i] = MSPLIT == 1 ? wg4[i] : wg4[(size_t)((t / MTL) * MTN + mz + t % MTL) * 64 + (i & 63)];
        }
        __syncthreads();
    }
    // (the weights go to LDS before anything of the half tile is requested: with the activation prefetch issued first, enc_down ran
    // 7 % slower — 1.58 vs 1.47 ms)
    // The per-half-tile body is a lambda called once.  That is deliberate: with the body inline the compiler schedules the down conv's
    // two-step prefetch loop worse (enc_down 1.54 ms inline, 1.47 ms as a lambda; every other instantiation unchanged) — measured, kept.
    // one step's weight pieces (KS taps) into LDS window `slot`, 1 KiB per wave-instruction.  A fixed number of instructions per wave
    // where the pieces divide evenly: behind a loop with a run-time trip count the compiler cannot count the outstanding loads and
    // puts s_waitcnt vmcnt(0) in front of the first use of ANY loaded register, i.e. waits for the pieces it has just requested.
    auto stream_weights = [&](int tap0, int slot) {
        constexpr int NPC = WSTEPL / 64;
        if (NPC % NWV == 0) {
#pragma unroll
            for (int k = 0; k < NPC / NWV; ++k) {
                const int t = wave + k * NWV;
                const int piece = (t / MTL) * MTN + mz + t % MTL;
                glds16(wg4 + (size_t)tap0 * WTAP + piece * 64 + lane, lds + slot * WSTEPL + t * 64);
            }
        } else {
            for (int t = wave; t < NPC; t += NWV) {
                const int piece = (t / MTL) * MTN + mz + t % MTL;
                glds16(wg4 + (size_t)tap0 * WTAP + piece * 64 + lane, lds + slot * WSTEPL + t * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto process = [&](int half) {   // NWV half tiles per workgroup (16 for LDS-resident weights: 4 waves/SIMD behind one copy)
    const bool active = (half >> 1) < A.n_tiles;
    // waves past the last half tile only help staging the weights; in the launches without fused statistics (small batches: a
    // workgroup may hold 1 live wave and 7 idle ones) they also stay off the MFMA pipe their live neighbours need
    // PARTS (position-split launches with fused statistics): per-block partial sums go to ConvArgs::part_* instead of being finished
    static_assert(!PARTS || GOUT > 0 || CSUM, "PARTS: an instantiation with fused statistics");
    const bool work = (!PARTS && (GOUT > 0 || CSUM)) ? true : active;
    if (!active) half = 2 * A.n_tiles - 1;
    const int tile = half >> 1;
    const int jj = (lane & 15) + 16 * (half & 1), q4 = lane >> 4;
    f32x4 ta[INMODE == 1 ? CBN : 1], tb[INMODE == 1 ? CBN : 1];   // input channel 16cb + 4q4 + i, GroupNorm(8, CIN)
    if (INMODE == 1) {
#pragma unroll
        for (int cb = 0; cb < CBN; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * cb + 4 * q4 + i, g = c / (CIN / 8);
                const float mean = A.in_mean[((size_t)tile * 8 + g) * 32 + jj];
                const float rstd = A.in_rstd[((size_t)tile * 8 + g) * 32 + jj];
                ta[cb][i] = rstd * A.in_gamma[c];
                tb[cb][i] = __builtin_fmaf(-mean, ta[cb][i], A.in_beta[c]);
            }
    }
    GnAcc st[GOUT > 0 ? MTN : 1];
#pragma unroll
    for (int k = 0; k < (GOUT > 0 ? MTN : 1); ++k) st[k].init();
    f32x4 cs[CSUM ? MTN : 1], csb[CSUM ? MTN : 1];   // channel sums: closed blocks / open block (same 16-block rule, fp32)
#pragma unroll
    for (int k = 0; k < (CSUM ? MTN : 1); ++k) cs[k] = csb[k] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // input position `pos`, channel block cb of this lane: wave-uniform 64-bit base + 32-bit lane offset, so that the load takes the
    // scalar-base form (one address register per lane instead of a 64-bit pair, no per-row 64-bit vector add)
    const f32x4* in_u = (const f32x4*)A.in + (size_t)((ABL & 64) ? 0 : tile) * NPI * (CIN / 4) * 32;
    const unsigned lane_b = (unsigned)(q4 * 32 + jj) * 16u;
    const vq_buf xrs = buf_of(in_u);
    auto ldx = [&](int pos, int cb) -> f32x4 { return buf_ld16(xrs, lane_b + cb * 2048, (unsigned)pos * (CIN / 4) * 512u); };
    const vq_buf outb = buf_of((const f32x4*)A.out + (size_t)tile * NPO * (COUT / 4) * 32);
    const vq_buf skb = buf_of(RESID ? (const f32x4*)A.skip + (size_t)tile * NPO * (COUT / 4) * 32 : (const f32x4*)A.out);
    f32x4* out4 = (f32x4*)A.out + (size_t)tile * NPO * (COUT / 4) * 32 + q4 * 32 + jj;
    const f32x4* bias4 = (const f32x4*)A.bias_frag;   // plain [COUT]: quad 4mt + q4
    const int NS = A.n_steps;
    int g0, g1;
    split_range<SO * SO>(g0, g1);
    int si = gridDim.y > 1 ? A.grp_start[g0] : 0;
    int4 e = steps[si];
    int4 en = steps[si + 1 < NS ? si + 1 : NS - 1];
    f32x4 xr[SI][CBN];
    f32x4 xq[PF2 ? SI : 1][PF2 ? CBN : 1];   // PF2: the row of step+1 waits here while the row of step+2 is in flight
    // Streamed weights: the pieces of a step are requested BEFORE the input positions of the same step, here and in the loop, so that
    // "all but the last NLD vector-memory operations have completed" (s_waitcnt vmcnt(NLD), loads return in order) means "my weight
    // pieces have landed" without also waiting for the positions re-loaded at the end of the previous step.
    if (REGW) {
#pragma unroll
        for (int k = 0; k < NPCW; ++k) {
            const int t = wave + k * NWV;
            lds[(si & 1) * WSTEPL + t * 64 + lane] = wg4[(size_t)e.y * WTAP + ((t / MTL) * MTN + mz + t % MTL) * 64 + lane];
        }
    } else if (!RESIDENT) {
        stream_weights(e.y, si & 1);
    }
#pragma unroll
    for (int iw = 0; iw < SI; ++iw)
#pragma unroll
        for (int cb = 0; cb < CBN; ++cb) {
            xr[iw][cb] = ldx(e.x + iw, cb);
            if (PF2) xq[iw][cb] = ldx(en.x + iw, cb);
        }
    // The first row is waited for here, in full: entering the step loop with loads in flight in an order of the compiler's choosing
    // makes it answer the loop's own first-use dependences with the worst case of both paths, i.e. s_waitcnt vmcnt(0) in every step.
    __builtin_amdgcn_s_waitcnt(0x0f70);
    // GroupNorm + ReLU of one arrived position / channel block, applied right before its first MFMA of the step (not for the whole row
    // at the top of the step: that made every step begin by waiting for the position re-loaded last, one L2 round trip with the matrix
    // pipe idle, both waves of a SIMD at once because the barrier keeps them in step)
    auto arrive = [&](int iw, int cb) {
        f32x4 v = xr[iw][cb];
        xr[iw][cb] = gn_relu4(v, ta[cb], tb[cb]);
    };
    for (int row = g0; row < g1; ++row) {
        f32x4 acc[SO][MTL];
#pragma unroll
        for (int ow = 0; ow < SO; ++ow)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) acc[ow][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        bool last;
        do {
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
            if (REGW) {
                if (!(ABL & 1)) __syncthreads();      // every wave's pieces of W(step) are in LDS; every wave done reading W(step-1)
            } else if (!RESIDENT) {
                // this wave's weight pieces have landed: everything but the NLD position loads a working wave has issued since
                constexpr int NLD = (PF2 || (ABL & 8)) ? 0 : SI * CBN;
                static_assert(NLD < 64, "vmcnt is a 6-bit counter");
                if (work) __builtin_amdgcn_s_waitcnt(0x0f70 | (NLD & 15) | ((NLD >> 4) << 14));
                else __builtin_amdgcn_s_waitcnt(0x0f70);
                if (!(ABL & 1)) __syncthreads();      // every wave's pieces of W(step) landed; every wave done reading W(step-1)
                if (!(ABL & 2)) stream_weights(en.y, (si + 1) & 1);   // requested before any position of this step (see NLD)
            }
            const f32x4* wl = (RESIDENT ? lds + (size_t)e.y * WTAPL : lds + (si & 1) * WSTEPL) + lane;
            if (KWO) {
                // kw-outer order: the A fragments of one (kw, group of channel blocks) are read once from LDS and feed every output of the
                // row that has this kw (3-4 outputs x CP blocks x MTL x 4 MFMAs per CP*MTL ds_read_b128, instead of one block's MTL x 4).
                // Per output the taps still arrive with kw ascending and, inside a tap, the channel blocks ascending: same arithmetic.
                static_assert(!KWO || !PF2, "kw-outer: single rolling buffer");
                constexpr int CP = CPW, NG = NGRP;
                if (work) {
                    // APF: the fragments of group g+1 are read while group g computes (two sets of CP*MTL registers).  All eight waves reach
                    // a group boundary together (the barrier keeps them in step), so reads issued AT the boundary queue behind each other
                    // on the one LDS port with the matrix pipe idle; issued a group ahead nobody waits for them.
                    f32x4 a[APF ? 2 : 1][CP][MTL];
                    auto read_group = [&](int g, int set) {
                        const int kw = g / (CBN / CP), cbp = (g % (CBN / CP)) * CP;
                        if (!RESIDENT && !REGW) {   // window filled by global_load_lds: hand-ordered reads (see lds_frag_read)
                            const unsigned wa = lds_offset_of(wl);
#pragma unroll
                            for (int c2 = 0; c2 < CP; ++c2)
#pragma unroll
                                for (int mt = 0; mt < MTL; ++mt) a[set][c2][mt] = lds_frag_read(wa + ((kw * CBN + cbp + c2) * MTL + mt) * 1024);
                            lds_frags_wait();
#pragma unroll
                            for (int c2 = 0; c2 < CP; ++c2)
#pragma unroll
                                for (int mt = 0; mt < MTL; ++mt) lds_frag_use(a[set][c2][mt]);
                        } else {
#pragma unroll
                            for (int c2 = 0; c2 < CP; ++c2)
#pragma unroll
                                for (int mt = 0; mt < MTL; ++mt) a[set][c2][mt] = wl[((kw * CBN + cbp + c2) * MTL + mt) * 64];
                        }
                    };
                    if (APF) read_group(0, 0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const int kw = g / (CBN / CP), cbp = (g % (CBN / CP)) * CP, set = APF ? (g & 1) : 0;
                        if (INMODE == 1 && !(ABL & 16)) {   // positions whose first tap of the row is this kw arrive here
#pragma unroll
                            for (int iw = 0; iw < SI; ++iw) {
                                int firstkw = -1;
#pragma unroll
                                for (int k2 = KS - 1; k2 >= 0; --k2) {
                                    const int num = iw + PAD - k2;
                                    if (num >= 0 && num % STRIDE == 0 && num / STRIDE < SO) firstkw = k2;
                                }
                                if (firstkw == kw) {
#pragma unroll
                                    for (int c2 = 0; c2 < CP; ++c2) arrive(iw, cbp + c2);
                                }
                            }
                        }
                        // REGW: piece k of this wave's NPCW travels with group k*NG/NPCW (requested here, stored after the group's MFMAs)
                        f32x4 wnext[PPG];
                        if (REGW) {
#pragma unroll
                            for (int j = 0; j < PPG; ++j) {
                                const int k = regw_first(g) + j;
                                if (k < regw_first(g + 1)) {
                                    const int t = wave + k * NWV;
                                    wnext[j] = wg4[(size_t)en.y * WTAP + ((t / MTL) * MTN + mz + t % MTL) * 64 + lane];
                                }
                            }
                        }
                        if (APF) {
                            if (g + 1 < NG) read_group(g + 1, set ^ 1);
                        } else {
                            read_group(g, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // requests first, a whole group ahead of their use (not sunk next to it)
                        static_assert(SO % OWI == 0, "outputs interleaved per MFMA run");
#pragma unroll
                        for (int ow0 = 0; ow0 < SO; ow0 += OWI) {
                            // k outer, (output, cout tile) inner: consecutive MFMAs go to different accumulators (each accumulator still sees
                            // its k steps in ascending order), so neither the dependent-issue latency nor an instruction the scheduler
                            // drops between two of them lands between an MFMA and the one that needs its result.  OWI outputs share a
                            // run: MTL x OWI accumulators in rotation (4 -> 8 at OWI = 2: profiles/r04_mfma_chain_depth.txt)
#pragma unroll
                            for (int c2 = 0; c2 < CP; ++c2)
#pragma unroll
                                for (int k = 0; k < 4; ++k)
#pragma unroll
                                    for (int oo = 0; oo < OWI; ++oo) {
                                        const int ow = ow0 + oo, iw = ow * STRIDE - PAD + kw;
                                        if (iw < 0 || iw >= SI) continue;
#pragma unroll
                                        for (int mt = 0; mt < MTL; ++mt) acc[ow][mt] = mfma16(a[set][c2][mt][k], xr[iw][cbp + c2][k], acc[ow][mt]);
                                    }
                            // was this kw the position's last tap of the row?  Then these channel blocks are re-loaded for the next step
                            // right here, between the outputs' MFMA runs (not all of a group's re-loads in one burst at its end: the
                            // eight waves of a workgroup run in step, and a burst of requests stalls their issue)
#pragma unroll
                            for (int oo = 0; oo < OWI; ++oo) {
                                const int ow = ow0 + oo, iw = ow * STRIDE - PAD + kw;
                                if (iw < 0 || iw >= SI) continue;
                                int lastkw = -1;
#pragma unroll
                                for (int k2 = 0; k2 < KS; ++k2) {
                                    const int num = iw + PAD - k2;
                                    if (num >= 0 && num % STRIDE == 0 && num / STRIDE < SO) lastkw = k2;
                                }
                                if (lastkw == kw && !(ABL & 8)) {
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int c2 = 0; c2 < CP; ++c2)
                                        xr[iw][cbp + c2] = ldx(en.x + iw, (cbp + c2));   // next step's row (index clamped)
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (REGW) {
#pragma unroll
                            for (int j = 0; j < PPG; ++j) {
                                const int k = regw_first(g) + j;
                                if (k < regw_first(g + 1)) {
                                    const int t = wave + k * NWV;
                                    lds[((si + 1) & 1) * WSTEPL + t * 64 + lane] = wnext[j];
                                }
                            }
                        }
                        // positions no output of the row reads (strided layers) still follow the rolling buffer
#pragma unroll
                        for (int iw = 0; iw < SI; ++iw) {
                            int lastkw = -1;
#pragma unroll
                            for (int k2 = 0; k2 < KS; ++k2) {
                                const int num = iw + PAD - k2;
                                if (num >= 0 && num % STRIDE == 0 && num / STRIDE < SO) lastkw = k2;
                            }
                            if (lastkw < 0 && kw == 0) {
#pragma unroll
                                for (int c2 = 0; c2 < CP; ++c2)
                                    if (!(ABL & 8)) xr[iw][cbp + c2] = ldx(en.x + iw, (cbp + c2));   // next step's row (index clamped)
                            }
                        }
                    }
                } else if (REGW) {   // a wave without a half tile still carries its share of the weights
#pragma unroll
                    for (int k = 0; k < NPCW; ++k) {
                        const int t = wave + k * NWV;
                        lds[((si + 1) & 1) * WSTEPL + t * 64 + lane] = wg4[(size_t)en.y * WTAP + ((t / MTL) * MTN + mz + t % MTL) * 64 + lane];
                    }
                }
            } else
            if (work)
#pragma unroll
            for (int iw = 0; iw < SI; ++iw) {
                if (INMODE == 1 && !(ABL & 16)) {
                    bool used = false;
#pragma unroll
                    for (int ow = 0; ow < SO; ++ow) used = used || (iw - ow * STRIDE + PAD >= 0 && iw - ow * STRIDE + PAD < KS);
                    if (used) {
#pragma unroll
                        for (int cb = 0; cb < CBN; ++cb) arrive(iw, cb);
                    }
                }
#pragma unroll
                for (int ow = 0; ow < SO; ++ow) {
                    const int kw = iw - ow * STRIDE + PAD;
                    if (kw < 0 || kw >= KS) continue;
#pragma unroll
                    for (int cb = 0; cb < CBN; ++cb) {
                        f32x4 a[MTL];   // the cout tiles of this (tap, channel block): MTN LDS reads ahead of their MFMAs, no further
                        if (!RESIDENT && !(ABL & 4)) {   // streamed window: hand-ordered reads (see lds_frag_read)
                            const unsigned wa = lds_offset_of(wl);
#pragma unroll
                            for (int mt = 0; mt < MTL; ++mt) a[mt] = lds_frag_read(wa + ((kw * CBN + cb) * MTL + mt) * 1024);
                            lds_frags_wait();
#pragma unroll
                            for (int mt = 0; mt < MTL; ++mt) lds_frag_use(a[mt]);
                        } else {
#pragma unroll
                        for (int mt = 0; mt < MTL; ++mt) a[mt] = (ABL & 4) ? xr[(iw + mt) % SI][cb] : wl[((kw * CBN + cb) * MTL + mt) * 64];
                        }
#pragma unroll
                        for (int mt = 0; mt < MTL; ++mt) {
                            acc[ow][mt] = mfma16(a[mt].x, xr[iw][cb].x, acc[ow][mt]);
                            acc[ow][mt] = mfma16(a[mt].y, xr[iw][cb].y, acc[ow][mt]);
                            acc[ow][mt] = mfma16(a[mt].z, xr[iw][cb].z, acc[ow][mt]);
                            acc[ow][mt] = mfma16(a[mt].w, xr[iw][cb].w, acc[ow][mt]);
                        }
                        if (MTL >= 4) __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < CBN; ++cb) {
                    if (PF2) {
                        xr[iw][cb] = xq[iw][cb];
                        xq[iw][cb] = ldx(en2.x + iw, cb);   // the row after next (index clamped)
                    } else if (!(ABL & 8)) {
                        xr[iw][cb] = ldx(en.x + iw, cb);   // next step's row (index clamped)
                    }
                }
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
        // ---- epilogue: the SO positions of the row, ascending ----
        if (ABL & 32) {   // keep the accumulators alive, nothing else
            float t = 0.0f;
#pragma unroll
            for (int ow = 0; ow < SO; ++ow)
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt) t += acc[ow][mt].x + acc[ow][mt].w;
            if (t == 12345.678f) out4[0] = acc[0][0];
            continue;
        }
        f32x4 bq[MTL];   // this lane's bias quads, once per row (left in the expression below they are re-loaded for every position,
#pragma unroll           // each load followed by s_waitcnt vmcnt(0): the stores in between may alias as far as the compiler knows)
        for (int mt = 0; mt < MTL; ++mt) bq[mt] = bias4[4 * (mz + mt) + q4];
        // residual input: the loads of position ow+1 are in flight while position ow is finished (one exposed round trip per row
        // instead of one per position)
        f32x4 sk[RESID ? 2 : 1][RESID ? MTL : 1];
        auto load_skip = [&](int ow) {
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) sk[ow & 1][mt] = buf_ld16(skb, lane_b + (mz + mt) * 2048, (unsigned)(row * SO + ow) * (COUT / 4) * 512u);
        };
        if (RESID) load_skip(0);
#pragma unroll
        for (int ow = 0; ow < SO; ++ow) {
            if (RESID && ow + 1 < SO) {
                load_skip(ow + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) {
                f32x4 v = acc[ow][mt] + bq[mt];
                if (RESID) {
                    const f32x4 u = v * 0.1f;
                    v = sk[ow & 1][mt] + u;
                }
                if (active) buf_st16_nt(v, outb, lane_b + (mz + mt) * 2048, (unsigned)(row * SO + ow) * (COUT / 4) * 512u);   // streaming store (see conv_first_k)
                if (GOUT > 0) {
                    st[mt].add(v.x);
                    st[mt].add(v.y);
                    st[mt].add(v.z);
                    st[mt].add(v.w);
                }
                if (CSUM) csb[mt] = csb[mt] + v;
            }
        }
        // one output row = 4 positions = one statistics block of a 4^3 layer
        static_assert((GOUT == 0 && !CSUM) || NPO == 64, "statistics blocks are rows of 4 positions");
        if (GOUT > 0) {   // slot = 4-channel quad of the output (group of 4 at COUT = 32, half of a group of 8 at COUT = 64)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) {
                if (PARTS) st[mt].fold_store(active ? A.part_s : nullptr, A.part_q, part_index(tile, row, 4 * (mz + mt) + q4, jj));
                else st[mt].fold();
            }
        }
        if (CSUM) {
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) {
                if (PARTS && active) {
                    const float v4[4] = {csb[mt].x, csb[mt].y, csb[mt].z, csb[mt].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) A.part_c[(((size_t)tile * 16 + row) * 64 + 16 * (mz + mt) + 4 * q4 + r) * 32 + jj] = v4[r];
                }
                cs[mt] = cs[mt] + csb[mt], csb[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    }
    if (!active || PARTS) return;   // PARTS: gn_combine_k / csum_combine_k finish
    if (GOUT > 0) {
        constexpr int CPGO = COUT / 8;
        static_assert(GOUT == 0 || (GOUT == 8 && (CPGO == 4 || CPGO == 8)), "GroupNorm(8, COUT) with COUT = 32 or 64");
        const double inv_n = 1.0 / (double)(CPGO * NPO);
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            double S = st[mt].s, Q = st[mt].q;
            if (CPGO == 8) {   // group = quads (2g, 2g+1) = this lane's quad 4mt+q4 and its q4^1 neighbour (16 lanes away): low + high
                const int lo = __double2loint(S), hi = __double2hiint(S), lo2 = __double2loint(Q), hi2 = __double2hiint(Q);
                S = S + __hiloint2double(__shfl_xor(hi, 16, 64), __shfl_xor(lo, 16, 64));
                Q = Q + __hiloint2double(__shfl_xor(hi2, 16, 64), __shfl_xor(lo2, 16, 64));
            }
            float m, r;
            gn_finish(S, Q, inv_n, m, r);
            const int grp = CPGO == 8 ? 2 * mt + (q4 >> 1) : 4 * mt + q4;
            if (CPGO == 4 || (q4 & 1) == 0) {
                A.out_mean[((size_t)tile * 8 + grp) * 32 + jj] = m;
                A.out_rstd[((size_t)tile * 8 + grp) * 32 + jj] = r;
            }
        }
    }
    if (CSUM) {
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            const float v[4] = {cs[mt].x, cs[mt].y, cs[mt].z, cs[mt].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) A.out_csum[((size_t)tile * COUT + 16 * mt + 4 * q4 + r) * 32 + jj] = v[r];
        }
    }
    };
    process((int)(blockIdx.x * NWV) + wave);
}

// ------------------------------------------------------------------------------------------
// Encoder tail: ChannelAttention(32) (VQVAE_v2.py:242) -> Conv3d(32->128,k1) (:243) -> nearest
// codebook row (:358-367), with the projection FOLDED into the search:  ||z||^2 is the same for
// every code and z.e_k = x'.(P^T e_k) + b.e_k, so
//     argmin_k dist_k = argmax_k [ h_k + x'.Ep_k ],   Ep = E P  (256 x 32),  h_k = b.e_k - ||e_k||^2 / 2
// (Ep, h built in fp64 at vqhip_create; the fmaf chain of candidate k starts at h_k: it is the MFMA's C operand).  Per position: 8 code tiles x 16 MFMAs on the 32 gated
// channels instead of 64 (projection) + 512 (128-d distances); the 128-channel latent is never
// formed.  Ep fragments (32 KB) and h (1 KB) live in LDS.  First maximum = torch.argmin's first minimum.
// ------------------------------------------------------------------------------------------
struct VqArgs {
    const float* in;        // x11 L4 [tile][64][8][32][4]
    const float* se_csum;   // [tile][32][32]
    const float* se_fc0;    // [8][32]
    const float* se_fc2;    // [32][8]
    const float* se_gate;   // optional: gates precomputed per tile (position-split launches), [tile][32][32]
    const float* epfrag;    // A fragments of Ep: [u=4][ct=8][64][4]
    const float* ck_frag;   // h_k in D-fragment order [(ct*2+q)*16 + r]
    uint8_t* idx;           // [n_leaves][64]
    int64_t n_leaves;
    int n_tiles;
};

// SCAN: how the winning tile's scores are kept.  0: sixteen selects per tile (v_cndmask).  1: under the lanes-that-won EXEC mask
// (a divergent branch: the copies are plain moves, two registers each where the compiler finds v_pk_mov_b32 / v_mov_b64 — it
// if-converts the branch back into the selects).  2: in LDS — the lanes that won store the tile's 16 scores into their own 64-byte
// slot (4 ds_write_b128 under the EXEC mask: no vector-ALU slot at all on the pipe the MFMAs share), read back once per position.
// ABL (tools/ablate only): 1 no scan at all, 2 no score keeping, 4 no MFMAs
template <int NW, int SCAN = 0, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 4) void vq_folded_k(VqArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    __shared__ f32x4 ldsE[4 * 8 * 64];  // 32 KB
    __shared__ f32x4 ldsC[8 * 2 * 4];   // 1 KB
    __shared__ f32x4 ldsS[SCAN == 2 ? NW * 4 * 64 : 1];   // SCAN 2: [wave][score quad g][lane], 4 KB per wave
    for (int i = threadIdx.x; i < 4 * 8 * 64; i += NW * 64) ldsE[i] = ((const f32x4*)A.epfrag)[i];
    for (int i = threadIdx.x; i < 8 * 2 * 4; i += NW * 64) ldsC[i] = ((const f32x4*)A.ck_frag)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * NW + wave;
    if (tile >= A.n_tiles) return;
    const int j = lane & 31, q = lane >> 5;
    f32x4 gate[4];   // (vectors: the gate multiply below is two packed multiplies per float4)
    if (A.se_gate) {   // (uniform) split launches: every position range would repeat the same prologue
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) gate[u][i] = A.se_gate[((size_t)tile * 32 + 8 * u + 4 * q + i) * 32 + j];
    } else {
        float hid[8], gall[32];
        se_hidden<32>(A.se_csum + (size_t)tile * 32 * 32 + j, A.se_fc0, hid);
        se_gates<32>(hid, A.se_fc2, gall);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) gate[u][i] = q ? gall[8 * u + 4 + i] : gall[8 * u + i];
    }
    const f32x4* in4 = (const f32x4*)A.in + (size_t)tile * 64 * 8 * 32 + q * 32 + j;
    const int64_t leaf = (int64_t)tile * 32 + j;
    int p0 = 0, p1 = 64;
    if (gridDim.y > 1) p0 = (int)(blockIdx.y * 64 / gridDim.y), p1 = (int)((blockIdx.y + 1) * 64 / gridDim.y);
    const bool packed = ((p0 | p1) & 3) == 0 && ((uintptr_t)A.idx & 3) == 0;   // (uniform)
    unsigned pk = 0;
    f32x4 bn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bn[u] = in4[((size_t)p0 * 8 + 2 * u) * 32];
    for (int p = p0; p < p1; ++p) {
        f32x4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = bn[u] * gate[u];
        const int pn = p < 63 ? p + 1 : 63;
#pragma unroll
        for (int u = 0; u < 4; ++u) bn[u] = in4[((size_t)pn * 8 + 2 * u) * 32];
        // Scan (round 4).  Per code tile: the maximum of the lane's 16 candidates by a v_max3 tree (8 ops), "did the tile beat the running
        // best" (strictly: an earlier tile keeps a tie), the new best, the tile's number, and the tile's 16 scores kept where it won
        // (16 selects) — 27 vector ops per tile where compare + two selects per candidate took 48; the candidate's number inside the
        // winning tile is found ONCE per position (first score equal to the best: 32 ops).  Same result as the sequential scan: first
        // maximum = torch.argmin's first minimum.
        float best = -__builtin_inff();
        int bt = 0;
        f32x16 sv;
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = 0.0f;
        const f32x4* el = ldsE + lane;
#pragma unroll 2
        for (int ct = 0; ct < 8; ++ct) {
            // the 4 A-fragments and the 16 code constants of this code tile, requested ahead of the MFMAs
            const f32x4 a0 = el[(0 * 8 + ct) * 64], a1 = el[(1 * 8 + ct) * 64], a2 = el[(2 * 8 + ct) * 64], a3 = el[(3 * 8 + ct) * 64];
            f32x4 ck[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) ck[g] = ldsC[(ct * 2 + q) * 4 + g];
            // the chain of candidate k starts at h_k = -c_k / 2 (the MFMA's C operand) and ends as  h_k + x'.Ep_k = -score_k / 2
            f32x16 d;
#pragma unroll
            for (int g = 0; g < 4; ++g) d[4 * g + 0] = ck[g].x, d[4 * g + 1] = ck[g].y, d[4 * g + 2] = ck[g].z, d[4 * g + 3] = ck[g].w;
            if (!(ABL & 4)) {
            d = mfma32(a0.x, b[0].x, d);
            d = mfma32(a0.y, b[0].y, d);
            d = mfma32(a0.z, b[0].z, d);
            d = mfma32(a0.w, b[0].w, d);
            d = mfma32(a1.x, b[1].x, d);
            d = mfma32(a1.y, b[1].y, d);
            d = mfma32(a1.z, b[1].z, d);
            d = mfma32(a1.w, b[1].w, d);
            d = mfma32(a2.x, b[2].x, d);
            d = mfma32(a2.y, b[2].y, d);
            d = mfma32(a2.z, b[2].z, d);
            d = mfma32(a2.w, b[2].w, d);
            d = mfma32(a3.x, b[3].x, d);
            d = mfma32(a3.y, b[3].y, d);
            d = mfma32(a3.z, b[3].z, d);
            d = mfma32(a3.w, b[3].w, d);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] += a0.x * b[r & 3].x + a1.y * b[r & 3].y + a2.z + a3.w;
            }
            if (ABL & 1) {
                best += d[0] + d[5] + d[10] + d[15];
                continue;
            }
            const float t0 = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), d[2]), t1 = __builtin_fmaxf(__builtin_fmaxf(d[3], d[4]), d[5]);
            const float t2 = __builtin_fmaxf(__builtin_fmaxf(d[6], d[7]), d[8]), t3 = __builtin_fmaxf(__builtin_fmaxf(d[9], d[10]), d[11]);
            const float t4 = __builtin_fmaxf(__builtin_fmaxf(d[12], d[13]), d[14]);
            const float u0 = __builtin_fmaxf(__builtin_fmaxf(t0, t1), t2), u1 = __builtin_fmaxf(__builtin_fmaxf(t3, t4), d[15]);
            const float m = __builtin_fmaxf(u0, u1);
            const bool won = m > best;
            best = __builtin_fmaxf(best, m);
            bt = won ? ct : bt;
            if (ABL & 2) continue;
            if (SCAN == 2) {
                if (won) {   // (divergent: stores under the EXEC mask of the lanes whose best moved)
#pragma unroll
                    for (int g = 0; g < 4; ++g) ldsS[(wave * 4 + g) * 64 + lane] = (f32x4){d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                }
            } else if (SCAN == 1) {
                if (won) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[r] = d[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = won ? d[r] : sv[r];
            }
        }
        // register r of the winning tile holds code 32 bt + (r & 3) + 8 (r >> 2) (+ 4q): ascending in r, so the first r with sv[r] == best
        if (SCAN == 2) {   // (the wave's own stores, program order: every lane won at least tile 0, its slot is this position's)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t = ldsS[(wave * 4 + g) * 64 + lane];
                sv[4 * g] = t.x, sv[4 * g + 1] = t.y, sv[4 * g + 2] = t.z, sv[4 * g + 3] = t.w;
            }
        }
        int loc = 0;
#pragma unroll
        for (int r = 15; r >= 0; --r) loc = sv[r] == best ? (r & 3) + 8 * (r >> 2) : loc;
        if (SCAN == 2) loc = best > -__builtin_inff() ? loc : 0;   // a lane that never won (every score NaN or -inf) holds an older position's scores: code 0, as above
        int bk = 32 * bt + loc;
        bk += 4 * q;
        const float ob = __shfl_xor(best, 32, 64);
        const int ok = __shfl_xor(bk, 32, 64);
        if (ob > best || (ob == best && ok < bk)) bk = ok;
        // four consecutive positions of a leaf leave as one 32-bit store (a byte store per position touches 32 cache lines for 32 bytes)
        if (packed) {
            pk |= (unsigned)bk << (8 * (p & 3));
            if ((p & 3) == 3) {
                if (q == 0 && leaf < A.n_leaves) *(unsigned*)(A.idx + leaf * 64 + (p & ~3)) = pk;
                pk = 0;
            }
        } else if (q == 0 && leaf < A.n_leaves) {
            A.idx[leaf * 64 + p] = (uint8_t)bk;
        }
    }
}

// ------------------------------------------------------------------------------------------
// D0+D1: embedding gather (VQVAE_v2.py:373-375) + decoder stem Conv3d(128->64,k3,p1) (:257).
// The stem only ever sees the 256 codebook rows, so by linearity
//     y[co][p] = bias + sum_{valid taps} T[tap][idx[p+tap]][co],   T[tap][k][co] = sum_cin W[co][cin][tap] * E[k][cin]
// T (27 x 256 x 64 fp32 = 1.77 MB, L2-resident) is built once per codec from the weights by
// build_stem_lut_k with the contract's cin-chain ("P8", from 0); the per-leaf work drops from 14.2 M
// MACs to ~1000 table-row adds per leaf position and the 32 KiB/leaf gathered tensor never exists.
// Accumulation order per output: valid taps ascending, plain adds from 0, then + bias (oracle:
// conv_tapsum).  Also produces the GroupNorm(8,64) statistics of the output.
// Lane (leaf j, half h) owns couts [32h, 32h+32).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void build_stem_lut_k(const float* __restrict__ W /*[64][128][27]*/, const float* __restrict__ E /*[256][128]*/,
                                                       float* __restrict__ T /*[27][256][64]*/)
{
    const int tap = blockIdx.x / 256, k = blockIdx.x % 256, co = threadIdx.x;
    float p = 0.0f;
    for (int cc = 0; cc < 128; ++cc) {
        const int ci = (cc & ~7) + ((cc & 1) ? 4 : 0) + ((cc & 7) >> 1);  // P8: 0,4,1,5,2,6,3,7
        p = __builtin_fmaf(W[((size_t)co * 128 + ci) * 27 + tap], E[k * 128 + ci], p);
    }
    T[((size_t)tap * 256 + k) * 64 + co] = p;
}

// The same table on the matrix pipe (training: the table follows the weights AND the codebook, every step): one wave per (tap, cout
// tile of 32, code tile of 32), A = the stem's forward fragments (frag32: [tap][u 16][mt 2][lane][4]), B = codebook rows.  The K order
// of a 32x32x2 chain over u, i is 8u+i, 8u+4+i — "P8", the order of build_stem_lut_k's fmaf chain — so the two kernels agree bit for
// bit (tests/test_gpu_fulltrain.py::test_stem_table_kernels_agree); 0.17 ms -> a few microseconds.
__global__ __launch_bounds__(64) void build_stem_lut_mfma_k(const float* __restrict__ wfrag, const float* __restrict__ E /*[256][128]*/,
                                                            float* __restrict__ T /*[27][256][64]*/)
{
    const int tap = blockIdx.x >> 4, mt = blockIdx.x & 1, ct = (blockIdx.x >> 1) & 7;
    const int lane = threadIdx.x, j = lane & 31, q = lane >> 5;
    const f32x4* a4 = (const f32x4*)wfrag + ((size_t)tap * 16 * 2 + mt) * 64 + lane;   // + u * 2 * 64
    const f32x4* b4 = (const f32x4*)E + (size_t)(ct * 32 + j) * 32 + q;                 // + 2u
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const f32x4 a = a4[u * 2 * 64], b = b4[2 * u];
        d = mfma32(a.x, b.x, d);
        d = mfma32(a.y, b.y, d);
        d = mfma32(a.z, b.z, d);
        d = mfma32(a.w, b.w, d);
    }
    float* t = T + ((size_t)tap * 256 + ct * 32 + j) * 64 + 32 * mt + 4 * q;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[(r & 3) + 8 * (r >> 2)] = d[r];
}

__global__ __launch_bounds__(256) void stem_lut_k(const uint8_t* __restrict__ idx, const float* __restrict__ T, const float* __restrict__ bias,
                                                  float* __restrict__ out, float* __restrict__ out_mean, float* __restrict__ out_rstd,
                                                  const int4* __restrict__ steps, int n_steps, int64_t n_leaves, int n_tiles,
                                                  const int* __restrict__ grp_start, double* __restrict__ part_s = nullptr,
                                                  double* __restrict__ part_q = nullptr)
{
    // The kernel is a gather from the 1.8 MB table (L2-resident), bound by the L1's tag rate: a wave covers 8 leaves x one
    // 32-channel half of the table row, lane = (leaf l, 16-byte chunk c) with c fastest, so that the 8 lanes of a leaf read one
    // whole 128-byte line per load (one lane per leaf reading its 128 bytes as 8 loads cost 8 tag lookups per line: 1.2 ms
    // per 65536 leaves).  8 waves = one 32-leaf tile; three steps of table rows in flight per wave.
    __shared__ uint8_t sidx[4][64 * 8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave;          // (tile, leaf octet o, half h) = gw / 8, (gw / 2) % 4, gw % 2
    const int tile = gw >> 3, o = (gw >> 1) & 3, h = gw & 1;
    if (tile >= n_tiles) return;
    const int l = lane >> 3, c = lane & 7;
    const int jt = 8 * o + l;                       // leaf within the tile
    const int64_t leaf = (int64_t)tile * 32 + jt;
    uint8_t* my = sidx[wave];
    for (int p = c; p < 64; p += 8) my[p * 8 + l] = leaf < n_leaves ? idx[leaf * 64 + p] : 0;
    // (same wave writes and reads its own LDS region: program order suffices)
    const f32x4* T4 = (const f32x4*)T + h * 8 + c;
    const f32x4 b4 = ((const f32x4*)bias)[h * 8 + c];
    GnAcc st;
    st.init();
    f32x4* out4 = (f32x4*)out + (size_t)tile * 64 * 16 * 32 + (size_t)(h * 8 + c) * 32 + jt;

    int g0, g1;
    split_range<64>(g0, g1);
    int si = gridDim.y > 1 ? grp_start[g0] : 0;
    const int NSm = n_steps - 1;
    int4 e = steps[si], e1 = steps[min(si + 1, NSm)], e2 = steps[min(si + 2, NSm)], e3 = steps[min(si + 3, NSm)];
#define STEM_ROW(E) T4[((size_t)(E).y * 256 + my[(E).x * 8 + l]) * 16]
    f32x4 r0 = STEM_ROW(e), r1 = STEM_ROW(e1), r2 = STEM_ROW(e2), r3;
    int po = g0;
    f32x4 acc = {0, 0, 0, 0};
    bool done = false;
#define STEM_STEP(RC, RN)                                                         \
    if (!done) {                                                                  \
        RN = STEM_ROW(e3); /* step si+3 into the register freed by the previous step */ \
        const int4 e4 = steps[min(si + 4, NSm)];                                  \
        acc = acc + RC;                                                           \
        const bool last = (e.w & 2) != 0;                                         \
        e = e1, e1 = e2, e2 = e3, e3 = e4;                                        \
        ++si;                                                                     \
        if (last) {                                                               \
            const f32x4 v = acc + b4;                                             \
            out4[(size_t)po * 16 * 32] = v;                                       \
            st.add(v.x);                                                          \
            st.add(v.y);                                                          \
            st.add(v.z);                                                          \
            st.add(v.w);                                                          \
            acc = (f32x4){0, 0, 0, 0};                                            \
            if ((po & 3) == 3) st.fold_store(part_s, part_q, part_index(tile, po >> 2, h * 8 + c, jt)); /* 4 positions = one block */ \
            done = ++po == g1;                                                    \
        }                                                                         \
    }
    while (!done) {
        STEM_STEP(r0, r3)
        STEM_STEP(r1, r0)
        STEM_STEP(r2, r1)
        STEM_STEP(r3, r2)
    }
#undef STEM_STEP
#undef STEM_ROW
    if (!out_mean) return;  // position-split launch: statistics by gn_stats_seq_k
    // GroupNorm(8,64): group = 8 channels = the 4-channel partials of lanes c (even, low) and c+1 (high): low + high
    const double S = st.s + __shfl_xor(st.s, 1, 64), Q = st.q + __shfl_xor(st.q, 1, 64);
    if ((c & 1) == 0) {
        float m, r;
        gn_finish(S, Q, 1.0 / 512.0, m, r);
        out_mean[((size_t)tile * 8 + h * 4 + (c >> 1)) * 32 + jt] = m;
        out_rstd[((size_t)tile * 8 + h * 4 + (c >> 1)) * 32 + jt] = r;
    }
}

// ------------------------------------------------------------------------------------------
// D0-D2 fused (large passes): embedding gather + decoder stem conv (as stem_lut_k) + GroupNorm(8,64) + ReLU (VQVAE_v2.py:258-259)
// + the statistics of the result for ResidualBlock.gn1 (:205), in ONE kernel: the stem output never goes to HBM.
// GroupNorm needs the statistics of the whole leaf before any element can be normalised, so a workgroup owns FOUR leaves for all 64
// positions: phase 1 gathers (wave w = positions 8w .. 8w+7, lane = (leaf, 16-byte chunk), a leaf's 16 lanes read the whole
// 256-byte table row) into a 64 KB LDS tile and chains the statistics per 4-position block; the 16 block sums are added in order
// through LDS (the 16-block rule, same numbers as stem_lut_k); phase 2 normalises out of LDS, stores d2 in the L4 layout and chains
// the statistics of d2 (same decomposition as gn_relu_stats_k: one accumulator per channel quad).  80 KB of LDS -> two workgroups
// per CU: one gathers (L1-bound) while the other writes (HBM-bound).  Replaces stem_lut_k + gn_relu_stats_k: 1 GB less written,
// 1 GB less read per 65 536 leaves.
// ------------------------------------------------------------------------------------------
struct StemFusedArgs {
    const uint8_t* idx;
    const float* T;          // (tap, code) table [27][256][64]
    const float* bias;       // stem bias [64]
    const float* gamma;      // stem GroupNorm weight / bias [64]
    const float* beta;
    float* d2;               // out: relu(GroupNorm(stem)) L4 [tile][64][16][32][4]
    float* out_mean;         // statistics of d2 for the residual block's gn1: [tile][8][32]
    float* out_rstd;
    float* ystem_dbg;        // optional: raw stem output (debug fetch), same layout as d2
    const int4* steps;
    const int* grp_start;
    int n_steps;
    int64_t n_leaves;
    int n_tiles;
};

#ifndef STEM_DEPTH
#define STEM_DEPTH 8
#endif
// NW waves per workgroup: 8 (eight positions = two statistics blocks per wave) for full chunks; 16 (four positions = one block per wave,
// half the serial gather chain) for SOP-sized passes, where a handful of workgroups is all there is and the chain is the latency
template <int NW = 8>
__global__ __launch_bounds__(NW * 64) void stem_fused_k(StemFusedArgs A)
{
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    constexpr int PW = 64 / NW, BW = PW / 4;   // positions / statistics blocks per wave
    __shared__ f32x4 ys[64 * 4 * 16];          // [pos][leaf 4][chunk 16]
    __shared__ double part[16][64];            // [block][lane of (leaf, chunk)]: the sums, then (after a barrier) the sums of squares —
                                               // 8 KB instead of 16: 74 KB per workgroup, two workgroups per CU
    __shared__ uint8_t sidx[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x >> 3, lg = blockIdx.x & 7;
    const int l = lane >> 4, c = lane & 15;
    const int jt = 4 * lg + l;
    if (tid < 256) {
        const int64_t leaf = (int64_t)tile * 32 + 4 * lg + (tid >> 6);
        sidx[tid >> 6][tid & 63] = leaf < A.n_leaves ? A.idx[leaf * 64 + (tid & 63)] : 0;
    }
    __syncthreads();
    const f32x4 b4 = ((const f32x4*)A.bias)[c];
    const uint8_t* my = sidx[l];

    // ---- phase 1: gather + bias -> LDS, statistics per 4-position block ----
    double bs0 = 0.0, bq0 = 0.0, bs1 = 0.0, bq1 = 0.0;   // this wave's two blocks (2w, 2w+1) of the tensor being reduced
    // ordered sum of the 16 blocks of every lane's accumulator: sums through LDS, barrier, sums of squares through the same buffer
    auto ordered_totals = [&](double& S, double& Q) {
        part[BW * wave][lane] = bs0;
        if (BW == 2) part[2 * wave + 1][lane] = bs1;
        __syncthreads();
        S = 0.0;
#pragma unroll
        for (int b = 0; b < 16; ++b) S += part[b][lane];
        __syncthreads();
        part[BW * wave][lane] = bq0;
        if (BW == 2) part[2 * wave + 1][lane] = bq1;
        __syncthreads();
        Q = 0.0;
#pragma unroll
        for (int b = 0; b < 16; ++b) Q += part[b][lane];
        __syncthreads();
    };
    {
        const int g0 = wave * PW, g1 = g0 + PW;
        int si = A.grp_start[g0];
        const int NSm = A.n_steps - 1;
        // D - 1 table rows in flight per wave (ring of D registers and D schedule entries, static indices).  The gather is bound by the
        // L2 -> L1 path: 64 B/clk x ~700 ns of latency is ~100 KB in flight per CU; 16 waves x 3 rows x 1 KiB was half of that.
        constexpr int D = STEM_DEPTH;
        const vq_buf tb = buf_of(A.T);
        // row of (tap, code): wave-uniform tap offset, lane offset = code row + this lane's 16-byte chunk
        auto row = [&](const int4& E) -> f32x4 { return buf_ld16(tb, ((unsigned)my[E.x] * 64u + (unsigned)c * 4u) * 4u, (unsigned)E.y * 65536u); };
        int4 ent[D];
#pragma unroll
        for (int i = 0; i < D; ++i) ent[i] = A.steps[min(si + i, NSm)];
        f32x4 r[D];
#pragma unroll
        for (int i = 0; i < D - 1; ++i) r[i] = row(ent[i]);
        int po = g0;
        f32x4 acc = {0, 0, 0, 0};
        GnAcc st;
        st.init();
        bool done = false;
        // (the wave's two block sums wait in registers: bs0/bq0, bs1/bq1)
        while (!done) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (!done) {
                    r[(k + D - 1) % D] = row(ent[(k + D - 1) % D]);
                    acc = acc + r[k];
                    const bool last = (ent[k].w & 2) != 0;
                    ent[k] = A.steps[min(si + D, NSm)];
                    ++si;
                    if (last) {
                        const f32x4 v = acc + b4;
                        ys[(po * 4 + l) * 16 + c] = v;
                        st.add(v.x);
                        st.add(v.y);
                        st.add(v.z);
                        st.add(v.w);
                        acc = (f32x4){0, 0, 0, 0};
                        if ((po & 3) == 3) {
                            if ((po >> 2) & (BW - 1)) bs1 = st.bs, bq1 = st.bq;
                            else bs0 = st.bs, bq0 = st.bq;
                            st.init();
                        }
                        done = ++po == g1;
                    }
                }
            }
        }
    }
    // the 16 block sums in order, then low quad + high quad of the 8-channel group (lanes c and c ^ 1)
    float ia[4], ib[4];
    {
        double S, Q;
        ordered_totals(S, Q);   // (its barriers also make the LDS tile visible to phase 2)
        const double S2 = __shfl_xor(S, 1, 64), Q2 = __shfl_xor(Q, 1, 64);
        float m, r;
        gn_finish((c & 1) ? S2 + S : S + S2, (c & 1) ? Q2 + Q : Q + Q2, 1.0 / 512.0, m, r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ia[i] = r * A.gamma[4 * c + i];
            ib[i] = __builtin_fmaf(-m, ia[i], A.beta[4 * c + i]);
        }
    }

    // ---- phase 2: normalise + ReLU out of LDS -> d2, statistics of d2 per block ----
    {
        f32x4* out4 = (f32x4*)A.d2 + ((size_t)tile * 64 * 16 + c) * 32 + jt;
        f32x4* dbg4 = A.ystem_dbg ? (f32x4*)A.ystem_dbg + ((size_t)tile * 64 * 16 + c) * 32 + jt : nullptr;
        GnAcc st;
        st.init();
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int po = wave * PW + k;
            const f32x4 v = ys[(po * 4 + l) * 16 + c];
            if (dbg4) dbg4[(size_t)po * 16 * 32] = v;
            f32x4 y;
            y.x = fmaxf(__builtin_fmaf(v.x, ia[0], ib[0]), 0.0f);
            y.y = fmaxf(__builtin_fmaf(v.y, ia[1], ib[1]), 0.0f);
            y.z = fmaxf(__builtin_fmaf(v.z, ia[2], ib[2]), 0.0f);
            y.w = fmaxf(__builtin_fmaf(v.w, ia[3], ib[3]), 0.0f);
            __builtin_nontemporal_store(y, &out4[(size_t)po * 16 * 32]);
            st.add(y.x);
            st.add(y.y);
            st.add(y.z);
            st.add(y.w);
            if ((k & 3) == 3) {
                if ((k >> 2) & (BW - 1)) bs1 = st.bs, bq1 = st.bq;
                else bs0 = st.bs, bq0 = st.bq;
                st.init();
            }
        }
    }
    {
        double S, Q;
        ordered_totals(S, Q);
        const double S2 = __shfl_xor(S, 1, 64), Q2 = __shfl_xor(Q, 1, 64);
        if (wave == 0 && (c & 1) == 0) {
            float m, r;
            gn_finish(S + S2, Q + Q2, 1.0 / 512.0, m, r);
            A.out_mean[((size_t)tile * 8 + (c >> 1)) * 32 + jt] = m;
            A.out_rstd[((size_t)tile * 8 + (c >> 1)) * 32 + jt] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stand-alone statistics of a stored activation, by the 16-block rule (same results as the fused ones).  The training step uses
// them (its forward keeps every activation and recomputes the statistics the backward pass needs); the inference split path
// fuses its statistics as per-block partials instead.
//   gn_stats_seq_k<C, NP, CPG>: mean / rstd of GroupNorm with CPG (2, 4 or 8) channels per group, [tile][C/CPG][32]
//   csum_seq_k<C, NP>         : per-channel fp32 sums over the positions (ChannelAttention), [tile][C][32]
//   ew_gn_relu_k<C, G>        : y = relu(GroupNorm_G(x)) elementwise (first-conv output, VQVAE_v2.py:236-237)
// Wave w of a workgroup owns channel quads 2w, 2w+1 (lane half h -> quad 2w+h) of one leaf tile.
// ------------------------------------------------------------------------------------------
template <int C, int NP, int CPG>
__global__ __launch_bounds__(64 * C / 8) void gn_stats_seq_k(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd)
{
    static_assert(CPG == 2 || CPG == 4 || CPG == 8, "2/4: lane-local groups; 8: low quad + high quad of one wave");
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int quad = 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
    const int tile = blockIdx.x;
    const f32x4* in4 = (const f32x4*)x + (size_t)tile * NP * (C / 4) * 32 + quad * 32 + j;
    constexpr int NACC = CPG == 2 ? 2 : 1;
    GnAcc st[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) st[k].init();
    // latency-bound twice over (one dependent fp64 chain; a lone wave's loads): keep 32 loads in flight ahead of the chain
    constexpr int NB = 32;
    static_assert(NP % NB == 0, "positions in batches of 32");
    f32x4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = in4[(size_t)k * (C / 4) * 32];
    for (int p0 = 0; p0 < NP; p0 += NB) {
        f32x4 u[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) u[k] = v[k];
        const int pn = p0 + NB < NP ? p0 + NB : p0;   // last batch: re-request valid data
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = in4[(size_t)(pn + k) * (C / 4) * 32];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            st[0].add(u[k].x);
            st[0].add(u[k].y);
            st[CPG == 2 ? 1 : 0].add(u[k].z);
            st[CPG == 2 ? 1 : 0].add(u[k].w);
            if ((k + 1) % (NP / 16 < NB ? NP / 16 : NB) == 0 && (NP / 16 <= NB)) {   // statistics block boundary (NP/16 = 4 or 32 positions)
#pragma unroll
                for (int a = 0; a < NACC; ++a) st[a].fold();
            }
        }
    }
    static_assert(NP / 16 <= 32 && 32 % (NP / 16) == 0, "blocks of NP/16 positions inside batches of 32");
    constexpr int G = C / CPG;
    if (CPG == 8) {  // group = this wave's two quads: low-quad partial + high-quad partial
        float m, r;
        gn_finish(st[0].s + shfl_xor32_f64(st[0].s), st[0].q + shfl_xor32_f64(st[0].q), 1.0 / (double)(8 * NP), m, r);
        if ((lane >> 5) == 0) {
            mean[((size_t)tile * G + (quad >> 1)) * 32 + j] = m;
            rstd[((size_t)tile * G + (quad >> 1)) * 32 + j] = r;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        float m, r;
        gn_finish(st[k].s, st[k].q, 1.0 / (double)(CPG * NP), m, r);
        const int g = quad * (4 / CPG) + k;
        mean[((size_t)tile * G + g) * 32 + j] = m;
        rstd[((size_t)tile * G + g) * 32 + j] = r;
    }
}

// ... and for the tensor whose statistics blocks are its 128 output HALF rows, added row-major (conv1 output of the 16-channel
// residual block): wave = channel quads (2w, 2w+1), GroupNorm(8,16): 2 channels per group.
__global__ __launch_bounds__(128) void gn_stats_rows16_k(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd)
{
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int quad = 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
    const int tile = blockIdx.x;
    const f32x4* in4 = (const f32x4*)x + (size_t)tile * 512 * 4 * 32 + quad * 32 + j;
    double S[2] = {0.0, 0.0}, Q[2] = {0.0, 0.0};
    for (int o = 0; o < 16; ++o) {   // half row (oh, hw) = o
        double ts[2] = {0.0, 0.0}, tq[2] = {0.0, 0.0};
        for (int od = 0; od < 8; ++od) {
            f32x4 v[4];
#pragma unroll
            for (int ow = 0; ow < 4; ++ow) v[ow] = in4[(size_t)((od * 16 + o) * 4 + ow) * 4 * 32];
            GnAcc a[2];
            a[0].init(), a[1].init();
#pragma unroll
            for (int ow = 0; ow < 4; ++ow) a[0].add(v[ow].x), a[0].add(v[ow].y), a[1].add(v[ow].z), a[1].add(v[ow].w);
            ts[0] += a[0].bs, tq[0] += a[0].bq, ts[1] += a[1].bs, tq[1] += a[1].bq;
        }
        S[0] += ts[0], Q[0] += tq[0], S[1] += ts[1], Q[1] += tq[1];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float m, r;
        gn_finish(S[k], Q[k], 1.0 / 1024.0, m, r);
        mean[((size_t)tile * 8 + 2 * quad + k) * 32 + j] = m;
        rstd[((size_t)tile * 8 + 2 * quad + k) * 32 + j] = r;
    }
}

// ChannelAttention gates of a tile, once, with the C/4 hidden units and the C gates of a leaf spread over the workgroup's C/4
// (quad) threads of that leaf: the same fmaf chains as se_hidden / se_gates (vq_device.h), which every consumer wave would
// otherwise run serially in its prologue (2 x C x C/4 fmafs behind C dependent loads).  s = this thread's 4 channel sums.
template <int C>
__device__ __forceinline__ void se_gates_block(int tile, int quad, int j, f32x4 s, const float* __restrict__ fc0, const float* __restrict__ fc2,
                                               float* __restrict__ gates)
{
    constexpr int R = C / 4;
    __shared__ float cs[C][32], hs[R][32];
    cs[4 * quad + 0][j] = s.x, cs[4 * quad + 1][j] = s.y, cs[4 * quad + 2][j] = s.z, cs[4 * quad + 3][j] = s.w;
    __syncthreads();
    float h = 0.0f;
    for (int c = 0; c < C; ++c) h = __builtin_fmaf(fc0[quad * C + c], cs[c][j] * (1.0f / 64.0f), h);
    hs[quad][j] = h > 0.0f ? h : 0.0f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * quad + i;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < R; ++k) a = __builtin_fmaf(fc2[c * R + k], hs[k][j], a);
        gates[((size_t)tile * C + c) * 32 + j] = vq_sigmoid(a);
    }
}

// thread `tid` of the 64 C / 8 threads that own tile `tile`: lane = (leaf j, channel quad)
template <int C, int NP>
__device__ __forceinline__ f32x4 csum_seq_thread(const float* __restrict__ x, float* __restrict__ csum, int tile, int tid)
{
    const int lane = tid & 63, j = lane & 31;
    const int quad = 2 * (tid >> 6) + (lane >> 5);
    const f32x4* in4 = (const f32x4*)x + (size_t)tile * NP * (C / 4) * 32 + quad * 32 + j;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll(NP <= 64 ? 16 : 1)
    for (int b = 0; b < 16; ++b) {   // 16-block rule: block sums from zero, added in block order
        f32x4 sb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int p = b * (NP / 16); p < (b + 1) * (NP / 16); ++p) sb = sb + in4[(size_t)p * (C / 4) * 32];
        s = s + sb;
    }
    csum[((size_t)tile * C + 4 * quad + 0) * 32 + j] = s.x;
    csum[((size_t)tile * C + 4 * quad + 1) * 32 + j] = s.y;
    csum[((size_t)tile * C + 4 * quad + 2) * 32 + j] = s.z;
    csum[((size_t)tile * C + 4 * quad + 3) * 32 + j] = s.w;
    return s;
}
template <int C, int NP>
__global__ __launch_bounds__(64 * C / 8) void csum_seq_k(const float* __restrict__ x, float* __restrict__ csum, const float* __restrict__ fc0 = nullptr,
                                                         const float* __restrict__ fc2 = nullptr, float* __restrict__ gates = nullptr)
{
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int quad = 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
    const int tile = blockIdx.x;
    const f32x4 s = csum_seq_thread<C, NP>(x, csum, tile, (int)threadIdx.x);
    if (gates) se_gates_block<C>(tile, quad, j, s, fc0, fc2, gates);   // (uniform)
}

// Position-split launches with fused statistics: add the 16 block sums of every accumulator slot in block order (the fold()
// chain of the one-wave-per-tile launch) and finish.  PAIR: groups of 8 channels = low quad + high quad (slots 2g, 2g+1).
template <bool PAIR, bool ROWS = false>
__global__ __launch_bounds__(512) void gn_combine_k(const double* __restrict__ ps, const double* __restrict__ pq, float* __restrict__ mean,
                                                    float* __restrict__ rstd, int n_groups, double inv_n)
{
    const int tile = blockIdx.x, slot = threadIdx.x >> 5, j = threadIdx.x & 31;
    double S = 0.0, Q = 0.0;
    if (ROWS) {   // the tensor with 128 half-row blocks (8 slots), added row-major: conv1 output of the 16-channel residual block
        for (int o = 0; o < 16; ++o) {   // half row (oh, hw) = o
            double ts = 0.0, tq = 0.0;
#pragma unroll
            for (int od = 0; od < 8; ++od) {
                ts += ps[part_index_rows(tile, od * 16 + o, slot, j)];
                tq += pq[part_index_rows(tile, od * 16 + o, slot, j)];
            }
            S += ts;
            Q += tq;
        }
    } else {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            S += ps[part_index(tile, b, slot, j)];
            Q += pq[part_index(tile, b, slot, j)];
        }
    }
    if (PAIR) {   // slots 2g (lanes 0..31 of the wave) and 2g+1 (lanes 32..63)
        S = S + shfl_xor32_f64(S);
        Q = Q + shfl_xor32_f64(Q);
        if (slot & 1) return;
    }
    float m, r;
    gn_finish(S, Q, inv_n, m, r);
    const int g = PAIR ? slot >> 1 : slot;
    mean[((size_t)tile * n_groups + g) * 32 + j] = m;
    rstd[((size_t)tile * n_groups + g) * 32 + j] = r;
}

// ... and the channel sums (+ attention gates, like csum_seq_k)
template <int C>
__global__ __launch_bounds__(64 * C / 8) void csum_combine_k(const float* __restrict__ pc, float* __restrict__ csum, const float* __restrict__ fc0,
                                                             const float* __restrict__ fc2, float* __restrict__ gates)
{
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int quad = 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
    const int tile = blockIdx.x;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] + pc[(((size_t)tile * 16 + b) * 64 + 4 * quad + r) * 32 + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[((size_t)tile * C + 4 * quad + r) * 32 + j] = v[r];
    se_gates_block<C>(tile, quad, j, (f32x4){v[0], v[1], v[2], v[3]}, fc0, fc2, gates);
}

template <int C, int NP, int G>
__global__ __launch_bounds__(256) void ew_gn_relu_k(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta)
{
    // one thread per float4 of the tile: index = (p*(C/4) + quad)*32 + j
    const int tile = blockIdx.x;
    constexpr int CPG = C / G;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < NP * (C / 4) * 32; i += gridDim.y * 256) {
        const int j = i & 31, quad = (i >> 5) % (C / 4);
        f32x4 v = ((const f32x4*)x)[(size_t)tile * NP * (C / 4) * 32 + i];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * quad + k, g = c / CPG;
            const float ia = rstd[((size_t)tile * G + g) * 32 + j] * gamma[c];
            const float ib = __builtin_fmaf(-mean[((size_t)tile * G + g) * 32 + j], ia, beta[c]);
            o[k] = fmaxf(__builtin_fmaf(o[k], ia, ib), 0.0f);
        }
        ((f32x4*)y)[(size_t)tile * NP * (C / 4) * 32 + i] = (f32x4){o[0], o[1], o[2], o[3]};
    }
}

// ------------------------------------------------------------------------------------------
// Hardware-assumption probe: fp32 MFMA == k-ordered fmaf chain (cdna guide §3).
// ------------------------------------------------------------------------------------------
__global__ void mfma_probe_k(const float* __restrict__ a32, const float* __restrict__ b32, const float* __restrict__ a16,
                             const float* __restrict__ b16, int ksteps, unsigned long long* mism)
{
    const int lane = threadIdx.x;
    // 32x32x2: A[32][2*ksteps] row-major, B[2*ksteps][32]
    {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int s = 0; s < ksteps; ++s) {
            const int k = 2 * s + (lane >> 5);
            acc = mfma32(a32[(lane & 31) * 2 * ksteps + k], b32[k * 32 + (lane & 31)], acc);
        }
        unsigned long long bad = 0;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            float ref = 0.0f;
            for (int k = 0; k < 2 * ksteps; ++k) ref = __builtin_fmaf(a32[row * 2 * ksteps + k], b32[k * 32 + col], ref);
            bad += (__float_as_uint(ref) != __float_as_uint(acc[r]));
        }
        if (bad) atomicAdd(mism, bad);
    }
    {
        f32x4 acc = {0, 0, 0, 0};
        for (int s = 0; s < ksteps; ++s) {
            const int k = 4 * s + (lane >> 4);
            acc = mfma16(a16[(lane & 15) * 4 * ksteps + k], b16[k * 16 + (lane & 15)], acc);
        }
        unsigned long long bad = 0;
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;
            float ref = 0.0f;
            for (int k = 0; k < 4 * ksteps; ++k) ref = __builtin_fmaf(a16[row * 4 * ksteps + k], b16[k * 16 + col], ref);
            bad += (__float_as_uint(ref) != __float_as_uint(acc[r]));
        }
        if (bad) atomicAdd(mism + 1, bad);
    }
}
