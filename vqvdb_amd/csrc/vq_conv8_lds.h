// vq_conv8_lds.h — Conv3d(16->16,k3,p1) @8^3 of ResidualBlock(16) (VQVAE_v2.py:190-210, :238) with the input PLANES staged in LDS.
//
// conv8_c16_k (vq_kernels.h) gives a wave four output rows and re-reads every input row 3.4x from L2/HBM: 7-9.5 GB fetched per
// 65 536-leaf launch for a 2.1 GB input, 3.2-3.8 TB/s sustained — the two 16-channel convs (half of encode) sit on the HBM
// roofline as much as on the MFMA one.  Here a WORKGROUP (8 waves) owns a 16-leaf half tile and walks its 8 output planes; the input
// planes travel global -> registers -> (GroupNorm + ReLU, once) -> LDS exactly once:
//
//   LDS: 27 KB weight fragments | 2 plane slots of 64 KB ([pos 64][q4 4][leaf 16] float4), plane `id` lives in slot id & 1
//   per output plane od:   taps kd=0 (plane od-1)            | barrier | write plane od+1 into the slot plane od-1 just left,
//                          issue the global loads of plane od+2 | taps kd=1 (plane od) | barrier | taps kd=2 (plane od+1) | epilogue
//
// Wave w owns output row oh = w of the plane (8 accumulators of 16 couts x 16 leaves); per (kd,kh) it reads the 8 positions of input
// row oh+kh-1 from LDS (conflict-free 1 KB per instruction) and the three kw weight fragments, and issues, position by position,
// the 4-MFMA channel chains of the (up to) three outputs the position feeds, interleaved across those outputs.  Per output the taps
// still arrive in ascending (kd,kh,kw) order and the channels in P16 order: the arithmetic is conv8_c16_k's and the oracle's, bit
// for bit.  The workgroup is persistent: it walks half tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the plane pipeline runs
// straight across the boundary (the next half tile's plane 0 is prefetched during the last plane of the current one).
//
// Statistics (STATS, conv1): each output HALF ROW is one statistics block of this tensor (128 half-row blocks added row-major, DESIGN
// 4): a wave adds the 4 positions of a half row in one fp64 chain from zero, adds these blocks over the 8 planes in its registers,
// and stores the per-half-row totals (16 per half tile); gn_combine_k<false> adds them in (oh, hw) order.
#pragma once
#include "vq_kernels.h"

// workgroup barrier that waits for this wave's LDS operations only: s_waitcnt lgkmcnt(0) (vmcnt / expcnt fields at their maxima) + s_barrier
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

constexpr size_t LDS_CONV8 = (size_t)(27 * 64 + 2 * 4096) * 16;   // 155 648 B
static_assert(LDS_CONV8 <= 160 * 1024, "gfx950: 160 KB of LDS per workgroup, all of it dynamic here: the kernel must stay free of static __shared__");

// ABL: timing-only ablations for tools/ablate/conv8_lds_ablate.hip (0 in the library): 1 no barriers, 2 no epilogue, 4 no plane
// write / prefetch, 8 no LDS operand reads, 16 no MFMAs
template <bool RESID, bool STATS, int NW = 8, int ABL = 0, bool LD2 = false>
__global__ __launch_bounds__(NW * 64, 1) void conv8_lds_k(ConvArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    static_assert(!LD2 || NW == 8, "border-row loaders: the 8-wave variant");
    static_assert(NW == 8 || NW == 16, "8 waves: one output row each; 16 waves: one half row (4 positions) each");
    constexpr int NT = NW * 64, OWN = 64 / NW;   // threads; output positions (and loader positions) per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* wl = (f32x4*)smem_raw;          // [27 taps][64 lanes]
    f32x4* slots = wl + 27 * 64;           // [2][64 pos][4 q4][16 leaves]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, j16 = lane & 15;
    for (int i = tid; i < 27 * 64; i += NT) wl[i] = ((const f32x4*)A.wfrag)[i];
    const int n_half = 2 * A.n_tiles;
    if ((int)blockIdx.x >= n_half) return;
    const int n_my = (n_half - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // half tiles of this workgroup
    const int NPL = n_my * 8;                                                             // planes it walks
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    float gam[4], bet[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gam[i] = A.in_gamma[4 * q4 + i], bet[i] = A.in_beta[4 * q4 + i];
    // NW = 8: wave = output row oh of every plane.  NW = 16: wave = half row (oh, 4 positions from ow0): every SIMD then hosts one
    // border half row (2 of 3 kh taps) and three inner ones — equal MFMA work per SIMD between barriers.  As a loader a wave
    // brings in OWN consecutive input positions of every plane.
    const int oh = NW == 8 ? wave : wave >> 1, ow0 = NW == 8 ? 0 : (wave & 1) * 4;
    const unsigned lane_b = (unsigned)(q4 * 32 + j16) * 16u;   // this lane inside a [pos][4 q4][32 leaves] float4 position (see buf_ld16)

    // ---- plane prefetch: LPOS positions per loading wave, this lane's channel quad, + the GroupNorm statistics of that half tile ----
    // LD2: only the two border-row waves (rows 0 and 7 have 2 of the 3 kh taps: a third less MFMA work) stage the planes, 32
    // positions each — the staging then sits in their slack instead of on every wave's critical path.
    constexpr int LPOS = LD2 ? 32 : OWN;
    const bool loader = !LD2 || wave == 0 || wave == NW - 1;
    const int lbase = LD2 ? (wave == 0 ? 0 : 32) : wave * OWN;   // first position this wave brings in
    f32x4 pf[LPOS];
    float pm0, pm1, pr0, pr1;
    auto issue_prefetch = [&](int P) {
        if (!loader) return;   // (wave-uniform)
        const int hh = (int)blockIdx.x + (P >> 3) * (int)gridDim.x, id = P & 7;
        const int tile = hh >> 1, jj = j16 + 16 * (hh & 1);
        const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * 512 * 4 * 32 + 16 * (hh & 1));
#pragma unroll
        for (int k = 0; k < LPOS; ++k) pf[k] = buf_ld16(inb, lane_b, (unsigned)(id * 64 + lbase + k) * 2048u);
        // channels 4q4 .. 4q4+3 -> GroupNorm(8,16) groups 2q4 (channels 0,1 of the quad) and 2q4+1 (channels 2,3): the statistics
        // change with the half tile only, so they travel with its first plane (they used to be re-loaded with every plane: four
        // per-lane-pointer loads per wave and plane in a kernel whose staging costs 6 %)
        if (id == 0) {   // (wave-uniform)
            pm0 = A.in_mean[((size_t)tile * 8 + 2 * q4) * 32 + jj], pr0 = A.in_rstd[((size_t)tile * 8 + 2 * q4) * 32 + jj];
            pm1 = A.in_mean[((size_t)tile * 8 + 2 * q4 + 1) * 32 + jj], pr1 = A.in_rstd[((size_t)tile * 8 + 2 * q4 + 1) * 32 + jj];
        }
    };
    float ia[4], ib[4];   // GroupNorm scale / shift of this lane's channel quad for the half tile being staged
    auto write_plane = [&](int P) {   // relu(GroupNorm(x)) once per element, then into slot P & 1
        if (!loader) return;   // (wave-uniform)
        if ((P & 7) == 0) {   // (wave-uniform) first plane of a half tile: its statistics arrived with this plane's prefetch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ia[i] = (i < 2 ? pr0 : pr1) * gam[i];
                ib[i] = __builtin_fmaf(-(i < 2 ? pm0 : pm1), ia[i], bet[i]);
            }
        }
        f32x4* dst = slots + (P & 1) * 4096 + (lbase * 4 + q4) * 16 + j16;
#pragma unroll
        for (int k = 0; k < LPOS; ++k) {
            f32x4 v = pf[k];
            v.x = fmaxf(__builtin_fmaf(v.x, ia[0], ib[0]), 0.0f);
            v.y = fmaxf(__builtin_fmaf(v.y, ia[1], ib[1]), 0.0f);
            v.z = fmaxf(__builtin_fmaf(v.z, ia[2], ib[2]), 0.0f);
            v.w = fmaxf(__builtin_fmaf(v.w, ia[3], ib[3]), 0.0f);
            dst[k * 4 * 16] = v;
        }
    };

    f32x4 acc[OWN];
    // STATS: running sums of this wave's half row(s) over the planes of the current half tile ([half row of the wave][group of the quad])
    constexpr int HW = NW == 8 ? 2 : 1;
    GnAcc st[HW][2];
#pragma unroll
    for (int h = 0; h < HW; ++h) st[h][0].init(), st[h][1].init();
    // the taps of one kd: input plane in `slot`, rows oh-1 .. oh+1
    // valid kh of this wave's row (zero padding: the other taps do not exist): k0 .. k0 + nk - 1
    const int k0 = oh == 0 ? 1 : 0, nk = (oh == 0 || oh == 7) ? 2 : 3;
    auto taps = [&](const f32x4* slot, int kd) {
        // input positions iw = ow0 - 1 + li, li = 0 .. OWN + 1 (the row's positions for NW = 8 are li = 1 .. 8); position li feeds
        // output li (local index) with kw = 0, li - 1 with kw = 1, li - 2 with kw = 2: per output the kw arrive ascending (li
        // ascends), the four channel steps of a tap in order; the (up to) three outputs interleave (independent accumulators).
        f32x4 xbuf[1][OWN + 2];
        auto load_row = [&](f32x4 (&x)[OWN + 2], int kh) {
            const f32x4* src = slot + (((oh + kh - 1) * 8 + ow0 - 1) * 4 + q4) * 16 + j16;
#pragma unroll
            for (int li = 0; li < OWN + 2; ++li) {
                const int iw = ow0 - 1 + li;
                if (iw >= 0 && iw <= 7) x[li] = (ABL & 8) ? (f32x4){bias4.x, bias4.y, bias4.z, bias4.w} : src[li * 4 * 16];   // (wave-uniform)
            }
        };
        auto compute = [&](const f32x4 (&x)[OWN + 2], int kh) {
            const f32x4* wp = wl + ((kd * 3 + kh) * 3) * 64 + lane;
            const f32x4 w0 = wp[0], w1 = wp[64], w2 = wp[128];
#pragma unroll
            for (int li = 0; li < OWN + 2; ++li) {
                const int iw = ow0 - 1 + li;
                if (iw < 0 || iw > 7) continue;   // (wave-uniform) zero padding
#pragma unroll
                for (int comp = 0; comp < 4; ++comp) {
                    const float xv = comp == 0 ? x[li].x : comp == 1 ? x[li].y : comp == 2 ? x[li].z : x[li].w;
                    if (ABL & 16) {
                        acc[li < OWN ? li : 0].x += xv;
                        continue;
                    }
                    if (li < OWN) acc[li] = mfma16(comp == 0 ? w0.x : comp == 1 ? w0.y : comp == 2 ? w0.z : w0.w, xv, acc[li]);
                    if (li >= 1 && li <= OWN) acc[li - 1] = mfma16(comp == 0 ? w1.x : comp == 1 ? w1.y : comp == 2 ? w1.z : w1.w, xv, acc[li - 1]);
                    if (li >= 2) acc[li - 2] = mfma16(comp == 0 ? w2.x : comp == 1 ? w2.y : comp == 2 ? w2.z : w2.w, xv, acc[li - 2]);
                }
            }
        };
        // (reading row kh+1 while row kh's MFMAs issue was tried: no gain — the SIMD's other wave already covers the LDS latency —
        // and 70 more VGPRs)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i >= nk) break;   // (wave-uniform)
            load_row(xbuf[0], k0 + i);
            compute(xbuf[0], k0 + i);
        }
    };

    // ---- prologue: plane 0 into slot 0, plane 1 in flight ----
    issue_prefetch(0);
    write_plane(0);
    if (NPL > 1) issue_prefetch(1);

    for (int P = 0; P < NPL; ++P) {
        const int hh = (int)blockIdx.x + (P >> 3) * (int)gridDim.x, od = P & 7;
        const int tile = hh >> 1, jj = j16 + 16 * (hh & 1);
#pragma unroll
        for (int ow = 0; ow < OWN; ++ow) acc[ow] = (f32x4){0, 0, 0, 0};
        if (od > 0) taps(slots + ((P - 1) & 1) * 4096, 0);
        // barriers order LDS traffic only: wait for this wave's LDS operations (lgkmcnt), NOT for its global stores / prefetches in
        // flight (vmcnt) as __syncthreads() would — the epilogue's stores of the previous plane are still on their way here
        if (!(ABL & 1)) lds_barrier();   // every wave is done with plane od-1 (and, at od = 0, plane 0 written before this barrier is visible)
        if (P + 1 < NPL && !(ABL & 4)) {
            write_plane(P + 1);                      // ... into the slot plane od-1 just left (next half tile's plane 0 at od = 7)
            if (P + 2 < NPL) issue_prefetch(P + 2);   // two planes ahead: a whole plane of MFMAs hides the latency
        }
        f32x4 sk[RESID ? OWN : 1];
        if (RESID) {   // residual input of this row, consumed in the epilogue
            const vq_buf skb = buf_of((const f32x4*)A.skip + (size_t)tile * 512 * 4 * 32 + 16 * (hh & 1));
#pragma unroll
            for (int ow = 0; ow < OWN; ++ow) sk[ow] = buf_ld16(skb, lane_b, (unsigned)((od * 8 + oh) * 8 + ow0 + ow) * 2048u);
        }
        taps(slots + (P & 1) * 4096, 1);
        if (!(ABL & 1)) lds_barrier();   // plane od+1 is visible
        if (od < 7) taps(slots + ((P + 1) & 1) * 4096, 2);

        // ---- epilogue: this wave's positions of row (od, oh), ascending ----
        f32x4* out4 = (f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj;
        const vq_buf outb = buf_of((f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + 16 * (hh & 1));
#pragma unroll
        for (int ow = 0; ow < OWN; ++ow) {
            f32x4 v = acc[ow] + bias4;
            if (RESID) {
                const f32x4 u = v * 0.1f;
                v = sk[ow] + u;
            }
            if (!(ABL & 2)) buf_st16_nt(v, outb, lane_b, (unsigned)((od * 8 + oh) * 8 + ow0 + ow) * 2048u);
            if ((ABL & 2) && ow == 0 && v.x == 12345.678f) out4[0] = v;   // keep the accumulators alive
            if (STATS && !(ABL & 2)) {
                constexpr int HP = OWN / HW;   // positions per half row of this wave (4)
                st[ow / HP][0].add(v.x);
                st[ow / HP][0].add(v.y);
                st[ow / HP][1].add(v.z);
                st[ow / HP][1].add(v.w);
            }
        }
        if (STATS && !(ABL & 2)) {
            // HALF a row (4 positions) is one statistics block; a wave owns half row (oh, hw) of every plane (both halves with 8 waves),
            // and the contract adds the 128 half-row blocks row-major: t_(oh,hw) = sum over od (from zero, od ascending) lives in this
            // wave's registers, the sixteen t go to the 16 blocks of the partial buffer and gn_combine_k<false> adds them in order
#pragma unroll
            for (int h = 0; h < HW; ++h) st[h][0].fold(), st[h][1].fold();
            if (od == 7) {
#pragma unroll
                for (int h = 0; h < HW; ++h)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int blk = NW == 8 ? 2 * oh + h : wave;
                        A.part_s[part_index(tile, blk, 2 * q4 + k, jj)] = st[h][k].s, A.part_q[part_index(tile, blk, 2 * q4 + k, jj)] = st[h][k].q;
                        st[h][k].init();
                    }
            }
        }
    }
}
