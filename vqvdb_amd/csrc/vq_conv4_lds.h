// vq_conv4_lds.h — Conv3d(32->32,k3,p1) @4^3 of the encoder's ResidualBlock(32) (VQVAE_v2.py:190-210, :240) with the INPUT PLANES
// staged in LDS (round 4).
//
// conv_rows16_k keeps this layer's weights in LDS (108 KB) and gives every wave a half tile of its own: each input row is then
// re-loaded 6.25 times, every re-load a miss (between two uses of a row by one wave the 256 waves of an XCD pull ~37 MB through a
// 4 MB L2: hit rate 0.12-0.14, 3.3 / 3.9 GB fetched per 65 536-leaf launch for a 0.54 GB input, profiles/r03_v3_pmc_l2_hit_miss.txt).
// LDS holds the weights OR a half tile's input (128 KB), not both.  Here it holds the input, and the weights — which every workgroup
// of the chip reads in the same order, 108 KB that never leave the L2 / the CU's L1 — come straight from global memory as MFMA A
// fragments, one 1 KiB buffer load per 13 MFMAs:
//
//   LDS: ring of FOUR plane slots of 32 KB ([pos 16][quad 8][leaf 16] float4), plane P of the workgroup's plane stream in slot P & 3;
//        32 KB exchange buffer for the block statistics.  GroupNorm + ReLU is applied ONCE per element, on the way into LDS
//        (the row kernel re-applies it on each of the 6.25 loads: 2.5 % of its time, tools/ablate/conv_rows16_ablate.hip ABL 16).
//   per output plane P:  barrier | plane P+2 (prefetched during plane P-1) -> slot of plane P-2 | global loads of plane P+3 |
//                        taps kd = 0 (plane P-1), 1 (plane P), 2 (plane P+1) | epilogue
//   ONE barrier per plane; the plane stream runs straight across half-tile boundaries (the workgroup is persistent).
//
// A workgroup of 8 waves owns a 16-leaf half tile; wave = (output row oh of the plane, 16-cout tile mt): 4 accumulators (the row's
// 4 positions).  Per (kd, kh): the 4 input positions x 2 channel blocks of row oh+kh-1 come from LDS (8 conflict-free ds_read_b128),
// the 3 kw x 2 channel-block A fragments from global memory (6 buffer loads, L1 / L2 hits), then 80 MFMAs, kw-outer, consecutive
// MFMAs on different accumulators.  Per output the taps arrive in ascending (kd, kh, kw) order, inside a tap the channel blocks
// ascending, inside a block "P16": conv_rows16_k's arithmetic and the oracle's, bit for bit.  Border rows (oh = 0, 3: two of three
// kh) share a SIMD with inner rows (waves w and w+4 of a workgroup land on one SIMD), so every SIMD carries the same MFMA work.
//
// Statistics: one output row = one block of the 16-block contract.  A wave passes the block sums of its rows through LDS to the wave
// oh = 0 of its cout tile, which adds the sixteen in block order (GnAcc::fold's chain) after the first barrier of the next half tile and
// finishes mean / rstd (STATS, conv1) or the channel sums (CSUM, conv2).
#pragma once
#include "vq_conv8_lds.h"

constexpr size_t LDS_CONV4 = (size_t)4 * 2048 * 16 + (size_t)2 * 4 * 4 * 2 * 64 * 8;   // 131 072 + 32 768 B = all 160 KB
static_assert(LDS_CONV4 <= 160 * 1024, "gfx950: 160 KB of LDS per workgroup, all of it dynamic here: the kernel must stay free of static __shared__");

// ABL (tools/ablate only): 1 no barriers, 2 no epilogue, 4 no plane write / prefetch, 8 no LDS B reads, 16 no A-fragment loads, 32 no MFMAs
// STG: who stages the planes.  0: every wave two positions, the two waves of a SIMD at different points of the plane.  1: only the four
// border-row waves (oh = 0, 3: two of three kh, a third less matrix work than the inner-row wave they share a SIMD with), four positions
// each, at the top of the plane — the HBM round trips land in their slack, and the inner-row waves have no HBM request in their queue
// (conv1) or only the residual row's (conv2).
template <bool RESID, bool STATS, bool CSUM, int ABL = 0, int STG = 1>
__global__ __launch_bounds__(512, 1) void conv4_lds_k(ConvArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    static_assert(!(STATS && CSUM), "conv1 carries GroupNorm statistics, conv2 channel sums");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* slots = (f32x4*)smem_raw;                                   // [4][16 pos][8 quads][16 leaves]
    unsigned char* xch = smem_raw + (size_t)4 * 2048 * 16;              // [mt 2][oh 4][od 4][2][64 lanes] x 8 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, j16 = lane & 15;
    // waves w and w+4 share a SIMD: (oh, mt) = (w, 0) and (perm(w-4), 1) with perm = 1,0,3,2 -> one border and one inner row per SIMD
    const int mt = wave >> 2, oh = mt ? ((wave & 3) ^ 1) : wave;
    const int k0 = oh == 0 ? 1 : 0, nk = (oh == 0 || oh == 3) ? 2 : 3;   // valid kh: k0 .. k0 + nk - 1
    const int n_half = 2 * A.n_tiles;
    if ((int)blockIdx.x >= n_half) return;
    const int n_my = (n_half - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int NPL = n_my * 4;
    const unsigned lane_b = (unsigned)(q4 * 32 + j16) * 16u;              // this lane inside a [quad 4][32 leaves] float4 block of the L4 layout
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[4 * mt + q4];
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;
    float gam[2][4], bet[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) gam[cb][i] = A.in_gamma[16 * cb + 4 * q4 + i], bet[cb][i] = A.in_beta[16 * cb + 4 * q4 + i];

    // ---- plane staging: a staging wave brings in NPP consecutive positions (x 2 channel blocks) of every plane ----
    constexpr int NPP = STG == 1 ? 4 : 2, NPF = 2 * NPP;
    const bool border = oh == 0 || oh == 3;                               // (waves 0, 3, 5, 6)
    const int pos0 = STG == 1 ? 4 * ((oh == 3 ? 1 : 0) + 2 * mt) : 2 * wave;   // first position this wave stages
    f32x4 pf[NPF];
    float pm[2], pr[2];
    auto issue_prefetch = [&](int P) __attribute__((always_inline)) {
        const int hh = (int)blockIdx.x + (P >> 2) * (int)gridDim.x, id = P & 3;
        const int tile = hh >> 1, jj = j16 + 16 * (hh & 1);
        const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * 64 * 8 * 32 + 16 * (hh & 1));
#pragma unroll
        for (int k = 0; k < NPF; ++k) pf[k] = buf_ld16(inb, lane_b + (k & 1) * 2048, (unsigned)(id * 16 + pos0 + (k >> 1)) * 4096u);
        // GroupNorm(8,32): the group of quad 4cb + q4 is the quad itself.  The statistics change with the half tile only, but they travel with
        // EVERY plane: a load inside a branch makes the compiler count the loads in flight for the worse of the two paths, and every
        // fragment wait of the plane would then also wait for part of this HBM prefetch
        const vq_buf mb = buf_of(A.in_mean + (size_t)tile * 8 * 32), rb = buf_of(A.in_rstd + (size_t)tile * 8 * 32);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) pm[cb] = buf_ld4(mb, (unsigned)((4 * cb + q4) * 32 + jj) * 4u, 0), pr[cb] = buf_ld4(rb, (unsigned)((4 * cb + q4) * 32 + jj) * 4u, 0);
    };
    float ia[2][4], ib[2][4];
    auto write_plane = [&](int P) __attribute__((always_inline)) {
        if ((P & 3) == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ia[cb][i] = pr[cb] * gam[cb][i];
                    ib[cb][i] = __builtin_fmaf(-pm[cb], ia[cb][i], bet[cb][i]);
                }
        }
        f32x4* dst = slots + (P & 3) * 2048 + q4 * 16 + j16;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int cb = k & 1;
            f32x4 v = pf[k];
            v.x = fmaxf(__builtin_fmaf(v.x, ia[cb][0], ib[cb][0]), 0.0f);
            v.y = fmaxf(__builtin_fmaf(v.y, ia[cb][1], ib[cb][1]), 0.0f);
            v.z = fmaxf(__builtin_fmaf(v.z, ia[cb][2], ib[cb][2]), 0.0f);
            v.w = fmaxf(__builtin_fmaf(v.w, ia[cb][3], ib[cb][3]), 0.0f);
            dst[((pos0 + (k >> 1)) * 8 + 4 * cb) * 16] = v;
        }
    };

    f32x4 acc[4];
    // Operands of one (kd, kh) step: a[kw][cb] = the A fragments of taps (kd*3 + kh)*3 + kw for this wave's cout tile (global memory:
    // L1 / L2 hits), x[iw][cb] = the 4 positions x 2 channel blocks of input row ih = oh + kh - 1 of the plane in `slot` (LDS).
    // ONE register set: every operand is re-requested for the NEXT step of the stream right after its last MFMA of this step (below).
    // The stream runs plane after plane; every request is unconditional (a request inside a branch makes the compiler count the
    // operations in flight for the worse path, and every operand wait becomes vmcnt(0)).
    f32x4 a[3][2], x[4][2];
    int rp = 0, rkd = 1, ri = 0;      // the step whose operands are being requested: plane, kd, kh - k0
    auto advance = [&]() __attribute__((always_inline)) {
        if (++ri == nk) {
            ri = 0;
            if (++rkd > ((rp & 3) == 3 ? 1 : 2)) ++rp, rkd = (rp & 3) == 0 ? 1 : 0;   // zero padding along d: kd = 1.. at od = 0, ..1 at od = 3
        }
    };
    auto ld_a = [&](int kw) __attribute__((always_inline)) {
        const unsigned t = (unsigned)((rkd * 3 + k0 + ri) * 3 + kw);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) a[kw][cb] = (ABL & 16) ? bias4 : buf_ld16(wb, lane_w, ((t * 2 + cb) * 2 + mt) * 1024u);
    };
    auto ld_x = [&](int iw) __attribute__((always_inline)) {
        const f32x4* src = slots + ((rp + rkd - 1) & 3) * 2048 + (((oh + k0 + ri - 1) * 4 + iw) * 8 + q4) * 16 + j16;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) x[iw][cb] = (ABL & 8) ? bias4 : src[4 * cb * 16];
    };
    // The step's ten (output ow, tap kw) pairs — input position iw = ow + kw - 1 inside the row — run as five groups of two pairs whose
    // MFMAs alternate (two accumulators: the dependent-issue latency of the 16x16x4 MFMA is hidden), each pair's 8 MFMAs in (cb, k)
    // order.  Every accumulator sees its kw ascending.  The order lets each operand be re-requested for the next step right after its
    // last use with at least one group (512 cycles: LDS) or two (global memory) to go before its first use there:
    //   G1 (1,0) (2,0) | G2 (3,0) (0,1): a[0] x[0] done | G3 (1,1) (2,1) | G4 (3,1) (0,2): a[1] x[1] done | G5 (1,2) (2,2): a[2] x[2] x[3] done
    auto group = [&](int owA, int kwA, int owB, int kwB) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ABL & 32) {
                    acc[owA].x += a[kwA][cb][k] + x[owA + kwA - 1][cb][k], acc[owB].x += a[kwB][cb][k] + x[owB + kwB - 1][cb][k];
                } else {
                    acc[owA] = mfma16(a[kwA][cb][k], x[owA + kwA - 1][cb][k], acc[owA]);
                    acc[owB] = mfma16(a[kwB][cb][k], x[owB + kwB - 1][cb][k], acc[owB]);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&]() __attribute__((always_inline)) {
        advance();   // (rp, rkd, ri): the NEXT step — what the re-requests below fetch
        group(1, 0, 2, 0);
        group(3, 0, 0, 1);
        ld_a(0);
        ld_x(0);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 1, 2, 1);
        group(3, 1, 0, 2);
        ld_a(1);
        ld_x(1);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 2, 2, 2);
        ld_a(2);
        ld_x(2);
        ld_x(3);
        __builtin_amdgcn_sched_barrier(0);
    };

    // block sums of this wave's four rows of the current half tile
    // A row's block travels through LDS ([mt 2][oh 4][od 4] slots) to the wave oh = 0 of its cout tile, which reads the sixteen after
    // the first barrier of the NEXT half tile.  The block of plane od is written one plane late (at the end of plane od + 1; od = 3: at
    // once), so no slot of the next half tile is written before that reader's plane is over: one block in registers, not four.
    double hold_s = 0.0, hold_q = 0.0;
    f32x4 hold_c = {0.0f, 0.0f, 0.0f, 0.0f};
    // the ordered sum over the sixteen blocks of half tile hh (wave oh = 0 of each cout tile; the others' blocks come through LDS)
    auto finish_stats = [&](int hh) __attribute__((always_inline)) {
        if (oh != 0) return;   // (wave-uniform)
        const int tile = hh >> 1, jj = j16 + 16 * (hh & 1);
        if (STATS) {
            const double* xs = (const double*)xch + (size_t)mt * 4 * 4 * 2 * 64;
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                const int o = blk & 3, d = blk >> 2;
                S += xs[((o * 4 + d) * 2 + 0) * 64 + lane];
                Q += xs[((o * 4 + d) * 2 + 1) * 64 + lane];
            }
            float m, r;
            gn_finish(S, Q, 1.0 / 256.0, m, r);   // GroupNorm(8,32): 4 channels x 64 positions
            A.out_mean[((size_t)tile * 8 + 4 * mt + q4) * 32 + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 4 * mt + q4) * 32 + jj] = r;
        }
        if (CSUM) {
            const f32x4* xs = (const f32x4*)xch + (size_t)mt * 4 * 4 * 64;
            f32x4 cs = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                const int o = blk & 3, d = blk >> 2;
                cs = cs + xs[(o * 4 + d) * 64 + lane];
            }
            const float v[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) A.out_csum[((size_t)tile * 32 + 16 * mt + 4 * q4 + r) * 32 + jj] = v[r];
        }
    };

    // ---- prologue: planes 0 and 1 into their slots, plane 2 in flight ----
    if (STG == 0 || border) {
        issue_prefetch(0);
        write_plane(0);
        if (NPL > 1) {
            issue_prefetch(1);
            write_plane(1);
        }
        if (NPL > 2) issue_prefetch(2);
    }
    lds_barrier();           // planes 0 and 1 are visible
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) ld_a(kw);      // plane 0 (od = 0): kd starts at 1
#pragma unroll
    for (int iw = 0; iw < 4; ++iw) ld_x(iw);

    // The persistent plane loop exists twice, once per staging point (see `stage` below): with both copies of the plane body inside ONE loop the
    // compiler reconciles the operand registers of the two paths by copying all 56 of them at every plane boundary — behind a wait for
    // the next plane's operands, requested on purpose one step early.
    auto run = [&](auto ROLE_C) __attribute__((always_inline)) {
    constexpr bool ROLE = decltype(ROLE_C)::value;   // STG 0: stages after the plane's first step (else after its third); STG 1: stages at all
    for (int P = 0; P < NPL; ++P) {
        const int hh = (int)blockIdx.x + (P >> 2) * (int)gridDim.x, od = P & 3;
        const int tile = hh >> 1;
        // every wave is done with plane P-1's taps (so the slot of plane P-2 is free) and plane P+1, written during plane P-1, is visible
        if (!(ABL & 1)) lds_barrier();
        if (od == 0 && P > 0 && (STATS || CSUM)) finish_stats(hh - (int)gridDim.x);
        const int ns = ((od == 0 || od == 3) ? 2 : 3) * nk;   // kd x kh steps of this plane (4, 6 or 9)
        f32x4 sk[RESID ? 4 : 1];
        const vq_buf skb = buf_of(RESID ? (const f32x4*)A.skip + (size_t)tile * 64 * 8 * 32 + 16 * (hh & 1) : (const f32x4*)A.in);
#pragma unroll
        for (int ow = 0; ow < 4; ++ow) acc[ow] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        // Staging of plane P+2 (registers -> GroupNorm + ReLU -> LDS) and the HBM requests for plane P+3 (index clamped) and the residual
        // row.  Vector-memory operations return IN ORDER: the first operand wait behind these requests also waits for their HBM round trip.
        // The persistent loop exists once per role (straight-line code each: a request inside a branch would make the compiler count the
        // operations in flight for the worse path and turn every operand wait into vmcnt(0); two roles inside one loop make it copy all
        // 56 operand registers at every plane boundary).
        auto stage = [&]() __attribute__((always_inline)) {
            if (ABL & 4) return;
            if (P + 2 < NPL) write_plane(P + 2);   // (LDS stores only)
            issue_prefetch(P + 3 < NPL ? P + 3 : NPL - 1);
        };
        auto residual = [&]() __attribute__((always_inline)) {
            if (RESID) {
#pragma unroll
                for (int ow = 0; ow < 4; ++ow) sk[ow] = buf_ld16(skb, lane_b + mt * 2048, (unsigned)((od * 4 + oh) * 4 + ow) * 4096u);
            }
        };
        if (STG == 1) {
            if (ROLE) stage();
            residual();
            for (int st = 0; st < ns; ++st) step();
        } else {
            step();
            if (ROLE) stage(), residual();
            step(), step();                  // (every plane has at least four steps)
            if (!ROLE) stage(), residual();
            for (int st = 3; st < ns; ++st) step();
        }
        // ---- epilogue: the row's 4 positions, ascending ----
        const vq_buf outb = buf_of((const f32x4*)A.out + (size_t)tile * 64 * 8 * 32 + 16 * (hh & 1));
        GnAcc st;
        st.init();
        f32x4 csb = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!(ABL & 2)) {
#pragma unroll
            for (int ow = 0; ow < 4; ++ow) {
                f32x4 v = acc[ow] + bias4;
                if (RESID) {
                    const f32x4 u = v * 0.1f;
                    v = sk[ow] + u;
                }
                buf_st16_nt(v, outb, lane_b + mt * 2048, (unsigned)((od * 4 + oh) * 4 + ow) * 4096u);
                if (STATS) {
                    st.add(v.x);
                    st.add(v.y);
                    st.add(v.z);
                    st.add(v.w);
                }
                if (CSUM) csb = csb + v;
            }
        } else {
            float t = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 4; ++ow) t += acc[ow].x + acc[ow].w;
            if (t == 12345.678f) ((f32x4*)A.out)[tid] = acc[0];
        }
        // this row's block goes to LDS one plane late (see hold_*)
        if (STATS) {
            double* xs = (double*)xch + (size_t)mt * 4 * 4 * 2 * 64;
            if (od > 0) xs[((oh * 4 + od - 1) * 2 + 0) * 64 + lane] = hold_s, xs[((oh * 4 + od - 1) * 2 + 1) * 64 + lane] = hold_q;
            if (od == 3) xs[((oh * 4 + 3) * 2 + 0) * 64 + lane] = st.bs, xs[((oh * 4 + 3) * 2 + 1) * 64 + lane] = st.bq;
            hold_s = st.bs, hold_q = st.bq;
        }
        if (CSUM) {
            f32x4* xs = (f32x4*)xch + (size_t)mt * 4 * 4 * 64;
            if (od > 0) xs[(oh * 4 + od - 1) * 64 + lane] = hold_c;
            if (od == 3) xs[(oh * 4 + 3) * 64 + lane] = csb;
            hold_c = csb;
        }
    }
    };
    if (STG == 1 ? border : mt == 0) run(std::true_type{});
    else run(std::false_type{});
    if (STATS || CSUM) {
        lds_barrier();
        finish_stats((int)blockIdx.x + (n_my - 1) * (int)gridDim.x);
    }
}
