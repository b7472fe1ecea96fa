// vq_convdown_lds.h — the encoder's down conv, Conv3d(16->32,k4,s2,p1) 8^3 -> 4^3 (VQVAE_v2.py:239), with the INPUT PLANES streamed
// through LDS exactly once (round 4).
//
// conv_rows16_k (weights LDS-resident, 128 KB) gives every wave a half tile and walks (output row, (kd,kh)) steps: each of the 64
// input rows of a leaf is loaded 3.06 times, every re-load a miss (L2 hit rate 0.08, 6.6 GB fetched per 65 536-leaf launch for a
// 2.15 GB input: profiles/r03_v3_pmc_l2_hit_miss.txt).  Here the INPUT planes are the outer loop.  Input plane id feeds exactly two
// output planes: od_a = (id+1)/2 with kd = (id+1)%2 and od_b = od_a - 1 with kd + 2.  A workgroup of 8 waves owns a 16-leaf half tile;
// wave = (parity g of the output planes it owns, output row oh): per input plane it has ONE (od, kd) to do — od of parity g among
// {od_a, od_b} — and keeps that row's accumulators (4 positions x 2 cout tiles) over the four input planes 2od-1 .. 2od+2 that feed
// it, so kd arrives ascending.  The planes travel global -> registers -> LDS once (two slots of 64 KB, plane Q in slot Q & 1, the
// next plane written while this one is read; one barrier per input plane; the stream runs across half-tile boundaries), the weights
// (128 KB, the same for every workgroup: L1 / L2 hits) come straight from global memory as MFMA A fragments.
//
// One step = (kd, kh): input row ih = 2 oh - 1 + kh (8 positions, LDS) and the 4 kw x 2 cout-tile fragments; 14 (ow, kw) pairs
// (iw = 2 ow - 1 + kw inside the row) x 8 MFMAs, kw-major, the two cout tiles of a pair alternating; every operand is re-requested
// for the next step right after its last use.  Per output the taps arrive in ascending (kd, kh, kw) order, the 16 channels of a tap
// in "P16" order: conv_rows16_k's arithmetic and the oracle's, bit for bit.  Border rows (oh = 0, 3: three of four kh) share a
// SIMD with inner rows.
//
// Statistics of the output (GroupNorm(8,32) of the residual block that follows): one output row = one block of the 16-block
// contract; a wave keeps its two rows' block sums, all waves pass them through LDS at the end of the half tile and wave 0 adds the
// sixteen in block order after the next barrier.
#pragma once
#include "vq_conv8_lds.h"

constexpr size_t LDS_CONVDOWN = (size_t)2 * 4096 * 16 + (size_t)8 * 2 * 2 * 2 * 64 * 8;   // 131 072 + 32 768 B = all 160 KB
static_assert(LDS_CONVDOWN <= 160 * 1024, "gfx950: 160 KB of LDS per workgroup, all of it dynamic here: the kernel must stay free of static __shared__");

// ABL (tools/ablate only): 1 no barriers, 2 no epilogue, 4 no plane write / prefetch, 8 no LDS B reads, 16 no A-fragment loads
// NW: 8 waves (a wave holds both 16-cout tiles of its row: 8 accumulators) or 16 (one tile each: four waves per SIMD hide one another's
// staging, epilogue and barrier skew; the A-fragment traffic is the same, the B operands are read twice).
template <int ABL = 0, bool TWOB = false, int STAGE_AT = 1, int NW = 16>
__global__ __launch_bounds__(NW * 64, 1) void conv_down_lds_k(ConvArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* slots = (f32x4*)smem_raw;                                    // [2][64 pos][4 quads][16 leaves]
    double* xch = (double*)(smem_raw + (size_t)2 * 4096 * 16);          // [wave NW][row 2][tile of the wave MTW][2][64 lanes]
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    constexpr int MTW = NW == 8 ? 2 : 1, NPW = 64 / NW;                  // cout tiles per wave; positions a wave stages per plane
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, j16 = lane & 15;
    // waves w and w+4 share a SIMD: (g, oh) = (0, w) and (1, perm(w-4)), perm = 1,0,3,2 -> one border and one inner row per SIMD
    const int w8 = wave & 7, mt0 = NW == 8 ? 0 : wave >> 3;              // (16 waves: waves w and w+8 = the two cout tiles of one row)
    const int g = w8 >> 2, oh = g ? ((w8 & 3) ^ 1) : w8;
    const int k0 = oh == 0 ? 1 : 0, nk = (oh == 0 || oh == 3) ? 3 : 4;   // valid kh (ih = 2 oh - 1 + kh in 0..7): k0 .. k0 + nk - 1
    const int n_half = 2 * A.n_tiles;
    if ((int)blockIdx.x >= n_half) return;
    const int n_my = (n_half - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int NPL = n_my * 8;                                             // input planes this workgroup walks
    const unsigned lane_b = (unsigned)(q4 * 32 + j16) * 16u;
    f32x4 bias4[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) bias4[m] = ((const f32x4*)A.bias_frag)[4 * (mt0 + m) + q4];
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;

    // ---- plane staging: wave w brings in NPW consecutive positions of every plane: a plain copy, the input is raw ----
    f32x4 pf[NPW];
    auto issue_prefetch = [&](int Q) __attribute__((always_inline)) {
        const int hh = (int)blockIdx.x + (Q >> 3) * (int)gridDim.x, id = Q & 7;
        const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)(hh >> 1) * 512 * 4 * 32 + 16 * (hh & 1));
#pragma unroll
        for (int k = 0; k < NPW; ++k) pf[k] = buf_ld16(inb, lane_b, (unsigned)(id * 64 + wave * NPW + k) * 2048u);
    };
    auto write_plane = [&](int Q) __attribute__((always_inline)) {
        f32x4* dst = slots + (Q & 1) * 4096 + ((wave * NPW) * 4 + q4) * 16 + j16;
#pragma unroll
        for (int k = 0; k < NPW; ++k) dst[k * 4 * 16] = pf[k];
    };

    // which (od, kd) of this wave input plane id feeds: od of parity g among od_a = (id+1)/2 (kd = (id+1)%2) and od_a - 1 (kd + 2)
    auto od_of = [&](int id) { const int oa = (id + 1) >> 1; return (oa & 1) == g ? oa : oa - 1; };
    auto kd_of = [&](int id) { const int oa = (id + 1) >> 1; return ((id + 1) & 1) + ((oa & 1) == g ? 0 : 2); };
    auto works = [&](int id) { const int o = od_of(id); return o >= 0 && o <= 3; };

    f32x4 acc[4][MTW];
    // Operands of one (kd, kh) step: a[kw][mt] = A fragments of taps (kd*4 + kh)*4 + kw (global memory: L1 / L2 hits), x[iw] = the 8
    // positions of input row ih = 2 oh - 1 + kh of the plane (LDS).  One register set; every operand is re-requested for the next step
    // of the stream (this plane's next kh, else the first kh of this wave's next working plane) right after its last MFMA.  Requests
    // are unconditional.  The LDS re-requests at a plane's last step read a slot that is still being written: they are repeated
    // after the barrier (x_all), the early copy is never used.
    f32x4 a[4][MTW], x[8];
    int rq = 0, ri = 0;                 // the step being requested: input plane, kh - k0
    auto advance = [&]() __attribute__((always_inline)) {
        if (++ri == nk) {
            ri = 0;
            ++rq;
            if (!works(rq & 7)) ++rq;   // (a wave idles at id = 0 (g = 1) or id = 7 (g = 0): never two planes in a row)
        }
    };
    auto ld_a = [&](int kw) __attribute__((always_inline)) {
        const unsigned t = (unsigned)((kd_of(rq & 7) * 4 + k0 + ri) * 4 + kw);
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[kw][m] = (ABL & 16) ? bias4[m] : buf_ld16(wb, lane_w, (t * 2 + mt0 + m) * 1024u);
    };
    auto ld_x = [&](int iw) __attribute__((always_inline)) {
        x[iw] = (ABL & 8) ? bias4[0] : slots[(rq & 1) * 4096 + (((2 * oh - 1 + k0 + ri) * 8 + iw) * 4 + q4) * 16 + j16];
    };
    // The step's fourteen (ow, kw) pairs in kw-major order — kw 0: ow 1,2,3 | kw 1: ow 0..3 | kw 2: ow 0..3 | kw 3: ow 0,1,2: every accumulator
    // sees its kw ascending — run as seven groups of two pairs whose MFMAs alternate (different outputs: independent accumulators), each
    // pair's MFMAs in k order.  Every operand is re-requested for the next step right after its last use, at least two groups (~1 k
    // cycles) before its first use there:
    //   G1 (1,0)(2,0) | G2 (3,0)(0,1): a0 x0 | G3 (1,1)(2,1) | G4 (3,1)(0,2): a1 x1 | G5 (1,2)(2,2): x3 x5 | G6 (3,2)(0,3): a2 x7 x2 | G7 (1,3)(2,3): a3 x4 x6
    auto group = [&](int owA, int kwA, int owB, int kwB) __attribute__((always_inline)) {
        const int iwA = 2 * owA - 1 + kwA, iwB = 2 * owB - 1 + kwB;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                acc[owA][m] = mfma16(a[kwA][m][k], x[iwA][k], acc[owA][m]);
                acc[owB][m] = mfma16(a[kwB][m][k], x[iwB][k], acc[owB][m]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&]() __attribute__((always_inline)) {
        advance();   // (rq, ri): the NEXT step — what the re-requests below fetch
        group(1, 0, 2, 0);
        group(3, 0, 0, 1);
        ld_a(0), ld_x(0);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 1, 2, 1);
        group(3, 1, 0, 2);
        ld_a(1), ld_x(1);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 2, 2, 2);
        ld_x(3), ld_x(5);
        __builtin_amdgcn_sched_barrier(0);
        group(3, 2, 0, 3);
        ld_a(2), ld_x(7), ld_x(2);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 3, 2, 3);
        ld_a(3), ld_x(4), ld_x(6);
        __builtin_amdgcn_sched_barrier(0);
    };

    // (the block sums of this wave's two rows (od = g, g + 2) go to LDS as each row completes: the readers come two barriers later at the earliest)
    auto finish_stats = [&](int hh) __attribute__((always_inline)) {   // the row (g, oh) = (0, 0) wave(s): the sixteen blocks of half tile hh in block order
        if (w8 != 0) return;
        const int tile = hh >> 1, jj = j16 + 16 * (hh & 1);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = mt0 + m;
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                const int o = blk & 3, d = blk >> 2;                  // block = output row (od = d, oh = o)
                const int wv = ((d & 1) ? 4 + (o ^ 1) : o) + (NW == 8 ? 0 : 8 * mt);   // its wave: g = d & 1, oh = o (, cout tile)
                S += xch[(((wv * 2 + (d >> 1)) * MTW + m) * 2 + 0) * 64 + lane];
                Q += xch[(((wv * 2 + (d >> 1)) * MTW + m) * 2 + 1) * 64 + lane];
            }
            float mm, r;
            gn_finish(S, Q, 1.0 / 256.0, mm, r);   // GroupNorm(8,32): 4 channels x 64 positions
            A.out_mean[((size_t)tile * 8 + 4 * mt + q4) * 32 + jj] = mm;
            A.out_rstd[((size_t)tile * 8 + 4 * mt + q4) * 32 + jj] = r;
        }
    };

    // ---- prologue: plane 0 into slot 0, plane 1 in flight; the first step's operands ----
    issue_prefetch(0);
    write_plane(0);
    if (NPL > 1) issue_prefetch(1);
    rq = works(0) ? 0 : 1;
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) ld_a(kw);

    for (int Q = 0; Q < NPL; ++Q) {
        const int hh = (int)blockIdx.x + (Q >> 3) * (int)gridDim.x, id = Q & 7;
        const int tile = hh >> 1;
        // every wave is done with plane Q-1 (its slot is free) and plane Q, written during plane Q-1, is visible
        if (!(ABL & 1)) lds_barrier();
        if (id == 0 && Q > 0) finish_stats(hh - (int)gridDim.x);
        // staging of plane Q+1 (registers -> LDS) and the requests for plane Q+2: at the top of the plane (STAGE_AT 0) or behind the wave's
        // first step (1) — the wait for the staged registers is a vmcnt(0) (the number of younger requests varies), and at the top it also
        // waits for the write acknowledgements of the epilogue that has just run
        auto stage = [&]() __attribute__((always_inline)) {
            if (ABL & 4) return;
            if (Q + 1 < NPL) write_plane(Q + 1);
            issue_prefetch(Q + 2 < NPL ? Q + 2 : NPL - 1);
        };
        if (STAGE_AT == 0 || TWOB || !works(id)) stage();
        // a second barrier right behind the staging (the waves have just met: little skew): plane Q+1 is visible from HERE, so the last
        // step of this plane can already request the first B operands of the next one
        if (TWOB && !(ABL & 1)) lds_barrier();
        if (!works(id)) continue;   // (wave-uniform)
        const int od = od_of(id), kd = kd_of(id);
        if (kd == 0 || (od == 0 && kd == 1)) {   // first plane of this output row
#pragma unroll
            for (int ow = 0; ow < 4; ++ow)
#pragma unroll
                for (int m = 0; m < MTW; ++m) acc[ow][m] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
        // (rq, ri) = this plane's first step.  One barrier per plane: its B operands can only be requested now (the early copy read a slot
        // that was still being written).  Two: only behind a plane this wave sat out (the early copy was requested two planes ahead).
        if (!TWOB || Q == 0 || !works((id + 7) & 7)) {
#pragma unroll
            for (int iw = 0; iw < 8; ++iw) ld_x(iw);
        }
        step();
        if (STAGE_AT == 1 && !TWOB) stage();
        for (int st = 1; st < nk; ++st) step();
        if (!(kd == 3 || (od == 3 && kd == 2))) continue;   // the row is complete after its last plane
        // ---- epilogue: the row's 4 positions, ascending; the two cout tiles ----
        const vq_buf outb = buf_of((const f32x4*)A.out + (size_t)tile * 64 * 8 * 32 + 16 * (hh & 1));
        GnAcc st[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) st[m].init();
        if (!(ABL & 2)) {
#pragma unroll
            for (int ow = 0; ow < 4; ++ow)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const f32x4 v = acc[ow][m] + bias4[m];
                    buf_st16_nt(v, outb, lane_b + (mt0 + m) * 2048, (unsigned)((od * 4 + oh) * 4 + ow) * 4096u);
                    st[m].add(v.x);
                    st[m].add(v.y);
                    st[m].add(v.z);
                    st[m].add(v.w);
                }
        } else {
            float t = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 4; ++ow) t += acc[ow][0].x + acc[ow][MTW - 1].w;
            if (t == 12345.678f) ((f32x4*)A.out)[tid] = acc[0][0];
        }
        // this row's block: read by the (0, 0) waves after the first barrier of the next half tile (od = g completes at id = 2 + 2g, at
        // least two barriers after that reader ran for the previous half tile)
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            xch[(((wave * 2 + (od >> 1)) * MTW + m) * 2 + 0) * 64 + lane] = st[m].bs;
            xch[(((wave * 2 + (od >> 1)) * MTW + m) * 2 + 1) * 64 + lane] = st[m].bq;
        }
    }
    lds_barrier();
    finish_stats((int)blockIdx.x + (n_my - 1) * (int)gridDim.x);
}
