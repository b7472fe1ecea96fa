// vq_stem_taps.h — decoder front of large passes with the (tap, code) table streamed through LDS tap by tap (round 3).
//
// D0-D2 of the decoder: F.embedding + stem Conv3d(128->64,k3,p1) @4^3 (VQVAE_v2.py:257, :371-375) as look-ups in the (tap, code)
// table T[27][256][64] (build_stem_lut_k) + GroupNorm(8,64) + ReLU (:258-259) + the statistics of the result for ResidualBlock.gn1
// (:205) — what stem_fused_k does, same arithmetic contract bit for bit (valid taps ascending, plain adds from 0, + bias; statistics
// by the 16-block rule).  stem_fused_k walks POSITIONS and gathers every tap's table row of a position from L2 through the L1: 256 KB
// per leaf, 16.8 GB per 65 536 leaves, and the L2 -> L1 path delivers about half of its 64 B/clk to such a gather — 0.80 ms, 8 % of
// decode without one MFMA.  Here the TAPS are the outer loop: a persistent workgroup keeps the accumulators of 16 leaves x 64
// positions x 64 channels in registers (128 VGPRs per lane) and streams the 27 table slices of 64 KB (one tap: 256 codes x 64
// channels) through a double-buffered LDS window with wide coalesced loads; the gather itself reads LDS, where the 16 lanes of a
// leaf fetch one whole 256-byte row per access (conflict-free, 128 B/clk).  Table traffic: 1.77 MB per 16 leaves = 7.2 GB per
// 65 536 leaves, every byte of it an L2 hit moved by a full-width load.
//
// MEASURED (r03, 65 536 leaves): 0.91 ms against stem_fused_k's 0.82 ms — NOT the default (VQHIP_STEM=taps selects it; results are
// bit-identical, tests/test_gpu_parity.py::test_large_path_kernel_variants_agree).  What both kernels are bound by is the L2's OUTPUT:
// 16 channels x 64 B/clk = 1 KiB/clk per XCD, 19.7 TB/s over the chip.  The gather pulls 16.8 GB through it (0.85 ms: what it
// takes), this kernel 7.2 GB (0.37 ms) — but here nothing else overlaps with that stream: one workgroup per CU (134 KB of LDS),
// every wave of every CU waiting for the same slice at the same barrier, statistics and stores of a group between two streams.
// 24 leaves per group (12 waves, slices by LDS-DMA instead of through registers) would cut the stream to 4.8 GB; the floor of the
// approach is about 0.5 ms.  Kept as the starting point for that.
//
// Wave w = (leaf quad w / 2, half h = w % 2: positions 32h .. 32h + 31); lane = (leaf l of the quad, 16-byte chunk c): the lane owns
// channels 4c .. 4c + 3 of its leaf at its wave's 32 positions.  Loop nest: (kd, kh) at run time (9 iterations: the eight neighbour
// rows' codes arrive as one aligned dword each from the index block in LDS), kw static (the three neighbour codes of a position
// are bytes of that dword), rows and positions static (accumulator registers need static indices); a row whose neighbour row lies
// outside the leaf is skipped by a wave-uniform branch — zero-padding taps are skipped exactly, as everywhere.
#pragma once
#include "vq_kernels.h"

constexpr size_t LDS_STEM_TAPS = (size_t)2 * 65536 + 16 * 128 + 4 * 64 * 2 * sizeof(double);   // slices | index block with halo | block-sum exchange

__global__ __launch_bounds__(512, 2) void stem_taps_k(StemFusedArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* slice = (f32x4*)smem_raw;                                  // [2][256 codes][16 chunks]
    unsigned char* sidx = smem_raw + 2 * 65536;                      // [16 leaves][128]: the leaf's 64 codes at bytes 32 .. 95 (halo: never used, only read)
    double* xch = (double*)(smem_raw + 2 * 65536 + 16 * 128);        // [4 quads][64 lanes][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = wave >> 1, h = wave & 1;
    const int l = lane >> 4, c = lane & 15;
    const int ls = 4 * quad + l;                                      // leaf slot of this lane in the group of 16
    const f32x4 b4 = ((const f32x4*)A.bias)[c];
    float gam[4], bet[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gam[i] = A.gamma[4 * c + i], bet[i] = A.beta[4 * c + i];
    const vq_buf tb = buf_of(A.T);
    const unsigned lane_t = (unsigned)lane * 16u;                     // this lane's 16 bytes of a 1 KiB piece
    const unsigned nb_base = (unsigned)(ls * 128 + 32 + 32 * h);      // byte offset of this wave's first position's code in sidx
    double* xq = xch + ((size_t)quad * 64 + lane) * 2;
    const int n_groups = 2 * A.n_tiles;

    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int tile = g >> 1, jt = 16 * (g & 1) + ls;
        // ---- the group's codes -> LDS (two bytes per thread) ----
        {
            const int li = tid >> 5, b0 = (tid & 31) * 2;
            const int64_t leaf = (int64_t)tile * 32 + 16 * (g & 1) + li;
            unsigned char v0 = 0, v1 = 0;
            if (leaf < A.n_leaves) v0 = A.idx[leaf * 64 + b0], v1 = A.idx[leaf * 64 + b0 + 1];
            sidx[li * 128 + 32 + b0] = v0, sidx[li * 128 + 32 + b0 + 1] = v1;
        }
        // ---- slice 0 -> window 0 ----
        {
            f32x4 st[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) st[k] = buf_ld16(tb, lane_t, (unsigned)(k * 8 + wave) * 1024u);
#pragma unroll
            for (int k = 0; k < 8; ++k) slice[(k * 8 + wave) * 64 + lane] = st[k];
        }
        __syncthreads();

        f32x4 acc[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) acc[p] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

        for (int kdkh = 0; kdkh < 9; ++kdkh) {
            const int kd = kdkh / 3, kh = kdkh - 3 * kd;
            // the codes of the eight neighbour rows (pd + kd - 1, ph + kh - 1): one aligned dword per row
            unsigned nb[8];
            {
                const unsigned char* src = sidx + nb_base + (kd - 1) * 16 + (kh - 1) * 4;
#pragma unroll
                for (int r = 0; r < 8; ++r) nb[r] = *(const unsigned*)(src + 4 * r);
            }
            // rows of this wave whose neighbour row exists (wave-uniform)
            bool rv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) rv[r] = (unsigned)(2 * h + (r >> 2) + kd - 1) < 4u && (unsigned)((r & 3) + kh - 1) < 4u;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kdkh * 3 + kw;
                const bool more = t < 26;                                  // (uniform) slice t + 1 exists
                f32x4 st[8];
                if (more) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) st[k] = buf_ld16(tb, lane_t, (unsigned)(t + 1) * 65536u + (unsigned)(k * 8 + wave) * 1024u);
                }
                const unsigned wbase = (unsigned)(t & 1) * 65536u + (unsigned)c * 16u;   // window of slice t + this lane's chunk
                const unsigned char* win = smem_raw + wbase;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (rv[r]) {   // (wave-uniform)
#pragma unroll
                        for (int pw = 0; pw < 4; ++pw) {
                            const int qw = pw + kw - 1;
                            if (qw < 0 || qw > 3) continue;   // (static) zero padding along w
                            const unsigned code = (nb[r] >> (8 * qw)) & 0xffu;
                            const f32x4 row = *(const f32x4*)(win + code * 256u);
                            acc[4 * r + pw] = acc[4 * r + pw] + row;
                        }
                    }
                }
                if (more) {
                    f32x4* dst = slice + ((t + 1) & 1) * 4096;
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[(k * 8 + wave) * 64 + lane] = st[k];
                }
                __syncthreads();   // slice t + 1 visible; every wave is done with the window of slice t (slice t + 2 overwrites it)
            }
        }

        // ---- statistics of y = acc + bias: one fp64 chain per row of 4 positions (a statistics block), rows added in order ----
        double bs[8], bq[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            GnAcc st;
            st.init();
#pragma unroll
            for (int pw = 0; pw < 4; ++pw) {
                const f32x4 v = acc[4 * r + pw] + b4;
                acc[4 * r + pw] = v;
                st.add(v.x);
                st.add(v.y);
                st.add(v.z);
                st.add(v.w);
            }
            bs[r] = st.bs, bq[r] = st.bq;
        }
        // blocks 0..7 live in half 0, blocks 8..15 in half 1: the ordered total passes through LDS
        auto ordered_totals = [&](double& S, double& Q) {
            if (h == 0) {
                S = 0.0, Q = 0.0;
#pragma unroll
                for (int r = 0; r < 8; ++r) S += bs[r], Q += bq[r];
                xq[0] = S, xq[1] = Q;
            }
            __syncthreads();
            if (h == 1) {
                S = xq[0], Q = xq[1];
#pragma unroll
                for (int r = 0; r < 8; ++r) S += bs[r], Q += bq[r];
                xq[0] = S, xq[1] = Q;
            }
            __syncthreads();
            if (h == 0) S = xq[0], Q = xq[1];
            __syncthreads();   // (the exchange buffer is reused right away)
        };
        float ia[4], ib[4];
        {
            double S, Q;
            ordered_totals(S, Q);
            const double S2 = __shfl_xor(S, 1, 64), Q2 = __shfl_xor(Q, 1, 64);   // low quad + high quad of the 8-channel group
            float m, r;
            gn_finish((c & 1) ? S2 + S : S + S2, (c & 1) ? Q2 + Q : Q + Q2, 1.0 / 512.0, m, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ia[i] = r * gam[i];
                ib[i] = __builtin_fmaf(-m, ia[i], bet[i]);
            }
        }
        // ---- normalise + ReLU -> d2, statistics of d2 ----
        {
            const vq_buf outb = buf_of((const f32x4*)A.d2 + (size_t)tile * 64 * 16 * 32);
            const vq_buf dbgb = buf_of(A.ystem_dbg ? (const f32x4*)A.ystem_dbg + (size_t)tile * 64 * 16 * 32 : (const f32x4*)A.d2);
            const unsigned lane_o = (unsigned)(c * 32 + jt) * 16u;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                GnAcc st;
                st.init();
#pragma unroll
                for (int pw = 0; pw < 4; ++pw) {
                    const f32x4 v = acc[4 * r + pw];
                    const unsigned po = (unsigned)(32 * h + 4 * r + pw);
                    if (A.ystem_dbg) buf_st16(v, dbgb, lane_o, po * 8192u);
                    f32x4 y;
                    y.x = fmaxf(__builtin_fmaf(v.x, ia[0], ib[0]), 0.0f);
                    y.y = fmaxf(__builtin_fmaf(v.y, ia[1], ib[1]), 0.0f);
                    y.z = fmaxf(__builtin_fmaf(v.z, ia[2], ib[2]), 0.0f);
                    y.w = fmaxf(__builtin_fmaf(v.w, ia[3], ib[3]), 0.0f);
                    buf_st16_nt(y, outb, lane_o, po * 8192u);
                    st.add(y.x);
                    st.add(y.y);
                    st.add(y.z);
                    st.add(y.w);
                }
                bs[r] = st.bs, bq[r] = st.bq;
            }
        }
        {
            double S, Q;
            ordered_totals(S, Q);
            const double S2 = __shfl_xor(S, 1, 64), Q2 = __shfl_xor(Q, 1, 64);
            if (h == 0 && (c & 1) == 0) {
                float m, r;
                gn_finish(S + S2, Q + Q2, 1.0 / 512.0, m, r);
                A.out_mean[((size_t)tile * 8 + (c >> 1)) * 32 + jt] = m;
                A.out_rstd[((size_t)tile * 8 + (c >> 1)) * 32 + jt] = r;
            }
        }
    }
}
