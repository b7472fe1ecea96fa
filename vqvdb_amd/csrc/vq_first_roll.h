// vq_first_roll.h — Conv3d(1->16,k3,p1) @8^3 + GroupNorm(4,16) + ReLU (VQVAE_v2.py:235-237) with a ROLLING 3 x 3 ROW WINDOW in registers
// (round 4).
//
// conv_first_k (vq_kernels.h) walks (output row, kd) steps and loads the step's three input rows every time: 9 row loads per output
// row, every input row fetched nine times (0.69 / 0.86 GB per pass for a 0.2 GB input, profiles/r04_ablate_conv_first.txt: the loads
// are 0.03 ms of the statistics pass and 0.05 ms of the store-bound normalising pass).  Here a wave owns ONE 16-leaf sub-tile and keeps,
// for each of the three input planes of the current output plane, the rows ih-1, ih, ih+1 in registers (slot = ih mod 3): moving
// down one output row costs three row loads (one per plane) instead of nine.  The loads are unconditional and sit at fixed places of a
// straight-line row body (clamped rows / planes where the neighbour does not exist: the MFMAs of such taps are skipped by wave-uniform
// branches, as everywhere), so every operand wait is an exact count.  The row body exists three times (slot rotation oh mod 3: register
// indices must be static).  Same MFMAs in the same order per accumulator — (kd, kh) ascending, K slots = kw — so the same bits as
// conv_first_k (tests/test_gpu_parity.py::test_large_path_kernel_variants_agree, VQHIP_FIRST=steps selects the old kernel).
//   MODE 0: statistics of y1 = conv(x) + bias for GroupNorm(4,16).   MODE 1: recompute, a1 = relu(gn(y1)) -> store, statistics of a1.
#pragma once
#include "vq_kernels.h"

template <int MODE, bool RAW = false>   // RAW: the caller's float[leaf][512] layout read directly (see conv_first_k)
__global__ __launch_bounds__(256, 2) void conv_first_roll_k(ConvArgs A)
{
    static_assert(MODE == 0 || MODE == 1, "statistics pass / normalising pass");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = blockIdx.x * 4 + wave;          // 16-leaf sub-tile
    const int tile = half >> 1, sb = half & 1;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    // K slot q4 = kw (3 = pad): a lane's eight B operands of a row are the 8 floats from element kw of the row's record (0, x0..x7, 0);
    // the pad slot's lane offset lies beyond the descriptor's range and reads zeros (see conv_first_k)
    const vq_buf xb = RAW ? buf_of_n(A.in + (size_t)tile * 32 * 512, (unsigned)min((int64_t)32, A.n_leaves - (int64_t)tile * 32) * 2048u)
                          : buf_of(A.in + (size_t)tile * VQ_XR_TILE);
    const unsigned lane_x = q4 < 3 ? (RAW ? (unsigned)((sb * 16 + jj) * 512) * 4u : (unsigned)(jj * VQ_XR_REC + q4) * 4u) : 0x80000000u;   // (RAW: the leaf in the LANE offset, see conv_first_k)
    const bool halo_lo = q4 == 0, halo_hi = q4 == 2;
    const unsigned raw_lo = lane_x + (q4 < 3 ? (unsigned)(q4 > 0 ? q4 - 1 : 0) * 4u : 0u);            // (no load starts outside its leaf: see conv_first_k)
    const unsigned raw_hi = lane_x + (q4 < 3 ? (unsigned)(4 + (q4 == 2 ? 0 : q4 - 1)) * 4u : 0u);
    auto ldrow = [&](int r, int hf) __attribute__((always_inline)) -> f32x4 {
        if (!RAW) return buf_ld16(xb, lane_x, (unsigned)((r * 32 + sb * 16) * VQ_XR_REC + hf * 4) * 4u);
        f32x4 v = buf_ld16(xb, hf ? raw_hi : raw_lo, (unsigned)(r * 8) * 4u);
        if (hf == 0) v = halo_lo ? (f32x4){0.0f, v.x, v.y, v.z} : v;
        else v = halo_hi ? (f32x4){v.y, v.z, v.w, 0.0f} : v;
        return v;
    };
    const bool has_out = A.out != nullptr;
    const vq_buf outb = buf_of(has_out ? (const f32x4*)A.out + (size_t)tile * 512 * 4 * 32 : (const f32x4*)A.in);
    const unsigned lane_o = (unsigned)(q4 * 32 + jj) * 16u;
    f32x4 ia = {0, 0, 0, 0}, ib = {0, 0, 0, 0};   // MODE 1: GroupNorm(4,16) of y1, group = q4 (this lane's 4 couts)
    if (MODE == 1) {
        const float mean = A.in_mean[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
        const float rstd = A.in_rstd[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ia[i] = rstd * A.in_gamma[4 * q4 + i];
            ib[i] = __builtin_fmaf(-mean, ia[i], A.in_beta[4 * q4 + i]);
        }
    }
    GnAcc st[MODE == 1 ? 2 : 1];
#pragma unroll
    for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) st[k].init();

    int g0, g1;
    split_range<64>(g0, g1);
    f32x4 win[3][3][2];   // [kd][row slot = ih mod 3][half row]
    // row ih of the three input planes of output plane od into slot SLOT (static: the window must stay in registers; clamped where the
    // plane / row does not exist)
    auto load_rows = [&](int od, int ih, auto SLOT_C) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOT_C)::value;
        const int ihc = max(0, min(ih, 7));
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int idc = max(0, min(od + kd - 1, 7));
            win[kd][SLOT][0] = ldrow(idc * 8 + ihc, 0);
            win[kd][SLOT][1] = ldrow(idc * 8 + ihc, 1);
        }
    };
    // one output row with slot rotation PH = oh mod 3: input row oh + kh - 1 sits in slot (PH + kh + 2) mod 3
    auto row_body = [&](auto PH_C, int row) __attribute__((always_inline)) {
        constexpr int PH = decltype(PH_C)::value;
        const int od = row >> 3, oh = row & 7;
        if (row == g0 || oh == 0) {   // (wave-uniform) a new output plane (or the start of this workgroup's range): rows oh - 1, oh, oh + 1
            load_rows(od, oh - 1, std::integral_constant<int, (PH + 2) % 3>{});
            load_rows(od, oh, std::integral_constant<int, PH>{});
            load_rows(od, oh + 1, std::integral_constant<int, (PH + 1) % 3>{});
        }
        f32x4 acc[8];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow] = (f32x4){0, 0, 0, 0};
        const int ihn = min(oh + 2, 7);   // the row the NEXT output row adds to the window (clamped: at oh = 6, 7 nobody reads it)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const bool pv = (unsigned)(od + kd - 1) < 8u;   // (wave-uniform) does this input plane exist?
            const int idc = max(0, min(od + kd - 1, 7));
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if (pv && (unsigned)(oh + kh - 1) < 8u) {
                    const float wv = w[kd * 3 + kh];
                    const f32x4 x0 = win[kd][(PH + kh + 2) % 3][0], x1 = win[kd][(PH + kh + 2) % 3][1];
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) acc[ow] = mfma16(wv, ow < 4 ? x0[ow & 3] : x1[ow & 3], acc[ow]);
                }
                if (kh == 0) {
                    // this plane's row for the next output row, into the slot its kh = 0 tap has just left; unconditional (clamped)
                    __builtin_amdgcn_sched_barrier(0);
                    win[kd][(PH + 2) % 3][0] = ldrow(idc * 8 + ihn, 0);
                    win[kd][(PH + 2) % 3][1] = ldrow(idc * 8 + ihn, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) {
            f32x4 v = acc[ow] + bias4;
            if (MODE == 0) {
                if (has_out) buf_st16(v, outb, lane_o, (unsigned)((row * 8 + ow) * 128 + 16 * sb) * 16u);   // debug only
                st[0].add(v.x);
                st[0].add(v.y);
                st[0].add(v.z);
                st[0].add(v.w);
            } else {
                v = gn_relu4(v, ia, ib);
                buf_st16_nt(v, outb, lane_o, (unsigned)((row * 8 + ow) * 128 + 16 * sb) * 16u);   // streaming store (see conv_first_k)
                st[0].add(v.x);
                st[0].add(v.y);
                st[1].add(v.z);
                st[1].add(v.w);
            }
        }
        if ((row & 3) == 3) {   // 4 rows = 32 positions = one statistics block
#pragma unroll
            for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k)
                st[k].fold_store(A.part_s, A.part_q, part_index(tile, row >> 2, MODE == 1 ? 2 * q4 + k : q4, 16 * sb + jj));
        }
    };
    for (int row = g0; row < g1; ++row) {
        const int ph = (row & 7) % 3;
        if (ph == 0) row_body(std::integral_constant<int, 0>{}, row);
        else if (ph == 1) row_body(std::integral_constant<int, 1>{}, row);
        else row_body(std::integral_constant<int, 2>{}, row);
    }
    if (A.part_s) return;   // split launch: gn_combine_k finishes
    if (MODE == 0) {
        float m, r;
        gn_finish(st[0].s, st[0].q, 1.0 / 2048.0, m, r);
        A.out_mean[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj] = m;
        A.out_rstd[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj] = r;
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float m, r;
            gn_finish(st[k].s, st[k].q, 1.0 / 1024.0, m, r);
            A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
        }
    }
}
