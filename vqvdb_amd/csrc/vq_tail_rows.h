// vq_tail_rows.h — the folded decoder tail of full chunks (round 5): up_conv 64->256 k3 @4^3 -> PixelShuffle3D(2) -> final 32->1
// k3 @8^3 -> sigmoid (python/VQVAE_v2.py:265-268,274-275,172-187) as ONE linear map 64ch@4^3 -> 1ch@8^3 (see build_folded_tail),
// on the 16x16x4 MFMA with the structural zeros of the composite operator skipped along D AND H.
//
// Why another kernel.  An output voxel (od,oh,ow) reaches, per axis, the coarse cells of o-1..o+1 and their neighbours: 2,3,3,4,4,3,3,2
// of the 4 input planes / rows / columns (0.75 per axis, 0.42 of the dense 2 097 152 MAC/leaf over three axes).  conv_mfma32_k<OUTMODE 2>
// tiles M as 32 voxels = (od, four oh rows, 8 ow): the union of four rows' reach is everything, only depth can be skipped
// (1 572 864 MAC/leaf issued).  Here an M tile is 16 voxels = TWO (od,oh) cells x 8 ow whose reach in (pd,ph) is the same box, so the
// tile's MFMAs run over exactly the input rows (pd,ph) the box holds:
//   * od in {1,2}, {3,4}, {5,6} pair up along depth (identical reach): tile oh = cells (od0,oh),(od0+1,oh)         — "pair" units
//   * od = 0 and od = 7 have no partner in depth; their cells pair up along H: (1,2), (3,4), (5,6) and the two
//     corner cells (0,7), whose reach in H is disjoint (the one tile that pays for rows it does not need)             — "single" units
//   296 tile-rows x 64 MFMAs per 16 leaves = 1 212 416 MAC/leaf issued (exact D x H would be 288 tile-rows = 1 179 648: the four corner
//   cells (od,oh) in {0,7}^2 have boxes no other cell shares; W cannot be skipped with all 8 ow of a cell in one tile).
//
// Arithmetic = the oracle's tail_apply_ex (its skip_rows = 1 form restates this kernel; both forms give the same bits): per voxel the input rows of its planes ascending, the four positions of a W-row one
// fmaf chain from zero (channels in P8 order), row sums added in row order.  A row outside the voxel's reach in H has all-zero
// composite weights there: its chain is fmaf(0, x, .) = +0 for the finite activations a decoder produces, and tot + (+0) = tot
// (tot is never -0: it starts at +0 and x + (-x) = +0), so not running the row is bit-identical to running it.
//
// Structure.  A wave owns a 16-leaf HALF tile; a workgroup = 8 waves (2 per SIMD) = 128 leaves behind one stream of weights.
// Five units of output planes, od = {0}, {1,2}, {3,4}, {5,6}, {7}; a unit walks the input rows (pd in its reach, ph = 0..3) once:
//   row = 4 phases (pw = 0..3); phase = the active tiles' 16 MFMAs each (K = 64 channels of one input position), tiles interleaved
//   (3-4 accumulators in rotation), fragments of item j+2 requested from LDS when item j has issued (two register sets).
//   The activations of ONE position sit in 16 registers (one per MFMA slot), the next position's in 16 more: requested in the first
//   half of a phase as four dwordx4 (a lane loads a channel quad of its leaf; the P8 order wants element pairs (e, e+2) of it: lanes L
//   and L+32 exchange halves with v_permlane32_swap), gated (ChannelAttention, v_pk_mul) and swapped at the start of the next phase.
//   Weights: one 32 KB slice per phase (dense list of the active tiles' 4 KB blocks; laid out by the host in consumption order),
//   global -> register (one phase) -> LDS into a ring of three slices, written two phases ahead; ONE barrier per phase publishes
//   slice t+1 and frees the slot of slice t-1, so the first fragments of the next phase are read BEFORE its barrier.
// Everything inside a unit is static (the row's active tiles depend on ph only): exact s_waitcnt counts, no branches but the pd loop.
#pragma once
#include <utility>
#include "vq_kernels.h"

template <int... I, typename F>
__device__ __forceinline__ void tr_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void tr_static_for(F&& f) { tr_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int TR_SLICE = 8 * 4096;                                   // bytes per weight slice: up to 8 tile blocks of 4 KB (3..7 used)
constexpr int TR_RING = 3;
constexpr size_t LDS_TAIL_ROWS = (size_t)TR_RING * TR_SLICE;          // 96 KB
constexpr int TR_PHASES = 224;                                        // (2 + 3 + 4 + 3 + 2 planes) x 4 rows x 4 positions
constexpr int TR_STREAM_SLICES = TR_PHASES + 3;                       // the kernel requests slices up to three phases ahead: padding
constexpr int TR_TILE_ROWS = 296;                                     // (tile, input row) pairs that are issued: 64 MFMAs each per 16 leaves

// reach of output coordinate o (0..7 at 8^3) in input coordinates (0..3 at 4^3): final taps o-1..o+1 -> coarse cells -> +-1 (clamped)
constexpr int tr_lo(int o) { const int c = (o > 0 ? o - 1 : 0) >> 1; return c > 0 ? c - 1 : 0; }
constexpr int tr_hi(int o) { const int c = (o < 7 ? o + 1 : 7) >> 1; return c < 3 ? c + 1 : 3; }
// tiles of a unit that input row ph feeds.  Pair units: tile oh (0..7).  Single units: tile 0 = rows (1,2), 1 = (3,4), 2 = (5,6), 3 = (0,7).
constexpr unsigned tr_pair_mask(int ph)
{
    unsigned m = 0;
    for (int oh = 0; oh < 8; ++oh)
        if (ph >= tr_lo(oh) && ph <= tr_hi(oh)) m |= 1u << oh;
    return m;
}
constexpr unsigned tr_single_mask(int ph) { return (ph <= 2 ? 1u : 0u) | 2u | (ph >= 1 ? 4u : 0u) | 8u; }
constexpr unsigned tr_mask(bool pair, int ph) { return pair ? tr_pair_mask(ph) : tr_single_mask(ph); }
constexpr int tr_popc(unsigned m) { int n = 0; for (; m; m &= m - 1) ++n; return n; }
constexpr int tr_nth(unsigned m, int i) { for (int b = 0; b < 32; ++b) if (m >> b & 1) { if (i == 0) return b; --i; } return -1; }
// A phase's fragment requests run up to two items ahead; in the last phase of a unit they fall on the NEXT unit's slice with the ending unit's mask
// (the values are discarded and re-requested at the start of run_unit): with at most 7 of the slot's 8 blocks ever addressed they stay inside the slot.
constexpr int tr_max_blocks()
{
    int m = 0;
    for (int pair = 0; pair < 2; ++pair)
        for (int ph = 0; ph < 4; ++ph) m = tr_popc(tr_mask(pair != 0, ph)) > m ? tr_popc(tr_mask(pair != 0, ph)) : m;
    return m;
}
static_assert((tr_max_blocks() + 1) * 4096 <= TR_SLICE, "tail_rows16_k: a weight slice's blocks (+1 for the look-ahead across a unit boundary) must fit the ring slot");
static_assert(LDS_TAIL_ROWS <= 160 * 1024, "gfx950: 160 KB of LDS per workgroup");
// the two (od,oh) cells of tile `tid` of a unit whose first plane is od0: rows 0-7 of the tile = cell A, rows 8-15 = cell B
constexpr int tr_cell_a(bool pair, int od0, int tid) { return pair ? od0 * 8 + tid : od0 * 8 + (tid == 3 ? 0 : 2 * tid + 1); }
constexpr int tr_cell_b(bool pair, int od0, int tid) { return pair ? (od0 + 1) * 8 + tid : od0 * 8 + (tid == 3 ? 7 : 2 * tid + 2); }
// items of a phase with N active tiles: N <= 4: one item per fragment group (4 groups of 4 MFMA slots), all tiles; else two items
// per group (first ceil(N/2) tiles, rest).  Item j uses fragment register set j & 1.
constexpr int tr_parts(int n) { return n > 4 ? 2 : 1; }
constexpr int tr_part_lo(int n, int part) { return tr_parts(n) == 1 ? 0 : (part == 0 ? 0 : (n + 1) / 2); }
constexpr int tr_part_hi(int n, int part) { return tr_parts(n) == 1 ? n : (part == 0 ? (n + 1) / 2 : n); }

// ABL (tools/ablate/tail_rows_ablate.hip only): 1 no barriers, 2 no weight streaming, 4 no LDS fragment reads, 8 no activation re-loads,
// 16 no gate multiply, 32 no epilogue, 64 no lane swap either, 128 no MFMAs
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void tail_rows16_k(ConvArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, k = lane >> 4;
    int half = blockIdx.x * 8 + wave;
    const bool active = half < 2 * A.n_tiles;
    if (!active) half = 2 * A.n_tiles - 1;   // a wave without a half tile re-computes the last one (it carries its share of the weights)
    const int tile = half >> 1, jj = 16 * (half & 1) + n;
    const bool store = active && (int64_t)tile * 32 + jj < A.n_leaves;

    // Activations: element (pos, channel c, leaf) at pos*8192 + (c>>2)*512 + leaf*16 + (c&3)*4 bytes of the tile.  Lane (n, k) loads
    // the whole channel quad 4j + k of its leaf (j = 0..3: four dwordx4 per position, 1 KB each — every vector-memory instruction
    // issued into the MFMA stream costs the wave ~100 cycles, their number is what counts); MFMA slot s = 2u + mf of lane (n, k) wants
    // channel 8u + 4(k&1) + (k>>1) + 2mf: for octet u = 2j the lanes k = 0,1 hold the right quad and want its elements (0,2), the
    // lanes k = 2,3 want elements (1,3) of the quad their partner lane L-32 holds; for octet 2j+1 it is the other way round, so
    // v_permlane32_swap of (x,y) and of (z,w) turns one float4 into (octet 2j mf 0, octet 2j+1 mf 0, octet 2j mf 1, octet 2j+1 mf 1).
    // ChannelAttention gates are applied BEFORE the swap, to the quad the lane loaded.
    f32x4 tg[4];
    {
        float hid[16], gall[64];
        se_hidden<64>(A.se_csum + (size_t)tile * 64 * 32 + jj, A.se_fc0, hid);
        se_gates<64>(hid, A.se_fc2, gall);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (bit selects: written as ?: the compiler turns the four-way choice into an indexed read of gall[] from scratch)
                const int c0 = 16 * j + e;
                const unsigned m1 = 0u - (unsigned)(k & 1), m2 = 0u - (unsigned)(k >> 1);
                const unsigned lo = (__float_as_uint(gall[c0 + 4]) & m1) | (__float_as_uint(gall[c0]) & ~m1);
                const unsigned hi = (__float_as_uint(gall[c0 + 12]) & m1) | (__float_as_uint(gall[c0 + 8]) & ~m1);
                tg[j][e] = __uint_as_float((hi & m2) | (lo & ~m2));
            }
    }
    const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * 64 * 16 * 32);
    const unsigned lane_x = (unsigned)(k * 512 + jj * 16);
    // Bc = the position this phase's MFMAs read (gated, swapped: Bc[j] = (slot 4j, slot 4j+2, slot 4j+1, slot 4j+3)), Bn = the next
    // position, raw, requested in the first half of the phase
    f32x4 Bc[4], Bn[4];
    auto reload1 = [&](int j, int pos) {
        if (ABL & 8) return;
        Bn[j] = buf_ld16(inb, lane_x + j * 2048, (unsigned)pos * 8192u);
    };
    auto arrive = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            f32x4 v = Bn[j];
            if (ABL & 64) { Bc[j] = v; continue; }
            if (!(ABL & 16)) v = v * tg[j];
            const u32x2 r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
            const u32x2 r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
            Bc[j] = (f32x4){__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y)};
        }
    };
    // weights: slice t at t*TR_SLICE; this wave moves the 1 KB pieces wave, wave + 8, .. of a slice, as many as cover the slice's N tile
    // blocks (ceil(N / 2) of 4: a slice of 3 .. 7 blocks is not copied as if it had 8; the count is a compile-time constant per phase)
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;
    f32x4* const lds_w = (f32x4*)(smem_raw + wave * 1024) + lane;   // + slot*TR_SLICE/16 + kp*512
    const f32x4* const lds_r = (const f32x4*)smem_raw + lane;       // + slot*TR_SLICE/16 + (tile i*4 + g)*64
    const vq_buf outb = buf_of(A.out + (size_t)tile * 32 * 512);
    const float* bias = A.bias_frag;   // plain per voxel [512]

    int t = 0, sl = 0;   // phase (= slice) counter and t % 3
    auto slot_of = [&](int ahead) { const int s = sl + ahead; return s >= TR_RING ? s - TR_RING : s; };

    // ---- prologue: slices 0 and 1 into the ring, slice 2 into registers, positions 0..2 of the first row ----
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_w[s * (TR_SLICE / 16) + j * 512] = buf_ld16(wb, lane_w + j * 8192, (unsigned)(s * TR_SLICE + wave * 1024));
    f32x4 wreg[4];   // this wave's share of the slice two phases ahead: loaded in phase t-1, written to the ring in phase t
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = buf_ld16(wb, lane_w + j * 8192, (unsigned)(2 * TR_SLICE + wave * 1024));
#pragma unroll
    for (int j = 0; j < 4; ++j) reload1(j, 0);   // unit 0 starts at plane 0, row 0, position 0
    __builtin_amdgcn_s_waitcnt(0x0f70);               // enter the loops with nothing in flight
    __syncthreads();

    f32x4 fa[2][4];   // A-fragment register sets (item j -> set j & 1)
    if (ABL & 4) {    // (ablation: fragments read once, the MFMAs keep real operands)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[i >> 2][i & 3] = lds_r[i * 64];
            asm volatile("" : "+v"(fa[i >> 2][i & 3]));
        }
    }
    // fragments of item `it` of a phase with active mask M, from ring slot `slot`
    auto frag_req = [&](auto mc, auto itc, int slot) {
        constexpr unsigned M = decltype(mc)::value;
        constexpr int it = decltype(itc)::value;
        constexpr int N = tr_popc(M), P = tr_parts(N), g = it / P, part = it % P;
        if (ABL & 4) return;
#pragma unroll
        for (int i = tr_part_lo(N, part); i < tr_part_hi(N, part); ++i) fa[it & 1][i - tr_part_lo(N, part)] = lds_r[slot * (TR_SLICE / 16) + (i * 4 + g) * 64];
    };

    auto run_unit = [&](auto pair_c, const int od0, const int pd_lo, const int pd_hi, const int p_next_unit) {
        constexpr bool PAIR = decltype(pair_c)::value;
        constexpr int NT = PAIR ? 8 : 4;
        f32x4 tot[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) tot[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        // the unit's first phase: its slice was published by the previous barrier
        frag_req(std::integral_constant<unsigned, tr_mask(PAIR, 0)>{}, std::integral_constant<int, 0>{}, sl);
        frag_req(std::integral_constant<unsigned, tr_mask(PAIR, 0)>{}, std::integral_constant<int, 1>{}, sl);
#pragma nounroll
        for (int pd = pd_lo; pd <= pd_hi; ++pd) {
            auto row = [&](auto phc) {
                constexpr int PH = decltype(phc)::value;
                constexpr unsigned MASK = tr_mask(PAIR, PH), NMASK = tr_mask(PAIR, (PH + 1) & 3);
                constexpr int N = tr_popc(MASK), P = tr_parts(N), NI = 4 * P;
                const int pcur = (pd * 4 + PH) * 4;
                const int pnext = (PH == 3 && pd == pd_hi) ? p_next_unit : pcur + 4;
                f32x4 acc[NT];
                auto phase = [&](auto pwc) {
                    constexpr int PW = decltype(pwc)::value;
                    constexpr unsigned NEXT = PW < 3 ? MASK : NMASK;
                    const int qpos = PW < 3 ? pcur + PW + 1 : pnext;   // the position the next phase reads
                    t = __builtin_amdgcn_readfirstlane(t), sl = __builtin_amdgcn_readfirstlane(sl);   // (loop-carried counters: keep them scalar, the slice offset is the loads' scalar offset)
                    if (!(ABL & 1)) __syncthreads();   // slice t+1 visible to every wave; every wave is done with slice t-1's slot
                    arrive();
                    __builtin_amdgcn_sched_barrier(0);
                    const int ws = slot_of(2);
                    tr_static_for<NI>([&](auto itc) {
                        constexpr int it = decltype(itc)::value;
                        constexpr int g = it / P, part = it % P, i0 = tr_part_lo(N, part), i1 = tr_part_hi(N, part);
                        tr_static_for<4>([&](auto ec) {
                            constexpr int e = decltype(ec)::value, s = 4 * g + e;
                            tr_static_for<i1 - i0>([&](auto ic) {
                                constexpr int i = i0 + decltype(ic)::value, tid = tr_nth(MASK, i);
                                const float a = fa[it & 1][i - i0][e], b = Bc[s >> 2][2 * (s & 1) + ((s >> 1) & 1)];   // slot s = 2u + mf, u = 2j + (u & 1)
                                if constexpr ((ABL & 128) != 0) {
                                    if constexpr (PW == 0 && s == 0) acc[tid][0] = a * b;
                                    else acc[tid][0] += a * b;
                                } else if constexpr (PW == 0 && s == 0) {
                                    acc[tid] = mfma16(a, b, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});   // a W-row's chain starts from zero
                                } else {
                                    acc[tid] = mfma16(a, b, acc[tid]);
                                }
                            });
                        });
                        __builtin_amdgcn_sched_barrier(0);
                        // item it+2 of this phase, or item (it+2-NI) of the next one (its slice is in the ring since the last barrier)
                        if constexpr (it + 2 < NI) frag_req(std::integral_constant<unsigned, MASK>{}, std::integral_constant<int, it + 2>{}, sl);
                        else frag_req(std::integral_constant<unsigned, NEXT>{}, std::integral_constant<int, it + 2 - NI>{}, slot_of(1));
                        // this item's share of the phase's memory traffic: piece k of slice t+2 (in registers since the last phase) goes to
                        // the ring and its registers take piece k of slice t+3
                        if constexpr (it % (NI / 4) == 0 && (ABL & 2) == 0) {
                            constexpr int kp = it / (NI / 4);
                            // pieces per wave of slices t+2 / t+3: ceil(N / 2) while the slice lies in this plane (its N is static), all four
                            // behind it (the next plane's first row, or another unit's)
                            constexpr int q2 = PH * 4 + PW + 2, q3 = PH * 4 + PW + 3;
                            constexpr int c2 = q2 < 16 ? (tr_popc(tr_mask(PAIR, q2 / 4)) + 1) / 2 : 4, c3 = q3 < 16 ? (tr_popc(tr_mask(PAIR, q3 / 4)) + 1) / 2 : 4;
                            if constexpr (kp < c2) lds_w[ws * (TR_SLICE / 16) + kp * 512] = wreg[kp];
                            if constexpr (kp < c3) wreg[kp] = buf_ld16(wb, lane_w + kp * 8192, (unsigned)((t + 3) * TR_SLICE + wave * 1024));
                        }
                        // the next position's four quads, requested between the MFMA runs of the first half of the phase
                        tr_static_for<4>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            if constexpr (j * (NI / 2) / 4 == it) reload1(j, qpos);
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    ++t;
                    sl = slot_of(1);
                };
                phase(std::integral_constant<int, 0>{});
                phase(std::integral_constant<int, 1>{});
                phase(std::integral_constant<int, 2>{});
                phase(std::integral_constant<int, 3>{});
                tr_static_for<N>([&](auto ic) {
                    constexpr int tid = tr_nth(MASK, decltype(ic)::value);
                    tot[tid] = tot[tid] + acc[tid];   // row sums in row order
                });
            };
            row(std::integral_constant<int, 0>{});
            row(std::integral_constant<int, 1>{});
            row(std::integral_constant<int, 2>{});
            row(std::integral_constant<int, 3>{});
        }
        // ---- epilogue: per-voxel bias, sigmoid, store into the caller's leaf-major [n][512] buffer (VQVAECodec.cpp:182-192) ----
        if (ABL & 32) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < NT; ++i) s += tot[i][0] + tot[i][3];
            if (s == 12345.678f) A.out[threadIdx.x] = s;
            return;
        }
#pragma unroll
        for (int tid = 0; tid < NT; ++tid) {
            // rows 4k .. 4k+3 of the tile: k < 2 cell A, else cell B; ow = 4(k&1) .. +3
            const int ca = PAIR ? od0 * 8 + tid : od0 * 8 + (tid == 3 ? 0 : 2 * tid + 1);
            const int cb = PAIR ? (od0 + 1) * 8 + tid : od0 * 8 + (tid == 3 ? 7 : 2 * tid + 2);
            const int vox = (k < 2 ? ca : cb) * 8 + (k & 1) * 4;
            const f32x4 bv = *(const f32x4*)(bias + vox);
            f32x4 sg;
            sg.x = vq_sigmoid(tot[tid].x + bv.x), sg.y = vq_sigmoid(tot[tid].y + bv.y);
            sg.z = vq_sigmoid(tot[tid].z + bv.z), sg.w = vq_sigmoid(tot[tid].w + bv.w);
            if (store) buf_st16(sg, outb, (unsigned)(jj * 512 + vox) * 4u, 0u);
        }
    };

    // od0 / planes of the five units: {0}: 0..1, {1,2}: 0..2, {3,4}: 0..3, {5,6}: 1..3, {7}: 2..3; a unit's last row re-loads the
    // first row of the next unit (the last unit: its own last row again)
#pragma nounroll
    for (int unit = 0; unit < 5; ++unit) {
        const int od0 = unit == 0 ? 0 : unit == 4 ? 7 : 2 * unit - 1;
        const int pd_lo = unit <= 2 ? 0 : unit - 2, pd_hi = unit >= 2 ? 3 : unit + 1;
        const int nlo = unit + 1 <= 2 ? 0 : unit - 1;   // first plane of unit + 1
        const int p_next_unit = unit < 4 ? nlo * 16 : 60;
        if (unit == 0 || unit == 4) run_unit(std::false_type{}, od0, pd_lo, pd_hi, p_next_unit);
        else run_unit(std::true_type{}, od0, pd_lo, pd_hi, p_next_unit);
    }
}

// ------------------------------------------------------------------------------------------
// The same operator with a whole 32-leaf tile per wave and ONE wave per SIMD (512 registers): every weight fragment read from LDS
// feeds two MFMAs (the tile's two 16-leaf halves), so the LDS reads, the fragment waits and the lock-step of two waves sharing a
// matrix pipe are halved; nothing but this wave's own instruction stream sits between its MFMAs, and that stream is static.
// A workgroup = 4 waves = 128 leaves behind one weight stream (as above).  Activations: the tile's natural float4 (lane = (leaf,
// quad), fully coalesced 1 KB per load, eight per position); v_permlane32_swap of (x,y) and (z,w) turns a float4 into the four
// operands (half 0 mf 0, half 1 mf 0, half 0 mf 1, half 1 mf 1) after the gate multiply.
// ------------------------------------------------------------------------------------------
constexpr int TR32_WAVES = 4;
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void tail_rows32_k(ConvArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, k = lane >> 4;
    int tile = blockIdx.x * TR32_WAVES + wave;
    const bool active = tile < A.n_tiles;
    if (!active) tile = A.n_tiles - 1;   // a wave without a tile re-computes the last one (it carries its share of the weights)
    // load lane = (leaf jl of the tile, quad ql): lanes 0-31 hold half 0, lanes 32-63 half 1
    const int jl = 16 * (lane >> 5) + n, ql = k & 1;

    // ---- ChannelAttention gates of the four channels 8u + 4 ql + {0..3} of leaf jl, applied before the lane swap ----
    f32x4 tg[8];
    {
        float hid[16], gall[64];
        se_hidden<64>(A.se_csum + (size_t)tile * 64 * 32 + jl, A.se_fc0, hid);
        se_gates<64>(hid, A.se_fc2, gall);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) tg[u][e] = ql ? gall[8 * u + 4 + e] : gall[8 * u + e];
    }
    const vq_buf inb = buf_of((const f32x4*)A.in + (size_t)tile * 64 * 16 * 32);
    const unsigned lane_x = (unsigned)(ql * 512 + jl * 16);
    // Two register sets: Bc = the position the MFMAs of this phase read (gated, swapped), Bn = the next position in flight (raw):
    // requested in the first half of a phase, it has the other half to arrive (an HBM round trip is a tenth of a phase)
    f32x4 Bc[8], Bn[8];   // Bc[u] = (half 0 slot 2u, half 1 slot 2u, half 0 slot 2u+1, half 1 slot 2u+1)
    auto reload1 = [&](int u, int pos) {
        if (ABL & 8) return;
        Bn[u] = buf_ld16(inb, lane_x + u * 1024, (unsigned)pos * 8192u);
    };
    auto arrive = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            f32x4 v = Bn[u];
            if (ABL & 64) { Bc[u] = v; continue; }
            if (!(ABL & 16)) v = v * tg[u];
            const u32x2 r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
            const u32x2 r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
            Bc[u] = (f32x4){__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y)};
        }
    };
    // weights: slice t at t*TR_SLICE; this wave moves bytes [wave*8192, +8192) of every slice
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;
    f32x4* const lds_w = (f32x4*)(smem_raw + wave * 8192) + lane;   // + slot*TR_SLICE/16 + j*64
    const f32x4* const lds_r = (const f32x4*)smem_raw + lane;       // + slot*TR_SLICE/16 + (tile i*4 + g)*64
    const vq_buf outb = buf_of(A.out + (size_t)tile * 32 * 512);
    const float* bias = A.bias_frag;

    int t = 0, sl = 0;
    auto slot_of = [&](int ahead) { const int s = sl + ahead; return s >= TR_RING ? s - TR_RING : s; };
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_w[s * (TR_SLICE / 16) + j * 64] = buf_ld16(wb, lane_w + j * 1024, (unsigned)(s * TR_SLICE + wave * 8192));
    f32x4 wreg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wreg[j] = buf_ld16(wb, lane_w + j * 1024, (unsigned)(2 * TR_SLICE + wave * 8192));
#pragma unroll
    for (int u = 0; u < 8; ++u) reload1(u, 0);   // unit 0 starts at plane 0, row 0, position 0
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    f32x4 fa[2][4];
    if (ABL & 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[i >> 2][i & 3] = lds_r[i * 64];
            asm volatile("" : "+v"(fa[i >> 2][i & 3]));
        }
    }
    auto frag_req = [&](auto mc, auto itc, int slot) {
        constexpr unsigned M = decltype(mc)::value;
        constexpr int it = decltype(itc)::value;
        constexpr int N = tr_popc(M), P = tr_parts(N), g = it / P, part = it % P;
        if (ABL & 4) return;
#pragma unroll
        for (int i = tr_part_lo(N, part); i < tr_part_hi(N, part); ++i) fa[it & 1][i - tr_part_lo(N, part)] = lds_r[slot * (TR_SLICE / 16) + (i * 4 + g) * 64];
    };

    auto run_unit = [&](auto pair_c, const int od0, const int pd_lo, const int pd_hi, const int p_next_unit) {
        constexpr bool PAIR = decltype(pair_c)::value;
        constexpr int NT = PAIR ? 8 : 4;
        f32x4 tot[2][NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) tot[0][i] = tot[1][i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        frag_req(std::integral_constant<unsigned, tr_mask(PAIR, 0)>{}, std::integral_constant<int, 0>{}, sl);
        frag_req(std::integral_constant<unsigned, tr_mask(PAIR, 0)>{}, std::integral_constant<int, 1>{}, sl);
#pragma nounroll
        for (int pd = pd_lo; pd <= pd_hi; ++pd) {
            auto row = [&](auto phc) {
                constexpr int PH = decltype(phc)::value;
                constexpr unsigned MASK = tr_mask(PAIR, PH), NMASK = tr_mask(PAIR, (PH + 1) & 3);
                constexpr int N = tr_popc(MASK), P = tr_parts(N), NI = 4 * P;
                const int pcur = (pd * 4 + PH) * 4;
                const int pnext = (PH == 3 && pd == pd_hi) ? p_next_unit : pcur + 4;
                f32x4 acc[2][NT];
                auto phase = [&](auto pwc) {
                    constexpr int PW = decltype(pwc)::value;
                    constexpr unsigned NEXT = PW < 3 ? MASK : NMASK;
                    const int qpos = PW < 3 ? pcur + PW + 1 : pnext;   // the position the next phase reads
                    t = __builtin_amdgcn_readfirstlane(t), sl = __builtin_amdgcn_readfirstlane(sl);
                    if (!(ABL & 1)) __syncthreads();
                    arrive();
                    __builtin_amdgcn_sched_barrier(0);
                    const int ws = slot_of(2);
                    tr_static_for<NI>([&](auto itc) {
                        constexpr int it = decltype(itc)::value;
                        constexpr int g = it / P, part = it % P, i0 = tr_part_lo(N, part), i1 = tr_part_hi(N, part);
                        tr_static_for<4>([&](auto ec) {
                            constexpr int e = decltype(ec)::value, s = 4 * g + e;
                            tr_static_for<i1 - i0>([&](auto ic) {
                                constexpr int i = i0 + decltype(ic)::value, tid = tr_nth(MASK, i);
                                const float a = fa[it & 1][i - i0][e];
                                tr_static_for<2>([&](auto hc) {
                                    constexpr int h = decltype(hc)::value;
                                    const float b = Bc[s >> 1][2 * (s & 1) + h];
                                    if constexpr ((ABL & 128) != 0) {
                                        if constexpr (PW == 0 && s == 0) acc[h][tid][0] = a * b;
                                        else acc[h][tid][0] += a * b;
                                    } else if constexpr (PW == 0 && s == 0) {
                                        acc[h][tid] = mfma16(a, b, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
                                    } else {
                                        acc[h][tid] = mfma16(a, b, acc[h][tid]);
                                    }
                                });
                            });
                        });
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (it + 2 < NI) frag_req(std::integral_constant<unsigned, MASK>{}, std::integral_constant<int, it + 2>{}, sl);
                        else frag_req(std::integral_constant<unsigned, NEXT>{}, std::integral_constant<int, it + 2 - NI>{}, slot_of(1));
                        if constexpr ((ABL & 2) == 0) {
#pragma unroll
                            for (int kp = it * (8 / NI); kp < (it + 1) * (8 / NI); ++kp) {
                                lds_w[ws * (TR_SLICE / 16) + kp * 64] = wreg[kp];
                                wreg[kp] = buf_ld16(wb, lane_w + kp * 1024, (unsigned)((t + 3) * TR_SLICE + wave * 8192));
                            }
                        }
                        // the next position: all of it in the first half of the phase
                        if constexpr (it < NI / 2) {
#pragma unroll
                            for (int u = it * (16 / NI); u < (it + 1) * (16 / NI); ++u) reload1(u, qpos);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    ++t;
                    sl = slot_of(1);
                };
                phase(std::integral_constant<int, 0>{});
                phase(std::integral_constant<int, 1>{});
                phase(std::integral_constant<int, 2>{});
                phase(std::integral_constant<int, 3>{});
                tr_static_for<N>([&](auto ic) {
                    constexpr int tid = tr_nth(MASK, decltype(ic)::value);
                    tot[0][tid] = tot[0][tid] + acc[0][tid];
                    tot[1][tid] = tot[1][tid] + acc[1][tid];
                });
            };
            row(std::integral_constant<int, 0>{});
            row(std::integral_constant<int, 1>{});
            row(std::integral_constant<int, 2>{});
            row(std::integral_constant<int, 3>{});
        }
        if (ABL & 32) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < NT; ++i) s += tot[0][i][0] + tot[0][i][3] + tot[1][i][0] + tot[1][i][3];
            if (s == 12345.678f) A.out[threadIdx.x] = s;
            return;
        }
#pragma unroll
        for (int tid = 0; tid < NT; ++tid) {
            const int ca = PAIR ? od0 * 8 + tid : od0 * 8 + (tid == 3 ? 0 : 2 * tid + 1);
            const int cb = PAIR ? (od0 + 1) * 8 + tid : od0 * 8 + (tid == 3 ? 7 : 2 * tid + 2);
            const int vox = (k < 2 ? ca : cb) * 8 + (k & 1) * 4;
            const f32x4 bv = *(const f32x4*)(bias + vox);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 sg;
                sg.x = vq_sigmoid(tot[h][tid].x + bv.x), sg.y = vq_sigmoid(tot[h][tid].y + bv.y);
                sg.z = vq_sigmoid(tot[h][tid].z + bv.z), sg.w = vq_sigmoid(tot[h][tid].w + bv.w);
                const int jj = 16 * h + n;
                if (active && (int64_t)tile * 32 + jj < A.n_leaves) buf_st16(sg, outb, (unsigned)(jj * 512 + vox) * 4u, 0u);
            }
        }
    };
#pragma nounroll
    for (int unit = 0; unit < 5; ++unit) {
        const int od0 = unit == 0 ? 0 : unit == 4 ? 7 : 2 * unit - 1;
        const int pd_lo = unit <= 2 ? 0 : unit - 2, pd_hi = unit >= 2 ? 3 : unit + 1;
        const int nlo = unit + 1 <= 2 ? 0 : unit - 1;
        const int p_next_unit = unit < 4 ? nlo * 16 : 60;
        if (unit == 0 || unit == 4) run_unit(std::false_type{}, od0, pd_lo, pd_hi, p_next_unit);
        else run_unit(std::true_type{}, od0, pd_lo, pd_hi, p_next_unit);
    }
}
