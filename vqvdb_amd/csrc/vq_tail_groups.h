// vq_tail_groups.h — the folded decoder tail of full chunks, third cut (round 5): the D x H zero-skipping tiling of vq_tail_rows.h
// (16-voxel tiles of two (od,oh) cells with one reach box, 16x16x4 MFMA, 296 (tile, input row) pairs = 1 212 416 MAC/leaf) with the
// output planes in THREE groups instead of five units, so that every input plane is read 2.5x instead of 3.5x and gated / swapped
// 40 times instead of 56 per half tile:
//     group 0 = od {0} + {1,2}: input planes 0..2, the four od = 0 tiles only while pd <= 1
//     group 1 = od {3,4}      : input planes 0..3
//     group 2 = od {5,6} + {7}: input planes 1..3, the four od = 7 tiles only while pd >= 2
// A group's row (pd, ph) feeds up to 7 "pair" tiles (tile oh = cells (od,oh),(od+1,oh)) and up to 4 "single" tiles (od = 0 or 7: cells
// paired along H, see vq_tail_rows.h): twelve tile slots, tile ids 0..7 / 8..11.  The pd loop body exists twice, with and without
// the single tiles (both fully static: exact waits, no branch inside a row); 160 phases instead of 224.
// Arithmetic, fragment layout and the weight ring (global -> register for a phase -> LDS, three slices, one barrier per phase) are
// those of tail_rows16_k; a slice is 48 KB here (12 blocks).  Activations arrive as four dwordx4 per position (see below).
#pragma once
#include "vq_tail_rows.h"

constexpr int TG_SLOTS = 12;
constexpr int TG_SLICE = TG_SLOTS * 4096;                            // bytes per weight slice (one phase): up to 11 tile blocks used
constexpr size_t LDS_TAIL_GROUPS = (size_t)TR_RING * TG_SLICE;       // 144 KB
static_assert(LDS_TAIL_GROUPS <= 160 * 1024, "gfx950: 160 KB of LDS per workgroup, all of it dynamic here: the kernel must stay free of static __shared__");
constexpr int TG_PHASES = 160;                                       // (3 + 4 + 3 planes) x 4 rows x 4 positions
constexpr int TG_STREAM_SLICES = TG_PHASES + 3;                      // the kernel requests slices up to three phases ahead: padding
constexpr int TG_PIECES = TG_SLICE / 1024 / 8;                       // 1 KB pieces per wave and phase (8 waves)

constexpr unsigned tg_mask(bool with_single, int ph) { return tr_pair_mask(ph) | (with_single ? tr_single_mask(ph) << 8 : 0u); }
// groups: first plane of the pair part / plane of the single part / input planes / which pd carry the single tiles
constexpr int tg_od_pair(int g) { return 1 + 2 * g; }
constexpr int tg_od_single(int g) { return g == 0 ? 0 : 7; }
constexpr int tg_pd_lo(int g) { return g == 2 ? 1 : 0; }
constexpr int tg_pd_hi(int g) { return g == 0 ? 2 : 3; }
constexpr bool tg_with_single(int g, int pd) { return g == 0 ? pd <= 1 : g == 2 ? pd >= 2 : false; }
constexpr int tg_cell_a(int g, int tid) { return tid < 8 ? tr_cell_a(true, tg_od_pair(g), tid) : tr_cell_a(false, tg_od_single(g), tid - 8); }
constexpr int tg_cell_b(int g, int tid) { return tid < 8 ? tr_cell_b(true, tg_od_pair(g), tid) : tr_cell_b(false, tg_od_single(g), tid - 8); }
// items of a phase with N active tiles: P = ceil(N / 4) parts per fragment group (four groups of four MFMA slots), at most four
// tiles each; item j = (group j / P, part j % P) uses fragment register set j & 1
constexpr int tg_parts(int n) { return (n + 3) / 4; }
constexpr int tg_part_lo(int n, int part) { return part * n / tg_parts(n); }
constexpr int tg_part_hi(int n, int part) { return (part + 1) * n / tg_parts(n); }

// ABL (tools/ablate/tail_rows_ablate.hip only): 1 no barriers, 2 no weight streaming, 4 no LDS fragment reads, 8 no activation re-loads,
// 16 no gate multiply, 32 no epilogue, 64 no lane swap either, 128 no MFMAs, 256 activation re-loads from two fixed positions (cache hits)
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void tail_groups16_k(ConvArgs A)
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
        if (ABL & 256) pos &= 1;   // (ablation: every re-load hits the L1 / L2)
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
    // weights: slice t at t*TG_SLICE; this wave moves bytes [wave*6144, +6144) of every slice
    const vq_buf wb = buf_of(A.wfrag);
    const unsigned lane_w = (unsigned)lane * 16u;
    f32x4* const lds_w = (f32x4*)(smem_raw + wave * (TG_PIECES * 1024)) + lane;   // + slot*TG_SLICE/16 + j*64
    const f32x4* const lds_r = (const f32x4*)smem_raw + lane;                      // + slot*TG_SLICE/16 + (tile i*4 + g)*64
    const vq_buf outb = buf_of(A.out + (size_t)tile * 32 * 512);
    const float* bias = A.bias_frag;   // plain per voxel [512]

    int t = 0, sl = 0;   // phase (= slice) counter and t % 3
    auto slot_of = [&](int ahead) { const int s = sl + ahead; return s >= TR_RING ? s - TR_RING : s; };

    // ---- prologue: slices 0 and 1 into the ring, slice 2 into registers, position 0 of the first row ----
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TG_PIECES; ++j)
            lds_w[s * (TG_SLICE / 16) + j * 64] = buf_ld16(wb, lane_w + j * 1024, (unsigned)(s * TG_SLICE + wave * (TG_PIECES * 1024)));
    f32x4 wreg[TG_PIECES];   // this wave's share of the slice two phases ahead: loaded in phase t-1, written to the ring in phase t
#pragma unroll
    for (int j = 0; j < TG_PIECES; ++j) wreg[j] = buf_ld16(wb, lane_w + j * 1024, (unsigned)(2 * TG_SLICE + wave * (TG_PIECES * 1024)));
#pragma unroll
    for (int j = 0; j < 4; ++j) reload1(j, 0);   // group 0 starts at plane 0, row 0, position 0
    __builtin_amdgcn_s_waitcnt(0x0f70);          // enter the loops with nothing in flight
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
        constexpr int N = tr_popc(M);
        if constexpr (N > 0 && (ABL & 4) == 0) {
            constexpr int P = tg_parts(N), g = it / P, part = it % P;
#pragma unroll
            for (int i = tg_part_lo(N, part); i < tg_part_hi(N, part); ++i) fa[it & 1][i - tg_part_lo(N, part)] = lds_r[slot * (TG_SLICE / 16) + (i * 4 + g) * 64];
        }
    };

    f32x4 tot[TG_SLOTS];   // row sums of the group's tiles, added in row order
    // one input plane of a group: four rows x four positions, with (WS) or without the group's single tiles
    auto rowset = [&](auto ws_c, const int pd, const int p_after) {
        constexpr bool WS = decltype(ws_c)::value;
        // the first phase's slice was published by the previous barrier
        frag_req(std::integral_constant<unsigned, tg_mask(WS, 0)>{}, std::integral_constant<int, 0>{}, sl);
        frag_req(std::integral_constant<unsigned, tg_mask(WS, 0)>{}, std::integral_constant<int, 1>{}, sl);
        auto row = [&](auto phc) {
            constexpr int PH = decltype(phc)::value;
            constexpr unsigned MASK = tg_mask(WS, PH), NMASK = PH < 3 ? tg_mask(WS, PH + 1) : 0u;   // (the next plane starts afresh)
            constexpr int N = tr_popc(MASK), P = tg_parts(N), NI = 4 * P;
            const int pcur = (pd * 4 + PH) * 4;
            const int pnext = PH == 3 ? p_after : pcur + 4;
            f32x4 acc[TG_SLOTS];
            auto phase = [&](auto pwc) {
                constexpr int PW = decltype(pwc)::value;
                constexpr unsigned NEXT = PW < 3 ? MASK : NMASK;
                const int qpos = PW < 3 ? pcur + PW + 1 : pnext;   // the position the next phase reads
                t = __builtin_amdgcn_readfirstlane(t), sl = __builtin_amdgcn_readfirstlane(sl);   // (loop-carried counters: keep them scalar)
                if (!(ABL & 1)) __syncthreads();   // slice t+1 visible to every wave; every wave is done with slice t-1's slot
                arrive();
                __builtin_amdgcn_sched_barrier(0);
                const int ws = slot_of(2);
                tr_static_for<NI>([&](auto itc) {
                    constexpr int it = decltype(itc)::value;
                    constexpr int g = it / P, part = it % P, i0 = tg_part_lo(N, part), i1 = tg_part_hi(N, part);
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
                    // this item's share of the phase's memory traffic: pieces of slice t+2 (in registers since the last phase) go to the
                    // ring and their registers take the pieces of slice t+3; the next position's octets in the first half of the phase
                    if constexpr ((ABL & 2) == 0) {
                        tr_static_for<TG_PIECES>([&](auto kc) {
                            constexpr int kp = decltype(kc)::value;
                            if constexpr (kp * NI / TG_PIECES == it) {
                                lds_w[ws * (TG_SLICE / 16) + kp * 64] = wreg[kp];
                                wreg[kp] = buf_ld16(wb, lane_w + kp * 1024, (unsigned)((t + 3) * TG_SLICE + wave * (TG_PIECES * 1024)));
                            }
                        });
                    }
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
    };

#pragma nounroll
    for (int g = 0; g < 3; ++g) {
        const int pd_lo = tg_pd_lo(g), pd_hi = tg_pd_hi(g);
        const int p_next_group = g == 0 ? 0 : g == 1 ? 16 : 60;   // first row of the next group (the last group: its own last row again)
#pragma unroll
        for (int i = 0; i < TG_SLOTS; ++i) tot[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma nounroll
        for (int pd = pd_lo; pd <= pd_hi; ++pd) {
            const int p_after = pd == pd_hi ? p_next_group : (pd + 1) * 16;
            if (tg_with_single(g, pd)) rowset(std::true_type{}, pd, p_after);
            else rowset(std::false_type{}, pd, p_after);
        }
        // ---- epilogue: per-voxel bias, sigmoid, store into the caller's leaf-major [n][512] buffer (VQVAECodec.cpp:182-192) ----
        if (ABL & 32) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < TG_SLOTS; ++i) s += tot[i][0] + tot[i][3];
            if (s == 12345.678f) A.out[threadIdx.x] = s;
            continue;
        }
        const int od_p = tg_od_pair(g), od_s = tg_od_single(g);
        tr_static_for<TG_SLOTS>([&](auto tc) {
            constexpr int tid = decltype(tc)::value;
            if (tid >= 8 && g == 1) return;   // (uniform) group 1 has no single tiles
            // rows 4k .. 4k+3 of the tile: k < 2 cell A, else cell B; ow = 4(k&1) .. +3
            const int ca = tid < 8 ? od_p * 8 + tid : od_s * 8 + (tid == 11 ? 0 : 2 * (tid - 8) + 1);
            const int cb = tid < 8 ? (od_p + 1) * 8 + tid : od_s * 8 + (tid == 11 ? 7 : 2 * (tid - 8) + 2);
            const int vox = (k < 2 ? ca : cb) * 8 + (k & 1) * 4;
            const f32x4 bv = *(const f32x4*)(bias + vox);
            f32x4 sg;
            sg.x = vq_sigmoid(tot[tid].x + bv.x), sg.y = vq_sigmoid(tot[tid].y + bv.y);
            sg.z = vq_sigmoid(tot[tid].z + bv.z), sg.w = vq_sigmoid(tot[tid].w + bv.w);
            if (store) buf_st16(sg, outb, (unsigned)(jj * 512 + vox) * 4u, 0u);
        });
    }
}
