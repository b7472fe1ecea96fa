// vq_stem_taps.h — decoder front of large passes with the (tap, code) table streamed through LDS tap by tap (round 3).
//
// D0-D2 of the decoder: F.embedding + stem Conv3d(128->64,k3,p1) @4^3 (VQVAE_v2.py:257, :371-375) as look-ups in the (tap, code)
// table T[27][256][64] (build_stem_lut_k) + GroupNorm(8,64) + ReLU (:258-259) + the statistics of the result for ResidualBlock.gn1
// (:205) — what stem_fused_k does, same arithmetic contract bit for bit (valid taps ascending, plain adds from 0, + bias; statistics
// by the 16-block rule; tests/test_gpu_parity.py::test_large_path_kernel_variants_agree compares the two).
//
// stem_fused_k walks POSITIONS and gathers every tap's 256-byte table row of a position through the L1: 256 KB per leaf, 16.8 GB per
// 65 536 leaves, and that gather is bound by the L1 (80 B/clk/CU; 8.7 GB of it miss to the L2, profiles/r03_archive/r03_v1_pmc_l2_hit_miss.txt):
// 0.81 ms, 8 % of decode without one MFMA.  Here the TAPS are the outer loop.  A persistent workgroup owns a whole 32-leaf tile and ONE
// CHANNEL HALF at a time (GroupNorm groups of 8 channels never straddle the halves, so the two halves of a tile are independent
// passes): the accumulators of 32 leaves x 64 positions x 32 channels live in registers (128 VGPRs per lane), a slice of the table is
// (tap, 32 channels) = 256 codes x 128 bytes = 32 KB, and the 27 slices of a pass stream through a ring of FOUR LDS slots filled by
// LDS-DMA (global_load_lds, no staging registers) three taps ahead of their use.  The gather itself reads LDS: the 8 lanes of a leaf
// fetch one 128-byte row per access.  Table traffic: 1.77 MB per 32 leaves = 3.6 GB per 65 536 leaves, every byte an L2 hit moved
// by full-width loads.
//
// Wave w = (leaf octet w / 2, position half w % 2: positions 32h .. 32h + 31); lane = (leaf l of the octet, 16-byte chunk c of the
// channel half).  Loop nest: (kd, kh) at run time (9 iterations: the eight neighbour rows' codes arrive as one aligned dword each
// from the tile's index block in LDS), kw static (the three neighbour codes of a position are bytes of that dword), rows and positions
// static (accumulator registers need static indices); a row whose neighbour row lies outside the leaf is skipped by a wave-uniform
// branch — zero-padding taps are skipped exactly, as everywhere.  One s_barrier per tap: it publishes slice t (every wave has waited
// for its own DMA pieces: vmcnt with the two newer slices still in flight) and frees the slot of slice t - 1 for slice t + 3.  The
// LDS reads of the gather go through lds_frag_read / lds_frags_wait (vq_kernels.h): behind an LDS-DMA in flight the compiler answers
// every LDS read it can see with s_waitcnt vmcnt(0) (DESIGN 3b rule 3).
//
// MEASURED (65 536 leaves): 0.81 ms (stem_fused_k) -> 0.57 ms; decode 6.64 -> 6.79 M leaves/s.  The tap loop is LDS-bound now: two
// leaves share a 16-lane access group, their rows collide in the banks for half of the code pairs (1.5x), 8 waves x ~15 row reads +
// the DMA's 32 KB per tap.  First cut (kept in the history, not in the tree): 16 leaves per workgroup, 64 KB slices, two windows
// filled through registers — one tap of look-ahead against an L2 round trip of thousands of cycles when 256 CUs ask for the same
// slice at once, and ~700 cycles of gather to hide it behind: 0.91 ms, slower than the L1 gather.
#pragma once
#include "vq_kernels.h"

constexpr size_t LDS_STEM_TAPS = (size_t)4 * 32768 + 32 * 128 + 4 * 64 * 2 * sizeof(double);

__device__ __forceinline__ unsigned lds_u32_read(unsigned byte_addr)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}

// lgkmcnt-only wait: LDS reads return in order, so "at most N outstanding" = "all but the N youngest have landed"
template <int N>
__device__ __forceinline__ void lds_frags_wait_n() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// PIPE: rows of a tap whose gather reads are in flight ahead of the row being accumulated (0 = read, wait, add, row by row — round 3).
// HMAP: wave -> (leaf octet, position half): 0 = (w >> 1, w & 1); 1 = (w & 3, w >> 2), so that the two waves of a SIMD (w and w + 4)
// are the two position halves of one octet — for kd = 0 / kd = 2 one half has half the rows of the other, and the per-tap barrier
// makes a SIMD that hosts two light waves wait for one that hosts two heavy ones.
// RELAX: the first three taps of a pass wait for their own table slices only, not for the previous pass's output stores (which are
// younger than those slices in the wave's vector-memory queue: vmcnt counts in order).  STAG: workgroups start in four groups
// STAG x 4096 cycles apart, so that the 256 KB output bursts of the chip's 256 workgroups — every pass ends with one, and the passes
// of a persistent launch are equally long, i.e. stay in lockstep — do not hit HBM at the same moment.
// CHK: which 32 of the tile's 64 positions a wave owns.  false: a position half (planes 2h, 2h+1) — for kd = 0 and kd = 2 one half then has
// half the valid rows of the other, and the per-tap barrier makes the lighter waves wait (no barrier: 0.57 -> 0.49 ms).  true: the rows
// (od, oh) with (od + oh) % 2 == h, a checkerboard over the 16 rows: both waves of an octet have 4-5 or 6 or 8 valid rows in every tap.
// ABL (tools/ablate only): 1 no gather reads, 2 no adds, 4 no table DMA, 8 no epilogue, 16 no per-tap barrier, 32 no output stores.
template <int PIPE = 0, int HMAP = 0, int ABL = 0, bool RELAX = false, int STAG = 0, bool CHK = false>
__global__ __launch_bounds__(512, 2) void stem_taps_k(StemFusedArgs A)
{
    static_assert(ABL == 0 || VQ_ABLATE, "ABL is a timing-only ablation switch (tools/ablate, -DVQ_ABLATE=1)");
    static_assert(PIPE >= 0 && PIPE <= 2, "0, 1 or 2 rows ahead");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* ring = (f32x4*)smem_raw;                                  // [4 slots][256 codes][8 chunks]
    unsigned char* sidx = smem_raw + 4 * 32768;                      // [32 leaves][128]: codes at bytes 32 .. 95
    double* xch = (double*)(smem_raw + 4 * 32768 + 32 * 128);        // [4 octets][64 lanes][2]
    const unsigned ring_off = lds_offset_of(ring), sidx_off = lds_offset_of(sidx);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oct = HMAP ? (wave & 3) : (wave >> 1), h = HMAP ? (wave >> 2) : (wave & 1);
    const int l = lane >> 3, c = lane & 7;
    const int jt = 8 * oct + l;                                       // leaf of this lane in the tile
    const unsigned nb_addr = sidx_off + (unsigned)(jt * 128 + 32 + (CHK ? 0 : 32 * h));
    // row r = 0..7 of this wave -> (od, oh).  CHK: od = r / 2, oh = 2 (r % 2) + (od + h) % 2 (h is wave-uniform: scalar arithmetic)
    auto od_of = [&](int r) { return CHK ? (r >> 1) : 2 * h + (r >> 2); };
    auto oh_of = [&](int r) { return CHK ? 2 * (r & 1) + (((r >> 1) + h) & 1) : (r & 3); };
    double* xq = xch + ((size_t)oct * 64 + lane) * 2;
    // slice (t, half) -> slot: DMA instruction i of a slice covers codes 8i .. 8i+7 (lane = (code, chunk)); wave w issues i = w, w+8, w+16, w+24.
    // Buffer addressing: descriptor in SGPRs, one constant per-lane offset, the rest wave-uniform (a per-lane 64-bit pointer per piece
    // costs register pairs the accumulators need)
    const vq_buf tb = buf_of(A.T);
    const unsigned lane_t = (unsigned)((lane >> 3) * 64 + (lane & 7) * 4) * 4u;
    // vector-memory operations per wave the RELAX wait below counts on: DMA instructions per slice, output stores per pass
    constexpr int DMA_PER_SLICE = 4, STORES_PER_PASS = 8 * 4;
    auto issue_slice = [&](int t, int half, int slot) {
        if (ABL & 4) return;
#pragma unroll
        for (int k = 0; k < DMA_PER_SLICE; ++k) {
            const int i = k * 8 + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tb, (__attribute__((address_space(3))) void*)(ring + slot * 2048 + i * 64), 16, (int)lane_t,
                                                 (int)(((unsigned)t * 16384u + (unsigned)i * 512u + (unsigned)half * 32u) * 4u), 0, 0);
        }
    };
    // the tile's codes -> LDS (four bytes per thread)
    auto stage_codes = [&](int tile) {
        const int li = tid >> 4, b0 = (tid & 15) * 4;
        const int64_t leaf = (int64_t)tile * 32 + li;
#pragma unroll
        for (int b = 0; b < 4; ++b) sidx[li * 128 + 32 + b0 + b] = leaf < A.n_leaves ? A.idx[leaf * 64 + b0 + b] : (unsigned char)0;
    };
    if (STAG > 0) {
        const int g = ((int)blockIdx.x >> 3) & 3;   // (eight consecutive workgroups = one per XCD)
        for (int i = 0; i < g * STAG; ++i) __builtin_amdgcn_s_sleep(64);   // 64 x 64 cycles
    }
    bool after_stores = false;   // (wave-uniform) has this wave's queue output stores behind the first three slices of the pass?
    if ((int)blockIdx.x < A.n_tiles) {
        stage_codes(blockIdx.x);
        issue_slice(0, 0, 0);
        issue_slice(1, 0, 1);
        issue_slice(2, 0, 2);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        for (int half = 0; half < 2; ++half) {
            const int cg = 8 * half + c;                              // this lane's chunk of the 16
            const f32x4 b4 = ((const f32x4*)A.bias)[cg];
            f32x4 acc[32];
#pragma unroll
            for (int p = 0; p < 32; ++p) acc[p] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

            for (int kdkh = 0; kdkh < 9; ++kdkh) {
                const int kd = kdkh / 3, kh = kdkh - 3 * kd;
                unsigned nb[8];
                {
                    const unsigned a0 = nb_addr + (unsigned)((kd - 1) * 16 + (kh - 1) * 4);
#pragma unroll
                    for (int r = 0; r < 8; ++r) nb[r] = lds_u32_read(a0 + (CHK ? (unsigned)(16 * od_of(r) + 4 * oh_of(r)) : 4u * r));
                }
                bool rv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) rv[r] = (unsigned)(od_of(r) + kd - 1) < 4u && (unsigned)(oh_of(r) + kh - 1) < 4u;
                auto tap = [&](auto KW) {   // kw as a compile-time constant: the reads per row (and the wait counts) depend on it
                    constexpr int kw = decltype(KW)::value;
                    const int t = kdkh * 3 + kw;
                    // this wave's pieces of slice t have landed (two newer slices = 8 DMA instructions may still be in flight; the first
                    // three slices of a pass were requested before the previous pass's epilogue, whose stores are younger: wait for all)
                    // RELAX: slices 0-2 were requested BEFORE the previous pass's 32 output stores per wave: "all but the youngest 8 + 32"
                    if (RELAX && t <= 2) {
                        // vmcnt(2 younger slices + the pass's stores): the epilogue below must issue exactly STORES_PER_PASS stores per wave
                        // (unconditional, one per (row, pw)) between the requests of slices 0-2 and this wait
                        static_assert(2 * DMA_PER_SLICE + STORES_PER_PASS == 40, "the RELAX wait is vmcnt(40): update the immediate with the counts");
                        if (after_stores && !A.ystem_dbg) __builtin_amdgcn_s_waitcnt(0x8F78);   // vmcnt(40)
                        else if (t == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
                        else __builtin_amdgcn_s_waitcnt(0x0F78);
                    } else
                    if (t == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
                    else if (t <= 24) __builtin_amdgcn_s_waitcnt(0x0F78);
                    else if (t == 25) __builtin_amdgcn_s_waitcnt(0x0F74);
                    else __builtin_amdgcn_s_waitcnt(0x0F70);
                    if (!(ABL & 16)) __builtin_amdgcn_s_barrier();   // slice t visible to every wave; every wave is done with slice t - 1's slot
                    asm volatile("" ::: "memory");
                    if (t + 3 <= 26) issue_slice(t + 3, half, (t + 3) & 3);
                    const unsigned wbase = ring_off + (unsigned)(t & 3) * 32768u + (unsigned)c * 16u;
                    if (kw == 0) {   // the codes requested above are first needed here
                        lds_frags_wait();
#pragma unroll
                        for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(nb[r]));
                    }
                    // one row's gather: the (up to) four neighbour codes' 16-byte chunks of this lane; NRD reads per row (static per kw)
                    constexpr int NRD = kw == 1 ? 4 : 3;
                    f32x4 rowb[PIPE + 1][4];
                    auto issue_row = [&](int r, f32x4 (&row)[4]) {
#pragma unroll
                        for (int pw = 0; pw < 4; ++pw) {
                            const int qw = pw + kw - 1;
                            if (qw < 0 || qw > 3) continue;   // (static) zero padding along w
                            if (ABL & 1) row[pw] = b4;
                            else row[pw] = lds_frag_read(wbase + ((nb[r] >> (8 * qw)) & 0xffu) * 128u);
                        }
                    };
                    auto add_row = [&](int r, f32x4 (&row)[4]) {
#pragma unroll
                        for (int pw = 0; pw < 4; ++pw) {
                            const int qw = pw + kw - 1;
                            if (qw < 0 || qw > 3) continue;
                            lds_frag_use(row[pw]);
                            if (ABL & 2) acc[4 * r + pw].x = row[pw].x;
                            else acc[4 * r + pw] = acc[4 * r + pw] + row[pw];
                        }
                    };
                    if (PIPE == 0) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if (rv[r]) {   // (wave-uniform)
                                issue_row(r, rowb[0]);
                                lds_frags_wait();
                                add_row(r, rowb[0]);
                            }
                        }
                    } else {
                        // rows PIPE ahead: row r's reads are issued, then row r - PIPE is accumulated once everything older than the
                        // (valid) rows issued since has landed.  Buffers are indexed statically by r; rv[] is wave-uniform, so the
                        // wait counts are picked by scalar branches.  Per accumulator the taps still arrive in ascending order.
#pragma unroll
                        for (int r = 0; r < 8 + PIPE; ++r) {
                            if (r < 8 && rv[r]) issue_row(r, rowb[r % (PIPE + 1)]);
                            const int rc = r - PIPE;
                            if (rc >= 0 && rv[rc]) {
                                int younger = 0;   // valid rows issued after rc (each NRD reads)
#pragma unroll
                                for (int k = 1; k <= PIPE; ++k) younger += (rc + k < 8 && rv[rc + k]) ? 1 : 0;
                                if (younger == 0) lds_frags_wait_n<0>();
                                else if (younger == 1) lds_frags_wait_n<NRD>();
                                else lds_frags_wait_n<2 * NRD>();
                                add_row(rc, rowb[rc % (PIPE + 1)]);
                            }
                        }
                    }
                };
                tap(std::integral_constant<int, 0>{});
                tap(std::integral_constant<int, 1>{});
                tap(std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();   // every wave is done with the ring and with the tile's codes; no DMA in flight
            // (this pass's GroupNorm affine first: requested after the slices, their first use would wait for the slices as well)
            const f32x4 gm4 = ((const f32x4*)A.gamma)[cg], bt4 = ((const f32x4*)A.beta)[cg];
            // the NEXT pass's first three slices (and, at a tile change, its codes) travel while this pass's statistics and stores run
            {
                const int ntile = half == 0 ? tile : tile + (int)gridDim.x;
                if (ntile < A.n_tiles) {
                    if (half == 1) stage_codes(ntile);
                    issue_slice(0, half ^ 1, 0);
                    issue_slice(1, half ^ 1, 1);
                    issue_slice(2, half ^ 1, 2);
                }
            }

            if (ABL & 8) {   // keep the accumulators alive, nothing else
                float t = 0.0f;
#pragma unroll
                for (int p = 0; p < 32; ++p) t += acc[p].x + acc[p].w;
                if (t == 12345.678f) A.d2[tid] = t;
                __syncthreads();
                continue;
            }
            // ---- statistics of y = acc + bias (as in stem_fused_k) ----
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
            // the sixteen block sums (block = output row od*4 + oh) in block order.  Position halves: wave h = 0 holds blocks 0..7, h = 1 blocks
            // 8..15 — a running total handed from one to the other.  Checkerboard: the owners alternate, so the h = 1 wave passes its eight
            // sums through the ring's fourth slot (free until the next pass's tap 0 has passed its barrier) and the h = 0 wave adds all sixteen.
            auto ordered_totals = [&](double& S, double& Q) {
                if (CHK) {
                    double* sc = (double*)(ring + 3 * 2048) + (size_t)oct * 8 * 64 * 2 + lane;   // [oct][row 8][s | q][lane]
                    if (h == 1) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) sc[(r * 2 + 0) * 64] = bs[r], sc[(r * 2 + 1) * 64] = bq[r];
                    }
                    __syncthreads();
                    if (h == 0) {
                        S = 0.0, Q = 0.0;
#pragma unroll
                        for (int b = 0; b < 16; ++b) {
                            const int od = b >> 2, oh = b & 3, r = od * 2 + (oh >> 1);   // (static) the owner's row number
                            if (((od + oh) & 1) == 0) S += bs[r], Q += bq[r];
                            else S += sc[(r * 2 + 0) * 64], Q += sc[(r * 2 + 1) * 64];
                        }
                        xq[0] = S, xq[1] = Q;
                    }
                    __syncthreads();
                    if (h == 1) S = xq[0], Q = xq[1];
                    return;
                }
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
                __syncthreads();
            };
            float ia[4], ib[4];
            {
                double S, Q;
                ordered_totals(S, Q);
                const double S2 = __shfl_xor(S, 1, 64), Q2 = __shfl_xor(Q, 1, 64);
                float m, r;
                gn_finish((c & 1) ? S2 + S : S + S2, (c & 1) ? Q2 + Q : Q + Q2, 1.0 / 512.0, m, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ia[i] = r * gm4[i];
                    ib[i] = __builtin_fmaf(-m, ia[i], bt4[i]);
                }
            }
            {
                const vq_buf outb = buf_of((const f32x4*)A.d2 + (size_t)tile * 64 * 16 * 32);
                const vq_buf dbgb = buf_of(A.ystem_dbg ? (const f32x4*)A.ystem_dbg + (size_t)tile * 64 * 16 * 32 : (const f32x4*)A.d2);
                const unsigned lane_o = (unsigned)(cg * 32 + jt) * 16u;
                static_assert(STORES_PER_PASS == 8 * 4, "one output store per (row, pw): the RELAX wait of the tap loop counts them");
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    GnAcc st;
                    st.init();
#pragma unroll
                    for (int pw = 0; pw < 4; ++pw) {
                        const f32x4 v = acc[4 * r + pw];
                        const unsigned po = (unsigned)(16 * od_of(r) + 4 * oh_of(r) + pw);
                        if (A.ystem_dbg) buf_st16(v, dbgb, lane_o, po * 8192u);
                        f32x4 y;
                        y.x = fmaxf(__builtin_fmaf(v.x, ia[0], ib[0]), 0.0f);
                        y.y = fmaxf(__builtin_fmaf(v.y, ia[1], ib[1]), 0.0f);
                        y.z = fmaxf(__builtin_fmaf(v.z, ia[2], ib[2]), 0.0f);
                        y.w = fmaxf(__builtin_fmaf(v.w, ia[3], ib[3]), 0.0f);
                        if (!(ABL & 32)) buf_st16_nt(y, outb, lane_o, po * 8192u);
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
                    A.out_mean[((size_t)tile * 8 + (cg >> 1)) * 32 + jt] = m;
                    A.out_rstd[((size_t)tile * 8 + (cg >> 1)) * 32 + jt] = r;
                }
            }
            after_stores = true;
            __syncthreads();   // the next pass reads the codes staged above
        }
    }
}
