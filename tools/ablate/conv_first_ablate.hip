// Where does conv_first_k's time go?  Timing-only variants of the kernel (results are garbage by design), one launch
// each at 2048 tiles.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vqvdb_amd/csrc tools/ablate/conv_first_ablate.hip -o gpurun_out/ablate_cf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "vq_kernels.h"

// ABL bits: 1 = no input loads inside the loop, 2 = no MFMAs, 4 = no statistics, 8 = no zero-select (cndmask), 16 = no stores
template <int MODE, int ABL, int WPS>
__global__ __launch_bounds__(256, WPS) void cf_k(ConvArgs A, const int4* __restrict__ steps)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    const float* x = A.in + (size_t)tile * 512 * 32 + jj;
    f32x4* out4 = A.out ? (f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj : nullptr;
    int off[8];
    bool ok[8];
#pragma unroll
    for (int ow = 0; ow < 8; ++ow) {
        const int iw = ow + q4 - 1;
        ok[ow] = (q4 < 3) && iw >= 0 && iw < 8;
        off[ow] = (ok[ow] ? iw : 0) * 32;
    }
    float ia[2][4], ib[2][4];
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
    GnAcc st[2][2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < 2; ++k) st[sb][k].init();
    const int NS = A.n_steps;
    int si = 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    float xn[3][8][2];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int rb = max(0, min(e.x + (kh - 1) * 8, 504)) * 32;
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) {
            xn[kh][ow][0] = x[rb + off[ow]];
            xn[kh][ow][1] = x[rb + off[ow] + 16];
        }
    }
    for (int row = 0; row < 64; ++row) {
        f32x4 acc[8][2];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow][0] = acc[ow][1] = (f32x4){0, 0, 0, 0};
        bool last;
        do {
            float xc[3][8][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int ow = 0; ow < 8; ++ow) {
                    xc[kh][ow][0] = (ABL & 8) ? xn[kh][ow][0] : (ok[ow] ? xn[kh][ow][0] : 0.0f);
                    xc[kh][ow][1] = (ABL & 8) ? xn[kh][ow][1] : (ok[ow] ? xn[kh][ow][1] : 0.0f);
                }
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
            if (!(ABL & 1)) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int rb = max(0, min(en.x + (kh - 1) * 8, 504)) * 32;
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) {
                        xn[kh][ow][0] = x[rb + off[ow]];
                        xn[kh][ow][1] = x[rb + off[ow] + 16];
                    }
                }
            }
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
                            acc[ow][0].x += wv * xc[kh][ow][0];
                            acc[ow][1].x += wv * xc[kh][ow][1];
                        } else {
                            acc[ow][0] = mfma16(wv, xc[kh][ow][0], acc[ow][0]);
                            acc[ow][1] = mfma16(wv, xc[kh][ow][1], acc[ow][1]);
                        }
                    }
                }
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            float rs0 = 0.0f, rq0 = 0.0f, rs1 = 0.0f, rq1 = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                f32x4 v = acc[ow][sb] + bias4;
                if (MODE == 1) {
                    v.x = fmaxf(__builtin_fmaf(v.x, ia[sb][0], ib[sb][0]), 0.0f);
                    v.y = fmaxf(__builtin_fmaf(v.y, ia[sb][1], ib[sb][1]), 0.0f);
                    v.z = fmaxf(__builtin_fmaf(v.z, ia[sb][2], ib[sb][2]), 0.0f);
                    v.w = fmaxf(__builtin_fmaf(v.w, ia[sb][3], ib[sb][3]), 0.0f);
                }
                if (MODE == 1 && !(ABL & 16)) out4[((size_t)(row * 8 + ow) * 4) * 32 + 16 * sb] = v;
                if (ABL & 4) {
                    rs0 += v.x + v.y + v.z + v.w;   // keep the values alive
                } else {
                    rs0 = rs0 + v.x, rq0 = __builtin_fmaf(v.x, v.x, rq0);
                    rs0 = rs0 + v.y, rq0 = __builtin_fmaf(v.y, v.y, rq0);
                    if (MODE == 1) {
                        rs1 = rs1 + v.z, rq1 = __builtin_fmaf(v.z, v.z, rq1);
                        rs1 = rs1 + v.w, rq1 = __builtin_fmaf(v.w, v.w, rq1);
                    } else {
                        rs0 = rs0 + v.z, rq0 = __builtin_fmaf(v.z, v.z, rq0);
                        rs0 = rs0 + v.w, rq0 = __builtin_fmaf(v.w, v.w, rq0);
                    }
                }
            }
            st[sb][0].add_row(rs0, rq0);
            if (MODE == 1) st[sb][1].add_row(rs1, rq1);
        }
        if ((row & 3) == 3) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < 2; ++k) st[sb][k].fold();
        }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) {
            float m, r;
            gn_finish(st[sb][k].s, st[sb][k].q, 1.0 / 1024.0, m, r);
            A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
        }
}

// LOADMODE 1: input in the row layout xr[tile][row 64][leaf 32][12] = (0, x0..x7, 0, pad, pad); lane (q4, jj) reads the 8 floats
// starting at element min(q4, 2): its B operands x[ow + q4 - 1] for ow = 0..7, halo zeros included -> no selects, 12 wide loads per step
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void cf2_k(ConvArgs A, const int4* __restrict__ steps)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    const float* x = A.in + (size_t)tile * 64 * 32 * 12 + jj * 12 + (q4 < 3 ? q4 : 2);
    f32x4* out4 = A.out ? (f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj : nullptr;
    float ia[2][4], ib[2][4];
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
    GnAcc st[2][2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < 2; ++k) st[sb][k].init();
    const int NS = A.n_steps;
    int si = 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    f32x4 xn[3][2][2];   // [kh][sb][half row]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int r = max(0, min((e.x >> 3) + (kh - 1), 63));
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            xn[kh][sb][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
            xn[kh][sb][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
        }
    }
    for (int row = 0; row < 64; ++row) {
        f32x4 acc[8][2];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow][0] = acc[ow][1] = (f32x4){0, 0, 0, 0};
        bool last;
        do {
            f32x4 xc[3][2][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) xc[kh][sb][0] = xn[kh][sb][0], xc[kh][sb][1] = xn[kh][sb][1];
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = max(0, min((en.x >> 3) + (kh - 1), 63));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    xn[kh][sb][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
                    xn[kh][sb][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
                }
            }
            const float w0 = e.y == 0 ? w[0] : (e.y == 1 ? w[3] : w[6]);
            const float w1 = e.y == 0 ? w[1] : (e.y == 1 ? w[4] : w[7]);
            const float w2 = e.y == 0 ? w[2] : (e.y == 1 ? w[5] : w[8]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if ((e.w >> (8 + kh)) & 1) {
                    const float wv = kh == 0 ? w0 : (kh == 1 ? w1 : w2);
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) {
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
        for (int sb = 0; sb < 2; ++sb) {
            float rs0 = 0.0f, rq0 = 0.0f, rs1 = 0.0f, rq1 = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                f32x4 v = acc[ow][sb] + bias4;
                if (MODE == 1) {
                    v.x = fmaxf(__builtin_fmaf(v.x, ia[sb][0], ib[sb][0]), 0.0f);
                    v.y = fmaxf(__builtin_fmaf(v.y, ia[sb][1], ib[sb][1]), 0.0f);
                    v.z = fmaxf(__builtin_fmaf(v.z, ia[sb][2], ib[sb][2]), 0.0f);
                    v.w = fmaxf(__builtin_fmaf(v.w, ia[sb][3], ib[sb][3]), 0.0f);
                }
                if (MODE == 1) out4[((size_t)(row * 8 + ow) * 4) * 32 + 16 * sb] = v;
                rs0 = rs0 + v.x, rq0 = __builtin_fmaf(v.x, v.x, rq0);
                rs0 = rs0 + v.y, rq0 = __builtin_fmaf(v.y, v.y, rq0);
                if (MODE == 1) {
                    rs1 = rs1 + v.z, rq1 = __builtin_fmaf(v.z, v.z, rq1);
                    rs1 = rs1 + v.w, rq1 = __builtin_fmaf(v.w, v.w, rq1);
                } else {
                    rs0 = rs0 + v.z, rq0 = __builtin_fmaf(v.z, v.z, rq0);
                    rs0 = rs0 + v.w, rq0 = __builtin_fmaf(v.w, v.w, rq0);
                }
            }
            st[sb][0].add_row(rs0, rq0);
            if (MODE == 1) st[sb][1].add_row(rs1, rq1);
        }
        if ((row & 3) == 3) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < 2; ++k) st[sb][k].fold();
        }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) {
            float m, r;
            gn_finish(st[sb][k].s, st[sb][k].q, 1.0 / 1024.0, m, r);
            A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
        }
}

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void cf4_k(ConvArgs A, const int4* __restrict__ steps)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    const float* x = A.in + (size_t)tile * 64 * 32 * 12 + jj * 12 + (q4 < 3 ? q4 : 2);
    f32x4* out4 = A.out ? (f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj : nullptr;
    float ia[2][4], ib[2][4];
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
    GnAcc st[2][2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < 2; ++k) st[sb][k].init();
    const int NS = A.n_steps;
    int si = 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    f32x4 xn[3][2][2];   // [kh][sb][half row]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int r = max(0, min((e.x >> 3) + (kh - 1), 63));
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            xn[kh][sb][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
            xn[kh][sb][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
        }
    }
    for (int row = 0; row < 64; ++row) {
        f32x4 acc[8][2];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow][0] = acc[ow][1] = (f32x4){0, 0, 0, 0};
        bool last;
        do {
            f32x4 xc[3][2][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) xc[kh][sb][0] = xn[kh][sb][0], xc[kh][sb][1] = xn[kh][sb][1];
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = max(0, min((en.x >> 3) + (kh - 1), 63));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    xn[kh][sb][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
                    xn[kh][sb][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
                }
            }
            const float w0 = e.y == 0 ? w[0] : (e.y == 1 ? w[3] : w[6]);
            const float w1 = e.y == 0 ? w[1] : (e.y == 1 ? w[4] : w[7]);
            const float w2 = e.y == 0 ? w[2] : (e.y == 1 ? w[5] : w[8]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if ((e.w >> (8 + kh)) & 1) {
                    const float wv = kh == 0 ? w0 : (kh == 1 ? w1 : w2);
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) {
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
        for (int sb = 0; sb < 2; ++sb) {
            float rs0 = 0.0f, rq0 = 0.0f, rs1 = 0.0f, rq1 = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                f32x4 v = acc[ow][sb] + bias4;
                if (MODE == 1) {
                    v.x = fmaxf(__builtin_fmaf(v.x, ia[sb][0], ib[sb][0]), 0.0f);
                    v.y = fmaxf(__builtin_fmaf(v.y, ia[sb][1], ib[sb][1]), 0.0f);
                    v.z = fmaxf(__builtin_fmaf(v.z, ia[sb][2], ib[sb][2]), 0.0f);
                    v.w = fmaxf(__builtin_fmaf(v.w, ia[sb][3], ib[sb][3]), 0.0f);
                }
                if (MODE == 1) out4[((size_t)(row * 8 + ow) * 4) * 32 + 16 * sb] = v;
                st[sb][0].add(v.x), st[sb][0].add(v.y);
                if (MODE == 1) st[sb][1].add(v.z), st[sb][1].add(v.w);
                else st[sb][0].add(v.z), st[sb][0].add(v.w);
            }
        }
        if ((row & 3) == 3) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < 2; ++k) st[sb][k].fold();
        }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) {
            float m, r;
            gn_finish(st[sb][k].s, st[sb][k].q, 1.0 / 1024.0, m, r);
            A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
        }
}

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void cf3_k(ConvArgs A, const int4* __restrict__ steps)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = blockIdx.x * 4 + wave;
    const int tile = half >> 1, SB = half & 1;
    if (tile >= A.n_tiles) return;
    const int jj = lane & 15, q4 = lane >> 4;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = A.wfrag[t * 64 + lane];
    const f32x4 bias4 = ((const f32x4*)A.bias_frag)[q4];
    const float* x = A.in + (size_t)tile * 64 * 32 * 12 + jj * 12 + (q4 < 3 ? q4 : 2);
    f32x4* out4 = A.out ? (f32x4*)A.out + (size_t)tile * 512 * 4 * 32 + q4 * 32 + jj : nullptr;
    float ia[1][4], ib[1][4];
    if (MODE == 1) {
#pragma unroll
        for (int sb = SB; sb < SB + 1; ++sb) {
            const float mean = A.in_mean[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
            const float rstd = A.in_rstd[((size_t)tile * 4 + q4) * 32 + 16 * sb + jj];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ia[0][i] = rstd * A.in_gamma[4 * q4 + i];
                ib[0][i] = __builtin_fmaf(-mean, ia[0][i], A.in_beta[4 * q4 + i]);
            }
        }
    }
    GnAcc st[1][2];
    st[0][0].init(), st[0][1].init();
    const int NS = A.n_steps;
    int si = 0;
    int4 e = steps[si];
    int4 en = steps[si + 1];
    f32x4 xn[3][1][2];   // [kh][sb][half row]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int r = max(0, min((e.x >> 3) + (kh - 1), 63));
#pragma unroll
        for (int sb = SB; sb < SB + 1; ++sb) {
            xn[kh][0][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
            xn[kh][0][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
        }
    }
    for (int row = 0; row < 64; ++row) {
        f32x4 acc[8][1];
#pragma unroll
        for (int ow = 0; ow < 8; ++ow) acc[ow][0] = (f32x4){0, 0, 0, 0};
        bool last;
        do {
            f32x4 xc[3][1][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int sb = SB; sb < SB + 1; ++sb) xc[kh][0][0] = xn[kh][0][0], xc[kh][0][1] = xn[kh][0][1];
            const int4 en2 = steps[si + 2 < NS ? si + 2 : NS - 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = max(0, min((en.x >> 3) + (kh - 1), 63));
#pragma unroll
                for (int sb = SB; sb < SB + 1; ++sb) {
                    xn[kh][0][0] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12);
                    xn[kh][0][1] = *(const f32x4u*)(x + ((size_t)r * 32 + 16 * sb) * 12 + 4);
                }
            }
            const float w0 = e.y == 0 ? w[0] : (e.y == 1 ? w[3] : w[6]);
            const float w1 = e.y == 0 ? w[1] : (e.y == 1 ? w[4] : w[7]);
            const float w2 = e.y == 0 ? w[2] : (e.y == 1 ? w[5] : w[8]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if ((e.w >> (8 + kh)) & 1) {
                    const float wv = kh == 0 ? w0 : (kh == 1 ? w1 : w2);
#pragma unroll
                    for (int ow = 0; ow < 8; ++ow) {
                        acc[ow][0] = mfma16(wv, xc[kh][0][ow >> 2][ow & 3], acc[ow][0]);
                    }
                }
            }
            last = (e.w & 2) != 0;
            e = en;
            en = en2;
            ++si;
        } while (!last);
#pragma unroll
        for (int sb = SB; sb < SB + 1; ++sb) {
            float rs0 = 0.0f, rq0 = 0.0f, rs1 = 0.0f, rq1 = 0.0f;
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                f32x4 v = acc[ow][0] + bias4;
                if (MODE == 1) {
                    v.x = fmaxf(__builtin_fmaf(v.x, ia[0][0], ib[0][0]), 0.0f);
                    v.y = fmaxf(__builtin_fmaf(v.y, ia[0][1], ib[0][1]), 0.0f);
                    v.z = fmaxf(__builtin_fmaf(v.z, ia[0][2], ib[0][2]), 0.0f);
                    v.w = fmaxf(__builtin_fmaf(v.w, ia[0][3], ib[0][3]), 0.0f);
                }
                if (MODE == 1) out4[((size_t)(row * 8 + ow) * 4) * 32 + 16 * sb] = v;
                rs0 = rs0 + v.x, rq0 = __builtin_fmaf(v.x, v.x, rq0);
                rs0 = rs0 + v.y, rq0 = __builtin_fmaf(v.y, v.y, rq0);
                if (MODE == 1) {
                    rs1 = rs1 + v.z, rq1 = __builtin_fmaf(v.z, v.z, rq1);
                    rs1 = rs1 + v.w, rq1 = __builtin_fmaf(v.w, v.w, rq1);
                } else {
                    rs0 = rs0 + v.z, rq0 = __builtin_fmaf(v.z, v.z, rq0);
                    rs0 = rs0 + v.w, rq0 = __builtin_fmaf(v.w, v.w, rq0);
                }
            }
            st[0][0].add_row(rs0, rq0);
            if (MODE == 1) st[0][1].add_row(rs1, rq1);
        }
        if ((row & 3) == 3) {
#pragma unroll
            for (int sb = SB; sb < SB + 1; ++sb)
#pragma unroll
                for (int k = 0; k < 2; ++k) st[0][k].fold();
        }
    }
#pragma unroll
    for (int sb = SB; sb < SB + 1; ++sb)
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 2 : 1); ++k) {
            float m, r;
            gn_finish(st[0][k].s, st[0][k].q, 1.0 / 1024.0, m, r);
            A.out_mean[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = m;
            A.out_rstd[((size_t)tile * 8 + 2 * q4 + k) * 32 + 16 * sb + jj] = r;
        }
}

static std::vector<int> steps_rows8_kd()
{
    std::vector<int> t;
    for (int od = 0; od < 8; ++od)
        for (int oh = 0; oh < 8; ++oh) {
            const size_t first = t.size();
            for (int kd = 0; kd < 3; ++kd) {
                const int id = od + kd - 1;
                if (id < 0 || id > 7) continue;
                int mask = 0;
                for (int kh = 0; kh < 3; ++kh)
                    if (oh + kh - 1 >= 0 && oh + kh - 1 <= 7) mask |= 1 << kh;
                t.insert(t.end(), {(id * 8 + oh) * 8, kd, (od * 8 + oh) * 8, mask << 8});
            }
            t[first + 3] |= 1;
            t[t.size() - 1] |= 2;
        }
    return t;
}

template <typename K>
static void run(const char* name, K k, ConvArgs A, const int4* steps, int nt)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3((nt + 3) / 4), dim3(256), 0, 0, A, steps);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3((nt + 3) / 4), dim3(256), 0, 0, A, steps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.4f ms  (%s)\n", name, ms / 10, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int nt = 2048;
    float *x, *out, *mean, *rstd, *om, *orr, *w, *bias, *gam, *bet;
    hipMalloc(&x, (size_t)nt * 64 * 32 * 12 * 4 + 64);
    hipMalloc(&out, (size_t)nt * 512 * 16 * 32 * 4);
    hipMalloc(&mean, (size_t)nt * 8 * 32 * 4), hipMalloc(&rstd, (size_t)nt * 8 * 32 * 4);
    hipMalloc(&om, (size_t)nt * 8 * 32 * 4), hipMalloc(&orr, (size_t)nt * 8 * 32 * 4);
    hipMalloc(&w, 9 * 64 * 4), hipMalloc(&bias, 64), hipMalloc(&gam, 64), hipMalloc(&bet, 64);
    hipMemset(x, 0, (size_t)nt * 64 * 32 * 12 * 4 + 64);
    hipMemset(mean, 0, (size_t)nt * 8 * 32 * 4), hipMemset(rstd, 0, (size_t)nt * 8 * 32 * 4);
    hipMemset(w, 0, 9 * 64 * 4), hipMemset(bias, 0, 64), hipMemset(gam, 0, 64), hipMemset(bet, 0, 64);
    std::vector<int> t = steps_rows8_kd();
    int4* steps;
    hipMalloc(&steps, t.size() * 4);
    hipMemcpy(steps, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    ConvArgs A{};
    A.in = x, A.out = out, A.wfrag = w, A.bias_frag = bias, A.in_mean = mean, A.in_rstd = rstd, A.in_gamma = gam, A.in_beta = bet;
    A.out_mean = om, A.out_rstd = orr, A.n_tiles = nt, A.n_steps = (int)t.size() / 4;
#define R(MODE, ABL, WPS) run("MODE " #MODE " ABL " #ABL " waves/SIMD " #WPS, cf_k<MODE, ABL, WPS>, A, steps, nt)
    R(0, 0, 2); R(0, 1, 2); R(0, 2, 2); R(0, 4, 2); R(0, 8, 2); R(0, 3, 2); R(0, 7, 2); R(0, 15, 2);
    R(0, 0, 1); R(0, 0, 3); R(0, 0, 4);
    R(1, 0, 2); R(1, 1, 2); R(1, 2, 2); R(1, 4, 2); R(1, 16, 2); R(1, 20, 2); R(1, 23, 2);
    R(1, 0, 3); R(1, 0, 4);
    run("row layout, wide loads: MODE 0", cf2_k<0, 2>, A, steps, nt);
    run("row layout, wide loads: MODE 1", cf2_k<1, 2>, A, steps, nt);
    run("row layout, wide loads, fp64-per-value stats: MODE 0", cf4_k<0, 2>, A, steps, nt);
    run("row layout, wide loads, fp64-per-value stats: MODE 1", cf4_k<1, 2>, A, steps, nt);
    run("row layout, half tiles (2x waves): MODE 0", cf3_k<0, 4>, A, steps, 2 * nt);
    run("row layout, half tiles (2x waves): MODE 1", cf3_k<1, 4>, A, steps, 2 * nt);
    run("row layout, half tiles, 3 waves/SIMD regs: MODE 1", cf3_k<1, 3>, A, steps, 2 * nt);
    return 0;
}
