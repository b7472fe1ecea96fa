// vq_device.h — shared device-side definitions for the gfx950 leaf-codec kernels.
//
// Data layout in HBM ("leaf-tile" layout): leaves are grouped in tiles of LT = 32; inside a
// tile the leaf index is (almost) the fastest axis so that one wavefront row of an MFMA
// B-operand (32 leaves x 1 channel) is a contiguous 128-byte run:
//
//   act[tile][pos][C/4][32 leaves][4 channels]            (float; "L4" layout)
//
// i.e. element (leaf L, position p, channel c) lives at
//   (((L/32)*NPOS + p)*(C/4) + c/4)*128 + (L%32)*4 + c%4 .
// One lane reads/writes a float4 = 4 consecutive channels of one leaf; a 32-lane half-wave
// covers 512 contiguous bytes.  The single-channel input uses act[tile][pos][32].
// Per-leaf scalars (GroupNorm mean/rstd, channel sums) are stored [tile][k][32].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// One target: every kernel of this library assumes gfx950 (160 KB of LDS per workgroup, its fp32 MFMA shapes, v_permlane32_swap, LDS-DMA
// of 16 bytes per lane).  A multi-arch build would silently produce kernels that cannot launch; it stops here instead.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the kernels of this library are written for gfx950 (MI355X): build with --offload-arch=gfx950 only"
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VQ_LT 32  // leaves per tile

// Timing-only ablation switches (template parameter ABL of conv8_lds_k / conv_rows16_k, the STEM_DEPTH / WGRAD_ABL / ROWS4_PIPE macros) exist
// for tools/ablate/*.hip, which define VQ_ABLATE before including the kernel headers; a library build cannot instantiate them.
#ifndef VQ_ABLATE
#define VQ_ABLATE 0
#if defined(STEM_DEPTH) || defined(WGRAD_ABL) || defined(ROWS4_PIPE)
#error "STEM_DEPTH / WGRAD_ABL / ROWS4_PIPE are ablation switches: define VQ_ABLATE 1 (tools/ablate only)"
#endif
#endif

// Buffer addressing for the activation traffic of the MFMA kernels: a wave-uniform base (descriptor in four SGPRs) + a wave-uniform
// byte offset (one SGPR) + a per-lane byte offset that never changes (one VGPR).  The same access written as a per-lane 64-bit
// pointer costs a 64-bit vector add per row and an address register PAIR per request, and each such global_load/global_store
// issued into a stream of MFMAs held the matrix pipe for ~100-200 cycles (measured: the decoder's 64->64 conv went from 3.91 to
// 3.70 ms, pure MFMA time 3.55, by this change alone).  Offsets are bytes; the range is 2 GiB from the base.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t vq_buf;
__device__ __forceinline__ vq_buf buf_of(const void* uniform_base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_base, 0, 0x7fffffff, 0x00020000);
}
// ... with a range: loads beyond `bytes` return zeros (ragged last tile of a caller's buffer)
__device__ __forceinline__ vq_buf buf_of_n(const void* uniform_base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld16(vq_buf b, unsigned lane_bytes, unsigned uniform_bytes)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)lane_bytes, (int)uniform_bytes, 0));
}
__device__ __forceinline__ float buf_ld4(vq_buf b, unsigned lane_bytes, unsigned uniform_bytes)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)lane_bytes, (int)uniform_bytes, 0));
}
// Stores: the uniform offset is ADDED INTO THE VECTOR OFFSET (one 32-bit add), the scalar-offset field stays 0.  With a register in
// that field the compiler assumes the "store of more than 64 bits, then a write of its data registers" hazard does not exist and
// drops the wait state — on gfx950 it does exist: v_pk_add_f32 into the data registers straight after buffer_store_dwordx4 ...
// s5 offen corrupted one component of four lanes per row (caught by the bit-exact intermediate comparison of the encoder).
__device__ __forceinline__ void buf_st16(f32x4 v, vq_buf b, unsigned lane_bytes, unsigned uniform_bytes)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, (int)(lane_bytes + uniform_bytes), 0, 0);
}
// streaming (non-temporal) store: the output is read next by another kernel, from HBM anyway
__device__ __forceinline__ void buf_st16_nt(f32x4 v, vq_buf b, unsigned lane_bytes, unsigned uniform_bytes)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, (int)(lane_bytes + uniform_bytes), 0, 2);
}

// D = A(32x2) * B(2x32) + C, exact fp32 (k-ordered fmaf chain), 64 cycles/SIMD.
// Lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31];
// result reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// D = A(16x4) * B(4x16) + C.  Lane l supplies A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
// result reg r is D[row = 4*(l>>4) + r][col = l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// exp with a fixed operation sequence shared with the CPU oracle's restatement
// (oracle/vqvae_oracle.c vq_expf): Cephes-style polynomial, explicit fmaf only.
__device__ __forceinline__ float vq_expf(float x)
{
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float e = __builtin_fmaf(p, r2, r) + 1.0f;
    return __int_as_float(__float_as_int(e) + (((int)n) << 23));
}
__device__ __forceinline__ float vq_sigmoid(float x) { return 1.0f / (1.0f + vq_expf(-x)); }

// GroupNorm statistics accumulator (fp64; same order as the oracle's gn_stats).
// GroupNorm statistics under the 16-block contract (oracle gn_stats): the positions of a leaf form 16 equal blocks;
// inside a block one sequential fp64 chain (positions ascending, channels ascending) starting from zero, block sums added in
// block order.  add() feeds the open block, fold() closes it; s / q are the totals over the closed blocks.
struct GnAcc {
    double s, q, bs, bq;
    __device__ __forceinline__ void init() { s = 0.0; q = 0.0; bs = 0.0; bq = 0.0; }
    __device__ __forceinline__ void add(float v)
    {
        const double d = (double)v;
        bs += d;
        bq = fma(d, d, bq);
    }
    __device__ __forceinline__ void fold()
    {
        s += bs;
        q += bq;
        bs = 0.0;
        bq = 0.0;
    }
    // position-split launches: the block's sums also go to the partial buffers (gn_combine_k adds the 16 blocks in order)
    __device__ __forceinline__ void fold_store(double* ps, double* pq, size_t i)
    {
        if (ps) ps[i] = bs, pq[i] = bq;
        fold();
    }
};
// index of (tile, statistics block, accumulator slot, leaf) in the partial buffers: 16 blocks x 16 slots x 32 leaves per tile
__device__ __forceinline__ size_t part_index(int tile, int blk, int slot, int leaf) { return (((size_t)tile * 16 + blk) * 16 + slot) * 32 + leaf; }
// ... and for the one tensor with 128 half-row blocks (conv1 output of the 16-channel residual block): half row hrow = position / 4
// = (od*8 + oh)*2 + hw; 128 half rows x 8 slots x 32 leaves per tile
__device__ __forceinline__ size_t part_index_rows(int tile, int hrow, int slot, int leaf) { return (((size_t)tile * 128 + hrow) * 8 + slot) * 32 + leaf; }
// mean / rstd from total sums over n = 2^k elements (fp64, rounded to fp32 at the end)
__device__ __forceinline__ void gn_finish(double S, double Q, double inv_n, float& mean, float& rstd)
{
    const double m = S * inv_n, ex2 = Q * inv_n;
    double var = fma(-m, m, ex2);
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
}

// relu(v * a + b) on four channels of one element: the multiply-adds as two packed fmas (v_pk_fma_f32: two IEEE fmas per vector-ALU
// slot — the same roundings as four fmaf, half the issue slots on the pipe the MFMAs share); v_max_f32 has no packed form
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 gn_relu4(f32x4 v, f32x4 a, f32x4 b)
{
    const f32x2 lo = __builtin_elementwise_fma((f32x2){v.x, v.y}, (f32x2){a.x, a.y}, (f32x2){b.x, b.y});
    const f32x2 hi = __builtin_elementwise_fma((f32x2){v.z, v.w}, (f32x2){a.z, a.w}, (f32x2){b.z, b.w});
    return (f32x4){fmaxf(lo.x, 0.0f), fmaxf(lo.y, 0.0f), fmaxf(hi.x, 0.0f), fmaxf(hi.y, 0.0f)};
}

__device__ __forceinline__ double shfl_xor32_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, 32, 64);
    hi = __shfl_xor(hi, 32, 64);
    return __hiloint2double(hi, lo);
}

// Squeeze-excite gate for one leaf (ChannelAttention, python/VQVAE_v2.py:213-228):
// gate[c] = sigmoid( fc2[c][:] . relu( fc0 . mean ) ), mean[c] = csum[c]/64.
// csum: this leaf's channel sums, element c at csum[c*32].  All lanes evaluate every channel
// with wave-uniform weight addresses (scalar loads); callers pick the channels they need.
template <int C>
__device__ __forceinline__ void se_hidden(const float* __restrict__ csum, const float* __restrict__ fc0, float (&hid)[C / 4])
{
    constexpr int R = C / 4;
#pragma unroll
    for (int j = 0; j < R; ++j) hid[j] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float m = csum[c * VQ_LT] * (1.0f / 64.0f);
#pragma unroll
        for (int j = 0; j < R; ++j) hid[j] = __builtin_fmaf(fc0[j * C + c], m, hid[j]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) hid[j] = hid[j] > 0.0f ? hid[j] : 0.0f;
}
// All C gates of this lane's leaf; fc2 is addressed wave-uniformly (scalar loads), each lane
// then selects the channels of its K-slot.  Same fmaf chain per channel as the oracle.
template <int C>
__device__ __forceinline__ void se_gates(const float (&hid)[C / 4], const float* __restrict__ fc2, float (&gate)[C])
{
    constexpr int R = C / 4;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < R; ++j) a = __builtin_fmaf(fc2[c * R + j], hid[j], a);
        gate[c] = vq_sigmoid(a);
    }
}
