// vq_train_kernels.h — codebook-training kernels (stage 1): the training-mode forward of
// VectorQuantizerEMA (python/VQVAE_v2.py:107-156) on encoder outputs, for gfx950.
//
//   latent_assign_k   z = proj(gate * x11) + b materialised (flat [row][128], row = leaf*64 + pos, the reference's
//                     `flat` view, :113-114) and assigned with the reference's expanded distance against the LIVE
//                     codebook (:117-124): the arithmetic of the oracle's "faithful" path.
//   vq_ema_partials_k / vq_ema_reduce_k
//                     encodings_sum, dw = encodings^T @ flat, sum (z - e)^2   (:134-137,146) without one-hots: one wave
//                     per (code, 8192-row segment) scans the indices and adds up its member rows in ascending order,
//                     segments are added ascending (deterministic, no atomics).
//   vq_ema_update_k   cluster_size / embed_avg EMA and embedding = embed_avg / clamp(cluster_size, eps) (:135-144)
//   codebook_frag_k   live codebook -> MFMA A-fragment order + code norms for latent_assign_k
#pragma once
#include "vq_device.h"

#define VQ_STATS_COUNTS 0
#define VQ_STATS_DW 256
#define VQ_STATS_SQ (256 + 256 * 128)
#define VQ_STATS_ROWS (256 + 256 * 128 + 256)
#define VQ_STATS_FLOATS (256 + 256 * 128 + 256 + 1)

// efrag[(u*8+ct)*64+lane][i] = E[32ct + (lane&31)][8u + 4(lane>>5) + i];  ee_frag[(ct*2+q)*16+r] = ||E[32ct+(r&3)+8(r>>2)+4q]||^2
__global__ __launch_bounds__(256) void codebook_frag_k(const float* __restrict__ E, float* __restrict__ efrag, float* __restrict__ ee_frag)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 16 * 8 * 64) {
        const int u = t / (8 * 64), ct = (t / 64) % 8, lane = t % 64;
        const float* src = E + (32 * ct + (lane & 31)) * 128 + 8 * u + 4 * (lane >> 5);
        ((f32x4*)efrag)[t] = (f32x4){src[0], src[1], src[2], src[3]};
    } else if (t < 16 * 8 * 64 + 256) {
        const int f = t - 16 * 8 * 64;
        const int ct = f / 32, q = (f / 16) % 2, r = f % 16;
        const float* e = E + (32 * ct + (r & 3) + 8 * (r >> 2) + 4 * q) * 128;
        float s = 0.0f;
        for (int c = 0; c < 128; ++c) s = __builtin_fmaf(e[c], e[c], s);
        ee_frag[f] = s;
    }
}

struct LatentArgs {
    const float* in;        // x11 L4 [tile][64][8][32][4]
    const float* se_csum;   // [tile][32][32]
    const float* se_fc0;    // [8][32]
    const float* se_fc2;    // [32][8]
    const float* wproj;     // frag [u=4][mt=4][64][4]
    const float* bproj;     // D-fragment order [(mt*2+q)*16 + r]
    const float* efrag;     // frag [u=16][ct=8][64][4]
    const float* ee_frag;   // [(ct*2+q)*16 + r]
    uint8_t* idx;           // [n_leaves][64]
    float* z;               // flat [n_leaves*64][128]
    float* z4;              // optional: the same latent in the L4 tile layout [tile][64][32][32][4] (full training step)
    int64_t n_leaves;
    int n_tiles;
};

// One wave per 32-leaf tile.  Per position: z (128 ch) by 64 MFMAs with the gated activations as B operand; z is then
// directly the B operand of the distance GEMM (D-layout rows (r&3)+8(r>>2)+4q give the contract's "P8" k-order).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void latent_assign_k(LatentArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* ldsE = (f32x4*)smem_raw;          // 16*8*64 float4 = 128 KB
    f32x4* ldsP = ldsE + 16 * 8 * 64;        // 4*4*64 float4  = 16 KB
    for (int i = threadIdx.x; i < 16 * 8 * 64; i += NW * 64) ldsE[i] = ((const f32x4*)A.efrag)[i];
    for (int i = threadIdx.x; i < 4 * 4 * 64; i += NW * 64) ldsP[i] = ((const f32x4*)A.wproj)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * NW + wave;
    if (tile >= A.n_tiles) return;
    const int j = lane & 31, q = lane >> 5;
    float gate[4][4];
    {
        float hid[8], gall[32];
        se_hidden<32>(A.se_csum + (size_t)tile * 32 * 32 + j, A.se_fc0, hid);
        se_gates<32>(hid, A.se_fc2, gall);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) gate[u][i] = q ? gall[8 * u + 4 + i] : gall[8 * u + i];
    }
    const f32x4* in4 = (const f32x4*)A.in + (size_t)tile * 64 * 8 * 32 + q * 32 + j;
    const f32x4* bp4 = (const f32x4*)A.bproj;
    const f32x4* ee4 = (const f32x4*)A.ee_frag;
    const int64_t leaf = (int64_t)tile * 32 + j;
    const bool live = leaf < A.n_leaves;
    int p0 = 0, p1 = 64;
    if (gridDim.y > 1) p0 = (int)(blockIdx.y * 64 / gridDim.y), p1 = (int)((blockIdx.y + 1) * 64 / gridDim.y);
    for (int p = p0; p < p1; ++p) {
        f32x16 z[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[t][r] = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 b = in4[((size_t)p * 8 + 2 * u) * 32];
            b.x = b.x * gate[u][0];
            b.y = b.y * gate[u][1];
            b.z = b.z * gate[u][2];
            b.w = b.w * gate[u][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 w = ldsP[(u * 4 + t) * 64 + lane];
                z[t] = mfma32(w.x, b.x, z[t]);
                z[t] = mfma32(w.y, b.y, z[t]);
                z[t] = mfma32(w.z, b.z, z[t]);
                z[t] = mfma32(w.w, b.w, z[t]);
            }
        }
        float zzp = 0.0f;
        f32x4* zrow = (f32x4*)(A.z + ((size_t)leaf * 64 + p) * 128);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = bp4[(t * 2 + q) * 4 + g];
                z[t][4 * g + 0] = z[t][4 * g + 0] + bias.x;
                z[t][4 * g + 1] = z[t][4 * g + 1] + bias.y;
                z[t][4 * g + 2] = z[t][4 * g + 2] + bias.z;
                z[t][4 * g + 3] = z[t][4 * g + 3] + bias.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) zzp = __builtin_fmaf(z[t][4 * g + i], z[t][4 * g + i], zzp);
                {  // channels 32t + 8g + 4q .. +3 of this leaf's row
                    f32x4 v;
                    v.x = z[t][4 * g + 0], v.y = z[t][4 * g + 1], v.z = z[t][4 * g + 2], v.w = z[t][4 * g + 3];
                    if (live) zrow[8 * t + 2 * g + q] = v;
                    if (A.z4) ((f32x4*)A.z4)[(((size_t)tile * 64 + p) * 32 + 8 * t + 2 * g + q) * 32 + j] = v;
                }
            }
        const float zzo = __shfl_xor(zzp, 32, 64);
        const float zz = q == 0 ? zzp + zzo : zzo + zzp;  // partial(c&4==0) + partial(c&4!=0)
        float best = __builtin_inff();
        int bk = 0;
        for (int ct = 0; ct < 8; ++ct) {
            f32x4 eev[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) eev[g] = ee4[(ct * 2 + q) * 4 + g];
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 e = ldsE[((4 * t + g) * 8 + ct) * 64 + lane];
                    d = mfma32(e.x, z[t][4 * g + 0], d);
                    d = mfma32(e.y, z[t][4 * g + 1], d);
                    d = mfma32(e.z, z[t][4 * g + 2], d);
                    d = mfma32(e.w, z[t][4 * g + 3], d);
                    if (g == 3) __builtin_amdgcn_sched_barrier(0);  // keep LDS reads from piling up (VGPR budget)
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ee = eev[g];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float een = i == 0 ? ee.x : (i == 1 ? ee.y : (i == 2 ? ee.z : ee.w));
                    const float t1 = zz + een;
                    const float dist = t1 - 2.0f * d[4 * g + i];
                    const int k = 32 * ct + i + 8 * g + 4 * q;
                    if (dist < best) {
                        best = dist;
                        bk = k;
                    }
                }
            }
        }
        const float ob = __shfl_xor(best, 32, 64);
        const int ok = __shfl_xor(bk, 32, 64);
        if (ob < best || (ob == best && ok < bk)) bk = ok;
        if (q == 0 && live) A.idx[leaf * 64 + p] = (uint8_t)bk;
    }
}

// Statistics in two deterministic passes.  Rows are cut into segments of VQ_SEG_ROWS consecutive rows (the last may be
// short).  Pass 1: one wave per (code k, segment): lane l owns channels 2l, 2l+1 and adds up the segment's member rows
// in ascending order (fp32), with the squared distance to e_k as an fmaf chain; partials go to scratch.  Pass 2: one
// workgroup per code adds the segment partials in ascending order (dw in fp32 from 0, the error in fp64).
// Output (local to this rank, summed across ranks by the host's all-reduce):
//   stats[COUNTS+k], stats[DW + k*128 + c], stats[SQ+k] = sum over the code's rows and channels of (z-e)^2
#define VQ_SEG_ROWS 8192

__global__ __launch_bounds__(1024) void vq_ema_partials_k(const float* __restrict__ z, const uint8_t* __restrict__ idx, const float* __restrict__ E,
                                                           int64_t n_rows, int n_seg, float* __restrict__ part, double* __restrict__ sqpart,
                                                           int* __restrict__ cntpart)
{
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.y * 16 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int64_t r0 = (int64_t)seg * VQ_SEG_ROWS, r1 = r0 + VQ_SEG_ROWS < n_rows ? r0 + VQ_SEG_ROWS : n_rows;
    const float e0 = E[k * 128 + 2 * lane], e1 = E[k * 128 + 2 * lane + 1];
    float a0 = 0.0f, a1 = 0.0f, sq = 0.0f;
    int cnt = 0;
    for (int64_t r = r0; r < r1; r += 64) {
        const int64_t rr = r + lane;
        const bool hit = rr < r1 && idx[rr] == (uint8_t)k;
        unsigned long long m = __ballot(hit);
        cnt += __popcll(m);
        while (m) {
            // up to four member rows in flight; accumulation stays in ascending row order
            int b[4];
            float2 v[4];
            int nb = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b[i] = m ? __builtin_ctzll(m) : -1;
                if (m) {
                    m &= m - 1;
                    ++nb;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nb) v[i] = *(const float2*)(z + (size_t)(r + b[i]) * 128 + 2 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nb) {
                    a0 = a0 + v[i].x;
                    a1 = a1 + v[i].y;
                    const float d0 = v[i].x - e0, d1 = v[i].y - e1;
                    sq = __builtin_fmaf(d0, d0, sq);
                    sq = __builtin_fmaf(d1, d1, sq);
                }
        }
    }
    float2* dst = (float2*)(part + ((size_t)k * n_seg + seg) * 128);
    dst[lane] = make_float2(a0, a1);
    // per-segment error: lanes ascending, fp64
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += (double)__shfl(sq, l, 64);
    if (lane == 0) {
        sqpart[(size_t)k * n_seg + seg] = s;
        cntpart[(size_t)k * n_seg + seg] = cnt;
    }
}

// Round 5: the same partial sums without every (code, segment) wave walking the segment's 8192 index bytes (256 x the index traffic,
// and a dependent ballot -> gather chain per 64 rows: 0.19 ms at 2048 leaves, a quarter of the codebook-training step).
//   vq_ema_lists_k   one wave per segment: a STABLE counting sort of the segment's rows by code (LDS histogram, exclusive scan, scatter
//                    with each lane's rank among the equal codes of its 64-row chunk from eight ballots) -> per code the member rows of
//                    the segment in ascending order (16-bit offsets inside the segment) + start / count per (segment, code)
//   vq_ema_gather_k  one wave per (code, segment) as before, but its member rows are known up front: up to eight rows in flight, added in
//                    ascending order — the arithmetic of vq_ema_partials_k, bit for bit (VQHIP_TRAIN_EMA=scan keeps that kernel)
__global__ __launch_bounds__(64) void vq_ema_lists_k(const uint8_t* __restrict__ idx, int64_t n_rows, unsigned short* __restrict__ lists /*[n_seg][VQ_SEG_ROWS]*/,
                                                     int* __restrict__ starts /*[n_seg][256][2]: start, count*/)
{
    __shared__ int cnt[256], off[256];
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int64_t r0 = (int64_t)seg * VQ_SEG_ROWS;
    const int nr = (int)(r0 + VQ_SEG_ROWS < n_rows ? VQ_SEG_ROWS : n_rows - r0);
    for (int i = lane; i < 256; i += 64) cnt[i] = 0;
    __syncthreads();
    for (int r = lane; r < nr; r += 64) atomicAdd(&cnt[idx[r0 + r]], 1);
    __syncthreads();
    {   // exclusive scan over the 256 codes: four codes per lane, then across the lanes
        const int c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
        const int own = c0 + c1 + c2 + c3;
        int incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int base = incl - own;
        off[4 * lane] = base, off[4 * lane + 1] = base + c0, off[4 * lane + 2] = base + c0 + c1, off[4 * lane + 3] = base + c0 + c1 + c2;
        int* st = starts + ((size_t)seg * 256 + 4 * lane) * 2;
        st[0] = base, st[1] = c0, st[2] = base + c0, st[3] = c1, st[4] = base + c0 + c1, st[5] = c2, st[6] = base + c0 + c1 + c2, st[7] = c3;
    }
    __syncthreads();
    unsigned short* dst = lists + (size_t)seg * VQ_SEG_ROWS;
    for (int rb = 0; rb < nr; rb += 64) {
        const int r = rb + lane;
        const bool live = r < nr;
        const int code = live ? (int)idx[r0 + r] : 0;
        // lanes of this chunk with the same code (and live): eight ballots
        unsigned long long m = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((code >> b) & 1);
            m &= ((code >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        const int base = off[code];                       // (every lane of the group reads it before the group's first lane advances it)
        if (live) dst[base + rank] = (unsigned short)r;
        if (live && rank == 0) off[code] = base + __popcll(m);
        __syncthreads();                                  // (one wave: orders the LDS update before the next chunk's reads)
    }
}

__global__ __launch_bounds__(1024) void vq_ema_gather_k(const float* __restrict__ z, const unsigned short* __restrict__ lists, const int* __restrict__ starts,
                                                         const float* __restrict__ E, int64_t n_rows, int n_seg, float* __restrict__ part,
                                                         double* __restrict__ sqpart, int* __restrict__ cntpart)
{
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.y * 16 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int64_t r0 = (int64_t)seg * VQ_SEG_ROWS;
    const int start = starts[((size_t)seg * 256 + k) * 2], cnt = starts[((size_t)seg * 256 + k) * 2 + 1];
    const unsigned short* lst = lists + (size_t)seg * VQ_SEG_ROWS + start;
    const float e0 = E[k * 128 + 2 * lane], e1 = E[k * 128 + 2 * lane + 1];
    float a0 = 0.0f, a1 = 0.0f, sq = 0.0f;
    for (int b0 = 0; b0 < cnt; b0 += 64) {
        const int nb = cnt - b0 < 64 ? cnt - b0 : 64;
        const int mine = lane < nb ? (int)lst[b0 + lane] : 0;   // this batch's member rows, one per lane
        for (int j0 = 0; j0 < nb; j0 += 8) {
            float2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = __builtin_amdgcn_readlane(mine, (j0 + i) & 63);
                if (j0 + i < nb) v[i] = *(const float2*)(z + (size_t)(r0 + r) * 128 + 2 * lane);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (j0 + i < nb) {
                    a0 = a0 + v[i].x;
                    a1 = a1 + v[i].y;
                    const float d0 = v[i].x - e0, d1 = v[i].y - e1;
                    sq = __builtin_fmaf(d0, d0, sq);
                    sq = __builtin_fmaf(d1, d1, sq);
                }
        }
    }
    float2* dst = (float2*)(part + ((size_t)k * n_seg + seg) * 128);
    dst[lane] = make_float2(a0, a1);
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += (double)__shfl(sq, l, 64);
    if (lane == 0) {
        sqpart[(size_t)k * n_seg + seg] = s;
        cntpart[(size_t)k * n_seg + seg] = cnt;
    }
}

__global__ __launch_bounds__(128) void vq_ema_reduce_k(const float* __restrict__ part, const double* __restrict__ sqpart, const int* __restrict__ cntpart,
                                                        int64_t n_rows, int n_seg, float* __restrict__ stats)
{
    const int k = blockIdx.x, c = threadIdx.x;
    float s = 0.0f;
    for (int g = 0; g < n_seg; ++g) s = s + part[((size_t)k * n_seg + g) * 128 + c];
    stats[VQ_STATS_DW + k * 128 + c] = s;
    if (c == 0) {
        double q = 0.0;
        int n = 0;
        for (int g = 0; g < n_seg; ++g) {
            q += sqpart[(size_t)k * n_seg + g];
            n += cntpart[(size_t)k * n_seg + g];
        }
        stats[VQ_STATS_SQ + k] = (float)q;
        stats[VQ_STATS_COUNTS + k] = (float)n;
        if (k == 0) stats[VQ_STATS_ROWS] = (float)n_rows;
    }
}

// EMA update from the (all-reduced) statistics; one workgroup per code, one thread per channel.
__global__ __launch_bounds__(128) void vq_ema_update_k(const float* __restrict__ stats, float decay, float alpha, float eps,
                                                        float* __restrict__ cluster_size, float* __restrict__ embed_avg, float* __restrict__ embedding)
{
    const int k = blockIdx.x, c = threadIdx.x;
    const float cs = __builtin_fmaf(alpha, stats[VQ_STATS_COUNTS + k], cluster_size[k] * decay);
    const float avg = __builtin_fmaf(alpha, stats[VQ_STATS_DW + k * 128 + c], embed_avg[k * 128 + c] * decay);
    embed_avg[k * 128 + c] = avg;
    embedding[k * 128 + c] = avg / (cs < eps ? eps : cs);
    __syncthreads();  // every thread has read cluster_size[k]
    if (c == 0) cluster_size[k] = cs;
}

// Validation losses of the reference loop (python/training.py:183-199: F.mse_loss / F.l1_loss of the reconstruction):
// sums of (y-x)^2 and |y-x| over all voxels.  Fixed launch geometry (RL_BLOCKS x 256 threads, grid-stride), fp64 partials,
// block tree in LDS, then one thread adds the block partials in order -> deterministic.
#define RL_BLOCKS 1024
__global__ __launch_bounds__(256) void recon_loss_partials_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n_elems,
                                                              double* __restrict__ part /*[RL_BLOCKS][2]*/)
{
    __shared__ double s2[256], s1[256];
    double a2 = 0.0, a1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)RL_BLOCKS * 256) {
        const double d = (double)y[i] - (double)x[i];
        a2 = fma(d, d, a2);
        a1 += d < 0.0 ? -d : d;
    }
    s2[threadIdx.x] = a2;
    s1[threadIdx.x] = a1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s2[threadIdx.x] += s2[threadIdx.x + w];
            s1[threadIdx.x] += s1[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s2[0];
        part[2 * blockIdx.x + 1] = s1[0];
    }
}
__global__ void recon_loss_reduce_k(const double* __restrict__ part, int64_t n_elems, float* __restrict__ out /*[3]: sum sq, sum abs, elems*/)
{
    double a2 = 0.0, a1 = 0.0;
    for (int b = 0; b < RL_BLOCKS; ++b) {
        a2 += part[2 * b];
        a1 += part[2 * b + 1];
    }
    out[0] = (float)a2;
    out[1] = (float)a1;
    out[2] = (float)n_elems;
}
