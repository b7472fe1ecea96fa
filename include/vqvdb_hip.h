/*
 * vqvdb_hip.h — C ABI of libvqvdb_hip.so, the MI355X (gfx950) backend for VQVDB's
 * VQ-VAE leaf codec.  No C++ / torch / ONNX types cross this boundary.
 *
 * This is what a `HipBackend final : IVQVAECodec` (sibling of the reference's
 * src/backends/torch/TorchBackend.{hpp,cpp} and src/backends/onnx/OnnxBackend_Cuda.cpp)
 * binds; the adapter lives in include/vqvdb_hip_backend.hpp, the factory hook and the
 * maintainer-side diff are in INTEGRATION.md.
 *
 * Tensor contracts (reference: src/core/IVQVAECodec.hpp:114-135, layout fixed by the
 * orchestrator's packing loops src/orchestrator/VQVAECodec.cpp:36-59,182-192):
 *   leaves  : float32 [n_leaves][512]  = TensorView shape [B,1,8,8,8]; leaf i occupies floats
 *             [i*512,(i+1)*512) in OpenVDB leaf-buffer order (offset = d*64 + h*8 + w).
 *   indices : uint8   [n_leaves][64]   = Tensor shape [B,4,4,4]; position = d*16 + h*4 + w.
 *
 * Non-finite voxels (policy): NaN / +-Inf / values whose activations overflow are not rejected — they propagate
 * through the leaf that holds them exactly as in the reference's fp32 graph, and what that leaf encodes to is unspecified (some
 * valid uint8 per position; torch.argmin of a NaN row is unspecified too).  Every OTHER leaf of the batch is unaffected bit for bit
 * (no arithmetic mixes leaves; tests/test_gpu_parity.py::test_nan_inf_poison_stays_inside_its_own_leaf), and decode, whose input is
 * indices, always returns finite voxels.  (Inside decode, should an activation overflow to Inf / NaN — possible only with weights
 * far outside any trained range — the full-chunk folded tail (tail_rows16_k) does not multiply the composite weights that are
 * structurally zero, so voxels outside the poisoned activation's reach stay finite, as in the reference's unfolded conv chain;
 * the slab kernel (VQHIP_TAIL=slab) and the oracle's skip_rows = 0 form compute 0 x Inf = NaN there.  For finite activations all
 * forms are bit-identical; tests/test_gpu_parity.py::test_large_path_kernel_variants_agree pins that.)
 *
 * Every function returns VQHIP_OK (0) or a negative status; the message is available from
 * vqhip_last_error().  Nothing throws, nothing aborts.  A codec handle owns its device
 * buffers and streams; distinct handles may be used from distinct threads concurrently,
 * one handle is used by one thread at a time and has ONE call in flight (its activation
 * workspace is shared by consecutive calls; the reference orchestrator calls encode/decode
 * serially: VQVAECodec.cpp:108-127,166-196).
 */
#ifndef VQVDB_HIP_H
#define VQVDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQHIP_OK 0
#define VQHIP_ERR_INVALID (-1)   /* bad argument / shape / dtype                      */
#define VQHIP_ERR_MODEL (-2)     /* weight pack missing, malformed or wrong shapes    */
#define VQHIP_ERR_DEVICE (-3)    /* HIP runtime error (message carries hipGetErrorString) */
#define VQHIP_ERR_NOMEM (-4)

#define VQHIP_LEAF_VOXELS 512
#define VQHIP_LATENT_VOXELS 64

typedef struct vqhip_codec vqhip_codec;

/* Replaces: TorchBackend::TorchBackend(const CodecConfig&) (TorchBackend.cpp:84-95) and
 * load_model() (TorchBackend.cpp:38-60).  Model source is a VQWPACK1 weight pack
 * (vqvdb_amd/weightpack.py): `pack_path` (CodecConfig.source = filesystem::path) or, when
 * pack_path is NULL, `pack_bytes`/`pack_size` (CodecConfig.source = EmbeddedModel).
 * device_id: HIP device ordinal.  On failure *out = NULL and the message is retrievable
 * with vqhip_last_error(NULL). */
int vqhip_create(const char* pack_path, const void* pack_bytes, size_t pack_size, int device_id, vqhip_codec** out);

void vqhip_destroy(vqhip_codec* codec);

/* Message of the last failure on this handle (codec == NULL: last failure of
 * vqhip_create on the calling thread).  Never NULL. */
const char* vqhip_last_error(const vqhip_codec* codec);

/* Replaces: getLatentShape() (IVQVAECodec.hpp:135) and the zero-leaf probe
 * initialize_latent_shape() (TorchBackend.cpp:97-119).  The shape is derived from the
 * loaded weights; out = {4,4,4}. */
int vqhip_latent_shape(const vqhip_codec* codec, int64_t out[3]);

/* Replaces: TorchBackend::encode (TorchBackend.cpp:133-164) / OnnxCudaBackend::encode_impl
 * (OnnxBackend_Cuda.cpp:83-123).  Host pointers, caller-owned; n_leaves >= 1. */
int vqhip_encode(vqhip_codec* codec, const float* leaves, int64_t n_leaves, uint8_t* indices);

/* Replaces: TorchBackend::decode (TorchBackend.cpp:166-194) / OnnxCudaBackend::decode_impl
 * (OnnxBackend_Cuda.cpp:125-165). */
int vqhip_decode(vqhip_codec* codec, const uint8_t* indices, int64_t n_leaves, float* leaves);

/* Device-resident variants (no reference counterpart: the reference backends always stage
 * through host memory).  Pointers are device pointers on the codec's device; work is
 * enqueued on `hip_stream` (a hipStream_t, NULL = the codec's own stream) and is NOT
 * synchronised on return. */
int vqhip_encode_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, uint8_t* indices_dev, void* hip_stream);
int vqhip_decode_device(vqhip_codec* codec, const uint8_t* indices_dev, int64_t n_leaves, float* leaves_dev, void* hip_stream);

/* Leaf-pointer variants (extension; no reference counterpart).  leaf_ptrs[i] points at the
 * 512 floats of leaf i (e.g. an OpenVDB LeafNode buffer, leaf.buffer().data()).  The library gathers /
 * scatters with its own host threads through pinned staging, which replaces the orchestrator's packing loop
 * and per-batch std::vector (VQVAECodec.cpp:36-59) and its unpack memcpy (VQVAECodec.cpp:182-192). */
int vqhip_encode_leaves(vqhip_codec* codec, const float* const* leaf_ptrs, int64_t n_leaves, uint8_t* indices);
int vqhip_decode_leaves(vqhip_codec* codec, const uint8_t* indices, int64_t n_leaves, float* const* leaf_ptrs);

/* ---- .vqvdb stream entry points (extension) -------------------------------------------
 * Whole-file compress / decompress with file I/O, GPU work and leaf insertion overlapped: the MI355X-native
 * replacement for the bodies of VQVAECodec::compress (src/orchestrator/VQVAECodec.cpp:78-134) and
 * VQVAECodec::decompress (:137-208), and of the record framing in VQVDB_Writer::writeBatch /
 * VQVDB_Reader::nextBatch (src/Utils/VQVDB_Reader.cpp:137-150,240-300).  File layout: .vqvdb v3, 76-byte
 * {int32 origin[3], uint8 indices[64]} records.  No OpenVDB types cross the ABI: the
 * caller creates grids/leaves in its callbacks and hands back plain float pointers. */
typedef struct vqhip_grid_info {
    const char* name;          /* NUL-terminated, valid during the callback only */
    float transform[16];       /* Mat4s as stored                                 */
    int64_t latent_shape[3];   /* {4,4,4}                                         */
    uint32_t num_embeddings;   /* header field (256)                              */
    uint64_t total_blocks;     /* leaves in this grid                             */
    int grid_index;
} vqhip_grid_info;
/* Called on the calling thread once per grid, before its leaves (create the grid, set name/transform). !=0 aborts. */
typedef int (*vqhip_grid_begin_fn)(void* user, const vqhip_grid_info* grid);
/* Called once per batch, in file order, from ONE library thread while the GPU decodes earlier batches: create
 * or look up the n leaves at origins[n][3] (e.g. tree.touchLeaf) and store where each leaf's 512 floats go
 * (leaf.buffer().data()).  The library fills them before vqhip_decompress_file returns.  !=0 aborts. */
typedef int (*vqhip_leaf_alloc_fn)(void* user, int grid_index, const int32_t* origins, int64_t n_leaves, float** leaf_ptrs);
typedef struct vqhip_stream_stats {
    int64_t leaves;
    int32_t grids;
    double wall_s;     /* whole call                                                             */
    double read_s;     /* file read/write + record (de)framing                                   */
    double alloc_s;    /* time inside leaf_alloc callbacks (reader thread)                       */
    double copy_s;     /* gather / scatter between leaf buffers and pinned staging               */
    double io_wait_s;  /* time the pipeline waited for the reader thread (0 = GPU-bound)         */
} vqhip_stream_stats;
int vqhip_decompress_file(vqhip_codec* codec, const char* path, int64_t batch_leaves, vqhip_grid_begin_fn grid_begin,
                          vqhip_leaf_alloc_fn leaf_alloc, void* user, vqhip_stream_stats* stats);
typedef struct vqhip_grid_source {
    const char* name;
    const float* transform;          /* 16 floats, NULL = identity              */
    const float* const* leaf_ptrs;   /* n_leaves pointers to 512 floats each    */
    const int32_t* origins;          /* [n_leaves][3]                           */
    int64_t n_leaves;                /* < 2^32 (file field is u32)              */
} vqhip_grid_source;
int vqhip_compress_file(vqhip_codec* codec, const char* path, const vqhip_grid_source* grids, int n_grids, int64_t batch_leaves,
                        vqhip_stream_stats* stats);

/* In-process multi-GPU front end (extension): one codec and one host thread per listed
 * device, device g of G takes the contiguous leaf range [g*ceil(n/G), min(n,(g+1)*ceil(n/G))) of every call
 * and writes to the same offsets of the caller's buffer.  Weights are replicated; there is no collective.
 * The host threads are persistent (created here, asleep between calls); each is bound to the cores of its GPU's NUMA
 * node when /sys exposes it (pinned staging is then allocated node-local), and all devices together fan out to at
 * most min(cores, 64) copy threads (override per thread with the environment variable VQHIP_COPY_THREADS).
 * (Process-per-GPU callers use one plain codec per rank instead: bench.py, vqvdb_amd/sharding.py.) */
typedef struct vqhip_multi vqhip_multi;
int vqhip_multi_create(const char* pack_path, const void* pack_bytes, size_t pack_size, const int* device_ids, int n_devices, vqhip_multi** out);
void vqhip_multi_destroy(vqhip_multi* multi);
/* What worker `index` (0 .. n_devices-1) was bound to: numa_node = -1 / cpus_bound = 0 if the topology is not exposed. */
int vqhip_multi_worker_info(const vqhip_multi* multi, int index, int* device_id, int* numa_node, int* cpus_bound);
const char* vqhip_multi_last_error(const vqhip_multi* multi);
int vqhip_multi_encode(vqhip_multi* multi, const float* leaves, int64_t n_leaves, uint8_t* indices);
int vqhip_multi_decode(vqhip_multi* multi, const uint8_t* indices, int64_t n_leaves, float* leaves);

/* Leaves processed per internal pass (default 65536).  Bounds the device workspace
 * (about 0.1 MB per leaf for inference: three shared 32 KiB-per-leaf activation regions; 0.24 MB per leaf while
 * debug mode keeps every intermediate; the full training step adds its own workspace of 0.60 MB per leaf — saved
 * activations and gradient buffers — sized for the training batch, not for the chunk).  If the device has less free memory than the chunk
 * needs (a GPU shared with a DCC application), the chunk is halved until workspace + I/O slots fit into 80 % of the free memory:
 * once, at the first host-pointer call (or vqhip_reserve) of the handle, and again at the call after a workspace allocation has
 * failed with VQHIP_ERR_NOMEM (that call itself fails; the handle stays usable).  Results never depend on the chunk size. */
int vqhip_set_chunk_leaves(vqhip_codec* codec, int64_t chunk_leaves);

/* Small passes run the position-split kernels: each layer's output rows are spread over 4-16x more workgroups (the tiniest
 * batches also split the output channels) and the GroupNorm statistics are fused as per-block partial sums (16-block rule), which cuts the
 * latency of small batches (the SOP default of 64 leaves, training batches of 2048) 10-20x with bit-identical results.
 * Default (-1): automatic, from the measured crossovers — up to 1800 tiles (57600 leaves) in both directions; one wave per
 * tile otherwise.  tiles >= 0 sets a plain threshold instead (encode `tiles`, decode 1.25x);
 * 0 disables the split path. */
int vqhip_set_small_batch_tiles(vqhip_codec* codec, int tiles);

/* Allocates up front what calls of up to n_leaves leaves (capped at the chunk size) need: the device workspace,
 * the device I/O slots and the pinned staging buffers.  Optional — every entry point allocates lazily — but it
 * moves the one-time cost (hundreds of ms for a 65536-leaf chunk: ~7 GB of HBM, 0.5 GB pinned) out of the first
 * cook, the way the reference backends pay model loading in IVQVAECodec::create. */
int vqhip_reserve(vqhip_codec* codec, int64_t n_leaves);
/* Bytes of device workspace the handle currently holds (activations + statistics; excludes weights and I/O slots), and the
 * chunk size in effect (it may have been halved to fit a shared GPU's free memory). */
int64_t vqhip_workspace_bytes(const vqhip_codec* codec);
int64_t vqhip_chunk_leaves(const vqhip_codec* codec);

/* ---- codebook training (extension; SURVEY.md §8 f-2, stage 1) ----------------------------------------------
 * The training-mode forward of VectorQuantizerEMA (python/VQVAE_v2.py:107-156) on the encoder's outputs: assign
 * with the reference's expanded distance against the LIVE codebook, accumulate encodings_sum / dw / the
 * commitment error, EMA-update cluster_size / embed_avg / embedding.  Data parallel: every rank computes the
 * statistics of its own leaves into one flat buffer, the host all-reduces that buffer (RCCL over xGMI:
 * torch.distributed, vqvdb_amd/codebook_training.py), then every rank applies the identical update — the only
 * collective on the path.  The encoder / decoder weights are not trained here.
 * stats layout (VQHIP_VQ_STATS_FLOATS floats, device memory):
 *   [0,256) encodings_sum | [256, 256+32768) dw[256][128] | [33024, 33280) sum over a code's rows of |z-e|^2 | [33280] rows */
#define VQHIP_VQ_STATS_FLOATS (256 + 256 * 128 + 256 + 1)
/* Start (or restart) training state: cluster_size[256] (NULL = ones) and embed_avg[256][128] (NULL = a copy of the
 * embedding), i.e. the reference's initial buffers (VQVAE_v2.py:103-105) or a checkpoint's. */
int vqhip_train_begin(vqhip_codec* codec, const float* cluster_size, const float* embed_avg);
/* Encoder forward + latent + assignment + local statistics for one batch (n_leaves <= chunk size) resident in HBM.
 * indices_dev ([n][64] uint8) and latent_dev ([n*64][128] float, the reference's `flat` rows) may be NULL. */
int vqhip_train_vq_stats_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, float* stats_dev, uint8_t* indices_dev,
                                float* latent_dev, void* hip_stream);
/* Validation forward (python/training.py:183-199; VQVAE.forward in eval mode, VQVAE_v2.py:344-348): the statistics of
 * vqhip_train_vq_stats_device without any update, plus the reconstruction through the decoder and
 * recon_sums_dev[3] = { sum (recon-x)^2, sum |recon-x|, voxels }.  recon_dev ([n][512]) may be NULL. */
int vqhip_train_eval_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, float* stats_dev, float* recon_sums_dev,
                            float* recon_dev, void* hip_stream);
/* EMA update from the (all-reduced) statistics: cluster_size = decay*cluster_size + (1-decay)*encodings_sum, embed_avg likewise
 * with dw, embedding = embed_avg / max(cluster_size, eps)  (VQVAE_v2.py:135-144; reference defaults decay 0.95, eps 1e-4). */
int vqhip_train_vq_update_device(vqhip_codec* codec, const float* stats_dev, float decay, float eps, void* hip_stream);
/* Host copies of the live state (any pointer may be NULL): checkpointing, dead-code reset (VQVAE_v2.py:382-417). */
int vqhip_train_get_state(vqhip_codec* codec, float* embedding, float* cluster_size, float* embed_avg);
int vqhip_train_set_state(vqhip_codec* codec, const float* embedding, const float* cluster_size, const float* embed_avg);
/* Rebuild the inference tables that derive from the codebook (folded VQ search, decoder stem table).  The inference
 * entry points do this on demand; calling it explicitly keeps the cost out of the first encode/decode. */
int vqhip_train_commit(vqhip_codec* codec);

/* ---- full training step (extension; SURVEY.md §8 f-2, stage 2; python/training.py:47-258) -------------------------
 * fp32 forward + backward + AdamW for the encoder / decoder weights and EMA for the codebook, data parallel like stage 1:
 *   vqhip_fulltrain_fwdbwd_device   this rank's batch -> flat gradient vector + auxiliary sums (local)
 *   [host: all-reduce(SUM) of both buffers over RCCL, vqvdb_amd/full_training.py]
 *   vqhip_fulltrain_apply_device    AdamW step on every rank + EMA codebook update + rebuild of all weight-derived tables
 * Loss = 0.8 mse + 0.2 l1 + vq_loss (training.py:147-155, fp32 instead of the reference's autocast), means over the GLOBAL
 * batch of n_global_leaves, so summing the ranks' gradients gives the gradient of the global loss.  Parameters live in one flat
 * vector in the reference's parameter order (model.parameters(); 995 905 floats).  Requires vqhip_fulltrain_begin (which also
 * starts the EMA state of vqhip_train_begin unless it is already running). */
#define VQHIP_FULLTRAIN_AUX_FLOATS (VQHIP_VQ_STATS_FLOATS + 3) /* VQ statistics | sum (recon-x)^2 | sum |recon-x| | voxels */
int vqhip_fulltrain_begin(vqhip_codec* codec);
int64_t vqhip_fulltrain_param_count(const vqhip_codec* codec);
/* The decoder tail (up_conv -> PixelShuffle3D -> final) is linear in its input and bilinear in its weights; by default (1) the training
 * step runs it as ONE folded operator, forward and backward, like inference does (vq_train_tail.h: fragments rebuilt on the device every
 * step, data gradient through the transposed operator, weight gradients by the chain rule through the fold).  0 selects the
 * layer-by-layer tail (every intermediate tensor and its gradient materialised: what the per-tensor gradient tests look at). */
int vqhip_fulltrain_set_folded_tail(vqhip_codec* codec, int on);
/* Training-mode forward only (test hook): every activation stays in the workspaces for vqhip_debug_fetch. */
int vqhip_fulltrain_forward_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, void* hip_stream);
/* Forward + backward of this rank's batch, enqueued on hip_stream (NULL: the codec's own).  Internally the weight / bias gradients,
 * the codebook statistics and the fold of the decoder tail run on a second stream of the codec beside the data-gradient chain; the
 * call returns with hip_stream made to wait for that stream, so work enqueued on hip_stream afterwards sees grads_dev / aux_dev
 * complete, exactly as with one stream (environment VQHIP_TRAIN_STREAMS=1 keeps everything on hip_stream; same bits either way). */
int vqhip_fulltrain_fwdbwd_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, int64_t n_global_leaves, float* grads_dev,
                                  float* aux_dev /* VQHIP_FULLTRAIN_AUX_FLOATS or NULL */, void* hip_stream);
/* The same, with a hook between the two halves of the backward pass: decoder_done(user) runs on the calling thread right after the
 * decoder's backward kernels have been enqueued (nothing is synchronised; non-zero return aborts).  From that point of the stream on,
 * grads_dev[vqhip_fulltrain_decoder_offset() ..] is final: a data-parallel caller records an event and all-reduces that slice on
 * another stream while the encoder half of the backward pass runs (vqvdb_amd/full_training.py).  WHICH stream to record that event on:
 * vqhip_fulltrain_ready_stream() — the codec's second stream when the weight gradients run there (it has been ordered after the
 * decoder's kernels on hip_stream at that point, without making hip_stream wait for it), NULL = hip_stream itself. */
typedef int (*vqhip_phase_fn)(void* user);
int vqhip_fulltrain_fwdbwd_overlap_device(vqhip_codec* codec, const float* leaves_dev, int64_t n_leaves, int64_t n_global_leaves, float* grads_dev,
                                          float* aux_dev, void* hip_stream, vqhip_phase_fn decoder_done, void* user);
int64_t vqhip_fulltrain_decoder_offset(const vqhip_codec* codec);
void* vqhip_fulltrain_ready_stream(const vqhip_codec* codec);
/* step counts from 1 (bias correction).  Reference hyper-parameters: lr 1e-4 (cosine schedule on the host), betas 0.9 / 0.999,
 * eps 1e-8, weight_decay 1e-4 (training.py:104-108); EMA decay 0.95, eps 1e-4.  aux_dev NULL leaves the codebook untouched. */
int vqhip_fulltrain_apply_device(vqhip_codec* codec, const float* grads_dev, const float* aux_dev, float lr, int64_t step, float beta1, float beta2,
                                 float adam_eps, float weight_decay, float ema_decay, float ema_eps, void* hip_stream);
/* host copies of the flat parameter vector (checkpoints, export to a weight pack); set also rebuilds the device tables */
int vqhip_fulltrain_get_params(vqhip_codec* codec, float* params);
int vqhip_fulltrain_set_params(vqhip_codec* codec, const float* params);
/* AdamW moments (param_count floats each, flat parameter order) so that a checkpoint resumes the optimizer like the reference's
 * does (python/training.py:216-226 saves optimizer and scheduler state); the step count is the caller's (apply_device's `step`). */
int vqhip_fulltrain_get_opt_state(vqhip_codec* codec, float* exp_avg, float* exp_avg_sq);
int vqhip_fulltrain_set_opt_state(vqhip_codec* codec, const float* exp_avg, const float* exp_avg_sq);

/* ---- measurement hooks (bench.py / tests) ---- */

/* Per-kernel timing with HIP events on the launch stream.  While enabled, every kernel
 * launch is bracketed by events; vqhip_profile_read() synchronises and returns, for up to
 * `cap` kernels, name / launch count / total milliseconds, and resets the counters. */
typedef struct vqhip_kernel_stat {
    char name[48];
    int64_t launches;
    double total_ms;
    double flops_per_leaf;     /* nominal dense FLOPs per leaf of the reference ops this kernel replaces (0: data movement / table lookups) */
    double eff_flops_per_leaf; /* FLOPs per leaf the kernel really issues (padding taps skipped, folded operators at folded cost) */
    int64_t leaves;            /* leaves processed over the counted launches */
} vqhip_kernel_stat;
int vqhip_profile_enable(vqhip_codec* codec, int enable);
int vqhip_profile_read(vqhip_codec* codec, vqhip_kernel_stat* stats, int cap, int* count);

/* Test hooks.  vqhip_debug_enable(1) makes encode also store the first conv's raw output (it is
 * otherwise recomputed, never written).  vqhip_debug_fetch copies an intermediate activation
 * of the LAST encode/decode chunk to host as float32 [n_leaves][C][positions] (NCDHW
 * flattened).  Names: e_y1 e_a1 e_y4 e_a6 e_x7 e_y9 e_x11 d_ystem d_d2 d_y4 d_x6. */
int vqhip_debug_enable(vqhip_codec* codec, int enable);
int vqhip_debug_fetch(vqhip_codec* codec, const char* name, int64_t n_leaves, float* out);

/* Hardware assumption check: runs fp32 MFMA chains against per-lane fmaf chains in K order
 * on the device; mismatches[0] = differing elements for 32x32x2, mismatches[1] for 16x16x4. */
int vqhip_selftest_mfma(vqhip_codec* codec, int64_t mismatches[2]);

const char* vqhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VQVDB_HIP_H */
