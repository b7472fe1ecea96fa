// vq_runtime.hip — host runtime and C ABI of libvqvdb_hip.so (include/vqvdb_hip.h).
//
// One vqhip_codec = one device context: fragment-ordered weights resident in HBM, a
// workspace of leaf-tile activations sized for one chunk of leaves, a compute stream and
// pinned staging buffers for the host-pointer entry points.  No PyTorch / ONNX / CPU
// fallback: if HIP is unavailable every entry point fails with VQHIP_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <memory>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vqvdb_hip.h"
#include "vq_kernels.h"
#include "vq_conv8_lds.h"
#include "vq_stem_taps.h"
#include "vq_conv4_lds.h"
#include "vq_first_roll.h"
#include "vq_convdown_lds.h"
#include "vq_train_kernels.h"
#include "vq_grad_kernels.h"
#include "vq_train_tail.h"
#include "vq_tail_rows.h"
#include "vq_tail_groups.h"

namespace {

thread_local std::string g_create_error = "";

struct PackTensor {
    std::vector<uint32_t> dims;
    const float* data = nullptr;
    size_t count = 0;
};

// Name -> value table with a cheap lookup: every launch resolves its ~45 buffers by name (a["e_a1"], w["r16c1.w"]); as
// std::map<std::string, ...> that was a heap of string constructions and tree walks per pass — measurable in the 0.14-0.2 ms
// calls of SOP-sized batches.  Open addressing over an FNV-1a hash of the characters, no allocation on lookup; entries keep
// their insertion order and their addresses' CONTENT stays put (values are only ever added), iteration yields .first / .second.
template <typename T>
struct NameMap {
    struct Ent {
        std::string first;
        T second;
    };
    NameMap() = default;
    NameMap(const NameMap&) = delete;              // (a copy would have capacity == size: the next insertion would look like an overflow)
    NameMap& operator=(const NameMap&) = delete;
    static constexpr size_t kMaxNames = 1024;   // (the largest table, the device-buffer names of a training handle, holds ~250)
    std::vector<Ent> ents;
    std::vector<int> slots;   // index into ents, -1 = empty; size is a power of two > 2 * ents.size()
    static uint32_t hash(const char* s)
    {
        uint32_t h = 2166136261u;
        for (; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
        return h;
    }
    int lookup(const char* s) const
    {
        if (slots.empty()) return -1;
        const uint32_t mask = (uint32_t)slots.size() - 1;
        for (uint32_t i = hash(s) & mask;; i = (i + 1) & mask) {
            const int e = slots[i];
            if (e < 0) return -1;
            if (ents[e].first == s) return e;
        }
    }
    void rehash()
    {
        size_t n = 64;
        while (n < 4 * ents.size()) n *= 2;
        slots.assign(n, -1);
        for (size_t e = 0; e < ents.size(); ++e) {
            uint32_t i = hash(ents[e].first.c_str()) & (uint32_t)(n - 1);
            while (slots[i] >= 0) i = (i + 1) & (uint32_t)(n - 1);
            slots[i] = (int)e;
        }
    }
    T& operator[](const char* s)
    {
        int e = lookup(s);
        if (e < 0) {
            // references handed out by operator[] must stay valid across later insertions (callers hold `auto& w = c->dw` and lambdas capture
            // it): the entries live in storage reserved once; a table that outgrows it is a programming error and stops loudly
            if (ents.capacity() == 0) ents.reserve(kMaxNames);
            if (ents.size() == ents.capacity()) {
                fprintf(stderr, "vqvdb_hip: NameMap overflow (%zu names) at '%s'\n", ents.size(), s);
                abort();
            }
            ents.push_back(Ent{s, T()});
            e = (int)ents.size() - 1;
            if (slots.size() < 4 * ents.size()) rehash();
            else {
                uint32_t i = hash(s) & (uint32_t)(slots.size() - 1);
                while (slots[i] >= 0) i = (i + 1) & (uint32_t)(slots.size() - 1);
                slots[i] = e;
            }
        }
        return ents[e].second;
    }
    T& operator[](const std::string& s) { return (*this)[s.c_str()]; }
    const Ent* find(const char* s) const { const int e = lookup(s); return e < 0 ? end() : &ents[e]; }
    const Ent* find(const std::string& s) const { return find(s.c_str()); }
    Ent* find(const char* s) { const int e = lookup(s); return e < 0 ? end() : &ents[e]; }
    Ent* find(const std::string& s) { return find(s.c_str()); }
    size_t count(const char* s) const { return lookup(s) >= 0; }
    size_t count(const std::string& s) const { return lookup(s.c_str()) >= 0; }
    Ent* begin() { return ents.data(); }
    Ent* end() { return ents.data() + ents.size(); }
    const Ent* begin() const { return ents.data(); }
    const Ent* end() const { return ents.data() + ents.size(); }
};

struct KernelTimer {
    std::string name;
    hipEvent_t start, stop;
    int64_t leaves;
};

struct KernelInfo {
    double flops, eff_flops;
};

// FLOPs per leaf of each compute kernel: {nominal, issued}.  nominal = dense count of the reference ops the
// kernel replaces (zero-padding taps included); issued = what the kernel really executes
// on the matrix pipe (padding taps skipped, folded/looked-up operators counted at their folded cost) — the
// honest utilisation numerator.
const std::map<std::string, KernelInfo>& kernel_info()
{
    static const std::map<std::string, KernelInfo> m = {
        {"enc_conv_first_stats", {0.0, 2.0 * 221184 * 0.7703}},   // recomputation: the op is counted once, on the second pass
        {"enc_conv_first_gn", {2.0 * 221184, 2.0 * 221184 * 0.7703}},
        {"enc_res16_conv1", {2.0 * 3538944, 2.0 * 3538944 * 0.7703}},
        {"enc_res16_conv2", {2.0 * 3538944, 2.0 * 3538944 * 0.7703}},
        {"enc_down", {2.0 * 2097152, 2.0 * 2097152 * 0.669922}},
        {"enc_res32_conv1", {2.0 * 1769472, 2.0 * 1769472 * 0.578704}},
        {"enc_res32_conv2", {2.0 * 1769472, 2.0 * 1769472 * 0.578704}},
        {"enc_vq", {2.0 * (262144 + 2097152 + 512), 2.0 * (524288 + 512)}},  // attn + proj + VQ; projection folded into the search
        {"dec_stem", {0.0, 0.0}},
        {"dec_stem_gn", {0.0, 0.0}},  // table lookups: the 14.2 M MAC/leaf of the reference op are not executed (stem_lut_k)
        {"dec_res64_conv1", {2.0 * 7077888, 2.0 * 7077888 * 0.578704}},
        {"dec_res64_conv2", {2.0 * 7077888, 2.0 * 7077888 * 0.578704}},
        // folded up_conv+pixshuf+final: nominal = the reference ops' dense count; "effective" = the MACs the
        // folded map really needs (884 736/leaf); the kernel issues (160 steps x 4 + 64 steps x 2 cout tiles) x 32 voxels x 64
        // channels = 1 572 864 MAC/leaf (round 3: the MFMAs of a voxel plane against input planes it cannot depend on are skipped;
        // 224 steps x 128 x 64 = 1 835 008 before)
        {"dec_tail_slab", {2.0 * (28311552 + 2048 + 442368), 2.0 * 1572864}},
        // round 5 (tail_rows16_k): 16-voxel tiles of two (od,oh) cells with one reach box: the structural zeros are skipped along D and H,
        // 296 (tile, input row) pairs x 64 MFMAs of 1024 MACs per 16 leaves = 1 212 416 MAC/leaf (exact D x H: 1 179 648; useful 884 736)
        {"dec_tail", {2.0 * (28311552 + 2048 + 442368), 2.0 * 1212416}},
        {"dec_tail_rows32", {2.0 * (28311552 + 2048 + 442368), 2.0 * 1212416}},   // the same tiles, a whole 32-leaf tile per wave
        {"dec_tail_groups", {2.0 * (28311552 + 2048 + 442368), 2.0 * 1212416}},   // the same (tile, input row) pairs in three plane groups
    };
    return m;
}

}  // namespace

struct vqhip_codec {
    int device = 0;
    std::string err;
    hipStream_t stream = nullptr;
    int64_t chunk = 65536;
    int n_cus = 256;         // compute units of the device (persistent-workgroup launches)
    bool stem_fused = true;  // decoder front of large passes: one kernel; VQHIP_STEM=split selects stem_lut_k + gn_relu_stats_k
    bool stem_taps = true;   // ... the (tap, code) table streamed through an LDS ring tap by tap (stem_taps_k, vq_stem_taps.h: 0.81 -> 0.57 ms); VQHIP_STEM=gather selects stem_fused_k (gather through the L1)
    bool tail_groups = false; // ... VQHIP_TAIL=groups: the output planes in three groups instead of five units, every input plane read 2.5x instead of 3.5x (tail_groups16_k, vq_tail_groups.h; measured equal in time, 33 spilled registers)
    bool tail_rows32 = false; // ... VQHIP_TAIL=rows32: a whole 32-leaf tile per wave, one wave per SIMD (tail_rows32_k; measured 4 % slower)
    bool tail_rows = true;   // folded decoder tail of full chunks: 16-voxel tiles, zeros skipped along D and H (tail_rows16_k, vq_tail_rows.h); VQHIP_TAIL=slab selects conv_mfma32_k<OUTMODE 2> (depth only)
    int host_split = 8;      // host-memory calls of one chunk are cut into up to host_split pieces of >= host_split_min leaves (VQHIP_HOST_SPLIT=n[,min]; 1 = off)
    int64_t host_split_min = 8192;
    int tail16_tiles = 48;   // small-batch folded tail on the 16x16x4 MFMA up to this many tiles (VQHIP_TAIL16_TILES; measured: 1024 leaves 90 -> 56 us, 2048 leaves 92 -> 103 us)
    bool r64s_resident = true;   // small-batch 64->64 convs: their quarter of the weights LDS-resident (VQHIP_R64S=stream: streamed)
    int vq_split = 2;        // position ranges per tile in the VQ search of full chunks (VQHIP_VQ_SPLIT)
    bool conv8_w16 = true;   // ... with 16 waves (one half row each; a half row is a statistics block); VQHIP_CONV8=w8 selects 8 (one row each)
    bool convdown_lds = true;   // down conv of large passes: input planes streamed through LDS once, weights from L1 / L2 (vq_convdown_lds.h); VQHIP_DOWN=rows selects the row kernel (input re-fetched 3.06x)
    bool conv4_lds = true;   // 32-channel 4^3 convs of large passes: input planes in an LDS ring, weights straight from L1 / L2 (vq_conv4_lds.h); VQHIP_CONV4=rows selects the row kernel (weights LDS-resident, every input row re-fetched 6.25x)
    bool first_roll_stats = false;   // ... for the statistics pass too (VQHIP_FIRST=roll0; measured slower)
    bool first_raw = false;  // VQHIP_FIRST_SRC=raw: the first conv reads the caller's float[leaf][512] layout itself and pack_leaves_k is not launched (round 6; measured SLOWER: 0.425 + 0.465 ms against 0.069 + 0.322 + 0.396 ms with the row layout — 16 cache lines per load instead of 6)
    const float* cur_leaves = nullptr;   // the current pass's input leaves (device), for the split path's first conv
    bool first_roll = true;  // first conv of large passes, normalising pass: rolling row window in registers (vq_first_roll.h); VQHIP_FIRST=steps selects conv_first_k ((row, kd) steps, nine row loads per output row)
    bool conv8_lds = true;   // 16-channel 8^3 convs of large passes: LDS-plane kernel (vq_conv8_lds.h); VQHIP_CONV8=rows selects the row-group kernel
    int split_tiles = -1;    // position-split path: -1 = automatic (measured crossovers, use_split), >= 0 = plain tile threshold

    // device weights
    NameMap<float*> dw;
    NameMap<int> nsteps;
    float e_final_bias = 0.0f;

    // workspace
    int64_t ws_tiles = 0;
    bool ws_full = false;    // workspace layout: full (every activation at its own address: debug, training) or compact (inference)
    bool chunk_fitted = false;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    NameMap<float*> act;  // named activation buffers inside ws
    NameMap<std::pair<int, int>> act_shape;

    // host-pointer entry points: two I/O slots so H2D(i+1), compute(i) and D2H(i-1) overlap
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    void* pin_out[2] = {nullptr, nullptr};   // pinned landing zone for results (chunk * 2048 B each)
    void* pin_in[2] = {nullptr, nullptr};    // pinned gather buffers of the leaf-pointer entry points (lazy)
    size_t pin_in_bytes = 0;
    float* dev_leaves[2] = {nullptr, nullptr};  // chunk * 512 floats
    uint8_t* dev_idx[2] = {nullptr, nullptr};   // chunk * 64 bytes
    int64_t dev_io_leaves = 0;

    // profiling
    bool profiling = false;
    std::vector<KernelTimer> timers;

    bool debug = false;

    // codebook training (vqhip_train_*): host copies of what the inference tables are rebuilt from, live state flag,
    // latent / index scratch for one training batch
    std::vector<float> h_proj_w, h_proj_b;
    bool training = false, tables_stale = false;
    float* tr_z = nullptr;
    uint8_t* tr_idx = nullptr;
    int64_t tr_leaves = 0;
    // full training step (vq_train_full.inc)
    std::vector<float> h_params;                                   // raw tensors, state_dict order
    std::map<std::string, std::pair<int64_t, int64_t>> p_off;      // name -> (offset, count) in the flat parameter vector
    float *ft_P = nullptr, *ft_M = nullptr, *ft_V = nullptr;       // parameters, AdamW moments (device, flat)
    void* ft_refrag_jobs = nullptr;                                 // device job table of refrag_multi_k (vq_train_full.inc: refrag_all)
    int ft_refrag_n = 0, ft_refrag_wgs = 0;
    const void* ft_refrag_P = nullptr;                              // the parameter block the cached job table points into
    char* ft_ws = nullptr;                                         // training workspace (saved activations, gradients)
    int64_t ft_tiles = 0;
    char* ft_part = nullptr;                                       // partial-gradient scratch
    size_t ft_part_bytes = 0;
    bool full_training = false, keep_y1 = false, weights_stale = false;
    bool train_gn_fused = true;      // GroupNorm + ReLU backward as one pass per layer (gn_bwd_fused_k); VQHIP_TRAIN_GNBWD=split: sums + finish + apply
    bool train_bias_main = true;     // bias gradients on the data-gradient stream when there is no reduction stream (VQHIP_TRAIN_BIAS=side: beside the weight gradients)
    bool train_red_stream = false;   // round 6, first attempt: bias sums, GroupNorm-affine / attention-weight reductions on a third stream (VQHIP_TRAIN_BIAS=third; fast or slow depending on the hardware queue the stream lands on)
    bool train_red_deferred = true;  // round 6: ... as two multi-job launches at the end of the data-gradient chain (csum_multi_k, reduce_multi_k); VQHIP_TRAIN_BIAS=main|side|third: the other arrangements
    bool train_ema_early = true;     // round 6: the codebook statistics start on the side stream as soon as the assignment exists, beside the decoder's forward (VQHIP_TRAIN_EMA_AT=backward: with the backward pass)
    bool train_side_stream = true;   // training backward: weight / bias gradients on a second stream beside the data-gradient chain (VQHIP_TRAIN_STREAMS=1: one stream)
    hipStream_t ft_side = nullptr;
    hipStream_t ft_red_shared = nullptr;   // ... the same on a plain stream (shares a hardware queue with the weight-gradient stream): large batches
    int64_t train_red_own_leaves = 2048;   // batches up to this size use ft_red (VQHIP_TRAIN_RED_OWN_LEAVES)
    bool train_r64_quarters = true;        // 64 -> 64 convs of small training batches in four cout quarters with resident weights (VQHIP_TRAIN_R64=whole: one workgroup per row)
    bool train_dgrad_small_lds = true;     // data gradients of the 4^3 layers with streamed weight windows (VQHIP_TRAIN_DGRAD=resident: LDS-resident weights)
    int train_wgrad_cu_pct = 100;          // wgrad_rows4_k: slices as a percentage of one-per-CU (VQHIP_TRAIN_WGRAD_CU_PCT)
    bool ft_egate_early = false;           // this step's encoder attention gates were computed beside the forward pass
    hipStream_t ft_red = nullptr;    // third stream of the training step (round 6): the small reductions nothing on the data-gradient chain waits for (bias sums, GroupNorm-affine and attention-weight reductions)
    std::vector<hipEvent_t> ft_ev;   // fork / join events of the side stream, reused every step
    size_t ft_ev_next = 0;
    bool train_wgrad_rows = true;    // training backward: weight gradients of the k3 layers at 4^3 by wgrad_rows4_k (VQHIP_TRAIN_WGRAD=pairs: wgrad32_k)
    bool train_stem_lut = true;      // training forward: decoder stem through the (tap, code) table rebuilt every step (VQHIP_TRAIN_STEM=conv: the real conv)
    bool train_ema_lists = true;     // codebook statistics from per-(segment, code) member lists (vq_ema_lists_k + vq_ema_gather_k); VQHIP_TRAIN_EMA=scan: every (code, segment) wave scans the indices
    bool train_folded_tail = true;   // training step: up_conv + PixelShuffle3D + final as one folded operator (vq_train_tail.h); VQHIP_TRAIN_TAIL=unfolded keeps the layer-by-layer tail
    float* z4_out = nullptr;   // set around encode_chunk by the training forward: latent also in the L4 layout
    char* tr_part = nullptr;   // per-(code, row segment) partial statistics
    size_t tr_part_bytes = 0;
    float* tr_recon = nullptr;  // reconstruction scratch of vqhip_train_eval_device
    int64_t tr_recon_leaves = 0;
    double* tr_loss_part = nullptr;
};

namespace {

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError(); /* the slot is sticky until read: do not let it fail the next launch check */ \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return VQHIP_ERR_DEVICE;                                                            \
        }                                                                                       \
    } while (0)

int fail(vqhip_codec* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

// ---------------- weight pack (vqvdb_amd/weightpack.py) ----------------
bool parse_pack(const unsigned char* p, size_t n, std::map<std::string, PackTensor>& out, std::string& err)
{
    if (n < 16 || std::memcmp(p, "VQWPACK1", 8) != 0) {
        err = "weight pack: bad magic (expected VQWPACK1)";
        return false;
    }
    uint32_t nt;
    std::memcpy(&nt, p + 8, 4);
    const size_t ent = 108;
    if (16 + (size_t)nt * ent > n) {
        err = "weight pack: truncated table";
        return false;
    }
    for (uint32_t i = 0; i < nt; ++i) {
        const unsigned char* e = p + 16 + (size_t)i * ent;
        char name[65];
        std::memcpy(name, e, 64);
        name[64] = 0;
        uint32_t ndim, dims[6];
        uint64_t off, cnt;
        std::memcpy(&ndim, e + 64, 4);
        std::memcpy(dims, e + 68, 24);
        std::memcpy(&off, e + 92, 8);
        std::memcpy(&cnt, e + 100, 8);
        // overflow-safe bounds: off and cnt come from the (user-supplied) file
        if (ndim > 6 || (off & 3) || off > n || cnt > (n - off) / 4) {
            err = std::string("weight pack: tensor '") + name + "' out of bounds";
            return false;
        }
        uint64_t prod = 1;
        for (uint32_t d = 0; d < ndim; ++d) {
            if (dims[d] != 0 && prod > UINT64_MAX / dims[d]) prod = UINT64_MAX;
            else prod *= dims[d];
        }
        if (prod != cnt) {   // every consumer reads prod(dims) floats from the tensor
            err = std::string("weight pack: tensor '") + name + "' count does not match its shape";
            return false;
        }
        PackTensor t;
        t.dims.assign(dims, dims + ndim);
        t.data = reinterpret_cast<const float*>(p + off);
        t.count = cnt;
        out[name] = t;
    }
    return true;
}

const PackTensor* need(const std::map<std::string, PackTensor>& pk, const char* name, std::initializer_list<uint32_t> dims, std::string& err)
{
    auto it = pk.find(name);
    if (it == pk.end()) {
        err = std::string("weight pack: missing tensor '") + name + "'";
        return nullptr;
    }
    if (it->second.dims != std::vector<uint32_t>(dims)) {
        err = std::string("weight pack: tensor '") + name + "' has unexpected shape";
        return nullptr;
    }
    return &it->second;
}

// ---------------- fragment repacks (host, once at create) ----------------
// 32x32x2 A-fragments: [tap][u][mt][lane][i] = W[cout = 32mt + (lane&31)][cin = 8u + 4(lane>>5) + i][tap]
std::vector<float> frag32(const float* W, int COUT, int CIN, int KT)
{
    const int NU = CIN / 8, NMT = COUT / 32;
    std::vector<float> f((size_t)KT * NU * NMT * 64 * 4);
    for (int tap = 0; tap < KT; ++tap)
        for (int u = 0; u < NU; ++u)
            for (int mt = 0; mt < NMT; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int co = 32 * mt + (lane & 31), ci = 8 * u + 4 * (lane >> 5) + i;
                        f[((((size_t)tap * NU + u) * NMT + mt) * 64 + lane) * 4 + i] = W[((size_t)co * CIN + ci) * KT + tap];
                    }
    return f;
}
// D-fragment order of a per-cout vector: [(mt*2+q)*16 + r] = v[32mt + (r&3) + 8(r>>2) + 4q]
std::vector<float> dfrag32(const float* v, int COUT)
{
    std::vector<float> f(COUT);
    for (int mt = 0; mt < COUT / 32; ++mt)
        for (int q = 0; q < 2; ++q)
            for (int r = 0; r < 16; ++r) f[(mt * 2 + q) * 16 + r] = v[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * q];
    return f;
}
// 16x16x4 A-fragments for Cin = Cout = 16: [tap][lane][i] = W[cout = lane&15][cin = 4(lane>>4) + i][tap]
std::vector<float> frag16(const float* W, int KT)
{
    std::vector<float> f((size_t)KT * 64 * 4);
    for (int tap = 0; tap < KT; ++tap)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) f[((size_t)tap * 64 + lane) * 4 + i] = W[((size_t)(lane & 15) * 16 + 4 * (lane >> 4) + i) * KT + tap];
    return f;
}
// 16x16x4 A-fragments of conv_rows16_k: [tap][cb][mt][lane][i] = W[16mt + (lane&15)][16cb + 4(lane>>4) + i][tap]
std::vector<float> frag16g(const float* W, int COUT, int CIN, int KT)
{
    const int CBN = CIN / 16, MTN = COUT / 16;
    std::vector<float> f((size_t)KT * CBN * MTN * 64 * 4);
    for (int tap = 0; tap < KT; ++tap)
        for (int cb = 0; cb < CBN; ++cb)
            for (int mt = 0; mt < MTN; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i)
                        f[((((size_t)tap * CBN + cb) * MTN + mt) * 64 + lane) * 4 + i] =
                            W[((size_t)(16 * mt + (lane & 15)) * CIN + 16 * cb + 4 * (lane >> 4) + i) * KT + tap];
    return f;
}
// first conv: [(kd*3+kh)][lane] = W[cout = lane&15][0][kd][kh][kw = lane>>4] (0 for the pad slot)
std::vector<float> frag_first(const float* W)
{
    std::vector<float> f(9 * 64);
    for (int t = 0; t < 9; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            const int kw = lane >> 4;
            f[t * 64 + lane] = kw < 3 ? W[(lane & 15) * 27 + t * 3 + kw] : 0.0f;
        }
    return f;
}

// Flattened (output, valid tap) schedules (StepEnt, vq_kernels.h).  Zero-padding taps never enter
// the table; taps appear in ascending (kd,kh,kw) order per output (arithmetic contract).
std::vector<int> steps_conv(int SI, int SO, int KS, int STRIDE, int PAD, int KWG)
{
    std::vector<int> t;
    for (int od = 0; od < SO; ++od)
        for (int oh = 0; oh < SO; ++oh)
            for (int ow = 0; ow < SO; ++ow) {
                const size_t first = t.size();
                for (int kd = 0; kd < KS; ++kd)
                    for (int kh = 0; kh < KS; ++kh) {
                        const int id = od * STRIDE - PAD + kd, ih = oh * STRIDE - PAD + kh;
                        if (id < 0 || id >= SI || ih < 0 || ih >= SI) continue;
                        int run = 0;  // valid taps along kw are contiguous in input position and tap index
                        for (int kw = 0; kw < KS; ++kw) {
                            const int iw = ow * STRIDE - PAD + kw;
                            if (iw < 0 || iw >= SI) continue;
                            if (run == 0 || run == KWG) {
                                t.insert(t.end(), {(id * SI + ih) * SI + iw, (kd * KS + kh) * KS + kw, (od * SO + oh) * SO + ow, 0});
                                run = 0;
                            }
                            ++run;
                            t[t.size() - 1] = (t[t.size() - 1] & 0xff) | (run << 8);
                        }
                    }
                t[first + 3] |= 1;
                t[t.size() - 1] |= 2;
            }
    return t;
}
// row schedule for conv_rows16_k: one step = (output row (od,oh) of SO positions, valid (kd,kh));
// x = input row base position, y = first tap (kw = 0) of the (kd,kh) run, z = output row base
std::vector<int> steps_rows(int SI, int SO, int KS, int STRIDE, int PAD)
{
    std::vector<int> t;
    for (int od = 0; od < SO; ++od)
        for (int oh = 0; oh < SO; ++oh) {
            const size_t first = t.size();
            for (int kd = 0; kd < KS; ++kd)
                for (int kh = 0; kh < KS; ++kh) {
                    const int id = od * STRIDE - PAD + kd, ih = oh * STRIDE - PAD + kh;
                    if (id < 0 || id >= SI || ih < 0 || ih >= SI) continue;
                    t.insert(t.end(), {(id * SI + ih) * SI, (kd * KS + kh) * KS, (od * SO + oh) * SO, 1 << 8});
                }
            t[first + 3] |= 1;
            t[t.size() - 1] |= 2;
        }
    return t;
}

// conv8_c16_k<NR>: one step = (od, output row group oh0 = NR*g, valid kd, input row ih in [oh0-1, oh0+NR]);
// w bits 8+4*rw.. = kh+1 of output row oh0+rw (0 = that row is not fed by this input row)
std::vector<int> steps_rowgroups8(int NR)
{
    std::vector<int> t;
    for (int od = 0; od < 8; ++od) {
        for (int g = 0; g < 8 / NR; ++g) {
            const int oh0 = NR * g;
            const size_t first = t.size();
            for (int kd = 0; kd < 3; ++kd) {
                const int id = od + kd - 1;
                if (id < 0 || id > 7) continue;
                for (int ih = oh0 - 1; ih <= oh0 + NR; ++ih) {
                    if (ih < 0 || ih > 7) continue;
                    int feeds = 0;
                    for (int rw = 0; rw < NR; ++rw) {
                        const int kh = ih - (oh0 + rw) + 1;
                        if (kh >= 0 && kh <= 2) feeds |= (kh + 1) << (8 + 4 * rw);
                    }
                    t.insert(t.end(), {(id * 8 + ih) * 8, kd * 3, (od * 8 + oh0) * 8, feeds});
                }
            }
            t[first + 3] |= 1;
            t[t.size() - 1] |= 2;
        }
    }
    return t;
}

// first conv: one step = (output row, valid kd); kh validity as a 3-bit mask
std::vector<int> steps_rows8_kd()
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

int upload_i(vqhip_codec* c, const char* name, const std::vector<int>& v)
{
    int* d = nullptr;
    HIPCHK(c, hipMalloc(&d, v.size() * sizeof(int)));
    HIPCHK(c, hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    c->dw[name] = reinterpret_cast<float*>(d);
    c->nsteps[name] = (int)(v.size() / 4);
    return VQHIP_OK;
}

// schedule + index of the first step of every output group (first-flagged steps), for position-split launches
int upload_steps(vqhip_codec* c, const std::string& name, const std::vector<int>& t)
{
    int rc = upload_i(c, name.c_str(), t);
    if (rc) return rc;
    std::vector<int> g;
    for (size_t i = 0; i < t.size() / 4; ++i)
        if (t[4 * i + 3] & 1) g.push_back((int)i);
    g.push_back((int)(t.size() / 4));
    while (g.size() % 4) g.push_back(g.back());  // upload_i counts int4 entries
    const int n_steps = c->nsteps[name];
    rc = upload_i(c, (name + ".grp").c_str(), g);
    c->nsteps[name] = n_steps;
    return rc;
}

int upload(vqhip_codec* c, const char* name, const std::vector<float>& v)
{
    float* d = nullptr;
    auto it = c->dw.find(name);
    if (it != c->dw.end()) {  // refresh of a derived table (same size by construction)
        HIPCHK(c, hipMemcpy(it->second, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return VQHIP_OK;
    }
    HIPCHK(c, hipMalloc(&d, v.size() * sizeof(float)));
    HIPCHK(c, hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    c->dw[name] = d;
    return VQHIP_OK;
}
int upload(vqhip_codec* c, const char* name, const PackTensor* t)
{
    return upload(c, name, std::vector<float>(t->data, t->data + t->count));
}

// Projection folded into the codebook (contract: oracle vqfold_build): Ep = E P in fp64 (j ascending),
// h_k = sum b e - (sum e^2) / 2, both rounded to fp32 once.  E = codebook [256][128] on the host.
int build_vq_fold(vqhip_codec* c, const float* E)
{
    std::vector<float> ep(256 * 32), ck(256);
    const float* P = c->h_proj_w.data();  // [128][32]
    const float* pb = c->h_proj_b.data();
    for (int k = 0; k < 256; ++k) {
        for (int ch = 0; ch < 32; ++ch) {
            double acc = 0.0;
            for (int jx = 0; jx < 128; ++jx) acc = __builtin_fma((double)E[k * 128 + jx], (double)P[jx * 32 + ch], acc);
            ep[k * 32 + ch] = (float)acc;
        }
        double cc = 0.0, bb = 0.0;
        for (int jx = 0; jx < 128; ++jx) {
            cc = __builtin_fma((double)E[k * 128 + jx], (double)E[k * 128 + jx], cc);
            bb = __builtin_fma((double)pb[jx], (double)E[k * 128 + jx], bb);
        }
        ck[k] = (float)(bb - 0.5 * cc);   // h_k = -c_k / 2: where the score chain of code k starts (oracle vqfold_build)
    }
    int rc;
    if ((rc = upload(c, "vq.ep", frag32(ep.data(), 256, 32, 1)))) return rc;
    return upload(c, "vq.ck", dfrag32(ck.data(), 256));
}

// Folded decoder tail: up_conv (64->256,k3 @4^3) -> PixelShuffle3D(2) -> final (32->1,k3 @8^3) is one
// linear map (no nonlinearity in between, VQVAE_v2.py:274-275).  Composite weights per output voxel,
// built in fp64 in the contract's order (see oracle tail_build) and rounded to fp32 once; laid out as
// MFMA A-fragments per (output slab d, input position p) with the 128 voxels of the slab as rows.
int build_folded_tail(vqhip_codec* c, const float* Wu, const float* bu, const float* Wf, const float* bf)
{
    std::vector<double> G((size_t)27 * 8 * 27 * 64);
    double Bg[27][8];
    for (int dl = 0; dl < 27; ++dl)
        for (int sb = 0; sb < 8; ++sb) {
            double b = 0.0;
            for (int oc = 0; oc < 32; ++oc) b = __builtin_fma((double)Wf[oc * 27 + dl], (double)bu[oc * 8 + sb], b);
            Bg[dl][sb] = b;
            for (int t = 0; t < 27; ++t)
                for (int ci = 0; ci < 64; ++ci) {
                    double a = 0.0;
                    for (int oc = 0; oc < 32; ++oc)
                        a = __builtin_fma((double)Wf[oc * 27 + dl], (double)Wu[((size_t)(oc * 8 + sb) * 64 + ci) * 27 + t], a);
                    G[(((size_t)dl * 8 + sb) * 27 + t) * 64 + ci] = a;
                }
        }
    std::vector<float> wc((size_t)512 * 64 * 64), bc(512);
    std::vector<double> acc(64 * 64);
    for (int od = 0; od < 8; ++od)
        for (int oh = 0; oh < 8; ++oh)
            for (int ow = 0; ow < 8; ++ow) {
                const int ov = (od * 8 + oh) * 8 + ow;
                double b = (double)bf[0];
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int dd = 0; dd < 3; ++dd)
                    for (int dh = 0; dh < 3; ++dh)
                        for (int dw = 0; dw < 3; ++dw) {
                            const int zd = od + dd - 1, zh = oh + dh - 1, zw = ow + dw - 1;
                            if (zd < 0 || zd > 7 || zh < 0 || zh > 7 || zw < 0 || zw > 7) continue;
                            const int dl = (dd * 3 + dh) * 3 + dw;
                            const int cd = zd >> 1, ch = zh >> 1, cw = zw >> 1, sb = (zd & 1) * 4 + (zh & 1) * 2 + (zw & 1);
                            b = b + Bg[dl][sb];
                            for (int td = 0; td < 3; ++td)
                                for (int th = 0; th < 3; ++th)
                                    for (int tw = 0; tw < 3; ++tw) {
                                        const int pd = cd + td - 1, ph = ch + th - 1, pw = cw + tw - 1;
                                        if (pd < 0 || pd > 3 || ph < 0 || ph > 3 || pw < 0 || pw > 3) continue;
                                        const double* g = &G[(((size_t)dl * 8 + sb) * 27 + (td * 3 + th) * 3 + tw) * 64];
                                        double* a = &acc[(size_t)((pd * 4 + ph) * 4 + pw) * 64];
                                        for (int ci = 0; ci < 64; ++ci) a[ci] = a[ci] + g[ci];
                                    }
                        }
                bc[ov] = (float)b;
                for (int i = 0; i < 64 * 64; ++i) wc[(size_t)ov * 4096 + i] = (float)acc[i];
            }
    // fragments + schedule: slab d visits positions pd in [max(0,d-2), min(3,d+2)] x all (ph,pw), ascending
    std::vector<float> frags, bias;
    std::vector<int> steps;
    std::vector<float> wslab((size_t)128 * 64);
    for (int d = 0; d < 4; ++d) {
        const size_t first = steps.size();
        for (int p = std::max(0, d - 2) * 16; p < (std::min(3, d + 2) + 1) * 16; ++p) {
            for (int row = 0; row < 128; ++row)
                for (int ci = 0; ci < 64; ++ci) wslab[(size_t)row * 64 + ci] = wc[((size_t)(d * 128 + row) * 64 + p) * 64 + ci];
            const std::vector<float> f = frag32(wslab.data(), 128, 64, 1);
            steps.insert(steps.end(), {p, (int)(frags.size() / f.size()), d, 1 << 8});
            frags.insert(frags.end(), f.begin(), f.end());
        }
        steps[first + 3] |= 1;
        steps[steps.size() - 1] |= 2;
        const std::vector<float> bfrag = dfrag32(bc.data() + d * 128, 128);
        bias.insert(bias.end(), bfrag.begin(), bfrag.end());
    }
    // the same operator as 16x16x4 fragments for the smallest batches (tail_small16_k): [step][voxel block 8][uu 4][lane][e]
    std::vector<float> frags16((size_t)224 * 8 * 4 * 64 * 4);
    {
        size_t step = 0;
        for (int d = 0; d < 4; ++d)
            for (int p = std::max(0, d - 2) * 16; p < (std::min(3, d + 2) + 1) * 16; ++p, ++step)
                for (int mb = 0; mb < 8; ++mb)
                    for (int uu = 0; uu < 4; ++uu)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int u = 2 * uu + (e >> 1), mf = e & 1, m = lane & 15, k = lane >> 4;
                                const int ch = 8 * u + 4 * (k & 1) + (k >> 1) + 2 * mf;
                                frags16[((((step * 8 + mb) * 4 + uu) * 64) + lane) * 4 + e] = wc[((size_t)(d * 128 + 16 * mb + m) * 64 + p) * 64 + ch];
                            }
    }
    // ... and as the weight stream of tail_rows16_k (vq_tail_rows.h): one 32 KB slice per phase (unit, pd, ph, pw) in the order the
    // kernel walks them, a slice = the 4 KB blocks [uu 4][lane][e] of the tiles that input row feeds, dense, ascending tile id
    std::vector<float> wrows((size_t)TR_STREAM_SLICES * (TR_SLICE / 4), 0.0f);   // (padded: the kernel requests slices three phases ahead)
    {
        size_t t = 0, tile_rows = 0;
        for (int unit = 0; unit < 5; ++unit) {
            const bool pair = unit >= 1 && unit <= 3;
            const int od0 = unit == 0 ? 0 : unit == 4 ? 7 : 2 * unit - 1;
            for (int pd = tr_lo(od0); pd <= tr_hi(od0); ++pd)
                for (int ph = 0; ph < 4; ++ph) {
                    const unsigned mask = tr_mask(pair, ph);
                    tile_rows += tr_popc(mask);
                    for (int pw = 0; pw < 4; ++pw, ++t) {
                        const int p = (pd * 4 + ph) * 4 + pw;
                        for (int i = 0; i < tr_popc(mask); ++i) {
                            const int tid = tr_nth(mask, i), ca = tr_cell_a(pair, od0, tid), cb = tr_cell_b(pair, od0, tid);
                            float* blk = &wrows[t * (TR_SLICE / 4) + (size_t)i * 1024];
                            for (int uu = 0; uu < 4; ++uu)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 4; ++e) {
                                        const int u = 2 * uu + (e >> 1), mf = e & 1, m = lane & 15, k = lane >> 4;
                                        const int ch = 8 * u + 4 * (k & 1) + (k >> 1) + 2 * mf;
                                        const int vox = (m < 8 ? ca : cb) * 8 + (m & 7);
                                        blk[(uu * 64 + lane) * 4 + e] = wc[((size_t)vox * 64 + p) * 64 + ch];
                                    }
                        }
                    }
                }
        }
        if (t != TR_PHASES || tile_rows != TR_TILE_ROWS) return fail(c, VQHIP_ERR_MODEL, "folded tail: row schedule does not match tail_rows16_k");
    }
    // ... and of tail_groups16_k (vq_tail_groups.h): three groups of output planes, 48 KB slices of up to 11 blocks (pair tiles, then the
    // group's single tiles while their planes last)
    // (built and uploaded only when VQHIP_TAIL=groups selects that kernel: 7.8 MB of host work per weight refresh otherwise spent for nothing)
    std::vector<float> wgroups;
    if (c->tail_groups) {
        wgroups.assign((size_t)TG_STREAM_SLICES * (TG_SLICE / 4), 0.0f);
        size_t t = 0, tile_rows = 0;
        for (int g = 0; g < 3; ++g)
            for (int pd = tg_pd_lo(g); pd <= tg_pd_hi(g); ++pd)
                for (int ph = 0; ph < 4; ++ph) {
                    const unsigned mask = tg_mask(tg_with_single(g, pd), ph);
                    tile_rows += tr_popc(mask);
                    for (int pw = 0; pw < 4; ++pw, ++t) {
                        const int p = (pd * 4 + ph) * 4 + pw;
                        for (int i = 0; i < tr_popc(mask); ++i) {
                            const int tid = tr_nth(mask, i), ca = tg_cell_a(g, tid), cb = tg_cell_b(g, tid);
                            float* blk = &wgroups[t * (TG_SLICE / 4) + (size_t)i * 1024];
                            for (int uu = 0; uu < 4; ++uu)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 4; ++e) {
                                        const int u = 2 * uu + (e >> 1), mf = e & 1, m = lane & 15, k = lane >> 4;
                                        const int ch = 8 * u + 4 * (k & 1) + (k >> 1) + 2 * mf;
                                        const int vox = (m < 8 ? ca : cb) * 8 + (m & 7);
                                        blk[(uu * 64 + lane) * 4 + e] = wc[((size_t)vox * 64 + p) * 64 + ch];
                                    }
                        }
                    }
                }
        if (t != TG_PHASES || tile_rows != TR_TILE_ROWS) return fail(c, VQHIP_ERR_MODEL, "folded tail: group schedule does not match tail_groups16_k");
    }
    int rc;
    if (c->tail_groups && (rc = upload(c, "tail.wgroups", wgroups))) return rc;
    if ((rc = upload(c, "tail.wrows", wrows))) return rc;
    if ((rc = upload(c, "tail.w", frags))) return rc;
    if ((rc = upload(c, "tail.b", bias))) return rc;
    if ((rc = upload(c, "tail.w16", frags16))) return rc;
    if ((rc = upload(c, "tail.braw", bc))) return rc;
    return c->dw.count("steps.tail") ? VQHIP_OK : upload_steps(c, "steps.tail", steps);  // the schedule does not depend on the weights
}

// the 44 trainable tensors in the reference's parameter order (model.parameters(), python/VQVAE_v2.py)
const char* const kTrainable[] = {
    "encoder.pre.0.weight", "encoder.pre.0.bias", "encoder.pre.1.weight", "encoder.pre.1.bias",
    "encoder.pre.3.gn1.weight", "encoder.pre.3.gn1.bias", "encoder.pre.3.conv1.weight", "encoder.pre.3.conv1.bias",
    "encoder.pre.3.gn2.weight", "encoder.pre.3.gn2.bias", "encoder.pre.3.conv2.weight", "encoder.pre.3.conv2.bias",
    "encoder.down.weight", "encoder.down.bias",
    "encoder.res_stack.0.gn1.weight", "encoder.res_stack.0.gn1.bias", "encoder.res_stack.0.conv1.weight", "encoder.res_stack.0.conv1.bias",
    "encoder.res_stack.0.gn2.weight", "encoder.res_stack.0.gn2.bias", "encoder.res_stack.0.conv2.weight", "encoder.res_stack.0.conv2.bias",
    "encoder.attn.fc.0.weight", "encoder.attn.fc.2.weight", "encoder.proj.weight", "encoder.proj.bias",
    "decoder.stem.0.weight", "decoder.stem.0.bias", "decoder.stem.1.weight", "decoder.stem.1.bias",
    "decoder.res_stack.0.gn1.weight", "decoder.res_stack.0.gn1.bias", "decoder.res_stack.0.conv1.weight", "decoder.res_stack.0.conv1.bias",
    "decoder.res_stack.0.gn2.weight", "decoder.res_stack.0.gn2.bias", "decoder.res_stack.0.conv2.weight", "decoder.res_stack.0.conv2.bias",
    "decoder.attn.fc.0.weight", "decoder.attn.fc.2.weight", "decoder.up_conv.weight", "decoder.up_conv.bias",
    "decoder.final.weight", "decoder.final.bias",
};

int load_weights(vqhip_codec* c, const std::map<std::string, PackTensor>& pk)
{
    std::string err;
#define NEED(var, name, ...)                                  \
    const PackTensor* var = need(pk, name, {__VA_ARGS__}, err); \
    if (!var) return fail(c, VQHIP_ERR_MODEL, err);
    NEED(e0w, "encoder.pre.0.weight", 16, 1, 3, 3, 3) NEED(e0b, "encoder.pre.0.bias", 16)
    NEED(eg0w, "encoder.pre.1.weight", 16) NEED(eg0b, "encoder.pre.1.bias", 16)
    NEED(r16g1w, "encoder.pre.3.gn1.weight", 16) NEED(r16g1b, "encoder.pre.3.gn1.bias", 16)
    NEED(r16c1w, "encoder.pre.3.conv1.weight", 16, 16, 3, 3, 3) NEED(r16c1b, "encoder.pre.3.conv1.bias", 16)
    NEED(r16g2w, "encoder.pre.3.gn2.weight", 16) NEED(r16g2b, "encoder.pre.3.gn2.bias", 16)
    NEED(r16c2w, "encoder.pre.3.conv2.weight", 16, 16, 3, 3, 3) NEED(r16c2b, "encoder.pre.3.conv2.bias", 16)
    NEED(edw, "encoder.down.weight", 32, 16, 4, 4, 4) NEED(edb, "encoder.down.bias", 32)
    NEED(r32g1w, "encoder.res_stack.0.gn1.weight", 32) NEED(r32g1b, "encoder.res_stack.0.gn1.bias", 32)
    NEED(r32c1w, "encoder.res_stack.0.conv1.weight", 32, 32, 3, 3, 3) NEED(r32c1b, "encoder.res_stack.0.conv1.bias", 32)
    NEED(r32g2w, "encoder.res_stack.0.gn2.weight", 32) NEED(r32g2b, "encoder.res_stack.0.gn2.bias", 32)
    NEED(r32c2w, "encoder.res_stack.0.conv2.weight", 32, 32, 3, 3, 3) NEED(r32c2b, "encoder.res_stack.0.conv2.bias", 32)
    NEED(efc0, "encoder.attn.fc.0.weight", 8, 32) NEED(efc2, "encoder.attn.fc.2.weight", 32, 8)
    NEED(epw, "encoder.proj.weight", 128, 32, 1, 1, 1) NEED(epb, "encoder.proj.bias", 128)
    NEED(dsw, "decoder.stem.0.weight", 64, 128, 3, 3, 3) NEED(dsb, "decoder.stem.0.bias", 64)
    NEED(dg0w, "decoder.stem.1.weight", 64) NEED(dg0b, "decoder.stem.1.bias", 64)
    NEED(r64g1w, "decoder.res_stack.0.gn1.weight", 64) NEED(r64g1b, "decoder.res_stack.0.gn1.bias", 64)
    NEED(r64c1w, "decoder.res_stack.0.conv1.weight", 64, 64, 3, 3, 3) NEED(r64c1b, "decoder.res_stack.0.conv1.bias", 64)
    NEED(r64g2w, "decoder.res_stack.0.gn2.weight", 64) NEED(r64g2b, "decoder.res_stack.0.gn2.bias", 64)
    NEED(r64c2w, "decoder.res_stack.0.conv2.weight", 64, 64, 3, 3, 3) NEED(r64c2b, "decoder.res_stack.0.conv2.bias", 64)
    NEED(dfc0, "decoder.attn.fc.0.weight", 16, 64) NEED(dfc2, "decoder.attn.fc.2.weight", 64, 16)
    NEED(duw, "decoder.up_conv.weight", 256, 64, 3, 3, 3) NEED(dub, "decoder.up_conv.bias", 256)
    NEED(dfw, "decoder.final.weight", 1, 32, 3, 3, 3) NEED(dfb, "decoder.final.bias", 1)
    NEED(cb, "quantizer.embedding", 256, 128)
#undef NEED
    // raw tensors in state_dict order, for the full training step (vq_train_full.inc): flat parameter vector + offsets
    c->h_params.clear();
    c->p_off.clear();
    for (const char* tn : kTrainable) {
        const PackTensor& t = pk.at(tn);
        c->p_off[tn] = {(int64_t)c->h_params.size(), (int64_t)t.count};
        c->h_params.insert(c->h_params.end(), t.data, t.data + t.count);
    }
    int rc;
#define UP(...)                                 \
    if ((rc = upload(c, __VA_ARGS__)) != VQHIP_OK) return rc;
    UP("e0.w", frag_first(e0w->data)) UP("e0.b", e0b) UP("eg0.w", eg0w) UP("eg0.b", eg0b)
    UP("r16g1.w", r16g1w) UP("r16g1.b", r16g1b) UP("r16c1.w", frag16(r16c1w->data, 27)) UP("r16c1.b", r16c1b)
    UP("r16g2.w", r16g2w) UP("r16g2.b", r16g2b) UP("r16c2.w", frag16(r16c2w->data, 27)) UP("r16c2.b", r16c2b)
    UP("ed.w", frag32(edw->data, 32, 16, 64)) UP("ed.b", dfrag32(edb->data, 32))
    UP("r32g1.w", r32g1w) UP("r32g1.b", r32g1b) UP("r32c1.w", frag32(r32c1w->data, 32, 32, 27)) UP("r32c1.b", dfrag32(r32c1b->data, 32))
    UP("r32g2.w", r32g2w) UP("r32g2.b", r32g2b) UP("r32c2.w", frag32(r32c2w->data, 32, 32, 27)) UP("r32c2.b", dfrag32(r32c2b->data, 32))
    UP("efc0", efc0) UP("efc2", efc2) 
    UP("ds.w", dsw) UP("ds.b", dsb) UP("dg0.w", dg0w) UP("dg0.b", dg0b)
    UP("r64g1.w", r64g1w) UP("r64g1.b", r64g1b) UP("r64c1.w", frag32(r64c1w->data, 64, 64, 27)) UP("r64c1.b", dfrag32(r64c1b->data, 64))
    UP("r64g2.w", r64g2w) UP("r64g2.b", r64g2b) UP("r64c2.w", frag32(r64c2w->data, 64, 64, 27)) UP("r64c2.b", dfrag32(r64c2b->data, 64))
    UP("r64c1.w16", frag16g(r64c1w->data, 64, 64, 27)) UP("r64c1.braw", r64c1b) UP("r64c2.w16", frag16g(r64c2w->data, 64, 64, 27)) UP("r64c2.braw", r64c2b)
    UP("r32c1.w16", frag16g(r32c1w->data, 32, 32, 27)) UP("r32c1.braw", r32c1b) UP("r32c2.w16", frag16g(r32c2w->data, 32, 32, 27)) UP("r32c2.braw", r32c2b)
    UP("ed.w16", frag16g(edw->data, 32, 16, 64)) UP("ed.braw", edb)
    UP("dfc0", dfc0) UP("dfc2", dfc2) if ((rc = build_folded_tail(c, duw->data, dub->data, dfw->data, dfb->data))) return rc;
    UP("cb", cb)
    c->h_proj_w.assign(epw->data, epw->data + 128 * 32);
    c->h_proj_b.assign(epb->data, epb->data + 128);
    UP("tr.wproj", frag32(epw->data, 128, 32, 1)) UP("tr.bproj", dfrag32(epb->data, 128))
    if ((rc = build_vq_fold(c, cb->data))) return rc;
#undef UP
    if ((rc = upload_steps(c, "steps.k3s1_4", steps_conv(4, 4, 3, 1, 1, 1)))) return rc;     // one tap per step (streamed layers)
    if ((rc = upload_steps(c, "steps.rowgroups8_4", steps_rowgroups8(4)))) return rc;
    if ((rc = upload_steps(c, "steps.rowgroups8_2", steps_rowgroups8(2)))) return rc;   // tiny batches: 32 groups of 2 rows
    if ((rc = upload_steps(c, "steps.rows_k3_4", steps_rows(4, 4, 3, 1, 1)))) return rc;
    if ((rc = upload_steps(c, "steps.rows_k4s2_8", steps_rows(8, 4, 4, 2, 1)))) return rc;
    if ((rc = upload_steps(c, "steps.rows8kd", steps_rows8_kd()))) return rc;
    {
        // decoder stem as a per-(tap, code) partial-sum table (stem_lut_k), built on the device once
        float* T = nullptr;
        HIPCHK(c, hipMalloc(&T, (size_t)27 * 256 * 64 * sizeof(float)));
        c->dw["ds.lut"] = T;
        hipLaunchKernelGGL(build_stem_lut_k, dim3(27 * 256), dim3(64), 0, c->stream, c->dw["ds.w"], c->dw["cb"], T);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VQHIP_OK;
}

// ---------------- workspace ----------------
// Inference keeps at most three large activations alive at a time (a conv reads its input and, for the second conv of a residual
// block, the skip tensor, and writes its output), and encode and decode never overlap on one handle — so the COMPACT layout packs
// every large activation of both directions into three regions of 32 KiB per leaf (0.1 MB per leaf, 6.8 GB for a 65 536-leaf
// chunk instead of 15 GB).  Debug mode and the training step need every intermediate at its own address: FULL layout.
struct ActSpec {
    const char* name;
    int C, NP;     // floats per leaf = C*NP ; C==0 -> per-leaf scalars, NP = count
    int region;    // compact layout: -1 own buffer, 0..2 shared region, -2 not allocated (full layout only)
};
const ActSpec kActs[] = {
    {"xr", 1, 768, -1},        // first-conv input, row layout with halo (pack_leaves_k): 64 rows x 12 floats per leaf
    {"xt", 1, 512, -2},        // position-major copy of the input: training / debug only
    // encode: xr -> a1 (R0) -> y4 (R1) -> a6 (R2, skip a1) -> x7 (R0) -> y9 (R1) -> x11 (R2, skip x7); y1 exists only in the
    // small-batch path (written and consumed before y4) and in debug / training
    {"e_y1", 16, 512, 1},      {"e_a1", 16, 512, 0},    {"e_y4", 16, 512, 1},    {"e_a6", 16, 512, 2},
    {"e_x7", 32, 64, 0},       {"e_y9", 32, 64, 1},     {"e_x11", 32, 64, 2},
    // decode: ystem (R0) -> d2 (R1) -> y4 (R2) -> x6 (R0, skip d2) -> voxels
    {"d_ystem", 64, 64, 0},    {"d_d2", 64, 64, 1},     {"d_y4", 64, 64, 2},     {"d_x6", 64, 64, 0},
    {"st_a.mean", 0, 8, -1},   {"st_a.rstd", 0, 8, -1}, {"st_b.mean", 0, 8, -1}, {"st_b.rstd", 0, 8, -1}, {"csum", 0, 64, -1}, {"gate", 0, 64, -1},
    {"part_s", 0, 1024, -1},   {"part_q", 0, 1024, -1},  {"part_c", 0, 1024, -1},   // per-block statistics partials of split launches (fp64 x 512: 16 blocks x 16 slots; fp32 x 1024); the 128 half-row partials of enc conv1 borrow e_a6
};
constexpr int kRegions = 3;

bool want_full_layout(const vqhip_codec* c) { return c->debug || c->training || c->full_training || c->keep_y1; }

size_t workspace_bytes(int64_t tiles, bool full)
{
    auto sz = [&](size_t per_leaf_floats) { return ((per_leaf_floats * sizeof(float) * 32 * (size_t)tiles) + 255) / 256 * 256; };
    size_t total = 0, region[kRegions] = {0, 0, 0};
    for (const ActSpec& a : kActs) {
        const size_t fl = a.C ? (size_t)a.C * a.NP : (size_t)a.NP;
        if (full || a.region == -1) total += sz(fl);
        else if (a.region >= 0) region[a.region] = std::max(region[a.region], sz(fl));
    }
    if (!full)
        for (size_t r : region) total += r;
    return total;
}

int ensure_workspace(vqhip_codec* c, int64_t n_leaves)
{
    const int64_t tiles = (n_leaves + 31) / 32;
    const bool full = want_full_layout(c);
    if (tiles <= c->ws_tiles && (c->ws_full || !full)) return VQHIP_OK;   // a full layout also serves inference
    // growing within a layout keeps the larger size; a switch to the full layout (debug, training: 2.4x the bytes per leaf) is sized
    // for what THIS call needs — a 32-leaf debug pass on a handle reserved for 65 536 leaves must not ask for 16 GB
    int64_t new_tiles = (full && !c->ws_full) ? tiles : std::max(tiles, c->ws_tiles);
    if (c->ws) {
        HIPCHK(c, hipDeviceSynchronize());   // the workspace may be in use on a caller's stream
        HIPCHK(c, hipFree(c->ws));
        c->ws = nullptr;
        c->ws_tiles = 0;
        c->ws_bytes = 0;
        for (const ActSpec& a : kActs) c->act[a.name] = nullptr;   // (only this workspace's names: the training step registers its own buffers in the same table)
    }
    size_t total = workspace_bytes(new_tiles, full);
    hipError_t e = hipMalloc(&c->ws, total);
    // a failed hipMalloc leaves its error in the thread's last-error slot (ROCm 7.2 keeps it until it is read): clear it, or the next
    // launch check reads "out of memory" for a launch that succeeded and turns a good retry / the promised smaller-chunk call into a failure
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess && new_tiles > tiles) {   // the larger-than-needed size did not fit: what the call needs
        new_tiles = tiles;
        total = workspace_bytes(new_tiles, full);
        e = hipMalloc(&c->ws, total);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e != hipSuccess) {
        // the handle holds no workspace now (ws_bytes 0, every activation pointer null); the chunk is re-fitted to the free memory
        // at the next host-pointer call, so a caller that retries after a transient shortage gets a smaller chunk instead of this error
        c->ws = nullptr;
        c->chunk_fitted = false;
        return fail(c, VQHIP_ERR_NOMEM, std::string("workspace hipMalloc of ") + std::to_string(total >> 20) + " MiB failed: " + hipGetErrorString(e));
    }
    c->ws_bytes = total;
    auto sz = [&](size_t per_leaf_floats) { return ((per_leaf_floats * sizeof(float) * 32 * (size_t)new_tiles) + 255) / 256 * 256; };
    size_t off = 0, region_off[kRegions] = {0, 0, 0};
    if (!full) {   // the three shared regions first
        size_t region[kRegions] = {0, 0, 0};
        for (const ActSpec& a : kActs)
            if (a.region >= 0) region[a.region] = std::max(region[a.region], sz((size_t)a.C * a.NP));
        for (int r = 0; r < kRegions; ++r) region_off[r] = off, off += region[r];
    }
    for (const ActSpec& a : kActs) {
        const size_t fl = a.C ? (size_t)a.C * a.NP : (size_t)a.NP;
        c->act_shape[a.name] = {a.C, a.NP};
        if (full || a.region == -1) {
            c->act[a.name] = reinterpret_cast<float*>(c->ws + off);
            off += sz(fl);
        } else if (a.region >= 0) {
            c->act[a.name] = reinterpret_cast<float*>(c->ws + region_off[a.region]);
        } else {
            c->act[a.name] = nullptr;
        }
    }
    c->ws_tiles = new_tiles;
    c->ws_full = full;
    return VQHIP_OK;
}

// A shared GPU may not have room for the default chunk: halve the chunk until workspace + I/O slots fit into 80 % of the free
// memory (never below 2048 leaves; results do not depend on the chunk size).
void fit_chunk_to_free_memory(vqhip_codec* c)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    free_b += c->ws_bytes;   // what this handle already holds can be re-used
    while (c->chunk > 2048) {
        const size_t need = workspace_bytes((c->chunk + 31) / 32, want_full_layout(c)) + (size_t)c->chunk * (2 * 2048 + 2 * 64);
        if (need <= free_b / 5 * 4) break;
        c->chunk = (c->chunk / 2 + 31) / 32 * 32;
    }
}

// ---------------- launches ----------------
struct Launcher {
    vqhip_codec* c;
    hipStream_t s;
    int64_t leaves;
    int rc = VQHIP_OK;
    int64_t on_main = 0;   // launch groups put on s so far (Bwd::fork skips the event when nothing new has been enqueued)
    template <typename F>
    void run(const char* name, F&& f)
    {
        run_on(s, name, f);
    }
    // the same for launches that f() puts on another stream (profiling events go to that stream)
    template <typename F>
    void run_on(hipStream_t st, const char* name, F&& f)
    {
        if (rc != VQHIP_OK) return;
        if (st == s) ++on_main;
        KernelTimer t;
        if (c->profiling) {
            t.name = name;
            t.leaves = leaves;
            hipEventCreate(&t.start);
            hipEventCreate(&t.stop);
            hipEventRecord(t.start, st);
        }
        f();
        if (c->profiling) {
            hipEventRecord(t.stop, st);
            c->timers.push_back(t);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            c->err = std::string("launch ") + name + ": " + hipGetErrorString(e);
            rc = VQHIP_ERR_DEVICE;
        }
    }
};

template <typename K>
int set_lds(vqhip_codec* c, K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VQHIP_OK;
}

// kernel instantiations -------------------------------------------------------------------
//                                          CIN COUT NPI NPO NW STREAM KWG INMODE GIN RESID GOUT CSUM  OUTMODE
//                                        CIN COUT SI SO KS ST PD NW INMODE GIN RESID GOUT CSUM
// decoder front of full chunks: gathers row by row (PIPE 0: the tap loop is bound by the LDS itself, reads in flight ahead bought nothing),
// the two position halves of a leaf octet on one SIMD (HMAP 1), the first taps of a pass not waiting for the previous pass's stores (RELAX)
constexpr auto k_stem_taps = stem_taps_k<0, 1, 0, true, 0, true>;   // (+ checkerboard row ownership: equal valid rows per wave in every tap)
constexpr auto k_dec_tail = conv_mfma32_k<64, 128, 64, 4, 8, true, 1, 2, 0, false, 0, false, 2>;  // folded up_conv+pixshuf+final
//                                           CIN COUT SI SO KS ST PD INMODE RESID GOUT CSUM
// large passes: kw-outer MFMA order (A fragments of a (kw, channel-block pair) read once for every output of the row)
constexpr auto k_dec_r64c1_r = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, false, 0, true>;   // row-blocked 16x16x4 convs (4^3 outputs)
constexpr auto k_dec_r64c2_r = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, false, 1, false, 8, false, 0, true>;
constexpr auto k_dec_r64c1_rs = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 0, false, false, 1, false, 8, false, 0, true>;  // ... without fused statistics (training forward, data gradients)
constexpr auto k_dec_r64c2_rs = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, false, false, 1, false, 8, false, 0, true>;
// position-split inference launches: fused statistics as per-block partials (PARTS)
constexpr auto k_dec_r64c1_rp = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 1, false, 8, true>;   // (kw-outer would spill here: the per-block statistics partials)
constexpr auto k_dec_r64c2_rp = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, false, 1, false, 8, true, 0, true>;   // kw-outer, register-staged weights: 309 -> 290 us at 4096 leaves
constexpr auto k_enc_down_r = conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true, 1, false, 8, false, 0, true>;      // 8 waves, kw-outer with fragments read a group ahead    // weights LDS-resident
constexpr auto k_enc_down_rs = conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true, 1, false, 8, true, 0, true>;   // position-split launches (8-wave workgroups), per-block partials
constexpr auto k_enc_r32c1_r = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 1, false, 16, false, 0, true>;   // 16 waves behind one LDS copy (4/SIMD), kw-outer
constexpr auto k_enc_r32c1_rs = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 1, false, 8, true, 0, true>;
constexpr auto k_enc_r32c2_r = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, true, 0, true, true, 1, false, 16, false, 0, true>;
constexpr auto k_enc_r32c2_rs = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, true, 0, true, true, 1, false, 8, true, 0, true>;
constexpr size_t LDS_ENC_DOWN_R = (size_t)64 * (1 * 2 * 64) * 16;   // 128 KB, resident
constexpr size_t LDS_ENC_R32R = (size_t)27 * (2 * 2 * 64) * 16;     // 108 KB, resident
constexpr size_t LDS_DEC_R64R = (size_t)2 * (3 * 16 * 64) * 16;      // 2 x 48 KB weight window
constexpr size_t LDS_DEC_TAIL = (size_t)2 * (8 * 4 * 64) * 16;    // 2 x 32 KB

// position-split variants that additionally split the output channels over gridDim.z (a wave's serial MFMA chain is the latency
// of a small batch): the 4^3 convs for the tiniest batches; the folded tail has its own small-batch kernel (tail_small_k)
constexpr auto k_dec_r64c1_rs4 = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false, 4, false, 8, true>;
constexpr auto k_dec_r64c2_rs4 = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, false, 4, false, 8, true>;
// ... with the quarter's weights of the whole layer LDS-resident (27 taps x 4 KB = 108 KB): no per-step barrier, no streaming
constexpr auto k_dec_r64c1_rs4r = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, true, 4, false, 8, true, 0, true>;
constexpr auto k_dec_r64c2_rs4r = conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, true, 4, false, 8, true, 0, true>;
constexpr size_t LDS_DEC_R64S4R = (size_t)27 * (4 * 1 * 64) * 16;
constexpr auto k_enc_down_rs2 = conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true, 2, false, 8, true, 0, true>;    // (statistics as per-block partials)
constexpr auto k_enc_r32c1_rs2 = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true, 2, false, 8, true, 0, true>;
constexpr auto k_enc_r32c2_rs2 = conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, true, 0, true, true, 2, false, 8, true, 0, true>;
// position-split (small-batch) variants: 2 tiles per workgroup, no fused statistics

constexpr size_t LDS_LATENT = (16 * 8 * 64 + 4 * 4 * 64) * 16;  // codebook + projection A-fragments (144 KB)

int init_kernel_attrs(vqhip_codec* c)
{
    int rc;
    if ((rc = set_lds(c, k_enc_down_r, LDS_ENC_DOWN_R))) return rc;
    if ((rc = set_lds(c, k_enc_down_rs, LDS_ENC_DOWN_R))) return rc;
    if ((rc = set_lds(c, k_enc_r32c1_r, LDS_ENC_R32R))) return rc;
    if ((rc = set_lds(c, k_enc_r32c1_rs, LDS_ENC_R32R))) return rc;
    if ((rc = set_lds(c, k_enc_r32c2_r, LDS_ENC_R32R))) return rc;
    if ((rc = set_lds(c, k_enc_r32c2_rs, LDS_ENC_R32R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c1_r, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c2_r, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c1_rs, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c2_rs, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c1_rs4r, LDS_DEC_R64S4R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c2_rs4r, LDS_DEC_R64S4R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c1_rp, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, k_dec_r64c2_rp, LDS_DEC_R64R))) return rc;
    if ((rc = set_lds(c, conv8_lds_k<false, true, 8, 0, true>, LDS_CONV8))) return rc;
    if ((rc = set_lds(c, conv8_lds_k<false, true, 16, 0, false>, LDS_CONV8))) return rc;
    if ((rc = set_lds(c, conv8_lds_k<true, false, 16, 0, false>, LDS_CONV8))) return rc;
    if ((rc = set_lds(c, conv8_lds_k<true, false, 8, 0, true>, LDS_CONV8))) return rc;
    if ((rc = set_lds(c, k_stem_taps, LDS_STEM_TAPS))) return rc;
    if ((rc = set_lds(c, tail_rows16_k<0>, LDS_TAIL_ROWS))) return rc;
    if ((rc = set_lds(c, tail_rows32_k<0>, LDS_TAIL_ROWS))) return rc;
    if ((rc = set_lds(c, tail_groups16_k<0>, LDS_TAIL_GROUPS))) return rc;
    if ((rc = set_lds(c, conv_down_lds_k<0>, LDS_CONVDOWN))) return rc;
    if ((rc = set_lds(c, conv4_lds_k<false, true, false, 0, 1>, LDS_CONV4))) return rc;
    if ((rc = set_lds(c, conv4_lds_k<true, false, true, 0, 0>, LDS_CONV4))) return rc;
    if ((rc = set_lds(c, latent_assign_k<8>, LDS_LATENT))) return rc;
    if ((rc = set_lds(c, latent_assign_k<2>, LDS_LATENT))) return rc;
    if ((rc = set_lds(c, latent_assign_k<4>, LDS_LATENT))) return rc;
    return VQHIP_OK;
}

// Where the encoder's GroupNorm statistics and channel sums live.  Inference ping-pongs two buffers (st_a / st_b); the full training
// step needs every layer's statistics again in its backward pass, so there each tensor's statistics go to their own buffer of the
// training workspace (ts_*) — the forward used to overwrite them and a pass of six sequential statistics kernels recomputed them
// from the stored activations (0.17 ms of a 7 ms step).
struct EncStats {
    float *y1m, *y1r, *a1m, *a1r, *y4m, *y4r, *x7m, *x7r, *y9m, *y9r, *csum;
};
EncStats enc_stats_of(vqhip_codec* c)
{
    auto& a = c->act;
    if (c->full_training && c->keep_y1)
        return {a["ts_gn0.mean"], a["ts_gn0.rstd"], a["ts_r16g1.mean"], a["ts_r16g1.rstd"], a["ts_r16g2.mean"], a["ts_r16g2.rstd"],
                a["ts_r32g1.mean"], a["ts_r32g1.rstd"], a["ts_r32g2.mean"], a["ts_r32g2.rstd"], a["ts_ecsum"]};
    return {a["st_a.mean"], a["st_a.rstd"], a["st_b.mean"], a["st_b.rstd"], a["st_a.mean"], a["st_a.rstd"],
            a["st_b.mean"], a["st_b.rstd"], a["st_a.mean"], a["st_a.rstd"], a["csum"]};
}

void launch_latent_assign(vqhip_codec* c, Launcher& L, int64_t n, uint8_t* d_idx, float* d_latent, hipStream_t s, int split)
{
    auto& a = c->act;
    auto& w = c->dw;
    const int nt = (int)((n + 31) / 32);
    // training forward: latent materialised, reference-faithful distance against the live codebook
    L.run("train_codebook_frag", [&] { hipLaunchKernelGGL(codebook_frag_k, dim3(33), dim3(256), 0, s, w["cb"], w["tr.efrag"], w["tr.ee"]); });
    LatentArgs A{};
    A.in = a["e_x11"], A.se_csum = enc_stats_of(c).csum, A.se_fc0 = w["efc0"], A.se_fc2 = w["efc2"];
    A.wproj = w["tr.wproj"], A.bproj = w["tr.bproj"], A.efrag = w["tr.efrag"], A.ee_frag = w["tr.ee"];
    A.idx = d_idx, A.z = d_latent, A.z4 = c->z4_out, A.n_leaves = n, A.n_tiles = nt;
    if (split <= 1 && nt >= 1024) L.run("train_latent_assign", [&] { hipLaunchKernelGGL(latent_assign_k<8>, dim3((nt + 7) / 8), dim3(512), LDS_LATENT, s, A); });
    else if (split > 1 && nt >= 8) {
        // small training batches: 144 KB of LDS = one workgroup per CU, so FOUR waves per workgroup (every SIMD busy; two left half of
        // the CU idle) and twice the position ranges: 0.236 -> 0.13 ms at 2048 leaves
        const int sp = std::min(64, 2 * split);
        L.run("train_latent_assign", [&] { hipLaunchKernelGGL(latent_assign_k<4>, dim3((nt + 3) / 4, sp), dim3(256), LDS_LATENT, s, A); });
    } else L.run("train_latent_assign", [&] { hipLaunchKernelGGL(latent_assign_k<2>, dim3((nt + 1) / 2, split > 1 ? split : 1), dim3(128), LDS_LATENT, s, A); });
}

// gridDim.y for a position-split launch of `wgs` workgroups: enough ranges to reach `target` workgroups (about two rounds
// of what the chip holds at once given the kernel's LDS / register footprint), at least `lo` (the factor used near the
// crossover), at most the kernel's number of output groups (a power of two)
int split_factor(int wgs, int lo, int n_groups, int target)
{
    int ps = lo;
    while (ps < n_groups && (int64_t)wgs * ps < target) ps *= 2;
    return std::min(ps, n_groups);
}

// Which path a pass of nt 32-leaf tiles takes.  One wave per tile fills the chip in rounds of 1024 tiles (1024 SIMDs), so its
// time is a staircase (encode: 7.7 ms at 1024 tiles, 10.2 ms at 1536); the split path is linear (about 6.4 us per tile for
// encode, 5.9 us for decode) and wins up to about 88 % of a full 2048-tile chunk.  Crossovers measured with
// tools/small_batch_probe.py.
bool use_split(const vqhip_codec* c, int nt, bool decode)
{
    if (c->split_tiles >= 0) return nt <= (decode ? 5 * c->split_tiles / 4 : c->split_tiles);
    return nt <= (decode ? 1728 : 1600);   // measured crossovers (r02 v8 kernels): encode 51 200 leaves, decode 55 296
}

// Small batches (too few leaf tiles to fill 1024 SIMDs with one wave per tile): every layer is launched with its output
// groups split over gridDim.y (and, for the tiniest batches, its output channels over gridDim.z) workgroups.  Each workgroup
// covers whole statistics blocks (16 per leaf, the contract's 16-block rule) and stores their partial sums; gn_combine_k /
// csum_combine_k add the blocks in order.  Same results bit for bit, 15-25x lower latency.
int encode_chunk_split(vqhip_codec* c, Launcher& L, int64_t n, uint8_t* d_idx, hipStream_t s, float* d_latent)
{
    const int nt = (int)((n + 31) / 32);
    auto& a = c->act;
    auto& w = c->dw;
    EncStats S = enc_stats_of(c);
    auto od = [&](const char* name) { return reinterpret_cast<const int*>(w[std::string(name) + ".grp"]); };
    const int g4 = (nt + 3) / 4, g2 = (nt + 1) / 2, gh = (2 * nt + 7) / 8;   // gh: workgroups of 8 half tiles (conv_rows16_k)
    double* ps = reinterpret_cast<double*>(a["part_s"]);
    double* pq = reinterpret_cast<double*>(a["part_q"]);
    // Every launch covers whole statistics blocks (16 per leaf) and stores their sums; gn_combine_k adds them in block order.
    auto combine = [&](const char* name, int groups, double inv_n, float* mean, float* rstd) {
        L.run(name, [&] { hipLaunchKernelGGL(gn_combine_k<false>, dim3(nt), dim3(groups * 32), 0, s, ps, pq, mean, rstd, groups, inv_n); });
    };
    if (nt >= 256) {
        // mid-size batches: the first conv twice (statistics, then recompute + normalise + store), like the one-wave-per-tile path,
        // instead of storing its raw output and normalising it in an elementwise pass (2 x 32 KiB per leaf less traffic)
        ConvArgs A{};
        const bool raw = c->cur_leaves != nullptr;
        A.in = raw ? c->cur_leaves : a["xr"], A.out = (c->debug || c->keep_y1) ? a["e_y1"] : nullptr, A.wfrag = w["e0.w"], A.bias_frag = w["e0.b"], A.n_tiles = nt, A.n_leaves = n;
        A.n_steps = c->nsteps["steps.rows8kd"], A.grp_start = od("steps.rows8kd"), A.part_s = ps, A.part_q = pq;
        const int psf = split_factor(g4, 8, 16, 1024);
        L.run("enc_conv_first_stats_s", [&] {
            if (raw) hipLaunchKernelGGL((conv_first_k<0, 0, true>), dim3(g4, psf), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
            else hipLaunchKernelGGL(conv_first_k<0>, dim3(g4, psf), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
        });
        combine("enc_stats_y1", 4, 1.0 / 2048.0, S.y1m, S.y1r);
        A.out = a["e_a1"], A.in_mean = S.y1m, A.in_rstd = S.y1r, A.in_gamma = w["eg0.w"], A.in_beta = w["eg0.b"];
        L.run("enc_conv_first_gn_s", [&] {   // (rolling row window: vq_first_roll.h)
            if (raw) hipLaunchKernelGGL((conv_first_roll_k<1, true>), dim3((2 * nt + 3) / 4, psf), dim3(256), 0, s, A);
            else if (c->first_roll) hipLaunchKernelGGL(conv_first_roll_k<1>, dim3((2 * nt + 3) / 4, psf), dim3(256), 0, s, A);
            else hipLaunchKernelGGL(conv_first_k<1>, dim3(g4, psf), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
        });
        combine("enc_stats_a1", 8, 1.0 / 1024.0, S.a1m, S.a1r);
    } else {
        ConvArgs A{};
        const bool raw = c->cur_leaves != nullptr;
        A.in = raw ? c->cur_leaves : a["xr"], A.out = a["e_y1"], A.wfrag = w["e0.w"], A.bias_frag = w["e0.b"], A.n_tiles = nt, A.n_leaves = n;
        A.n_steps = c->nsteps["steps.rows8kd"], A.grp_start = od("steps.rows8kd"), A.part_s = ps, A.part_q = pq;
        L.run("enc_conv_first_s", [&] {
            if (raw) hipLaunchKernelGGL((conv_first_k<2, 0, true>), dim3(g4, split_factor(g4, 8, 16, 1024)), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
            else hipLaunchKernelGGL(conv_first_k<2>, dim3(g4, split_factor(g4, 8, 16, 1024)), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
        });
        combine("enc_stats_y1", 4, 1.0 / 2048.0, S.y1m, S.y1r);
        ConvArgs B{};
        B.in = a["e_y1"], B.out = a["e_a1"], B.in_mean = S.y1m, B.in_rstd = S.y1r, B.in_gamma = w["eg0.w"], B.in_beta = w["eg0.b"];
        B.n_tiles = nt, B.part_s = ps, B.part_q = pq;
        L.run("enc_gn_relu_a1", [&] { hipLaunchKernelGGL((gn_relu_stats_k<16, 512, 4>), dim3(g4, split_factor(g4, 8, 16, 1024)), dim3(256), 0, s, B); });
        combine("enc_stats_a1", 8, 1.0 / 1024.0, S.a1m, S.a1r);
    }
    {
        ConvArgs A{};
        A.in = a["e_a1"], A.out = a["e_y4"], A.wfrag = w["r16c1.w"], A.bias_frag = w["r16c1.b"];
        A.in_mean = S.a1m, A.in_rstd = S.a1r, A.in_gamma = w["r16g1.w"], A.in_beta = w["r16g1.b"], A.n_tiles = nt;
        const int gq = (2 * nt + 3) / 4;
        // conv1 carries the statistics.  This tensor's statistics blocks are its 128 output half rows: 2 x 1024 doubles per leaf of
        // partials, which live in the region of conv2's output (e_a6: free until conv2 runs, after the combine)
        double* psr = reinterpret_cast<double*>(a["e_a6"]);
        double* pqr = psr + (size_t)nt * 128 * 8 * 32;
        // a handful of tiles take two-row groups (32 ranges, half the serial chain per wave; same taps in the same order) — both convs:
        // the statistics blocks are half rows, whatever the row grouping
        const bool two = gq * 32 <= 512;   // up to 1024 leaves (measured)
        const char* tab = two ? "steps.rowgroups8_2" : "steps.rowgroups8_4";
        A.n_steps = c->nsteps[tab], A.grp_start = od(tab), A.part_s = psr, A.part_q = pqr;
        L.run("enc_res16_conv1_s", [&] {
            if (two) hipLaunchKernelGGL((conv8_c16_k<2, false, true>), dim3(gq, 32), dim3(256), 0, s, A, (const int4*)w[tab]);
            else hipLaunchKernelGGL((conv8_c16_k<4, false, true>), dim3(gq, split_factor(gq, 8, 16, 1024)), dim3(256), 0, s, A, (const int4*)w[tab]);
        });
        L.run("enc_stats_y4", [&] { hipLaunchKernelGGL((gn_combine_k<false, true>), dim3(nt), dim3(8 * 32), 0, s, psr, pqr, S.y4m, S.y4r, 8, 1.0 / 1024.0); });
        A.part_s = nullptr, A.part_q = nullptr;
        A.in = a["e_y4"], A.out = a["e_a6"], A.wfrag = w["r16c2.w"], A.bias_frag = w["r16c2.b"], A.skip = a["e_a1"];
        A.in_mean = S.y4m, A.in_rstd = S.y4r, A.in_gamma = w["r16g2.w"], A.in_beta = w["r16g2.b"];
        L.run("enc_res16_conv2_s", [&] {
            if (two) hipLaunchKernelGGL((conv8_c16_k<2, true, false>), dim3(gq, 32), dim3(256), 0, s, A, (const int4*)w[tab]);
            else hipLaunchKernelGGL((conv8_c16_k<4, true, false>), dim3(gq, split_factor(gq, 8, 16, 1024)), dim3(256), 0, s, A, (const int4*)w[tab]);
        });
    }
    {
        ConvArgs A{};
        A.in = a["e_a6"], A.out = a["e_x7"], A.wfrag = w["ed.w16"], A.bias_frag = w["ed.braw"], A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rows_k4s2_8"], A.n_taps = 64, A.grp_start = od("steps.rows_k4s2_8"), A.part_s = ps, A.part_q = pq;
        const int psr = split_factor(gh, 4, 16, 512);
        L.run("enc_down_s", [&] {
            if (gh * psr * 2 <= 256) hipLaunchKernelGGL(k_enc_down_rs2, dim3(gh, psr, 2), dim3(512), LDS_ENC_DOWN_R / 2, s, A, (const int4*)w["steps.rows_k4s2_8"]);
            else hipLaunchKernelGGL(k_enc_down_rs, dim3(gh, psr), dim3(512), LDS_ENC_DOWN_R, s, A, (const int4*)w["steps.rows_k4s2_8"]);
        });
        combine("enc_stats_x7", 8, 1.0 / 256.0, S.x7m, S.x7r);
    }
    {
        ConvArgs A{};
        A.in = a["e_x7"], A.out = a["e_y9"], A.wfrag = w["r32c1.w16"], A.bias_frag = w["r32c1.braw"];
        A.in_mean = S.x7m, A.in_rstd = S.x7r, A.in_gamma = w["r32g1.w"], A.in_beta = w["r32g1.b"], A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27, A.grp_start = od("steps.rows_k3_4"), A.part_s = ps, A.part_q = pq;
        const int psr = split_factor(gh, 4, 16, 512);
        const bool ms = gh * psr * 2 <= 256;   // up to 1024 leaves (measured): also split the 32 couts over gridDim.z
        L.run("enc_res32_conv1_s", [&] {
            if (ms) hipLaunchKernelGGL(k_enc_r32c1_rs2, dim3(gh, psr, 2), dim3(512), LDS_ENC_R32R / 2, s, A, (const int4*)w["steps.rows_k3_4"]);
            else hipLaunchKernelGGL(k_enc_r32c1_rs, dim3(gh, psr), dim3(512), LDS_ENC_R32R, s, A, (const int4*)w["steps.rows_k3_4"]);
        });
        combine("enc_stats_y9", 8, 1.0 / 256.0, S.y9m, S.y9r);
        A.in = a["e_y9"], A.out = a["e_x11"], A.wfrag = w["r32c2.w16"], A.bias_frag = w["r32c2.braw"], A.skip = a["e_x7"];
        A.in_mean = S.y9m, A.in_rstd = S.y9r, A.in_gamma = w["r32g2.w"], A.in_beta = w["r32g2.b"];
        A.part_s = nullptr, A.part_q = nullptr, A.part_c = a["part_c"];
        L.run("enc_res32_conv2_s", [&] {
            if (ms) hipLaunchKernelGGL(k_enc_r32c2_rs2, dim3(gh, psr, 2), dim3(512), LDS_ENC_R32R / 2, s, A, (const int4*)w["steps.rows_k3_4"]);
            else hipLaunchKernelGGL(k_enc_r32c2_rs, dim3(gh, psr), dim3(512), LDS_ENC_R32R, s, A, (const int4*)w["steps.rows_k3_4"]);
        });
        L.run("enc_csum_x11", [&] { hipLaunchKernelGGL((csum_combine_k<32>), dim3(nt), dim3(256), 0, s, a["part_c"], S.csum, w["efc0"], w["efc2"], a["gate"]); });
    }
    if (d_latent) {
        launch_latent_assign(c, L, n, d_idx, d_latent, s, split_factor(g2, 8, 32, 512));
        return L.rc;
    }
    VqArgs A{};
    A.in = a["e_x11"], A.se_csum = S.csum, A.se_fc0 = w["efc0"], A.se_fc2 = w["efc2"];
    A.epfrag = w["vq.ep"], A.ck_frag = w["vq.ck"], A.idx = d_idx, A.n_leaves = n, A.n_tiles = nt;
    A.se_gate = a["gate"];   // computed once per tile by enc_csum_x11
    L.run("enc_vq_s", [&] {
        // mid-size batches: 8 tiles behind one copy of the folded codebook in LDS (33 KB) instead of 2
        if (nt >= 256) hipLaunchKernelGGL(vq_folded_k<8>, dim3((nt + 7) / 8, split_factor((nt + 7) / 8, 2, 32, 512)), dim3(512), 0, s, A);
        else hipLaunchKernelGGL(vq_folded_k<2>, dim3(g2, split_factor(g2, 8, 32, 2048)), dim3(128), 0, s, A);
    });
    return L.rc;
}

int encode_chunk(vqhip_codec* c, const float* d_leaves, int64_t n, uint8_t* d_idx, hipStream_t s, float* d_latent = nullptr)
{
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    const int nt = (int)((n + 31) / 32);
    auto& a = c->act;
    auto& w = c->dw;
    Launcher L{c, s, n};
    const int g4 = (nt + 3) / 4, g8 = (nt + 7) / 8;
    EncStats S = enc_stats_of(c);

    // the position-major copy xt is only read by the training step (loss, first-conv weight gradients) and by debug fetches
    float* xt = (c->training || c->full_training || c->debug) ? a["xt"] : nullptr;
    // raw: the first conv reads d_leaves itself; the row layout is still written when something else reads it (training, debug fetches, VQHIP_FIRST=steps' normalising pass)
    const bool raw = c->first_raw && c->first_roll && !c->training && !c->full_training;
    c->cur_leaves = raw ? d_leaves : nullptr;
    if (!raw || xt) L.run("pack_leaves", [&] { hipLaunchKernelGGL(pack_leaves_k, dim3(nt, nt <= 128 ? 8 : 1), dim3(256), 0, s, d_leaves, a["xr"], xt, n); });
    if (use_split(c, nt, false)) return encode_chunk_split(c, L, n, d_idx, s, d_latent);
    {
        // first conv twice: statistics pass, then recompute + GroupNorm(4,16) + ReLU + statistics for res.gn1
        ConvArgs A{};
        A.in = raw ? d_leaves : a["xr"], A.out = (c->debug || c->keep_y1) ? a["e_y1"] : nullptr, A.wfrag = w["e0.w"], A.bias_frag = w["e0.b"];
        A.out_mean = S.y1m, A.out_rstd = S.y1r, A.n_tiles = nt, A.n_steps = c->nsteps["steps.rows8kd"], A.n_leaves = n;
        // first_roll: a wave per 16-leaf sub-tile with a rolling 3 x 3 row window in registers (three row loads per output row instead of nine)
        const int gq = (2 * nt + 3) / 4;
        // (the normalising pass only: its stores compete with the loads — 0.458 -> 0.393 ms; the statistics pass runs 0.322 ms on the
        // (row, kd) kernel and 0.353 ms with the window, whose plane starts it cannot hide; VQHIP_FIRST=roll0 selects that as well)
        L.run("enc_conv_first_stats", [&] {
            if (raw && c->first_roll_stats) hipLaunchKernelGGL((conv_first_roll_k<0, true>), dim3(gq), dim3(256), 0, s, A);
            else if (raw) hipLaunchKernelGGL((conv_first_k<0, 0, true>), dim3(g4), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
            else if (c->first_roll_stats) hipLaunchKernelGGL(conv_first_roll_k<0>, dim3(gq), dim3(256), 0, s, A);
            else hipLaunchKernelGGL(conv_first_k<0>, dim3(g4), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
        });
        A.out = a["e_a1"], A.in_mean = S.y1m, A.in_rstd = S.y1r, A.in_gamma = w["eg0.w"], A.in_beta = w["eg0.b"];
        A.out_mean = S.a1m, A.out_rstd = S.a1r;
        L.run("enc_conv_first_gn", [&] {
            if (raw) hipLaunchKernelGGL((conv_first_roll_k<1, true>), dim3(gq), dim3(256), 0, s, A);
            else if (c->first_roll) hipLaunchKernelGGL(conv_first_roll_k<1>, dim3(gq), dim3(256), 0, s, A);
            else hipLaunchKernelGGL(conv_first_k<1>, dim3(g4), dim3(256), 0, s, A, (const int4*)w["steps.rows8kd"]);
        });
    }
    {
        ConvArgs A{};
        A.in = a["e_a1"], A.out = a["e_y4"], A.wfrag = w["r16c1.w"], A.bias_frag = w["r16c1.b"];
        A.in_mean = S.a1m, A.in_rstd = S.a1r, A.in_gamma = w["r16g1.w"], A.in_beta = w["r16g1.b"];
        A.out_mean = S.y4m, A.out_rstd = S.y4r, A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rowgroups8_4"];
        if (c->conv8_lds) {   // input planes staged in LDS by a persistent workgroup per CU; statistics as 16 half-row totals + combine
            A.part_s = reinterpret_cast<double*>(a["part_s"]), A.part_q = reinterpret_cast<double*>(a["part_q"]);
            if (c->conv8_w16) L.run("enc_res16_conv1", [&] { hipLaunchKernelGGL((conv8_lds_k<false, true, 16, 0, false>), dim3(std::min(2 * nt, c->n_cus)), dim3(1024), LDS_CONV8, s, A); });
            else L.run("enc_res16_conv1", [&] { hipLaunchKernelGGL((conv8_lds_k<false, true, 8, 0, true>), dim3(std::min(2 * nt, c->n_cus)), dim3(512), LDS_CONV8, s, A); });
            L.run("enc_stats_y4", [&] { hipLaunchKernelGGL((gn_combine_k<false>), dim3(nt), dim3(8 * 32), 0, s, A.part_s, A.part_q, S.y4m, S.y4r, 8, 1.0 / 1024.0); });
        } else {   // row-group kernel: one partial per output half row (in the region of conv2's output, free until then), added row-major by the combine
            A.part_s = reinterpret_cast<double*>(a["e_a6"]), A.part_q = A.part_s + (size_t)nt * 128 * 8 * 32;
            L.run("enc_res16_conv1", [&] { hipLaunchKernelGGL((conv8_c16_k<4, false, true>), dim3((2 * nt + 3) / 4), dim3(256), 0, s, A, (const int4*)w["steps.rowgroups8_4"]); });
            L.run("enc_stats_y4", [&] { hipLaunchKernelGGL((gn_combine_k<false, true>), dim3(nt), dim3(8 * 32), 0, s, A.part_s, A.part_q, S.y4m, S.y4r, 8, 1.0 / 1024.0); });
        }
    }
    {
        ConvArgs A{};
        A.in = a["e_y4"], A.out = a["e_a6"], A.wfrag = w["r16c2.w"], A.bias_frag = w["r16c2.b"], A.skip = a["e_a1"];
        A.in_mean = S.y4m, A.in_rstd = S.y4r, A.in_gamma = w["r16g2.w"], A.in_beta = w["r16g2.b"], A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rowgroups8_4"];
        if (c->conv8_lds && c->conv8_w16) L.run("enc_res16_conv2", [&] { hipLaunchKernelGGL((conv8_lds_k<true, false, 16, 0, false>), dim3(std::min(2 * nt, c->n_cus)), dim3(1024), LDS_CONV8, s, A); });
        else if (c->conv8_lds) L.run("enc_res16_conv2", [&] { hipLaunchKernelGGL((conv8_lds_k<true, false, 8, 0, true>), dim3(std::min(2 * nt, c->n_cus)), dim3(512), LDS_CONV8, s, A); });
        else L.run("enc_res16_conv2", [&] { hipLaunchKernelGGL((conv8_c16_k<4, true, false>), dim3((2 * nt + 3) / 4), dim3(256), 0, s, A, (const int4*)w["steps.rowgroups8_4"]); });
    }
    {
        ConvArgs A{};
        A.in = a["e_a6"], A.out = a["e_x7"], A.wfrag = w["ed.w16"], A.bias_frag = w["ed.braw"];
        A.out_mean = S.x7m, A.out_rstd = S.x7r, A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rows_k4s2_8"], A.n_taps = 64;
        if (c->convdown_lds) L.run("enc_down", [&] { hipLaunchKernelGGL(conv_down_lds_k<0>, dim3(std::min(2 * nt, c->n_cus)), dim3(1024), LDS_CONVDOWN, s, A); });
        else L.run("enc_down", [&] { hipLaunchKernelGGL(k_enc_down_r, dim3((2 * nt + 7) / 8), dim3(512), LDS_ENC_DOWN_R, s, A, (const int4*)w["steps.rows_k4s2_8"]); });
    }
    {
        ConvArgs A{};
        A.in = a["e_x7"], A.out = a["e_y9"], A.wfrag = w["r32c1.w16"], A.bias_frag = w["r32c1.braw"];
        A.in_mean = S.x7m, A.in_rstd = S.x7r, A.in_gamma = w["r32g1.w"], A.in_beta = w["r32g1.b"];
        A.out_mean = S.y9m, A.out_rstd = S.y9r, A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27;
        if (c->conv4_lds) L.run("enc_res32_conv1", [&] { hipLaunchKernelGGL((conv4_lds_k<false, true, false, 0, 1>), dim3(std::min(2 * nt, c->n_cus)), dim3(512), LDS_CONV4, s, A); });
        else L.run("enc_res32_conv1", [&] { hipLaunchKernelGGL(k_enc_r32c1_r, dim3((2 * nt + 15) / 16), dim3(1024), LDS_ENC_R32R, s, A, (const int4*)w["steps.rows_k3_4"]); });
    }
    {
        ConvArgs A{};
        A.in = a["e_y9"], A.out = a["e_x11"], A.wfrag = w["r32c2.w16"], A.bias_frag = w["r32c2.braw"], A.skip = a["e_x7"];
        A.in_mean = S.y9m, A.in_rstd = S.y9r, A.in_gamma = w["r32g2.w"], A.in_beta = w["r32g2.b"];
        A.out_csum = S.csum, A.n_tiles = nt;
        A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27;
        if (c->conv4_lds) L.run("enc_res32_conv2", [&] { hipLaunchKernelGGL((conv4_lds_k<true, false, true, 0, 0>), dim3(std::min(2 * nt, c->n_cus)), dim3(512), LDS_CONV4, s, A); });
        else L.run("enc_res32_conv2", [&] { hipLaunchKernelGGL(k_enc_r32c2_r, dim3((2 * nt + 15) / 16), dim3(1024), LDS_ENC_R32R, s, A, (const int4*)w["steps.rows_k3_4"]); });
    }

    if (d_latent) {
        launch_latent_assign(c, L, n, d_idx, d_latent, s, 1);
        return L.rc;
    }
    {
        VqArgs A{};
        A.in = a["e_x11"], A.se_csum = S.csum, A.se_fc0 = w["efc0"], A.se_fc2 = w["efc2"];
        A.epfrag = w["vq.ep"], A.ck_frag = w["vq.ck"], A.idx = d_idx, A.n_leaves = n, A.n_tiles = nt;
        // two position ranges per tile: 2 x n_tiles waves = 4 per SIMD on a full chunk (one wave per tile leaves 2, and the wave's
        // MFMA chain -> argmin scan -> next chain sequence has nobody to overlap with)
        const int vq_split = c->vq_split;
        L.run("enc_vq", [&] { hipLaunchKernelGGL(vq_folded_k<8>, dim3(g8, vq_split), dim3(512), 0, s, A); });
    }
    return L.rc;
}

// position-split decode for small batches (see encode_chunk_split)
int decode_chunk_split(vqhip_codec* c, Launcher& L, const uint8_t* d_idx, int64_t n, float* d_out, hipStream_t s)
{
    const int nt = (int)((n + 31) / 32);
    auto& a = c->act;
    auto& w = c->dw;
    auto od = [&](const char* name) { return reinterpret_cast<const int*>(w[std::string(name) + ".grp"]); };
    const int g4 = (nt + 3) / 4, g2 = (nt + 1) / 2;
    double* ps = reinterpret_cast<double*>(a["part_s"]);
    double* pq = reinterpret_cast<double*>(a["part_q"]);
    // fused statistics of split launches: per-block partials + gn_combine_k (16 slots = the 4-channel halves of GroupNorm(8,64))
    auto combine64 = [&](const char* name, float* mean, float* rstd) {
        L.run(name, [&] { hipLaunchKernelGGL(gn_combine_k<true>, dim3(nt), dim3(512), 0, s, ps, pq, mean, rstd, 8, 1.0 / 512.0); });
    };
    if (c->stem_fused && c->stem_taps && nt >= c->n_cus) {
        // mid-size passes (a tile per CU and more): the decoder front as ONE kernel (table through the LDS ring, stem_taps_k) instead of
        // gather + combine + normalise + combine; same d2 and statistics bit for bit
        StemFusedArgs F{};
        F.idx = d_idx, F.T = w["ds.lut"], F.bias = w["ds.b"], F.gamma = w["dg0.w"], F.beta = w["dg0.b"], F.d2 = a["d_d2"];
        F.out_mean = a["st_b.mean"], F.out_rstd = a["st_b.rstd"], F.ystem_dbg = c->debug ? a["d_ystem"] : nullptr;
        F.n_leaves = n, F.n_tiles = nt;
        L.run("dec_stem_gn_s", [&] { hipLaunchKernelGGL(k_stem_taps, dim3(std::min(nt, c->n_cus)), dim3(512), LDS_STEM_TAPS, s, F); });
    } else if (c->stem_fused && getenv("VQHIP_STEM_SMALL") == nullptr) {
        // SOP-sized passes: the decoder front as ONE kernel too — a workgroup per four leaves gathers its table rows through the L1,
        // normalises out of LDS and finishes both sets of statistics itself (stem_fused_k) — instead of gather + combine + normalise +
        // combine (four launches of 6-23 us each at 64 leaves); same d2 and statistics bit for bit
        StemFusedArgs F{};
        F.idx = d_idx, F.T = w["ds.lut"], F.bias = w["ds.b"], F.gamma = w["dg0.w"], F.beta = w["dg0.b"], F.d2 = a["d_d2"];
        F.out_mean = a["st_b.mean"], F.out_rstd = a["st_b.rstd"], F.ystem_dbg = c->debug ? a["d_ystem"] : nullptr;
        F.steps = (const int4*)w["steps.k3s1_4"], F.grp_start = reinterpret_cast<const int*>(w["steps.k3s1_4.grp"]), F.n_steps = c->nsteps["steps.k3s1_4"];
        F.n_leaves = n, F.n_tiles = nt;
        L.run("dec_stem_gn_s", [&] {
            if (nt <= 64) hipLaunchKernelGGL(stem_fused_k<16>, dim3(8 * nt), dim3(1024), 0, s, F);   // up to 2048 leaves: sixteen waves, half the gather chain each
            else hipLaunchKernelGGL(stem_fused_k<8>, dim3(8 * nt), dim3(512), 0, s, F);
        });
    } else {
        L.run("dec_stem_s", [&] {
            hipLaunchKernelGGL(stem_lut_k, dim3(2 * nt, split_factor(2 * nt, 2, 16, 2048)), dim3(256), 0, s, d_idx, w["ds.lut"], w["ds.b"], a["d_ystem"], (float*)nullptr, (float*)nullptr,
                               (const int4*)w["steps.k3s1_4"], c->nsteps["steps.k3s1_4"], n, nt, od("steps.k3s1_4"), ps, pq);
        });
        combine64("dec_stats_ystem", a["st_a.mean"], a["st_a.rstd"]);
        {
            ConvArgs A{};
            A.in = a["d_ystem"], A.out = a["d_d2"], A.in_mean = a["st_a.mean"], A.in_rstd = a["st_a.rstd"];
            A.in_gamma = w["dg0.w"], A.in_beta = w["dg0.b"], A.n_tiles = nt, A.part_s = ps, A.part_q = pq;
            L.run("dec_gn_relu_d2", [&] { hipLaunchKernelGGL((gn_relu_stats_k<64, 64, 8>), dim3(g4, split_factor(g4, 2, 16, 1024)), dim3(256), 0, s, A); });
        }
        combine64("dec_stats_d2", a["st_b.mean"], a["st_b.rstd"]);
    }
    {
        ConvArgs A{};
        A.in = a["d_d2"], A.out = a["d_y4"], A.wfrag = w["r64c1.w"], A.bias_frag = w["r64c1.b"];
        A.in_mean = a["st_b.mean"], A.in_rstd = a["st_b.rstd"], A.in_gamma = w["r64g1.w"], A.in_beta = w["r64g1.b"], A.n_tiles = nt;
        A.wfrag = w["r64c1.w16"], A.bias_frag = w["r64c1.braw"];
        A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27, A.grp_start = od("steps.rows_k3_4");
        A.part_s = ps, A.part_q = pq;
        const int gh = (2 * nt + 7) / 8, psr = split_factor(gh, 4, 16, 512);   // 8 half tiles per workgroup, 16 output rows to split
        const bool ms = gh * psr * 4 <= 1024;   // up to 2048 leaves (measured): also split the 64 couts over gridDim.z
        const bool r64res = c->r64s_resident;
        L.run("dec_res64_conv1_s", [&] {
            if (ms && r64res) hipLaunchKernelGGL(k_dec_r64c1_rs4r, dim3(gh, psr, 4), dim3(512), LDS_DEC_R64S4R, s, A, (const int4*)w["steps.rows_k3_4"]);
            else if (ms) hipLaunchKernelGGL(k_dec_r64c1_rs4, dim3(gh, psr, 4), dim3(512), LDS_DEC_R64R / 4, s, A, (const int4*)w["steps.rows_k3_4"]);
            else hipLaunchKernelGGL(k_dec_r64c1_rp, dim3(gh, psr), dim3(512), LDS_DEC_R64R, s, A, (const int4*)w["steps.rows_k3_4"]);
        });
        combine64("dec_stats_y4", a["st_a.mean"], a["st_a.rstd"]);
        A.in = a["d_y4"], A.out = a["d_x6"], A.wfrag = w["r64c2.w16"], A.bias_frag = w["r64c2.braw"], A.skip = a["d_d2"];
        A.in_mean = a["st_a.mean"], A.in_rstd = a["st_a.rstd"], A.in_gamma = w["r64g2.w"], A.in_beta = w["r64g2.b"];
        A.part_s = nullptr, A.part_q = nullptr, A.part_c = a["part_c"];
        L.run("dec_res64_conv2_s", [&] {
            if (ms && r64res) hipLaunchKernelGGL(k_dec_r64c2_rs4r, dim3(gh, psr, 4), dim3(512), LDS_DEC_R64S4R, s, A, (const int4*)w["steps.rows_k3_4"]);
            else if (ms) hipLaunchKernelGGL(k_dec_r64c2_rs4, dim3(gh, psr, 4), dim3(512), LDS_DEC_R64R / 4, s, A, (const int4*)w["steps.rows_k3_4"]);
            else hipLaunchKernelGGL(k_dec_r64c2_rp, dim3(gh, psr), dim3(512), LDS_DEC_R64R, s, A, (const int4*)w["steps.rows_k3_4"]);
        });
        L.run("dec_csum_x6", [&] { hipLaunchKernelGGL((csum_combine_k<64>), dim3(nt), dim3(512), 0, s, a["part_c"], a["csum"], w["dfc0"], w["dfc2"], a["gate"]); });
    }
    {
        ConvArgs A{};
        A.in = a["d_x6"], A.out = d_out, A.wfrag = w["tail.w"], A.bias_frag = w["tail.b"];
        A.se_csum = a["csum"], A.se_fc0 = w["dfc0"], A.se_fc2 = w["dfc2"], A.n_tiles = nt, A.n_leaves = n;
        A.n_steps = c->nsteps["steps.tail"], A.n_taps = 0, A.grp_start = od("steps.tail");
        A.se_gate = a["gate"];   // computed once per tile by dec_csum_x6
        // a few tiles: sixteen waves per (tile, slab) on the 16x16x4 MFMA (a quarter of the serial chain per wave)
        const int t16 = c->tail16_tiles;
        if (nt <= t16) {
            A.wfrag = w["tail.w16"], A.bias_frag = w["tail.braw"];
            L.run("dec_tail_s", [&] { hipLaunchKernelGGL(tail_small16_k<4>, dim3(nt, 4, 4), dim3(256), 0, s, A); });
        } else {
            L.run("dec_tail_s", [&] { hipLaunchKernelGGL(tail_small_k<false>, dim3(nt, 4), dim3(256), 0, s, A); });
        }
    }
    return L.rc;
}

int decode_chunk(vqhip_codec* c, const uint8_t* d_idx, int64_t n, float* d_out, hipStream_t s)
{
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    const int nt = (int)((n + 31) / 32);
    auto& a = c->act;
    auto& w = c->dw;
    Launcher L{c, s, n};
    const int g4 = (nt + 3) / 4, g8 = (nt + 7) / 8;

    if (use_split(c, nt, true)) return decode_chunk_split(c, L, d_idx, n, d_out, s);
    if (c->stem_fused) {   // gather + stem + GroupNorm + ReLU + statistics in one kernel: the stem output stays in LDS
        StemFusedArgs F{};
        F.idx = d_idx, F.T = w["ds.lut"], F.bias = w["ds.b"], F.gamma = w["dg0.w"], F.beta = w["dg0.b"], F.d2 = a["d_d2"];
        F.out_mean = a["st_b.mean"], F.out_rstd = a["st_b.rstd"], F.ystem_dbg = c->debug ? a["d_ystem"] : nullptr;
        F.steps = (const int4*)w["steps.k3s1_4"], F.grp_start = reinterpret_cast<const int*>(w["steps.k3s1_4.grp"]), F.n_steps = c->nsteps["steps.k3s1_4"];
        F.n_leaves = n, F.n_tiles = nt;
        if (c->stem_taps) L.run("dec_stem_gn", [&] { hipLaunchKernelGGL(k_stem_taps, dim3(std::min(nt, c->n_cus)), dim3(512), LDS_STEM_TAPS, s, F); });
        else L.run("dec_stem_gn", [&] { hipLaunchKernelGGL(stem_fused_k<8>, dim3(8 * nt), dim3(512), 0, s, F); });
    } else {
        L.run("dec_stem", [&] {
            hipLaunchKernelGGL(stem_lut_k, dim3(2 * nt), dim3(256), 0, s, d_idx, w["ds.lut"], w["ds.b"], a["d_ystem"], a["st_a.mean"], a["st_a.rstd"],
                               (const int4*)w["steps.k3s1_4"], c->nsteps["steps.k3s1_4"], n, nt, (const int*)nullptr);
        });
        {
            ConvArgs A{};
            A.in = a["d_ystem"], A.out = a["d_d2"], A.in_mean = a["st_a.mean"], A.in_rstd = a["st_a.rstd"];
            A.in_gamma = w["dg0.w"], A.in_beta = w["dg0.b"], A.out_mean = a["st_b.mean"], A.out_rstd = a["st_b.rstd"], A.n_tiles = nt;
            L.run("dec_gn_relu_stats", [&] { hipLaunchKernelGGL((gn_relu_stats_k<64, 64, 8>), dim3(g4), dim3(256), 0, s, A); });
        }
    }
    {
        ConvArgs A{};
        A.in = a["d_d2"], A.out = a["d_y4"], A.wfrag = w["r64c1.w"], A.bias_frag = w["r64c1.b"];
        A.in_mean = a["st_b.mean"], A.in_rstd = a["st_b.rstd"], A.in_gamma = w["r64g1.w"], A.in_beta = w["r64g1.b"];
        A.out_mean = a["st_a.mean"], A.out_rstd = a["st_a.rstd"], A.n_tiles = nt;
        A.wfrag = w["r64c1.w16"], A.bias_frag = w["r64c1.braw"], A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27;
        L.run("dec_res64_conv1", [&] { hipLaunchKernelGGL(k_dec_r64c1_r, dim3((2 * nt + 7) / 8), dim3(512), LDS_DEC_R64R, s, A, (const int4*)w["steps.rows_k3_4"]); });
    }
    {
        ConvArgs A{};
        A.in = a["d_y4"], A.out = a["d_x6"], A.wfrag = w["r64c2.w"], A.bias_frag = w["r64c2.b"], A.skip = a["d_d2"];
        A.in_mean = a["st_a.mean"], A.in_rstd = a["st_a.rstd"], A.in_gamma = w["r64g2.w"], A.in_beta = w["r64g2.b"];
        A.out_csum = a["csum"], A.n_tiles = nt;
        A.wfrag = w["r64c2.w16"], A.bias_frag = w["r64c2.braw"], A.n_steps = c->nsteps["steps.rows_k3_4"], A.n_taps = 27;
        L.run("dec_res64_conv2", [&] { hipLaunchKernelGGL(k_dec_r64c2_r, dim3((2 * nt + 7) / 8), dim3(512), LDS_DEC_R64R, s, A, (const int4*)w["steps.rows_k3_4"]); });
    }
    {
        ConvArgs A{};
        A.in = a["d_x6"], A.out = d_out, A.wfrag = w["tail.w"], A.bias_frag = w["tail.b"];
        A.se_csum = a["csum"], A.se_fc0 = w["dfc0"], A.se_fc2 = w["dfc2"], A.n_tiles = nt, A.n_leaves = n;
        A.n_steps = c->nsteps["steps.tail"], A.n_taps = 0;
        if (c->tail_rows) {
            A.wfrag = w["tail.wrows"], A.bias_frag = w["tail.braw"];
            if (c->tail_groups) {
                A.wfrag = w["tail.wgroups"];
                L.run("dec_tail_groups", [&] { hipLaunchKernelGGL(tail_groups16_k<0>, dim3((2 * nt + 7) / 8), dim3(512), LDS_TAIL_GROUPS, s, A); });
            } else if (c->tail_rows32) L.run("dec_tail_rows32", [&] { hipLaunchKernelGGL(tail_rows32_k<0>, dim3((nt + 3) / 4), dim3(256), LDS_TAIL_ROWS, s, A); });
            else L.run("dec_tail", [&] { hipLaunchKernelGGL(tail_rows16_k<0>, dim3((2 * nt + 7) / 8), dim3(512), LDS_TAIL_ROWS, s, A); });
        } else {
            L.run("dec_tail_slab", [&] { hipLaunchKernelGGL(k_dec_tail, dim3(g8), dim3(512), LDS_DEC_TAIL, s, A, (const int4*)w["steps.tail"]); });
        }
    }
    return L.rc;
}

int ensure_io(vqhip_codec* c, int64_t n)
{
    if (!c->s_in) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    if (n <= c->dev_io_leaves) return VQHIP_OK;
    for (int i = 0; i < 2; ++i) {
        if (c->dev_leaves[i]) hipFree(c->dev_leaves[i]);
        if (c->dev_idx[i]) hipFree(c->dev_idx[i]);
        if (c->pin_out[i]) hipHostFree(c->pin_out[i]);
        c->dev_leaves[i] = nullptr, c->dev_idx[i] = nullptr, c->pin_out[i] = nullptr;
    }
    c->dev_io_leaves = 0;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipMalloc(&c->dev_leaves[i], (size_t)n * 512 * sizeof(float)));
        HIPCHK(c, hipMalloc(&c->dev_idx[i], (size_t)n * 64));
        HIPCHK(c, hipHostMalloc(&c->pin_out[i], (size_t)n * 512 * sizeof(float), hipHostMallocDefault));
    }
    c->dev_io_leaves = n;
    return VQHIP_OK;
}

// pinned input staging (leaf blocks: gathered leaf buffers or the caller's pageable block; index blocks of single-chunk decodes)
int ensure_stage(vqhip_codec* c, size_t bytes)
{
    if (c->pin_in_bytes >= bytes) return VQHIP_OK;
    for (int i = 0; i < 2; ++i) {
        if (c->pin_in[i]) hipHostFree(c->pin_in[i]);
        c->pin_in[i] = nullptr;
    }
    c->pin_in_bytes = 0;
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipHostMalloc(&c->pin_in[i], bytes, hipHostMallocDefault));
    c->pin_in_bytes = bytes;
    return VQHIP_OK;
}

// Cap on the copy threads one calling thread may fan out to (host_parallel_for).  Default 16; the in-process multi-device front
// end lowers it per device worker so that G devices never run more than min(cores, 64) copy threads together; the environment
// variable VQHIP_COPY_THREADS overrides both.
thread_local int g_copy_threads_cap = 16;

int copy_threads_cap()
{
    static const int env = [] {
        const char* e = std::getenv("VQHIP_COPY_THREADS");
        return e ? std::max(1, std::atoi(e)) : 0;
    }();
    return env ? env : g_copy_threads_cap;
}

// parallel-for over [0,n) on host threads (the library's stand-in for the orchestrator's tbb::parallel_for)
template <typename F>
void host_parallel_for(int64_t n, F&& f, int64_t grain = 2048)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw / 2, (int64_t)copy_threads_cap(), n / grain}));
    if (nt <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) th.emplace_back([=, &f] { f(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    for (auto& x : th) x.join();
}

// multi-threaded memcpy for the large host-side block copies (pageable <-> pinned staging): one thread tops out
// near 10 GB/s, below both the PCIe link and the kernels' leaf rate
void host_parallel_copy(void* dst, const void* src, size_t bytes)
{
    constexpr size_t BLK = size_t(1) << 20;
    const int64_t nb = (int64_t)((bytes + BLK - 1) / BLK);
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    host_parallel_for(nb, [=](int64_t a, int64_t b) {
        const size_t lo = (size_t)a * BLK, hi = std::min(bytes, (size_t)b * BLK);
        if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
    }, 4);
}

int refresh_tables(vqhip_codec* c)
{
    std::vector<float> E(256 * 128);
    HIPCHK(c, hipDeviceSynchronize());  // the codebook may have been updated on a caller's stream
    HIPCHK(c, hipMemcpy(E.data(), c->dw["cb"], E.size() * sizeof(float), hipMemcpyDeviceToHost));
    int rc;
    if (c->weights_stale) {
        // full training changed the encoder / decoder weights: the device fragments were rebuilt after the optimizer step;
        // what is folded on the host in fp64 (decoder tail, projection inside the VQ search) is rebuilt here from the parameters
        HIPCHK(c, hipMemcpy(c->h_params.data(), c->ft_P, c->h_params.size() * sizeof(float), hipMemcpyDeviceToHost));
        auto hp = [&](const char* n) { return c->h_params.data() + c->p_off.at(n).first; };
        c->h_proj_w.assign(hp("encoder.proj.weight"), hp("encoder.proj.weight") + 128 * 32);
        c->h_proj_b.assign(hp("encoder.proj.bias"), hp("encoder.proj.bias") + 128);
        if ((rc = build_folded_tail(c, hp("decoder.up_conv.weight"), hp("decoder.up_conv.bias"), hp("decoder.final.weight"), hp("decoder.final.bias")))) return rc;
        c->weights_stale = false;
    }
    if ((rc = build_vq_fold(c, E.data()))) return rc;
    hipLaunchKernelGGL(build_stem_lut_k, dim3(27 * 256), dim3(64), 0, c->stream, c->dw["ds.w"], c->dw["cb"], c->dw["ds.lut"]);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tables_stale = false;
    return VQHIP_OK;
}

// inference entry points after codebook training: the folded VQ tables and the decoder stem table follow the live codebook
int ensure_tables(vqhip_codec* c)
{
    return c->tables_stale ? refresh_tables(c) : VQHIP_OK;
}

// Host-side pipeline shared by every entry point that takes host memory (the path the reference orchestrator
// calls, VQVAECodec.cpp:120,178).  Chunk i: produce() -> H2D on s_in -> kernels on the compute stream -> D2H into
// pinned memory on s_out; consume() of chunk i-1 runs on the calling thread while the GPU works on chunk i.
// encode: in = leaves (2048 B/leaf), out = indices (64 B/leaf); decode: the reverse.
//   produce(o, m, stage): returns the host address of chunk [o, o+m)'s input; `stage` is this slot's pinned
//                         buffer (nullptr unless want_stage) which produce may fill and return.
//   consume(o, m, result): result = pinned buffer holding the chunk's output.
using ProduceFn = std::function<const void*(int64_t, int64_t, void*)>;
using ConsumeFn = std::function<int(int64_t, int64_t, const void*)>;

int run_pipeline(vqhip_codec* c, bool is_encode, int64_t n, int64_t step, bool want_stage, const ProduceFn& produce, const ConsumeFn& consume)
{
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->chunk_fitted) fit_chunk_to_free_memory(c), c->chunk_fitted = true;
    step = std::min(step > 0 ? std::min(step, c->chunk) : c->chunk, n);
    // A call that fits ONE chunk but is large (a 65 536-leaf batch from the orchestrator loop or the leaf-pointer entry points) would run
    // gather -> H2D -> kernels -> D2H -> scatter back to back (17.5 ms for 65 536 leaves against a 9.8 ms device pass): it is cut into
    // host_split pieces so that the host phases of one piece overlap the device pass of its neighbours.  Results never depend on the cut.
    if (n <= step && c->host_split > 1 && n >= 2 * c->host_split_min && !c->debug) {   // (debug mode keeps a call's activations whole for vqhip_debug_fetch)
        const int64_t pieces = std::min<int64_t>(c->host_split, n / c->host_split_min);
        step = ((n + pieces - 1) / pieces + 31) / 32 * 32;
    }
    int rc = ensure_tables(c);
    if (!rc) rc = ensure_io(c, step);
    if (rc) return rc;
    const size_t in_b = is_encode ? 2048 : 64, out_b = is_encode ? 64 : 2048;
    if (want_stage && (rc = ensure_stage(c, (size_t)step * in_b))) return rc;
    if (n <= step) {
        // One chunk: there is nothing to overlap — H2D, kernels and D2H go down the compute stream in order and the call waits once.
        // (The three-stream form below costs such a call three cross-stream event hand-overs and as many extra API calls: the SOP's
        // default batch of 64 leaves is a 0.14-0.20 ms pass, so they showed: 0.26 / 0.21 ms per call through the adapter.)
        void* d_in = is_encode ? (void*)c->dev_leaves[0] : (void*)c->dev_idx[0];
        void* d_out = is_encode ? (void*)c->dev_idx[0] : (void*)c->dev_leaves[0];
        const void* src = produce(0, n, want_stage ? c->pin_in[0] : nullptr);
        if (!src) return c->err.empty() ? fail(c, VQHIP_ERR_INVALID, "input source failed") : VQHIP_ERR_INVALID;
        auto sync_fail = [&](int code) {   // leave no copy in flight that references caller or slot memory
            hipStreamSynchronize(c->stream);
            return code;
        };
        hipError_t e = hipMemcpyAsync(d_in, src, (size_t)n * in_b, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return sync_fail(fail(c, VQHIP_ERR_DEVICE, std::string("hipMemcpyAsync H2D: ") + hipGetErrorString(e)));
        rc = is_encode ? encode_chunk(c, c->dev_leaves[0], n, c->dev_idx[0], c->stream) : decode_chunk(c, c->dev_idx[0], n, c->dev_leaves[0], c->stream);
        if (rc) return sync_fail(rc);
        e = hipMemcpyAsync(c->pin_out[0], d_out, (size_t)n * out_b, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return sync_fail(fail(c, VQHIP_ERR_DEVICE, std::string("hipMemcpyAsync D2H: ") + hipGetErrorString(e)));
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, VQHIP_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        return consume(0, n, c->pin_out[0]);
    }
    int64_t prev_off = -1, prev_m = 0;
    int prev_slot = 0, i = 0;
    auto drain = [&]() -> int {
        if (prev_off < 0) return VQHIP_OK;
        HIPCHK(c, hipEventSynchronize(c->ev_out[prev_slot]));
        const int64_t o = prev_off;
        prev_off = -1;
        return consume(o, prev_m, c->pin_out[prev_slot]);
    };
    auto abort_run = [&](int code) {  // leave no work in flight that still references caller memory
        hipStreamSynchronize(c->s_in);
        hipStreamSynchronize(c->stream);
        hipStreamSynchronize(c->s_out);
        return code;
    };
    // inside the loop every error exit goes through abort_run: async copies may still reference caller / slot memory
#define PIPECHK(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) {                                              \
            c->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return abort_run(VQHIP_ERR_DEVICE);                              \
        }                                                                    \
    } while (0)
    for (int64_t o = 0; o < n; o += step, ++i) {
        const int64_t m = std::min(step, n - o);
        const int slot = i & 1;
        void* d_in = is_encode ? (void*)c->dev_leaves[slot] : (void*)c->dev_idx[slot];
        void* d_out = is_encode ? (void*)c->dev_idx[slot] : (void*)c->dev_leaves[slot];
        if (i >= 2) PIPECHK(hipStreamWaitEvent(c->s_in, c->ev_done[slot], 0));  // slot's previous input consumed
        if (want_stage && i >= 2) PIPECHK(hipEventSynchronize(c->ev_in[slot]));  // the H2D that last read this pinned buffer is done
        const void* src = produce(o, m, want_stage ? c->pin_in[slot] : nullptr);
        if (!src) return abort_run(c->err.empty() ? fail(c, VQHIP_ERR_INVALID, "input source failed") : VQHIP_ERR_INVALID);
        PIPECHK(hipMemcpyAsync(d_in, src, (size_t)m * in_b, hipMemcpyHostToDevice, c->s_in));
        PIPECHK(hipEventRecord(c->ev_in[slot], c->s_in));
        PIPECHK(hipStreamWaitEvent(c->stream, c->ev_in[slot], 0));
        if (i >= 2) PIPECHK(hipStreamWaitEvent(c->stream, c->ev_out[slot], 0));  // slot's previous output drained to pinned
        rc = is_encode ? encode_chunk(c, c->dev_leaves[slot], m, c->dev_idx[slot], c->stream)
                       : decode_chunk(c, c->dev_idx[slot], m, c->dev_leaves[slot], c->stream);
        if (rc) return abort_run(rc);
        PIPECHK(hipEventRecord(c->ev_done[slot], c->stream));
        if ((rc = drain())) return abort_run(rc);  // chunk i-1 -> caller, overlapped with chunk i on the GPU
        PIPECHK(hipStreamWaitEvent(c->s_out, c->ev_done[slot], 0));
        PIPECHK(hipMemcpyAsync(c->pin_out[slot], d_out, (size_t)m * out_b, hipMemcpyDeviceToHost, c->s_out));
        PIPECHK(hipEventRecord(c->ev_out[slot], c->s_out));
        prev_off = o, prev_m = m, prev_slot = slot;
    }
    if ((rc = drain())) return abort_run(rc);
    PIPECHK(hipStreamSynchronize(c->stream));
#undef PIPECHK
    return VQHIP_OK;
}

// gather per-leaf buffers into a contiguous (pinned) block / scatter a block into per-leaf buffers
void gather_leaves(float* stage, const float* const* lp, int64_t m)
{
    host_parallel_for(m, [=](int64_t a, int64_t b) {
        for (int64_t l = a; l < b; ++l) std::memcpy(stage + l * 512, lp[l], 2048);
    });
}
void scatter_leaves(float* const* dst, const float* src, int64_t m)
{
    host_parallel_for(m, [=](int64_t a, int64_t b) {
        for (int64_t l = a; l < b; ++l) std::memcpy(dst[l], src + l * 512, 2048);
    });
}

// contiguous blocks or leaf-pointer arrays (vqhip_encode / _decode / _encode_leaves / _decode_leaves)
int run_host_pipeline(vqhip_codec* c, bool is_encode, const void* in, void* out, int64_t n, const float* const* in_ptrs = nullptr,
                      float* const* out_ptrs = nullptr)
{
    const size_t in_b = is_encode ? 2048 : 64, out_b = is_encode ? 64 : 2048;
    // leaf blocks go through pinned staging (copied by threads when large), not the runtime's pageable-copy path; a call that fits one
    // chunk stages whatever it gets (decode: 64 B per leaf) so that its H2D is an asynchronous copy on the compute stream
    const bool stage_in = in_ptrs || (is_encode && n >= 4096) || n <= c->chunk;
    return run_pipeline(
        c, is_encode, n, 0, stage_in,
        [=](int64_t o, int64_t m, void* stage) -> const void* {
            const void* src = static_cast<const char*>(in) + (size_t)o * in_b;
            if (!stage) return src;
            if (in_ptrs) gather_leaves(static_cast<float*>(stage), in_ptrs + o, m);
            else host_parallel_copy(stage, src, (size_t)m * in_b);
            return stage;
        },
        [=](int64_t o, int64_t m, const void* res) -> int {
            if (out_ptrs) scatter_leaves(out_ptrs + o, static_cast<const float*>(res), m);
            else host_parallel_copy(static_cast<char*>(out) + (size_t)o * out_b, res, (size_t)m * out_b);
            return VQHIP_OK;
        });
}

// ---------------- .vqvdb v3 container (reference src/Utils/VQVDB_Reader.{hpp,cpp}) ----------------
// file : "VQVDB" | u8 version=3 | u8 numGrids | u32 numEmbeddings | u8 latentDimCount
// grid : u32 nameLength | name | f32 transform[16] | u16 latentShape[latentDimCount] | u32 totalBlocks
//        totalBlocks x { i32 origin[3] | u8 indices[64] }   (76 B per leaf)
constexpr size_t REC_BYTES = 76;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct FileCloser {
    FILE* f;
    ~FileCloser()
    {
        if (f) std::fclose(f);
    }
};

// One decoded-side batch travelling from the reader thread to the pipeline: indices de-framed from the records,
// origins, and the leaf addresses the caller's allocator returned for them.
struct StreamBatch {
    std::vector<uint8_t> idx;
    std::vector<int32_t> origins;
    std::vector<float*> ptrs;
    std::vector<unsigned char> raw;
};

}  // namespace

// =============================== C ABI ===============================
extern "C" {

const char* vqhip_version(void) { return "vqvdb-hip 0.1 (gfx950)"; }

const char* vqhip_last_error(const vqhip_codec* codec) { return codec ? codec->err.c_str() : g_create_error.c_str(); }

int vqhip_create(const char* pack_path, const void* pack_bytes, size_t pack_size, int device_id, vqhip_codec** out)
{
    if (!out) return fail(nullptr, VQHIP_ERR_INVALID, "vqhip_create: out is NULL");
    *out = nullptr;
    std::vector<unsigned char> file;
    const unsigned char* p = static_cast<const unsigned char*>(pack_bytes);
    size_t n = pack_size;
    if (pack_path) {
        std::ifstream f(pack_path, std::ios::binary);
        if (!f) return fail(nullptr, VQHIP_ERR_MODEL, std::string("Model file not found at path: ") + pack_path);
        file.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        p = file.data();
        n = file.size();
    }
    if (!p || !n) return fail(nullptr, VQHIP_ERR_MODEL, "vqhip_create: no weight pack given (embedded model absent from this build)");
    std::map<std::string, PackTensor> pk;
    std::string err;
    if (!parse_pack(p, n, pk, err)) return fail(nullptr, VQHIP_ERR_MODEL, err);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VQHIP_ERR_DEVICE, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, VQHIP_ERR_INVALID, "vqhip_create: device_id out of range");
    vqhip_codec* c = new vqhip_codec();
    c->device = device_id;
    auto bail = [&](int rc) {
        g_create_error = c->err;
        vqhip_destroy(c);
        return rc;
    };
    if (hipSetDevice(device_id) != hipSuccess) {
        c->err = "hipSetDevice failed";
        return bail(VQHIP_ERR_DEVICE);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        c->err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return bail(VQHIP_ERR_DEVICE);
    }
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = std::getenv("VQHIP_DOWN")) c->convdown_lds = std::strcmp(e, "rows") != 0;
    if (const char* e = std::getenv("VQHIP_CONV4")) c->conv4_lds = std::strcmp(e, "rows") != 0;
    if (const char* e = std::getenv("VQHIP_FIRST_SRC")) c->first_raw = std::strcmp(e, "raw") == 0;
    if (const char* e = std::getenv("VQHIP_FIRST")) c->first_roll = std::strcmp(e, "steps") != 0, c->first_roll_stats = std::strcmp(e, "roll0") == 0;
    if (const char* e = std::getenv("VQHIP_CONV8")) c->conv8_lds = std::strcmp(e, "rows") != 0, c->conv8_w16 = std::strcmp(e, "w8") != 0;
    if (const char* e = std::getenv("VQHIP_TAIL16_TILES")) c->tail16_tiles = std::atoi(e);
    if (const char* e = std::getenv("VQHIP_HOST_SPLIT")) {
        c->host_split = std::max(1, std::atoi(e));
        if (const char* m = std::strchr(e, ',')) c->host_split_min = std::max(32, std::atoi(m + 1));
    }
    if (const char* e = std::getenv("VQHIP_TAIL")) {
        if (std::strcmp(e, "rows16") && std::strcmp(e, "slab") && std::strcmp(e, "rows32") && std::strcmp(e, "groups") && *e) {
            c->err = std::string("VQHIP_TAIL=") + e + ": unknown folded-tail variant (rows16 | rows32 | groups | slab)";
            return bail(VQHIP_ERR_INVALID);
        }
        c->tail_rows = std::strcmp(e, "slab") != 0, c->tail_rows32 = std::strcmp(e, "rows32") == 0, c->tail_groups = std::strcmp(e, "groups") == 0;
    }
    if (const char* e = std::getenv("VQHIP_R64S")) c->r64s_resident = std::strcmp(e, "stream") != 0;
    if (const char* e = std::getenv("VQHIP_VQ_SPLIT")) c->vq_split = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("VQHIP_TRAIN_STEM")) c->train_stem_lut = std::strcmp(e, "conv") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_WGRAD")) c->train_wgrad_rows = std::strcmp(e, "pairs") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_STREAMS")) c->train_side_stream = std::strcmp(e, "1") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_GNBWD")) c->train_gn_fused = std::strcmp(e, "split") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_BIAS"))
        c->train_bias_main = std::strcmp(e, "side") != 0, c->train_red_stream = std::strcmp(e, "third") == 0, c->train_red_deferred = std::strcmp(e, "deferred") == 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_R64")) c->train_r64_quarters = std::strcmp(e, "whole") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_DGRAD")) c->train_dgrad_small_lds = std::strcmp(e, "resident") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_WGRAD_CU_PCT")) c->train_wgrad_cu_pct = std::min(100, std::max(10, std::atoi(e)));
    if (const char* e = std::getenv("VQHIP_TRAIN_RED_OWN_LEAVES")) c->train_red_own_leaves = std::atoll(e);
    if (const char* e = std::getenv("VQHIP_TRAIN_EMA_AT")) c->train_ema_early = std::strcmp(e, "backward") != 0;
    if (const char* e = std::getenv("VQHIP_STEM")) c->stem_fused = std::strcmp(e, "split") != 0, c->stem_taps = std::strcmp(e, "gather") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_TAIL")) c->train_folded_tail = std::strcmp(e, "unfolded") != 0;
    if (const char* e = std::getenv("VQHIP_TRAIN_EMA")) c->train_ema_lists = std::strcmp(e, "scan") != 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->err = "hipStreamCreate failed";
        return bail(VQHIP_ERR_DEVICE);
    }
    int rc = load_weights(c, pk);
    if (rc) return bail(rc);
    if ((rc = init_kernel_attrs(c))) return bail(rc);
    *out = c;
    return VQHIP_OK;
}

void vqhip_destroy(vqhip_codec* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& kv : c->dw) hipFree(kv.second);
    if (c->ws) hipFree(c->ws);
    if (c->tr_z) hipFree(c->tr_z);
    if (c->tr_idx) hipFree(c->tr_idx);
    if (c->tr_part) hipFree(c->tr_part);
    if (c->ft_refrag_jobs) hipFree(c->ft_refrag_jobs);
    if (c->ft_P) hipFree(c->ft_P);
    if (c->ft_M) hipFree(c->ft_M);
    if (c->ft_V) hipFree(c->ft_V);
    if (c->ft_ws) hipFree(c->ft_ws);
    if (c->ft_part) hipFree(c->ft_part);
    for (hipEvent_t e : c->ft_ev) hipEventDestroy(e);
    if (c->ft_side) hipStreamDestroy(c->ft_side);
    if (c->ft_red) hipStreamDestroy(c->ft_red);
    if (c->ft_red_shared) hipStreamDestroy(c->ft_red_shared);
    if (c->tr_recon) hipFree(c->tr_recon);
    if (c->tr_loss_part) hipFree(c->tr_loss_part);
    for (int i = 0; i < 2; ++i) {
        if (c->dev_leaves[i]) hipFree(c->dev_leaves[i]);
        if (c->dev_idx[i]) hipFree(c->dev_idx[i]);
        if (c->pin_out[i]) hipHostFree(c->pin_out[i]);
        if (c->pin_in[i]) hipHostFree(c->pin_in[i]);
        if (c->ev_in[i]) hipEventDestroy(c->ev_in[i]);
        if (c->ev_done[i]) hipEventDestroy(c->ev_done[i]);
        if (c->ev_out[i]) hipEventDestroy(c->ev_out[i]);
    }
    if (c->s_in) hipStreamDestroy(c->s_in);
    if (c->s_out) hipStreamDestroy(c->s_out);
    for (auto& t : c->timers) {
        hipEventDestroy(t.start);
        hipEventDestroy(t.stop);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int vqhip_latent_shape(const vqhip_codec* c, int64_t out[3])
{
    if (!c || !out) return VQHIP_ERR_INVALID;
    out[0] = out[1] = out[2] = 4;  // 8^3 leaf through the k4/s2 down-conv (weights validated at create)
    return VQHIP_OK;
}

int vqhip_set_chunk_leaves(vqhip_codec* c, int64_t chunk)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (chunk < 32 || chunk > (1 << 22)) return fail(c, VQHIP_ERR_INVALID, "chunk_leaves must be in [32, 4194304]");
    c->chunk = (chunk + 31) / 32 * 32;
    c->chunk_fitted = false;   // re-checked against the free device memory at the next call
    return VQHIP_OK;
}

int vqhip_set_small_batch_tiles(vqhip_codec* c, int tiles)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (tiles < -1) return fail(c, VQHIP_ERR_INVALID, "small_batch_tiles must be >= 0, or -1 for the automatic choice");
    c->split_tiles = tiles;
    return VQHIP_OK;
}

int vqhip_reserve(vqhip_codec* c, int64_t n)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (n < 1) return fail(c, VQHIP_ERR_INVALID, "reserve: n_leaves < 1");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->chunk_fitted) fit_chunk_to_free_memory(c), c->chunk_fitted = true;
    const int64_t m = std::min(n, c->chunk);
    int rc = ensure_workspace(c, m);
    if (!rc) rc = ensure_io(c, m);
    if (!rc) rc = ensure_stage(c, (size_t)m * 2048);
    return rc;
}

int64_t vqhip_workspace_bytes(const vqhip_codec* c) { return c ? (int64_t)c->ws_bytes : -1; }

int64_t vqhip_chunk_leaves(const vqhip_codec* c) { return c ? c->chunk : -1; }

int vqhip_encode_device(vqhip_codec* c, const float* d_leaves, int64_t n, uint8_t* d_idx, void* stream)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!d_leaves || !d_idx || n < 1) return fail(c, VQHIP_ERR_INVALID, "encode: null pointer or n_leaves < 1");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->chunk_fitted) fit_chunk_to_free_memory(c), c->chunk_fitted = true;
    if (int trc = ensure_tables(c)) return trc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (int64_t o = 0; o < n; o += c->chunk) {
        const int64_t m = std::min(c->chunk, n - o);
        int rc = encode_chunk(c, d_leaves + o * 512, m, d_idx + o * 64, s);
        if (rc) return rc;
    }
    return VQHIP_OK;
}

int vqhip_decode_device(vqhip_codec* c, const uint8_t* d_idx, int64_t n, float* d_out, void* stream)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!d_idx || !d_out || n < 1) return fail(c, VQHIP_ERR_INVALID, "decode: null pointer or n_leaves < 1");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->chunk_fitted) fit_chunk_to_free_memory(c), c->chunk_fitted = true;
    if (int trc = ensure_tables(c)) return trc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    for (int64_t o = 0; o < n; o += c->chunk) {
        const int64_t m = std::min(c->chunk, n - o);
        int rc = decode_chunk(c, d_idx + o * 64, m, d_out + o * 512, s);
        if (rc) return rc;
    }
    return VQHIP_OK;
}

int vqhip_encode(vqhip_codec* c, const float* leaves, int64_t n, uint8_t* indices)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!leaves || !indices || n < 1) return fail(c, VQHIP_ERR_INVALID, "encode: null pointer or n_leaves < 1");
    return run_host_pipeline(c, true, leaves, indices, n);
}

int vqhip_decode(vqhip_codec* c, const uint8_t* indices, int64_t n, float* leaves)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!leaves || !indices || n < 1) return fail(c, VQHIP_ERR_INVALID, "decode: null pointer or n_leaves < 1");
    return run_host_pipeline(c, false, indices, leaves, n);
}

int vqhip_encode_leaves(vqhip_codec* c, const float* const* leaf_ptrs, int64_t n, uint8_t* indices)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!leaf_ptrs || !indices || n < 1) return fail(c, VQHIP_ERR_INVALID, "encode_leaves: null pointer or n_leaves < 1");
    return run_host_pipeline(c, true, nullptr, indices, n, leaf_ptrs, nullptr);
}

int vqhip_decode_leaves(vqhip_codec* c, const uint8_t* indices, int64_t n, float* const* leaf_ptrs)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!leaf_ptrs || !indices || n < 1) return fail(c, VQHIP_ERR_INVALID, "decode_leaves: null pointer or n_leaves < 1");
    return run_host_pipeline(c, false, indices, nullptr, n, nullptr, leaf_ptrs);
}

// ---- .vqvdb stream entry points: file read || GPU decode || leaf insert ----
// Replaces the body of VQVAECodec::decompress (VQVAECodec.cpp:137-208): per grid, a reader thread reads and de-frames
// batch k+1..k+2 and asks the caller's allocator for their leaf buffers while the GPU decodes batch k and the calling
// thread scatters batch k-1 straight from pinned memory into those buffers.
int vqhip_decompress_file(vqhip_codec* c, const char* path, int64_t batch_leaves, vqhip_grid_begin_fn grid_begin, vqhip_leaf_alloc_fn leaf_alloc,
                          void* user, vqhip_stream_stats* stats)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!path || !leaf_alloc) return fail(c, VQHIP_ERR_INVALID, "decompress_file: null path or leaf allocator");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(c, VQHIP_ERR_INVALID, std::string("Cannot open input file: ") + path);
    FileCloser closer{f};
    const double t_start = now_s();
    vqhip_stream_stats st;
    std::memset(&st, 0, sizeof st);
    unsigned char h[12];
    if (std::fread(h, 1, 12, f) != 12) return fail(c, VQHIP_ERR_INVALID, "Failed to read file header.");
    if (std::memcmp(h, "VQVDB", 5) != 0) return fail(c, VQHIP_ERR_INVALID, "Invalid file magic; not a .vqvdb file.");
    if (h[5] != 3) return fail(c, VQHIP_ERR_INVALID, "Unsupported .vqvdb version " + std::to_string((int)h[5]) + " (expected 3).");
    const int n_grids = h[6], dim_count = h[11];
    uint32_t num_emb;
    std::memcpy(&num_emb, h + 7, 4);
    if (dim_count != 3) return fail(c, VQHIP_ERR_INVALID, "latent rank " + std::to_string(dim_count) + " in file; this codec decodes [4,4,4] latents");

    for (int g = 0; g < n_grids; ++g) {
        vqhip_grid_info gi;
        std::memset(&gi, 0, sizeof gi);
        uint32_t name_len = 0, total = 0;
        uint16_t shp[3];
        std::string name;
        if (std::fread(&name_len, 4, 1, f) != 1 || name_len > (1u << 20)) return fail(c, VQHIP_ERR_INVALID, "Failed to read grid name length.");
        name.resize(name_len);
        if (name_len && std::fread(&name[0], 1, name_len, f) != name_len) return fail(c, VQHIP_ERR_INVALID, "Failed to read grid name.");
        if (std::fread(gi.transform, 4, 16, f) != 16) return fail(c, VQHIP_ERR_INVALID, "Failed to read transform.");
        if (std::fread(shp, 2, 3, f) != 3) return fail(c, VQHIP_ERR_INVALID, "Failed to read latent shape.");
        if (std::fread(&total, 4, 1, f) != 1) return fail(c, VQHIP_ERR_INVALID, "File appears truncated, failed to read total block count.");
        if (shp[0] != 4 || shp[1] != 4 || shp[2] != 4)
            return fail(c, VQHIP_ERR_INVALID, "grid '" + name + "' has latent shape [" + std::to_string(shp[0]) + "," + std::to_string(shp[1]) + "," +
                                                  std::to_string(shp[2]) + "]; this codec decodes [4,4,4]");
        gi.name = name.c_str();
        gi.grid_index = g;
        for (int i = 0; i < 3; ++i) gi.latent_shape[i] = shp[i];
        gi.num_embeddings = num_emb;
        gi.total_blocks = total;
        if (grid_begin && grid_begin(user, &gi) != 0) return fail(c, VQHIP_ERR_INVALID, "grid_begin callback failed for grid '" + name + "'");
        ++st.grids;
        const int64_t n = total;
        if (n == 0) continue;
        const int64_t step = std::min(batch_leaves > 0 ? std::min(batch_leaves, c->chunk) : c->chunk, n);
        const int64_t nb = (n + step - 1) / step;

        constexpr int Q = 3;
        StreamBatch slot[Q];
        std::mutex mu;
        std::condition_variable cv;
        int64_t produced = 0, consumed = 0;
        bool failed = false, stop = false;
        std::string herr;
        double read_s = 0, alloc_s = 0, wait_s = 0, copy_s = 0;
        std::thread reader([&] {
            auto bail = [&](const std::string& m) {
                std::lock_guard<std::mutex> lk(mu);
                herr = m;
                failed = true;
                cv.notify_all();
            };
            for (int64_t k = 0; k < nb; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k - consumed < Q; });
                    if (stop) return;
                }
                const int64_t m = std::min(step, n - k * step);
                StreamBatch& B = slot[k % Q];
                double t = now_s();
                B.raw.resize((size_t)m * REC_BYTES);
                if (std::fread(B.raw.data(), REC_BYTES, (size_t)m, f) != (size_t)m) return bail("File truncated: incomplete block data.");
                B.idx.resize((size_t)m * 64);
                B.origins.resize((size_t)m * 3);
                B.ptrs.assign((size_t)m, nullptr);
                const unsigned char* p = B.raw.data();
                for (int64_t l = 0; l < m; ++l, p += REC_BYTES) {
                    std::memcpy(&B.origins[3 * l], p, 12);
                    std::memcpy(&B.idx[64 * l], p + 12, 64);
                }
                read_s += now_s() - t;
                t = now_s();
                const int rc = leaf_alloc(user, g, B.origins.data(), m, B.ptrs.data());
                alloc_s += now_s() - t;
                if (rc != 0) return bail("leaf_alloc callback failed (" + std::to_string(rc) + ")");
                for (int64_t l = 0; l < m; ++l)
                    if (!B.ptrs[l]) return bail("leaf_alloc callback left a null leaf pointer");
                std::lock_guard<std::mutex> lk(mu);
                produced = k + 1;
                cv.notify_all();
            }
        });
        const int rc = run_pipeline(
            c, false, n, step, false,
            [&](int64_t o, int64_t, void*) -> const void* {
                const int64_t k = o / step;
                const double t = now_s();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || produced > k; });
                wait_s += now_s() - t;
                if (produced <= k) {
                    c->err = herr;
                    return nullptr;
                }
                return slot[k % Q].idx.data();
            },
            [&](int64_t o, int64_t m, const void* res) -> int {
                const int64_t k = o / step;
                const double t = now_s();
                scatter_leaves(slot[k % Q].ptrs.data(), static_cast<const float*>(res), m);
                copy_s += now_s() - t;
                std::lock_guard<std::mutex> lk(mu);
                consumed = k + 1;
                cv.notify_all();
                return VQHIP_OK;
            });
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        reader.join();
        if (rc) return rc;
        st.leaves += n;
        st.read_s += read_s, st.alloc_s += alloc_s, st.io_wait_s += wait_s, st.copy_s += copy_s;
    }
    st.wall_s = now_s() - t_start;
    if (stats) *stats = st;
    return VQHIP_OK;
}

// Replaces the body of VQVAECodec::compress (VQVAECodec.cpp:78-134): gather the leaf buffers into pinned memory,
// encode on the GPU, frame {origin, 64 indices} records and append them to the file while the next batch encodes.
int vqhip_compress_file(vqhip_codec* c, const char* path, const vqhip_grid_source* grids, int n_grids, int64_t batch_leaves, vqhip_stream_stats* stats)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!path || !grids) return fail(c, VQHIP_ERR_INVALID, "compress_file: null path or grid list");
    if (n_grids < 1 || n_grids > 255) return fail(c, VQHIP_ERR_INVALID, "compress_file: a .vqvdb file holds 1..255 grids");
    for (int g = 0; g < n_grids; ++g) {
        const vqhip_grid_source& G = grids[g];
        if (!G.name || G.n_leaves < 0 || G.n_leaves > 0xFFFFFFFFll || (G.n_leaves > 0 && (!G.leaf_ptrs || !G.origins)))
            return fail(c, VQHIP_ERR_INVALID, "compress_file: grid " + std::to_string(g) + " has no name, no leaves/origins or more than 2^32-1 leaves");
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, VQHIP_ERR_INVALID, std::string("Cannot open output file: ") + path);
    FileCloser closer{f};
    const double t_start = now_s();
    vqhip_stream_stats st;
    std::memset(&st, 0, sizeof st);
    bool wfail = false;
    auto put = [&](const void* p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) wfail = true;
    };
    unsigned char h[12];
    std::memcpy(h, "VQVDB", 5);
    h[5] = 3;
    h[6] = (unsigned char)n_grids;
    const uint32_t num_emb = 256;
    std::memcpy(h + 7, &num_emb, 4);
    h[11] = 3;
    put(h, 12);
    std::vector<unsigned char> rec;
    for (int g = 0; g < n_grids; ++g) {
        const vqhip_grid_source& G = grids[g];
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const uint32_t name_len = (uint32_t)std::strlen(G.name), total = (uint32_t)G.n_leaves;
        const uint16_t shp[3] = {4, 4, 4};
        put(&name_len, 4);
        put(G.name, name_len);
        put(G.transform ? G.transform : ident, 64);
        put(shp, 6);
        put(&total, 4);
        ++st.grids;
        if (wfail) return fail(c, VQHIP_ERR_INVALID, "Failed to write to .vqvdb file.");
        if (G.n_leaves == 0) continue;
        double copy_s = 0, write_s = 0;
        const int rc = run_pipeline(
            c, true, G.n_leaves, batch_leaves, true,
            [&](int64_t o, int64_t m, void* stage) -> const void* {
                const double t = now_s();
                gather_leaves(static_cast<float*>(stage), G.leaf_ptrs + o, m);
                copy_s += now_s() - t;
                return stage;
            },
            [&](int64_t o, int64_t m, const void* res) -> int {
                const double t = now_s();
                rec.resize((size_t)m * REC_BYTES);
                const uint8_t* idx = static_cast<const uint8_t*>(res);
                unsigned char* p = rec.data();
                for (int64_t l = 0; l < m; ++l, p += REC_BYTES) {
                    std::memcpy(p, G.origins + 3 * (o + l), 12);
                    std::memcpy(p + 12, idx + 64 * l, 64);
                }
                put(rec.data(), rec.size());
                write_s += now_s() - t;
                return wfail ? fail(c, VQHIP_ERR_INVALID, "Failed to write to .vqvdb file.") : VQHIP_OK;
            });
        if (rc) return rc;
        st.leaves += G.n_leaves;
        st.copy_s += copy_s, st.read_s += write_s;
    }
    closer.f = nullptr;
    if (std::fclose(f) != 0) return fail(c, VQHIP_ERR_INVALID, "Error closing the output file.");
    st.wall_s = now_s() - t_start;
    if (stats) *stats = st;
    return VQHIP_OK;
}

// ---- codebook training (stage 1): VectorQuantizerEMA.forward in training mode, VQVAE_v2.py:107-156 ----
int vqhip_train_begin(vqhip_codec* c, const float* cluster_size, const float* embed_avg)
{
    if (!c) return VQHIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (!c->training) {
        if ((rc = upload(c, "tr.efrag", std::vector<float>(16 * 8 * 64 * 4, 0.0f)))) return rc;
        if ((rc = upload(c, "tr.ee", std::vector<float>(256, 0.0f)))) return rc;
    }
    std::vector<float> cs(256, 1.0f), avg(256 * 128);  // reference initial buffers (VQVAE_v2.py:104-105)
    if (cluster_size) cs.assign(cluster_size, cluster_size + 256);
    if (embed_avg) avg.assign(embed_avg, embed_avg + 256 * 128);
    else HIPCHK(c, hipMemcpy(avg.data(), c->dw["cb"], avg.size() * sizeof(float), hipMemcpyDeviceToHost));
    if ((rc = upload(c, "tr.cs", cs))) return rc;
    if ((rc = upload(c, "tr.avg", avg))) return rc;
    c->training = true;
    return VQHIP_OK;
}

// scratch of the codebook statistics (tr_part): per-(code, segment) partial sums, squared errors and counts, then the per-segment member
// lists of vq_ema_lists_k and their (start, count) table
struct EmaScratch {
    float* part;
    double* sqpart;
    int* cntpart;
    unsigned short* lists;
    int* starts;
};
size_t ema_scratch_bytes(int n_seg)
{
    return (size_t)256 * n_seg * (128 * sizeof(float) + sizeof(double) + sizeof(int)) + (size_t)n_seg * VQ_SEG_ROWS * sizeof(unsigned short) +
           (size_t)n_seg * 256 * 2 * sizeof(int);
}
EmaScratch ema_scratch(char* base, int n_seg)
{
    EmaScratch e;
    e.part = reinterpret_cast<float*>(base);
    e.sqpart = reinterpret_cast<double*>(base + (size_t)256 * n_seg * 128 * sizeof(float));
    e.cntpart = reinterpret_cast<int*>(base + (size_t)256 * n_seg * (128 * sizeof(float) + sizeof(double)));
    e.lists = reinterpret_cast<unsigned short*>(base + (size_t)256 * n_seg * (128 * sizeof(float) + sizeof(double) + sizeof(int)));
    e.starts = reinterpret_cast<int*>(reinterpret_cast<char*>(e.lists) + (size_t)n_seg * VQ_SEG_ROWS * sizeof(unsigned short));
    return e;
}
// encodings_sum / dw / squared error of one batch into `stats` (VQ_STATS_* layout): member lists per (segment, code), ordered partial sums,
// ordered reduction.  VQHIP_TRAIN_EMA=scan: every (code, segment) wave scans the segment's indices itself (rounds 1-4; the same bits)
void launch_vq_ema(vqhip_codec* c, hipStream_t s, const float* z, const uint8_t* idx, int64_t rows, int n_seg, float* stats)
{
    const EmaScratch e = ema_scratch(c->tr_part, n_seg);
    if (c->train_ema_lists) {
        hipLaunchKernelGGL(vq_ema_lists_k, dim3(n_seg), dim3(64), 0, s, idx, rows, e.lists, e.starts);
        hipLaunchKernelGGL(vq_ema_gather_k, dim3(256, (n_seg + 15) / 16), dim3(1024), 0, s, z, e.lists, e.starts, c->dw["cb"], rows, n_seg, e.part, e.sqpart, e.cntpart);
    } else {
        hipLaunchKernelGGL(vq_ema_partials_k, dim3(256, (n_seg + 15) / 16), dim3(1024), 0, s, z, idx, c->dw["cb"], rows, n_seg, e.part, e.sqpart, e.cntpart);
    }
    hipLaunchKernelGGL(vq_ema_reduce_k, dim3(256), dim3(128), 0, s, e.part, e.sqpart, e.cntpart, rows, n_seg, stats);
}

int vqhip_train_vq_stats_device(vqhip_codec* c, const float* d_leaves, int64_t n, float* d_stats, uint8_t* d_idx, float* d_latent, void* stream)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!c->training) return fail(c, VQHIP_ERR_INVALID, "train_vq_stats: call vqhip_train_begin first");
    if (!d_leaves || !d_stats || n < 1) return fail(c, VQHIP_ERR_INVALID, "train_vq_stats: null pointer or n_leaves < 1");
    if (n > c->chunk) return fail(c, VQHIP_ERR_INVALID, "train_vq_stats: a training batch may not exceed the chunk size (" + std::to_string(c->chunk) + " leaves)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if ((!d_latent || !d_idx) && c->tr_leaves < n) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->tr_z) hipFree(c->tr_z);
        if (c->tr_idx) hipFree(c->tr_idx);
        c->tr_z = nullptr, c->tr_idx = nullptr, c->tr_leaves = 0;
        HIPCHK(c, hipMalloc(&c->tr_z, (size_t)n * 64 * 128 * sizeof(float)));
        HIPCHK(c, hipMalloc(&c->tr_idx, (size_t)n * 64));
        c->tr_leaves = n;
    }
    {
        const size_t need = ema_scratch_bytes((int)((n * 64 + VQ_SEG_ROWS - 1) / VQ_SEG_ROWS));
        if (c->tr_part_bytes < need) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->tr_part) hipFree(c->tr_part);
            c->tr_part = nullptr, c->tr_part_bytes = 0;
            HIPCHK(c, hipMalloc(&c->tr_part, need));
            c->tr_part_bytes = need;
        }
    }
    float* z = d_latent ? d_latent : c->tr_z;
    uint8_t* idx = d_idx ? d_idx : c->tr_idx;
    int rc = encode_chunk(c, d_leaves, n, idx, s, z);
    if (rc) return rc;
    const int64_t rows = n * 64;
    const int n_seg = (int)((rows + VQ_SEG_ROWS - 1) / VQ_SEG_ROWS);
    Launcher L{c, s, n};
    L.run("train_vq_ema_stats", [&] { launch_vq_ema(c, s, z, idx, rows, n_seg, d_stats); });
    return L.rc;
}

int vqhip_train_eval_device(vqhip_codec* c, const float* d_leaves, int64_t n, float* d_stats, float* d_recon_sums, float* d_recon, void* stream)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!c->training) return fail(c, VQHIP_ERR_INVALID, "train_eval: call vqhip_train_begin first");
    if (!d_leaves || !d_stats || !d_recon_sums || n < 1) return fail(c, VQHIP_ERR_INVALID, "train_eval: null pointer or n_leaves < 1");
    if (n > c->chunk) return fail(c, VQHIP_ERR_INVALID, "train_eval: a batch may not exceed the chunk size (" + std::to_string(c->chunk) + " leaves)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // forward in eval mode (VQVAE.forward, VQVAE_v2.py:344-348): assignment + statistics exactly as in training, no update
    int rc = vqhip_train_vq_stats_device(c, d_leaves, n, d_stats, nullptr, nullptr, s);
    if (rc) return rc;
    if ((rc = ensure_tables(c))) return rc;  // the decoder's stem table follows the live codebook
    float* recon = d_recon;
    if (!recon) {
        if (c->tr_recon_leaves < n) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->tr_recon) hipFree(c->tr_recon);
            c->tr_recon = nullptr, c->tr_recon_leaves = 0;
            HIPCHK(c, hipMalloc(&c->tr_recon, (size_t)n * 512 * sizeof(float)));
            c->tr_recon_leaves = n;
        }
        recon = c->tr_recon;
    }
    if (!c->tr_loss_part) HIPCHK(c, hipMalloc(&c->tr_loss_part, (size_t)RL_BLOCKS * 2 * sizeof(double)));
    if ((rc = decode_chunk(c, c->tr_idx, n, recon, s))) return rc;
    Launcher L{c, s, n};
    L.run("train_recon_loss", [&] {
        hipLaunchKernelGGL(recon_loss_partials_k, dim3(RL_BLOCKS), dim3(256), 0, s, d_leaves, recon, n * 512, c->tr_loss_part);
        hipLaunchKernelGGL(recon_loss_reduce_k, dim3(1), dim3(1), 0, s, c->tr_loss_part, n * 512, d_recon_sums);
    });
    return L.rc;
}

int vqhip_train_vq_update_device(vqhip_codec* c, const float* d_stats, float decay, float eps, void* stream)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!c->training) return fail(c, VQHIP_ERR_INVALID, "train_vq_update: call vqhip_train_begin first");
    if (!d_stats || !(decay >= 0.0f && decay <= 1.0f) || !(eps > 0.0f)) return fail(c, VQHIP_ERR_INVALID, "train_vq_update: null stats, decay outside [0,1] or eps <= 0");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const float alpha = (float)(1.0 - (double)decay);
    Launcher L{c, s, 0};
    L.run("train_vq_ema_update", [&] {
        hipLaunchKernelGGL(vq_ema_update_k, dim3(256), dim3(128), 0, s, d_stats, decay, alpha, eps, c->dw["tr.cs"], c->dw["tr.avg"], c->dw["cb"]);
    });
    c->tables_stale = true;
    return L.rc;
}

int vqhip_train_get_state(vqhip_codec* c, float* embedding, float* cluster_size, float* embed_avg)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!c->training && (cluster_size || embed_avg)) return fail(c, VQHIP_ERR_INVALID, "train_get_state: call vqhip_train_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (embedding) HIPCHK(c, hipMemcpy(embedding, c->dw["cb"], 256 * 128 * sizeof(float), hipMemcpyDeviceToHost));
    if (cluster_size) HIPCHK(c, hipMemcpy(cluster_size, c->dw["tr.cs"], 256 * sizeof(float), hipMemcpyDeviceToHost));
    if (embed_avg) HIPCHK(c, hipMemcpy(embed_avg, c->dw["tr.avg"], 256 * 128 * sizeof(float), hipMemcpyDeviceToHost));
    return VQHIP_OK;
}

int vqhip_train_set_state(vqhip_codec* c, const float* embedding, const float* cluster_size, const float* embed_avg)
{
    if (!c) return VQHIP_ERR_INVALID;
    if (!c->training && (cluster_size || embed_avg)) return fail(c, VQHIP_ERR_INVALID, "train_set_state: call vqhip_train_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (embedding) {
        HIPCHK(c, hipMemcpy(c->dw["cb"], embedding, 256 * 128 * sizeof(float), hipMemcpyHostToDevice));
        c->tables_stale = true;
    }
    if (cluster_size) HIPCHK(c, hipMemcpy(c->dw["tr.cs"], cluster_size, 256 * sizeof(float), hipMemcpyHostToDevice));
    if (embed_avg) HIPCHK(c, hipMemcpy(c->dw["tr.avg"], embed_avg, 256 * 128 * sizeof(float), hipMemcpyHostToDevice));
    return VQHIP_OK;
}

int vqhip_train_commit(vqhip_codec* c)
{
    if (!c) return VQHIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    return c->tables_stale ? refresh_tables(c) : VQHIP_OK;
}

// ---- in-process multi-GPU front end: one codec + one host thread per device, contiguous leaf ranges ----
// One persistent host thread per device: it owns the device's handle for the lifetime of the front end (hipSetDevice once, pinned
// staging allocated from this thread, i.e. on the memory node of the cores it is bound to), sleeps on a condition variable between
// calls and runs one leaf range per call.  The thread is bound to the cores of its GPU's NUMA node when /sys exposes it; the
// copy threads a call fans out to inherit that mask.
struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = true, quit = false;
    bool is_encode = false;
    const void* in = nullptr;
    void* out = nullptr;
    int64_t n = 0;
    int rc = VQHIP_OK;
    int numa_node = -1, cpus_bound = 0;
};

struct vqhip_multi {
    std::vector<vqhip_codec*> dev;
    std::vector<std::unique_ptr<MultiWorker>> workers;
    std::string err;
};

// cores of the NUMA node the device hangs off (empty if unknown)
static std::vector<int> device_numa_cpus(int device, int* node_out)
{
    std::vector<int> cpus;
    char bus[64] = {0};
    *node_out = -1;
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return cpus;
    for (char* p = bus; *p; ++p) *p = (char)std::tolower(*p);
    std::ifstream fn(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
    int node = -1;
    if (!(fn >> node) || node < 0) return cpus;
    *node_out = node;
    std::ifstream fl("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string list;
    if (!std::getline(fl, list)) return cpus;
    size_t pos = 0;
    while (pos < list.size()) {
        size_t end = list.find(',', pos);
        if (end == std::string::npos) end = list.size();
        const std::string part = list.substr(pos, end - pos);
        const size_t dash = part.find('-');
        try {
            const int lo = std::stoi(part.substr(0, dash)), hi = dash == std::string::npos ? lo : std::stoi(part.substr(dash + 1));
            for (int cpu = lo; cpu <= hi; ++cpu) cpus.push_back(cpu);
        } catch (...) {
            return {};
        }
        pos = end + 1;
    }
    return cpus;
}

static void multi_worker_main(vqhip_multi* m, int g, int copy_cap)
{
    MultiWorker& w = *m->workers[g];
    g_copy_threads_cap = copy_cap;
    hipSetDevice(m->dev[g]->device);
    {   // NUMA-local: restrict this thread (and the copy threads it spawns) to the device's node, within the process's mask
        int node = -1;
        const std::vector<int> cpus = device_numa_cpus(m->dev[g]->device, &node);
        cpu_set_t cur, want;
        CPU_ZERO(&want);
        if (!cpus.empty() && sched_getaffinity(0, sizeof(cur), &cur) == 0) {
            int cnt = 0;
            for (int cpu : cpus)
                if (cpu < CPU_SETSIZE && CPU_ISSET(cpu, &cur)) CPU_SET(cpu, &want), ++cnt;
            if (cnt > 0 && sched_setaffinity(0, sizeof(want), &want) == 0) w.numa_node = node, w.cpus_bound = cnt;
        }
    }
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.cv.wait(lk, [&] { return w.has_job || w.quit; });
        if (w.quit) return;
        w.has_job = false;
        lk.unlock();
        const int rc = w.is_encode ? vqhip_encode(m->dev[g], static_cast<const float*>(w.in), w.n, static_cast<uint8_t*>(w.out))
                                   : vqhip_decode(m->dev[g], static_cast<const uint8_t*>(w.in), w.n, static_cast<float*>(w.out));
        lk.lock();
        w.rc = rc;
        w.done = true;
        w.cv.notify_all();
    }
}

int vqhip_multi_create(const char* pack_path, const void* pack_bytes, size_t pack_size, const int* device_ids, int n_devices, vqhip_multi** out)
{
    if (!out) return fail(nullptr, VQHIP_ERR_INVALID, "vqhip_multi_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1) return fail(nullptr, VQHIP_ERR_INVALID, "vqhip_multi_create: need at least one device id");
    vqhip_multi* m = new vqhip_multi();
    for (int i = 0; i < n_devices; ++i) {
        vqhip_codec* c = nullptr;
        const int rc = vqhip_create(pack_path, pack_bytes, pack_size, device_ids[i], &c);
        if (rc != VQHIP_OK) {
            for (vqhip_codec* d : m->dev) vqhip_destroy(d);
            delete m;
            return rc;  // message already in the thread-local create error
        }
        m->dev.push_back(c);
    }
    // all devices together never run more than min(cores, 64) copy threads: 8 GPUs x 16 threads each would oversubscribe the
    // memory controllers long before they help a 57 GB/s PCIe link
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int cap = std::max(2, std::min(16, std::min(hw, 64) / n_devices));
    for (int g = 0; g < n_devices; ++g) m->workers.emplace_back(new MultiWorker());
    for (int g = 0; g < n_devices; ++g) m->workers[g]->th = std::thread(multi_worker_main, m, g, cap);
    *out = m;
    return VQHIP_OK;
}

void vqhip_multi_destroy(vqhip_multi* m)
{
    if (!m) return;
    for (auto& w : m->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    for (vqhip_codec* d : m->dev) vqhip_destroy(d);
    delete m;
}

const char* vqhip_multi_last_error(const vqhip_multi* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int vqhip_multi_worker_info(const vqhip_multi* m, int index, int* device_id, int* numa_node, int* cpus_bound)
{
    if (!m || index < 0 || index >= (int)m->workers.size()) return VQHIP_ERR_INVALID;
    if (device_id) *device_id = m->dev[index]->device;
    if (numa_node) *numa_node = m->workers[index]->numa_node;
    if (cpus_bound) *cpus_bound = m->workers[index]->cpus_bound;
    return VQHIP_OK;
}

// rank g of G takes leaves [g*ceil(n/G), min(n,(g+1)*ceil(n/G))); results land at the same
// offsets of the caller's buffer, so leaf order is preserved and there is no collective.
static int multi_run(vqhip_multi* m, bool is_encode, const void* in, void* out, int64_t n)
{
    if (!m) return VQHIP_ERR_INVALID;
    if (!in || !out || n < 1) {
        m->err = "multi: null pointer or n_leaves < 1";
        return VQHIP_ERR_INVALID;
    }
    const int G = (int)m->dev.size();
    const int64_t per = (n + G - 1) / G;
    std::vector<char> posted(G, 0);
    for (int g = 0; g < G; ++g) {
        const int64_t lo = std::min(n, g * per), hi = std::min(n, lo + per);
        if (hi == lo) continue;
        MultiWorker& w = *m->workers[g];
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.is_encode = is_encode;
            w.in = is_encode ? static_cast<const void*>(static_cast<const float*>(in) + lo * 512) : static_cast<const void*>(static_cast<const uint8_t*>(in) + lo * 64);
            w.out = is_encode ? static_cast<void*>(static_cast<uint8_t*>(out) + lo * 64) : static_cast<void*>(static_cast<float*>(out) + lo * 512);
            w.n = hi - lo;
            w.done = false;
            w.has_job = true;
        }
        w.cv.notify_all();
        posted[g] = 1;
    }
    int first_bad = -1, rc_bad = VQHIP_OK;
    for (int g = 0; g < G; ++g) {
        if (!posted[g]) continue;
        MultiWorker& w = *m->workers[g];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        if (w.rc != VQHIP_OK && first_bad < 0) first_bad = g, rc_bad = w.rc;
    }
    if (first_bad >= 0) {
        m->err = "device " + std::to_string(m->dev[first_bad]->device) + ": " + m->dev[first_bad]->err;
        return rc_bad;
    }
    return VQHIP_OK;
}

int vqhip_multi_encode(vqhip_multi* m, const float* leaves, int64_t n, uint8_t* indices) { return multi_run(m, true, leaves, indices, n); }
int vqhip_multi_decode(vqhip_multi* m, const uint8_t* indices, int64_t n, float* leaves) { return multi_run(m, false, indices, leaves, n); }

int vqhip_debug_enable(vqhip_codec* c, int enable)
{
    if (!c) return VQHIP_ERR_INVALID;
    c->debug = enable != 0;
    return VQHIP_OK;
}

int vqhip_profile_enable(vqhip_codec* c, int enable)
{
    if (!c) return VQHIP_ERR_INVALID;
    c->profiling = enable != 0;
    return VQHIP_OK;
}

int vqhip_profile_read(vqhip_codec* c, vqhip_kernel_stat* stats, int cap, int* count)
{
    if (!c || !count) return VQHIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<std::string> order;
    std::map<std::string, vqhip_kernel_stat> agg;
    for (auto& t : c->timers) {
        HIPCHK(c, hipEventSynchronize(t.stop));
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, t.start, t.stop));
        auto it = agg.find(t.name);
        if (it == agg.end()) {
            vqhip_kernel_stat s{};
            std::snprintf(s.name, sizeof(s.name), "%s", t.name.c_str());
            auto ki = kernel_info().find(t.name);
            if (ki != kernel_info().end()) s.flops_per_leaf = ki->second.flops, s.eff_flops_per_leaf = ki->second.eff_flops;
            it = agg.emplace(t.name, s).first;
            order.push_back(t.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.leaves += t.leaves;
        hipEventDestroy(t.start);
        hipEventDestroy(t.stop);
    }
    c->timers.clear();
    *count = (int)order.size();
    for (int i = 0; i < (int)order.size() && i < cap && stats; ++i) stats[i] = agg[order[i]];
    return VQHIP_OK;
}

int vqhip_debug_fetch(vqhip_codec* c, const char* name, int64_t n, float* out)
{
    if (!c || !name || !out || n < 1) return VQHIP_ERR_INVALID;
    auto it = c->act.find(name);
    if (it == c->act.end()) return fail(c, VQHIP_ERR_INVALID, std::string("debug_fetch: unknown activation '") + name + "'");
    const int C = c->act_shape[name].first, NP = c->act_shape[name].second;
    if (std::strcmp(name, "xr") == 0) return fail(c, VQHIP_ERR_INVALID, "debug_fetch: 'xr' is the first conv's row layout [tile][64 rows][32 leaves][12], not a [tile][pos][32] activation; fetch 'xt'");
    if ((C != 1 && C < 4) || n > std::max(c->ws_tiles, c->ft_tiles) * 32) return fail(c, VQHIP_ERR_INVALID, "debug_fetch: not a tile activation or n too large");
    for (const ActSpec& a : kActs)   // compact (inference) layout: large activations share three regions and overwrite each other
        if (!c->ws_full && a.region != -1 && std::strcmp(a.name, name) == 0)
            return fail(c, VQHIP_ERR_INVALID, std::string("debug_fetch: '") + name + "' is not kept in the compact inference workspace; call vqhip_debug_enable(1) before the pass");
    if (!it->second) return fail(c, VQHIP_ERR_INVALID, std::string("debug_fetch: '") + name + "' is not allocated");
    if (C == 1) {  // [tile][NP][32]
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        const int64_t tl = (n + 31) / 32;
        std::vector<float> raw1((size_t)tl * 32 * NP);
        HIPCHK(c, hipMemcpy(raw1.data(), it->second, raw1.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t l = 0; l < n; ++l)
            for (int p = 0; p < NP; ++p) out[(size_t)l * NP + p] = raw1[((size_t)(l / 32) * NP + p) * 32 + (l % 32)];
        return VQHIP_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    const int64_t tiles = (n + 31) / 32;
    std::vector<float> raw((size_t)tiles * 32 * C * NP);
    HIPCHK(c, hipMemcpy(raw.data(), it->second, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t l = 0; l < n; ++l)
        for (int ch = 0; ch < C; ++ch)
            for (int p = 0; p < NP; ++p)
                out[((size_t)l * C + ch) * NP + p] = raw[((((size_t)(l / 32) * NP + p) * (C / 4) + ch / 4) * 32 + (l % 32)) * 4 + ch % 4];
    return VQHIP_OK;
}

int vqhip_selftest_mfma(vqhip_codec* c, int64_t* mismatches)
{
    if (!c || !mismatches) return VQHIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const int ks = 24;
    std::vector<float> h((size_t)(32 * 2 * ks) * 2 + (size_t)(16 * 4 * ks) * 2);
    uint32_t x = 12345u;
    for (auto& v : h) {
        x = x * 1664525u + 1013904223u;
        v = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 22)) * (1.0f + (float)(x & 7));
    }
    float* d = nullptr;
    unsigned long long* dm = nullptr;
    HIPCHK(c, hipMalloc(&d, h.size() * sizeof(float)));
    HIPCHK(c, hipMalloc(&dm, 2 * sizeof(unsigned long long)));
    HIPCHK(c, hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(dm, 0, 2 * sizeof(unsigned long long)));
    const float* a32 = d;
    const float* b32 = a32 + 32 * 2 * ks;
    const float* a16 = b32 + 32 * 2 * ks;
    const float* b16 = a16 + 16 * 4 * ks;
    hipLaunchKernelGGL(mfma_probe_k, dim3(1), dim3(64), 0, c->stream, a32, b32, a16, b16, ks, dm);
    HIPCHK(c, hipGetLastError());
    unsigned long long hm[2] = {0, 0};
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost));
    hipFree(d);
    hipFree(dm);
    mismatches[0] = (int64_t)hm[0];
    mismatches[1] = (int64_t)hm[1];
    return VQHIP_OK;
}

}  // extern "C"

#include "vq_train_full.inc"
