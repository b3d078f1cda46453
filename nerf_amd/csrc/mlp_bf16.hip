// bf16 variant of the fused positional-encoding + NeRF MLP query: v_mfma_f32_32x32x16_bf16, gfx950.
//
// Same transposed-product chaining as the f32 kernel (mlp_fused.hip): out^T[channel, point] = W * act^T, the
// accumulators of one layer ARE the B operand of the next after a pairwise f32->bf16 pack (no LDS, no lane
// movement for activations).  What changes with a 16x faster matrix pipe is the weight traffic: per-wave
// streaming from L2 would need ~128 B/clk/CU, more than L2 delivers.  So here
//   * a wave owns 64 points (two 32-point MFMA column tiles): every weight fragment feeds two MFMAs;
//   * the 4 waves of a workgroup share the weights through LDS: the net's weights are ONE linear stream of
//     16 KiB chunks (16 A-fragments of 1 KiB, in the exact order the MFMAs consume them); each wave copies a
//     quarter of the next chunk global->LDS with buffer_load ... lds (no registers) while the current chunk is
//     being consumed; one barrier per chunk (1024 MFMA cycles);
//   * the bias rides in the GEMM as a constant-1 input channel (the encoders' pad slot, or one extra k-step);
//   * alpha / rgb heads are one-output-tile GEMMs on the same stream;
//   * gamma(x): one accurate sincosf per coordinate at the lane's base frequency, higher octaves by the
//     double-angle recurrence (error ~2^-16, far below bf16's 2^-9 rounding of the encoded value).
// The plain 16-bit kernels (bf16, f16) are inference only.  Further down: the bf16x3 kernels (three bf16 MFMAs per product on
// hi/residual operands: f32-grade) -- inference, the training forward that saves its activations (f32 T8 or f16 P16
// sections), and the activation-gradient chain; the weight-gradient GEMMs are in mlp_bwd.hip.
#include "mlp_device.h"
#include <stdlib.h>
#include <utility>
#include <type_traits>

namespace nrf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// Plain-bf16 kernels: the ENCODED INPUT blocks (gamma(x), gamma(d), gamma(t) and their weight columns) go through the f16
// MFMA (v_mfma_f32_32x32x16_f16: same rate, same f32 accumulators) instead of the bf16 one.  Encoder values lie in [-1, 1]
// and scene coordinates within a few units, so f16's range is no concern there, and its 11-bit mantissa cuts the rounding of
// the inputs and of those weights by 8x.  Measured on a trained field ('A trained field'): input rounding was
// the largest single term of the bf16 image error.  Hidden activations stay bf16 (range).
#ifndef NRF_BF16_F16_INPUTS
#define NRF_BF16_F16_INPUTS 1
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

constexpr int kChunk = 16384;   // bytes per LDS stage: 16 A-fragments (one 32x16 bf16 tile = 1 KiB per wave load)
constexpr int kFrag = 1024;

#ifdef NRF_BF16_PROFILE
static unsigned long long *g_prof_host_ptr = nullptr;
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif


// ---- channel maps (shared by the pack kernel and the query kernel) -----------------------------------
// A k-step is one MFMA (16 input channels): lane half h supplies elements j = 0..7 (k = 8h + j).
// Encoded inputs: lane half h owns a LOCAL list of 8*KS slots; k-step s takes slots 8s..8s+7.
//   gamma(3-vector), L frequencies: slot 0: x | z, slot 1: y | CONST 1 (bias), then per frequency f = 2i + h:
//   sin(2^f x y z), cos(2^f x y z).   Returns the reference channel (positional_encoding.py:11-36), -1 = zero
//   pad, -2 = the constant-1 channel.
NRF_HD inline int bf_pe3_chan(int L, int h, int slot) {
    if (slot == 0) return h ? 2 : 0;
    if (slot == 1) return h ? -2 : 1;
    int q = slot - 2, i = q / 6, r = q - 6 * i, f = 2 * i + h;
    return f < L ? 3 + 6 * f + r : -1;
}
//   gamma(scalar) (D-NeRF time): slot 0: t | pad, then per frequency f = 2i + h: sin, cos.
NRF_HD inline int bf_pe1_chan(int L, int h, int slot) {
    if (slot == 0) return h ? -1 : 0;
    int q = slot - 1, i = q >> 1, r = q & 1, f = 2 * i + h;
    return f < L ? 1 + 2 * f + r : -1;
}
//   previous layer's accumulators: k-step s = (tile s>>1, register half s&1); element j = register 8(s&1) + j
NRF_HD inline int bf_hid_chan(int h, int s, int j) {
    int t = s >> 1, r = 8 * (s & 1) + j;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

enum { BFB_NONE = 0, BFB_HIDDEN = 1, BFB_PE3 = 2, BFB_PE1 = 3, BFB_BIAS = 4 };
struct BfBlock { int32_t type, param, ksteps, col0, ncols; };
struct BfLayer {
    const float *W, *b;
    int32_t ld, n_out, nt_out, nblk;
    int32_t tail;    // the last `tail` k-steps are stored tile-outer ([out tile][k-step]), the rest [k-step][out tile]
    int32_t split;   // bf16x3: every fragment is followed by its residual fragment, lo = bf16(w - float(bf16(w)))
    int32_t transposed;   // backward chain: rows = W's INPUT channels col0 + out (n_out valid), contraction over W's output
                          // channels (blk.ncols of them): the fragment holds W^T
    int32_t f16;          // the f16 variant: EVERY block (hidden weights and the bias step too) is consumed by the f16 MFMA
    BfBlock blk[3];
    int64_t dst;   // byte offset in the stream
};

// one thread = one lane's 16 bytes (8 bf16) of one A fragment; blockIdx.y picks the layer of the batch (a training step
// repacks three streams per net after every optimizer step: one launch per 16 layers instead of one per layer)
constexpr int kPackBatch = 16;
struct BfPackBatch { BfLayer layer[kPackBatch]; };
__global__ void pack_bf16_kernel(BfPackBatch batch, char *stream) {
    const BfLayer &a = batch.layer[blockIdx.y];
    int ks_total = 0;
    for (int i = 0; i < a.nblk; ++i) ks_total += a.blk[i].ksteps;
    const int64_t nthreads = (int64_t)ks_total * a.nt_out * 64 * (a.split ? 2 : 1);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < nthreads; idx += (int64_t)gridDim.x * blockDim.x) {
        const int lane = idx & 63, sfrag = (int)(idx >> 6);          // fragment index in the stream
        const int frag = a.split ? sfrag >> 1 : sfrag, lo_part = a.split ? sfrag & 1 : 0;
        const int head = (ks_total - a.tail) * a.nt_out;
        int t, s;
        if (frag < head) { t = frag % a.nt_out; s = frag / a.nt_out; }
        else { t = (frag - head) / a.tail; s = ks_total - a.tail + (frag - head) % a.tail; }
        int bi = 0;
        while (s >= a.blk[bi].ksteps) { s -= a.blk[bi].ksteps; ++bi; }
        const BfBlock B = a.blk[bi];
        const int m = lane & 31, h = lane >> 5, out = 32 * t + m;
        bf16x8 v;
        for (int j = 0; j < 8; ++j) {
            int ch = -1;
            switch (B.type) {
                case BFB_HIDDEN: ch = bf_hid_chan(h, s, j); break;
                case BFB_PE3: ch = bf_pe3_chan(B.param, h, 8 * s + j); break;
                case BFB_PE1: ch = bf_pe1_chan(B.param, h, 8 * s + j); break;
                case BFB_BIAS: ch = (h == 0 && j == 0) ? -2 : -1; break;
            }
            float w = 0.0f;
            if (out < a.n_out) {
                if (a.transposed) { if (ch >= 0 && ch < B.ncols) w = a.W[(int64_t)ch * a.ld + B.col0 + out]; }
                else if (ch >= 0 && ch < B.ncols) w = a.W[(int64_t)out * a.ld + B.col0 + ch];
                else if (ch == -2) w = a.b[out];
            }
            const __bf16 hi = (__bf16)w;
            v[j] = lo_part ? (__bf16)(w - (float)hi) : hi;
            if (NRF_BF16_F16_INPUTS && !a.split && !a.transposed && (B.type == BFB_PE3 || B.type == BFB_PE1))
                v[j] = __builtin_bit_cast(__bf16, (_Float16)w);      // this block is consumed by the f16 MFMA
            if (a.f16)    // saturating: a weight beyond f16's range stays finite (the activations are clamped the same way)
                v[j] = __builtin_bit_cast(__bf16, (_Float16)fminf(fmaxf(w, -65504.0f), 65504.0f));
        }
        *reinterpret_cast<bf16x8 *>(stream + a.dst + (int64_t)sfrag * kFrag + lane * 16) = v;
    }
}

// ---- host-side stream plan --------------------------------------------------------------------------
// Stream = the layers in consumption order, each layer's fragments contiguous ([k-step][out tile] order) and
// each layer padded to a whole number of 16 KiB chunks; then a small f32 table of head biases.
constexpr int kBiasTable = 64;   // floats
struct BfBiasEntry { const float *src; int32_t n, dst; };
struct BfPlan {
    BfLayer layer[2 * NRF_MAX_D + 12];
    BfBiasEntry bias[8];
    int n = 0, nbias = 0;
    int split = 0;      // bf16x3 stream: hi and residual fragments interleaved
    int f16 = 0;        // the f16 variant's stream (BfLayer::f16)
    int64_t bytes = 0;
    void add(const float *W, const float *b, int ld, int n_out, int nt_out, int tail, BfBlock b0, BfBlock b1 = {}, BfBlock b2 = {}) {
        BfLayer &L = layer[n++];
        L.W = W; L.b = b; L.ld = ld; L.n_out = n_out; L.nt_out = nt_out; L.tail = tail; L.split = split; L.transposed = 0;
        L.f16 = f16;
        L.blk[0] = b0; L.blk[1] = b1; L.blk[2] = b2;
        L.nblk = 1 + (b1.type != BFB_NONE) + (b2.type != BFB_NONE);
        L.dst = bytes;
        int64_t ks = b0.ksteps + b1.ksteps + b2.ksteps;
        bytes += (ks * nt_out * (split ? 2 : 1) * kFrag + kChunk - 1) / kChunk * kChunk;
    }
    // W^T of the columns col0 .. col0 + k_valid of an nn.Linear with n_src outputs, as a layer of nt_out output tiles
    void add_T(const float *W, int ld, int col0, int k_valid, int nt_out, int n_src) {
        add(W, W, ld, k_valid, nt_out, 0, BfBlock{BFB_HIDDEN, 0, n_src / 16, col0, n_src});
        layer[n - 1].transposed = 1;
    }
    void add_bias(const float *src, int n_, int dst) { bias[nbias++] = BfBiasEntry{src, n_, dst}; }
};

constexpr int kBfKsX = 4, kBfKsD = 2, kBfKsT = 2;
constexpr int kBfTail = 0, kBfTail0 = 0, kBfTail0T = 0;   // (tile-outer layer tails: measured slower)
// head-bias table slots (per net: +0 canonical net, +32 time net)
constexpr int kHbOut = 0, kHbAlpha = 8, kHbRgb = 9, kHbTime = 32;

static const char *bf_check_net(const nrf_net_t *net, const nrf_embed_t *emb, bool time_net) {
    if (emb->mode != 0) return "the bf16 kernels take raw points (embed mode 0) only";
    if (net->use_layernorm) return "LayerNorm nets run on the f32 kernels only";
    if (net->W != 128 && net->W != 256) return "the bf16 kernels are built for W = 128 and 256";
    if (net->D < 1 || net->D > NRF_MAX_D) return "D out of range";
    if (emb->multires < 0 || emb->multires > 10 || net->input_ch != 3 + 6 * emb->multires)
        return "input_ch must equal 3 + 6*multires with multires <= 10";
    if (time_net) {
        if (emb->multires_time < 0 || emb->multires_time > 10 || net->input_ch_views != 1 + 2 * emb->multires_time)
            return "time channels must equal 1 + 2*multires_time with multires_time <= 10";
    } else if (net->use_viewdirs) {
        if (emb->multires_views < 0 || emb->multires_views > 4 || net->input_ch_views != 3 + 6 * emb->multires_views)
            return "input_ch_views must equal 3 + 6*multires_views with multires_views <= 4";
    } else if (net->output_ch < 1 || net->output_ch > 8) {
        return "output_ch must be 1..8 without view directions";
    }
    return nullptr;
}

// Appends one net's layers in the order the kernel consumes them.  time_net: layer 0 reads cat[gamma(x), gamma(t)],
// the output layer is `out_w` (W -> 3).  Hidden layers carry their bias as a 17th k-step (constant-1 channel);
// layers fed by an encoded input take it from that input's constant slot; the one-tile heads add it on the VALU.
// straight (DNeRF_Straightforward, vanilla_dynamic/network.py:100-196; time channels in net->reserved): layer 0 and the skip
// layers read cat[gamma(x), gamma(t)] (+ h).
static void bf_plan_net(BfPlan &P, const nrf_net_t *net, const nrf_embed_t *emb, bool time_net, bool straight = false) {
    const int W = net->W, nt = W / 32, ic = net->input_ch, D = net->D;
    const BfBlock X{BFB_PE3, emb->multires, kBfKsX, 0, ic};
    const BfBlock H{BFB_HIDDEN, 0, 2 * nt, 0, W}, Bi{BFB_BIAS, 0, 1, 0, 0};
    const int tc = straight ? net->reserved : 0;
    const BfBlock T{BFB_PE1, emb->multires_time, kBfKsT, ic, tc};
    if (straight)
        P.add(net->pts_w[0], net->pts_b[0], ic + tc, W, nt, kBfTail0T, X, T);
    else if (time_net)
        P.add(net->pts_w[0], net->pts_b[0], ic + net->input_ch_views, W, nt, kBfTail0T, X,
              BfBlock{BFB_PE1, emb->multires_time, kBfKsT, ic, net->input_ch_views});
    else
        P.add(net->pts_w[0], net->pts_b[0], ic, W, nt, kBfTail0, X);
    for (int i = 1; i < D; ++i) {
        const bool skip = (net->skips_mask >> (i - 1)) & 1;   // network.py:52-53
        if (skip && straight)
            P.add(net->pts_w[i], net->pts_b[i], W + ic + tc, W, nt, kBfTail, X, T, BfBlock{BFB_HIDDEN, 0, 2 * nt, ic + tc, W});
        else if (skip)
            P.add(net->pts_w[i], net->pts_b[i], W + ic, W, nt, kBfTail, X, BfBlock{BFB_HIDDEN, 0, 2 * nt, ic, W});
        else
            P.add(net->pts_w[i], net->pts_b[i], W, W, nt, kBfTail, H, Bi);
    }
    if (time_net) {
        P.add(net->out_w, net->out_b, W, 3, 1, 0, H);                             // vanilla_dynamic/network.py:305
        P.add_bias(net->out_b, 3, kHbTime);
        return;
    }
    if (!net->use_viewdirs) {
        P.add(net->out_w, net->out_b, W, net->output_ch, 1, 0, H);                // network.py:67
        P.add_bias(net->out_b, net->output_ch, kHbOut);
        return;
    }
    P.add(net->alpha_w, net->alpha_b, W, 1, 1, 0, H);                             // network.py:56
    P.add_bias(net->alpha_b, 1, kHbAlpha);
    P.add(net->feature_w, net->feature_b, W, W, nt, kBfTail, H, Bi);                    // network.py:57
    P.add(net->views_w, net->views_b, W + net->input_ch_views, W / 2, nt / 2, kBfTail, H,
          BfBlock{BFB_PE3, emb->multires_views, kBfKsD, W, net->input_ch_views});   // network.py:58-62 (bias: CONST slot)
    P.add(net->rgb_w, net->rgb_b, W / 2, 3, 1, 0, BfBlock{BFB_HIDDEN, 0, nt, 0, W / 2});   // network.py:64
    P.add_bias(net->rgb_b, 3, kHbRgb);
}

// The activation-gradient chain's stream (nrf_query_bwd_bf16x3): W^T blocks in the order dgrad_x3_kernel consumes them --
// views_linears[0][:, :W]^T, feature_linear^T, then the hidden parts of the trunk layers D-1 .. 1.
static void bf_plan_net_bwd(BfPlan &P, const nrf_net_t *net) {
    const int W = net->W, nt = W / 32, ic = net->input_ch, D = net->D;
    if (net->use_viewdirs) {
        P.add_T(net->views_w, W + net->input_ch_views, 0, W, nt, W / 2);
        P.add_T(net->feature_w, W, 0, W, nt, W);
    }
    for (int l = D - 1; l >= 1; --l) {
        const bool skip = (net->skips_mask >> (l - 1)) & 1;
        P.add_T(net->pts_w[l], W + (skip ? ic : 0), skip ? ic : 0, W, nt, W);
    }
}

struct BfBiasBatch { BfBiasEntry e[8]; };
__global__ void pack_bias_kernel(BfBiasBatch b, float *table) {
    const BfBiasEntry &e = b.e[blockIdx.x];
    if ((int)threadIdx.x < e.n) table[e.dst + threadIdx.x] = e.src[threadIdx.x];
}

static int64_t bf_total_bytes(const BfPlan &P) { return P.bytes + kBiasTable * 4; }

static int bf_pack(const BfPlan &P, void *packed, hipStream_t st) {
    if (hipMemsetAsync(packed, 0, bf_total_bytes(P), st) != hipSuccess)
        return nrf::fail(NRF_ERR_LAUNCH, "bf16 pack: hipMemsetAsync failed");
    for (int i = 0; i < P.n; ++i)
        if (!P.layer[i].W || !P.layer[i].b)
            return nrf::fail(NRF_ERR_INVALID, "bf16 pack: a parameter pointer of the net is null (layer %d)", i);
    for (int i0 = 0; i0 < P.n; i0 += kPackBatch) {
        BfPackBatch b{};
        const int nb = P.n - i0 < kPackBatch ? P.n - i0 : kPackBatch;
        for (int i = 0; i < nb; ++i) b.layer[i] = P.layer[i0 + i];
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(32, nb), dim3(256), 0, st, b, (char *)packed);
    }
    if (P.nbias > 0) {
        BfBiasBatch bb{};
        for (int i = 0; i < P.nbias; ++i) bb.e[i] = P.bias[i];
        hipLaunchKernelGGL(pack_bias_kernel, dim3(P.nbias), dim3(64), 0, st, bb, (float *)((char *)packed + P.bytes));
    }
    return check_launch("nrf_mlp_bf16_pack_weights");
}

// ---- device side ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// encoded-input pair of the plain-bf16 kernels: f16 (see NRF_BF16_F16_INPUTS)
__device__ __forceinline__ uint32_t pack2in(float a, float b) {
#if NRF_BF16_F16_INPUTS
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2));
#else
    return pack2(a, b);
#endif
}
// ReLU on a packed bf16 pair: as signed 16-bit integers negative floats are negative, so max(.., 0) clears them
__device__ __forceinline__ uint32_t relu2(uint32_t p) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), (s16x2){0, 0}));
}

// The f16 variant of the plain 16-bit kernels (F16 = true; `precision = "f16"`): hidden activations and their weights in
// f16 as well.  Same matrix rate and the same f32 accumulators as bf16; 11 mantissa bits instead of 8, so all three
// rounding terms of the image error (inputs, weights, activations) are 8x smaller.  What bf16 buys is range, so the
// conversion SATURATES: round-to-nearest v_cvt_pk_f16_f32, ReLU as a signed 16-bit max (as for bf16), then an unsigned
// 16-bit min against 65504 on the now non-negative patterns -- +inf and NaN patterns included -- one more packed op per
// pair.  (v_cvt_pkrtz_f16_f32 saturates for free but rounds toward zero: a -2^-11 bias per layer that a ReLU chain
// compounds.)  The one layer without a ReLU (feature_linear) clamps in f32 before the conversion.
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2));
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sat2h(uint32_t p) {      // p: two non-negative f16 patterns
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, p), (u16x2){0x7BFF, 0x7BFF}));
}

// The weight stream as seen by one wave.  LDS holds FOUR 16 KiB stages: chunk c is consumed from stage c%4 while
// chunk c+1 is already complete in the next stage and chunks c+2, c+3 are in flight into the other two (the mid-chunk
// wait leaves the youngest chunk's four copies outstanding: loads return in order).  Two chunks in flight cover an L2
// miss: both nets of the D-NeRF chain as hi/residual pairs are 4.4 MB, more than one XCD's 4 MB L2, and with three
// stages that kernel stalled on its copies (0.886 -> 0.813 s per 800x800 view; no change where the stream fits in L2).
// Fragments are read
// kAhead ahead of the MFMA that uses them, across chunk and layer boundaries, so the matrix pipe never waits for
// an LDS round trip; the one barrier per chunk sits in the MIDDLE of a chunk, where nothing else is pending.
constexpr int kAhead = 4;
#ifndef NRF_STREAM_STAGES
#define NRF_STREAM_STAGES 4
#endif
constexpr int kStages = NRF_STREAM_STAGES;
template <int AHEAD>
struct WStreamT {
    static constexpr int kDepth = AHEAD;
    __amdgpu_buffer_rsrc_t rsrc;
    char *lds;
    uint32_t total;      // bytes of all chunks (the stream wraps there)
    uint32_t pf_off;     // stream offset of the next chunk to request
    uint32_t st_req;     // stage (byte offset in lds) the next request goes to
    uint32_t rd_cur, rd_nxt;   // this lane's byte offset of fragment 0 in the current / the next chunk's stage
    int lane16, wave;
    bf16x8 q[AHEAD];     // fragments in flight / ready: fragment i of a layer sits in q[i % AHEAD]

    // one of the four 1 KiB pieces this wave copies per chunk; the ring position advances after the last one
    __device__ __forceinline__ void request_piece(int i) {
        const uint32_t f = (uint32_t)wave + 4 * i;
#ifndef NRF_EXP_NODMA       // (timing experiments: results are wrong with any NRF_EXP_* macro)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds + st_req + f * kFrag),
                                                 16, lane16, (int)(pf_off + f * kFrag), 0, 0);
#endif
        if (i == 3) {
            pf_off += kChunk;
            if (pf_off == total) pf_off = 0;
            st_req += kChunk;
            if (st_req == kStages * kChunk) st_req = 0;
        }
    }
    __device__ __forceinline__ void request() {
#pragma unroll
        for (int i = 0; i < 4; ++i) request_piece(i);
    }
    // Mid-chunk: publish the NEXT chunk (own copies landed -> barrier -> everyone's landed, and everyone is past
    // the previous chunk, whose stage the copies of the chunk after next may now overwrite).  The caller then issues
    // the four copies one at a time between MFMAs (each VMEM issue stalls the wave; spread out, the matrix pipe
    // covers part of every stall instead of idling through four in a row).
    // YOUNGER: vector-memory operations known to have been issued AFTER the copies this publish waits for and, like the
    // youngest chunk's four copies, allowed to stay in flight: the activation stores of the layer before (stream_frags).
    // Without the allowance the in-order counter makes a layer's first chunks wait until the L2 has acknowledged the
    // previous layer's stores.  A lower bound is safe (it only waits longer); an overestimate is not.
    template <int YOUNGER = 0>
    __device__ __forceinline__ void publish() {
        // the copies of chunks beyond the next one (4 per chunk and wave, returned in order) may stay in flight
        static_assert((kStages - 3) * 4 + YOUNGER <= 63, "vmcnt is a 6-bit counter");
#ifdef NRF_EXP_NOBAR
        return;
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kStages - 3) * 4 + YOUNGER) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void rotate() {
        rd_cur = rd_nxt;
        rd_nxt += kChunk;
        if (rd_nxt >= (uint32_t)(kStages * kChunk)) rd_nxt -= kStages * kChunk;
    }
    __device__ __forceinline__ bf16x8 read(bool nxt, int f) const {
#ifdef NRF_EXP_NOLDS
        { bf16x8 v = q[f % AHEAD]; asm volatile("" : "+v"(v)); return v; }     // no LDS read, no register write, opaque
#endif
        return *reinterpret_cast<const bf16x8 *>(lds + (nxt ? rd_nxt : rd_cur) + f * kFrag);
    }
    __device__ __forceinline__ void start() {
        pf_off = 0; st_req = 0;
#pragma unroll
        for (int i = 0; i < kStages - 1; ++i) request();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rd_cur = (uint32_t)lane16;
        rd_nxt = (uint32_t)lane16 + kChunk;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) q[i] = read(false, i);
    }
};
#ifndef NRF_X3_AHEAD
#define NRF_X3_AHEAD 4
#endif
using WStream = WStreamT<kAhead>;
using WStreamX3 = WStreamT<NRF_X3_AHEAD>;   // 1.5 MFMAs per fragment instead of 2: the same latency cover needs more fragments in flight

// The N fragments of a layer, in stream order: body(i, a) gets fragment i between two scheduling barriers; around it
// the chunk protocol runs -- publish the next chunk mid-chunk, copy requests spread over the following fragment slots,
// LDS reads kAhead fragments ahead across chunk and layer boundaries.  The layer starts on a chunk boundary;
// N % kAhead == 0 keeps q's phase static.
// YOUNGER > 0: that many stores were issued between the previous layer's last copy request and this layer.  At this
// layer's first two publishes the copies waited for (requested two chunks earlier, i.e. before those stores) are older
// than the stores, so the stores may stay in flight; from the third publish on the awaited copies are younger than them.
template <int N, int YOUNGER = 0, class ST, class Body>
__device__ __forceinline__ void stream_frags(ST &st, Body &&body) {
    constexpr int kAhead = ST::kDepth;
    static_assert(N % kAhead == 0 && N >= kAhead, "fragment count of a layer must be a multiple of the read-ahead");
    static_assert(N % 16 == 0 || N % 16 >= kAhead, "a layer's last chunk must hold at least the read-ahead");
    constexpr int NCH = (N + 15) / 16;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int F = (N - 16 * c) < 16 ? (N - 16 * c) : 16;          // fragments in this chunk
        const int pos = (F / 2 < F - kAhead) ? F / 2 : F - kAhead;    // publish before the first read of the next chunk
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int i = 16 * c + f;
            if (f == pos) {
                if (YOUNGER > 0 && kStages == 4 && c < 2) st.template publish<YOUNGER>();
                else st.template publish<0>();
            }
            {   // copies of the chunk after next: one piece at each of four fragment slots from `pos` on
                const int span = F - pos, stp = span >= 8 ? 2 : 1;
                if (span >= 4) {
                    if (f >= pos && (f - pos) % stp == 0 && (f - pos) / stp < 4) st.request_piece((f - pos) / stp);
                } else if (f == pos) {
                    st.request();
                }
            }
            const bf16x8 a = st.q[i % kAhead];
            {
                const int j = i + kAhead;
                if (j < N) st.q[i % kAhead] = st.read((j >> 4) != c, j & 15);
                else st.q[i % kAhead] = st.read(true, j - N);          // the next layer's first fragments
            }
            __builtin_amdgcn_sched_barrier(0);
            body(i, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        st.rotate();
    }
}

// acc[p][t] (+)= W_layer * B(p, s) over the layer's S k-steps; B(p, s) = 8 bf16 of point tile p for k-step s.
// k-steps IN_LO .. IN_HI-1 are encoded-input steps: f16 operands on the f16 MFMA (NRF_BF16_F16_INPUTS)
template <int NT_OUT, int S, int IN_LO = 0, int IN_HI = 0, bool F16 = false, class BF>
__device__ __forceinline__ void gemm_layer(f32x16 (&acc)[2][NT_OUT], WStream &st, BF &&B) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    stream_frags<S * NT_OUT>(st, [&](int i, const bf16x8 &a) {
        const int s = i / NT_OUT, t = i % NT_OUT;
        if (F16 || (NRF_BF16_F16_INPUTS && s >= IN_LO && s < IN_HI)) {
            const f16x8 ah = __builtin_bit_cast(f16x8, a);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, B(0, s)), s == 0 ? zero : acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, B(1, s)), s == 0 ? zero : acc[1][t], 0, 0, 0);
        } else {
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(0, s), s == 0 ? zero : acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(1, s), s == 0 ? zero : acc[1][t], 0, 0, 0);
        }
    });
}

// bf16x3: acc[t] (+)= W * x with W = Wh + Wl, x = xh + xl (bf16 each) as Wh*xh + Wh*xl + Wl*xh.  The stream holds
// (hi, residual) fragment pairs; one 32-point tile per wave; all three MFMAs of a pair accumulate into the same tile.
// BIAS_LAST: the last k-step is the bias step, whose input is the exact constant 1 (residual 0): Wh * 0 is skipped.
// YOUNGER: stores issued since the previous layer's GEMM (stream_frags)
template <int NT_OUT, int S, bool BIAS_LAST = false, int YOUNGER = 0, class BH, class BL>
__device__ __forceinline__ void gemm_layer_x3(f32x16 (&acc)[NT_OUT], WStreamX3 &st, BH &&Bh, BL &&Bl) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    stream_frags<2 * S * NT_OUT, YOUNGER>(st, [&](int i, const bf16x8 &a) {
        const int j = i >> 1, s = j / NT_OUT, t = j % NT_OUT;
        if ((i & 1) == 0) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bh(s), s == 0 ? zero : acc[t], 0, 0, 0);
            if (!(BIAS_LAST && s == S - 1)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bl(s), acc[t], 0, 0, 0);
        } else {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bh(s), acc[t], 0, 0, 0);
        }
    });
}

// accumulators -> next layer's B fragments: k-step 2t+e = registers 8e..8e+7 of tile t, packed pairwise
template <int NT, bool F16 = false>
__device__ __forceinline__ void pack_acts(u32x4v (&hin)[2][2 * NT], const f32x16 (&acc)[2][NT], bool relu) {
    // One wave per SIMD: a dependent VALU chain (read -> cvt -> max through the same temporaries) issues at a
    // third of the rate of independent instructions.  So: 16 reads, then 8 converts, then 8 max, pinned in that
    // order -- every instruction's producer is >= 8 instructions back.
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float r[16];
            uint32_t c[8];
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r[k]) : "a"(acc[p][t][k]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (F16) {
                if (!relu) {   // no ReLU, no unsigned clamp: saturate in f32 (feature_linear only, once per net)
#pragma unroll
                    for (int k = 0; k < 16; ++k) r[k] = __builtin_amdgcn_fmed3f(r[k], -65504.0f, 65504.0f);
                }
#pragma unroll
                for (int w = 0; w < 8; ++w) c[w] = pack2h(r[2 * w], r[2 * w + 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (relu) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) c[w] = relu2(c[w]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < 8; ++w) c[w] = sat2h(c[w]);
                }
#pragma unroll
                for (int w = 0; w < 8; ++w) hin[p][2 * t + (w >> 2)][w & 3] = c[w];
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) c[w] = pack2(r[2 * w], r[2 * w + 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < 8; ++w) hin[p][2 * t + (w >> 2)][w & 3] = relu ? relu2(c[w]) : c[w];
            __builtin_amdgcn_sched_barrier(0);
        }
}

// gamma(v) of a 3-vector into KS k-step fragments (slot order of bf_pe3_chan)
template <int KS>
__device__ __forceinline__ void fill_pe3_bf16(u32x4v (&out)[KS], float x, float y, float z, int h) {
    constexpr int NF = (8 * KS - 2) / 6;
    float v[8 * KS];
    v[0] = h ? z : x;
    v[1] = h ? 1.0f : y;
    const float sc = pow2i(h);
    float s[3], c[3];
    sincos_dispatch(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) * 2.0f, [&](auto sc_fn) {
        sc_fn(x * sc, &s[0], &c[0]);
        sc_fn(y * sc, &s[1], &c[1]);
        sc_fn(z * sc, &s[2], &c[2]);
    });
#pragma unroll
    for (int i = 0; i < NF; ++i) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[2 + 6 * i + d] = s[d];
            v[2 + 6 * i + 3 + d] = c[d];
            if (i + 1 < NF) {   // two octaves up: angle * 4  (frequencies >= L meet zero weights)
                const float s2 = 2.0f * s[d] * c[d], c2 = 1.0f - 2.0f * s[d] * s[d];
                s[d] = 2.0f * s2 * c2;
                c[d] = 1.0f - 2.0f * s2 * s2;
            }
        }
    }
#pragma unroll
    for (int i = 2 + 6 * NF; i < 8 * KS; ++i) v[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) out[k][w] = pack2in(v[8 * k + 2 * w], v[8 * k + 2 * w + 1]);
}

__device__ __forceinline__ bf16x8 as_bf(const u32x4v &v) { return __builtin_bit_cast(bf16x8, v); }

// Trunk of one net (network.py:48-53): layer 0 on S0 encoded k-steps B0(p, s), then the hidden layers, each followed
// by the conversion of its outputs into the next layer's B fragments.
template <int NT, int S0, bool F16 = false, class B0>
__device__ __forceinline__ void bf_trunk(f32x16 (&acc)[2][NT], u32x4v (&hin)[2][2 * NT], const u32x4v (&xin)[2][kBfKsX],
                                         const u32x4v &one, WStream &st, int n_trunk, int skip_in_mask, B0 &&b0,
                                         long long *prof = nullptr) {
    (void)prof;
    PROF_T(t0);
    gemm_layer<NT, S0, 0, S0, F16>(acc, st, b0);
    PROF_T(t1);
    pack_acts<NT, F16>(hin, acc, true);
    PROF_T(t2);
    PROF_ADD(1, t0, t1); PROF_ADD(2, t1, t2);
    for (int l = 1; l < n_trunk; ++l) {
        PROF_T(t3);
        if ((skip_in_mask >> l) & 1)
            gemm_layer<NT, kBfKsX + 2 * NT, 0, kBfKsX, F16>(acc, st, [&](int p, int s) {
                return s < kBfKsX ? as_bf(xin[p][s < kBfKsX ? s : 0]) : as_bf(hin[p][s >= kBfKsX ? s - kBfKsX : 0]);
            });
        else
            gemm_layer<NT, 2 * NT + 1, 0, 0, F16>(acc, st, [&](int p, int s) {
                return s < 2 * NT ? as_bf(hin[p][s < 2 * NT ? s : 0]) : as_bf(one);
            });
        PROF_T(t4);
        pack_acts<NT, F16>(hin, acc, true);
        PROF_T(t5);
        PROF_ADD(1, t3, t4); PROF_ADD(2, t4, t5);
    }
}

// View-direction head of the canonical net (network.py:56-65) -> (rgb + alpha) of both point tiles, stored by lane half 0.
// `din` = gamma(d), built by the caller BEFORE the trunk: 16 registers held through it cost nothing, while building it here
// (its sincos temporaries on top of the live feature operand) spilled 60 registers.
template <int NT, bool F16 = false>
__device__ __forceinline__ void bf_views_head(f32x16 (&acc)[2][NT], u32x4v (&hin)[2][2 * NT], const u32x4v &one, WStream &st,
                                              const float *hb, const u32x4v (&din)[2][kBfKsD], const int64_t (&pp)[2],
                                              const int64_t (&pc)[2], int64_t n_points, int pts_per_ray, int h, float *raw,
                                              long long *prof = nullptr) {
    (void)prof;
    PROF_T(h0);
    auto Hb = [&](int p, int s) { return s < 2 * NT ? as_bf(hin[p][s < 2 * NT ? s : 0]) : as_bf(one); };
    auto Hn = [&](int p, int s) { return as_bf(hin[p][s]); };
    f32x16 acc1[2][1];
    gemm_layer<1, 2 * NT, 0, 0, F16>(acc1, st, Hn);                                        // network.py:56
    const float ab = hb[kHbAlpha];
    const float alpha0 = acc1[0][0][0] + ab, alpha1 = acc1[1][0][0] + ab;
    PROF_T(h1);
    PROF_ADD(5, h0, h1);
    gemm_layer<NT, 2 * NT + 1, 0, 0, F16>(acc, st, Hb);                                    // network.py:57
    pack_acts<NT, F16>(hin, acc, false);
    PROF_T(h2);
    PROF_T(h3);
    PROF_ADD(6, h2, h3);
    f32x16 acc2[2][NT / 2];
    gemm_layer<NT / 2, 2 * NT + kBfKsD, 2 * NT, 2 * NT + kBfKsD, F16>(acc2, st, [&](int p, int s) {   // network.py:58-62
        return s < 2 * NT ? as_bf(hin[p][s < 2 * NT ? s : 0]) : as_bf(din[p][s >= 2 * NT ? s - 2 * NT : 0]);
    });
    u32x4v hv[2][NT];
    pack_acts<NT / 2, F16>(hv, acc2, true);
    PROF_T(h4);
    gemm_layer<1, NT, 0, 0, F16>(acc1, st, [&](int p, int s) { return as_bf(hv[p][s]); });   // network.py:64
    PROF_T(h5);
    PROF_ADD(7, h4, h5);
    if (h == 0) {
        const float b0 = hb[kHbRgb], b1 = hb[kHbRgb + 1], b2 = hb[kHbRgb + 2];
        if (pp[0] < n_points)
            reinterpret_cast<float4 *>(raw)[pp[0]] = make_float4(acc1[0][0][0] + b0, acc1[0][0][1] + b1, acc1[0][0][2] + b2, alpha0);
        if (pp[1] < n_points)
            reinterpret_cast<float4 *>(raw)[pp[1]] = make_float4(acc1[1][0][0] + b0, acc1[1][0][1] + b1, acc1[1][0][2] + b2, alpha1);
    }
}

struct Bf16Args {
    const char *stream;
    int64_t stream_bytes;    // all chunks (the head-bias table follows)
    const float *pts, *viewdirs;
    float *raw;
    int64_t n_points;
    int32_t pts_per_ray, n_trunk, skip_in_mask, use_viewdirs, output_ch, dbg, pad_[2];
    unsigned long long *prof_out;
    float *acts;             // bf16x3 training forward: [n_points / 32 tiles][tl.a_tile] f32 T8 workspace
    TrainLayout tl;
};


// VIEWS is a template parameter: with both heads in one kernel the allocator spilled another 49 registers.
template <int NT, bool VIEWS, bool F16 = false>
__global__ void __launch_bounds__(256, 1) query_fwd_bf16_kernel(const Bf16Args a) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStream st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (a.dbg & 1) ? 0 : (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);

    const int64_t ntiles = (a.n_points + 255) >> 8;
    u32x4v one;   // the constant-1 channel of a bias k-step: element 0 of lane half 0 (1.0 as bf16, or as f16)
    one[0] = h == 0 ? (F16 ? 0x3C00u : 0x3F80u) : 0u; one[1] = 0; one[2] = 0; one[3] = 0;

#ifdef NRF_BF16_PROFILE
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk0 = clock64();
#else
    long long *prof = nullptr;
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        PROF_T(ta);
        int64_t pp[2], pc[2];
        u32x4v xin[2][kBfKsX];
        u32x4v din[2][VIEWS ? kBfKsD : 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pp[p] = tile * 256 + st.wave * 64 + p * 32 + n;
            pc[p] = pp[p] < a.n_points ? pp[p] : a.n_points - 1;
            const float *q = a.pts + pc[p] * 3;
            if constexpr (VIEWS) {
                const float *d = a.viewdirs + (pc[p] / a.pts_per_ray) * 3;
                fill_pe3_bf16<kBfKsD>(din[p], d[0], d[1], d[2], h);
            }
            fill_pe3_bf16<kBfKsX>(xin[p], q[0], q[1], q[2], h);
        }
        PROF_T(tb);
        PROF_ADD(0, ta, tb);
        f32x16 acc[2][NT];
        u32x4v hin[2][2 * NT];
        bf_trunk<NT, kBfKsX, F16>(acc, hin, xin, one, st, a.n_trunk, a.skip_in_mask,
                                  [&](int p, int s) { return as_bf(xin[p][s]); }, prof);          // network.py:48-53
        PROF_T(tc);
        if constexpr (VIEWS) {
            bf_views_head<NT, F16>(acc, hin, one, st, hb, din, pp, pc, a.n_points, a.pts_per_ray, h, a.raw, prof);
            PROF_T(td);
            PROF_ADD(3, tc, td);
        } else {
            f32x16 acc1[2][1];
            gemm_layer<1, 2 * NT, 0, 0, F16>(acc1, st, [&](int p, int s) { return as_bf(hin[p][s]); });   // network.py:67
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * h + r;   // rows 0-3 on lane half 0, 4-7 on half 1 (registers 0..3)
                    if (o < a.output_ch && pp[p] < a.n_points) a.raw[pp[p] * a.output_ch + o] = acc1[p][0][r] + hb[kHbOut + o];
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests in flight must land before the LDS is released
    __builtin_amdgcn_s_barrier();
#ifdef NRF_BF16_PROFILE
    if (threadIdx.x == 0 && a.prof_out) {
        prof[4] = clock64() - tk0;
        for (int i = 0; i < 8; ++i) atomicAdd(&a.prof_out[i], (unsigned long long)prof[i]);
    }
#endif
}

// ---- bf16x3: f32-grade accuracy from the bf16 matrix pipe ------------------------------------------------
// Every f32 operand v is carried as hi = bf16(v) and lo = bf16(v - hi): 16 mantissa bits.  A product W*x is the three
// MFMAs Wh*xh + Wh*xl + Wl*xh accumulated in f32 (the dropped Wl*xl is 2^-18 relative).  Measured against the f32
// reference the raw outputs differ by ~1e-5 of their scale (bf16 alone: 5e-3), i.e. the f32 parity bar, at a third of the
// bf16 kernel's point rate -- still ~3x the f32-MFMA kernel, whose matrix rate equals the vector rate on this chip.
// One 32-point tile per wave (the accumulators are shared by the three products), four waves per workgroup.
// The residuals a - float(hi.lo), b - float(hi.hi) come from v_dot2c_f32_bf16 against the pairs (-1, 0) and (0, -1) with a
// and b as the accumulators: one operation a value instead of an unpack and a subtract, and exact (the product by -1 and the
// sum are; `scripts/microbench/dot2bf.hip`: 0 mismatches in 2^20 values over 40 binades).  The pairs must NOT be compile-time
// constants: (-1, 0) would be emitted as the inline constant -1.0, which this instruction reads as the f32 bit pattern
// 0xBF800000 -- the pair (0, -1) -- so half the residuals come out as a - float(hi.hi) (the reason an earlier attempt at
// bias row sums through this instruction was 14 % off).  They live in non-const __constant__ memory.
#ifndef NRF_X3_SPLIT_DOT2
#define NRF_X3_SPLIT_DOT2 1
#endif
__constant__ uint32_t nrf_split_neg[2] = {0x0000BF80u, 0xBF800000u};
__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo) {
    hi = pack2(a, b);
#if NRF_X3_SPLIT_DOT2
    const bf16x2 h = __builtin_bit_cast(bf16x2, hi);
    lo = pack2(__builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, nrf_split_neg[0]), a, false),
               __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, nrf_split_neg[1]), b, false));
#else
    lo = pack2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
#endif
}

// Training mode (nrf_query_fwd_train_bf16x3): where a wave's 32-point tile saves what the backward pass needs, in the f32
// T8 workspace layout of the exact-f32 training kernels (TrainLayout, mlp_layout.h) -- so nrf_query_bwd consumes it as it is.
struct X3Save {
    TBuf tb;          // this wave's tile of `acts` (zero records for a tile beyond the workspace: stores are dropped)
    int lane;
};

#ifndef NRF_X3_QUAD_STORES
#define NRF_X3_QUAD_STORES 1
#endif

// a[i] of lane j (j = lane & 3)  <-  a[j] of lane i of the same lane quad: two rounds of DPP exchanges
__device__ __forceinline__ void quad_transpose(uint32_t (&a)[4], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {
        const uint32_t s = o1 ? a[p] : a[p + 1];
        const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
        a[p] = o1 ? got : a[p];
        a[p + 1] = o1 ? a[p + 1] : got;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t s = o2 ? a[p] : a[p + 2];
        const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
        a[p] = o2 ? got : a[p];
        a[p + 2] = o2 ? a[p + 2] : got;
    }
}

// 16-byte stores one call of store_tile_t8_quads (4 a tile) or store_tile_p16 (2 a tile) issues for nt tiles (the allowance of
// stream_frags: a LOWER bound is safe, an overestimate is not)
#ifndef NRF_X3_STORE_ALLOWANCE
#define NRF_X3_STORE_ALLOWANCE 1
#endif
constexpr int x3_tile_stores(int nt, bool p16 = false) { return (NRF_X3_QUAD_STORES && NRF_X3_STORE_ALLOWANCE) ? (p16 ? 2 : 4) * nt : 0; }

// One accumulator tile (16 values a lane, register r = channel 32t + (r&3) + 8(r>>2) + 4h of point lane&31) to its T8
// section.  A T8 line is four channels of eight points: the four registers of a group (r>>2) of a lane quad hold a 4x4
// block of it, so after a transpose inside the quad every lane owns four consecutive points of one channel and the block
// goes out as ONE 16-byte store a lane instead of four 4-byte ones -- a quarter of the store instructions, which is what
// the conversion phase of the 3-term kernels waits on.
__device__ __forceinline__ void store_tile_t8_quads(__amdgpu_buffer_rsrc_t rsrc, int lane, int C, uint32_t sec, int t,
                                                    const float (&r)[16]) {
    const int q = lane & 31, h = lane >> 5;
    const int voff = (((q >> 3) * C + 4 * h + (q & 3)) * 8 + (q & 4)) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t a[4] = {__float_as_uint(r[4 * g]), __float_as_uint(r[4 * g + 1]), __float_as_uint(r[4 * g + 2]),
                         __float_as_uint(r[4 * g + 3])};
        quad_transpose(a, lane);
        const u32x4 v = {a[0], a[1], a[2], a[3]};
        buffer_store_b128(v, rsrc, voff, (int)((sec + (32 * t + 8 * g) * 8) * 4));
    }
}

// The f16 sections (TrainLayout::half_acts / half_grads, training modes "a16" / "w16") are POINT-major ("P16"): per 32-point
// tile and 32-channel tile ct two 1 KiB blocks j = 0, 1 of [point n (32)][lane half h (2)][8 values], value e of (ct, j, h)
// being accumulator register 8 j + e of that lane half, i.e. channel 32 ct + ((8j+e) & 3) + 8 ((8j+e) >> 2) + 4 h:
//     byte offset in the section = ((2 ct + j) * 32 + n) * 32 + 16 h + 2 e,    section = 64 C bytes.
// A lane's eight registers 8j .. 8j+7 ARE one 16-byte piece and a wave's store is one contiguous KiB: no transpose, two
// stores a tile (the channel-major T8 form cost a 4x4 transpose per lane quad and register group: 8 operations even on packed
// pairs -- a quarter of the training conversion's VALU work -- and four 8-byte stores).  The price is paid where there is
// slack: the weight-gradient GEMMs, which contract over POINTS, read their operands with ds_read_b64_tr_b16 (mlp_bwd.hip).
// nonneg: the values went through a ReLU, so the saturation at 65504 (an infinity among the saved activations would turn a
// weight gradient into NaN) is one unsigned 16-bit min per pair; otherwise a clamp in f32 first.
__device__ __forceinline__ void store_tile_p16(__amdgpu_buffer_rsrc_t rsrc, int lane, uint32_t sec, int t, const float (&r)[16],
                                               bool nonneg) {
    const int voff = (lane & 31) * 32 + (lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint32_t p[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = r[8 * j + 2 * w], b = r[8 * j + 2 * w + 1];
            if (nonneg) p[w] = sat2h(pack2h(a, b));
            else p[w] = pack2h(__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f));
        }
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        buffer_store_b128(v, rsrc, voff, (int)(sec * 4 + (2 * t + j) * 1024));
    }
}

// sec: section (floats from the tile base; f32 T8 or f16 P16) the values are saved to; gate_slot >= 0: also the ReLU gate bits
// (bit order of store_gate_bits).  The stores sit in the conversion phase, where the matrix pipe is idle anyway.
// TRAIN: 0 inference, 1 save f32 (the exact kernels' workspace format), 2 save f16 (half_acts)
template <int NT, int TRAIN = 0>
__device__ __forceinline__ void pack_acts_x3(u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT], const f32x16 (&acc)[NT], bool relu,
                                             const X3Save *sv = nullptr, uint32_t sec = 0, int gate_slot = -1) {
    uint32_t gm[4] = {0u, 0u, 0u, 0u};
    int voff = 0;
    if constexpr (TRAIN) {
        voff = ((sv->tb.m * (32 * NT) + 4 * sv->tb.h) * 8 + sv->tb.q7) * 4;
        asm volatile("" : "+s"(sec));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r[k]) : "a"(acc[t][k]));
        __builtin_amdgcn_sched_barrier(0);
        // ReLU as ONE instruction per value: fmaxf() on a value the compiler knows nothing about (it comes out of the asm
        // above) is preceded by a canonicalising v_max_f32 x, x, x -- twice the work
        if (relu) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm("v_max_f32_e32 %0, 0, %1" : "=v"(r[k]) : "v"(r[k]));
        }
        if constexpr (TRAIN) {
#if NRF_X3_QUAD_STORES
            if constexpr (TRAIN == 2) store_tile_p16(sv->tb.rsrc, sv->lane, sec, t, r, relu);
            else store_tile_t8_quads(sv->tb.rsrc, sv->lane, 32 * NT, sec, t, r);
#else
            static_assert(TRAIN != 2, "the f16 workspace is written by the quad stores");
#pragma unroll
            for (int k = 0; k < 16; ++k)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r[k]), sv->tb.rsrc, voff,
                                                      (int)((sec + (32 * t + 8 * (k >> 2) + (k & 3)) * 8) * 4), 0);
#endif
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t hi, lo;
            split2(r[2 * w], r[2 * w + 1], hi, lo);
            hh[2 * t + (w >> 2)][w & 3] = hi;
            hl[2 * t + (w >> 2)][w & 3] = lo;
        }
        if constexpr (TRAIN) {
            // gate bit = the value's leading bf16 is non-zero (h > 0 after the ReLU): one packed min and one dot product per
            // PAIR, the dot's 16-bit weights being the pair's bit positions (store_gate_bits' order) -- half the operations of
            // a min and a shift-or per value
            if (gate_slot >= 0) {
                uint32_t bits = (t & 1) ? gm[t >> 1] : 0u;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const u16x2v m = __builtin_elementwise_min(__builtin_bit_cast(u16x2v, (uint32_t)hh[2 * t + (w >> 2)][w & 3]), (u16x2v){1, 1});
                    bits = __builtin_amdgcn_udot2(m, (u16x2v){(unsigned short)(1u << (15 - 2 * w)), (unsigned short)(1u << (14 - 2 * w))}, bits, false);
                }
                gm[t >> 1] = (t & 1) ? bits : (bits << 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TRAIN) {
        if (gate_slot >= 0) {
            u32x4 v = {gm[0], gm[1], gm[2], gm[3]};
            buffer_store_b128(v, sv->tb.rsrc, sv->lane * 16, gate_slot * 4);
        }
    }
}

// ---- conversion UNDER the next GEMM ("piped" layers) ---------------------------------------------------------------------
// Between two GEMMs of the chain a wave has ~150-200 VALU operations per 32-channel tile to do (accumulator reads, ReLU,
// hi/residual split; in training also the gate bits, the f16 conversion, a transpose and the store), and with one wave per
// SIMD nothing else runs meanwhile: the matrix pipe idled through 26 % (inference) to 48 % (training) of these kernels.
// But k-steps 2t, 2t+1 of the next GEMM only need tile t of this one.  So the next GEMM keeps its OWN accumulator set (two sets
// = all 256 accumulation registers), starts as soon as tile 0 is converted, and tile t+1 is converted in micro-slices of <= 6
// VALU operations placed after each of the 6 * NT_OUT MFMAs that consume tile t (measured: six independent VALU operations
// fit under a 32x32x16 MFMA at no cost, `scripts/microbench/mfma_fill.hip`, `profiles/r3_mfma_fill_microbench.txt`).  Only tile 0's conversion stays exposed (hidden layers) or
// runs under the encoded-input k-steps (skip layer).
#ifndef NRF_X3_PIPE
#define NRF_X3_PIPE 0
#endif
constexpr int kConvSlices = 48;      // per tile: 4 register groups x 12 stages

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


// the quad transpose of quad_transpose() in four parts of four operations (part 0, 1: first round; 2, 3: second round)
__device__ __forceinline__ void quad_transpose_part(uint32_t (&a)[4], bool o1, bool o2, int part) {
    if (part < 2) {
        const int p = 2 * part;
        const uint32_t s = o1 ? a[p] : a[p + 1];
        const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0xB1, 0xF, 0xF, true);
        a[p] = o1 ? got : a[p];
        a[p + 1] = o1 ? a[p + 1] : got;
    } else {
        const int p = part - 2;
        const uint32_t s = o2 ? a[p] : a[p + 2];
        const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x4E, 0xF, 0xF, true);
        a[p] = o2 ? got : a[p];
        a[p + 2] = o2 ? a[p + 2] : got;
    }
}

// State of a forward conversion across its micro-slices (registers; only one or two groups' worth is live at a time).
// TRAIN as in pack_acts_x3; RELU: the layer has one; GATE: its gate bits are saved (training).
template <int TRAIN, bool RELU, bool GATE>
struct ConvFwd {
    static constexpr int kTrain = TRAIN;
    static constexpr bool kGate = GATE && TRAIN != 0;
    static constexpr int kStoreStage = TRAIN == 2 ? 10 : 11;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t sec;        // section (floats from the tile base): f32 T8 (TRAIN 1) or f16 P16 (TRAIN 2)
    uint32_t gate_slot;
    int C;               // channels of the section
    int lane, voff;
    bool o1, o2;
    float r[16], ta[4], tb[4];
    uint32_t p[8], gm[4], bits;

    __device__ __forceinline__ void init(const X3Save *sv, uint32_t sec_, int gate_slot_, int C_) {
        if constexpr (TRAIN != 0) {
            rsrc = sv->tb.rsrc;
            lane = sv->lane;
            const int q = lane & 31, h = lane >> 5;
            voff = TRAIN == 2 ? q * 32 + h * 16 : (((q >> 3) * C_ + 4 * h + (q & 3)) * 8 + (q & 4)) * 4;
            o1 = lane & 1; o2 = lane & 2;
            sec = sec_;
            asm volatile("" : "+s"(sec));
            gate_slot = (uint32_t)gate_slot_;
            gm[0] = gm[1] = gm[2] = gm[3] = 0u;
        }
        C = C_;
    }
    template <int G>
    __device__ __forceinline__ void relu4() {
        if constexpr (RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) asm("v_max_f32_e32 %0, 0, %1" : "=v"(r[4 * G + k]) : "v"(r[4 * G + k]));
        }
    }
    template <int G>
    __device__ __forceinline__ void cvt_h(uint32_t &p0, uint32_t &p1) {
        if constexpr (RELU) {
            p0 = sat2h(pack2h(r[4 * G], r[4 * G + 1]));
            p1 = sat2h(pack2h(r[4 * G + 2], r[4 * G + 3]));
        } else {
            p0 = pack2h(__builtin_amdgcn_fmed3f(r[4 * G], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(r[4 * G + 1], -65504.0f, 65504.0f));
            p1 = pack2h(__builtin_amdgcn_fmed3f(r[4 * G + 2], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(r[4 * G + 3], -65504.0f, 65504.0f));
        }
    }
    // stores issued by micro-slice u of tile T (of NT_IN tiles)
    static constexpr int stores(int T, int u, int NT_IN) {
        if (TRAIN == 0) return 0;
        const bool st = TRAIN == 2 ? (u % 12 == 10 && ((u / 12) & 1)) : u % 12 == kStoreStage;
        return (st ? 1 : 0) + (kGate && T == NT_IN - 1 && u == kConvSlices - 1 ? 1 : 0);
    }
    static constexpr int stores_per_tile() { return TRAIN == 0 ? 0 : (TRAIN == 2 ? 2 : 4); }
};

// micro-slice U (0..47) of the conversion of accumulator tile T into operand k-steps 2T, 2T+1 (hh, hl).  T and U are template
// arguments: every register index below must be a constant in the front end already (left to the optimiser, the operand
// arrays went to scratch memory)
template <int NT, int T, int U, class CV>
__device__ __forceinline__ void conv_slice(CV &c, u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT], const f32x16 (&acc)[NT]) {
    constexpr int TRAIN = CV::kTrain;
    constexpr int g = U / 12, stg = U % 12;
    if constexpr (stg == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(c.r[4 * g + k]) : "a"(acc[T][4 * g + k]));
    } else if constexpr (stg == 1) {
        c.template relu4<g>();
    } else if constexpr (stg >= 2 && stg <= 5) {
        constexpr int q = (stg - 2) >> 1, w = 2 * g + q;
        if constexpr (((stg - 2) & 1) == 0) {
            const uint32_t hi = pack2(c.r[4 * g + 2 * q], c.r[4 * g + 2 * q + 1]);
            hh[2 * T + (w >> 2)][w & 3] = hi;
            c.ta[g] = __uint_as_float(hi << 16);
            c.tb[g] = __uint_as_float(hi & 0xffff0000u);
        } else {
            hl[2 * T + (w >> 2)][w & 3] = pack2(c.r[4 * g + 2 * q] - c.ta[g], c.r[4 * g + 2 * q + 1] - c.tb[g]);
        }
    } else if constexpr (stg == 6) {
        if constexpr (CV::kGate) {
            // gate bit = the value's leading bf16 is non-zero (h > 0): one packed min and one dot product per PAIR, the
            // dot's 16-bit weights being the pair's bit positions (tile t, register k -> dword t >> 1, bit 31 - (16 (t & 1) + k))
            if constexpr (g == 0) c.bits = (T & 1) ? c.gm[T >> 1] : 0u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int w = 2 * g + q;
                const u16x2v m = __builtin_elementwise_min(__builtin_bit_cast(u16x2v, (uint32_t)hh[2 * T + (w >> 2)][w & 3]), (u16x2v){1, 1});
                c.bits = __builtin_amdgcn_udot2(m, (u16x2v){(unsigned short)(1u << (15 - 2 * w)), (unsigned short)(1u << (14 - 2 * w))},
                                                c.bits, false);
            }
            if constexpr (g == 3) c.gm[T >> 1] = (T & 1) ? c.bits : (c.bits << 16);
        }
    } else if constexpr (TRAIN == 2) {
        // P16 (store_tile_p16): registers 8j .. 8j+7 = groups 2j, 2j+1 are one 16-byte piece
        uint32_t &p0 = c.p[2 * g], &p1 = c.p[2 * g + 1];
        if constexpr (stg == 7) {
            c.template cvt_h<g>(p0, p1);
        } else if constexpr (stg == 10 && (g & 1)) {
            const u32x4 v = {c.p[2 * g - 2], c.p[2 * g - 1], p0, p1};
            buffer_store_b128(v, c.rsrc, c.voff, (int)(c.sec * 4 + (2 * T + (g >> 1)) * 1024));
        }
    } else if constexpr (TRAIN == 1) {
        uint32_t a[4] = {__float_as_uint(c.r[4 * g]), __float_as_uint(c.r[4 * g + 1]), __float_as_uint(c.r[4 * g + 2]),
                         __float_as_uint(c.r[4 * g + 3])};
        if constexpr (stg <= 10) {
            quad_transpose_part(a, c.o1, c.o2, stg - 7);
#pragma unroll
            for (int k = 0; k < 4; ++k) c.r[4 * g + k] = __uint_as_float(a[k]);
        } else {
            const u32x4 v = {a[0], a[1], a[2], a[3]};
            buffer_store_b128(v, c.rsrc, c.voff, (int)((c.sec + (32 * T + 8 * g) * 8) * 4));
        }
    }
    if constexpr (CV::kGate && U == kConvSlices - 1 && T == NT - 1) {
        const u32x4 v = {c.gm[0], c.gm[1], c.gm[2], c.gm[3]};
        buffer_store_b128(v, c.rsrc, c.lane * 16, (int)(c.gate_slot * 4));
    }
}

// Where the micro-slices of a piped GEMM sit.  The GEMM has S_PRE k-steps that do not depend on the conversion (encoded
// inputs of the skip layer), then the 2 * NT_IN k-steps of the converted tiles, then S_POST (bias).  MFMA e (0, 1: the hi
// fragment's two; 2: the residual fragment's) of k-step s, output tile t is followed by the slices [lo, hi) of tile T:
//   pre region (S_PRE > 0): tile 0 spread evenly over its 3 * S_PRE * NT_OUT MFMAs;
//   k-steps of tile T - 1: tile T spread evenly over their 6 * NT_OUT MFMAs.
template <int NT_OUT, int NT_IN, int S_PRE>
struct ConvMap {
    int T, lo, hi;
    constexpr ConvMap(int s, int t, int e) : T(0), lo(0), hi(0) {
        if (S_PRE > 0 && s < S_PRE) {
            const int M = 3 * (S_PRE > 0 ? S_PRE : 1) * NT_OUT, mo = 3 * (s * NT_OUT + t) + e;
            T = 0; lo = mo * kConvSlices / M; hi = (mo + 1) * kConvSlices / M;
        } else {
            const int hs = s - S_PRE, M = 6 * NT_OUT;
            if (hs / 2 + 1 < NT_IN) {
                const int mo = (hs & 1) * 3 * NT_OUT + 3 * t + e;
                T = hs / 2 + 1; lo = mo * kConvSlices / M; hi = (mo + 1) * kConvSlices / M;
            }
        }
    }
};
// stores issued in the bodies of fragments [0, frag) of such a GEMM (fragment i: pair i >> 1 = (s, t); even: MFMAs 0, 1; odd: 2)
template <class CV, int NT_OUT, int NT_IN, int S_PRE>
struct ConvStores {
    static constexpr int exposed = S_PRE == 0 ? CV::stores_per_tile() : 0;   // tile 0's, issued just before fragment 0
    static constexpr int before(int frag) {
        int n = 0;
        for (int i = 0; i < frag; ++i) {
            const int j = i >> 1, s = j / NT_OUT, t = j % NT_OUT;
            for (int e = (i & 1) ? 2 : 0; e < ((i & 1) ? 3 : 2); ++e) {
                const ConvMap<NT_OUT, NT_IN, S_PRE> m(s, t, e);
                for (int u = m.lo; u < m.hi; ++u) n += CV::stores(m.T, u, NT_IN);
            }
        }
        return n;
    }
};
struct NoStores {
    static constexpr int exposed = 0;
    static constexpr int before(int) { return 0; }
};

// stream_frags with the store allowance of every publish computed from SC (stores issued by the bodies, per fragment): at the
// publish of chunk c the awaited copies are those of chunk c + 1, requested two chunks earlier; every store issued since the
// last of them may stay in flight beside the four copies of chunk c + 2.  In a layer's first two chunks those copies were
// requested before the layer began: all of its stores so far, and the `exposed` ones just before it, are younger.
template <int N, class SC, class ST, class Body>
__device__ __forceinline__ void stream_frags_sc(ST &st, Body &&body) {
    constexpr int kAhead = ST::kDepth;
    static_assert(kStages == 4, "the store allowance assumes copies two chunks ahead");
    static_assert(N % kAhead == 0 && N >= kAhead, "fragment count of a layer must be a multiple of the read-ahead");
    static_assert(N % 16 == 0 || N % 16 >= kAhead, "a layer's last chunk must hold at least the read-ahead");
    constexpr int NCH = (N + 15) / 16;
    static_for<NCH>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        constexpr int F = (N - 16 * c) < 16 ? (N - 16 * c) : 16;          // fragments in this chunk
        constexpr int pos = (F / 2 < F - kAhead) ? F / 2 : F - kAhead;    // publish before the first read of the next chunk
        constexpr int span = F - pos, stp = span >= 8 ? 2 : 1;
        // chunk c - 2 is a full one: its publish at 8, its copy requests at 8, 10, 12, 14
        constexpr int pos2 = (8 < 16 - kAhead) ? 8 : 16 - kAhead;
        constexpr int last2 = (16 - pos2 >= 8) ? pos2 + 6 : ((16 - pos2 >= 4) ? pos2 + 3 : pos2);
        constexpr int young = c >= 2 ? SC::before(16 * c + pos) - SC::before(16 * (c - 2) + last2)
                                     : SC::exposed + SC::before(16 * c + pos);
        static_for<F>([&](auto f_) {
            constexpr int f = decltype(f_)::value, i = 16 * c + f;
            if constexpr (f == pos) st.template publish<young>();
            if constexpr (span >= 4) {
                if constexpr (f >= pos && (f - pos) % stp == 0 && (f - pos) / stp < 4) st.request_piece((f - pos) / stp);
            } else if constexpr (f == pos) {
                st.request();
            }
            const bf16x8 a = st.q[i % kAhead];
            {
                constexpr int j = i + kAhead;
                if constexpr (j < N) st.q[i % kAhead] = st.read((j >> 4) != c, j & 15);
                else st.q[i % kAhead] = st.read(true, j - N);          // the next layer's first fragments
            }
            __builtin_amdgcn_sched_barrier(0);
            body(f_, std::integral_constant<int, i>{}, a);
            __builtin_amdgcn_sched_barrier(0);
        });
        st.rotate();
    });
}

// gemm_layer_x3 with the conversion of its hidden input (accumulator tiles `src` -> hh, hl, by `cv`) running under its own
// MFMAs.  k-steps: S_PRE from (PreH, PreL), then hh/hl[0 .. 2 NT_IN), then the bias step if BIAS_LAST.
template <int NT_OUT, int NT_IN, int S_PRE, bool BIAS_LAST, class CV, class PH, class PL>
__device__ __forceinline__ void gemm_layer_x3_conv(f32x16 (&acc)[NT_OUT], WStreamX3 &st, CV &cv, u32x4v (&hh)[2 * NT_IN],
                                                   u32x4v (&hl)[2 * NT_IN], const f32x16 (&src)[NT_IN], PH &&PreH, PL &&PreL,
                                                   const u32x4v &one, const u32x4v &none) {
    constexpr int S = S_PRE + 2 * NT_IN + (BIAS_LAST ? 1 : 0);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (S_PRE == 0) {
        // tile 0: nothing to run under; four groups interleaved (a dependent VALU chain issues at a third of the rate)
        static_for<12>([&](auto stg_) {
            static_for<4>([&](auto g_) { conv_slice<NT_IN, 0, 12 * decltype(g_)::value + decltype(stg_)::value>(cv, hh, hl, src); });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    auto B = [&](auto s_, auto lo_) -> bf16x8 {
        constexpr int s = decltype(s_)::value;
        constexpr bool lo = decltype(lo_)::value;
        if constexpr (s < S_PRE) return lo ? PreL(s) : PreH(s);
        else if constexpr (s < S_PRE + 2 * NT_IN) return as_bf(lo ? hl[s - S_PRE] : hh[s - S_PRE]);
        else return as_bf(lo ? none : one);
    };
    auto fill = [&](auto s_, auto t_, auto e_) {
        constexpr ConvMap<NT_OUT, NT_IN, S_PRE> m(decltype(s_)::value, decltype(t_)::value, decltype(e_)::value);
        constexpr int T = m.T, lo = m.lo, n = m.hi - m.lo;
        static_for<n>([&](auto k_) { conv_slice<NT_IN, T, lo + decltype(k_)::value>(cv, hh, hl, src); });
        __builtin_amdgcn_sched_barrier(0);
    };
    using std::integral_constant;
    stream_frags_sc<2 * S * NT_OUT, ConvStores<CV, NT_OUT, NT_IN, S_PRE>>(st, [&](auto, auto i_, const bf16x8 &a) {
        constexpr int i = decltype(i_)::value, j = i >> 1, s = j / NT_OUT, t = j % NT_OUT;
        const integral_constant<int, s> sc;
        const integral_constant<int, t> tc;
        if constexpr ((i & 1) == 0) {
            if constexpr (s == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(sc, std::false_type{}), zero, 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(sc, std::false_type{}), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fill(sc, tc, integral_constant<int, 0>{});
            if constexpr (!(BIAS_LAST && s == S - 1)) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(sc, std::true_type{}), acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fill(sc, tc, integral_constant<int, 1>{});
            }
        } else {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B(sc, std::false_type{}), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fill(sc, tc, integral_constant<int, 2>{});
        }
    });
}

// gamma(v) of a 3-vector (slot order of bf_pe3_chan), every frequency by an accurate sincos (nrf_sincos): the recurrence of
// fill_pe3_bf16 is only good to ~2^-16
template <int KS>
__device__ __forceinline__ void fill_pe3_x3_v(u32x4v (&oh)[KS], u32x4v (&ol)[KS], float (&v)[8 * KS], float x, float y, float z,
                                              int h) {
    constexpr int NF = (8 * KS - 2) / 6;
    v[0] = h ? z : x;
    v[1] = h ? 1.0f : y;
    const float co[3] = {x, y, z};
    sincos_dispatch(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) * pow2i(2 * NF - 1), [&](auto sc_fn) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const float sc = pow2i(2 * i + h);
#pragma unroll
            for (int d = 0; d < 3; ++d) sc_fn(co[d] * sc, &v[2 + 6 * i + d], &v[2 + 6 * i + 3 + d]);
        }
    });
#pragma unroll
    for (int i = 2 + 6 * NF; i < 8 * KS; ++i) v[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t hi, lo;
            split2(v[8 * k + 2 * w], v[8 * k + 2 * w + 1], hi, lo);
            oh[k][w] = hi;
            ol[k][w] = lo;
        }
}
template <int KS>
__device__ __forceinline__ void fill_pe3_x3(u32x4v (&oh)[KS], u32x4v (&ol)[KS], float x, float y, float z, int h) {
    float v[8 * KS];
    fill_pe3_x3_v<KS>(oh, ol, v, x, y, z, h);
}

// Training mode: gamma(v) into the T8 input section in the exact-f32 kernels' k-step order (blk_chan BLK_PE3 with KS32
// steps: steps 6j..6j+5 = sin(xyz), cos(xyz) of frequency 2j + h, step KS32-2 = x | y, step KS32-1 = z | pad), from the
// slot-ordered values of fill_pe3_x3_v (slot 2 + 6j + r <-> step 6j + r: both lists give lane half h the frequencies 2j + h).
// Frequencies >= L are stored as computed: the weight-gradient unpack drops those columns (blk_chan returns -1).
template <int KS, int KS32, bool HALF = false>
__device__ __forceinline__ void save_pe3_t8(const X3Save &sv, uint32_t sec, const float (&v)[8 * KS], float x, float y, float z) {
    constexpr int NF = (8 * KS - 2) / 6;
    static_assert(6 * NF <= KS32 - 2, "the f32 k-step list must hold every frequency pair");
    const int h = sv.tb.h;
    asm volatile("" : "+s"(sec));
    auto val = [&](int st) {
        float r = 0.0f;
        if (st < 6 * NF) r = v[2 + st];
        else if (st == KS32 - 2) r = h ? y : x;
        else if (st == KS32 - 1) r = h ? 0.0f : z;
        return r;
    };
    if constexpr (HALF) {
        // P16 (store_tile_p16): the lane's k-steps 8 pi .. 8 pi + 7 are one 16-byte piece, block (ct, j) = (pi >> 1, pi & 1);
        // the column of the weight-gradient GEMM that reads it is k-step-space channel 2 st + h (wgrad_kernel's epilogue).
        // Encoder values lie in [-1, 1], raw coordinates are scene-sized: no saturation needed
        static_assert(KS32 % 16 == 0, "whole 32-channel tiles");
        const int voff = (sv.lane & 31) * 32 + (sv.lane >> 5) * 16;
#pragma unroll
        for (int pi = 0; pi < KS32 / 8; ++pi) {
            const u32x4 p = {pack2h(val(8 * pi), val(8 * pi + 1)), pack2h(val(8 * pi + 2), val(8 * pi + 3)),
                             pack2h(val(8 * pi + 4), val(8 * pi + 5)), pack2h(val(8 * pi + 6), val(8 * pi + 7))};
            buffer_store_b128(p, sv.tb.rsrc, voff, (int)(sec * 4 + pi * 1024));
        }
    } else {
        const int voff = ((sv.tb.m * (2 * KS32) + h) * 8 + sv.tb.q7) * 4;
#pragma unroll
        for (int st = 0; st < KS32; ++st)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val(st)), sv.tb.rsrc, voff, (int)((sec + 2 * st * 8) * 4), 0);
    }
}

// Trunk of one net in bf16x3: layer 0 on S0 encoded k-steps (B0h, B0l), then the hidden layers.
template <int NT, int S0, int TRAIN = 0, class B0H, class B0L>
__device__ __forceinline__ void bf_trunk_x3(f32x16 (&acc)[NT], u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT],
                                            const u32x4v (&xh)[kBfKsX], const u32x4v (&xl)[kBfKsX], const u32x4v &one,
                                            const u32x4v &none, WStreamX3 &st, int n_trunk, int skip_in_mask, B0H &&b0h,
                                            B0L &&b0l, const X3Save *sv = nullptr, uint32_t a_trunk = 0, int a_mask = 0) {
    gemm_layer_x3<NT, S0>(acc, st, b0h, b0l);
    pack_acts_x3<NT, TRAIN>(hh, hl, acc, true, sv, a_trunk, a_mask);
    for (int l = 1; l < n_trunk; ++l) {
        constexpr int kSt = TRAIN ? x3_tile_stores(NT, TRAIN == 2) : 0;   // the layer before saved its activations
        if ((skip_in_mask >> l) & 1)
            gemm_layer_x3<NT, kBfKsX + 2 * NT, false, kSt>(
                acc, st,
                [&](int s) { return s < kBfKsX ? as_bf(xh[s < kBfKsX ? s : 0]) : as_bf(hh[s >= kBfKsX ? s - kBfKsX : 0]); },
                [&](int s) { return s < kBfKsX ? as_bf(xl[s < kBfKsX ? s : 0]) : as_bf(hl[s >= kBfKsX ? s - kBfKsX : 0]); });
        else
            gemm_layer_x3<NT, 2 * NT + 1, true, kSt>(
                acc, st, [&](int s) { return s < 2 * NT ? as_bf(hh[s < 2 * NT ? s : 0]) : as_bf(one); },
                [&](int s) { return s < 2 * NT ? as_bf(hl[s < 2 * NT ? s : 0]) : as_bf(none); });
        pack_acts_x3<NT, TRAIN>(hh, hl, acc, true, sv, a_trunk + l * ((TRAIN == 2 ? 16 : 32) * 32 * NT), a_mask + l * 256);
    }
}

// The trunk with every layer after the first converting its input under its own MFMAs (gemm_layer_x3_conv): the layers
// alternate between two accumulator sets; the last layer's output is converted in the open (the heads follow).
template <int NT, int S0, int TRAIN = 0, class B0H, class B0L>
__device__ __forceinline__ void bf_trunk_x3_piped(f32x16 (&acc)[NT], f32x16 (&accB)[NT], u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT],
                                                  const u32x4v (&xh)[kBfKsX], const u32x4v (&xl)[kBfKsX], const u32x4v &one,
                                                  const u32x4v &none, WStreamX3 &st, int n_trunk, int skip_in_mask, B0H &&b0h,
                                                  B0L &&b0l, const X3Save *sv = nullptr, uint32_t a_trunk = 0, int a_mask = 0) {
    gemm_layer_x3<NT, S0>(acc, st, b0h, b0l);
    constexpr uint32_t kSec = (TRAIN == 2 ? 16 : 32) * 32 * NT;
    auto layer = [&](f32x16 (&out)[NT], const f32x16 (&in)[NT], int l) __attribute__((always_inline)) {
        ConvFwd<TRAIN, true, true> cv;
        cv.init(sv, a_trunk + (uint32_t)(l - 1) * kSec, a_mask + (l - 1) * 256, 32 * NT);
        if ((skip_in_mask >> l) & 1)
            gemm_layer_x3_conv<NT, NT, kBfKsX, false>(
                out, st, cv, hh, hl, in, [&](int s) { return as_bf(xh[s]); }, [&](int s) { return as_bf(xl[s]); }, one, none);
        else
            gemm_layer_x3_conv<NT, NT, 0, true>(
                out, st, cv, hh, hl, in, [&](int) { return as_bf(one); }, [&](int) { return as_bf(none); }, one, none);
    };
    int l = 1;
    for (; l + 1 < n_trunk; l += 2) {
        layer(accB, acc, l);
        layer(acc, accB, l + 1);
    }
    const uint32_t sec_last = a_trunk + (uint32_t)(n_trunk - 1) * kSec;
    if (l < n_trunk) {
        layer(accB, acc, l);
        pack_acts_x3<NT, TRAIN>(hh, hl, accB, true, sv, sec_last, a_mask + (n_trunk - 1) * 256);
    } else {
        pack_acts_x3<NT, TRAIN>(hh, hl, acc, true, sv, sec_last, a_mask + (n_trunk - 1) * 256);
    }
}

// View-direction head in bf16x3 (network.py:56-65) -> (rgb, alpha) of the wave's point tile, stored by lane half 0.
template <int NT, int TRAIN = 0>
__device__ __forceinline__ void bf_views_head_x3(f32x16 (&acc)[NT], u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT],
                                                 const u32x4v (&dh)[kBfKsD], const u32x4v (&dl)[kBfKsD], const u32x4v &one,
                                                 const u32x4v &none, WStreamX3 &st, const float *hb, int64_t pp,
                                                 int64_t n_points, int h, float *raw, const X3Save *sv = nullptr,
                                                 uint32_t a_feat = 0, uint32_t a_hv = 0, int hv_gate_slot = -1) {
    f32x16 acc1[1];
    constexpr int kSt = TRAIN ? x3_tile_stores(NT, TRAIN == 2) : 0;   // the last trunk layer saved its activations
    gemm_layer_x3<1, 2 * NT, false, kSt>(acc1, st, [&](int s) { return as_bf(hh[s]); }, [&](int s) { return as_bf(hl[s]); });  // network.py:56
    const float alpha = acc1[0][0] + hb[kHbAlpha];
    gemm_layer_x3<NT, 2 * NT + 1, true>(acc, st, [&](int s) { return s < 2 * NT ? as_bf(hh[s < 2 * NT ? s : 0]) : as_bf(one); },
                                  [&](int s) { return s < 2 * NT ? as_bf(hl[s < 2 * NT ? s : 0]) : as_bf(none); });   // network.py:57
    pack_acts_x3<NT, TRAIN>(hh, hl, acc, false, sv, a_feat, -1);
    f32x16 acc2[NT / 2];
    gemm_layer_x3<NT / 2, 2 * NT + kBfKsD, false, kSt>(                                                              // network.py:58-62
        acc2, st,
        [&](int s) { return s < 2 * NT ? as_bf(hh[s < 2 * NT ? s : 0]) : as_bf(dh[s >= 2 * NT ? s - 2 * NT : 0]); },
        [&](int s) { return s < 2 * NT ? as_bf(hl[s < 2 * NT ? s : 0]) : as_bf(dl[s >= 2 * NT ? s - 2 * NT : 0]); });
    u32x4v vh[NT], vl[NT];
    pack_acts_x3<NT / 2, TRAIN>(vh, vl, acc2, true, sv, a_hv, hv_gate_slot);
    gemm_layer_x3<1, NT>(acc1, st, [&](int s) { return as_bf(vh[s]); }, [&](int s) { return as_bf(vl[s]); });         // network.py:64
    if (h == 0 && pp < n_points)
        reinterpret_cast<float4 *>(raw)[pp] = make_float4(acc1[0][0] + hb[kHbRgb], acc1[0][1] + hb[kHbRgb + 1],
                                                          acc1[0][2] + hb[kHbRgb + 2], alpha);
}

template <int NT, bool VIEWS, int TRAIN = 0>
__global__ void __launch_bounds__(256, 1) query_fwd_bf16x3_kernel(const Bf16Args a) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStreamX3 st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);
    const int64_t ntiles = (a.n_points + 127) >> 7;
    u32x4v one, none;   // the constant-1 channel of a bias k-step (element 0 of lane half 0) and its zero residual
    one[0] = h == 0 ? 0x3F80u : 0u; one[1] = 0; one[2] = 0; one[3] = 0;
    none[0] = 0; none[1] = 0; none[2] = 0; none[3] = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pp = tile * 128 + st.wave * 32 + n;
        const int64_t pc = pp < a.n_points ? pp : a.n_points - 1;
        u32x4v xh[kBfKsX], xl[kBfKsX];
        u32x4v dh[kBfKsD], dl[kBfKsD];
        X3Save sv;
        const X3Save *svp = nullptr;
        if constexpr (TRAIN) {
            // the workspace holds ceil(n_points / 32) tiles; a wave whose tile lies beyond it still runs (the weight
            // stream is shared) with a zero-length descriptor, so its stores are dropped by the bounds check
            const int64_t t32 = tile * 4 + st.wave;
            const bool have = t32 < ((a.n_points + 31) >> 5);
            sv.tb = make_tbuf(a.acts + (have ? t32 : 0) * a.tl.a_tile, have ? a.tl.a_tile : 0, lane);
            sv.lane = lane;
            svp = &sv;
        }
        {
            const float *q = a.pts + pc * 3;
            if constexpr (VIEWS) {
                const float *d = a.viewdirs + (pc / a.pts_per_ray) * 3;
                if constexpr (TRAIN) {
                    float v[8 * kBfKsD];
                    fill_pe3_x3_v<kBfKsD>(dh, dl, v, d[0], d[1], d[2], h);
                    save_pe3_t8<kBfKsD, kKsD, TRAIN == 2>(sv, (uint32_t)a.tl.a_din, v, d[0], d[1], d[2]);
                } else {
                    fill_pe3_x3<kBfKsD>(dh, dl, d[0], d[1], d[2], h);
                }
            }
            if constexpr (TRAIN) {
                float v[8 * kBfKsX];
                fill_pe3_x3_v<kBfKsX>(xh, xl, v, q[0], q[1], q[2], h);
                save_pe3_t8<kBfKsX, kKsX, TRAIN == 2>(sv, (uint32_t)a.tl.a_xin, v, q[0], q[1], q[2]);
            } else {
                fill_pe3_x3<kBfKsX>(xh, xl, q[0], q[1], q[2], h);
            }
        }
        f32x16 acc[NT];
        u32x4v hh[2 * NT], hl[2 * NT];
#if NRF_X3_PIPE
        f32x16 accB[NT];
        bf_trunk_x3_piped<NT, kBfKsX, TRAIN>(acc, accB, hh, hl, xh, xl, one, none, st, a.n_trunk, a.skip_in_mask,
                                             [&](int s) { return as_bf(xh[s]); }, [&](int s) { return as_bf(xl[s]); },   // network.py:48-53
                                             svp, (uint32_t)a.tl.a_trunk, (int)a.tl.a_mask);
#else
        bf_trunk_x3<NT, kBfKsX, TRAIN>(acc, hh, hl, xh, xl, one, none, st, a.n_trunk, a.skip_in_mask,
                                       [&](int s) { return as_bf(xh[s]); }, [&](int s) { return as_bf(xl[s]); },   // network.py:48-53
                                       svp, (uint32_t)a.tl.a_trunk, (int)a.tl.a_mask);
#endif
        if constexpr (VIEWS) {
            bf_views_head_x3<NT, TRAIN>(acc, hh, hl, dh, dl, one, none, st, hb, pp, a.n_points, h, a.raw, svp,
                                        (uint32_t)a.tl.a_feat, (uint32_t)a.tl.a_hv, (int)a.tl.a_mask + a.n_trunk * 256);
        } else {
            f32x16 acc1[1];
            gemm_layer_x3<1, 2 * NT>(acc1, st, [&](int s) { return as_bf(hh[s]); }, [&](int s) { return as_bf(hl[s]); });  // network.py:67
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * h + r;   // rows 0-3 on lane half 0, 4-7 on half 1 (registers 0..3)
                if (o < a.output_ch && pp < a.n_points) a.raw[pp * a.output_ch + o] = acc1[0][r] + hb[kHbOut + o];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests in flight must land before the LDS is released
    __builtin_amdgcn_s_barrier();
}

// ---- bf16x3 activation-gradient chain (the dgrad_kernel of mlp_bwd.hip on the bf16 matrix pipe) ------------------------
// A gradient block on its way from one GEMM of the chain to the next: accumulators (or VGPR values) [+ sc * head vector]
// -> ReLU gate -> saved to its section of `gbuf` (f32 T8, or f16 P16 with G16: what the weight-gradient GEMMs read) ->
// hi/residual operands.
// G16 (training mode "w16"): the saved copy is f16 (the chain's gradients are pre-scaled by a power of two so that the
// step's largest |d(loss)/d(raw)| is in [2^10, 2^11); saturating at 65504) -- the chain's own operands keep their hi/residual pair.
template <int NT, bool FROM_ACC, bool GATE, bool AXPY, bool G16 = false>
__device__ __forceinline__ void finish_grad_x3(u32x4v (&hh)[2 * NT], u32x4v (&hl)[2 * NT], const f32x16 (&src)[NT],
                                               const u32x4 &gate, const WBuf &wb, uint32_t vec, float sc, const TBuf &tg,
                                               uint32_t sec, int C) {
    const int voff = ((tg.m * C + 4 * tg.h) * 8 + tg.q7) * 4;
    asm volatile("" : "+s"(sec), "+s"(vec));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float r[16];
        f32x4 hv[4];
        if constexpr (AXPY) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) hv[gq] = wload(wb, wb.h16, (vec + ((t * 4 + gq) * 2) * 4) * 4);
        }
        if constexpr (FROM_ACC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r[k]) : "a"(src[t][k]));
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) r[k] = src[t][k];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AXPY) {
#pragma unroll
            for (int k = 0; k < 16; ++k) r[k] = fmaf(hv[k >> 2][k & 3], sc, r[k]);
        }
        if constexpr (GATE) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int mask = __builtin_amdgcn_sbfe((int)gate[t >> 1], 31 - (16 * (t & 1) + k), 1);
                r[k] = __uint_as_float(__float_as_uint(r[k]) & (uint32_t)mask);
            }
        }
#if NRF_X3_QUAD_STORES
        if constexpr (G16) store_tile_p16(tg.rsrc, (int)(threadIdx.x & 63), sec, t, r, false);
        else store_tile_t8_quads(tg.rsrc, (int)(threadIdx.x & 63), C, sec, t, r);
#else
        static_assert(!G16, "the f16 gradient workspace is written by the quad stores");
#pragma unroll
        for (int k = 0; k < 16; ++k)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r[k]), tg.rsrc, voff,
                                                  (int)((sec + (32 * t + 8 * (k >> 2) + (k & 3)) * 8) * 4), 0);
#endif
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t hi, lo;
            split2(r[2 * w], r[2 * w + 1], hi, lo);
            hh[2 * t + (w >> 2)][w & 3] = hi;
            hl[2 * t + (w >> 2)][w & 3] = lo;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// store_head_tile (mlp_device.h) into an f16 (P16, store_tile_p16) section: rows o < n <= 8 of the 32-channel tile at channel
// c0, zeros elsewhere.  Channel o sits in piece j = 0 of lane half (o >> 2) & 1 at element o & 3.
__device__ __forceinline__ void store_head_tile_h(const TBuf &tb, uint32_t sec, int c0, const float (&vals)[8], int n, int lane) {
    const int voff = (lane & 31) * 32 + (lane >> 5) * 16;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = e < n ? vals[e] : 0.0f, b = e + 4 < n ? vals[e + 4] : 0.0f;
        v[e] = __builtin_amdgcn_fmed3f(tb.h ? b : a, -65504.0f, 65504.0f);
    }
    const u32x4 p = {pack2h(v[0], v[1]), pack2h(v[2], v[3]), 0u, 0u};
    const u32x4 zero = {0u, 0u, 0u, 0u};
    buffer_store_b128(p, tb.rsrc, voff, (int)(sec * 4 + (c0 >> 5) * 2048));
    buffer_store_b128(zero, tb.rsrc, voff, (int)(sec * 4 + (c0 >> 5) * 2048 + 1024));
}

struct DgradX3Args {
    const unsigned *gmax;    // G16: bits of the step's largest |g_raw| (gmax_kernel, mlp_bwd.hip); else unused
    NetPlan plan;            // head-vector offsets into the f32 packed buffer
    TrainLayout tl;
    const float *packed;     // nrf_mlp_pack_weights
    const char *stream;      // nrf_mlp_bf16x3_bwd_pack_weights
    int64_t stream_bytes;
    const float *g_raw, *acts;
    float *gbuf;
    int64_t n_points;
};

// Autograd of network.py:46-69 w.r.t. every layer's pre-activations, written to `gbuf` exactly as dgrad_kernel writes it.
// Four waves share the W^T stream through LDS (the machinery of the query kernels); a wave owns one 32-point tile.
template <int NT, bool VIEWS, bool G16 = false>
__global__ void __launch_bounds__(256, 1) dgrad_x3_kernel(const DgradX3Args a) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const NetPlan &pl = a.plan;
    const TrainLayout &tl = a.tl;
    const int lane = threadIdx.x & 63, n = lane & 31;
    constexpr int W = 32 * NT;
    WStreamX3 st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    WBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, (int)(pl.total_floats * 4), 0x00020000);
    wb.lane16 = lane * 16;
    wb.h16 = (lane >> 5) * 16;
    const int64_t ntiles = (a.n_points + 127) >> 7, n32 = (a.n_points + 31) >> 5;
    const u32x4 nogate = {0u, 0u, 0u, 0u};
    // G16: the whole chain runs on gradients scaled by gs = 2^-E, E the exponent of the step's largest |g_raw| (the chain is
    // linear and a power of two commutes with every rounding in it); the reduction of the weight gradients divides by it
    float gs = 1.0f;
    constexpr int gpw = G16 ? 16 : 32;       // floats per channel of a gradient section
    if constexpr (G16) gs = nrf_grad_scale(*a.gmax);

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // a wave whose 32-point tile lies beyond the workspaces still walks the stream (it is shared); zero-length
        // descriptors drop its stores and return zeros for its loads
        const int64_t t32 = tile * 4 + st.wave;
        const bool have = t32 < n32;
        const int64_t p = t32 * 32 + n;
        const bool live = have && p < a.n_points;
        TBuf ta = make_tbuf(const_cast<float *>(a.acts) + (have ? t32 : 0) * tl.a_tile, have ? tl.a_tile : 0, lane);
        TBuf tg = make_tbuf(a.gbuf + (have ? t32 : 0) * tl.g_tile, have ? tl.g_tile : 0, lane);
        f32x16 acc[NT];
        u32x4v hh[2 * NT], hl[2 * NT];
        u32x4 gate = load_gate_bits(ta, (uint32_t)tl.a_mask + (pl.n_trunk - 1) * 256, lane);
        if constexpr (VIEWS) {
            float4 gr = live ? reinterpret_cast<const float4 *>(a.g_raw)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (G16) { gr.x *= gs; gr.y *= gs; gr.z *= gs; gr.w *= gs; }
            const float hv3[8] = {gr.x, gr.y, gr.z, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (G16) store_head_tile_h(tg, (uint32_t)tl.g_head, 0, hv3, 3, lane);
            else store_head_tile(tg, (uint32_t)tl.g_head, 32, 0, hv3, 3);            // d/d(rgb logits): rgb_linear's wgrad
            {   // through rgb_linear (network.py:64) and the view layer's ReLU: three-term sums, on the VALU in f32
                f32x16 g2[NT / 2];
                zero_acc<NT / 2>(g2);
                axpy_head<NT / 2>(g2, wb, (uint32_t)pl.rgb_vec + 0 * (NT / 2) * 32, gr.x);
                axpy_head<NT / 2>(g2, wb, (uint32_t)pl.rgb_vec + 1 * (NT / 2) * 32, gr.y);
                axpy_head<NT / 2>(g2, wb, (uint32_t)pl.rgb_vec + 2 * (NT / 2) * 32, gr.z);
                const u32x4 hvgate = load_gate_bits(ta, (uint32_t)tl.a_mask + pl.n_trunk * 256, lane);
                u32x4v vh[NT], vl[NT];
                finish_grad_x3<NT / 2, false, true, false, G16>(vh, vl, g2, hvgate, wb, 0u, 0.f, tg, (uint32_t)tl.g_hv, W / 2);
                // through views_linears[0][:, :W] (network.py:58-61) -> d/d(feature)
                gemm_layer_x3<NT, NT, false, x3_tile_stores(NT / 2, G16)>(acc, st, [&](int s) { return as_bf(vh[s]); },
                                                                     [&](int s) { return as_bf(vl[s]); });
            }
            finish_grad_x3<NT, true, false, false, G16>(hh, hl, acc, nogate, wb, 0u, 0.f, tg, (uint32_t)tl.g_feat, W + 32);
            const float ha[8] = {gr.w, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (G16) store_head_tile_h(tg, (uint32_t)tl.g_feat, W, ha, 1, lane);
            else store_head_tile(tg, (uint32_t)tl.g_feat, W + 32, W, ha, 1);          // d/d(alpha): alpha_linear's wgrad
            // through feature_linear and alpha_linear (network.py:56-57) -> d/d(h_{D-1}), then that layer's ReLU
            gemm_layer_x3<NT, 2 * NT, false, x3_tile_stores(NT, G16)>(acc, st, [&](int s) { return as_bf(hh[s]); },
                                                                 [&](int s) { return as_bf(hl[s]); });
            finish_grad_x3<NT, true, true, true, G16>(hh, hl, acc, gate, wb, (uint32_t)pl.alpha_vec, gr.w, tg,
                                                      (uint32_t)tl.g_trunk + (pl.n_trunk - 1) * (gpw * W), W);
        } else {
            float ho[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) ho[o] = (live && o < pl.output_ch) ? a.g_raw[p * pl.output_ch + o] * gs : 0.0f;
            if constexpr (G16) store_head_tile_h(tg, (uint32_t)tl.g_head, 0, ho, pl.output_ch, lane);
            else store_head_tile(tg, (uint32_t)tl.g_head, 32, 0, ho, pl.output_ch);
            f32x16 g[NT];
            zero_acc<NT>(g);
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < pl.output_ch) axpy_head<NT>(g, wb, (uint32_t)pl.out_vec + o * NT * 32, ho[o]);
            finish_grad_x3<NT, false, true, false, G16>(hh, hl, g, gate, wb, 0u, 0.f, tg,
                                                        (uint32_t)tl.g_trunk + (pl.n_trunk - 1) * (gpw * W), W);
        }
        // trunk in reverse (network.py:49-53): g_{l-1} = gate_{l-1} . (W_l[:, hidden]^T g_l)
        for (int l = pl.n_trunk - 1; l >= 1; --l) {
            gate = load_gate_bits(ta, (uint32_t)tl.a_mask + (l - 1) * 256, lane);       // in flight during the GEMM
            gemm_layer_x3<NT, 2 * NT, false, x3_tile_stores(NT, G16)>(acc, st, [&](int s) { return as_bf(hh[s]); },
                                                                 [&](int s) { return as_bf(hl[s]); });
            finish_grad_x3<NT, true, true, false, G16>(hh, hl, acc, gate, wb, 0u, 0.f, tg, (uint32_t)tl.g_trunk + (l - 1) * (gpw * W), W);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests in flight must land before the LDS is released
    __builtin_amdgcn_s_barrier();
}

// gamma(t) of a scalar into KS k-step fragments (slot order of bf_pe1_chan)
template <int KS>
__device__ __forceinline__ void fill_pe1_bf16(u32x4v (&out)[KS], float t, int h) {
    constexpr int NF = (8 * KS - 1) / 2;
    float v[8 * KS];
    v[0] = h ? 0.0f : t;
    float s, c;
    sincos_dispatch(fabsf(t) * 2.0f, [&](auto sc_fn) { sc_fn(t * pow2i(h), &s, &c); });
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        v[1 + 2 * i] = s;
        v[2 + 2 * i] = c;
        if (i + 1 < NF) {
            const float s2 = 2.0f * s * c, c2 = 1.0f - 2.0f * s * s;
            s = 2.0f * s2 * c2;
            c = 1.0f - 2.0f * s2 * s2;
        }
    }
#pragma unroll
    for (int i = 1 + 2 * NF; i < 8 * KS; ++i) v[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) out[k][w] = pack2in(v[8 * k + 2 * w], v[8 * k + 2 * w + 1]);
}

struct DnerfBf16Args {
    const char *stream;           // time net layers, canonical net layers, head-bias table
    int64_t stream_bytes;
    const float *pts, *viewdirs, *times;
    float *raw, *dx;
    int64_t n_points, netchunk;
    int32_t pts_per_ray, t_div, zero_canonical;
    int32_t t_trunk, t_skip_mask, c_trunk, c_skip_mask, pad_;
};

// K6 in bf16: dx = time_net(gamma(x), gamma(t)); raw = occ_net(gamma(x + dx), gamma(d)) -- vanilla_dynamic/network.py:307-341.
// dx and x + dx stay in f32; only the MLP operands are bf16.  With t == 0 for a whole netchunk (network.py:315-322) the
// time net still runs here (the weight stream is one fixed sequence) and its dx is replaced by zero.
template <int NT>
__global__ void __launch_bounds__(256, 1) dnerf_query_fwd_bf16_kernel(const DnerfBf16Args a) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStream st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);
    const int64_t ntiles = (a.n_points + 255) >> 8;
    u32x4v one;
    one[0] = h == 0 ? 0x3F80u : 0u; one[1] = 0; one[2] = 0; one[3] = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t pp[2], pc[2];
        float px[2][3], dv[2][3];
        u32x4v din[2][kBfKsD];
        bool skip[2];
        u32x4v xin[2][kBfKsX], tin[2][kBfKsT];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pp[p] = tile * 256 + st.wave * 64 + p * 32 + n;
            pc[p] = pp[p] < a.n_points ? pp[p] : a.n_points - 1;
            const float *q = a.pts + pc[p] * 3;
            px[p][0] = q[0]; px[p][1] = q[1]; px[p][2] = q[2];
            const int64_t chunk_first = (pc[p] / a.netchunk) * a.netchunk;       // network.py:315-322
            skip[p] = a.zero_canonical && a.times[chunk_first / a.t_div] == 0.0f;
            const float *d = a.viewdirs + (pc[p] / a.pts_per_ray) * 3;
            dv[p][0] = d[0]; dv[p][1] = d[1]; dv[p][2] = d[2];
            fill_pe3_bf16<kBfKsX>(xin[p], px[p][0], px[p][1], px[p][2], h);
            fill_pe1_bf16<kBfKsT>(tin[p], a.times[pc[p] / a.t_div], h);
        }
        f32x16 acc[2][NT];
        u32x4v hin[2][2 * NT];
        bf_trunk<NT, kBfKsX + kBfKsT>(acc, hin, xin, one, st, a.t_trunk, a.t_skip_mask, [&](int p, int s) {
            return s < kBfKsX ? as_bf(xin[p][s < kBfKsX ? s : 0]) : as_bf(tin[p][s >= kBfKsX ? s - kBfKsX : 0]);
        });
        {
            f32x16 acc1[2][1];
            gemm_layer<1, 2 * NT>(acc1, st, [&](int p, int s) { return as_bf(hin[p][s]); });       // network.py:305
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // rows 0..2 sit in registers 0..2 of lane half 0; the other half needs them too (it encodes the
                    // odd frequencies of the same point)
                    d[k] = __shfl(acc1[p][0][k] + hb[kHbTime + k], n);
                    if (skip[p]) d[k] = 0.0f;
                    px[p][k] += d[k];                                                             // network.py:331
                }
                if (h == 0 && pp[p] < a.n_points) {
                    a.dx[pp[p] * 3 + 0] = d[0]; a.dx[pp[p] * 3 + 1] = d[1]; a.dx[pp[p] * 3 + 2] = d[2];
                }
                fill_pe3_bf16<kBfKsX>(xin[p], px[p][0], px[p][1], px[p][2], h);                   // network.py:332-333
                fill_pe3_bf16<kBfKsD>(din[p], dv[p][0], dv[p][1], dv[p][2], h);   // here: gamma(t) is dead, the trunk spare
            }
        }
        bf_trunk<NT, kBfKsX>(acc, hin, xin, one, st, a.c_trunk, a.c_skip_mask, [&](int p, int s) { return as_bf(xin[p][s]); });
        bf_views_head<NT>(acc, hin, one, st, hb, din, pp, pc, a.n_points, a.pts_per_ray, h, a.raw);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// gamma(t) of a scalar in bf16x3 (slot order of bf_pe1_chan), every frequency by an accurate sincosf
template <int KS>
__device__ __forceinline__ void fill_pe1_x3(u32x4v (&oh)[KS], u32x4v (&ol)[KS], float t, int h) {
    constexpr int NF = (8 * KS - 1) / 2;
    float v[8 * KS];
    v[0] = h ? 0.0f : t;
    sincos_dispatch(fabsf(t) * pow2i(2 * NF - 1), [&](auto sc_fn) {
#pragma unroll
        for (int i = 0; i < NF; ++i) sc_fn(t * pow2i(2 * i + h), &v[1 + 2 * i], &v[2 + 2 * i]);
    });
#pragma unroll
    for (int i = 1 + 2 * NF; i < 8 * KS; ++i) v[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t hi, lo;
            split2(v[8 * k + 2 * w], v[8 * k + 2 * w + 1], hi, lo);
            oh[k][w] = hi;
            ol[k][w] = lo;
        }
}

// K6 in bf16x3 (f32-grade accuracy): one 32-point tile per wave, otherwise the structure of dnerf_query_fwd_bf16_kernel.
template <int NT>
__global__ void __launch_bounds__(256, 1) dnerf_query_fwd_bf16x3_kernel(const DnerfBf16Args a) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStreamX3 st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);
    const int64_t ntiles = (a.n_points + 127) >> 7;
    u32x4v one, none;
    one[0] = h == 0 ? 0x3F80u : 0u; one[1] = 0; one[2] = 0; one[3] = 0;
    none[0] = 0; none[1] = 0; none[2] = 0; none[3] = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pp = tile * 128 + st.wave * 32 + n;
        const int64_t pc = pp < a.n_points ? pp : a.n_points - 1;
        float px[3], dv[3];
        u32x4v xh[kBfKsX], xl[kBfKsX], th[kBfKsT], tl[kBfKsT], dh[kBfKsD], dl[kBfKsD];
        const float *q = a.pts + pc * 3;
        px[0] = q[0]; px[1] = q[1]; px[2] = q[2];
        const int64_t chunk_first = (pc / a.netchunk) * a.netchunk;                 // network.py:315-322
        const bool skip = a.zero_canonical && a.times[chunk_first / a.t_div] == 0.0f;
        const float *d = a.viewdirs + (pc / a.pts_per_ray) * 3;
        dv[0] = d[0]; dv[1] = d[1]; dv[2] = d[2];
        fill_pe3_x3<kBfKsX>(xh, xl, px[0], px[1], px[2], h);
        fill_pe1_x3<kBfKsT>(th, tl, a.times[pc / a.t_div], h);
        f32x16 acc[NT];
        u32x4v hh[2 * NT], hl[2 * NT];
        bf_trunk_x3<NT, kBfKsX + kBfKsT>(
            acc, hh, hl, xh, xl, one, none, st, a.t_trunk, a.t_skip_mask,
            [&](int s) { return s < kBfKsX ? as_bf(xh[s < kBfKsX ? s : 0]) : as_bf(th[s >= kBfKsX ? s - kBfKsX : 0]); },
            [&](int s) { return s < kBfKsX ? as_bf(xl[s < kBfKsX ? s : 0]) : as_bf(tl[s >= kBfKsX ? s - kBfKsX : 0]); });
        {
            f32x16 acc1[1];
            gemm_layer_x3<1, 2 * NT>(acc1, st, [&](int s) { return as_bf(hh[s]); }, [&](int s) { return as_bf(hl[s]); });  // network.py:305
            float dd[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dd[k] = __shfl(acc1[0][k] + hb[kHbTime + k], n);   // rows 0..2 live in lane half 0; both halves encode
                if (skip) dd[k] = 0.0f;
                px[k] += dd[k];                                                                     // network.py:331
            }
            if (h == 0 && pp < a.n_points) {
                a.dx[pp * 3 + 0] = dd[0]; a.dx[pp * 3 + 1] = dd[1]; a.dx[pp * 3 + 2] = dd[2];
            }
            fill_pe3_x3<kBfKsX>(xh, xl, px[0], px[1], px[2], h);                                    // network.py:332-333
            fill_pe3_x3<kBfKsD>(dh, dl, dv[0], dv[1], dv[2], h);
        }
        bf_trunk_x3<NT, kBfKsX>(acc, hh, hl, xh, xl, one, none, st, a.c_trunk, a.c_skip_mask,
                                [&](int s) { return as_bf(xh[s]); }, [&](int s) { return as_bf(xl[s]); });
        bf_views_head_x3<NT>(acc, hh, hl, dh, dl, one, none, st, hb, pp, a.n_points, h, a.raw);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// DNeRF_Straightforward in bf16x3 (vanilla_dynamic/network.py:151-196): the static query with a gamma(t) block beside
// gamma(x) at layer 0 and at the skip layer; (rgb, sigma) out, no deformation.
struct StraightX3Args {
    Bf16Args b;
    const float *times;
    int64_t t_div;
};

// The same network on plain bf16 MFMAs (tolerance 1e-2): the structure of query_fwd_bf16_kernel, 64 points per wave.
template <int NT>
__global__ void __launch_bounds__(256, 1) straight_query_fwd_bf16_kernel(const StraightX3Args sa) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const Bf16Args &a = sa.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStream st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);
    const int64_t ntiles = (a.n_points + 255) >> 8;
    u32x4v one;
    one[0] = h == 0 ? 0x3F80u : 0u; one[1] = 0; one[2] = 0; one[3] = 0;
    constexpr int KI = kBfKsX + kBfKsT;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t pp[2], pc[2];
        u32x4v xin[2][kBfKsX], tin[2][kBfKsT], din[2][kBfKsD];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pp[p] = tile * 256 + st.wave * 64 + p * 32 + n;
            pc[p] = pp[p] < a.n_points ? pp[p] : a.n_points - 1;
            const float *q = a.pts + pc[p] * 3;
            const float *d = a.viewdirs + (pc[p] / a.pts_per_ray) * 3;
            fill_pe3_bf16<kBfKsD>(din[p], d[0], d[1], d[2], h);
            fill_pe3_bf16<kBfKsX>(xin[p], q[0], q[1], q[2], h);
            fill_pe1_bf16<kBfKsT>(tin[p], sa.times[pc[p] / sa.t_div], h);
        }
        auto in = [&](int p, int s) { return s < kBfKsX ? as_bf(xin[p][s < kBfKsX ? s : 0]) : as_bf(tin[p][s >= kBfKsX ? s - kBfKsX : 0]); };
        f32x16 acc[2][NT];
        u32x4v hin[2][2 * NT];
        gemm_layer<NT, KI, 0, KI>(acc, st, in);                                                // network.py:168-172
        pack_acts<NT>(hin, acc, true);
        for (int l = 1; l < a.n_trunk; ++l) {
            if ((a.skip_in_mask >> l) & 1)                                                     // network.py:176-177
                gemm_layer<NT, KI + 2 * NT, 0, KI>(acc, st, [&](int p, int s) {
                    return s < KI ? in(p, s < KI ? s : 0) : as_bf(hin[p][s >= KI ? s - KI : 0]);
                });
            else
                gemm_layer<NT, 2 * NT + 1>(acc, st, [&](int p, int s) {
                    return s < 2 * NT ? as_bf(hin[p][s < 2 * NT ? s : 0]) : as_bf(one);
                });
            pack_acts<NT>(hin, acc, true);
        }
        bf_views_head<NT>(acc, hin, one, st, hb, din, pp, pc, a.n_points, a.pts_per_ray, h, a.raw, nullptr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int NT>
__global__ void __launch_bounds__(256, 1) straight_query_fwd_bf16x3_kernel(const StraightX3Args sa) {
    __shared__ __attribute__((aligned(1024))) char lds[kStages * kChunk];
    const Bf16Args &a = sa.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WStreamX3 st;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.stream), 0, (int)a.stream_bytes, 0x00020000);
    st.lds = lds;
    st.total = (uint32_t)a.stream_bytes;
    st.lane16 = lane * 16;
    st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    st.start();
    const float *hb = reinterpret_cast<const float *>(a.stream + a.stream_bytes);
    const int64_t ntiles = (a.n_points + 127) >> 7;
    u32x4v one, none;
    one[0] = h == 0 ? 0x3F80u : 0u; one[1] = 0; one[2] = 0; one[3] = 0;
    none[0] = 0; none[1] = 0; none[2] = 0; none[3] = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pp = tile * 128 + st.wave * 32 + n;
        const int64_t pc = pp < a.n_points ? pp : a.n_points - 1;
        u32x4v xh[kBfKsX], xl[kBfKsX], th[kBfKsT], tl[kBfKsT], dh[kBfKsD], dl[kBfKsD];
        {
            const float *q = a.pts + pc * 3;
            const float *d = a.viewdirs + (pc / a.pts_per_ray) * 3;
            fill_pe3_x3<kBfKsD>(dh, dl, d[0], d[1], d[2], h);
            fill_pe3_x3<kBfKsX>(xh, xl, q[0], q[1], q[2], h);
            fill_pe1_x3<kBfKsT>(th, tl, sa.times[pc / sa.t_div], h);
        }
        auto inh = [&](int s) { return s < kBfKsX ? as_bf(xh[s < kBfKsX ? s : 0]) : as_bf(th[s >= kBfKsX ? s - kBfKsX : 0]); };
        auto inl = [&](int s) { return s < kBfKsX ? as_bf(xl[s < kBfKsX ? s : 0]) : as_bf(tl[s >= kBfKsX ? s - kBfKsX : 0]); };
        constexpr int KI = kBfKsX + kBfKsT;
        f32x16 acc[NT];
        u32x4v hh[2 * NT], hl[2 * NT];
        gemm_layer_x3<NT, KI>(acc, st, inh, inl);                                              // network.py:168-172
        pack_acts_x3<NT>(hh, hl, acc, true);
        for (int l = 1; l < a.n_trunk; ++l) {
            if ((a.skip_in_mask >> l) & 1)                                                     // network.py:176-177
                gemm_layer_x3<NT, KI + 2 * NT>(
                    acc, st, [&](int s) { return s < KI ? inh(s < KI ? s : 0) : as_bf(hh[s >= KI ? s - KI : 0]); },
                    [&](int s) { return s < KI ? inl(s < KI ? s : 0) : as_bf(hl[s >= KI ? s - KI : 0]); });
            else
                gemm_layer_x3<NT, 2 * NT + 1, true>(
                    acc, st, [&](int s) { return s < 2 * NT ? as_bf(hh[s < 2 * NT ? s : 0]) : as_bf(one); },
                    [&](int s) { return s < 2 * NT ? as_bf(hl[s < 2 * NT ? s : 0]) : as_bf(none); });
            pack_acts_x3<NT>(hh, hl, acc, true);
        }
        bf_views_head_x3<NT>(acc, hh, hl, dh, dl, one, none, st, hb, pp, a.n_points, h, a.raw);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace nrf

using namespace nrf;

static int skip_in_mask_of(const nrf_net_t *net) {
    int m = 0;
    for (int i = 1; i < net->D; ++i) if ((net->skips_mask >> (i - 1)) & 1) m |= 1 << i;
    return m;
}

static const char *bf_check_dnerf(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb) {
    if (const char *e = bf_check_net(time_net, emb, true)) return e;
    if (const char *e = bf_check_net(occ_net, emb, false)) return e;
    if (!occ_net->use_viewdirs) return "the canonical net must use view directions";
    if (time_net->W != occ_net->W) return "time and canonical nets must have the same width";
    return nullptr;
}

// D-NeRF entry points, shared by the bf16 (split = 0) and bf16x3 (split = 1) variants
static int64_t dnerf_packed_bytes_impl(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb, int split,
                                       const char *who) {
    if (!time_net || !occ_net || !emb) return -1;
    if (const char *e = bf_check_dnerf(time_net, occ_net, emb)) { nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e); return -1; }
    BfPlan P;
    P.split = split;
    bf_plan_net(P, time_net, emb, true);
    bf_plan_net(P, occ_net, emb, false);
    return bf_total_bytes(P);
}

static int dnerf_pack_impl(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb, void *packed,
                           void *stream, int split, const char *who) {
    NRF_REQUIRE(time_net && occ_net && emb && packed, "%s: null pointer", who);
    if (const char *e = bf_check_dnerf(time_net, occ_net, emb)) return nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "%s: packed must be 16-byte aligned", who);
    BfPlan P;
    P.split = split;
    bf_plan_net(P, time_net, emb, true);
    bf_plan_net(P, occ_net, emb, false);
    return bf_pack(P, packed, as_stream(stream));
}

static int dnerf_query_impl(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb, const void *packed,
                            const float *pts, const float *viewdirs, const float *times, int64_t n_points,
                            int32_t pts_per_ray, int32_t times_per_point, int64_t netchunk, int32_t zero_canonical,
                            float *raw, float *dx, void *stream, int split, const char *who) {
    NRF_REQUIRE(time_net && occ_net && emb, "%s: null descriptor", who);
    if (n_points != 0) NRF_REQUIRE(time_net && occ_net && emb && packed && pts && viewdirs && times && raw && dx,
                "%s: null pointer", who);
    NRF_REQUIRE(n_points >= 0 && pts_per_ray >= 1 && netchunk >= 1, "%s: bad sizes", who);
    NRF_REQUIRE(n_points % pts_per_ray == 0, "%s: n_points must be a multiple of pts_per_ray", who);
    if (const char *e = bf_check_dnerf(time_net, occ_net, emb)) return nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0,
                "%s: packed and raw must be 16-byte aligned", who);
    if (n_points == 0) return NRF_OK;
    BfPlan P;
    P.split = split;
    bf_plan_net(P, time_net, emb, true);
    bf_plan_net(P, occ_net, emb, false);
    DnerfBf16Args a{};
    a.stream = (const char *)packed; a.stream_bytes = P.bytes;
    a.pts = pts; a.viewdirs = viewdirs; a.times = times; a.raw = raw; a.dx = dx;
    a.n_points = n_points; a.netchunk = netchunk; a.pts_per_ray = pts_per_ray;
    a.t_div = times_per_point ? 1 : pts_per_ray; a.zero_canonical = zero_canonical;
    a.t_trunk = time_net->D; a.t_skip_mask = skip_in_mask_of(time_net);
    a.c_trunk = occ_net->D; a.c_skip_mask = skip_in_mask_of(occ_net);
    const int per_wg = split ? 128 : 256;            // points per workgroup pass
    const int64_t ntiles = (n_points + per_wg - 1) / per_wg;
    const int blocks = (int)(ntiles < 256 ? ntiles : 256);
    hipStream_t st = as_stream(stream);
    switch (occ_net->W / 32 + (split ? 100 : 0)) {
        case 8: hipLaunchKernelGGL((dnerf_query_fwd_bf16_kernel<8>), dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((dnerf_query_fwd_bf16_kernel<4>), dim3(blocks), dim3(256), 0, st, a); break;
        case 108: hipLaunchKernelGGL((dnerf_query_fwd_bf16x3_kernel<8>), dim3(blocks), dim3(256), 0, st, a); break;
        case 104: hipLaunchKernelGGL((dnerf_query_fwd_bf16x3_kernel<4>), dim3(blocks), dim3(256), 0, st, a); break;
    }
    return check_launch(who);
}

extern "C" int64_t nrf_dnerf_bf16_packed_bytes(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb) {
    return dnerf_packed_bytes_impl(time_net, occ_net, emb, 0, "nrf_dnerf_bf16_packed_bytes");
}
extern "C" int nrf_dnerf_bf16_pack_weights(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                           void *packed, void *stream) {
    return dnerf_pack_impl(time_net, occ_net, emb, packed, stream, 0, "nrf_dnerf_bf16_pack_weights");
}
extern "C" int nrf_dnerf_query_fwd_bf16(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                        const void *packed, const float *pts, const float *viewdirs, const float *times,
                                        int64_t n_points, int32_t pts_per_ray, int32_t times_per_point, int64_t netchunk,
                                        int32_t zero_canonical, float *raw, float *dx, void *stream) {
    return dnerf_query_impl(time_net, occ_net, emb, packed, pts, viewdirs, times, n_points, pts_per_ray, times_per_point,
                            netchunk, zero_canonical, raw, dx, stream, 0, "nrf_dnerf_query_fwd_bf16");
}
extern "C" int64_t nrf_dnerf_bf16x3_packed_bytes(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb) {
    return dnerf_packed_bytes_impl(time_net, occ_net, emb, 1, "nrf_dnerf_bf16x3_packed_bytes");
}
extern "C" int nrf_dnerf_bf16x3_pack_weights(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                             void *packed, void *stream) {
    return dnerf_pack_impl(time_net, occ_net, emb, packed, stream, 1, "nrf_dnerf_bf16x3_pack_weights");
}
extern "C" int nrf_dnerf_query_fwd_bf16x3(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                          const void *packed, const float *pts, const float *viewdirs, const float *times,
                                          int64_t n_points, int32_t pts_per_ray, int32_t times_per_point, int64_t netchunk,
                                          int32_t zero_canonical, float *raw, float *dx, void *stream) {
    return dnerf_query_impl(time_net, occ_net, emb, packed, pts, viewdirs, times, n_points, pts_per_ray, times_per_point,
                            netchunk, zero_canonical, raw, dx, stream, 1, "nrf_dnerf_query_fwd_bf16x3");
}

extern "C" int64_t nrf_mlp_bf16_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net || !emb) return -1;
    if (const char *e = bf_check_net(net, emb, false)) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16_packed_bytes: %s", e); return -1; }
    BfPlan P;
    bf_plan_net(P, net, emb, false);
    return bf_total_bytes(P);
}

static int pack_16bit_impl(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream, int f16) {
    NRF_REQUIRE(net && emb && packed, "nrf_mlp_bf16_pack_weights: null pointer");
    if (const char *e = bf_check_net(net, emb, false)) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16_pack_weights: %s", e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "nrf_mlp_bf16_pack_weights: packed must be 16-byte aligned");
    BfPlan P;
    P.f16 = f16;
    bf_plan_net(P, net, emb, false);
    return bf_pack(P, packed, as_stream(stream));
}
extern "C" int nrf_mlp_bf16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    return pack_16bit_impl(net, emb, packed, stream, 0);
}
extern "C" int nrf_mlp_f16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    return pack_16bit_impl(net, emb, packed, stream, 1);
}

static int query_fwd_16bit_impl(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream, bool f16) {
    NRF_REQUIRE(net && emb, "nrf_query_fwd_bf16: null descriptor");
    if (n_points != 0) NRF_REQUIRE(net && emb && packed && pts && raw, "nrf_query_fwd_bf16: null pointer");
    NRF_REQUIRE(n_points >= 0, "nrf_query_fwd_bf16: negative n_points");
    if (const char *e = bf_check_net(net, emb, false)) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_fwd_bf16: %s", e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0,
                "nrf_query_fwd_bf16: packed and raw must be 16-byte aligned");
    if (net->use_viewdirs && n_points != 0) {
        NRF_REQUIRE(viewdirs, "nrf_query_fwd_bf16: viewdirs is null but the net uses view directions");
        NRF_REQUIRE(pts_per_ray >= 1 && n_points % pts_per_ray == 0,
                    "nrf_query_fwd_bf16: n_points (%lld) must be a multiple of pts_per_ray (%d)", (long long)n_points, pts_per_ray);
    }
    if (n_points == 0) return NRF_OK;
    BfPlan P;
    bf_plan_net(P, net, emb, false);
    Bf16Args a{};
    a.stream = (const char *)packed; a.stream_bytes = P.bytes;
    a.pts = pts; a.viewdirs = viewdirs; a.raw = raw; a.n_points = n_points;
    a.pts_per_ray = pts_per_ray > 0 ? pts_per_ray : 1;
    a.n_trunk = net->D;
    a.skip_in_mask = 0;
    for (int i = 1; i < net->D; ++i) if ((net->skips_mask >> (i - 1)) & 1) a.skip_in_mask |= 1 << i;
    a.use_viewdirs = net->use_viewdirs; a.output_ch = net->output_ch;
#ifdef NRF_BF16_PROFILE
    a.prof_out = nrf::g_prof_host_ptr;
#endif
    if (const char *d = getenv("NRF_BF16_DEBUG")) a.dbg = atoi(d);   // timing experiments only (1: drop the weight DMA)
    const int64_t ntiles = (n_points + 255) / 256;
    const int blocks = (int)(ntiles < 256 ? ntiles : 256);   // 512 registers per lane: one workgroup per CU
    hipStream_t st = as_stream(stream);
    switch ((net->W / 32) * (f16 ? -1 : 1)) {
        case 8:
            if (a.use_viewdirs) hipLaunchKernelGGL((query_fwd_bf16_kernel<8, true>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((query_fwd_bf16_kernel<8, false>), dim3(blocks), dim3(256), 0, st, a);
            break;
        case 4:
            if (a.use_viewdirs) hipLaunchKernelGGL((query_fwd_bf16_kernel<4, true>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((query_fwd_bf16_kernel<4, false>), dim3(blocks), dim3(256), 0, st, a);
            break;
        case -8:
            if (a.use_viewdirs) hipLaunchKernelGGL((query_fwd_bf16_kernel<8, true, true>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((query_fwd_bf16_kernel<8, false, true>), dim3(blocks), dim3(256), 0, st, a);
            break;
        case -4:
            if (a.use_viewdirs) hipLaunchKernelGGL((query_fwd_bf16_kernel<4, true, true>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((query_fwd_bf16_kernel<4, false, true>), dim3(blocks), dim3(256), 0, st, a);
            break;
    }
    return check_launch(f16 ? "nrf_query_fwd_f16" : "nrf_query_fwd_bf16");
}
extern "C" int nrf_query_fwd_bf16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                  const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream) {
    return query_fwd_16bit_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, stream, false);
}
extern "C" int nrf_query_fwd_f16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                 const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream) {
    return query_fwd_16bit_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, stream, true);
}

extern "C" int64_t nrf_mlp_bf16x3_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net || !emb) return -1;
    if (const char *e = bf_check_net(net, emb, false)) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16x3_packed_bytes: %s", e); return -1; }
    BfPlan P;
    P.split = 1;
    bf_plan_net(P, net, emb, false);
    return bf_total_bytes(P);
}

extern "C" int nrf_mlp_bf16x3_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    NRF_REQUIRE(net && emb && packed, "nrf_mlp_bf16x3_pack_weights: null pointer");
    if (const char *e = bf_check_net(net, emb, false)) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16x3_pack_weights: %s", e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "nrf_mlp_bf16x3_pack_weights: packed must be 16-byte aligned");
    BfPlan P;
    P.split = 1;
    bf_plan_net(P, net, emb, false);
    return bf_pack(P, packed, as_stream(stream));
}

static int query_fwd_x3_impl(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                             const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, float *acts,
                             void *stream, const char *who, int half_acts = 0) {
    NRF_REQUIRE(net && emb, "%s: null descriptor", who);
    if (n_points != 0) NRF_REQUIRE(net && emb && packed && pts && raw, "%s: null pointer", who);
    NRF_REQUIRE(n_points >= 0, "%s: negative n_points", who);
    if (const char *e = bf_check_net(net, emb, false)) return nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0,
                "%s: packed and raw must be 16-byte aligned", who);
    if (net->use_viewdirs && n_points != 0) {
        NRF_REQUIRE(viewdirs, "%s: viewdirs is null but the net uses view directions", who);
        NRF_REQUIRE(pts_per_ray >= 1 && n_points % pts_per_ray == 0,
                    "%s: n_points (%lld) must be a multiple of pts_per_ray (%d)", who, (long long)n_points, pts_per_ray);
    }
    if (n_points == 0) return NRF_OK;
    BfPlan P;
    P.split = 1;
    bf_plan_net(P, net, emb, false);
    Bf16Args a{};
    a.stream = (const char *)packed; a.stream_bytes = P.bytes;
    a.pts = pts; a.viewdirs = viewdirs; a.raw = raw; a.n_points = n_points;
    a.pts_per_ray = pts_per_ray > 0 ? pts_per_ray : 1;
    a.n_trunk = net->D;
    a.skip_in_mask = 0;
    for (int i = 1; i < net->D; ++i) if ((net->skips_mask >> (i - 1)) & 1) a.skip_in_mask |= 1 << i;
    a.use_viewdirs = net->use_viewdirs; a.output_ch = net->output_ch;
    a.acts = acts;
    a.tl = make_train_layout(net->W / 32, net->D, net->use_viewdirs, 0, half_acts);
    const int64_t ntiles = (n_points + 127) / 128;
    const int blocks = (int)(ntiles < 256 ? ntiles : 256);
    hipStream_t st = as_stream(stream);
    switch (net->W / 32 + (net->use_viewdirs ? 0 : 100) + (acts ? (half_acts ? 2000 : 1000) : 0)) {
        case 8: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 108: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, false>), dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 104: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, false>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1008: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, true, 1>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1108: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, false, 1>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1004: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, true, 1>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1104: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, false, 1>), dim3(blocks), dim3(256), 0, st, a); break;
        case 2008: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, true, 2>), dim3(blocks), dim3(256), 0, st, a); break;
        case 2108: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<8, false, 2>), dim3(blocks), dim3(256), 0, st, a); break;
        case 2004: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, true, 2>), dim3(blocks), dim3(256), 0, st, a); break;
        case 2104: hipLaunchKernelGGL((query_fwd_bf16x3_kernel<4, false, 2>), dim3(blocks), dim3(256), 0, st, a); break;
    }
    return check_launch(who);
}

extern "C" int nrf_query_fwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                    const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream) {
    return query_fwd_x3_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, nullptr, stream, "nrf_query_fwd_bf16x3");
}

extern "C" int nrf_query_fwd_train_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                          const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw,
                                          float *acts, void *stream) {
    NRF_REQUIRE(acts || n_points == 0, "nrf_query_fwd_train_bf16x3: acts workspace is null");
    return query_fwd_x3_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, acts, stream,
                             "nrf_query_fwd_train_bf16x3");
}

/* The same forward saving its activations as f16 (TrainLayout::half_acts; training mode "bf16x3_a16"): the workspace has the
 * size nrf_train_acts_floats gives, half of it is used; nrf_query_bwd_bf16x3_a16 consumes it. */
extern "C" int nrf_query_fwd_train_bf16x3_a16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                              const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw,
                                              float *acts, void *stream) {
    NRF_REQUIRE(acts || n_points == 0, "nrf_query_fwd_train_bf16x3_a16: acts workspace is null");
    return query_fwd_x3_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, acts, stream,
                             "nrf_query_fwd_train_bf16x3_a16", 1);
}

extern "C" int64_t nrf_mlp_bf16x3_bwd_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net || !emb) return -1;
    if (const char *e = bf_check_net(net, emb, false)) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16x3_bwd_packed_bytes: %s", e); return -1; }
    BfPlan P;
    P.split = 1;
    bf_plan_net_bwd(P, net);
    return bf_total_bytes(P);
}

extern "C" int nrf_mlp_bf16x3_bwd_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    NRF_REQUIRE(net && emb && packed, "nrf_mlp_bf16x3_bwd_pack_weights: null pointer");
    if (const char *e = bf_check_net(net, emb, false)) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_bf16x3_bwd_pack_weights: %s", e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "nrf_mlp_bf16x3_bwd_pack_weights: packed must be 16-byte aligned");
    BfPlan P;
    P.split = 1;
    bf_plan_net_bwd(P, net);
    return bf_pack(P, packed, as_stream(stream));
}

// Called by query_bwd_impl (mlp_bwd.hip) in place of the f32 dgrad launch.  NRF_ERR_UNSUPPORTED (with nothing launched)
// when the shape has no bf16x3 kernel: the caller then runs the f32 chain.
namespace nrf {
int launch_dgrad_x3(const nrf_net_t *net, const nrf_embed_t *emb, const NetPlan &plan, const float *packed,
                    const void *packed_bwd_x3, const float *g_raw, const float *acts, float *gbuf, int64_t n_points,
                    hipStream_t st, int half_acts, const unsigned *gmax) {
    if (bf_check_net(net, emb, false) || net->D < 2) return NRF_ERR_UNSUPPORTED;
    BfPlan P;
    P.split = 1;
    bf_plan_net_bwd(P, net);
    DgradX3Args a{};
    a.plan = plan;
    a.tl = make_train_layout(net->W / 32, net->D, net->use_viewdirs, 0, half_acts, gmax != nullptr);    // reads the gate bits only
    a.gmax = gmax;
    a.packed = packed; a.stream = (const char *)packed_bwd_x3; a.stream_bytes = P.bytes;
    a.g_raw = g_raw; a.acts = acts; a.gbuf = gbuf; a.n_points = n_points;
    const int64_t ntiles = (n_points + 127) / 128;
    const int blocks = (int)(ntiles < 256 ? ntiles : 256);
    switch (net->W / 32 + (net->use_viewdirs ? 0 : 100) + (gmax ? 1000 : 0)) {
        case 1008: hipLaunchKernelGGL((dgrad_x3_kernel<8, true, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1108: hipLaunchKernelGGL((dgrad_x3_kernel<8, false, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1004: hipLaunchKernelGGL((dgrad_x3_kernel<4, true, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 1104: hipLaunchKernelGGL((dgrad_x3_kernel<4, false, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((dgrad_x3_kernel<8, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 108: hipLaunchKernelGGL((dgrad_x3_kernel<8, false>), dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((dgrad_x3_kernel<4, true>), dim3(blocks), dim3(256), 0, st, a); break;
        case 104: hipLaunchKernelGGL((dgrad_x3_kernel<4, false>), dim3(blocks), dim3(256), 0, st, a); break;
        default: return NRF_ERR_UNSUPPORTED;
    }
    return check_launch("nrf_query_bwd_bf16x3 (dgrad)");
}
}  // namespace nrf

#ifdef NRF_BF16_PROFILE
extern "C" int nrf_bf16_profile_set(unsigned long long *device_counters) { nrf::g_prof_host_ptr = device_counters; return 0; }
#endif

/* DNeRF_Straightforward on the bf16x3 arithmetic (net->reserved = input_ch_time; raw points, embed mode 0). */
static const char *bf_check_straight(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net->use_viewdirs) return "the fused straightforward kernels need use_viewdirs";
    if (const char *e = bf_check_net(net, emb, false)) return e;
    if (emb->multires_time < 0 || emb->multires_time > 10 || net->reserved != 1 + 2 * emb->multires_time)
        return "time channels (net->reserved) must equal 1 + 2*multires_time with multires_time <= 10";
    if (__builtin_popcount((unsigned)net->skips_mask & ((1u << (net->D - 1)) - 1u)) > 1)
        return "the fused straightforward kernel takes one skip layer";
    if ((net->skips_mask >> (net->D - 1)) & 1) return "a skip at the last trunk layer is not supported (nor by the reference)";
    return nullptr;
}

static int64_t straight_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb, int split, const char *who) {
    if (!net || !emb) return -1;
    if (const char *e = bf_check_straight(net, emb)) { nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e); return -1; }
    BfPlan P;
    P.split = split;
    bf_plan_net(P, net, emb, false, true);
    return bf_total_bytes(P);
}

static int straight_pack(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream, int split, const char *who) {
    NRF_REQUIRE(net && emb && packed, "%s: null pointer", who);
    if (const char *e = bf_check_straight(net, emb)) return nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e);
    BfPlan P;
    P.split = split;
    bf_plan_net(P, net, emb, false, true);
    return bf_pack(P, packed, as_stream(stream));
}

static int straight_query(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                          const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                          int32_t times_per_point, float *raw, void *stream, int split, const char *who) {
    NRF_REQUIRE(net && emb, "%s: null descriptor", who);
    if (n_points != 0) NRF_REQUIRE(packed && pts && viewdirs && times && raw, "%s: null pointer", who);
    NRF_REQUIRE(n_points >= 0, "%s: negative n_points", who);
    if (const char *e = bf_check_straight(net, emb)) return nrf::fail(NRF_ERR_UNSUPPORTED, "%s: %s", who, e);
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0,
                "%s: packed and raw must be 16-byte aligned", who);
    NRF_REQUIRE(pts_per_ray >= 1 && n_points % pts_per_ray == 0, "%s: n_points (%lld) must be a multiple of pts_per_ray (%d)",
                who, (long long)n_points, pts_per_ray);
    if (n_points == 0) return NRF_OK;
    BfPlan P;
    P.split = split;
    bf_plan_net(P, net, emb, false, true);
    StraightX3Args sa{};
    Bf16Args &a = sa.b;
    a.stream = (const char *)packed; a.stream_bytes = P.bytes;
    a.pts = pts; a.viewdirs = viewdirs; a.raw = raw; a.n_points = n_points;
    a.pts_per_ray = pts_per_ray;
    a.n_trunk = net->D;
    a.skip_in_mask = 0;
    for (int i = 1; i < net->D; ++i) if ((net->skips_mask >> (i - 1)) & 1) a.skip_in_mask |= 1 << i;
    a.use_viewdirs = 1; a.output_ch = 4;
    sa.times = times; sa.t_div = times_per_point ? 1 : pts_per_ray;
    const int64_t ntiles = split ? (n_points + 127) / 128 : (n_points + 255) / 256;
    const int blocks = (int)(ntiles < 256 ? ntiles : 256);
    hipStream_t st = as_stream(stream);
    switch (net->W / 32 + (split ? 100 : 0)) {
        case 108: hipLaunchKernelGGL((straight_query_fwd_bf16x3_kernel<8>), dim3(blocks), dim3(256), 0, st, sa); break;
        case 104: hipLaunchKernelGGL((straight_query_fwd_bf16x3_kernel<4>), dim3(blocks), dim3(256), 0, st, sa); break;
        case 8: hipLaunchKernelGGL((straight_query_fwd_bf16_kernel<8>), dim3(blocks), dim3(256), 0, st, sa); break;
        case 4: hipLaunchKernelGGL((straight_query_fwd_bf16_kernel<4>), dim3(blocks), dim3(256), 0, st, sa); break;
    }
    return check_launch(who);
}

extern "C" int64_t nrf_dnerf_straight_bf16x3_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb) {
    return straight_packed_bytes(net, emb, 1, "nrf_dnerf_straight_bf16x3_packed_bytes");
}
extern "C" int nrf_dnerf_straight_bf16x3_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    return straight_pack(net, emb, packed, stream, 1, "nrf_dnerf_straight_bf16x3_pack_weights");
}
extern "C" int nrf_dnerf_straight_query_fwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed,
                                                   const float *pts, const float *viewdirs, const float *times,
                                                   int64_t n_points, int32_t pts_per_ray, int32_t times_per_point, float *raw,
                                                   void *stream) {
    return straight_query(net, emb, packed, pts, viewdirs, times, n_points, pts_per_ray, times_per_point, raw, stream, 1,
                          "nrf_dnerf_straight_query_fwd_bf16x3");
}
extern "C" int64_t nrf_dnerf_straight_bf16_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb) {
    return straight_packed_bytes(net, emb, 0, "nrf_dnerf_straight_bf16_packed_bytes");
}
extern "C" int nrf_dnerf_straight_bf16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream) {
    return straight_pack(net, emb, packed, stream, 0, "nrf_dnerf_straight_bf16_pack_weights");
}
extern "C" int nrf_dnerf_straight_query_fwd_bf16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed,
                                                 const float *pts, const float *viewdirs, const float *times, int64_t n_points,
                                                 int32_t pts_per_ray, int32_t times_per_point, float *raw, void *stream) {
    return straight_query(net, emb, packed, pts, viewdirs, times, n_points, pts_per_ray, times_per_point, raw, stream, 0,
                          "nrf_dnerf_straight_query_fwd_bf16");
}
