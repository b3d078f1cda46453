// Device + planner code shared by the fused forward (mlp_fused.hip) and backward (mlp_bwd.hip) kernels.
#pragma once
#include "nrf_common.h"
#include "mlp_layout.h"

namespace nrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// weight packing
struct PackSec {
    const float *src;   // nn.Linear weight [n_out, ld] (or bias [n_out])
    int64_t dst;        // offset in floats
    int32_t kind;       // 0 gemm block, 1 bias, 2 head vectors, 3 plain copy, 4 zero fill
    int32_t ld, n_out, nt_out;
    int32_t blk_type, blk_param, blk_ks, ch_off;
    int32_t count;      // floats in this section
    int32_t transposed; // gemm block of W^T (dgrad): output rows = W's input channels ch_off + 32t + row,
                        // contraction index = W's output channel (valid below n_src)
    int32_t n_src, pad_;
};
constexpr int kMaxSecs = 64;        // sections per pack launch (the table travels as a kernel argument)
constexpr int kPlanSecs = 256;      // sections of one plan: D = 16 with LayerNorm needs ~115
struct PackArgs {
    PackSec sec[kMaxSecs];
    int32_t nsec;
};
struct PackList {
    PackSec sec[kPlanSecs];
    int32_t nsec;
};

static __global__ void __launch_bounds__(256) pack_kernel(PackArgs a, float *__restrict__ packed) {
    const PackSec &s = a.sec[blockIdx.y];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < s.count; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (s.kind == 0) {  // [G][t][lane][i]
            int i = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
            int t = rest % s.nt_out, G = rest / s.nt_out;
            int h = lane >> 5, n = 32 * t + (lane & 31);
            int c = blk_chan(s.blk_type, s.blk_param, s.blk_ks, h, 4 * G + i);
            if (!s.transposed) {   // n_src > 0: live input channels of a hidden block whose last tile is zero padding
                if (c >= 0 && n < s.n_out && (s.n_src <= 0 || c < s.n_src)) v = s.src[(int64_t)n * s.ld + s.ch_off + c];
            } else {
                if (c >= 0 && c < s.n_src && n < s.n_out) v = s.src[(int64_t)c * s.ld + s.ch_off + n];
            }
        } else if (s.kind == 1) {  // bias [t][g][h][i]
            int i = e & 3, h = (e >> 2) & 1, g = (e >> 3) & 3, t = e >> 5;
            int n = 32 * t + 8 * g + 4 * h + i;
            if (n < s.n_out) v = s.src[n];
        } else if (s.kind == 2) {  // head vectors [o][t][g][h][i] over nt_out (= nt_in of the head) tiles
            int per = s.nt_out * 32;
            int o = e / per, q = e - o * per;
            int i = q & 3, h = (q >> 2) & 1, g = (q >> 3) & 3, t = q >> 5;
            int c = 32 * t + 8 * g + 4 * h + i;          // n_src: live input channels (the last tile may be padding)
            if (c < s.n_src) v = s.src[(int64_t)o * s.ld + s.ch_off + c];
        } else if (s.kind == 3) {
            v = s.src[e];
        }
        packed[s.dst + e] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host planner
struct Planner {
    NetPlan plan;
    PackList pack;
    int64_t cur = 0;
    bool want_pack;
    const char *err = nullptr;

    int64_t alloc(int64_t n) {
        int64_t o = cur;
        cur += (n + 3) & ~int64_t(3);  // keep every section 16-byte aligned
        return o;
    }
    void add(PackSec s) {
        if (pack.nsec >= kPlanSecs) { err = "too many weight sections"; return; }
        pack.sec[pack.nsec++] = s;
    }
    // k_valid > 0: a hidden input block of which only the first k_valid channels exist (W not a multiple of 32)
    int64_t gemm_block(const float *w, int ld, int n_out, int nt_out, int type, int param, int ks, int ch_off, int k_valid = 0) {
        int count = (ks / 4) * nt_out * 256;
        int64_t o = alloc(count);
        add(PackSec{w, o, 0, ld, n_out, nt_out, type, param, ks, ch_off, count, 0, k_valid, 0});
        return o;
    }
    // W^T block for dgrad: [4*nt_src groups][nt_out tiles]: out rows = W's input channels k_off.. (k_valid of
    // them), contraction over W's n_src output channels in accumulator (BLK_HIDDEN) order.
    int64_t gemm_block_T(const float *w, int ld, int k_off, int k_valid, int nt_out, int n_src, int nt_src) {
        int count = (4 * nt_src) * nt_out * 256;
        int64_t o = alloc(count);
        add(PackSec{w, o, 0, ld, k_valid, nt_out, BLK_HIDDEN, 0, 16 * nt_src, k_off, count, 1, n_src, 0});
        return o;
    }
    int64_t bias(const float *b, int n_out, int nt_out) {
        int count = nt_out * 32;
        int64_t o = alloc(count);
        add(PackSec{b, o, 1, 0, n_out, nt_out, 0, 0, 0, 0, count, 0, 0, 0});
        return o;
    }
    void head(const float *w, const float *b, int ld, int n_out, int nt_in, int ch_off, int64_t &vec,
              int64_t &bias_off, int n_in = -1) {
        int count = n_out * nt_in * 32;
        vec = alloc(count);
        add(PackSec{w, vec, 2, ld, n_out, nt_in, 0, 0, 0, ch_off, count, 0, n_in < 0 ? nt_in * 32 : n_in, 0});
        bias_off = alloc(n_out);
        add(PackSec{b, bias_off, 3, 0, n_out, 0, 0, 0, 0, 0, n_out, 0, 0, 0});
    }
    void zeros(int64_t count) {  // a section the kernels never read with live data; keep it defined
        int64_t o = alloc(count);
        add(PackSec{nullptr, o, 4, 0, 0, 0, 0, 0, 0, 0, (int32_t)count, 0, 0, 0});
    }
};

// All sections of a plan, kMaxSecs per launch.
static inline int launch_pack(const Planner &P, float *packed, hipStream_t st, const char *who) {
    for (int i = 0; i < P.pack.nsec; ++i)
        if (!P.pack.sec[i].src && P.pack.sec[i].kind != 4)
            return nrf::fail(NRF_ERR_INVALID, "%s: a parameter pointer of the net is null (section %d)", who, i);
    for (int i0 = 0; i0 < P.pack.nsec; i0 += kMaxSecs) {
        PackArgs a{};
        a.nsec = P.pack.nsec - i0 < kMaxSecs ? P.pack.nsec - i0 : kMaxSecs;
        for (int i = 0; i < a.nsec; ++i) a.sec[i] = P.pack.sec[i0 + i];
        hipLaunchKernelGGL(pack_kernel, dim3(64, a.nsec), dim3(256), 0, st, a, packed);
    }
    return check_launch(who);
}

// Builds the plan for one NeRF-shaped net.  time_ch > 0: deformation ("_time") net whose first layer
// also takes gamma(t) (vanilla_dynamic/network.py:254-305) and whose head is out_w/out_b (W -> 3).
// straight: DNeRF_Straightforward (vanilla_dynamic/network.py:100-196) -- a NeRF whose trunk input AND skip are
// cat[gamma(x), gamma(t)], with the usual view head.
static const char *plan_net(Planner &P, const nrf_net_t *net, const nrf_embed_t *emb, int time_ch, bool straight = false) {
    NetPlan &pl = P.plan;
    const int W = net->W, D = net->D, ic = net->input_ch, icv = net->input_ch_views;
    // any --netwidth up to 256 (config.py:20-27 takes anything): the kernels run whole 32-channel tiles, a width that is
    // not a multiple of 32 is zero-padded in the packed weights (zero rows and bias -> the extra channels stay 0 through
    // every ReLU; zero columns in the consumers) -- same sums, same results
    if (W < 2 || W > 256) return "W must be in [2, 256]";
    if (net->use_layernorm && W % 32) return "LayerNorm nets need W to be a multiple of 32 (the statistics run over whole tiles)";
    if (D < 1 || D > NRF_MAX_D) return "D must be in [1, 16]";
    // a skip index >= D never fires in the reference's loop (network.py:49-53) and is ignored here too; index D - 1
    // would widen the trunk's output, which the reference's heads cannot take either
    if ((net->skips_mask >> (D - 1)) & 1) return "a skip at the last trunk layer is not supported (nor by the reference)";
    const bool views = net->use_viewdirs != 0 && (time_ch == 0 || straight);
    if (straight && time_ch <= 0) return "a straightforward net needs time channels";
    if (straight && __builtin_popcount((unsigned)net->skips_mask & ((1u << (D - 1)) - 1u)) > 1)
        return "the fused straightforward kernel takes one skip layer";
    pl.straight = straight;
    // the view layer has W/2 outputs (network.py:40): nv whole tiles, the last one zero-padded when W/2 is not a multiple
    // of 32 (W = 96, 160, 224; --netwidth accepts anything, config.py:20-27)
    const int nv = ((W + 31) / 32 + 1) / 2;   // = the kernels' NV = (NT + 1) / 2 (>= the tiles W/2 needs)
    if (views && net->use_layernorm && (W / 2) % 32)
        return "LayerNorm nets with view directions need W/2 to be a multiple of 32 (the statistics run over whole tiles)";
    pl.nt = (W + 31) / 32;
    pl.n_trunk = D;
    pl.use_viewdirs = views;
    pl.output_ch = (time_ch && !straight) ? 3 : (views ? 4 : net->output_ch);
    if (pl.output_ch < 1 || pl.output_ch > kMaxHeadOut) return "output_ch must be in [1, 8]";
    if (emb->mode == 0) {
        if (ic != 3 + 6 * emb->multires || emb->multires < 0 || emb->multires > 10)
            return "fused encoding needs input_ch == 3 + 6*multires, multires <= 10";
        pl.x_type = BLK_PE3; pl.x_param = emb->multires;
        if (views) {
            if (icv != 3 + 6 * emb->multires_views || emb->multires_views < 0 || emb->multires_views > 4)
                return "fused encoding needs input_ch_views == 3 + 6*multires_views, multires_views <= 4";
            pl.d_type = BLK_PE3; pl.d_param = emb->multires_views;
        }
        if (time_ch) {
            if (time_ch != 1 + 2 * emb->multires_time || emb->multires_time < 0 || emb->multires_time > 10)
                return "fused encoding needs input_ch_time == 1 + 2*multires_time, multires_time <= 10";
            pl.t_type = BLK_PE1; pl.t_param = emb->multires_time;
        }
    } else {
        if (ic < 1 || ic > 2 * kKsX) return "embedded input: input_ch must be in [1, 64]";
        pl.x_type = BLK_RAW; pl.x_param = ic;
        if (views) {
            if (icv < 1 || icv > 2 * kKsD) return "embedded input: input_ch_views must be in [1, 32]";
            pl.d_type = BLK_RAW; pl.d_param = icv;
        }
        if (time_ch) {
            if (time_ch > 2 * kKsT) return "embedded input: input_ch_time must be <= 24";
            pl.t_type = BLK_RAW; pl.t_param = time_ch;
        }
    }
    const int nt = pl.nt;
    {   // layer 0 (network.py:49-51 with h = input_pts)
        int ld = ic + time_ch;
        pl.l0_w = P.gemm_block(net->pts_w[0], ld, W, nt, pl.x_type, pl.x_param, kKsX, 0);
        if (time_ch) pl.l0_w2 = P.gemm_block(net->pts_w[0], ld, W, nt, pl.t_type, pl.t_param, kKsT, ic);
        pl.l0_b = P.bias(net->pts_b[0], W, nt);
    }
    pl.trunk_base = P.cur;
    pl.trunk_stride = nt * 32 + 4 * nt * nt * 256 + (kKsX / 4) * nt * 256 + (straight ? (kKsT / 4) * nt * 256 : 0);
    pl.skip_in_mask = 0;
    const int skip_ch = ic + (straight ? time_ch : 0);   // channels a skip layer reads ahead of the hidden ones
    for (int i = 1; i < D; ++i) {
        bool skip = (net->skips_mask >> (i - 1)) & 1;  // network.py:52-53: layer i reads cat[input_pts, h]
        int ld = W + (skip ? skip_ch : 0);
        int64_t slot = P.cur;
        P.bias(net->pts_b[i], W, nt);
        P.gemm_block(net->pts_w[i], ld, W, nt, BLK_HIDDEN, 0, 16 * nt, skip ? skip_ch : 0, W);
        if (skip) {
            pl.skip_in_mask |= 1 << i;
            P.gemm_block(net->pts_w[i], ld, W, nt, pl.x_type, pl.x_param, kKsX, 0);
            if (straight) P.gemm_block(net->pts_w[i], ld, W, nt, pl.t_type, pl.t_param, kKsT, ic);
        } else {
            P.zeros((kKsX / 4) * nt * 256);
            if (straight) P.zeros((kKsT / 4) * nt * 256);
        }
        if (P.cur - slot != pl.trunk_stride) return "internal: trunk slot size mismatch";
    }
    if (views) {
        P.head(net->alpha_w, net->alpha_b, W, 1, nt, 0, pl.alpha_vec, pl.alpha_b, W);                  // network.py:56
        pl.feat_w = P.gemm_block(net->feature_w, W, W, nt, BLK_HIDDEN, 0, 16 * nt, 0, W);               // network.py:57
        pl.feat_b = P.bias(net->feature_b, W, nt);
        pl.views_w_hid = P.gemm_block(net->views_w, W + icv, W / 2, nv, BLK_HIDDEN, 0, 16 * nt, 0, W);  // :58-62
        pl.views_w_in = P.gemm_block(net->views_w, W + icv, W / 2, nv, pl.d_type, pl.d_param, kKsD, W);
        pl.views_b = P.bias(net->views_b, W / 2, nv);
        P.head(net->rgb_w, net->rgb_b, W / 2, 3, nv, 0, pl.rgb_vec, pl.rgb_b, W / 2);                // network.py:64
    } else {
        P.head(net->out_w, net->out_b, W, pl.output_ch, nt, 0, pl.out_vec, pl.out_b, W);               // network.py:67
    }
    pl.layernorm = net->use_layernorm != 0;
    if (pl.layernorm) {   // vanilla_dynamic/network.py:46-48 (or :275-277 for the time net): affine parameters in bias order
        pl.ln_base = P.cur;
        pl.ln_stride = 2 * nt * 32;
        for (int i = 0; i < D; ++i) {
            P.bias(net->ln_w[i], W, nt);
            P.bias(net->ln_b[i], W, nt);
        }
        if (views) {
            pl.views_ln = P.cur;
            P.bias(net->views_ln_w, W / 2, nv);
            P.bias(net->views_ln_b, W / 2, nv);
        }
    }
    // W^T blocks for the backward chain (dgrad): hidden parts only (inputs need no gradient here)
    pl.trunkT_base = P.cur;
    pl.trunkT_stride = 4 * nt * nt * 256;
    for (int i = 1; i < D; ++i) {
        bool skip = (net->skips_mask >> (i - 1)) & 1;
        P.gemm_block_T(net->pts_w[i], W + (skip ? skip_ch : 0), skip ? skip_ch : 0, W, nt, W, nt);
    }
    if (views) {
        pl.featT_w = P.gemm_block_T(net->feature_w, W, 0, W, nt, W, nt);
        pl.viewsT_w = P.gemm_block_T(net->views_w, W + icv, 0, W, nt, W / 2, nv);
    }
    // W_x^T blocks (gradient w.r.t. the encoded position, needed when the points themselves depend on parameters:
    // the D-NeRF chain x + dx).  Canonical nets with the fused encoding only.
    pl.xT_base = -1; pl.xT_stride = 0;
    if (emb->mode == 0 && time_ch == 0) {
        pl.xT_base = P.cur;
        pl.xT_stride = 4 * nt * 2 * 256;
        for (int i = 0; i < D; ++i) {
            bool skip = i > 0 && ((net->skips_mask >> (i - 1)) & 1);
            if (i == 0 || skip) P.gemm_block_T(net->pts_w[i], i == 0 ? ic : W + ic, 0, ic, 2, W, nt);
        }
    }
    pl.total_floats = P.cur;
    return P.err;
}

// ------------------------------------------------------------------------------------------------
// device side
//
// All weight traffic goes through ONE buffer descriptor (wave-uniform, in SGPRs): address =
// descriptor base + SGPR byte offset + per-lane VGPR offset.  With plain pointers the compiler
// materialises a 64-bit VGPR address per load of the unrolled layer and spills them.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane16;  // lane * 16: per-lane byte offset inside a 1 KiB wave load
    int h16;     // (lane >> 5) * 16: per-half offset into bias / head-vector tables
};

__device__ __forceinline__ f32x4 wload(const WBuf &wb, int voff, uint32_t soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wb.rsrc, voff, (int)soff_bytes, 0));
}
__device__ __forceinline__ float wload1(const WBuf &wb, uint32_t soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wb.rsrc, 0, (int)soff_bytes, 0));
}

// `b` etc. below are offsets in FLOATS from the start of the packed buffer (wave-uniform).
template <int NT_OUT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT_OUT], const WBuf &wb, uint32_t b) {
    __builtin_amdgcn_sched_barrier(0);  // the loads stay between the neighbouring stages (hoisted, they spill)
#pragma unroll
    for (int t = 0; t < NT_OUT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = wload(wb, wb.h16, (b + ((t * 4 + g) * 2) * 4) * 4);
            acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1]; acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
        }
}

// One k-group (4 k-steps) of weights for NT_OUT output tiles: NT_OUT coalesced 1 KiB wave loads.
template <int NT_OUT>
__device__ __forceinline__ void load_wgroup(f32x4 (&wv)[NT_OUT], const WBuf &wb, uint32_t w, int G) {
#pragma unroll
    for (int t = 0; t < NT_OUT; ++t) wv[t] = wload(wb, wb.lane16, (w + (G * NT_OUT + t) * 256) * 4);
}

// The k loop is fully unrolled (operand registers must be statically indexed).  Left alone, the
// scheduler hoists every weight load of the layer to the top and spills; the sched_barrier pins a
// one-group-ahead software pipeline instead: group G+1's loads are in flight during group G's
// 4*NT_OUT MFMAs (>= 1024 cycles, several L2 latencies).
#define NRF_GEMM_BODY(NG, SRC)                                                                          \
    asm volatile("" : "+s"(w)); /* opaque: keeps the per-load (w + const) SGPR adds inside the tile loop */ \
    f32x4 wa[NT_OUT], wb_[NT_OUT];                                                                      \
    load_wgroup<NT_OUT>(wa, wb, w, 0);                                                                  \
    _Pragma("unroll") for (int G = 0; G < (NG); ++G) {                                                  \
        f32x4(&cur)[NT_OUT] = (G & 1) ? wb_ : wa;                                                       \
        f32x4(&nxt)[NT_OUT] = (G & 1) ? wa : wb_;                                                       \
        if (G + 1 < (NG)) load_wgroup<NT_OUT>(nxt, wb, w, G + 1);                                       \
        NRF_GEMM_SIDE(G) /* after the loads: the in-order vmcnt wait for them must not wait for side stores */ \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        _Pragma("unroll") for (int t = 0; t < NT_OUT; ++t) _Pragma("unroll") for (int i = 0; i < 4; ++i) \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][i], SRC, acc[t], 0, 0, 0);             \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }

// acc[t] += W_block * xin   (input block held as one register per k-step)
#define NRF_GEMM_SIDE(G)
template <int KS, int NT_OUT>
__device__ __forceinline__ void gemm_in(f32x16 (&acc)[NT_OUT], const float (&xin)[KS], const WBuf &wb, uint32_t w) {
    NRF_GEMM_BODY(KS / 4, xin[4 * G + i])
}

// acc[t] += W_block * hin   (input = previous layer's accumulators, used as they stand)
template <int NT_IN, int NT_OUT>
__device__ __forceinline__ void gemm_hid(f32x16 (&acc)[NT_OUT], const f32x16 (&hin)[NT_IN], const WBuf &wb,
                                         uint32_t w) {
    NRF_GEMM_BODY(4 * NT_IN, hin[G >> 2][4 * (G & 3) + i])
}
#undef NRF_GEMM_SIDE

// The same, with a side job per k-group: side(G) is issued under group G's MFMAs (G is a constant after
// unrolling).  The training kernels save the operand they are consuming this way -- group G's four registers,
// four stores -- instead of a burst of 16 * NT_IN stores between layers that backs up the store path while the
// matrix pipe idles.
#define NRF_GEMM_SIDE(G) side(G);
template <int NT_IN, int NT_OUT, class Side>
__device__ __forceinline__ void gemm_hid_side(f32x16 (&acc)[NT_OUT], const f32x16 (&hin)[NT_IN], const WBuf &wb,
                                              uint32_t w, Side &&side) {
    NRF_GEMM_BODY(4 * NT_IN, hin[G >> 2][4 * (G & 3) + i])
}
#undef NRF_GEMM_SIDE

template <int NT>
__device__ __forceinline__ void relu_move(f32x16 (&dst)[NT], const f32x16 (&src)[NT], bool relu) {
    // One instruction per value beside the accumulator read: fmaxf() on an MFMA result is lowered to a canonicalising
    // v_max_f32 x, x, x followed by the max itself, and VALU work is not hidden under f32 MFMAs on this chip
    // (profiles/ubench/mfma_f32_issue.hip).  max(x, 0) on the bit pattern as a signed integer is the same function of
    // every non-NaN float (negative floats are negative integers, -0.0 included) and is a single v_max_i32.
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = __float_as_int(src[t][r]);
            if (relu) {
                dst[t][r] = __int_as_float(b > 0 ? b : 0);
            } else {   // a move, kept a distinct value: a bare copy lets the allocator alias the two register sets
                float v = src[t][r];   // and the render kernel then spills 32 registers
                asm("" : "+v"(v));
                dst[t][r] = v;
            }
        }
}

// LayerNorm over the 32 * NT channels of each point + ReLU, on the way out of the accumulators (vanilla_dynamic/network.py:
// 75-77): a lane holds 16 * NT of its point's channels, lane ^ 32 the others, so the two row statistics (mean, then the
// centred variance: biased, eps = 1e-5, as nn.LayerNorm) are lane-local sums finished by one cross-half shuffle each.
// `ln` = offset of (weight, bias) in bias order.  dst may alias src.
// SAVE (training forward): the normalised values (x - mean) * rstd go to the T8 section `xhat_sec` of `tb` and the point's
// rstd to float `rstd_off + point` -- the operands of the LayerNorm backward (ln_bwd, mlp_bwd.hip).
struct TBuf;
template <int NT, bool SAVE = false>
__device__ __forceinline__ void ln_relu_move(f32x16 (&dst)[NT], const f32x16 (&src)[NT], const WBuf &wb, uint32_t ln,
                                             const TBuf *tb = nullptr, uint32_t xhat_sec = 0, uint32_t rstd_off = 0);

// relu_move with the NEXT layer's bias loads folded in: as soon as tile t has left the accumulator file its
// registers are refilled with the next layer's bias, so those loads (an L2 round trip that the first MFMA of a layer
// otherwise waits for with the matrix pipe idle) are in flight under the remaining tiles' moves.  (Requesting the next
// layer's first weight group before the move as well was tried: 32 more live registers, and the kernel spills.)
template <int NT>
__device__ __forceinline__ void relu_move_bias(f32x16 (&dst)[NT], f32x16 (&acc)[NT], const WBuf &wb, uint32_t b) {
    asm volatile("" : "+s"(b));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bits = __float_as_int(acc[t][r]);
            dst[t][r] = __int_as_float(bits > 0 ? bits : 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = wload(wb, wb.h16, (b + ((t * 4 + g) * 2) * 4) * 4);
            acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1]; acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// lane-local part of  sum_c vec[c] * h[c]  over this lane's channels, then the other half's part.
// Loads are issued a whole tile (4 x 16 B) ahead of their use: left to itself the compiler waits for
// each load before the next (an L2 round trip per 16 bytes).
template <int NT_IN>
__device__ __forceinline__ float head_dot(const f32x16 (&hin)[NT_IN], const WBuf &wb, uint32_t vec) {
    float s = 0.0f;
    asm volatile("" : "+s"(vec));
    f32x4 va[4], vb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) va[g] = wload(wb, wb.h16, (vec + (g * 2) * 4) * 4);
#pragma unroll
    for (int t = 0; t < NT_IN; ++t) {
        f32x4(&cur)[4] = (t & 1) ? vb : va;
        f32x4(&nxt)[4] = (t & 1) ? va : vb;
        if (t + 1 < NT_IN) {
#pragma unroll
            for (int g = 0; g < 4; ++g) nxt[g] = wload(wb, wb.h16, (vec + (((t + 1) * 4 + g) * 2) * 4) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) s = fmaf(cur[g][i], hin[t][4 * g + i], s);
        __builtin_amdgcn_sched_barrier(0);
    }
    return s + __shfl_xor(s, 32);
}

// ---- T8 training-workspace access (TrainLayout in mlp_layout.h).  `sec` = section offset in floats from
// the tile base the descriptor points at; lane (h, q) owns point q of the tile.
struct TBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int m, h, q7;  // q >> 3, lane >> 5, q & 7
};

__device__ __forceinline__ TBuf make_tbuf(float *tile_base, int64_t tile_floats, int lane) {
    TBuf t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc(tile_base, 0, (int)(tile_floats * 4), 0x00020000);
    t.h = lane >> 5;
    t.m = (lane & 31) >> 3;
    t.q7 = lane & 7;
    return t;
}

template <int NT, bool SAVE>
__device__ __forceinline__ void ln_relu_move(f32x16 (&dst)[NT], const f32x16 (&src)[NT], const WBuf &wb, uint32_t ln,
                                             const TBuf *tb, uint32_t xhat_sec, uint32_t rstd_off) {
    constexpr float inv_n = 1.0f / (32 * NT);
    constexpr int C = 32 * NT;
    asm volatile("" : "+s"(ln));
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += src[t][r];
    const float mean = (s + __shfl_xor(s, 32)) * inv_n;
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = src[t][r] - mean;
            v = fmaf(d, d, v);
        }
    const float rstd = 1.0f / sqrtf((v + __shfl_xor(v, 32)) * inv_n + 1e-5f);
    int voff = 0;
    if constexpr (SAVE) {
        voff = ((tb->m * C + 4 * tb->h) * 8 + tb->q7) * 4;
        asm volatile("" : "+s"(xhat_sec), "+s"(rstd_off));
        if (tb->h == 0)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstd), tb->rsrc, (tb->m * 8 + tb->q7) * 4, (int)(rstd_off * 4), 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 w[4], b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            w[g] = wload(wb, wb.h16, (ln + ((t * 4 + g) * 2) * 4) * 4);
            b[g] = wload(wb, wb.h16, (ln + NT * 32 + ((t * 4 + g) * 2) * 4) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float xh = (src[t][r] - mean) * rstd;
            if constexpr (SAVE)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xh), tb->rsrc, voff,
                                                      (int)((xhat_sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
            const float y = fmaf(xh, w[r >> 2][r & 3], b[r >> 2][r & 3]);
            dst[t][r] = fmaxf(y, 0.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulator-layout registers (register r of tile t = channel 32t + (r&3) + 8(r>>2) + 4h) <-> T8 section
template <int NT>
__device__ __forceinline__ void store_acc_t8(const TBuf &tb, uint32_t sec, const f32x16 (&v)[NT]) {
    constexpr int C = 32 * NT;
    const int voff = ((tb.m * C + 4 * tb.h) * 8 + tb.q7) * 4;
    __builtin_amdgcn_sched_barrier(0);  // keep the stores (and their operand lifetimes) out of the GEMM loops
    asm volatile("" : "+s"(sec));       // opaque: (sec + const) must not be hoisted out of the tile loop
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[t][r]), tb.rsrc, voff,
                                                  (int)((sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
    __builtin_amdgcn_sched_barrier(0);
}

// EVERY 16-byte buffer store goes through here.  The store reads its data registers over more than one cycle, and a VALU
// write to them in the slots right after it lands in the stored value: seen on gfx950 as the last two dwords of the
// transposed T8 stores of the 3-term dgrad kernel replaced, run to run differently, wherever the compiler had placed a
// v_pk_mov to those registers directly behind the store.  The compiler pads that gap only for stores whose soffset is not
// a register; these have one.  The asm keeps the data registers live across two idle slots (and, through its memory
// clobber, behind the store), so nothing can be written to them before the store has read them.
__device__ __forceinline__ void buffer_store_b128(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 0);
    asm volatile("s_nop 2" : "+v"(v) : : "memory");
}

// group G's share of store_acc_t8: registers 4(G&3) .. +3 of tile G>>2 (the ones group G's MFMAs read)
template <int NT>
__device__ __forceinline__ void store_acc_t8_group(const TBuf &tb, uint32_t sec, const f32x16 (&v)[NT], int G) {
    constexpr int C = 32 * NT;
    const int t = G >> 2;
    const int voff = ((tb.m * C + 4 * tb.h) * 8 + tb.q7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * (G & 3) + i;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[t][r]), tb.rsrc, voff,
                                              (int)((sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
    }
}
// group G's share of the gate bits (same bit order as store_gate_bits); the last group stores the four dwords
template <int NT>
__device__ __forceinline__ void gate_bits_group(const TBuf &tb, uint32_t slot, const f32x16 (&h)[NT], int lane,
                                                uint32_t (&m)[4], int G) {
    const int t = G >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = min(__float_as_uint(h[t][4 * (G & 3) + i]), 1u);
        m[t >> 1] = (m[t >> 1] << 1) | b;
    }
    if (G == 4 * NT - 1) {
        if constexpr (NT & 1) m[NT >> 1] <<= 16;
        u32x4 v = {m[0], m[1], m[2], m[3]};
        buffer_store_b128(v, tb.rsrc, lane * 16, (int)(slot * 4));
    }
}

// ReLU gates of NT tiles as bits (the backward chain's only use of the trunk activations): tile t, register r ->
// dword t>>1, bit 31 - (16(t&1) + r).  h = max(z, 0) >= 0, so min(bits, 1) is 1 exactly where h > 0.
template <int NT>
__device__ __forceinline__ void store_gate_bits(const TBuf &tb, uint32_t slot, const f32x16 (&h)[NT], int lane) {
    uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t b = min(__float_as_uint(h[t][r]), 1u);
            m[t >> 1] = (m[t >> 1] << 1) | b;
        }
    if constexpr (NT & 1) m[NT >> 1] <<= 16;
    u32x4 v = {m[0], m[1], m[2], m[3]};
    buffer_store_b128(v, tb.rsrc, lane * 16, (int)(slot * 4));
}
__device__ __forceinline__ u32x4 load_gate_bits(const TBuf &tb, uint32_t slot, int lane) {
    return __builtin_amdgcn_raw_buffer_load_b128(tb.rsrc, lane * 16, (int)(slot * 4), 0);
}
template <int NT>
__device__ __forceinline__ void apply_gate_bits(const u32x4 &m, f32x16 (&g)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            g[t][r] = (int32_t)(m[t >> 1] << (16 * (t & 1) + r)) < 0 ? g[t][r] : 0.0f;
}

// dst = gate ? src : 0 in ONE pass out of the accumulator file (relu_move + apply_gate_bits fused): the gate bit is
// sign-extended into a mask (v_bfe_i32) and ANDed onto the value's bits -- two VALU ops per value beside the
// accumulator read, instead of a move pass (read, max) followed by a gate pass (shift, compare, select).
template <int NT>
__device__ __forceinline__ void move_gated(f32x16 (&dst)[NT], const f32x16 (&src)[NT], const u32x4 &m) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mask = __builtin_amdgcn_sbfe((int)m[t >> 1], 31 - (16 * (t & 1) + r), 1);
            dst[t][r] = __uint_as_float(__float_as_uint(src[t][r]) & (uint32_t)mask);
        }
}

template <int NT>
__device__ __forceinline__ void load_acc_t8(const TBuf &tb, uint32_t sec, f32x16 (&v)[NT]) {
    constexpr int C = 32 * NT;
    const int voff = ((tb.m * C + 4 * tb.h) * 8 + tb.q7) * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                tb.rsrc, voff, (int)((sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0));
}

// input-block registers (one per k-step; channel index in k-space = 2*step + half)
template <int KS>
__device__ __forceinline__ void store_in_t8(const TBuf &tb, uint32_t sec, const float (&x)[KS]) {
    constexpr int C = 2 * KS;
    const int voff = ((tb.m * C + tb.h) * 8 + tb.q7) * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[s]), tb.rsrc, voff,
                                              (int)((sec + 2 * s * 8) * 4), 0);
}

// ---- pieces of the backward chain shared by the f32 (mlp_bwd.hip) and bf16x3 (mlp_bf16.hip) dgrad kernels
template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = 0.0f;
}

// head tile: 32 channels starting at channel c0 of a section with C channels; channel c0+o = vals[o] (o < n), else 0
__device__ __forceinline__ void store_head_tile(const TBuf &tb, uint32_t sec, int C, int c0, const float (&vals)[8],
                                                int n) {
    const int voff = ((tb.m * C + c0 + 16 * tb.h) * 8 + tb.q7) * 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v = 0.0f;
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (tb.h == 0 && j == o && o < n) v = vals[o];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), tb.rsrc, voff, (int)((sec + j * 8) * 4), 0);
    }
}

// g[t][r] += vec[channel] * s   (head weight vectors in accumulator order), loads one tile ahead
template <int NT>
__device__ __forceinline__ void axpy_head(f32x16 (&g)[NT], const WBuf &wb, uint32_t vec, float s) {
    asm volatile("" : "+s"(vec));
    f32x4 va[4], vb[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) va[gq] = wload(wb, wb.h16, (vec + (gq * 2) * 4) * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4(&cur)[4] = (t & 1) ? vb : va;
        f32x4(&nxt)[4] = (t & 1) ? va : vb;
        if (t + 1 < NT) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) nxt[gq] = wload(wb, wb.h16, (vec + (((t + 1) * 4 + gq) * 2) * 4) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int i = 0; i < 4; ++i) g[t][4 * gq + i] = fmaf(cur[gq][i], s, g[t][4 * gq + i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ float pow2i(int f) { return __int_as_float((127 + f) << 23); }

// Training mode "w16": the power of two that brings the step's largest |d(loss)/d(raw)| (its bits: gmax_kernel) into
// [2^10, 2^11) -- the scale of the f16 gradient workspace: gradients of inner layers are mostly SMALLER than d/d(raw) (sums of
// products with weights ~ 1/16), so the window is placed high: 2^5 of headroom to f16's 65504 (beyond it the store
// saturates) and full 11-bit precision down to 2^-24 of the largest value.  (Measured on the test nets: the window's place
// -- maximum at 1 or at 2^10 -- does not move the error, which is the 2^-12 rounding of the two operands.)  inv = true: the
// factor that takes a weight gradient back.
constexpr int kGradWindowLog2 = 10;
__host__ __device__ __forceinline__ int nrf_grad_exponent(unsigned maxbits) {
    int e = (int)((maxbits >> 23) & 0xffu) - 127 - kGradWindowLog2;
    if (maxbits == 0u) e = 0;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}
__device__ __forceinline__ float nrf_grad_scale(unsigned maxbits, bool inv = false) {
    const int e = nrf_grad_exponent(maxbits);
    return pow2i(inv ? e : -e);
}

// sin and cos of one encoder argument, |a| <= kSincosMax (every frequency of a scene-sized coordinate: x * 2^9 with
// |x| <= 32; callers test their largest argument and use the library's sincosf beyond it).  Two-FMA Cody-Waite
// reduction by pi/2 (n * c is exact inside the FMA), degree-7/8 minimax polynomials on [-pi/4, pi/4]: max abs error
// 9.3e-8 over x * 2^k, x in [-6, 6], k = 0..9 (libm's float sin: 7e-8) -- about half the instructions of ocml's
// sincosf, whose large-argument path rides along in every inlined copy.
constexpr float kSincosMax = 16384.0f;
__device__ __forceinline__ void nrf_sincos(float a, float *sn, float *cs) {
    const float n = rintf(a * 0.636619747f);
    float r = fmaf(-n, 1.57079637f, a);
    r = fmaf(-n, -4.37113883e-8f, r);
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    const float s = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    const float c = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = __int_as_float(__float_as_int(ss) ^ ((q & 2) << 30));
    *cs = __int_as_float(__float_as_int(cc) ^ (((q + 1) & 2) << 30));
}
// body(fn) with fn = the fast routine when every lane's largest argument is within range, else the library's
template <class Body>
__device__ __forceinline__ void sincos_dispatch(float amax, Body &&body) {
    if (__builtin_expect(__all(amax <= kSincosMax), 1)) body([](float a, float *s, float *c) { nrf_sincos(a, s, c); });
    else body([](float a, float *s, float *c) { sincosf(a, s, c); });
}

// gamma(v) for a 3-vector into the static k-step slots of BLK_PE3 (see blk_chan)
template <int KS>
__device__ __forceinline__ void fill_pe3(float (&xin)[KS], float x, float y, float z, int L, int h) {
    constexpr int NPAIR = (KS - 2) / 6;
    const float amax = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) * pow2i(L > 0 ? L - 1 : 0);
    sincos_dispatch(amax, [&](auto sc_fn) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
            const int f = 2 * j + h;
            const bool valid = f < L;
            const float sc = valid ? pow2i(f) : 0.0f;
            float sx, cx, sy, cy, sz, cz;
            sc_fn(x * sc, &sx, &cx);   // x * 2^f is exact, as in positional_encoding.py:29
            sc_fn(y * sc, &sy, &cy);
            sc_fn(z * sc, &sz, &cz);
            xin[6 * j + 0] = sx; xin[6 * j + 1] = sy; xin[6 * j + 2] = sz;
            xin[6 * j + 3] = valid ? cx : 0.0f; xin[6 * j + 4] = valid ? cy : 0.0f; xin[6 * j + 5] = valid ? cz : 0.0f;
        }
    });
#pragma unroll
    for (int s = 6 * NPAIR; s < KS - 2; ++s) xin[s] = 0.0f;
    xin[KS - 2] = h == 0 ? x : y;
    xin[KS - 1] = h == 0 ? z : 0.0f;
}

// gamma(t) for a scalar into the static k-step slots of BLK_PE1 (see blk_chan)
template <int KS>
__device__ __forceinline__ void fill_pe1(float (&tin)[KS], float t, int L, int h) {
    constexpr int NPAIR = (KS - 1) / 2;
    sincos_dispatch(fabsf(t) * pow2i(L > 0 ? L - 1 : 0), [&](auto sc_fn) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
            const int f = 2 * j + h;
            const bool valid = f < L;
            float s, c;
            sc_fn(t * (valid ? pow2i(f) : 0.0f), &s, &c);
            tin[2 * j] = s;
            tin[2 * j + 1] = valid ? c : 0.0f;
        }
    });
#pragma unroll
    for (int s = 2 * NPAIR; s < KS - 1; ++s) tin[s] = 0.0f;
    tin[KS - 1] = h == 0 ? t : 0.0f;
}

template <int KS>
__device__ __forceinline__ void fill_raw(float (&xin)[KS], const float *__restrict__ row, int n, int h) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        int c = 2 * s + h;
        xin[s] = c < n ? row[c] : 0.0f;
    }
}

}  // namespace nrf
