// Packed-weight layout shared by the pack kernel, the fused query kernels and their host-side planner.
//
// The fused kernels compute every layer TRANSPOSED:  out^T[channel, point] = W[channel, k] * act^T[k, point]
// with v_mfma_f32_32x32x2_f32, weights as the A operand and activations as the B operand.  A wave owns
// 32 points (the MFMA's 32 columns = lane & 31); the accumulator of one 32-channel output tile holds,
// in register r of lane l, channel  32*t + (r&3) + 8*(r>>2) + 4*(l>>5)  of point l&31 -- which is
// exactly the B-operand shape (k-slot = l>>5, column = l&31) of the NEXT layer.  So activations never
// leave registers between layers; only the k ORDER seen by the weights is permuted, and that
// permutation is baked into the packed weights here.
//
// A "k-step" is one MFMA (k = 2: lanes 0-31 supply one input channel, lanes 32-63 another); a
// "k-group" is 4 consecutive steps = one 16-byte weight load per lane.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NRF_HD __host__ __device__
#else
#define NRF_HD
#endif

namespace nrf {

enum BlkType : int32_t { BLK_NONE = 0, BLK_HIDDEN = 1, BLK_PE3 = 2, BLK_PE1 = 3, BLK_RAW = 4 };

// Input channel (within the block's own channel range) supplied by lane-half h at k-step s; -1 = zero pad.
//  BLK_HIDDEN: previous layer's accumulators, tile s>>4, register s&15.
//  BLK_PE3   : gamma(x) of a 3-vector with L = param frequencies, reference channel order
//              [x y z | sin(2^0 x y z) | cos(2^0 x y z) | sin(2^1 ..) ...] (positional_encoding.py:11-36).
//              Lane-half h evaluates the frequencies f = 2j+h, so each lane needs one sincos per (j, coord).
//              `ks` = k-steps reserved for the block (static positions keep the kernel's registers
//              statically indexed whatever L is).
//  BLK_PE1   : the same for a scalar (D-NeRF time embedding, vanilla_dynamic/positional_encoding.py:71-97).
//  BLK_RAW   : an already-embedded input row of `param` channels; channel c sits at step c>>1, half c&1.
NRF_HD inline int blk_chan(int type, int param, int ks, int h, int s) {
    switch (type) {
        case BLK_HIDDEN: {
            int t = s >> 4, r = s & 15;
            return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
        }
        case BLK_PE3: {  // steps 6j..6j+5: sin(xyz), cos(xyz) of frequency 2j+h; last two steps: x|y, z|pad
            int L = param;
            if (s == ks - 2) return h == 0 ? 0 : 1;
            if (s == ks - 1) return h == 0 ? 2 : -1;
            int npair = (ks - 2) / 6;
            if (s >= 6 * npair) return -1;
            int j = s / 6, r = s - 6 * j, f = 2 * j + h;
            return f < L ? 3 + 6 * f + r : -1;
        }
        case BLK_PE1: {  // steps 2j, 2j+1: sin, cos of frequency 2j+h; last step: t|pad
            int L = param;
            if (s == ks - 1) return h == 0 ? 0 : -1;
            int npair = (ks - 1) / 2;
            if (s >= 2 * npair) return -1;
            int j = s >> 1, r = s & 1, f = 2 * j + h;
            return f < L ? 1 + 2 * f + r : -1;
        }
        case BLK_RAW: {
            int c = 2 * s + h;
            return c < param ? c : -1;
        }
    }
    return -1;
}

constexpr int kKsX = 32;  // k-steps reserved for the position input (gamma(x), L <= 10 -> 63 channels)
constexpr int kKsD = 16;  // k-steps for the direction input (L <= 4 -> 27 channels)
constexpr int kKsT = 12;  // k-steps for the time input (L <= 10 -> 21 channels)
constexpr int kMaxHeadOut = 8;

// Offsets (in floats from the start of the packed buffer) of one net's sections.  Scalars only: the
// struct travels as a by-value kernel argument and must never be indexed dynamically (that would
// send it to scratch memory).
//   GEMM block    [k-group][out tile][64 lanes][4]      (one f32x4 per lane = 4 k-steps)
//   bias          [out tile][4][2][4]                   (accumulator order: channel 32t + 8g + 4h + i)
//   head vectors  [out][in tile][4][2][4] + bias [out]
// Trunk layers 1..D-1 sit in equal slots of `trunk_stride` floats:
//   slot + 0: bias | + nt*32: hidden block | + nt*32 + 4*nt*nt*256: skip (input) block, zeros unless
//   the layer reads cat[input_pts, h] (network.py:52-53).
struct NetPlan {
    int32_t nt;          // W / 32
    int32_t n_trunk;     // number of trunk layers (D)
    int32_t skip_in_mask;  // bit l set <=> trunk layer l also consumes the position input block
    int32_t use_viewdirs, output_ch;
    int32_t x_type, x_param, d_type, d_param, t_type, t_param;   // input block kinds
    int32_t straight;    // DNeRF_Straightforward: skip layers read cat[gamma(x), gamma(t), h]; the trunk slots then carry a
                         // time block after the position block (trunk_stride grows by (kKsT / 4) * nt * 256)
    int64_t l0_w, l0_w2, l0_b;          // layer 0: position block, time block (D-NeRF), bias
    int64_t trunk_base, trunk_stride;   // slot of layer l >= 1: trunk_base + (l-1)*trunk_stride
    int64_t feat_w, feat_b;             // feature_linear (network.py:57)
    int64_t views_w_hid, views_w_in, views_b;   // views_linears[0] (network.py:58-62)
    int64_t alpha_vec, alpha_b;         // alpha_linear (network.py:56)
    int64_t rgb_vec, rgb_b;             // rgb_linear (network.py:64)
    int64_t out_vec, out_b;             // output_linear (network.py:67) or _time_out (deformation net)
    int64_t trunkT_base, trunkT_stride; // W^T (hidden part) of trunk layer l >= 1: trunkT_base + (l-1)*stride
    int64_t featT_w, viewsT_w;          // W^T of feature_linear; of views_linears[0][:, :W]
    int64_t xT_base, xT_stride;         // W[:, :input_ch]^T of the layers that read the position block (layer 0, skip
                                        // layers, in layer order): input gradients for the D-NeRF chain; -1 if absent
    int64_t ln_base, ln_stride;         // LayerNorm (weight, bias) of trunk layer l in bias order: ln_base + l*ln_stride, +nt*32
    int64_t views_ln;                   // the view layer's (weight, bias): views_ln, + (nt/2)*32
    int32_t layernorm, pad2_;
    int64_t total_floats;
};

// Training workspace, per 32-point tile.  Every section is a [4][C][8] block ("T8"): element
// (channel c, point q of the tile) at ((q >> 3) * C + c) * 8 + (q & 7).  The forward kernel writes it from
// its accumulator registers (32-byte runs), the backward chain reads it back the same way, and the
// weight-gradient GEMM -- which contracts over POINTS -- reads whole 1 KiB pieces of it as MFMA operands
// (lane (s, c) takes the 4 points 8m+4s.. of channel c).
struct TrainLayout {
    int32_t nt, n_trunk, use_viewdirs;
    int32_t half_acts;               // 1: the activation sections (inputs, trunk, feature, view layer) hold f16 values, two
                                     // bytes each and POINT-major (P16: store_tile_p16 in mlp_bf16.hip), so a section of C
                                     // channels takes 16 * C floats (the bf16x3 training mode "a16": half the bytes the
                                     // forward writes and wgrad reads back)
    // activations (floats from the tile's base)
    int64_t a_xin, a_din, a_tin;     // input blocks in k-step space: channel 2*step + half (64 / 32 / 24 ch)
    int64_t a_trunk;                 // D sections of W channels: post-ReLU outputs of trunk layer l
    int64_t a_feat, a_hv;            // feature_linear output (W), views layer output (W/2 padded to whole tiles, post-ReLU)
    int64_t a_mask;                  // ReLU gates as bits: D + 1 slots (trunk layers, view layer) of [64 lanes][4 dwords]
    int64_t a_xhat, a_xhat_v;        // LayerNorm nets only: the normalised pre-activations (x - mean) * rstd of trunk layer l
                                     // (D sections of W) and of the view layer: what the LayerNorm backward differentiates
    int64_t a_rstd;                  // LayerNorm nets only: D + 1 slots of 32 floats: 1 / sqrt(var + eps) of each point
    int64_t a_tile;                  // floats per tile
    int32_t half_grads, pad2_;       // 1: the gradient sections g_trunk / g_feat / g_hv / g_head hold f16 values, P16 (pre-scaled by a
                                     // power of two, training mode "w16"): 16 * C floats per section of C channels
    // gradients w.r.t. pre-activations
    int64_t g_trunk;                 // D sections of W
    int64_t g_feat;                  // W + 32: the extra tile's row 0 is d/d(alpha)
    int64_t g_hv;                    // W/2 padded to whole tiles
    int64_t g_head;                  // 32: rows 0..2 d/d(rgb)  (or d/d(output_linear out) rows 0..output_ch-1)
    int64_t g_xin;                   // 64: d/d(gamma(x)) in the reference's channel order (only written on request)
    int64_t g_ln, g_ln_v;            // LayerNorm nets only: d/d(LayerNorm output) of trunk layer l (D sections of W) and of the
                                     // view layer: with a_xhat, the operands of the affine parameters' gradients
    int64_t g_tile;
};

NRF_HD inline TrainLayout make_train_layout(int nt, int D, int use_viewdirs, int layernorm = 0, int half_acts = 0,
                                            int half_grads = 0) {
    TrainLayout t{};
    t.nt = nt; t.n_trunk = D; t.use_viewdirs = use_viewdirs;
    const int W = 32 * nt;
    const int Wv = 32 * ((nt + 1) / 2);     // the view layer's W/2 outputs in whole 32-channel tiles (W = 96: 48 -> 64, zero-padded)
    t.half_acts = half_acts;
    const int pp = half_acts ? 16 : 32;     // floats per channel of a 32-point activation section
    int64_t c = 0;
    t.a_xin = c; c += pp * 2 * kKsX;
    t.a_din = c; c += pp * 2 * kKsD;
    t.a_tin = c; c += pp * 2 * kKsT;
    t.a_trunk = c; c += (int64_t)pp * W * D;
    t.a_feat = c; c += pp * W;
    t.a_hv = c; c += pp * Wv;
    t.a_mask = c; c += (int64_t)256 * (D + 1);
    t.a_xhat = t.a_xhat_v = t.a_rstd = -1;
    if (layernorm) {
        t.a_xhat = c; c += (int64_t)32 * W * D;
        t.a_xhat_v = c; c += 32 * Wv;
        t.a_rstd = c; c += (int64_t)32 * (D + 1);
    }
    t.a_tile = c;
    t.half_grads = half_grads;
    const int gp = half_grads ? 16 : 32;
    c = 0;
    t.g_trunk = c; c += (int64_t)gp * W * D;
    t.g_feat = c; c += gp * (W + 32);
    t.g_hv = c; c += gp * Wv;
    t.g_head = c; c += gp * 32;
    t.g_xin = c; c += 32 * 64;
    t.g_ln = t.g_ln_v = -1;
    if (layernorm) {
        t.g_ln = c; c += (int64_t)32 * W * D;
        t.g_ln_v = c; c += 32 * Wv;
    }
    t.g_tile = c;
    return t;
}

}  // namespace nrf
