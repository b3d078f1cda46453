// K3b: backward of the fused NeRF MLP on f32 MFMA, gfx950.  Autograd of network.py:46-69 w.r.t. the
// nn.Linear parameters (train.py:242), in three kinds of launches:
//
//  1. dgrad chain (one wave = 32 points, no LDS, no barriers -- the forward's structure run in reverse):
//     d/d(pre-activation) of every layer, obtained by chaining  g_in^T = W^T g_out^T  on the MFMA with
//     the accumulators re-used as the next B operand, ReLU masks read back from the saved activations.
//     Each layer's gradient tile is written to the T8 workspace.
//  2. wgrad GEMMs:  dW[n][k] = sum_points g[n][p] x[k][p]  contracts over POINTS; both operands are
//     read from the T8 workspace as whole 1 KiB wave loads (lane (s, c) = 4 points of channel c), a wave
//     owns a 128x128 block of dW (256 accumulator registers) for a chunk of points and writes a partial slab.
//     Bias gradients are the row sums of the same A operand.
//  3. reduce + unpack: sums the slabs over chunks (fixed order: deterministic) and scatters into the
//     nn.Linear [out, in] gradient layout, undoing the k-space permutation of the input blocks.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "mlp_device.h"

namespace nrf {

// store / load accumulator-layout registers into a T8 section of C channels starting at channel c0
template <int NT>
__device__ __forceinline__ void store_acc_t8c(const TBuf &tb, uint32_t sec, int C, const f32x16 (&v)[NT]) {
    const int voff = ((tb.m * C + 4 * tb.h) * 8 + tb.q7) * 4;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+s"(sec));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[t][r]), tb.rsrc, voff,
                                                  (int)((sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
    __builtin_amdgcn_sched_barrier(0);
}

// group G's share of store_acc_t8c (registers 4(G&3)..+3 of tile G>>2), issued under the GEMM that consumes v
template <int NT>
__device__ __forceinline__ void store_acc_t8c_group(const TBuf &tb, uint32_t sec, int voff, const f32x16 (&v)[NT], int G) {
    const int t = G >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * (G & 3) + i;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[t][r]), tb.rsrc, voff,
                                              (int)((sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
    }
}
__device__ __forceinline__ int t8c_voff(const TBuf &tb, int C) { return ((tb.m * C + 4 * tb.h) * 8 + tb.q7) * 4; }

// acc = act > 0 ? acc : 0, with act read back from the saved activations.  One 32-channel tile (16 loads)
// is in flight ahead of the tile being gated.
template <int NT>
__device__ __forceinline__ void relu_mask_t8(const TBuf &ta, uint32_t sec, f32x16 (&g)[NT]) {
    constexpr int C = 32 * NT;
    const int voff = ((ta.m * C + 4 * ta.h) * 8 + ta.q7) * 4;
    asm volatile("" : "+s"(sec));
    float aa[16], ab[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        aa[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            ta.rsrc, voff, (int)((sec + (8 * (r >> 2) + (r & 3)) * 8) * 4), 0));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float(&cur)[16] = (t & 1) ? ab : aa;
        float(&nxt)[16] = (t & 1) ? aa : ab;
        if (t + 1 < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                nxt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    ta.rsrc, voff, (int)((sec + (32 * (t + 1) + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = cur[r] > 0.0f ? g[t][r] : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LayerNorm backward in registers (autograd of vanilla_dynamic/network.py:75-76 / :88-89 / :298-299): on entry g holds
// d/d(LayerNorm output) of this lane's channels (already through the ReLU gate); on exit d/d(Linear output):
//     a = g * gamma;   g_z = rstd * (a - mean_c(a) - xhat * mean_c(a * xhat))
// with xhat = (z - mean) * rstd and rstd as the training forward saved them.  The incoming g is also written to `gy_sec`
// of the gradient workspace: with xhat it gives the affine parameters' gradients (ln_param_grad_kernel).
// Two passes over the tile's saved xhat (the second re-read hits in L2); a lane holds 16 * NT of its point's channels,
// lane ^ 32 the rest, so both channel means are lane-local sums finished by one cross-half shuffle.
template <int NT>
__device__ __forceinline__ void ln_bwd(f32x16 (&g)[NT], const TBuf &ta, const TBuf &tg, const WBuf &wb, uint32_t xhat_sec,
                                       uint32_t rstd_off, uint32_t gamma, uint32_t gy_sec) {
    constexpr int C = 32 * NT;
    constexpr float inv_n = 1.0f / C;
    const int voff = ((ta.m * C + 4 * ta.h) * 8 + ta.q7) * 4;
    asm volatile("" : "+s"(xhat_sec), "+s"(rstd_off), "+s"(gamma), "+s"(gy_sec));
    const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ta.rsrc, (ta.m * 8 + ta.q7) * 4,
                                                                                      (int)(rstd_off * 4), 0));
    float s1 = 0.0f, s2 = 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float xh[16];
        f32x4 w[4];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xh[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                ta.rsrc, voff, (int)((xhat_sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = wload(wb, wb.h16, (gamma + ((t * 4 + q) * 2) * 4) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g[t][r]), tg.rsrc, voff,
                                                  (int)((gy_sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = g[t][r] * w[r >> 2][r & 3];
            g[t][r] = a;
            s1 += a;
            s2 = fmaf(a, xh[r], s2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    s1 = (s1 + __shfl_xor(s1, 32)) * inv_n;
    s2 = (s2 + __shfl_xor(s2, 32)) * inv_n;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float xh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xh[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                ta.rsrc, voff, (int)((xhat_sec + (32 * t + 8 * (r >> 2) + (r & 3)) * 8) * 4), 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = rstd * ((g[t][r] - s1) - xh[r] * s2);
        __builtin_amdgcn_sched_barrier(0);
    }
}

int launch_dgrad_x3(const nrf_net_t *net, const nrf_embed_t *emb, const NetPlan &plan, const float *packed,
                    const void *packed_bwd_x3, const float *g_raw, const float *acts, float *gbuf, int64_t n_points,
                    hipStream_t st, int half_acts, const unsigned *gmax);      // mlp_bf16.hip

struct DgradArgs {
    NetPlan plan;
    TrainLayout tl;
    const float *packed;
    const float *g_raw;  // [P, output_ch]
    const float *acts;
    float *gbuf;
    int64_t n_points;
};

// LN: the use_layernorm nets -- every gated gradient passes through the LayerNorm backward (ln_bwd) on its way to the
// Linear's output, where the chain and the weight-gradient GEMMs take it as before
template <int NT, bool GX = false, bool LN = false>
__global__ void __launch_bounds__(256, 1) dgrad_kernel(const DgradArgs a) {
    const NetPlan &pl = a.plan;
    const TrainLayout &tl = a.tl;
    const int lane = threadIdx.x & 63, n = lane & 31;
    WBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, (int)(pl.total_floats * 4), 0x00020000);
    wb.lane16 = lane * 16;
    wb.h16 = (lane >> 5) * 16;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (a.n_points + 31) >> 5;
    constexpr int W = 32 * NT;

    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t p = tile * 32 + n;
        const bool live = p < a.n_points;
        TBuf ta = make_tbuf(const_cast<float *>(a.acts) + tile * tl.a_tile, tl.a_tile, lane);
        TBuf tg = make_tbuf(a.gbuf + tile * tl.g_tile, tl.g_tile, lane);
        f32x16 g[NT], nx[NT];
        f32x16 gx[2];   // GX: d/d(gamma(x)), 64 channels in the reference's order, summed over the layers that read it
        int xi = __builtin_popcount(((unsigned)pl.skip_in_mask | 1u) & ((1u << pl.n_trunk) - 1u));
        if constexpr (GX) zero_acc<2>(gx);
        zero_acc<NT>(g);
        if (pl.use_viewdirs) {
            {
                constexpr int NV = (NT + 1) / 2;     // tiles of the view layer (zero-padded when NT is odd)
                float4 gr = live ? reinterpret_cast<const float4 *>(a.g_raw)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
                float hv[8] = {gr.x, gr.y, gr.z, 0.f, 0.f, 0.f, 0.f, 0.f};
                store_head_tile(tg, (uint32_t)tl.g_head, 32, 0, hv, 3);          // d/d(rgb logits): rgb_linear wgrad
                // through rgb_linear (network.py:64) and the view layer's ReLU
                f32x16 g2[NV];
                zero_acc<NV>(g2);
                axpy_head<NV>(g2, wb, (uint32_t)pl.rgb_vec + 0 * (NV) * 32, gr.x);
                axpy_head<NV>(g2, wb, (uint32_t)pl.rgb_vec + 1 * (NV) * 32, gr.y);
                axpy_head<NV>(g2, wb, (uint32_t)pl.rgb_vec + 2 * (NV) * 32, gr.z);
                apply_gate_bits<NV>(load_gate_bits(ta, (uint32_t)tl.a_mask + pl.n_trunk * 256, lane), g2);
                if constexpr (LN)
                    ln_bwd<NV>(g2, ta, tg, wb, (uint32_t)tl.a_xhat_v, (uint32_t)tl.a_rstd + pl.n_trunk * 32,
                               (uint32_t)pl.views_ln, (uint32_t)tl.g_ln_v);
                // every gradient block is saved (for wgrad) by the GEMM that consumes it, four registers per k-group
                // through views_linears[0][:, :W] (network.py:58-61) -> d/d(feature)
                {
                    uint32_t sec = (uint32_t)tl.g_hv;
                    asm volatile("" : "+s"(sec));
                    const int voff = t8c_voff(tg, 32 * NV);
                    gemm_hid_side<NV, NT>(g, g2, wb, (uint32_t)pl.viewsT_w,
                                              [&](int G) { store_acc_t8c_group<NV>(tg, sec, voff, g2, G); });
                }
                float ha[8] = {gr.w, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store_head_tile(tg, (uint32_t)tl.g_feat, W + 32, W, ha, 1);       // d/d(alpha): alpha_linear wgrad
                // through feature_linear and alpha_linear (network.py:56-57) -> d/d(h_{D-1})
                zero_acc<NT>(nx);
                {
                    uint32_t sec = (uint32_t)tl.g_feat;
                    asm volatile("" : "+s"(sec));
                    const int voff = t8c_voff(tg, W + 32);
                    gemm_hid_side<NT, NT>(nx, g, wb, (uint32_t)pl.featT_w,
                                          [&](int G) { store_acc_t8c_group<NT>(tg, sec, voff, g, G); });
                }
                axpy_head<NT>(nx, wb, (uint32_t)pl.alpha_vec, gr.w);
                relu_move<NT>(g, nx, false);
            }
        } else {
            float ho[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) ho[o] = (live && o < pl.output_ch) ? a.g_raw[p * pl.output_ch + o] : 0.0f;
            store_head_tile(tg, (uint32_t)tl.g_head, 32, 0, ho, pl.output_ch);
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < pl.output_ch) axpy_head<NT>(g, wb, (uint32_t)pl.out_vec + o * NT * 32, ho[o]);
        }
        // trunk in reverse (network.py:49-53)
        u32x4 gate = load_gate_bits(ta, (uint32_t)tl.a_mask + (pl.n_trunk - 1) * 256, lane);
        apply_gate_bits<NT>(gate, g);
        auto ln_layer = [&](int l) {
            ln_bwd<NT>(g, ta, tg, wb, (uint32_t)tl.a_xhat + l * (32 * W), (uint32_t)tl.a_rstd + l * 32,
                       (uint32_t)(pl.ln_base + (int64_t)l * pl.ln_stride), (uint32_t)tl.g_ln + l * (32 * W));
        };
        if constexpr (LN) ln_layer(pl.n_trunk - 1);
        for (int l = pl.n_trunk - 1; l >= 0; --l) {
            // g = gated gradient w.r.t. layer l's pre-activations; the next layer's gates are in flight during this GEMM
            if (l > 0) gate = load_gate_bits(ta, (uint32_t)tl.a_mask + (l - 1) * 256, lane);
            if constexpr (GX) {
                if (l == 0 || ((pl.skip_in_mask >> l) & 1)) {   // g_x += W_l[:, :input_ch]^T g_l
                    --xi;
                    gemm_hid<NT, 2>(gx, g, wb, (uint32_t)(pl.xT_base + (int64_t)xi * pl.xT_stride));
                }
            }
            if (l > 0) {
                zero_acc<NT>(nx);
                uint32_t sec = (uint32_t)tl.g_trunk + l * (32 * W);
                asm volatile("" : "+s"(sec));
                const int voff = t8c_voff(tg, W);
                gemm_hid_side<NT, NT>(nx, g, wb, (uint32_t)(pl.trunkT_base + (int64_t)(l - 1) * pl.trunkT_stride),
                                      [&](int G) { store_acc_t8c_group<NT>(tg, sec, voff, g, G); });
                move_gated<NT>(g, nx, gate);     // through layer l-1's ReLU, on the way out of the accumulators
                if constexpr (LN) ln_layer(l - 1);
            } else {
                store_acc_t8c<NT>(tg, (uint32_t)tl.g_trunk, W, g);      // layer 0's block has no consumer
            }
        }
        if constexpr (GX) store_acc_t8c<2>(tg, (uint32_t)tl.g_xin, 64, gx);
    }
}

// d(loss)/d(x) from d(loss)/d(gamma(x)) (positional_encoding.py:24-33 differentiated): channel 3+6f+d is
// sin(2^f x_d), 3+6f+3+d is cos(2^f x_d).  One thread per point; reads the point's 64-channel column of the T8 section.
__global__ void __launch_bounds__(256) pe3_bwd_t8_kernel(const float *gbuf, int64_t g_tile, int64_t g_sec, const float *pts,
                                                         int L, int64_t n_points, float *g_pts) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const int q = (int)(p & 31);
    const float *g = gbuf + (p >> 5) * g_tile + g_sec + ((q >> 3) * 64) * 8 + (q & 7);   // channel c at g[c * 8]
    float out[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float x = pts[p * 3 + d];
        float acc = g[d * 8];
        for (int f = 0; f < L; ++f) {
            const float sc = pow2i(f);
            float sn, cs;
            sincosf(x * sc, &sn, &cs);
            acc += sc * (cs * g[(3 + 6 * f + d) * 8] - sn * g[(3 + 6 * f + 3 + d) * 8]);
        }
        out[d] = acc;
    }
    g_pts[p * 3 + 0] = out[0]; g_pts[p * 3 + 1] = out[1]; g_pts[p * 3 + 2] = out[2];
}

// ------------------------------------------------------------------------------------------------
// wgrad: slab[chunk][n][k] = sum_{points in chunk} G[n][p] * X[k][p]
//
// A workgroup owns a super-block of up to 8 x 8 channel tiles (256 x 256 of dW); each wave accumulates a rectangle
// of up to 4 x 4 tiles (256 accumulator registers).  Per step (8 points of one 32-point tile) the operands are the
// super-block's <= 8 G pieces and <= 8 X pieces of 1 KiB; the four waves copy them global -> LDS ONCE
// (buffer_load ... lds, a 3-stage ring, one barrier per step) and every wave reads its 4 + 4 fragments from LDS.
// Without the sharing each wave pulled its own 8 KiB per 64 MFMAs = 8 B/clk/CU, the per-CU load limit (measured 60 %
// of peak); shared it is 4 B/clk/CU.  X may be the concatenation of two T8 sections (cat[h, gamma(x)] of a skip
// layer, cat[feature, gamma(d)] of the view layer), so one launch covers a whole nn.Linear.
struct WgradEntry {       // one nn.Linear's weight-gradient GEMM
    float *slab;         // [nchunks][NP*KP] partial dW, then [nchunks][NP] partial db
    int64_t g_sec;
    int64_t a_sec[2];        // X = section 0 (k_tiles0 tiles) followed by section 1
    int32_t Cg, Ca[2];       // channels in the sections (f32 T8, or f16 P16)
    int32_t n_tiles, k_tiles, k_tiles0;  // 32-channel tiles of G (rows of dW) and of X (columns)
    int32_t x_pe[2];         // f16 (P16) X sections: 1 = an encoded-input section (columns in k-step space), 0 = hidden
};
// ALL GEMMs of a backward pass in one launch: a block is (pair, chunk), a pair = (GEMM, super-block) -- pairs sorted
// by tile count, largest first, so the skinny super-blocks (layer 0, the skip / view slices, the heads: HBM- and
// latency-bound) run on CUs the full 8 x 8 ones (MFMA-bound) have left, instead of each kind holding the whole chip in turn.
constexpr int kWgEntries = 20, kWgPairs = 64;     // D <= 16 trunk layers + 3 heads; <= 4 super-blocks each, typically 1-2
struct WgradArgs {
    const float *gbuf;   // tile stride g_tile
    const float *acts;   // tile stride a_tile
    int64_t g_tile, a_tile, ntiles;
    int32_t chunk_tiles, nchunks;
    WgradEntry e[kWgEntries];
    uint16_t pair[kWgPairs];     // entry << 8 | super-block
};

#ifndef NRF_WG_M
#define NRF_WG_M 2
#endif
constexpr int kWgM = NRF_WG_M;                // 8-point groups per step: a step covers 8 * kWgM points of one tile
constexpr int kWgStage = 16384 * kWgM;        // per group: <= 8 G pieces then <= 8 X pieces of 1 KiB

// The step loop of wgrad_kernel for a wave that owns NA x KA tiles (0 x 0: a wave with no tiles still copies its share
// of the pieces and takes part in the barriers).  Fragments are double-buffered in registers: step s+1's are read from
// LDS under step s's MFMAs, so the matrix pipe restarts right after the barrier instead of waiting out the LDS latency.
// Iteration `step`: stage(step+1) has landed for everyone (its fragments are read now, for the next iteration);
// stage(step-1) == stage(step+2) was last read during iteration step-2, so the request for step+2 may overwrite it.
// (Measured and dropped: copies three steps ahead with a partial vmcnt wait -- 6 % slower on full super-blocks, a
// gain only on the few-tile ones, 0.4 % of the step overall.)
template <int NA, int KA, class Req>
__device__ __forceinline__ void wgrad_steps(f32x16 (&acc)[4][4], float (&bs)[4], const char *lds, int lane, int nt0, int kt0,
                                            int64_t nsteps, Req &&request) {
    constexpr int NAr = NA ? NA : 1, KAr = KA ? KA : 1;
    f32x4 A0[kWgM][NAr], B0[kWgM][KAr], A1[kWgM][NAr], B1[kWgM][KAr];
    auto read_frags = [&](f32x4 (&A)[kWgM][NAr], f32x4 (&B)[kWgM][KAr], int stage) {
#pragma unroll
        for (int m = 0; m < kWgM; ++m) {
            const char *base = lds + stage * kWgStage + m * 16384 + lane * 16;
#pragma unroll
            for (int x = 0; x < NA; ++x) A[m][x] = *reinterpret_cast<const f32x4 *>(base + (nt0 + x) * 1024);
#pragma unroll
            for (int y = 0; y < KA; ++y) B[m][y] = *reinterpret_cast<const f32x4 *>(base + (8 + kt0 + y) * 1024);
        }
    };
    auto land = [&]() {   // own copies have landed; after the barrier everyone's have
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int st_nxt = 1, st_req = 2;
    auto body = [&](int64_t step, f32x4 (&A)[kWgM][NAr], f32x4 (&B)[kWgM][KAr], f32x4 (&An)[kWgM][NAr],
                    f32x4 (&Bn)[kWgM][KAr]) {
        land();
        if (step + 2 < nsteps) request(step + 2, st_req);   // first: its latency is the step's critical path
        if constexpr (NA > 0 && KA > 0) {
            read_frags(An, Bn, st_nxt);     // past the last step this re-reads a stale stage: unused
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < kWgM; ++m)
#pragma unroll
                for (int x = 0; x < NA; ++x) {
                    bs[x] += (A[m][x][0] + A[m][x][1]) + (A[m][x][2] + A[m][x][3]);
#pragma unroll
                    for (int y = 0; y < KA; ++y)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m][x][j], B[m][y][j], acc[x][y], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        st_nxt = st_nxt == 2 ? 0 : st_nxt + 1;
        st_req = st_req == 2 ? 0 : st_req + 1;
    };
    if (nsteps > 0) {
        request(0, 0);
        request(1, 1);          // nsteps is even
        land();
        if constexpr (NA > 0 && KA > 0) read_frags(A0, B0, 0);
    }
    for (int64_t step = 0; step < nsteps; step += 2) {
        body(step, A0, B0, A1, B1);
        body(step + 1, A1, B1, A0, B0);
    }
}

// ---- bf16x3 wgrad: the same super-blocks, pieces, LDS ring and slabs; the contraction over points runs on
// v_mfma_f32_32x32x16_bf16 with every f32 operand split into hi = bf16(v), lo = bf16(v - hi) as it leaves LDS:
//   dW += Gh Xh^T + Gh Xl^T + Gl Xh^T      (the dropped Gl Xl^T is 2^-18 relative)
// One MFMA covers the step's 16 points: lane (channel c, half s) supplies the 8 points of group s -- two 16-byte LDS reads
// (the piece of group s holds points 0-3 of every channel in its first 512 bytes, points 4-7 in the second).  At three
// MFMAs of 32 cycles per tile pair a full super-block asks for ~21 B/clk/CU of operands, twice what HBM delivers to 256
// CUs: this kernel is HBM-bound (64 FLOP/B), the matrix pipe waits, and the conversion VALU work fills the gaps.
#ifndef NRF_WG_X3_PIPE
#define NRF_WG_X3_PIPE 4      // VALU ops pinned behind each MFMA of a step (0: leave the order to the scheduler)
#endif
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
struct WgFrag { u32x4 hi, lo; };

__device__ __forceinline__ void wg_split(const f32x4 &p0, const f32x4 &p1, WgFrag &f) {
    const float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((wg_f32x2){v[2 * w], v[2 * w + 1]}, wg_bf16x2));
        const float ra = v[2 * w] - __uint_as_float(hi << 16), rb = v[2 * w + 1] - __uint_as_float(hi & 0xffff0000u);
        f.hi[w] = hi;
        f.lo[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector((wg_f32x2){ra, rb}, wg_bf16x2));
    }
}

// ---- f16 sections are POINT-major ("P16", store_tile_p16 in mlp_bf16.hip): per 32-channel tile two 1 KiB blocks j = 0, 1 of
// [point (32)][lane half h (2)][8 values].  A copy request brings 16 points of both blocks, eight lanes (128 contiguous
// bytes: four points of one block) at a time, so the LDS image of a tile is units of 256 bytes: [4 points of j = 0][the same 4
// points of j = 1] (eight lanes per 128-byte line; alternating single points between the blocks coalesces worse).  A
// lane's MFMA fragment -- 8 consecutive points of ONE slot -- is two ds_read_b64_tr_b16: lane (j = (lane >> 4) & 1,
// s = lane >> 5, i = lane & 15) passes the address of point row q = i >> 2 of its block, columns 4 (i & 3).., and receives
// slot 16 j + i of points 8 s .. 8 s + 3 (the instruction's mapping verified with an indexed image,
// scripts/microbench/tr_test.hip); the two 16-lane groups of a half read 128 contiguous bytes each, in disjoint banks.
// MFMA row / column m = 16 j + 8 h + e of a tile is therefore NOT channel m: the epilogue permutes.
// The reads are inline assembly: through the builtin the compiler cannot tell them from the in-flight LDS copies and puts
// `s_waitcnt vmcnt(0)` in front of every group of them -- which drains the copy pipeline the counted waits exist for (w16
// wgrad 2.69 -> 3.44 ms).  So the wait for the DATA is ours too: p16_wait() before the first use of a P16Frag.
typedef uint32_t wg_u32x2 __attribute__((ext_vector_type(2)));
struct P16Frag { wg_u32x2 lo, hi; };     // points 8s .. 8s+3, 8s+4 .. 8s+7 of the lane's slot
__device__ __forceinline__ P16Frag p16_read8(const char *row0) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)row0;
    P16Frag f;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(f.hi) : "v"(addr));
    return f;
}
// (the fragment passes THROUGH the wait: a register copy of it cannot be scheduled in front)
__device__ __forceinline__ void p16_wait(P16Frag &f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi)::"memory"); }
__device__ __forceinline__ u32x4 p16_value(const P16Frag &f) {
    const u32x4 v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return v;
}
// this lane's byte offset inside a tile image for the k-half it supplies (points 8 (lane >> 5) ..)
__device__ __forceinline__ int p16_lane_off(int lane) {
    return (lane >> 5) * 512 + ((lane >> 4) & 1) * 128 + ((lane & 15) >> 2) * 32 + (lane & 3) * 8;
}
// per-lane source offset of a copy request inside a tile's 2 KiB of the section (16 points from point 0)
__device__ __forceinline__ int p16_copy_off(int lane) { return ((lane >> 3) & 1) * 1024 + (lane >> 4) * 128 + (lane & 7) * 16; }
// MFMA index m of a tile -> channel within the tile: hidden sections (accumulator order), encoded-input sections (k-step space)
__device__ __forceinline__ int p16_chan_hidden(int m) {
    const int r = 8 * (m >> 4) + (m & 7);
    return (r & 3) + 8 * (r >> 2) + 4 * ((m >> 3) & 1);
}
__device__ __forceinline__ int p16_chan_input(int m) { return 2 * (8 * (m >> 4) + (m & 7)) + ((m >> 3) & 1); }

// the same from eight f16 values (TrainLayout::half_acts): an f16 is exactly hi + lo (11 significant bits)
typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
// (The elements go through scalar copies: `__builtin_bit_cast(wg_f16x2, p[w])` on a vector ELEMENT expression took element 0
// for every w -- one dword of the fragment used for all four, every weight gradient off by O(1) with the right norm; found
// with delta patterns in a crafted workspace.)
__device__ __forceinline__ void wg_split_h(const u32x4 p, WgFrag &f) {
    const uint32_t pw[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const wg_f32x2 v = __builtin_convertvector(__builtin_bit_cast(wg_f16x2, pw[w]), wg_f32x2);
        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wg_bf16x2));
        const float ra = v[0] - __uint_as_float(hi << 16), rb = v[1] - __uint_as_float(hi & 0xffff0000u);
        f.hi[w] = hi;
        f.lo[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector((wg_f32x2){ra, rb}, wg_bf16x2));
    }
}

// XH: the X operand's sections hold f16, point-major (P16 above): a tile's 16 points of a step are one 1 KiB copy and a
// lane's fragment two transposing 8-byte reads; G stays f32.
template <int NA, int KA, bool XH, class Req>
__device__ __forceinline__ void wgrad_steps_x3(f32x16 (&acc)[4][4], float (&bs)[4], const char *lds, int lane, int nt0, int kt0,
                                               int64_t nsteps, Req &&request) {
    static_assert(kWgM == 2, "the bf16x3 step covers one 32x32x16 MFMA: two 8-point groups");
    constexpr int NAr = NA ? NA : 1, KAr = KA ? KA : 1;
    WgFrag A0[NAr], B0[KAr], A1[NAr], B1[KAr];
    float R0[NAr], R1[NAr];                      // row sums of the G fragments (bias gradients)
    auto read_frags = [&](WgFrag (&A)[NAr], WgFrag (&B)[KAr], float (&R)[NAr], int stage) {
        const char *base = lds + stage * kWgStage + (lane >> 5) * 16384 + (lane & 31) * 16;
        P16Frag xf[KAr];
        if constexpr (XH) {         // the step's 16 points of X tile y: a 1 KiB P16 image in the first group's X area
#pragma unroll
            for (int y = 0; y < KA; ++y) xf[y] = p16_read8(lds + stage * kWgStage + 8192 + (kt0 + y) * 1024 + p16_lane_off(lane));
        }
#pragma unroll
        for (int x = 0; x < NA; ++x) {
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(base + (nt0 + x) * 1024);
            const f32x4 p1 = *reinterpret_cast<const f32x4 *>(base + (nt0 + x) * 1024 + 512);
            R[x] = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
            wg_split(p0, p1, A[x]);
        }
#pragma unroll
        for (int y = 0; y < KA; ++y) {
            if constexpr (XH) {
                p16_wait(xf[y]);
                wg_split_h(p16_value(xf[y]), B[y]);
            } else {
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(base + (8 + kt0 + y) * 1024);
                const f32x4 p1 = *reinterpret_cast<const f32x4 *>(base + (8 + kt0 + y) * 1024 + 512);
                wg_split(p0, p1, B[y]);
            }
        }
    };
    auto land = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int st_nxt = 1, st_req = 2;
    auto body = [&](int64_t step, WgFrag (&A)[NAr], WgFrag (&B)[KAr], float (&R)[NAr], WgFrag (&An)[NAr], WgFrag (&Bn)[KAr],
                    float (&Rn)[NAr]) {
        land();
        if (step + 2 < nsteps) request(step + 2, st_req);
        if constexpr (NA > 0 && KA > 0) {
            // the next step's fragments are read and split while this step's MFMAs run (the scheduler is free to interleave:
            // the matrix pipe has slack here); past the last step this re-reads a stale stage: unused
            read_frags(An, Bn, Rn, st_nxt);
#pragma unroll
            for (int x = 0; x < NA; ++x) {
                bs[x] += R[x];
#pragma unroll
                for (int y = 0; y < KA; ++y) {
                    const wg_bf16x8 ah = __builtin_bit_cast(wg_bf16x8, A[x].hi), al = __builtin_bit_cast(wg_bf16x8, A[x].lo);
                    const wg_bf16x8 bh = __builtin_bit_cast(wg_bf16x8, B[y].hi), bl = __builtin_bit_cast(wg_bf16x8, B[y].lo);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[x][y], 0, 0, 0);
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[x][y], 0, 0, 0);
                }
            }
#if NRF_WG_X3_PIPE
            // the split results are only used after the next barrier, and pure arithmetic gets sunk to its use: pin it here
#pragma unroll
            for (int x = 0; x < NA; ++x) asm volatile("" : "+v"(An[x].hi), "+v"(An[x].lo), "+v"(Rn[x]));
#pragma unroll
            for (int y = 0; y < KA; ++y) asm volatile("" : "+v"(Bn[y].hi), "+v"(Bn[y].lo));
            // A wave issues in order: an MFMA behind a busy matrix pipe holds back everything after it, so the split of the
            // next step's operands only overlaps the MFMAs if it sits BETWEEN them in program order.  Left alone the
            // scheduler built one of the two unrolled steps as 36 MFMAs in a row with the whole split after the barrier
            // (matrix pipe idle for ~250 VALU ops).  Pin the shape: the 16 LDS reads first, then per MFMA its share of the
            // ~190 VALU ops.
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA + (XH ? 1 : 2) * KA, 0);
#pragma unroll
            for (int i = 0; i < 3 * NA * KA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NRF_WG_X3_PIPE, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        st_nxt = st_nxt == 2 ? 0 : st_nxt + 1;
        st_req = st_req == 2 ? 0 : st_req + 1;
    };
    if (nsteps > 0) {
        request(0, 0);
        request(1, 1);          // nsteps is even
        land();
        if constexpr (NA > 0 && KA > 0) read_frags(A0, B0, R0, 0);
    }
    for (int64_t step = 0; step < nsteps; step += 2) {
        body(step, A0, B0, R0, A1, B1, R1);
        body(step + 1, A1, B1, R1, A0, B0, R0);
    }
}

// ---- training mode "w16": BOTH operands f16 (X as in XH; G pre-scaled by a power of two and saved as f16 by the dgrad
// chain), so the contraction over points is ONE v_mfma_f32_32x32x16_f16 per tile pair and 16 points on the fragments as they
// leave LDS -- no split, a third of the matrix work, half the bytes on both sides.  With so little arithmetic per byte the
// loop is bound by how many copies are in flight, so it is laid out for that (the first version kept the x3 loop's shape --
// 16 points a step, copies one step ahead, vmcnt(0) -- and ran at 3.6 TB/s with the matrix pipe 19 % busy):
//   * a step is a whole 32-point tile (four 8-point groups, two MFMAs per tile pair): half the barriers;
//   * FOUR LDS stages of 32 KiB (G tile images at x * 2048, X tile images at 16384 + y * 2048; a 1 KiB copy brings 16 points of
//     one tile), copies FOUR steps ahead (the running step's operands sit in registers, so its stage is free), and the wait
//     before the barrier is COUNTED: vector-memory operations retire in order, so vmcnt(2 * nd) -- nd = this wave's copies
//     per step -- returns when the step about to be read has landed and leaves the two younger steps in flight: 32 to 48 KiB
//     per CU, 8 to 12 MiB over the chip.
// Bias gradients = row sums of the G fragments (f32 adds).
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
#ifndef NRF_WG_H_STAGES
#define NRF_WG_H_STAGES 4      // 5 (all 160 KiB of LDS, 64 KiB of copies in flight per CU): measured no change (2.72 vs 2.71 ms)
#endif
constexpr int kHStage = 32768, kHStages = NRF_WG_H_STAGES, kHX = 16384;     // a stage: 8 G tile images of 2 KiB, then 8 X tile images
template <int NA, int KA, class Req>
__device__ __forceinline__ void wgrad_steps_h(f32x16 (&acc)[4][4], float (&bs)[4], const char *lds, int lane, int nt0, int kt0,
                                              int64_t nsteps, int nd, Req &&request) {
    constexpr int NAr = NA ? NA : 1, KAr = KA ? KA : 1;
    P16Frag A0[2][NAr], B0[2][KAr], A1[2][NAr], B1[2][KAr];
    auto read_frags = [&](P16Frag (&A)[2][NAr], P16Frag (&B)[2][KAr], int stage) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {        // k-block kb: points 16 kb .. 16 kb + 15 of the tile images (rows of 64 bytes)
            const char *base = lds + stage * kHStage + kb * 1024 + p16_lane_off(lane);
#pragma unroll
            for (int x = 0; x < NA; ++x) A[kb][x] = p16_read8(base + (nt0 + x) * 2048);
#pragma unroll
            for (int y = 0; y < KA; ++y) B[kb][y] = p16_read8(base + kHX + (kt0 + y) * 2048);
        }
    };
    // own copies of the step about to be read have landed (`younger` later steps may stay in flight); after the barrier
    // everyone's have
    auto land = [&](int younger) {
        switch (younger * nd) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;       // 4 * 8
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int st_nxt = 1, st_req = 0;
    auto body = [&](int64_t step, P16Frag (&A)[2][NAr], P16Frag (&B)[2][KAr], P16Frag (&An)[2][NAr], P16Frag (&Bn)[2][KAr]) {
        // requests issued so far: steps <= step + kHStages - 1; the one read now is step + 1
        const int64_t last = nsteps - 1 < step + kHStages - 1 ? nsteps - 1 : step + kHStages - 1;
        land(last > step + 1 ? (int)(last - (step + 1)) : 0);
        // stage (step + kHStages) % kHStages is this step's own: its fragments went to registers a step ago, and every wave is
        // past that read
        if (step + kHStages < nsteps) request(step + kHStages, st_req);
        if constexpr (NA > 0 && KA > 0) {
            // this step's fragments were read a step ago by inline assembly: the wait for them is ours
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int x = 0; x < NA; ++x) p16_wait(A[kb][x]);
#pragma unroll
                for (int y = 0; y < KA; ++y) p16_wait(B[kb][y]);
            }
            read_frags(An, Bn, st_nxt);     // past the last step this re-reads a stale stage: unused
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int x = 0; x < NA; ++x) {
                    float r = 0.0f;
                    const u32x4 av = p16_value(A[kb][x]);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t e = av[w];   // a scalar copy first: __builtin_bit_cast of a vector ELEMENT reads element 0
                        const wg_f32x2 v = __builtin_convertvector(__builtin_bit_cast(wg_f16x2, e), wg_f32x2);
                        r += v[0] + v[1];
                    }
                    bs[x] += r;
#pragma unroll
                    for (int y = 0; y < KA; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, av),
                                                                          __builtin_bit_cast(wg_f16x8, p16_value(B[kb][y])), acc[x][y], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        st_nxt = st_nxt + 1 == kHStages ? 0 : st_nxt + 1;
        st_req = st_req + 1 == kHStages ? 0 : st_req + 1;
    };
    if (nsteps > 0) {
#pragma unroll
        for (int i = 0; i < kHStages; ++i)
            if (i < nsteps) request(i, i);
        land(nsteps > kHStages - 1 ? kHStages - 1 : (int)(nsteps - 1));
        if constexpr (NA > 0 && KA > 0) read_frags(A0, B0, 0);
    }
    for (int64_t step = 0; step < nsteps; step += 2) {
        body(step, A0, B0, A1, B1);
        if (step + 1 < nsteps) body(step + 1, A1, B1, A0, B0);
    }
    // The last step's read-ahead is still in flight and nobody uses it: to the compiler its destination registers are dead
    // from here on, and an inline-assembly read that lands late would overwrite whatever they are reused for.  Keep them alive
    // through a wait (every fragment passes through it).
    if constexpr (NA > 0 && KA > 0) {
        if (nsteps > 0) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int x = 0; x < NA; ++x) { p16_wait(A0[kb][x]); p16_wait(A1[kb][x]); }
#pragma unroll
                for (int y = 0; y < KA; ++y) { p16_wait(B0[kb][y]); p16_wait(B1[kb][y]); }
            }
        }
    }
}

template <bool X3, bool XH = false, bool GH = false>
__global__ void __launch_bounds__(256, 1) wgrad_kernel(const WgradArgs b_arg) {
    __shared__ __attribute__((aligned(1024))) char lds[GH ? kHStages * kHStage : 3 * kWgStage];
    // The argument block is indexed by a block-uniform value: read it where it lies, in the kernel-argument segment
    // (constant address space -> scalar loads).  Indexing the by-value parameter made the compiler copy all 1.3 KB of it
    // to per-lane scratch memory at the top of every workgroup.
    typedef const WgradArgs __attribute__((address_space(4))) *ArgPtr;
    const WgradArgs __attribute__((address_space(4))) &b = *(ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)b_arg;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pair = blockIdx.x / b.nchunks, chunk = blockIdx.x - pair * b.nchunks;
    const int code = b.pair[pair];
    const WgradEntry __attribute__((address_space(4))) &ea = b.e[code >> 8];
    // this block's GEMM and the pass-wide fields as local scalars: read once here, not again inside the step loop
    WgradEntry a;
    a.slab = ea.slab; a.g_sec = ea.g_sec; a.a_sec[0] = ea.a_sec[0]; a.a_sec[1] = ea.a_sec[1];
    a.Cg = ea.Cg; a.Ca[0] = ea.Ca[0]; a.Ca[1] = ea.Ca[1];
    a.n_tiles = ea.n_tiles; a.k_tiles = ea.k_tiles; a.k_tiles0 = ea.k_tiles0;
    const float *const gbuf = b.gbuf, *const acts = b.acts;
    const int64_t g_tile = b.g_tile, a_tile = b.a_tile;
    const int sblk = code & 255;
    const int k_sblks = (a.k_tiles + 7) >> 3;
    const int nsb = sblk / k_sblks, ksb = sblk - nsb * k_sblks;
    const int gn = min(8, a.n_tiles - nsb * 8), xn = min(8, a.k_tiles - ksb * 8);   // pieces of the super-block
    // the super-block's gn x xn tiles are split over the four waves as a WY x WX grid of rectangles of <= 4 x 4
    // tiles; the grid (2x2, 1x4 or 4x1) with the fewest tiles per wave wins, so skinny GEMMs (layer 0, the skip and
    // view slices, the one-tile heads) keep all four matrix pipes busy
    int WY = 2, best = 1 << 30;
    for (int c = 0; c < 3; ++c) {
        const int wy_c = c == 0 ? 2 : (c == 1 ? 1 : 4), wx_c = 4 / wy_c;
        const int np = (gn + wy_c - 1) / wy_c, kp = (xn + wx_c - 1) / wx_c;
        if (np <= 4 && kp <= 4 && np * kp < best) { best = np * kp; WY = wy_c; }
    }
    const int WX = 4 / WY, nper = (gn + WY - 1) / WY, kper = (xn + WX - 1) / WX;
    const int nt0 = (wib / WX) * nper, kt0 = (wib % WX) * kper;   // this wave's first tile within the super-block
    const int n0 = nsb * 8 + nt0, k0 = ksb * 8 + kt0;
    const int na = max(0, min(nper, gn - nt0)), ka = max(0, min(kper, xn - kt0));
    const bool active = na > 0 && ka > 0;
    const int NP = a.n_tiles * 32, KP = a.k_tiles * 32;
    const int lane_src = ((lane & 31) * 8 + (lane >> 5) * 4) * 4;  // bytes in a piece: channel lane&31, points 4s..4s+3

    f32x16 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};

    const int64_t t0 = (int64_t)chunk * b.chunk_tiles;
    const int64_t t1 = min(b.ntiles, t0 + b.chunk_tiles);
    constexpr int kSpt = GH ? 1 : 4 / kWgM;        // steps per 32-point tile
    const int64_t nsteps = (t1 - t0) * kSpt;       // one step = kWgM groups of 8 points of one tile (GH: the whole tile)
    // GH: wave w copies G tiles 2w, 2w+1 and X tiles 2w, 2w+1 (two 16-point copies each): 4 copies a side
    const bool h_g = 2 * wib < gn, h_x = 2 * wib < xn;
    const int nd = GH ? 4 * ((int)h_g + (int)h_x) : 0;

    // copy this wave's share of step `step`'s pieces into `stage`: piece i (G pieces first, then X) by wave i & 3
    auto request = [&](int64_t step, int stage) {
        const int64_t tile = t0 + step / kSpt;
        const int m0 = (int)(step % kSpt) * kWgM;
        if constexpr (GH) {
            // both sides f16, point-major: a 1 KiB copy = 16 points of one 32-channel tile (p16_copy_off); a tile beyond its
            // section reads zeros.  Wave w: G tiles 2w, 2w+1 and X tiles 2w, 2w+1, two copies each
            const int csrc = p16_copy_off(lane);
            if (h_g) {
                __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(gbuf) + tile * g_tile + a.g_sec, 0, a.Cg * 32 * 2, 0x00020000);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = 2 * wib + (u >> 1), half = u & 1;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rg, (__attribute__((address_space(3))) void *)(lds + stage * kHStage + x * 2048 + half * 1024), 16, csrc,
                        (nsb * 8 + x) * 2048 + half * 512, 0, 0);
                }
            }
            if (h_x) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int y = 2 * wib + (u >> 1), half = u & 1;
                    const int kt = ksb * 8 + y;
                    const int sec = kt >= a.k_tiles0;
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(acts) + tile * a_tile + a.a_sec[sec], 0, a.Ca[sec] * 32 * 2, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rx, (__attribute__((address_space(3))) void *)(lds + stage * kHStage + kHX + y * 2048 + half * 1024), 16, csrc,
                        (sec ? kt - a.k_tiles0 : kt) * 2048 + half * 512, 0, 0);
                }
            }
            return;
        }
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(gbuf) + tile * g_tile + a.g_sec, 0, a.Cg * 32 * 4, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = wib + 4 * q;
            char *dst = lds + stage * kWgStage + i * 1024;
            if (i < gn) {
#pragma unroll
                for (int mm = 0; mm < kWgM; ++mm)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void *)(dst + mm * 16384),
                                                             16, lane_src,
                                                             (((m0 + mm) * a.Cg + 32 * (nsb * 8 + i)) * 8) * 4, 0, 0);
            } else if (XH) {
                // f16 X sections, point-major: one 1 KiB copy = the step's 16 points of one 32-channel tile (p16_copy_off),
                // into the first group's X area (8 tiles x 1 KiB)
                if (i >= 8 && i - 8 < xn) {
                    const int kt = ksb * 8 + (i - 8);
                    const int sec = kt >= a.k_tiles0;
                    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(acts) + tile * a_tile + a.a_sec[sec], 0, a.Ca[sec] * 32 * 2, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rx, (__attribute__((address_space(3))) void *)(lds + stage * kWgStage + 8192 + (i - 8) * 1024), 16,
                        p16_copy_off(lane), (sec ? kt - a.k_tiles0 : kt) * 2048 + m0 * 8 * 32, 0, 0);
                }
            } else if (i - 8 < xn && i >= 8) {
                const int kt = ksb * 8 + (i - 8);
                const int sec = kt >= a.k_tiles0;
                const int c0 = 32 * (sec ? kt - a.k_tiles0 : kt);
                __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(acts) + tile * a_tile + a.a_sec[sec], 0, a.Ca[sec] * 32 * 4, 0x00020000);
#pragma unroll
                for (int mm = 0; mm < kWgM; ++mm)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(dst + mm * 16384),
                                                             16, lane_src, (((m0 + mm) * a.Ca[sec] + c0) * 8) * 4, 0, 0);
            }
        }
    };
    // The step loop is compiled once per tile rectangle (NA x KA known at compile time -> straight-line bodies with
    // exact wait counts); waves of one workgroup may run different instances, all with the same barrier sequence.
    auto run = [&](auto na_c, auto ka_c) {
        constexpr int NA = decltype(na_c)::value, KA = decltype(ka_c)::value;
        if constexpr (GH) wgrad_steps_h<NA, KA>(acc, bs, lds, lane, nt0, kt0, nsteps, nd, request);
        else if constexpr (X3) wgrad_steps_x3<NA, KA, XH>(acc, bs, lds, lane, nt0, kt0, nsteps, request);
        else wgrad_steps<NA, KA>(acc, bs, lds, lane, nt0, kt0, nsteps, request);
    };
    auto pick_k = [&](auto na_c) {
        switch (ka) {
            case 1: run(na_c, std::integral_constant<int, 1>{}); break;
            case 2: run(na_c, std::integral_constant<int, 2>{}); break;
            case 3: run(na_c, std::integral_constant<int, 3>{}); break;
            default: run(na_c, std::integral_constant<int, 4>{}); break;
        }
    };
    if (!active) {
        if constexpr (GH) wgrad_steps_h<0, 0>(acc, bs, lds, lane, nt0, kt0, nsteps, nd, request);
        else wgrad_steps<0, 0>(acc, bs, lds, lane, nt0, kt0, nsteps, request);
    } else {
        switch (na) {
            case 1: pick_k(std::integral_constant<int, 1>{}); break;
            case 2: pick_k(std::integral_constant<int, 2>{}); break;
            case 3: pick_k(std::integral_constant<int, 3>{}); break;
            default: pick_k(std::integral_constant<int, 4>{}); break;
        }
    }
    if (!active) return;
    float *slab = a.slab + (int64_t)chunk * NP * KP;
#pragma unroll
    for (int x = 0; x < 4; ++x)
        if (x < na) {
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (y < ka) {
                    // f16 (P16) operands: MFMA row / column m is not channel m (p16_chan_*): a permutation inside the tile
                    int col = lane & 31;
                    if constexpr (XH) col = ea.x_pe[(k0 + y) >= a.k_tiles0] ? p16_chan_input(col) : p16_chan_hidden(col);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int mrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if constexpr (GH) mrow = p16_chan_hidden(mrow);
                        slab[(int64_t)(32 * (n0 + x) + mrow) * KP + 32 * (k0 + y) + col] = acc[x][y][r];
                    }
                }
            if (k0 == 0) {
                float v = bs[x] + __shfl_xor(bs[x], 32);
                const int brow = GH ? p16_chan_hidden(lane & 31) : lane;
                if (lane < 32) a.slab[(int64_t)b.nchunks * NP * KP + (int64_t)chunk * NP + 32 * (n0 + x) + brow] = v;
            }
        }
}

// dst[n][ch_off + chan(k)] = sum_chunks slab[chunk][row0 + n][k]; dst_b[n] = sum_chunks slab_b[chunk][row0 + n]
struct ReduceArgs {
    const float *slab;
    float *dst_w;   // [n_valid, ld]
    float *dst_b;   // [n_valid] or null
    int32_t nchunks, NP, KP, row0, n_valid, ld, ch_off;
    int32_t x_type, x_param, x_ks;  // how slab column kc + k maps to an input channel (BLK_HIDDEN: identity)
    int32_t k_valid;                // columns holding live channels (BLK_HIDDEN)
    int32_t kc, kn;                 // the slab columns [kc, kc + kn) this launch unpacks
};

// Bandwidth-bound (nchunks x dW bytes): 4 lanes share one output quad, each summing a quarter of the chunks with
// 8 float4 loads in flight, combined by shuffles in a FIXED order (bit-reproducible gradients).
constexpr int kReduceBatch = 24;
struct ReduceBatch {
    ReduceArgs e[kReduceBatch];
    const unsigned *gmax;      // training mode "w16": the slabs hold gradients scaled by nrf_grad_scale(*gmax); else null
};

// All reductions of a backward pass in one launch: blockIdx.y picks the entry (every GEMM keeps its own slab region,
// so nothing is reused between them); blocks beyond an entry's work leave at once.
__global__ void __launch_bounds__(256) reduce_unpack_kernel(const ReduceBatch batch) {
    const ReduceArgs &a = batch.e[blockIdx.y];
    const float unscale = batch.gmax ? nrf_grad_scale(*batch.gmax, true) : 1.0f;   // a power of two: exact
    const int kq = a.kn >> 2;                       // quads per row (kn is a multiple of 32)
    const int total_q = a.n_valid * kq;
    const int sub = threadIdx.x & 3;
    const int64_t stride = (int64_t)a.NP * a.KP;
    const int c_lo = (a.nchunks * sub) >> 2, c_hi = (a.nchunks * (sub + 1)) >> 2;
    for (int q = blockIdx.x * 64 + (threadIdx.x >> 2); q < total_q + ((a.n_valid + 3) & ~3); q += gridDim.x * 64) {
        if (q < total_q) {
            const int n = q / kq, k = (q - n * kq) * 4;
            const float *s = a.slab + (int64_t)(a.row0 + n) * a.KP + a.kc + k;
            f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
            int c = c_lo;
            for (; c + 8 <= c_hi; c += 8) {
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4 *>(s + (c + u) * stride);
#pragma unroll
                for (int u = 0; u < 8; u += 2) { v0 += t[u]; v1 += t[u + 1]; }
            }
            for (; c < c_hi; ++c) v0 += *reinterpret_cast<const f32x4 *>(s + c * stride);
            v0 += v1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = v0[i];
                x += __shfl_xor(x, 1);
                x += __shfl_xor(x, 2);
                v0[i] = x;
            }
            if (sub == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = k + i;
                    const int chan = (a.x_type == BLK_HIDDEN) ? (kk < a.k_valid ? kk : -1)
                                                              : blk_chan(a.x_type, a.x_param, a.x_ks, kk & 1, kk >> 1);
                    if (chan >= 0) a.dst_w[(int64_t)n * a.ld + a.ch_off + chan] = v0[i] * unscale;
                }
            }
        } else if (a.dst_b) {   // bias rows: one lane group per row
            const int n = q - total_q;
            float v = 0.0f;
            if (n < a.n_valid) {
                const float *s = a.slab + (int64_t)a.nchunks * a.NP * a.KP + a.row0 + n;
                for (int c = c_lo; c < c_hi; ++c) v += s[(int64_t)c * a.NP];
            }
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (sub == 0 && n < a.n_valid) a.dst_b[n] = v * unscale;
        }
    }
}

// ---- gradients of the LayerNorm affine parameters (vanilla_dynamic/network.py:46-48, :275-277):
//   d(weight)[c] = sum_points g_y[c, p] * xhat[c, p],   d(bias)[c] = sum_points g_y[c, p]
// from the T8 sections the dgrad chain (g_y) and the training forward (xhat) wrote.  HBM-bound (8 B per channel and point),
// two stages with fixed summation orders (bit-reproducible): a block = (chunk of tiles, layer), thread = channel -- for a
// fixed channel a tile holds 4 runs of 8 points, and neighbouring threads read neighbouring 32-byte runs.
constexpr int kLnEntries = NRF_MAX_D + 1;     // trunk layers + the view layer
constexpr int kLnChunks = 64;
struct LnGradEntry {
    int64_t g_sec, a_sec;
    int32_t C, n_valid;
    float *dst_w, *dst_b;
};
struct LnGradArgs {
    const float *gbuf, *acts;
    float *partial;          // [entry][chunk][2][256]
    int64_t g_tile, a_tile, ntiles;
    int32_t chunk_tiles, nchunks;
    LnGradEntry e[kLnEntries];
};

__global__ void __launch_bounds__(256) ln_param_grad_kernel(const LnGradArgs a) {
    const LnGradEntry &e = a.e[blockIdx.y];
    const int c = threadIdx.x, chunk = blockIdx.x;
    float sw = 0.0f, sb = 0.0f;
    if (c < e.C) {
        const int64_t t0 = (int64_t)chunk * a.chunk_tiles, t1 = min(a.ntiles, t0 + a.chunk_tiles);
        for (int64_t tile = t0; tile < t1; ++tile) {
            const float *g = a.gbuf + tile * a.g_tile + e.g_sec + c * 8;
            const float *x = a.acts + tile * a.a_tile + e.a_sec + c * 8;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g + m * e.C * 8), g1 = *reinterpret_cast<const f32x4 *>(g + m * e.C * 8 + 4);
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(x + m * e.C * 8), x1 = *reinterpret_cast<const f32x4 *>(x + m * e.C * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { sb += g0[i]; sw = fmaf(g0[i], x0[i], sw); }
#pragma unroll
                for (int i = 0; i < 4; ++i) { sb += g1[i]; sw = fmaf(g1[i], x1[i], sw); }
            }
        }
    }
    float *out = a.partial + ((int64_t)blockIdx.y * a.nchunks + chunk) * 512;
    out[c] = sw;
    out[256 + c] = sb;
}

__global__ void __launch_bounds__(256) ln_param_reduce_kernel(const LnGradArgs a) {
    const LnGradEntry &e = a.e[blockIdx.x];
    const int c = threadIdx.x;
    if (c >= e.n_valid) return;
    const float *in = a.partial + (int64_t)blockIdx.x * a.nchunks * 512;
    float sw = 0.0f, sb = 0.0f;
    for (int k = 0; k < a.nchunks; ++k) { sw += in[k * 512 + c]; sb += in[k * 512 + 256 + c]; }
    e.dst_w[c] = sw;
    e.dst_b[c] = sb;
}

// Training mode "w16": bits of max |g_raw| over the pass (non-negative floats order as unsigned integers; *out zeroed by
// the host side before the launch; the maximum does not depend on the order of the atomics)
__global__ void __launch_bounds__(256) gmax_kernel(const float *__restrict__ g, int64_t n, unsigned *out) {
    __shared__ unsigned part[4];
    unsigned m = 0u;
    // float4 loads, four in flight per lane, where g is 16-byte aligned; scalars for the rest
    const int64_t nq = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? n >> 2 : 0, stride = (int64_t)gridDim.x * blockDim.x;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nq; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = g4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) m = max(m, __float_as_uint(v[u][c]) & 0x7fffffffu);
    }
    for (; i < nq; i += stride) {
        const f32x4 v = g4[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) m = max(m, __float_as_uint(v[c]) & 0x7fffffffu);
    }
    for (i = 4 * nq + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
        m = max(m, __float_as_uint(g[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m != 0u) atomicMax(out, m);     // one per block
    }
}

struct Bwd {
    const nrf_net_t *net;
    const nrf_net_t *grads;
    NetPlan pl;
    TrainLayout tl;
    const float *acts;
    const float *gbuf;
    float *slab;
    int64_t ntiles;
    int nchunks, chunk_tiles;
    hipStream_t st;
    int64_t slab_off = 0, slab_cap = 0;    // floats: every GEMM gets its own region of `slab`
    bool x3 = false;                       // wgrad on the bf16 matrix pipe (three bf16 MFMAs per product)
    bool xh = false;                       // x3 only: the activation sections hold f16 (TrainLayout::half_acts)
    const unsigned *gmax = nullptr;        // "w16": gradient sections hold scaled f16 too; the weight GEMMs are one f16 MFMA
    ReduceBatch batch;
    int nbatch = 0, max_blocks = 1;
    WgradArgs wb{};
    int nent = 0, npairs = 0;
    int pair_size[kWgPairs];

    int flush() {   // the GEMMs queued so far in one launch, then their reductions in one launch
        if (nent > 0) {
            // largest super-blocks first (stable insertion sort: equal sizes keep layer order)
            for (int i = 1; i < npairs; ++i) {
                const uint16_t c = wb.pair[i];
                const int sz = pair_size[i];
                int j = i;
                for (; j > 0 && pair_size[j - 1] < sz; --j) { wb.pair[j] = wb.pair[j - 1]; pair_size[j] = pair_size[j - 1]; }
                wb.pair[j] = c; pair_size[j] = sz;
            }
            wb.gbuf = gbuf; wb.acts = acts; wb.g_tile = tl.g_tile; wb.a_tile = tl.a_tile; wb.ntiles = ntiles;
            wb.chunk_tiles = chunk_tiles; wb.nchunks = nchunks;
            const unsigned blocks = (unsigned)npairs * (unsigned)nchunks;
            if (gmax) hipLaunchKernelGGL((wgrad_kernel<true, true, true>), dim3(blocks), dim3(256), 0, st, wb);
            else if (x3 && xh) hipLaunchKernelGGL((wgrad_kernel<true, true>), dim3(blocks), dim3(256), 0, st, wb);
            else if (x3) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(blocks), dim3(256), 0, st, wb);
            else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(blocks), dim3(256), 0, st, wb);
            nent = 0; npairs = 0;
            int rc = check_launch("nrf_query_bwd (wgrad)");
            if (rc) return rc;
        }
        if (nbatch == 0) return NRF_OK;
        batch.gmax = gmax;
        hipLaunchKernelGGL(reduce_unpack_kernel, dim3(max_blocks, nbatch), dim3(256), 0, st, batch);
        nbatch = 0; max_blocks = 1;
        return check_launch("nrf_query_bwd (slab reduction)");
    }

    // One wgrad GEMM (a whole nn.Linear: G x cat[X parts]); its reductions are queued for flush().
    struct Dst { int row0, n_valid; const float *w; const float *b; };
    struct XPart { int64_t sec; int C, tiles, type, param, ks, k_valid, ch_off; };   // a T8 section and how its columns unpack
    int run(int64_t g_sec, int Cg, int n_tiles, const XPart *xp, int nparts, int ld, const Dst *dsts, int ndst) {
        const int k_tiles = xp[0].tiles + (nparts > 1 ? xp[1].tiles : 0);
        const int n_sb = (n_tiles + 7) / 8, k_sb = (k_tiles + 7) / 8;
        if (nent == kWgEntries || npairs + n_sb * k_sb > kWgPairs || nbatch + ndst * nparts > kReduceBatch) {
            int rc = flush();
            if (rc) return rc;
        }
        float *region = slab + slab_off;
        WgradEntry &w = wb.e[nent];
        w = WgradEntry{};
        w.slab = region; w.g_sec = g_sec; w.Cg = Cg; w.n_tiles = n_tiles;
        w.a_sec[0] = xp[0].sec; w.Ca[0] = xp[0].C; w.k_tiles0 = xp[0].tiles; w.k_tiles = k_tiles;
        w.a_sec[1] = xp[0].sec; w.Ca[1] = xp[0].C;
        w.x_pe[0] = w.x_pe[1] = xp[0].type != BLK_HIDDEN;
        if (nparts > 1) { w.a_sec[1] = xp[1].sec; w.Ca[1] = xp[1].C; w.x_pe[1] = xp[1].type != BLK_HIDDEN; }
        slab_off += (int64_t)nchunks * ((int64_t)n_tiles * 32 * k_tiles * 32 + n_tiles * 32);
        if (slab_off > slab_cap) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd: slab workspace too small (internal)");
        for (int sb = 0; sb < n_sb * k_sb; ++sb) {
            const int gn = std::min(8, n_tiles - (sb / k_sb) * 8), xn = std::min(8, k_tiles - (sb % k_sb) * 8);
            pair_size[npairs] = gn * xn;
            wb.pair[npairs++] = (uint16_t)(nent << 8 | sb);
        }
        ++nent;
        for (int i = 0; i < ndst; ++i) {
            int kc = 0;
            for (int q = 0; q < nparts; ++q) {
                ReduceArgs &r = batch.e[nbatch++];
                r = ReduceArgs{};
                r.slab = region; r.dst_w = const_cast<float *>(dsts[i].w);
                r.dst_b = q == 0 ? const_cast<float *>(dsts[i].b) : nullptr;
                r.nchunks = nchunks; r.NP = n_tiles * 32; r.KP = k_tiles * 32; r.row0 = dsts[i].row0;
                r.n_valid = dsts[i].n_valid; r.ld = ld; r.ch_off = xp[q].ch_off;
                r.x_type = xp[q].type; r.x_param = xp[q].param; r.x_ks = xp[q].ks; r.k_valid = xp[q].k_valid;
                r.kc = kc; r.kn = xp[q].tiles * 32;
                kc += r.kn;
                int total_q = r.n_valid * (r.kn / 4) + r.n_valid + 4;
                max_blocks = std::max(max_blocks, (int)grid_for(total_q, 64, 4096));
            }
        }
        return NRF_OK;
    }
};

}  // namespace nrf

using namespace nrf;

// Partial slabs per GEMM: every chunk costs a slab of dW bytes written and read back, so small passes take fewer (a
// chunk keeps >= kWgChunkTiles tiles while that leaves >= 32 chunks); the count is a function of the pass size
// alone, so gradients stay bit-reproducible.
// Measured (one box, ms per step f32 / bf16x3): 512 rays 4.45 / 2.64 with 256 chunks -> 4.02 / 2.07; 32, 64 and 128 tiles
// per chunk within 1 %, 256 and more slower again (too few workgroups).
// The bf16-pipe weight-gradient kernels are 2-4x faster than the f32 one, so the per-chunk costs (a 256 KiB slab written per
// block and GEMM, read back by the reduction; the copy pipeline's fill and drain) weigh more: fewer, longer chunks.  Measured,
// ms per 4096-ray step, wgrad + reduction: w16 2.92 / 2.83 / 2.69 / 2.55 / 2.67 / 2.64 at 256 / 192 / 128 / 64 / 48 / 32 chunks;
// a16 4.73 / 4.60 / 4.73 at 256 / 128 / 64.
#ifndef NRF_WG_H_MAXCHUNKS
#define NRF_WG_H_MAXCHUNKS 64
#endif
#ifndef NRF_WG_X3_MAXCHUNKS
#define NRF_WG_X3_MAXCHUNKS 128
#endif
constexpr int kWgChunkTiles = 64;
static int64_t pick_chunks(int64_t ntiles, int *chunk_tiles, int max_chunks = 256) {
    int64_t want = std::max<int64_t>(32, ntiles / kWgChunkTiles);
    int64_t nchunks = std::min<int64_t>(std::min<int64_t>(ntiles, max_chunks), want);
    *chunk_tiles = (int)((ntiles + nchunks - 1) / nchunks);
    return (ntiles + *chunk_tiles - 1) / *chunk_tiles;
}

extern "C" int64_t nrf_train_grads_floats(const nrf_net_t *net, int64_t n_points) {
    if (!net || net->W < 2 || net->W > 256 || n_points < 0) return -1;
    TrainLayout tl = make_train_layout((net->W + 31) / 32, net->D, net->use_viewdirs, net->use_layernorm);
    return ((n_points + 31) / 32) * tl.g_tile;
}

constexpr int kGmaxFloats = 64;     // the slab workspace ends with the scalar of gmax_kernel (training mode "w16")
// LayerNorm nets: the partial sums of ln_param_grad_kernel live behind the weight-gradient slabs
static int64_t ln_partial_floats(const nrf_net_t *net) {
    return net->use_layernorm ? (int64_t)kLnEntries * kLnChunks * 512 : 0;
}

static int64_t slab_floats(const nrf_net_t *net, int64_t n_points, int max_chunks) {
    if (!net || net->W < 2 || net->W > 256 || n_points < 0) return -1;
    int ct;
    int64_t nchunks = pick_chunks((n_points + 31) / 32, &ct, max_chunks);
    // every GEMM of the pass (<= D trunk layers + 3 heads) keeps its own slab region; bound on each:
    // (feature + alpha row) x cat[hidden, gamma(x), gamma(t)]
    const int64_t wp = (net->W + 31) / 32 * 32;      // the kernels' width: whole 32-channel tiles
    int64_t np = wp + 32, kp = wp + 2 * kKsX + 32;
    return nchunks * (net->D + 3) * (np * kp + np) + ln_partial_floats(net) + kGmaxFloats;
}

extern "C" int64_t nrf_train_slab_floats(const nrf_net_t *net, int64_t n_points) { return slab_floats(net, n_points, 256); }

// time_ch > 0: `net` describes a deformation ("time") net (layer 0 also reads gamma(t)).  pts/g_pts non-null: also
// return d(loss)/d(pts) (canonical nets with the fused encoding).
static int query_bwd_impl(const nrf_net_t *net, const nrf_embed_t *emb, int time_ch, const float *packed,
                          const float *g_raw, const float *acts, float *gbuf, float *slab, int64_t n_points,
                          const nrf_net_t *grads, const float *pts, float *g_pts, void *stream, int max_chunks = 256,
                          bool x3 = false, const void *packed_bwd_x3 = nullptr, bool straight = false, bool half_acts = false,
                          bool half_grads = false) {
    NRF_REQUIRE(net && emb && grads, "nrf_query_bwd: null descriptor");
    if (n_points != 0) NRF_REQUIRE(net && emb && packed && g_raw && acts && gbuf && slab && grads, "nrf_query_bwd: null pointer");
    NRF_REQUIRE(n_points >= 0, "nrf_query_bwd: negative n_points");
    Planner P{};
    const char *e = plan_net(P, net, emb, time_ch, straight);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd: %s", e);
    if (P.plan.layernorm && (x3 || straight || (P.plan.nt != 2 && P.plan.nt != 4 && P.plan.nt != 8)))
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd: the LayerNorm backward is built in f32 for W = 64, 128, 256");
    if (n_points == 0) return NRF_OK;
    const bool want_gx = g_pts != nullptr;
    if (want_gx) {
        NRF_REQUIRE(pts, "nrf_query_bwd_input: pts is null");
        if (P.plan.xT_base < 0)
            return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd_input: input gradients need the fused encoding (embed mode 0)");
    }
    const NetPlan &pl = P.plan;
    // W: the net's width (valid rows / columns of the gradients); Wp: the kernels' width in whole 32-channel tiles (the
    // T8 sections' channel count; equal unless W is zero-padded)
    const int nt = pl.nt, W = net->W, Wp = 32 * nt, D = net->D, ic = net->input_ch, icv = net->input_ch_views;
    hipStream_t st = as_stream(stream);
    DgradArgs d{};
    if (half_acts && (!x3 || !packed_bwd_x3 || want_gx || straight || time_ch || pl.layernorm || emb->mode != 0))
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd: the f16 activation workspace belongs to the static nets' bf16x3 "
                         "training kernels (parameter gradients only)");
    NRF_REQUIRE(!half_grads || half_acts, "nrf_query_bwd: the f16 gradient workspace goes with the f16 activation workspace");
    d.plan = pl; d.tl = make_train_layout(nt, D, pl.use_viewdirs, pl.layernorm, half_acts, half_grads);
    // "w16": the scalar the chain scales by lives in the last kGmaxFloats of the slab workspace
    unsigned *gmax = nullptr;
    if (half_grads) {
        gmax = reinterpret_cast<unsigned *>(slab + slab_floats(net, n_points, max_chunks) - kGmaxFloats);
        hipError_t me = hipMemsetAsync(gmax, 0, sizeof(unsigned), st);
        if (me != hipSuccess) return nrf::fail(NRF_ERR_LAUNCH, "nrf_query_bwd: hipMemsetAsync: %s", hipGetErrorString(me));
        const int64_t nraw = n_points * pl.output_ch;
        hipLaunchKernelGGL(gmax_kernel, dim3((unsigned)std::min<int64_t>((nraw + 1023) / 1024, 512)), dim3(256), 0, st, g_raw, nraw, gmax);
    }
    d.packed = packed; d.g_raw = g_raw; d.acts = acts; d.gbuf = gbuf; d.n_points = n_points;
    const int64_t ntiles = (n_points + 31) / 32;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > 256) blocks = 256;
    int x3rc = NRF_ERR_UNSUPPORTED;
    // the chain on the bf16 matrix pipe, where a kernel exists: static nets, and the deformation net (its chain ends at its
    // own parameters: nothing downstream amplifies the 1e-5 of this arithmetic, unlike the canonical net's d/dx)
    if (x3 && packed_bwd_x3 && !want_gx && !straight)
        x3rc = launch_dgrad_x3(net, emb, pl, packed, packed_bwd_x3, g_raw, acts, gbuf, n_points, st, half_acts, gmax);
    if (half_acts && x3rc != NRF_OK)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd_bf16x3_a16: this shape has no bf16x3 kernels (W = 128, 256; D >= 2)");
    if (x3rc != NRF_OK && x3rc != NRF_ERR_UNSUPPORTED) return x3rc;
    if (x3rc == NRF_OK) goto dgrad_done;
    if (pl.layernorm) {
        switch (want_gx ? -nt : nt) {
            case 8: hipLaunchKernelGGL((dgrad_kernel<8, false, true>), dim3(blocks), dim3(256), 0, st, d); break;
            case 4: hipLaunchKernelGGL((dgrad_kernel<4, false, true>), dim3(blocks), dim3(256), 0, st, d); break;
            case 2: hipLaunchKernelGGL((dgrad_kernel<2, false, true>), dim3(blocks), dim3(256), 0, st, d); break;
            case -8: hipLaunchKernelGGL((dgrad_kernel<8, true, true>), dim3(blocks), dim3(256), 0, st, d); break;
            case -4: hipLaunchKernelGGL((dgrad_kernel<4, true, true>), dim3(blocks), dim3(256), 0, st, d); break;
            default: hipLaunchKernelGGL((dgrad_kernel<2, true, true>), dim3(blocks), dim3(256), 0, st, d); break;
        }
        goto dgrad_done;
    }
    switch (want_gx ? -nt : nt) {
        case 8: hipLaunchKernelGGL(dgrad_kernel<8>, dim3(blocks), dim3(256), 0, st, d); break;
        case 4: hipLaunchKernelGGL(dgrad_kernel<4>, dim3(blocks), dim3(256), 0, st, d); break;
        case 2: hipLaunchKernelGGL(dgrad_kernel<2>, dim3(blocks), dim3(256), 0, st, d); break;
        case 1: hipLaunchKernelGGL(dgrad_kernel<1>, dim3(blocks), dim3(256), 0, st, d); break;
        case 3: hipLaunchKernelGGL(dgrad_kernel<3>, dim3(blocks), dim3(256), 0, st, d); break;
        case 5: hipLaunchKernelGGL(dgrad_kernel<5>, dim3(blocks), dim3(256), 0, st, d); break;
        case 6: hipLaunchKernelGGL(dgrad_kernel<6>, dim3(blocks), dim3(256), 0, st, d); break;
        case 7: hipLaunchKernelGGL(dgrad_kernel<7>, dim3(blocks), dim3(256), 0, st, d); break;
        case -1: hipLaunchKernelGGL((dgrad_kernel<1, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -3: hipLaunchKernelGGL((dgrad_kernel<3, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -5: hipLaunchKernelGGL((dgrad_kernel<5, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -6: hipLaunchKernelGGL((dgrad_kernel<6, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -7: hipLaunchKernelGGL((dgrad_kernel<7, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -8: hipLaunchKernelGGL((dgrad_kernel<8, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -4: hipLaunchKernelGGL((dgrad_kernel<4, true>), dim3(blocks), dim3(256), 0, st, d); break;
        case -2: hipLaunchKernelGGL((dgrad_kernel<2, true>), dim3(blocks), dim3(256), 0, st, d); break;
        default: return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd: W = %d has no fused kernel", W);
    }
dgrad_done:
    int rc = check_launch("nrf_query_bwd (dgrad)");
    if (rc) return rc;
    if (want_gx) {
        hipLaunchKernelGGL(pe3_bwd_t8_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, gbuf, d.tl.g_tile,
                           d.tl.g_xin, pts, pl.x_param, n_points, g_pts);
        rc = check_launch("nrf_query_bwd_input (encoding backward)");
        if (rc) return rc;
    }

    Bwd B{};
    B.net = net; B.grads = grads; B.pl = pl; B.tl = d.tl; B.acts = acts; B.gbuf = gbuf; B.slab = slab;
    B.ntiles = ntiles; B.st = st; B.x3 = x3; B.xh = half_acts; B.gmax = gmax;
    B.nchunks = (int)pick_chunks(ntiles, &B.chunk_tiles, half_grads ? std::min(max_chunks, NRF_WG_H_MAXCHUNKS) : (x3 ? std::min(max_chunks, NRF_WG_X3_MAXCHUNKS) : max_chunks));
    B.slab_cap = slab_floats(net, n_points, max_chunks) - ln_partial_floats(net) - kGmaxFloats;
    if (pl.layernorm) {   // the affine parameters' gradients: independent of the weight-gradient GEMMs below
        LnGradArgs la{};
        la.gbuf = gbuf; la.acts = acts; la.partial = slab + B.slab_cap;     // (then the kGmaxFloats tail)
        la.g_tile = d.tl.g_tile; la.a_tile = d.tl.a_tile; la.ntiles = ntiles;
        la.nchunks = (int)std::min<int64_t>(ntiles, kLnChunks);
        la.chunk_tiles = (int)((ntiles + la.nchunks - 1) / la.nchunks);
        la.nchunks = (int)((ntiles + la.chunk_tiles - 1) / la.chunk_tiles);
        int ne = 0;
        for (int l = 0; l < D; ++l) {
            NRF_REQUIRE(grads->ln_w[l] && grads->ln_b[l], "nrf_query_bwd: grads->ln_w/ln_b[%d] is null", l);
            la.e[ne++] = LnGradEntry{d.tl.g_ln + (int64_t)l * 32 * Wp, d.tl.a_xhat + (int64_t)l * 32 * Wp, Wp, W,
                                     const_cast<float *>(grads->ln_w[l]), const_cast<float *>(grads->ln_b[l])};
        }
        if (pl.use_viewdirs) {
            NRF_REQUIRE(grads->views_ln_w && grads->views_ln_b, "nrf_query_bwd: grads->views_ln_w/views_ln_b is null");
            la.e[ne++] = LnGradEntry{d.tl.g_ln_v, d.tl.a_xhat_v, W / 2, W / 2, const_cast<float *>(grads->views_ln_w),
                                     const_cast<float *>(grads->views_ln_b)};
        }
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3(la.nchunks, ne), dim3(256), 0, st, la);
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ne), dim3(256), 0, st, la);
        rc = check_launch("nrf_query_bwd (LayerNorm parameter gradients)");
        if (rc) return rc;
    }
    const TrainLayout &tl = d.tl;
    const int pp = tl.half_acts ? 16 : 32;             // floats per channel of an activation section (f16: two bytes a value)
    const int xk = 2 * kKsX / 32, dk = 2 * kKsD / 32;  // 32-channel tiles of the k-space input blocks
    const Bwd::XPart Xin{tl.a_xin, 2 * kKsX, xk, pl.x_type, pl.x_param, kKsX, 0, 0};
    for (int l = 0; l < D; ++l) {
        NRF_REQUIRE(grads->pts_w[l] && grads->pts_b[l], "nrf_query_bwd: grads->pts_w/pts_b[%d] is null", l);
        const int64_t gsec = tl.g_trunk + (int64_t)l * (tl.half_grads ? 16 : 32) * Wp;
        bool skip = l > 0 && ((net->skips_mask >> (l - 1)) & 1);
        const int skip_ch = ic + (straight ? time_ch : 0);     // a straightforward net's skip reads cat[gamma(x), gamma(t), h]
        int ld = l == 0 ? ic + time_ch : (W + (skip ? skip_ch : 0));
        Bwd::Dst dst{0, W, grads->pts_w[l], grads->pts_b[l]};
        const Bwd::XPart Tin{tl.a_tin, 2 * kKsT, 1, pl.t_type, pl.t_param, kKsT, 0, ic};   // 24 k-space channels of gamma(t)
        Bwd::XPart parts[2];
        int np = 0;
        if (l == 0) {
            parts[np++] = Xin;
            if (time_ch) parts[np++] = Tin;   // columns fed by gamma(t) (vanilla_dynamic/network.py:296, :176)
        } else {
            parts[np++] = Bwd::XPart{tl.a_trunk + (int64_t)(l - 1) * pp * Wp, Wp, nt, BLK_HIDDEN, 0, 0, W, skip ? skip_ch : 0};
            if (skip) parts[np++] = Xin;   // network.py:52-53: cat[input_pts, h]
        }
        rc = B.run(gsec, Wp, nt, parts, np, ld, &dst, 1);
        if (rc) return rc;
        if (l > 0 && skip && straight) {   // the gamma(t) columns of the skip layer: a GEMM of their own (no bias)
            Bwd::Dst dt{0, W, grads->pts_w[l], nullptr};
            rc = B.run(gsec, Wp, nt, &Tin, 1, ld, &dt, 1);
            if (rc) return rc;
        }
    }
    const int64_t a_last = tl.a_trunk + (int64_t)(D - 1) * pp * Wp;
    const Bwd::XPart Hlast{a_last, Wp, nt, BLK_HIDDEN, 0, 0, W, 0};
    if (pl.use_viewdirs) {
        NRF_REQUIRE(grads->feature_w && grads->feature_b && grads->alpha_w && grads->alpha_b && grads->views_w &&
                    grads->views_b && grads->rgb_w && grads->rgb_b, "nrf_query_bwd: a head gradient pointer is null");
        Bwd::Dst fd[2] = {{0, W, grads->feature_w, grads->feature_b}, {Wp, 1, grads->alpha_w, grads->alpha_b}};   // d/d(alpha): row Wp
        rc = B.run(tl.g_feat, Wp + 32, nt + 1, &Hlast, 1, W, fd, 2);
        if (rc) return rc;
        const int nv = (nt + 1) / 2;          // tiles of the view layer: W/2 live channels, zero-padded to 32 * nv
        Bwd::Dst vd{0, W / 2, grads->views_w, grads->views_b};
        Bwd::XPart vparts[2] = {{tl.a_feat, Wp, nt, BLK_HIDDEN, 0, 0, W, 0},
                                {tl.a_din, 2 * kKsD, dk, pl.d_type, pl.d_param, kKsD, 0, W}};   // network.py:58: cat[feature, views]
        rc = B.run(tl.g_hv, 32 * nv, nv, vparts, 2, W + icv, &vd, 1);
        if (rc) return rc;
        Bwd::Dst rd{0, 3, grads->rgb_w, grads->rgb_b};
        Bwd::XPart hv{tl.a_hv, 32 * nv, nv, BLK_HIDDEN, 0, 0, W / 2, 0};
        rc = B.run(tl.g_head, 32, 1, &hv, 1, W / 2, &rd, 1);
        if (rc) return rc;
    } else {
        NRF_REQUIRE(grads->out_w && grads->out_b, "nrf_query_bwd: grads->out_w/out_b is null");
        Bwd::Dst od{0, pl.output_ch, grads->out_w, grads->out_b};
        rc = B.run(tl.g_head, 32, 1, &Hlast, 1, W, &od, 1);
        if (rc) return rc;
    }
    return B.flush();
}

extern "C" int nrf_query_bwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                             const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                             void *stream) {
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, gbuf, slab, n_points, grads, nullptr, nullptr, stream);
}

extern "C" int nrf_query_bwd_input(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                                   const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                                   const float *pts, float *g_pts, void *stream) {
    NRF_REQUIRE(pts && g_pts, "nrf_query_bwd_input: pts / g_pts is null");
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, gbuf, slab, n_points, grads, pts, g_pts, stream);
}

extern "C" int nrf_query_bwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed,
                                    const void *packed_bwd_x3, const float *g_raw, const float *acts, float *gbuf,
                                    float *slab, int64_t n_points, const nrf_net_t *grads, const float *pts, float *g_pts,
                                    void *stream) {
    NRF_REQUIRE((pts == nullptr) == (g_pts == nullptr), "nrf_query_bwd_bf16x3: pts and g_pts go together");
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, gbuf, slab, n_points, grads, pts, g_pts, stream, 256, true,
                          packed_bwd_x3);
}

/* nrf_query_bwd_bf16x3 for a workspace nrf_query_fwd_train_bf16x3_a16 wrote (activations saved as f16): the weight-gradient
 * GEMMs read half the bytes for their X operand.  Parameter gradients of static nets; weight gradients then carry the
 * activations' f16 rounding (2^-12 relative per value: ~2e-4 of a gradient's norm, the gradient tests' bar is 2e-3). */
extern "C" int nrf_query_bwd_bf16x3_a16(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed,
                                        const void *packed_bwd_x3, const float *g_raw, const float *acts, float *gbuf,
                                        float *slab, int64_t n_points, const nrf_net_t *grads, void *stream) {
    NRF_REQUIRE(packed_bwd_x3, "nrf_query_bwd_bf16x3_a16: packed_bwd_x3 is null");
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, gbuf, slab, n_points, grads, nullptr, nullptr, stream, 256, true,
                          packed_bwd_x3, false, true);
}

/* Training mode "w16": nrf_query_bwd_bf16x3_a16 with the gradient blocks saved as f16 as well -- the activation-gradient
 * chain (bf16x3, unchanged arithmetic) runs on gradients scaled by the power of two that brings the pass's largest
 * |g_raw| into [2^10, 2^11) and saves its blocks as f16 (saturating), the weight-gradient GEMMs are ONE f16 MFMA per tile pair
 * on the two f16 workspaces, and the slab reduction undoes the scale.  What mixed-precision training calls loss scaling;
 * weight AND bias gradients then carry a 2^-12 rounding of both operands. */
extern "C" int nrf_query_bwd_bf16x3_w16(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed,
                                        const void *packed_bwd_x3, const float *g_raw, const float *acts, float *gbuf,
                                        float *slab, int64_t n_points, const nrf_net_t *grads, void *stream) {
    NRF_REQUIRE(packed_bwd_x3, "nrf_query_bwd_bf16x3_w16: packed_bwd_x3 is null");
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, gbuf, slab, n_points, grads, nullptr, nullptr, stream, 256, true,
                          packed_bwd_x3, false, true, true);
}

extern "C" int nrf_dnerf_time_bwd_bf16x3(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time,
                                         const void *packed_bwd_x3, const float *g_dx, const float *acts, float *gbuf,
                                         float *slab, int64_t n_points, const nrf_net_t *grads, void *stream) {
    NRF_REQUIRE(time_net && !time_net->use_viewdirs && time_net->input_ch_views > 0,
                "nrf_dnerf_time_bwd_bf16x3: not a time-net descriptor");
    return query_bwd_impl(time_net, emb, time_net->input_ch_views, packed_time, g_dx, acts, gbuf, slab, n_points, grads,
                          nullptr, nullptr, stream, 256, true, packed_bwd_x3);
}

extern "C" int nrf_dnerf_time_bwd(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time,
                                  const float *g_dx, const float *acts, float *gbuf, float *slab, int64_t n_points,
                                  const nrf_net_t *grads, void *stream) {
    NRF_REQUIRE(time_net && !time_net->use_viewdirs && time_net->input_ch_views > 0,
                "nrf_dnerf_time_bwd: not a time-net descriptor");
    return query_bwd_impl(time_net, emb, time_net->input_ch_views, packed_time, g_dx, acts, gbuf, slab, n_points, grads,
                          nullptr, nullptr, stream);
}

/* DNeRF_Straightforward training: parameter gradients from the workspace nrf_dnerf_straight_fwd_train saved
 * (net->reserved = input_ch_time; `packed` from nrf_dnerf_straight_pack_weights). */
extern "C" int nrf_dnerf_straight_bwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                                      const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                                      void *stream) {
    NRF_REQUIRE(net && net->use_viewdirs && net->reserved > 0, "nrf_dnerf_straight_bwd: not a straightforward-net descriptor");
    return query_bwd_impl(net, emb, net->reserved, packed, g_raw, acts, gbuf, slab, n_points, grads, nullptr, nullptr, stream,
                          256, false, nullptr, true);
}

// ------------------------------------------------------------------------------------------------
// Allocation-free variant of nrf_query_bwd for callers that cannot allocate device memory (code under HIP graph
// capture, tight serving loops, harnesses that forbid hipMalloc): the two backward workspaces (`gbuf`, `slab`) live in
// a file-scope device array instead of caller-provided buffers.  Every GPU the library runs on holds its own copy of
// the array (64 MiB) until the process ends; the array keeps the previous call's contents -- nothing in this pass needs
// zeroed scratch (dgrad writes every gradient block of every tile before wgrad reads it, wgrad writes every slab entry
// the reduction reads) -- and calls that overlap on one GPU race on it.
constexpr int64_t kStaticScratchFloats = (int64_t)16 << 20;      // 64 MiB
constexpr int kStaticMaxChunks = 8;                              // wgrad partial slabs per GEMM (256 in nrf_query_bwd)
static __device__ __align__(256) float scratch[kStaticScratchFloats];

static std::mutex g_scratch_mu;
static float *g_scratch_ptr[64] = {};   // per device: each GPU has the array at its own address

// `scratch` in host code is a stand-in symbol, not a device address: resolve it once per device, with that device current.
static int static_scratch(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64)
        return nrf::fail(NRF_ERR_LAUNCH, "nrf_query_bwd_static: hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return nrf::fail(NRF_ERR_LAUNCH, "nrf_query_bwd_static: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                             hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float *>(p);
    }
    *out = g_scratch_ptr[dev];
    return NRF_OK;
}

extern "C" int nrf_query_bwd_static(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                                    const float *acts, int64_t n_points, const nrf_net_t *grads, void *stream) {
    NRF_REQUIRE(net && emb && grads, "nrf_query_bwd_static: null descriptor");
    NRF_REQUIRE(n_points >= 0, "nrf_query_bwd_static: negative n_points");
    const int64_t raw_g = nrf_train_grads_floats(net, n_points);
    const int64_t need_s = slab_floats(net, n_points, kStaticMaxChunks);
    if (raw_g < 0 || need_s < 0)      // tested BEFORE the rounding below, which would turn the -1 of an unsupported shape into 0
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_bwd_static: unsupported network shape (D %d, W %d: the backward "
                         "kernels take W up to 256 and D <= 16)", net->D, net->W);
    const int64_t need_g = (raw_g + 63) & ~(int64_t)63;     // keep `slab` 256-byte aligned
    if (need_g + need_s > kStaticScratchFloats) {
        // the slab part is constant beyond kStaticMaxChunks tiles; gbuf grows by g_tile floats per 32 points
        const int64_t per_tile = nrf_train_grads_floats(net, 32);
        const int64_t fixed = slab_floats(net, (int64_t)32 * kStaticMaxChunks, kStaticMaxChunks);
        const int64_t max_pts = per_tile > 0 && fixed < kStaticScratchFloats ? (kStaticScratchFloats - fixed - 64) / per_tile * 32 : 0;
        return nrf::fail(NRF_ERR_UNSUPPORTED,
                         "nrf_query_bwd_static: %lld points need %lld bytes of backward workspace, the static scratch holds "
                         "%lld (at most %lld points for this network); use nrf_query_bwd with caller-owned workspaces",
                         (long long)n_points, (long long)((need_g + need_s) * 4), (long long)(kStaticScratchFloats * 4),
                         (long long)max_pts);
    }
    float *base = nullptr;
    int rc = static_scratch(&base);
    if (rc) return rc;
    return query_bwd_impl(net, emb, 0, packed, g_raw, acts, base, base + need_g, n_points, grads, nullptr, nullptr, stream,
                          kStaticMaxChunks);
}
