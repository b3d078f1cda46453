// K2+K3: fused positional-encoding + NeRF MLP query on v_mfma_f32_32x32x2_f32 (exact f32), gfx950.
//
// Replaces run_network (render.py:31-56) + Embedder.embed (positional_encoding.py:35-36) +
// NeRF.forward (network.py:46-69).  Design (see mlp_layout.h for the register/weight layout):
//   * one wavefront = 32 points, all layers, no LDS, no barriers: the accumulators of layer l are the
//     B operand of layer l+1 as they stand (transposed product, k order baked into the packed weights);
//   * weights are streamed from L2 as fully coalesced 1 KiB wave loads (16 B per lane = 4 k-steps of
//     one 32-channel output tile), one load per 4 MFMAs; 2.4 MB per net stays resident in every
//     XCD's 4 MiB L2;
//   * bias is the accumulator's initial value, ReLU is the v_max that moves the accumulator into the
//     next layer's operand registers, the 1-wide sigma head and the 3-wide rgb head are lane-local
//     dot products (VALU) finished with one cross-half shuffle;
//   * sin/cos use the accurate range-reducing device functions (arguments reach ~3000 rad).
// The kernel is MFMA-bound: 593 920 MAC per point issued (593 408 algorithmic, 99.9 %).
#include <array>
#include <type_traits>

#include "mlp_device.h"

namespace nrf {

struct QueryArgs {
    NetPlan plan;
    const float *packed;
    const float *pts;       // [P,3]   (mode 0)  |  x [P, x_stride] (mode 1)
    const float *viewdirs;  // [P / pts_per_ray, 3] (mode 0)
    float *raw;             // [P, output_ch]
    int64_t n_points;
    int32_t pts_per_ray, x_stride, input_ch, pad_;
    float *acts;            // training mode: [n_tiles][tl.a_tile] saved inputs / activations (T8 sections)
    TrainLayout tl;
    const float *times;     // deformation ("time") net: [P / t_div] scalar input of the second layer-0 block
    int64_t t_div;          // (straightforward net with embedded inputs: rows [P, t_stride] of gamma(t), t_div = 1)
    int32_t t_stride, pad2_;
};

// STRAIGHT: DNeRF_Straightforward -- the time block also feeds the skip layers (so it stays live through the trunk)
// LN: the use_layernorm variant (vanilla_dynamic/network.py:46-48): LayerNorm between every trunk Linear and its ReLU, and
// after the view layer; inference
template <int NT, bool EMB, bool TRAIN, bool TIME = false, bool STRAIGHT = false, bool LN = false>
__global__ void __launch_bounds__(256, 1) query_fwd_kernel(const QueryArgs a) {
    const NetPlan &pl = a.plan;
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WBuf wb;
    wb.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, (int)(pl.total_floats * 4), 0x00020000);
    wb.lane16 = lane * 16;
    wb.h16 = h * 16;
    // readfirstlane: make the wave index (and everything derived from it: tile, buffer descriptors)
    // provably wave-uniform, or every buffer op through a per-tile descriptor gets a waterfall loop
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (a.n_points + 31) >> 5;
    const uint32_t hid_off = NT * 32, skip_off = NT * 32 + 4 * NT * NT * 256;
    const uint32_t skipt_off = skip_off + (kKsX / 4) * NT * 256;

    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t p = tile * 32 + n;
        const int64_t pc = p < a.n_points ? p : a.n_points - 1;
        float xin[kKsX];
        if constexpr (EMB) {
            fill_raw<kKsX>(xin, a.pts + pc * a.x_stride, pl.x_param, h);
        } else {
            const float *q = a.pts + pc * 3;
            fill_pe3<kKsX>(xin, q[0], q[1], q[2], pl.x_param, h);
        }

        TBuf tb;
        if constexpr (TRAIN) {
            tb = make_tbuf(a.acts + tile * a.tl.a_tile, a.tl.a_tile, lane);
            store_in_t8<kKsX>(tb, (uint32_t)a.tl.a_xin, xin);
        }
        f32x16 hin[NT], acc[NT];
        // trunk (network.py:48-53)
        load_bias<NT>(acc, wb, (uint32_t)pl.l0_b);
        gemm_in<kKsX, NT>(acc, xin, wb, (uint32_t)pl.l0_w);
        if constexpr (TIME) {   // vanilla_dynamic/network.py:296: h = cat[gamma(x), gamma(t)]
            float tin[kKsT];
            if constexpr (EMB) fill_raw<kKsT>(tin, a.times + pc * a.t_stride, pl.t_param, h);
            else fill_pe1<kKsT>(tin, a.times[pc / a.t_div], pl.t_param, h);
            if constexpr (TRAIN) store_in_t8<kKsT>(tb, (uint32_t)a.tl.a_tin, tin);
            gemm_in<kKsT, NT>(acc, tin, wb, (uint32_t)pl.l0_w2);
        }
        // every trunk move also fetches the bias of the layer that follows (the next trunk slot, then feature_linear;
        // with no further NT-tile layer the refill is a harmless re-read of a valid slot)
        auto bias_after = [&](int l) -> uint32_t {
            if (l + 1 < pl.n_trunk) return (uint32_t)(pl.trunk_base + (int64_t)l * pl.trunk_stride);
            return pl.use_viewdirs ? (uint32_t)pl.feat_b : (uint32_t)pl.l0_b;
        };
        // out of the accumulators: ReLU (LN: LayerNorm first), and the next layer's bias goes in
        auto activate = [&](int l) {
            if constexpr (LN) {
                const uint32_t ln = (uint32_t)(pl.ln_base + (int64_t)l * pl.ln_stride);
                if constexpr (TRAIN)     // the backward differentiates the normalisation from (x - mean) * rstd and rstd
                    ln_relu_move<NT, true>(hin, acc, wb, ln, &tb, (uint32_t)a.tl.a_xhat + l * (32 * 32 * NT),
                                           (uint32_t)a.tl.a_rstd + l * 32);
                else
                    ln_relu_move<NT>(hin, acc, wb, ln);
                load_bias<NT>(acc, wb, bias_after(l));
            } else {
                relu_move_bias<NT>(hin, acc, wb, bias_after(l));
            }
        };
        activate(0);
        // TRAIN: a layer's activations (and their ReLU gate bits) are saved by the GEMM that consumes them, four
        // registers per k-group under that group's MFMAs (save_hin); only a layer with no consumer saves in a burst
        auto save_hin = [&](int l) {
            uint32_t sec = (uint32_t)a.tl.a_trunk + l * (32 * 32 * NT), slot = (uint32_t)a.tl.a_mask + l * 256;
            asm volatile("" : "+s"(sec), "+s"(slot));
            return [&tb, &hin, sec, slot, lane, m = std::array<uint32_t, 4>{}](int G) mutable {
                store_acc_t8_group<NT>(tb, sec, hin, G);
                gate_bits_group<NT>(tb, slot, hin, lane, reinterpret_cast<uint32_t(&)[4]>(*m.data()), G);
            };
        };
        // one trunk layer l >= 1; WITH_T (straightforward nets): its skip also reads gamma(t), network.py:176-177
        auto trunk_layer = [&](int l, auto with_t) {
            const uint32_t slot = (uint32_t)(pl.trunk_base + (int64_t)(l - 1) * pl.trunk_stride);   // acc holds its bias
            if ((pl.skip_in_mask >> l) & 1) gemm_in<kKsX, NT>(acc, xin, wb, slot + skip_off);
            if constexpr (decltype(with_t)::value) {   // gamma(t) is rebuilt here: 12 registers the trunk cannot keep live
                float tin[kKsT];
                if constexpr (EMB) fill_raw<kKsT>(tin, a.times + pc * a.t_stride, pl.t_param, h);
                else fill_pe1<kKsT>(tin, a.times[pc / a.t_div], pl.t_param, h);
                gemm_in<kKsT, NT>(acc, tin, wb, slot + skipt_off);
            }
            if constexpr (TRAIN) gemm_hid_side<NT, NT>(acc, hin, wb, slot + hid_off, save_hin(l - 1));
            else gemm_hid<NT, NT>(acc, hin, wb, slot + hid_off);
            activate(l);
        };
        if constexpr (STRAIGHT) {
            // the time block sits in straight-line code between two plain layer loops (inside the loop, under a runtime
            // condition, the register allocator gave up: 1100 spilled registers); one skip layer, as every shipped config
            const int ls = 31 - __builtin_clz((unsigned)pl.skip_in_mask | 1u);      // the skip layer (0: none)
            for (int l = 1; l < ls; ++l) trunk_layer(l, std::false_type{});
            if (ls >= 1) trunk_layer(ls, std::true_type{});
            for (int l = (ls >= 1 ? ls + 1 : 1); l < pl.n_trunk; ++l) trunk_layer(l, std::false_type{});
        } else {
            for (int l = 1; l < pl.n_trunk; ++l) trunk_layer(l, std::false_type{});
        }
        if (pl.use_viewdirs) {
            {
                constexpr int NV = (NT + 1) / 2;     // tiles of the view layer (W/2 outputs; zero-padded when NT is odd)
                const float alpha = head_dot<NT>(hin, wb, (uint32_t)pl.alpha_vec) + wload1(wb, (uint32_t)pl.alpha_b * 4);
                // network.py:57 -- feature_linear's bias is already in acc (fetched by the last trunk move)
                if constexpr (TRAIN) gemm_hid_side<NT, NT>(acc, hin, wb, (uint32_t)pl.feat_w, save_hin(pl.n_trunk - 1));
                else gemm_hid<NT, NT>(acc, hin, wb, (uint32_t)pl.feat_w);
                // the feature vector becomes an MFMA operand: move it out of the accumulator file (no ReLU)
                relu_move<NT>(hin, acc, false);
                f32x16 acc2[NV];
                load_bias<NV>(acc2, wb, (uint32_t)pl.views_b);                               // network.py:58-62
                if constexpr (TRAIN) {
                    uint32_t sec = (uint32_t)a.tl.a_feat;
                    asm volatile("" : "+s"(sec));
                    gemm_hid_side<NT, NV>(acc2, hin, wb, (uint32_t)pl.views_w_hid,
                                              [&](int G) { store_acc_t8_group<NT>(tb, sec, hin, G); });
                } else {
                    gemm_hid<NT, NV>(acc2, hin, wb, (uint32_t)pl.views_w_hid);
                }
                // the direction encoding is built here, not at the top: 16 registers the trunk cannot spare
                float din[kKsD];
                if constexpr (EMB) {
                    fill_raw<kKsD>(din, a.pts + pc * a.x_stride + a.input_ch, pl.d_param, h);
                } else {
                    const float *d = a.viewdirs + (pc / a.pts_per_ray) * 3;
                    fill_pe3<kKsD>(din, d[0], d[1], d[2], pl.d_param, h);
                }
                if constexpr (TRAIN) store_in_t8<kKsD>(tb, (uint32_t)a.tl.a_din, din);
                gemm_in<kKsD, NV>(acc2, din, wb, (uint32_t)pl.views_w_in);
                if constexpr (LN && TRAIN)                                                       // network.py:88-90
                    ln_relu_move<NV, true>(acc2, acc2, wb, (uint32_t)pl.views_ln, &tb, (uint32_t)a.tl.a_xhat_v,
                                           (uint32_t)a.tl.a_rstd + pl.n_trunk * 32);
                else if constexpr (LN) ln_relu_move<NV>(acc2, acc2, wb, (uint32_t)pl.views_ln);
                else relu_move<NV>(acc2, acc2, true);
                if constexpr (TRAIN) {
                    store_acc_t8<NV>(tb, (uint32_t)a.tl.a_hv, acc2);
                    store_gate_bits<NV>(tb, (uint32_t)a.tl.a_mask + pl.n_trunk * 256, acc2, lane);
                }
                const uint32_t rv = (uint32_t)pl.rgb_vec, rb = (uint32_t)pl.rgb_b * 4;
                float4 o;                                                                        // network.py:64-65
                o.x = head_dot<NV>(acc2, wb, rv + 0 * (NV) * 32) + wload1(wb, rb + 0);
                o.y = head_dot<NV>(acc2, wb, rv + 1 * (NV) * 32) + wload1(wb, rb + 4);
                o.z = head_dot<NV>(acc2, wb, rv + 2 * (NV) * 32) + wload1(wb, rb + 8);
                o.w = alpha;
                if (h == 0 && p < a.n_points) reinterpret_cast<float4 *>(a.raw)[p] = o;
            }
        } else {
            if constexpr (TRAIN) {      // no GEMM consumes the last trunk layer here: save it in one go
                store_acc_t8<NT>(tb, (uint32_t)a.tl.a_trunk + (pl.n_trunk - 1) * (32 * 32 * NT), hin);
                store_gate_bits<NT>(tb, (uint32_t)a.tl.a_mask + (pl.n_trunk - 1) * 256, hin, lane);
            }
            for (int o = 0; o < pl.output_ch; ++o) {                                             // network.py:67
                float v = head_dot<NT>(hin, wb, (uint32_t)pl.out_vec + o * NT * 32) +
                          wload1(wb, ((uint32_t)pl.out_b + o) * 4);
                if (h == 0 && p < a.n_points) a.raw[p * pl.output_ch + o] = v;
            }
        }
    }
}

template <int NT, bool EMB>
static int launch_query(const QueryArgs &a, hipStream_t st) {
    int64_t ntiles = (a.n_points + 31) / 32;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > 256) blocks = 256;  // 512 registers per lane -> one 4-wave workgroup per CU
    if (a.plan.straight) {
        if constexpr (!EMB) {
            if (a.acts) {
                hipLaunchKernelGGL((query_fwd_kernel<NT, false, true, true, true>), dim3(blocks), dim3(256), 0, st, a);
                return check_launch("nrf_dnerf_straight_fwd_train");
            }
        }
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, false, true, true>), dim3(blocks), dim3(256), 0, st, a);
        return check_launch("nrf_dnerf_straight_query_fwd");
    }
    if (a.plan.layernorm) {   // the time net alone only has a training entry point (inference: one piece at a time through it)
        if constexpr (!EMB) {
            if (a.times) {
                hipLaunchKernelGGL((query_fwd_kernel<NT, false, true, true, false, true>), dim3(blocks), dim3(256), 0, st, a);
                return check_launch("nrf_dnerf_time_fwd_train (LayerNorm)");
            }
        }
        if (a.acts) {
            hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, true, false, false, true>), dim3(blocks), dim3(256), 0, st, a);
            return check_launch("nrf_query_fwd_train (LayerNorm)");
        }
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, false, false, false, true>), dim3(blocks), dim3(256), 0, st, a);
        return check_launch("nrf_query_fwd (LayerNorm)");
    }
    if constexpr (!EMB) {
        if (a.times) {
            hipLaunchKernelGGL((query_fwd_kernel<NT, false, true, true>), dim3(blocks), dim3(256), 0, st, a);
            return check_launch("nrf_dnerf_time_fwd_train");
        }
    }
    if (a.acts)
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, true>), dim3(blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, false>), dim3(blocks), dim3(256), 0, st, a);
    return check_launch("nrf_query_fwd");
}

// Widths other than 64 / 128 / 256: the static NeRF's kernels (inference and training forward); the D-NeRF time /
// straightforward variants and the LayerNorm instantiations are built for the three shipped widths only.
template <int NT, bool EMB>
static int launch_query_plain(const QueryArgs &a, hipStream_t st) {
    if (a.plan.straight || a.plan.layernorm || a.times)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_fwd: %d channel tiles: the D-NeRF time / straightforward and LayerNorm "
                         "kernels are built for 2, 4 and 8 tiles (W = 64, 128, 256 and the widths padded to them)", NT);
    int64_t ntiles = (a.n_points + 31) / 32;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > 256) blocks = 256;
    if (a.acts)
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, true>), dim3(blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((query_fwd_kernel<NT, EMB, false>), dim3(blocks), dim3(256), 0, st, a);
    return check_launch("nrf_query_fwd");
}

}  // namespace nrf

using namespace nrf;

extern "C" int64_t nrf_mlp_packed_floats(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net || !emb) return -1;
    Planner P{};
    const char *e = plan_net(P, net, emb, 0);
    if (e) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_packed_floats: %s", e); return -1; }
    return P.plan.total_floats;
}

extern "C" int nrf_mlp_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, float *packed, void *stream) {
    NRF_REQUIRE(net && emb && packed, "nrf_mlp_pack_weights: null pointer");
    Planner P{};
    const char *e = plan_net(P, net, emb, 0);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_mlp_pack_weights: %s", e);
    return launch_pack(P, packed, as_stream(stream), "nrf_mlp_pack_weights");
}

static int query_fwd_impl(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                          const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, float *acts,
                          void *stream, const float *times = nullptr, int64_t t_div = 1, bool straight = false) {
    NRF_REQUIRE(net && emb, "nrf_query_fwd: null descriptor");
    if (n_points != 0) NRF_REQUIRE(net && emb && packed && pts && raw, "nrf_query_fwd: null pointer");
    NRF_REQUIRE(n_points >= 0, "nrf_query_fwd: negative n_points");
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0,
                "nrf_query_fwd: packed and raw must be 16-byte aligned");
    Planner P{};
    const char *e = straight ? plan_net(P, net, emb, net->reserved, true)
                             : plan_net(P, net, emb, times ? net->input_ch_views : 0);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_fwd: %s", e);
    if (n_points == 0) return NRF_OK;
    QueryArgs a{};
    a.times = times; a.t_div = t_div; a.t_stride = straight ? net->reserved : 1;
    a.plan = P.plan;
    a.packed = packed; a.pts = pts; a.viewdirs = viewdirs; a.raw = raw; a.n_points = n_points;
    a.input_ch = net->input_ch;
    a.x_stride = net->input_ch + net->input_ch_views;
    if (emb->mode == 0 && P.plan.use_viewdirs && n_points != 0) {
        NRF_REQUIRE(viewdirs, "nrf_query_fwd: viewdirs is null but the net uses view directions");
        NRF_REQUIRE(pts_per_ray >= 1 && n_points % pts_per_ray == 0,
                    "nrf_query_fwd: n_points (%lld) must be a multiple of pts_per_ray (%d)", (long long)n_points,
                    pts_per_ray);
    }
    a.pts_per_ray = pts_per_ray > 0 ? pts_per_ray : 1;
    a.acts = acts;
    a.tl = make_train_layout(P.plan.nt, P.plan.n_trunk, P.plan.use_viewdirs, P.plan.layernorm);
    hipStream_t st = as_stream(stream);
    const bool emb1 = emb->mode != 0;
    switch (P.plan.nt) {
        case 8: return emb1 ? launch_query<8, true>(a, st) : launch_query<8, false>(a, st);
        case 4: return emb1 ? launch_query<4, true>(a, st) : launch_query<4, false>(a, st);
        case 2: return emb1 ? launch_query<2, true>(a, st) : launch_query<2, false>(a, st);
        // every other --netwidth that is a multiple of 32 (config.py:20-27 accepts any): the plain NeRF kernels
        case 1: return emb1 ? launch_query_plain<1, true>(a, st) : launch_query_plain<1, false>(a, st);
        case 3: return emb1 ? launch_query_plain<3, true>(a, st) : launch_query_plain<3, false>(a, st);
        case 5: return emb1 ? launch_query_plain<5, true>(a, st) : launch_query_plain<5, false>(a, st);
        case 6: return emb1 ? launch_query_plain<6, true>(a, st) : launch_query_plain<6, false>(a, st);
        case 7: return emb1 ? launch_query_plain<7, true>(a, st) : launch_query_plain<7, false>(a, st);
        default:
            return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_query_fwd: W = %d has no fused kernel (widths up to 256 do)", net->W);
    }
}

extern "C" int nrf_query_fwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                             const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw,
                             void *stream) {
    return query_fwd_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, nullptr, stream);
}

extern "C" int64_t nrf_train_acts_floats(const nrf_net_t *net, int64_t n_points) {
    if (!net || net->W < 2 || net->W > 256 || n_points < 0) return -1;
    TrainLayout tl = make_train_layout((net->W + 31) / 32, net->D, net->use_viewdirs, net->use_layernorm);
    return ((n_points + 31) / 32) * tl.a_tile;
}

extern "C" int nrf_query_fwd_train(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed,
                                   const float *pts, const float *viewdirs, int64_t n_points, int32_t pts_per_ray,
                                   float *raw, float *acts, void *stream) {
    NRF_REQUIRE(acts, "nrf_query_fwd_train: acts workspace is null");
    return query_fwd_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, acts, stream);
}

/* D-NeRF training: forward of the deformation ("time") net alone, saving activations -- dx = _time_out(trunk(
 * cat[gamma(x), gamma(t)])), vanilla_dynamic/network.py:281-305.  `packed_time` from nrf_dnerf_pack_time_weights. */
extern "C" int nrf_dnerf_time_fwd_train(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time,
                                        const float *pts, const float *times, int64_t n_points, int32_t pts_per_ray,
                                        int32_t times_per_point, float *dx, float *acts, void *stream) {
    NRF_REQUIRE(time_net && emb, "nrf_dnerf_time_fwd_train: null descriptor");
    if (n_points != 0) NRF_REQUIRE(time_net && emb && times && acts && dx, "nrf_dnerf_time_fwd_train: null pointer");
    NRF_REQUIRE(emb->mode == 0 && !time_net->use_viewdirs, "nrf_dnerf_time_fwd_train: needs embed mode 0 and a time-net descriptor");
    NRF_REQUIRE(pts_per_ray >= 1 && n_points % pts_per_ray == 0, "nrf_dnerf_time_fwd_train: n_points must be a multiple of pts_per_ray");
    return query_fwd_impl(time_net, emb, packed_time, pts, nullptr, n_points, pts_per_ray, dx, acts, stream, times,
                          times_per_point ? 1 : pts_per_ray);
}

/* DNeRF_Straightforward (vanilla_dynamic/network.py:100-196): a NeRF whose trunk input and skip are cat[gamma(x), gamma(t)].
 * net->reserved carries the time channel count (input_ch_time).  Inference. */
extern "C" int64_t nrf_dnerf_straight_packed_floats(const nrf_net_t *net, const nrf_embed_t *emb) {
    if (!net || !emb) return -1;
    Planner P{};
    const char *e = plan_net(P, net, emb, net->reserved, true);
    if (e) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_straight_packed_floats: %s", e); return -1; }
    return P.plan.total_floats;
}

extern "C" int nrf_dnerf_straight_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, float *packed, void *stream) {
    NRF_REQUIRE(net && emb && packed, "nrf_dnerf_straight_pack_weights: null pointer");
    Planner P{};
    const char *e = plan_net(P, net, emb, net->reserved, true);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_straight_pack_weights: %s", e);
    return launch_pack(P, packed, as_stream(stream), "nrf_dnerf_straight_pack_weights");
}

extern "C" int nrf_dnerf_straight_query_fwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                                            const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                            int32_t times_per_point, float *raw, void *stream) {
    NRF_REQUIRE(net && emb, "nrf_dnerf_straight_query_fwd: null descriptor");
    if (n_points != 0) NRF_REQUIRE(times, "nrf_dnerf_straight_query_fwd: times is null");
    NRF_REQUIRE(net->use_viewdirs, "nrf_dnerf_straight_query_fwd: the fused kernel needs use_viewdirs");
    return query_fwd_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, nullptr, stream, times,
                          emb->mode != 0 || times_per_point ? 1 : pts_per_ray, true);
}

/* The same forward on raw points (embed mode 0), saving inputs and activations for nrf_dnerf_straight_bwd
 * (workspace size: nrf_train_acts_floats). */
extern "C" int nrf_dnerf_straight_fwd_train(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                                            const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                            int32_t times_per_point, float *raw, float *acts, void *stream) {
    NRF_REQUIRE(net && emb, "nrf_dnerf_straight_fwd_train: null descriptor");
    if (n_points != 0) NRF_REQUIRE(times && acts, "nrf_dnerf_straight_fwd_train: times / acts is null");
    NRF_REQUIRE(net->use_viewdirs && emb->mode == 0,
                "nrf_dnerf_straight_fwd_train: needs use_viewdirs and raw points (embed mode 0)");
    return query_fwd_impl(net, emb, packed, pts, viewdirs, n_points, pts_per_ray, raw, acts, stream, times,
                          times_per_point ? 1 : pts_per_ray, true);
}
