// K6: D-NeRF deformation chain, fused, f32 MFMA, gfx950.
//
// Replaces run_network_dnerf (vanilla_dynamic/render.py:49-124) + DNeRF_Deformation.forward
// (vanilla_dynamic/network.py:307-341): per point
//     dx   = _time_out(time-net(gamma(x), gamma(t)))          network.py:281-305
//     raw  = _occ(gamma(x + dx), gamma(d))                    network.py:331-339
// One wave carries its 32 points through BOTH networks without leaving registers: the time net's three
// outputs come out of a lane-local head, the deformed point is re-embedded in-kernel and fed to the
// canonical net.  Structure, weight streaming and register layout are those of mlp_fused.hip.
//
// Reference quirk kept (network.py:315-322): the reference decides per *netchunk* whether to skip the
// time net -- if the FIRST point of the chunk has t == 0 (and zero_canonical) the whole chunk gets dx = 0.
#include "mlp_device.h"

namespace nrf {

struct DnerfArgs {
    NetPlan tplan, cplan;         // time net (head = _time_out), canonical net
    const float *packed_t, *packed_c;
    const float *pts;             // [P, 3]
    const float *viewdirs;        // [P / pts_per_ray, 3]
    const float *times;           // [P / t_div]
    float *raw;                   // [P, 4]
    float *dx;                    // [P, 3]
    int64_t n_points, netchunk;
    int32_t pts_per_ray, t_div, zero_canonical, pad_;
};

template <int NT>
__device__ __forceinline__ void trunk_fwd(const NetPlan &pl, const WBuf &wb, const float (&xin)[kKsX],
                                          const float (&tin)[kKsT], bool has_t, f32x16 (&hin)[NT], f32x16 (&acc)[NT]) {
    const uint32_t hid_off = NT * 32, skip_off = NT * 32 + 4 * NT * NT * 256;
    load_bias<NT>(acc, wb, (uint32_t)pl.l0_b);
    gemm_in<kKsX, NT>(acc, xin, wb, (uint32_t)pl.l0_w);
    if (has_t) gemm_in<kKsT, NT>(acc, tin, wb, (uint32_t)pl.l0_w2);
    relu_move<NT>(hin, acc, true);
    for (int l = 1; l < pl.n_trunk; ++l) {
        const uint32_t slot = (uint32_t)(pl.trunk_base + (int64_t)(l - 1) * pl.trunk_stride);
        load_bias<NT>(acc, wb, slot);
        if ((pl.skip_in_mask >> l) & 1) gemm_in<kKsX, NT>(acc, xin, wb, slot + skip_off);
        gemm_hid<NT, NT>(acc, hin, wb, slot + hid_off);
        relu_move<NT>(hin, acc, true);
    }
}

template <int NT>
__global__ void __launch_bounds__(256, 1) dnerf_query_fwd_kernel(const DnerfArgs a) {
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    WBuf wt, wc;
    wt.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed_t), 0, (int)(a.tplan.total_floats * 4), 0x00020000);
    wc.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed_c), 0, (int)(a.cplan.total_floats * 4), 0x00020000);
    wt.lane16 = wc.lane16 = lane * 16;
    wt.h16 = wc.h16 = h * 16;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (a.n_points + 31) >> 5;

    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t p = tile * 32 + n;
        const int64_t pc = p < a.n_points ? p : a.n_points - 1;
        const float *q = a.pts + pc * 3;
        float px = q[0], py = q[1], pz = q[2];
        const float t = a.times[pc / a.t_div];
        // network.py:315-322: the chunk's first point decides for the whole chunk
        const int64_t chunk_first = (pc / a.netchunk) * a.netchunk;
        const bool skip = a.zero_canonical && a.times[chunk_first / a.t_div] == 0.0f;

        float xin[kKsX], tin[kKsT];
        f32x16 hin[NT], acc[NT];
        float dx0 = 0.0f, dx1 = 0.0f, dx2 = 0.0f;
        if (!__all(skip)) {
            fill_pe3<kKsX>(xin, px, py, pz, a.tplan.x_param, h);
            fill_pe1<kKsT>(tin, t, a.tplan.t_param, h);
            trunk_fwd<NT>(a.tplan, wt, xin, tin, true, hin, acc);
            const uint32_t ov = (uint32_t)a.tplan.out_vec, ob = (uint32_t)a.tplan.out_b * 4;   // network.py:305
            dx0 = head_dot<NT>(hin, wt, ov + 0 * NT * 32) + wload1(wt, ob + 0);
            dx1 = head_dot<NT>(hin, wt, ov + 1 * NT * 32) + wload1(wt, ob + 4);
            dx2 = head_dot<NT>(hin, wt, ov + 2 * NT * 32) + wload1(wt, ob + 8);
            if (skip) { dx0 = 0.0f; dx1 = 0.0f; dx2 = 0.0f; }
        }
        if (h == 0 && p < a.n_points) { a.dx[p * 3 + 0] = dx0; a.dx[p * 3 + 1] = dx1; a.dx[p * 3 + 2] = dx2; }
        // network.py:331-333: deformed point, re-embedded; with dx == 0 this is the original embedding
        px = px + dx0; py = py + dx1; pz = pz + dx2;
        fill_pe3<kKsX>(xin, px, py, pz, a.cplan.x_param, h);
        trunk_fwd<NT>(a.cplan, wc, xin, tin, false, hin, acc);
        if constexpr (NT >= 2) {
            const NetPlan &pl = a.cplan;
            const float alpha = head_dot<NT>(hin, wc, (uint32_t)pl.alpha_vec) + wload1(wc, (uint32_t)pl.alpha_b * 4);
            load_bias<NT>(acc, wc, (uint32_t)pl.feat_b);
            gemm_hid<NT, NT>(acc, hin, wc, (uint32_t)pl.feat_w);
            relu_move<NT>(hin, acc, false);   // feature vector -> operand registers (no ReLU)
            f32x16 acc2[NT / 2];
            load_bias<NT / 2>(acc2, wc, (uint32_t)pl.views_b);
            gemm_hid<NT, NT / 2>(acc2, hin, wc, (uint32_t)pl.views_w_hid);
            float din[kKsD];                  // built here: the trunk has no registers to spare for it
            const float *d = a.viewdirs + (pc / a.pts_per_ray) * 3;
            fill_pe3<kKsD>(din, d[0], d[1], d[2], a.cplan.d_param, h);
            gemm_in<kKsD, NT / 2>(acc2, din, wc, (uint32_t)pl.views_w_in);
            relu_move<NT / 2>(acc2, acc2, true);
            const uint32_t rv = (uint32_t)pl.rgb_vec, rb = (uint32_t)pl.rgb_b * 4;
            float4 o;
            o.x = head_dot<NT / 2>(acc2, wc, rv + 0 * (NT / 2) * 32) + wload1(wc, rb + 0);
            o.y = head_dot<NT / 2>(acc2, wc, rv + 1 * (NT / 2) * 32) + wload1(wc, rb + 4);
            o.z = head_dot<NT / 2>(acc2, wc, rv + 2 * (NT / 2) * 32) + wload1(wc, rb + 8);
            o.w = alpha;
            if (h == 0 && p < a.n_points) reinterpret_cast<float4 *>(a.raw)[p] = o;
        }
    }
}

}  // namespace nrf

using namespace nrf;

// time net described with nrf_net_t: pts_w/pts_b = _time layers, out_w/out_b = _time_out (W -> 3),
// input_ch = position channels, input_ch_views = TIME channels (first layer in = input_ch + time channels),
// use_viewdirs = 0.
static const char *plan_time(Planner &P, const nrf_net_t *tnet, const nrf_embed_t *emb) {
    return plan_net(P, tnet, emb, tnet->input_ch_views);
}

extern "C" int64_t nrf_dnerf_time_packed_floats(const nrf_net_t *time_net, const nrf_embed_t *emb) {
    if (!time_net || !emb) return -1;
    Planner P{};
    const char *e = plan_time(P, time_net, emb);
    if (e) { nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_time_packed_floats: %s", e); return -1; }
    return P.plan.total_floats;
}

extern "C" int nrf_dnerf_pack_time_weights(const nrf_net_t *time_net, const nrf_embed_t *emb, float *packed,
                                           void *stream) {
    NRF_REQUIRE(time_net && emb && packed, "nrf_dnerf_pack_time_weights: null pointer");
    Planner P{};
    const char *e = plan_time(P, time_net, emb);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_pack_time_weights: %s", e);
    return launch_pack(P, packed, as_stream(stream), "nrf_dnerf_pack_time_weights");
}

extern "C" int nrf_dnerf_query_fwd(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                   const float *packed_time, const float *packed_occ, const float *pts,
                                   const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                   int32_t times_per_point, int64_t netchunk, int32_t zero_canonical, float *raw,
                                   float *dx, void *stream) {
    NRF_REQUIRE(time_net && occ_net && emb, "nrf_dnerf_query_fwd: null descriptor");
    if (n_points != 0) NRF_REQUIRE(time_net && occ_net && emb && packed_time && packed_occ && pts && viewdirs && times && raw && dx,
                "nrf_dnerf_query_fwd: null pointer");
    NRF_REQUIRE(n_points >= 0 && pts_per_ray >= 1 && netchunk >= 1, "nrf_dnerf_query_fwd: bad sizes");
    NRF_REQUIRE(n_points % pts_per_ray == 0, "nrf_dnerf_query_fwd: n_points must be a multiple of pts_per_ray");
    NRF_REQUIRE(emb->mode == 0, "nrf_dnerf_query_fwd: the deformation chain re-embeds x + dx, so it needs the fused "
                                "encoding (mode 0)");
    Planner Pt{}, Pc{};
    const char *e = plan_time(Pt, time_net, emb);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd (time net): %s", e);
    e = plan_net(Pc, occ_net, emb, 0);
    if (e) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd (canonical net): %s", e);
    if (!Pc.plan.use_viewdirs) return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd: canonical net must use view directions");
    if (Pt.plan.nt != Pc.plan.nt)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd: time and canonical nets must have the same width");
    if (Pt.plan.layernorm || Pc.plan.layernorm)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd: LayerNorm nets run as nrf_dnerf_time_fwd_train + "
                                              "nrf_query_fwd (the single-launch chain has no normalisation stage)");
    if (n_points == 0) return NRF_OK;
    DnerfArgs a{};
    a.tplan = Pt.plan; a.cplan = Pc.plan;
    a.packed_t = packed_time; a.packed_c = packed_occ; a.pts = pts; a.viewdirs = viewdirs; a.times = times;
    a.raw = raw; a.dx = dx; a.n_points = n_points; a.netchunk = netchunk; a.pts_per_ray = pts_per_ray;
    a.t_div = times_per_point ? 1 : pts_per_ray; a.zero_canonical = zero_canonical;
    int64_t ntiles = (n_points + 31) / 32;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > 256) blocks = 256;
    hipStream_t st = as_stream(stream);
    switch (Pc.plan.nt) {
        case 8: hipLaunchKernelGGL(dnerf_query_fwd_kernel<8>, dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(dnerf_query_fwd_kernel<4>, dim3(blocks), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(dnerf_query_fwd_kernel<2>, dim3(blocks), dim3(256), 0, st, a); break;
        default: return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_dnerf_query_fwd: W = %d has no fused kernel", occ_net->W);
    }
    return check_launch("nrf_dnerf_query_fwd");
}
