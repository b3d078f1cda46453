// K4 / K4b: alpha compositing (raw2outputs, render.py:82-131) and its hand-written backward, gfx950.
//
// One wavefront per ray.  Lane l owns samples l, l+64, l+128, ...: every global access of the wave is a
// contiguous row piece (raw: 64 x float4 = 1 KiB, z / weights: 256 B), i.e. fully coalesced.  The
// transmittance T_i = prod_{j<i}(1 - alpha_j + 1e-10) is a wavefront prefix product (6 shuffle steps
// per 64 samples, carried across the ray's 64-sample pieces); the per-ray sums are butterfly
// reductions.  HBM-bound: S*(16+4)+12 bytes read and S*4+24 written per ray.
#include "nrf_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Sample {
    float alpha, om, dist;
};

__device__ __forceinline__ Sample eval_sample(const float *__restrict__ z, int i, int S, float nrm, float sigma_raw,
                                              float noise) {
    Sample s;
    float zi = z[i];
    float d = (i + 1 < S) ? z[i + 1] - zi : 1e10f;  // render.py:102-104
    s.dist = d * nrm;                               // render.py:106
    float sg = fmaxf(sigma_raw + noise, 0.0f);      // relu, render.py:100
    s.alpha = 1.0f - expf(-sg * s.dist);
    s.om = (1.0f - s.alpha) + 1e-10f;               // render.py:121
    return s;
}

__global__ void __launch_bounds__(256) composite_fwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z_vals, const float *__restrict__ rays_d, int d_stride,
    int64_t N, int S, const float *__restrict__ noise, int white_bkgd, float *__restrict__ rgb_map,
    float *__restrict__ disp_map, float *__restrict__ acc_map, float *__restrict__ weights,
    float *__restrict__ depth_map) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < N; r += nwaves) {
        const float *d = rays_d + r * d_stride;
        const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float *z = z_vals + r * S;
        float carry = 1.0f;  // transmittance entering this 64-sample piece
        float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
        for (int base = 0; base < S; base += 64) {
            const int i = base + lane;
            const bool ok = i < S;
            float4 p = ok ? raw[r * S + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            Sample s = {0.f, 1.f, 0.f};
            if (ok) s = eval_sample(z, i, S, nrm, p.w, noise ? noise[r * S + i] : 0.0f);
            // inclusive prefix product of om over the wave
            float inc = s.om;
            for (int o = 1; o < 64; o <<= 1) {
                float q = __shfl_up(inc, o);
                if (lane >= o) inc *= q;
            }
            float excl = __shfl_up(inc, 1);
            if (lane == 0) excl = 1.0f;
            const float T = carry * excl;
            carry *= __shfl(inc, 63);
            const float w = s.alpha * T;
            if (ok) {
                weights[r * S + i] = w;
                sr += w * sigmoidf_(p.x);  // render.py:108, :122
                sg += w * sigmoidf_(p.y);
                sb += w * sigmoidf_(p.z);
                sd += w * z[i];            // render.py:124
                sa += w;                   // render.py:126
            }
        }
        sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
        if (lane == 0) {
            float q = sd / sa;                                    // render.py:125; torch.max propagates NaN
            float m = (q > 1e-10f || q != q) ? q : 1e-10f;
            disp_map[r] = 1.0f / m;
            acc_map[r] = sa;
            depth_map[r] = sd;
            if (white_bkgd) { float bg = 1.0f - sa; sr += bg; sg += bg; sb += bg; }  // render.py:128-129
            rgb_map[r * 3 + 0] = sr; rgb_map[r * 3 + 1] = sg; rgb_map[r * 3 + 2] = sb;
        }
    }
}

// Backward.  With q_i = dL/dw_i (all paths: rgb, acc, depth, weights, white background) and
// w_i = alpha_i T_i:   dL/dalpha_i = T_i q_i - (sum_{j>i} w_j q_j) / om_i ,
// dalpha_i/draw3_i = dist_i (1 - alpha_i) [raw3+noise > 0],  dL/draw_k = w_i g_rgb_k c_k (1 - c_k).
// The suffix sum is a wavefront reverse scan, pieces walked from the far end of the ray; forward
// quantities are recomputed from raw/z (cheaper than storing them: the kernel is HBM-bound).
constexpr int kMaxPieces = 16;  // S <= 1024

__global__ void __launch_bounds__(256) composite_bwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z_vals, const float *__restrict__ rays_d, int d_stride,
    int64_t N, int S, const float *__restrict__ noise, int white_bkgd, const float *__restrict__ g_rgb,
    const float *__restrict__ g_acc, const float *__restrict__ g_depth, const float *__restrict__ g_weights,
    float4 *__restrict__ g_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int npieces = (S + 63) >> 6;
    for (int64_t r = wave0; r < N; r += nwaves) {
        const float *d = rays_d + r * d_stride;
        const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float *z = z_vals + r * S;
        const float gr = g_rgb[r * 3 + 0], gg = g_rgb[r * 3 + 1], gb = g_rgb[r * 3 + 2];
        const float ga = (g_acc ? g_acc[r] : 0.0f) - (white_bkgd ? (gr + gg + gb) : 0.0f);
        const float gd = g_depth ? g_depth[r] : 0.0f;
        // pass 1: transmittance entering each piece
        float carry_in[kMaxPieces];
        {
            float carry = 1.0f;
#pragma unroll
            for (int pc = 0; pc < kMaxPieces; ++pc) {
                carry_in[pc] = carry;
                if (pc < npieces) {
                    const int i = pc * 64 + lane;
                    float om = 1.0f;
                    if (i < S) om = eval_sample(z, i, S, nrm, raw[r * S + i].w, noise ? noise[r * S + i] : 0.0f).om;
                    float prod = om;
                    for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o);
                    carry *= prod;
                }
            }
        }
        // pass 2: far -> near
        float suffix_carry = 0.0f;  // sum of w_j q_j over all later pieces
#pragma unroll
        for (int pc = kMaxPieces - 1; pc >= 0; --pc) {
            if (pc >= npieces) continue;
            const int i = pc * 64 + lane;
            const bool ok = i < S;
            float4 p = ok ? raw[r * S + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float nz = (ok && noise) ? noise[r * S + i] : 0.0f;
            Sample s = {0.f, 1.f, 0.f};
            if (ok) s = eval_sample(z, i, S, nrm, p.w, nz);
            float inc = s.om;
            for (int o = 1; o < 64; o <<= 1) {
                float q = __shfl_up(inc, o);
                if (lane >= o) inc *= q;
            }
            float excl = __shfl_up(inc, 1);
            if (lane == 0) excl = 1.0f;
            const float T = carry_in[pc] * excl;
            const float w = s.alpha * T;
            const float cr = sigmoidf_(p.x), cg = sigmoidf_(p.y), cb = sigmoidf_(p.z);
            float q = gr * cr + gg * cg + gb * cb + ga + (ok ? gd * z[i] : 0.0f) +
                      ((ok && g_weights) ? g_weights[r * S + i] : 0.0f);
            float wq = ok ? w * q : 0.0f;
            // exclusive suffix sum over the wave: sum_{lane' > lane} wq
            float suf = wq;
            for (int o = 1; o < 64; o <<= 1) {
                float t = __shfl_down(suf, o);
                if (lane + o < 64) suf += t;
            }
            const float total = __shfl(suf, 0);
            const float after = (suf - wq) + suffix_carry;
            suffix_carry += total;
            if (ok) {
                const float dLda = T * q - after / s.om;
                const float dads = (p.w + nz > 0.0f) ? s.dist * (1.0f - s.alpha) : 0.0f;
                float4 g;
                g.x = w * gr * cr * (1.0f - cr);
                g.y = w * gg * cg * (1.0f - cg);
                g.z = w * gb * cb * (1.0f - cb);
                g.w = dLda * dads;
                g_raw[r * S + i] = g;
            }
        }
    }
}

}  // namespace

extern "C" int nrf_composite_fwd(const float *raw, const float *z_vals, const float *rays_d, int32_t d_stride,
                                 int64_t n_rays, int32_t n_samples, const float *noise, int32_t white_bkgd,
                                 float *rgb_map, float *disp_map, float *acc_map, float *weights, float *depth_map,
                                 void *stream) {
    if (n_rays != 0) NRF_REQUIRE(raw && z_vals && rays_d && rgb_map && disp_map && acc_map && weights && depth_map,
                "nrf_composite_fwd: null pointer");
    NRF_REQUIRE(n_rays >= 0 && n_samples >= 1 && d_stride >= 3, "nrf_composite_fwd: bad sizes");
    NRF_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nrf_composite_fwd: raw must be 16-byte aligned");
    if (n_rays == 0) return NRF_OK;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(nrf::grid_for(n_rays, 4)), dim3(256), 0, nrf::as_stream(stream),
                       reinterpret_cast<const float4 *>(raw), z_vals, rays_d, d_stride, n_rays, n_samples, noise,
                       white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map);
    return nrf::check_launch("nrf_composite_fwd");
}

extern "C" int nrf_composite_bwd(const float *raw, const float *z_vals, const float *rays_d, int32_t d_stride,
                                 int64_t n_rays, int32_t n_samples, const float *noise, int32_t white_bkgd,
                                 const float *g_rgb, const float *g_acc, const float *g_depth,
                                 const float *g_weights, float *g_raw, void *stream) {
    if (n_rays != 0) NRF_REQUIRE(raw && z_vals && rays_d && g_rgb && g_raw, "nrf_composite_bwd: null pointer");
    NRF_REQUIRE(n_rays >= 0 && n_samples >= 1 && d_stride >= 3, "nrf_composite_bwd: bad sizes");
    NRF_REQUIRE(((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(g_raw)) & 15) == 0,
                "nrf_composite_bwd: raw/g_raw must be 16-byte aligned");
    if (n_samples > 64 * kMaxPieces)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_composite_bwd: n_samples <= %d", 64 * kMaxPieces);
    if (n_rays == 0) return NRF_OK;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(nrf::grid_for(n_rays, 4)), dim3(256), 0, nrf::as_stream(stream),
                       reinterpret_cast<const float4 *>(raw), z_vals, rays_d, d_stride, n_rays, n_samples, noise,
                       white_bkgd, g_rgb, g_acc, g_depth, g_weights, reinterpret_cast<float4 *>(g_raw));
    return nrf::check_launch("nrf_composite_bwd");
}
