// Sampling kernels (K1 coarse sampler, K5 hierarchical resampler) for gfx950.
//
// Sample positions must be BIT-EXACT with the reference's CPU path.  HIP's
// __fmul_rn/__fadd_rn are plain * and + to the compiler, so a*b+c would still be contracted into an
// FMA under hipcc's default -ffp-contract=fast: this file is compiled with -ffp-contract=off (Makefile)
// AND carries the pragma below; the only fused operation is the explicit __fmaf_rn of linspace.
//
// K1: one thread per (ray, sample) element; coalesced [N,S] stores.           HBM-bound, tiny.
// K5: one wavefront per ray.  The ray's bins / weights / cdf live in LDS; the 62-term weight sum is
//     evaluated in ATen's fixed order, the cdf is an f64 running sum (exact, see below), each lane
//     inverts the cdf for its samples by binary search, and the merged depth list is produced by
//     rank-merge (or a bitonic sort when the new samples are not monotone).  Loads and stores are
//     whole contiguous rows per wave (256..768 B), i.e. coalesced.
#include "nrf_common.h"

#pragma clang fp contract(off)

namespace {

// torch.linspace(0, 1, n)[i] on CPU: step = rn(1/(n-1)); two-sided, one fused rounding per element.
__device__ __forceinline__ float linspace01(int i, int n, float step) {
    if (n == 1) return 0.0f;
    return (i < n / 2) ? __fmaf_rn(step, (float)i, 0.0f) : __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
}

__device__ __forceinline__ float coarse_z(float near, float far, int i, int S, float step, bool lindisp) {
    float t = linspace01(i, S, step);
    float omt = __fsub_rn(1.0f, t);
    if (!lindisp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));                // render.py:188
    float a = __fmul_rn(__fdiv_rn(1.0f, near), omt), b = __fmul_rn(__fdiv_rn(1.0f, far), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));                                                // render.py:190
}

__global__ void __launch_bounds__(256) sample_coarse_kernel(const float *__restrict__ rays, int64_t total, int ncol,
                                                            int S, int lindisp,
                                                            const float *__restrict__ t_rand,
                                                            float *__restrict__ z_vals) {
    const float step = (S > 1) ? __fdiv_rn(1.0f, (float)(S - 1)) : 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / S;
        int i = (int)(e - r * S);
        float near = rays[r * ncol + 6], far = rays[r * ncol + 7];
        float z = (S > 1) ? coarse_z(near, far, i, S, step, lindisp) : near;
        if (t_rand) {  // stratified jitter, render.py:194-208
            float zl = (i > 0) ? coarse_z(near, far, i - 1, S, step, lindisp) : z;
            float zu = (i + 1 < S) ? coarse_z(near, far, i + 1, S, step, lindisp) : z;
            float upper = (i + 1 < S) ? __fmul_rn(0.5f, __fadd_rn(zu, z)) : z;
            float lower = (i > 0) ? __fmul_rn(0.5f, __fadd_rn(z, zl)) : z;
            z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[e]));
        }
        z_vals[e] = z;
    }
}

// Four consecutive samples of one ray per thread (S % 4 == 0, 16-byte aligned rows): one 16-byte load of depths,
// three 16-byte stores, no per-element 64-bit divisions -- the same two rounded operations per value.
__global__ void __launch_bounds__(256) points4_kernel(const float *__restrict__ rays, int64_t n_quads, int ncol, int S4,
                                                      const float4 *__restrict__ z4, float4 *__restrict__ pts4) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = n_quads < (1ll << 31) ? (int64_t)((uint32_t)q / (uint32_t)S4) : q / S4;
        const float *ray = rays + r * ncol;
        const float o0 = ray[0], o1 = ray[1], o2 = ray[2], d0 = ray[3], d1 = ray[4], d2 = ray[5];
        const float4 z = z4[q];
        const float zz[4] = {z.x, z.y, z.z, z.w};
        float p[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[3 * k + 0] = __fadd_rn(o0, __fmul_rn(d0, zz[k]));      // render.py:210: separate mul and add
            p[3 * k + 1] = __fadd_rn(o1, __fmul_rn(d1, zz[k]));
            p[3 * k + 2] = __fadd_rn(o2, __fmul_rn(d2, zz[k]));
        }
        pts4[3 * q + 0] = make_float4(p[0], p[1], p[2], p[3]);
        pts4[3 * q + 1] = make_float4(p[4], p[5], p[6], p[7]);
        pts4[3 * q + 2] = make_float4(p[8], p[9], p[10], p[11]);
    }
}

__global__ void __launch_bounds__(256) points_kernel(const float *__restrict__ rays, int64_t total, int ncol, int S,
                                                     const float *__restrict__ z_vals, float *__restrict__ pts) {
    // one thread per output float: pts[(r*S+i)*3+c] = o[c] + d[c]*z  (render.py:210: separate mul and add)
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = e / 3;
        int c = (int)(e - s * 3);
        int64_t r = s / S;
        pts[e] = __fadd_rn(rays[r * ncol + c], __fmul_rn(rays[r * ncol + 3 + c], z_vals[s]));
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int kMaxBins = 256;     // S-1 <= 256
constexpr int kMaxMerged = 1024;  // S + n_importance <= 1024

// Per-wave LDS working set, sized by the instantiation: the common configurations (S <= 64 coarse samples,
// S + n_importance <= 256: 64+128, 64+64) need 2.5 KB per wave instead of the general 11 KB, which lifts the
// occupancy of this latency-bound kernel from 3 to 4 waves per SIMD (then register-limited).
template <int MAXB, int MAXM>
struct FineSmem {
    float bins[MAXB];
    float cdf[MAXB];
    float w[MAXB];
    float zs[MAXM];      // new samples
    float merged[MAXM];  // sort buffer / staging (a power of two >= S + n_importance)
};

__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per ray.  mode_fine: bins/weights are derived from (z_vals, full weights) as render.py:219-220
// does; otherwise `zin` IS bins [nb] and `win` IS weights [nb-1] (plain sample_pdf).
// The waves of a block never share data: a wave's LDS accesses execute in order, so lanes only need the compiler not
// to move LDS accesses across the hand-over points.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int MAXB, int MAXM>
__global__ void __launch_bounds__(256) sample_fine_kernel(const float *__restrict__ zin,
                                                          const float *__restrict__ win, int64_t N, int S, int ns,
                                                          int mode_fine, const float *__restrict__ u_in,
                                                          float *__restrict__ z_samples,
                                                          float *__restrict__ z_merged, float *__restrict__ z_std) {
    __shared__ FineSmem<MAXB, MAXM> sm[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    FineSmem<MAXB, MAXM> &m = sm[wave];
    const int nb = mode_fine ? S - 1 : S;  // number of bins
    const int nw = nb - 1;                 // number of pdf weights
    const int zrow = mode_fine ? S : nb;   // row length of zin
    const int wrow = mode_fine ? S : nw;   // row length of win
    const float ustep = (ns > 1) ? __fdiv_rn(1.0f, (float)(ns - 1)) : 0.0f;
    const int total = S + ns;

    for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < N; r0 += (int64_t)gridDim.x * 4) {
        const int64_t r = r0 + wave;
        const bool live = r < N;  // wave-uniform
        if (live) {
            const float *z = zin + r * zrow;
            const float *wt = win + r * wrow;
            for (int i = lane; i < nb; i += 64)
                m.bins[i] = mode_fine ? __fmul_rn(0.5f, __fadd_rn(z[i + 1], z[i])) : z[i];   // render.py:219
            for (int i = lane; i < nw; i += 64)
                m.w[i] = __fadd_rn(mode_fine ? wt[i + 1] : wt[i], 1e-5f);                    // sampling.py:19
            if (mode_fine && z_merged)   // the merge's binary searches run over LDS, not over global memory
                for (int i = lane; i < S; i += 64) m.merged[i] = z[i];
        }
        wsync();
        float s = 0.0f;
        if (live) {
            // sampling.py:20 torch.sum(weights, -1): ATen vectorized_inner_sum order (8 lanes x ILP 4).
            // Lanes 0..7 each own one vector lane k; the final fold over k is done by every lane.
            const int nv = nw >> 3, rounds = nv >> 2;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const int k = lane & 7;
            for (int i = 0; i < rounds; ++i) {
                a0 = __fadd_rn(a0, m.w[(i * 4 + 0) * 8 + k]);
                a1 = __fadd_rn(a1, m.w[(i * 4 + 1) * 8 + k]);
                a2 = __fadd_rn(a2, m.w[(i * 4 + 2) * 8 + k]);
                a3 = __fadd_rn(a3, m.w[(i * 4 + 3) * 8 + k]);
            }
            for (int v = rounds * 4; v < nv; ++v) a0 = __fadd_rn(a0, m.w[v * 8 + k]);
            a0 = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
            for (int i = nv * 8; i < nw; ++i) s = __fadd_rn(s, m.w[i]);
            for (int kk = 0; kk < 8; ++kk) s = __fadd_rn(s, __shfl(a0, kk));
            // sampling.py:20-22 pdf = w / s; cdf = [0, cumsum(pdf)].  ATen's CPU cumsum keeps an f64 running
            // sum and rounds each prefix to f32.  Each pdf is an f32 in [~1e-5/s, 1], so every f64 partial
            // sum is exact (<= 47 significant bits) and a parallel prefix gives the sequential result.
            double carry = 0.0;
            for (int base = 0; base < nw; base += 64) {
                int i = base + lane;
                double p = (i < nw) ? (double)__fdiv_rn(m.w[i], s) : 0.0;
                for (int o = 1; o < 64; o <<= 1) {
                    double q = __shfl_up(p, o);
                    if (lane >= o) p += q;
                }
                p += carry;
                if (i < nw) m.cdf[i + 1] = (float)p;
                carry = __shfl(p, 63);
            }
            if (lane == 0) m.cdf[0] = 0.0f;
        }
        wsync();
        bool sorted_ok = true;
        double sum1 = 0.0;
        if (live) {
            float prev_last = -INFINITY;
            for (int base = 0; base < ns; base += 64) {
                int j = base + lane;
                float out = INFINITY;
                if (j < ns) {
                    float u = u_in ? u_in[r * ns + j] : linspace01(j, ns, ustep);            // sampling.py:25-29
                    int lo = 0, hi = nb;  // searchsorted(cdf, u, right=True): first idx with cdf[idx] > u
                    while (lo < hi) {
                        int mid = (lo + hi) >> 1;
                        if (m.cdf[mid] > u) hi = mid; else lo = mid + 1;
                    }
                    int below = max(lo - 1, 0), above = min(lo, nb - 1);                     // sampling.py:45-46
                    float c0 = m.cdf[below], c1 = m.cdf[above], b0 = m.bins[below], b1 = m.bins[above];
                    float denom = __fsub_rn(c1, c0);
                    if (denom < 1e-5f) denom = 1.0f;                                         // sampling.py:56
                    float t = __fdiv_rn(__fsub_rn(u, c0), denom);
                    out = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));                    // sampling.py:58
                    m.zs[j] = out;
                    if (z_samples) z_samples[r * ns + j] = out;
                    sum1 += (double)out;
                }
                // monotonicity check (rank-merge below needs a sorted list)
                float left = __shfl_up(out, 1);
                if (lane == 0) left = prev_last;
                if (j < ns && !(out >= left)) sorted_ok = false;   // NaN compares false: non-finite rows take the sort path
                prev_last = __shfl(out, 63);
            }
            sorted_ok = __all(sorted_ok);
        }
        wsync();
        if (live && z_std) {  // render.py:238 torch.std(z_samples, unbiased=False); tolerance-level
            double mean = wave_sum_f64(sum1) / (double)ns;
            double v = 0.0;
            for (int j = lane; j < ns; j += 64) {
                double d = (double)m.zs[j] - mean;
                v += d * d;
            }
            v = wave_sum_f64(v);
            if (lane == 0) z_std[r] = (float)sqrt(v / (double)ns);
        }
        if (mode_fine && z_merged) {  // render.py:223 sort(cat([z_vals, z_samples]))  (values only)
            const float *z = zin + (live ? r : 0) * zrow;
            bool zsorted = true;
            if (live) {
                for (int i = lane; i + 1 < S; i += 64)
                    if (!(m.merged[i + 1] >= m.merged[i])) zsorted = false;   // (so every output slot is written)
                zsorted = __all(zsorted);
            }
            const bool fast = live && sorted_ok && zsorted;  // wave-uniform
            if (fast) {
                // rank-merge of two sorted lists: rank(z[i]) = i + #{zs < z[i]}, rank(zs[j]) = j + #{z <= zs[j]}
                float *out = z_merged + r * total;
                for (int i = lane; i < S; i += 64) {
                    float v = m.merged[i];
                    int lo = 0, hi = ns;
                    while (lo < hi) { int mid = (lo + hi) >> 1; if (m.zs[mid] < v) lo = mid + 1; else hi = mid; }
                    out[i + lo] = v;
                }
                for (int j = lane; j < ns; j += 64) {
                    float v = m.zs[j];
                    int lo = 0, hi = S;
                    while (lo < hi) { int mid = (lo + hi) >> 1; if (m.merged[mid] <= v) lo = mid + 1; else hi = mid; }
                    out[j + lo] = v;
                }
            }
            // slow path: bitonic sort of the wave's own array
            if (live && !fast) {   // wave-uniform: every wave works on its own LDS arrays, no block-level coupling
                int T2 = 1;
                while (T2 < total) T2 <<= 1;
                for (int i = lane; i < T2; i += 64)
                    m.merged[i] = (i < S) ? z[i] : (i < total ? m.zs[i - S] : INFINITY);
                wsync();
                for (int k = 2; k <= T2; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < T2; i += 64) {
                            int p = i ^ j;
                            if (p > i) {
                                float a = m.merged[i], b = m.merged[p];
                                bool up = (i & k) == 0;
                                if ((a > b) == up) { m.merged[i] = b; m.merged[p] = a; }
                            }
                        }
                        wsync();
                    }
                for (int i = lane; i < total; i += 64) z_merged[r * total + i] = m.merged[i];
            }
        }
        wsync();
    }
}

}  // namespace

// K0: the [N, 8|11] ray matrix of render() straight from (pixel range, intrinsics, pose) --
// get_rays rays.py:5-27, viewdirs render.py:283-290, ndc_rays rays.py:52-83, bounds render.py:298-304 -- one
// thread per ray, the reference's f32 operation order (this file is built with -ffp-contract=off).
struct GenRaysArgs {
    const float *c2w, *c2w_static;   // device, rows of `stride` floats ([:3, :4] views); static may be null
    int32_t stride, stride_static;
    int32_t H, W, row0, nrows;
    float fx, fy, cx, cy;
    float near, far;
    int32_t use_viewdirs, ndc;
    float ndc_focal, ndc_near;
    float *rays;                     // [nrows * W, 8 | 11]
};

__device__ __forceinline__ void pose_ray(const float *m, int stride, float x, float y, float (&o)[3], float (&d)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *r = m + k * stride;
        // torch.sum(dirs[..., None, :] * c2w[:3, :3], -1): rounded products, summed left to right; dirs_z = -1
        d[k] = __fadd_rn(__fadd_rn(__fmul_rn(x, r[0]), __fmul_rn(y, r[1])), __fmul_rn(-1.0f, r[2]));
        o[k] = r[3];
    }
}

__global__ void __launch_bounds__(256) gen_rays_kernel(const GenRaysArgs a) {
    const int64_t total = (int64_t)a.nrows * a.W;
    const int ncol = a.use_viewdirs ? 11 : 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int row = a.row0 + (int)(e / a.W), col = (int)(e % a.W);
        const float x = __fdiv_rn(__fsub_rn((float)col, a.cx), a.fx);
        const float y = __fdiv_rn(-__fsub_rn((float)row, a.cy), a.fy);
        float o[3], d[3], v[3] = {0.f, 0.f, 0.f};
        pose_ray(a.c2w, a.stride, x, y, o, d);
        if (a.use_viewdirs) {   // view directions come from c2w even when the camera is c2w_staticcam
            const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
            v[0] = __fdiv_rn(d[0], nrm); v[1] = __fdiv_rn(d[1], nrm); v[2] = __fdiv_rn(d[2], nrm);
            if (a.c2w_static) pose_ray(a.c2w_static, a.stride_static, x, y, o, d);
        }
        if (a.ndc) {            // rays.py:66-78
            const float t = __fdiv_rn(-__fadd_rn(a.ndc_near, o[2]), d[2]);
            o[0] = __fadd_rn(o[0], __fmul_rn(t, d[0]));
            o[1] = __fadd_rn(o[1], __fmul_rn(t, d[1]));
            o[2] = __fadd_rn(o[2], __fmul_rn(t, d[2]));
            const float sx = (float)(-1.0 / ((double)a.W / (2.0 * (double)a.ndc_focal)));
            const float sy = (float)(-1.0 / ((double)a.H / (2.0 * (double)a.ndc_focal)));
            const float tn = (float)(2.0 * (double)a.ndc_near);
            const float o0 = __fdiv_rn(__fmul_rn(sx, o[0]), o[2]);
            const float o1 = __fdiv_rn(__fmul_rn(sy, o[1]), o[2]);
            const float o2 = __fadd_rn(1.0f, __fdiv_rn(tn, o[2]));
            const float d0 = __fmul_rn(sx, __fsub_rn(__fdiv_rn(d[0], d[2]), __fdiv_rn(o[0], o[2])));
            const float d1 = __fmul_rn(sy, __fsub_rn(__fdiv_rn(d[1], d[2]), __fdiv_rn(o[1], o[2])));
            const float d2 = __fdiv_rn(-tn, o[2]);
            o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
        }
        float *r = a.rays + e * ncol;
        r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = d[0]; r[4] = d[1]; r[5] = d[2]; r[6] = a.near; r[7] = a.far;
        if (a.use_viewdirs) { r[8] = v[0]; r[9] = v[1]; r[10] = v[2]; }
    }
}

extern "C" int nrf_gen_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float *c2w, int32_t c2w_stride,
                            const float *c2w_static, int32_t static_stride, int32_t row0, int32_t nrows, float near,
                            float far, int32_t use_viewdirs, int32_t ndc, float ndc_focal, float ndc_near, float *rays,
                            void *stream) {
    NRF_REQUIRE(H >= 1 && W >= 1 && row0 >= 0 && nrows >= 0 && row0 + nrows <= H, "nrf_gen_rays: bad image / row range");
    NRF_REQUIRE(c2w_stride >= 4 && (!c2w_static || static_stride >= 4), "nrf_gen_rays: pose rows need >= 4 floats");
    if (nrows == 0) return NRF_OK;
    NRF_REQUIRE(c2w && rays, "nrf_gen_rays: null pointer");
    GenRaysArgs a{};
    a.c2w = c2w; a.c2w_static = use_viewdirs ? c2w_static : nullptr; a.stride = c2w_stride; a.stride_static = static_stride;
    a.H = H; a.W = W; a.row0 = row0; a.nrows = nrows; a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    a.near = near; a.far = far; a.use_viewdirs = use_viewdirs; a.ndc = ndc; a.ndc_focal = ndc_focal; a.ndc_near = ndc_near;
    a.rays = rays;
    hipLaunchKernelGGL(gen_rays_kernel, dim3(nrf::grid_for((int64_t)nrows * W, 256)), dim3(256), 0, nrf::as_stream(stream), a);
    return nrf::check_launch("nrf_gen_rays");
}

// K0b: rows of train.py's shuffled rays_rgb table (train.py:147-153, batches :183-185) without the
// table: row r of the unshuffled table is (train image r / (H*W), pixel r % (H*W)); its ray comes from get_rays_np
// (rays.py:43-48) and its target from the image.  `f64`: the arithmetic the reference's numpy performs -- a float64
// intrinsics matrix promotes get_rays_np to float64 under NumPy >= 2, rounded once by .astype(float32) (1); NumPy 1.x
// keeps float32 (0, the operation order of get_rays).
struct RayBatchArgs {
    const int64_t *rows;
    const int32_t *img_index;        // table slot -> image / pose index (i_train); null = identity
    const float *poses, *images;     // poses: 3 rows of >= 4 floats per image; images [*, H, W, 3]
    int64_t n, pose_pitch;
    int32_t pose_stride, H, W, n_slots, n_images, f64;
    double fx, fy, cx, cy;
    float *rays, *rgb;               // [2, n, 3], [n, 3]
};

__global__ void __launch_bounds__(256) ray_batch_kernel(const RayBatchArgs a) {
    const int64_t hw = (int64_t)a.H * a.W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = a.rows[e];
        float *ro = a.rays + e * 3, *rd = a.rays + (a.n + e) * 3, *c = a.rgb + e * 3;
        if (r < 0 || r >= hw * a.n_slots) {   // a row outside the table: make it visible instead of reading out of bounds
            const float q = __builtin_nanf("");
            for (int k = 0; k < 3; ++k) { ro[k] = q; rd[k] = q; c[k] = q; }
            continue;
        }
        const int slot = (int)(r / hw);
        const int64_t pix = r - slot * hw;
        const int img = a.img_index ? a.img_index[slot] : slot;
        if (img < 0 || img >= a.n_images) {   // a slot that names no image
            const float q = __builtin_nanf("");
            for (int k = 0; k < 3; ++k) { ro[k] = q; rd[k] = q; c[k] = q; }
            continue;
        }
        const float col = (float)(pix % a.W), row = (float)(pix / a.W);
        const float *m = a.poses + img * a.pose_pitch;
        if (a.f64) {
            const double x = ((double)col - a.cx) / a.fx, y = -((double)row - a.cy) / a.fy;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float *q = m + k * a.pose_stride;
                rd[k] = (float)(__dadd_rn(__dadd_rn(__dmul_rn(x, (double)q[0]), __dmul_rn(y, (double)q[1])),
                                          __dmul_rn(-1.0, (double)q[2])));
                ro[k] = q[3];
            }
        } else {
            float o[3], d[3];
            pose_ray(m, a.pose_stride, __fdiv_rn(__fsub_rn(col, (float)a.cx), (float)a.fx),
                     __fdiv_rn(-__fsub_rn(row, (float)a.cy), (float)a.fy), o, d);
            for (int k = 0; k < 3; ++k) { ro[k] = o[k]; rd[k] = d[k]; }
        }
        const float *px = a.images + ((int64_t)img * hw + pix) * 3;
        c[0] = px[0]; c[1] = px[1]; c[2] = px[2];
    }
}

extern "C" int nrf_gather_ray_batch(const int64_t *rows, int64_t n, const int32_t *img_index, int32_t n_slots,
                                    const float *poses, int32_t pose_stride, int64_t pose_pitch, const float *images,
                                    int32_t n_images, int32_t H, int32_t W, double fx, double fy, double cx, double cy,
                                    int32_t f64, float *rays, float *rgb, void *stream) {
    NRF_REQUIRE(H >= 1 && W >= 1 && n >= 0 && n_slots >= 1 && n_images >= 1, "nrf_gather_ray_batch: bad image / table size");
    NRF_REQUIRE(img_index || n_slots <= n_images, "nrf_gather_ray_batch: more table slots than images");
    NRF_REQUIRE(pose_stride >= 4 && pose_pitch >= 2 * (int64_t)pose_stride + 4, "nrf_gather_ray_batch: pose layout");
    if (n == 0) return NRF_OK;
    NRF_REQUIRE(rows && poses && images && rays && rgb, "nrf_gather_ray_batch: null pointer");
    RayBatchArgs a{};
    a.rows = rows; a.img_index = img_index; a.poses = poses; a.images = images; a.n = n; a.pose_pitch = pose_pitch;
    a.pose_stride = pose_stride; a.H = H; a.W = W; a.n_slots = n_slots; a.n_images = n_images; a.f64 = f64;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.rays = rays; a.rgb = rgb;
    hipLaunchKernelGGL(ray_batch_kernel, dim3(nrf::grid_for(n, 256)), dim3(256), 0, nrf::as_stream(stream), a);
    return nrf::check_launch("nrf_gather_ray_batch");
}

extern "C" int nrf_sample_coarse(const float *rays, int64_t n_rays, int32_t ncol, int32_t n_samples,
                                 int32_t lindisp, const float *t_rand, float *z_vals, void *stream) {
    if (n_rays != 0) NRF_REQUIRE(rays && z_vals, "nrf_sample_coarse: null pointer");
    NRF_REQUIRE(n_rays >= 0 && ncol >= 8 && n_samples >= 1, "nrf_sample_coarse: bad sizes (n_rays=%lld ncol=%d S=%d)",
                (long long)n_rays, ncol, n_samples);
    if (n_rays == 0) return NRF_OK;
    int64_t total = n_rays * n_samples;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3(nrf::grid_for(total, 256)), dim3(256), 0, nrf::as_stream(stream),
                       rays, total, ncol, n_samples, lindisp, t_rand, z_vals);
    return nrf::check_launch("nrf_sample_coarse");
}

extern "C" int nrf_points(const float *rays, int64_t n_rays, int32_t ncol, const float *z_vals, int32_t n_samples,
                          float *pts, void *stream) {
    if (n_rays != 0) NRF_REQUIRE(rays && z_vals && pts, "nrf_points: null pointer");
    NRF_REQUIRE(n_rays >= 0 && ncol >= 6 && n_samples >= 1, "nrf_points: bad sizes");
    if (n_rays == 0) return NRF_OK;
    int64_t total = n_rays * n_samples * 3;
    if (n_samples % 4 == 0 && ((uintptr_t)z_vals & 15) == 0 && ((uintptr_t)pts & 15) == 0) {
        const int64_t n_quads = n_rays * (n_samples / 4);
        hipLaunchKernelGGL(points4_kernel, dim3(nrf::grid_for(n_quads, 256, 256 * 32)), dim3(256), 0, nrf::as_stream(stream),
                           rays, n_quads, ncol, n_samples / 4, reinterpret_cast<const float4 *>(z_vals),
                           reinterpret_cast<float4 *>(pts));
        return nrf::check_launch("nrf_points");
    }
    hipLaunchKernelGGL(points_kernel, dim3(nrf::grid_for(total, 256)), dim3(256), 0, nrf::as_stream(stream), rays,
                       total, ncol, n_samples, z_vals, pts);
    return nrf::check_launch("nrf_points");
}

extern "C" int nrf_sample_fine(const float *z_vals, const float *weights, int64_t n_rays, int32_t n_samples,
                               int32_t n_importance, const float *u, float *z_samples, float *z_merged,
                               float *z_std, void *stream) {
    if (n_rays != 0) NRF_REQUIRE(z_vals && weights && z_merged, "nrf_sample_fine: null pointer");
    NRF_REQUIRE(n_rays >= 0 && n_samples >= 3 && n_importance >= 1,
                "nrf_sample_fine: bad sizes (n_rays=%lld, n_samples=%d: at least 3 coarse samples, i.e. one interior "
                "weight, are needed; n_importance=%d: at least 1)", (long long)n_rays, n_samples, n_importance);
    if (n_samples - 1 > kMaxBins || n_samples + n_importance > kMaxMerged)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_sample_fine: N_samples-1 <= %d and N_samples+N_importance <= %d",
                         kMaxBins, kMaxMerged);
    if (n_rays == 0) return NRF_OK;
    if (n_samples - 1 <= 64 && n_samples + n_importance <= 256)
        hipLaunchKernelGGL((sample_fine_kernel<64, 256>), dim3(nrf::grid_for(n_rays, 4)), dim3(256), 0, nrf::as_stream(stream),
                           z_vals, weights, n_rays, n_samples, n_importance, 1, u, z_samples, z_merged, z_std);
    else
        hipLaunchKernelGGL((sample_fine_kernel<kMaxBins, kMaxMerged>), dim3(nrf::grid_for(n_rays, 4)), dim3(256), 0,
                           nrf::as_stream(stream), z_vals, weights, n_rays, n_samples, n_importance, 1, u, z_samples,
                           z_merged, z_std);
    return nrf::check_launch("nrf_sample_fine");
}

extern "C" int nrf_sample_pdf(const float *bins, const float *weights, int64_t n_rows, int32_t n_bins,
                              int32_t n_samples, const float *u, float *samples, void *stream) {
    if (n_rows != 0) NRF_REQUIRE(bins && weights && samples, "nrf_sample_pdf: null pointer");
    NRF_REQUIRE(n_rows >= 0 && n_bins >= 2 && n_samples >= 1, "nrf_sample_pdf: bad sizes");
    if (n_bins > kMaxBins || n_samples > kMaxMerged)
        return nrf::fail(NRF_ERR_UNSUPPORTED, "nrf_sample_pdf: n_bins <= %d, n_samples <= %d", kMaxBins, kMaxMerged);
    if (n_rows == 0) return NRF_OK;
    if (n_bins <= 64 && n_samples <= 256)
        hipLaunchKernelGGL((sample_fine_kernel<64, 256>), dim3(nrf::grid_for(n_rows, 4)), dim3(256), 0, nrf::as_stream(stream),
                           bins, weights, n_rows, n_bins, n_samples, 0, u, samples, (float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((sample_fine_kernel<kMaxBins, kMaxMerged>), dim3(nrf::grid_for(n_rows, 4)), dim3(256), 0,
                           nrf::as_stream(stream), bins, weights, n_rows, n_bins, n_samples, 0, u, samples,
                           (float *)nullptr, (float *)nullptr);
    return nrf::check_launch("nrf_sample_pdf");
}
