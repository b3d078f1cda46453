// A1: Embedder.embed (positional_encoding.py:5-36) as a standalone kernel, gfx950.
// Used when a caller wants the embedding itself (get_embedder()(x), train_2d.py's Embedder) -- the
// render path never materialises it: there the encoding is fused into the network query kernel.
// One thread per OUTPUT element: stores are fully coalesced rows of out_dim floats; the input row is
// re-read from L1/L2.  sinf/cosf are the accurate (range-reducing) device functions: arguments reach
// 2^(L-1) * |x| ~ 3000 rad, where the fast hardware approximations lose all precision.
#include "nrf_common.h"

namespace {

__device__ __forceinline__ float linspace_at(float start, float end, int n, int i) {
    if (n == 1) return start;
    float step = __fdiv_rn(__fsub_rn(end, start), (float)(n - 1));
    return (i < n / 2) ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(n - 1 - i), end);
}

__global__ void __launch_bounds__(256) pe_fwd_kernel(const float *__restrict__ x, int64_t total, int dims,
                                                     int num_freqs, float max_freq_log2, int log_sampling,
                                                     int include_input, int out_dim, float *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = e / out_dim;
        int c = (int)(e - p * out_dim);
        float v;
        if (include_input && c < dims) {
            v = x[p * dims + c];
        } else {
            int cc = c - (include_input ? dims : 0);
            int f = cc / (2 * dims);
            int rem = cc - f * 2 * dims;
            int k = rem % dims;
            bool is_cos = rem >= dims;
            float freq = log_sampling ? exp2f(linspace_at(0.0f, max_freq_log2, num_freqs, f))       // :23
                                      : linspace_at(1.0f, exp2f(max_freq_log2), num_freqs, f);      // :25
            float arg = __fmul_rn(x[p * dims + k], freq);                                           // :29
            v = is_cos ? cosf(arg) : sinf(arg);
        }
        out[e] = v;
    }
}

// d(loss)/dx of the encoding above: one thread per INPUT element (point, coordinate).
//   d sin(f x)/dx = f cos(f x),  d cos(f x)/dx = -f sin(f x)   (autograd of positional_encoding.py:24-33)
__global__ void __launch_bounds__(256) pe_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                     int64_t total, int dims, int num_freqs, float max_freq_log2,
                                                     int log_sampling, int include_input, int out_dim,
                                                     float *__restrict__ g_x) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / dims;
        const int k = (int)(e - p * dims);
        const float xv = x[e];
        const float *gr = g + p * out_dim;
        float acc = include_input ? gr[k] : 0.0f;
        const int base = include_input ? dims : 0;
        for (int f = 0; f < num_freqs; ++f) {
            float freq = log_sampling ? exp2f(linspace_at(0.0f, max_freq_log2, num_freqs, f))
                                      : linspace_at(1.0f, exp2f(max_freq_log2), num_freqs, f);
            float sn, cs;
            sincosf(__fmul_rn(xv, freq), &sn, &cs);
            acc += freq * (cs * gr[base + 2 * dims * f + k] - sn * gr[base + 2 * dims * f + dims + k]);
        }
        g_x[e] = acc;
    }
}

}  // namespace

extern "C" int nrf_pe_bwd(const float *x, const float *g_out, int64_t n_points, int32_t dims, int32_t num_freqs,
                          float max_freq_log2, int32_t log_sampling, int32_t include_input, float *g_x, void *stream) {
    if (n_points != 0) NRF_REQUIRE(x && g_out && g_x, "nrf_pe_bwd: null pointer");
    NRF_REQUIRE(n_points >= 0 && dims >= 1 && num_freqs >= 0, "nrf_pe_bwd: bad sizes");
    int out_dim = dims * ((include_input ? 1 : 0) + 2 * num_freqs);
    NRF_REQUIRE(out_dim > 0, "nrf_pe_bwd: empty embedding");
    if (n_points == 0) return NRF_OK;
    int64_t total = n_points * dims;
    hipLaunchKernelGGL(pe_bwd_kernel, dim3(nrf::grid_for(total, 256)), dim3(256), 0, nrf::as_stream(stream), x, g_out,
                       total, dims, num_freqs, max_freq_log2, log_sampling, include_input, out_dim, g_x);
    return nrf::check_launch("nrf_pe_bwd");
}

extern "C" int nrf_pe_fwd(const float *x, int64_t n_points, int32_t dims, int32_t num_freqs, float max_freq_log2,
                          int32_t log_sampling, int32_t include_input, float *out, void *stream) {
    if (n_points != 0) NRF_REQUIRE(x && out, "nrf_pe_fwd: null pointer");
    NRF_REQUIRE(n_points >= 0 && dims >= 1 && num_freqs >= 0, "nrf_pe_fwd: bad sizes");
    int out_dim = dims * ((include_input ? 1 : 0) + 2 * num_freqs);
    NRF_REQUIRE(out_dim > 0, "nrf_pe_fwd: empty embedding");
    if (n_points == 0) return NRF_OK;
    int64_t total = n_points * out_dim;
    hipLaunchKernelGGL(pe_fwd_kernel, dim3(nrf::grid_for(total, 256)), dim3(256), 0, nrf::as_stream(stream), x, total,
                       dims, num_freqs, max_freq_log2, log_sampling, include_input, out_dim, out);
    return nrf::check_launch("nrf_pe_fwd");
}
