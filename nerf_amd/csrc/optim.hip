// Fused Adam step over a list of parameter tensors: the step right after the path -- torch.optim.Adam
// as the reference configures it (model.py:52, vanilla_dynamic/model.py:115: betas (0.9, 0.999), eps 1e-8, no weight
// decay, no amsgrad) -- in ONE launch for all tensors, with an optional device-resident gradient scale (the
// clip_grad_norm_ coefficient of vanilla_dynamic/train.py:273-274) so that no host synchronisation is needed.
#include "nrf_common.h"

namespace {

constexpr int kMaxTensors = 48;
struct AdamArgs {
    float *p[kMaxTensors];
    const float *g[kMaxTensors];
    float *m[kMaxTensors], *v[kMaxTensors];
    int32_t blk0[kMaxTensors + 1];     // first block of each tensor (1024 elements per block)
    int64_t n[kMaxTensors];
    int32_t nt;
    float step_size[kMaxTensors], inv_bc2_sqrt[kMaxTensors];   // per tensor: torch keeps a step count per parameter
    float lerp_w, beta2, one_minus_beta2, eps;
    const float *grad_scale;           // device scalar or null
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float gs, bool scaled, float lerp_w, float beta2,
                                          float omb2, float inv_bc2_sqrt, float eps, float step_size) {
    const float gi = scaled ? g * gs : g;
    const float mi = m + lerp_w * (gi - m);                       // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v * beta2 + omb2 * (gi * gi);                // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;           // sqrt(v) / sqrt(bias_correction2) + eps
    m = mi;
    v = vi;
    p = p - step_size * (mi / denom);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// A block = 1024 consecutive elements of one tensor; the tensor is found by a scalar search over blk0.  The argument
// block is indexed by that block-uniform value: it is read where it lies, in the kernel-argument segment (constant
// address space -> scalar loads) -- indexing the by-value parameter made the compiler copy all 2 KB of it to per-lane
// scratch memory at the top of every block (55 us per step for 33 MB of traffic; 4.8 MB of parameters: see bench.py).
// Tensors whose four pointers are 16-byte aligned go through 16-byte accesses (one float4 per thread).
__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a_arg) {
    typedef const AdamArgs __attribute__((address_space(4))) *ArgPtr;
    const AdamArgs __attribute__((address_space(4))) &a = *(ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_arg;
    int t = 0;
    while (t + 1 < a.nt && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
    const int64_t base = (int64_t)((int)blockIdx.x - a.blk0[t]) * 1024;
    const float *gsp = a.grad_scale;
    const bool scaled = gsp != nullptr;
    const float gs = scaled ? *gsp : 1.0f;
    float *p = a.p[t], *m = a.m[t], *v = a.v[t];
    const float *g = a.g[t];
    const int64_t n = a.n[t];
    const float lerp_w = a.lerp_w, beta2 = a.beta2, omb2 = a.one_minus_beta2, eps = a.eps;
    const float ibc = a.inv_bc2_sqrt[t], ss = a.step_size[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t i4 = base + 4 * (int64_t)threadIdx.x;
    if (vec && i4 + 4 <= n) {
        float4 pv = *reinterpret_cast<float4 *>(p + i4), mv = *reinterpret_cast<float4 *>(m + i4);
        float4 vv = *reinterpret_cast<float4 *>(v + i4);
        const float4 gv = *reinterpret_cast<const float4 *>(g + i4);
        adam_elem(pv.x, gv.x, mv.x, vv.x, gs, scaled, lerp_w, beta2, omb2, ibc, eps, ss);
        adam_elem(pv.y, gv.y, mv.y, vv.y, gs, scaled, lerp_w, beta2, omb2, ibc, eps, ss);
        adam_elem(pv.z, gv.z, mv.z, vv.z, gs, scaled, lerp_w, beta2, omb2, ibc, eps, ss);
        adam_elem(pv.w, gv.w, mv.w, vv.w, gs, scaled, lerp_w, beta2, omb2, ibc, eps, ss);
        *reinterpret_cast<float4 *>(m + i4) = mv;
        *reinterpret_cast<float4 *>(v + i4) = vv;
        *reinterpret_cast<float4 *>(p + i4) = pv;
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = vec ? i4 + u : base + u * 256 + threadIdx.x;     // vec: this thread's ragged last quad
        if (i >= n) return;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, gs, scaled, lerp_w, beta2, omb2, ibc, eps, ss);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

}  // namespace

extern "C" int nrf_adam_step(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                             float *const *exp_avg_sq, const int64_t *numel, const int64_t *steps, float lr, float beta1,
                             float beta2, float eps, const float *grad_scale, void *stream) {
    NRF_REQUIRE(n_tensors >= 0, "nrf_adam_step: bad n_tensors");
    if (n_tensors == 0) return NRF_OK;
    NRF_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && steps, "nrf_adam_step: null pointer table");
    for (int t0 = 0; t0 < n_tensors; t0 += kMaxTensors) {
        AdamArgs a{};
        a.nt = n_tensors - t0 < kMaxTensors ? n_tensors - t0 : kMaxTensors;
        int blocks = 0;
        for (int t = 0; t < a.nt; ++t) {
            NRF_REQUIRE(params[t0 + t] && grads[t0 + t] && exp_avg[t0 + t] && exp_avg_sq[t0 + t] && numel[t0 + t] >= 0,
                        "nrf_adam_step: tensor %d has a null pointer or a negative size", t0 + t);
            a.p[t] = params[t0 + t]; a.g[t] = grads[t0 + t]; a.m[t] = exp_avg[t0 + t]; a.v[t] = exp_avg_sq[t0 + t];
            a.n[t] = numel[t0 + t];
            NRF_REQUIRE(steps[t0 + t] >= 1, "nrf_adam_step: step count of tensor %d must be >= 1", t0 + t);
            const double bc1 = 1.0 - pow((double)beta1, (double)steps[t0 + t]);
            const double bc2 = 1.0 - pow((double)beta2, (double)steps[t0 + t]);
            a.step_size[t] = (float)((double)lr / bc1);
            a.inv_bc2_sqrt[t] = (float)(1.0 / sqrt(bc2));
            a.blk0[t] = blocks;
            blocks += (int)((numel[t0 + t] + 1023) / 1024);
        }
        a.blk0[a.nt] = blocks;
        a.lerp_w = (float)(1.0 - (double)beta1);
        a.beta2 = beta2;
        a.one_minus_beta2 = (float)(1.0 - (double)beta2);
        a.eps = eps;
        a.grad_scale = grad_scale;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, nrf::as_stream(stream), a);
    }
    return nrf::check_launch("nrf_adam_step");
}
