// microbench: how many VALU fillers fit under v_mfma_f32_32x32x16_bf16, by accumulator pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int PATTERN>   // PATTERN 0: 3 consecutive MFMAs per tile (x3 order); 1: round robin over 8 tiles; 2: one dependent chain
__global__ void __launch_bounds__(256, 1) k(float *out, int iters, long long *cyc) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const int t = PATTERN == 0 ? m / 3 : (PATTERN == 1 ? m % 8 : 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("v_max_f32_e32 %0, 0x3f000000, %0" : "+v"(v[f % 8]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int F, int P>
void run(float *out, long long *cyc, const char *name) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F, P>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F, P>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s F=%d: %.1f ns per MFMA (%.3f ms), s_memtime ticks per MFMA %.1f\n", name, F, ms * 1e6 / (iters * 24.0), ms, (double)c / (iters * 24.0));
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
#define ROW(P, name) run<0, P>(out, cyc, name); run<2, P>(out, cyc, name); run<3, P>(out, cyc, name); run<4, P>(out, cyc, name); run<5, P>(out, cyc, name); run<6, P>(out, cyc, name); run<7, P>(out, cyc, name); run<8, P>(out, cyc, name);
    ROW(0, "x3-order") ROW(1, "round-robin") ROW(2, "one-chain")
    return 0;
}
