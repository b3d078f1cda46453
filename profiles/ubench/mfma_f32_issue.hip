// Microbenchmark: issue cadence of v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles) at one wave per SIMD:
// dependent chains vs independent accumulators, B operand from VGPR vs AGPR, VALU fillers between MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: one accumulator (dependent chain); 1: runs of 4 on one accumulator, 8 accumulators (the fused kernel's
// order); 2: 8 accumulators round robin; BA: B operand in an AGPR; NFILL VALU ops after each MFMA
template <int MODE, int BA, int NFILL, int THREADS = 256>
__global__ void __launch_bounds__(THREADS, 1) k(float *out, int iters) {
    float a = threadIdx.x * 0.001f, b = 0.5f;
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int t = MODE == 0 ? 0 : (MODE == 1 ? u / 4 : u % 8);
            if (BA) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "a"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NFILL; ++f) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[f]) : "v"(a));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int BA, int NFILL>
void run(const char *name, float *d) {
    hipLaunchKernelGGL((k<MODE, BA, NFILL>), dim3(256), dim3(256), 0, 0, d, 500);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BA, NFILL>), dim3(256), dim3(256), 0, 0, d, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (20000 * 32.0);
    printf("%-34s B in %s fill=%2d  %.2f ns/MFMA = %.1f cycles at 2.4 GHz  (%.1f TFLOP/s chip)\n", name, BA ? "AGPR" : "VGPR", NFILL,
           ns, ns * 2.4, 4096.0 * 1024 / ns / 1e3);
}
// two waves per SIMD (512 threads, <= 256 registers per wave): cycles per MFMA of the SIMD's matrix pipe
template <int MODE, int BA, int NFILL>
void run2(const char *name, float *d) {
    hipLaunchKernelGGL((k<MODE, BA, NFILL, 512>), dim3(256), dim3(512), 0, 0, d, 500);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BA, NFILL, 512>), dim3(256), dim3(512), 0, 0, d, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (20000 * 32.0 * 2);
    printf("2 waves/SIMD %-21s B in %s fill=%2d  %.2f ns/MFMA = %.1f cycles at 2.4 GHz  (%.1f TFLOP/s chip)\n", name, BA ? "AGPR" : "VGPR",
           NFILL, ns, ns * 2.4, 4096.0 * 1024 / ns / 1e3);
}
int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0, 0>("one accumulator", d);
    run<1, 0, 0>("runs of 4 per accumulator", d);
    run<2, 0, 0>("8 accumulators round robin", d);
    run<1, 1, 0>("runs of 4 per accumulator", d);
    run<2, 1, 0>("8 accumulators round robin", d);
    run<1, 1, 2>("runs of 4 per accumulator", d);
    run<1, 1, 4>("runs of 4 per accumulator", d);
    run<1, 1, 8>("runs of 4 per accumulator", d);
    run<1, 1, 12>("runs of 4 per accumulator", d);
    run<2, 1, 8>("8 accumulators round robin", d);
    run<2, 1, 12>("8 accumulators round robin", d);
    run2<1, 1, 0>("runs of 4", d);
    run2<1, 1, 4>("runs of 4", d);
    run2<1, 1, 8>("runs of 4", d);
    run2<1, 1, 12>("runs of 4", d);
    run2<1, 1, 16>("runs of 4", d);
    return 0;
}
