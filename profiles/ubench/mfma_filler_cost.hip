// Microbenchmark: cost of VALU fillers beside v_mfma_f32_32x32x16_bf16 at one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NFILL>
__global__ void __launch_bounds__(256, 1) k(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc0 = {0}, acc1 = {0}, other = {0};
    for (int i = 0; i < 16; ++i) other[i] = threadIdx.x + i;
    asm volatile("" : "+a"(other));
    float r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            if (MODE == 1) {   // AGPR reads of registers no MFMA touches
                if (NFILL > 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r0) : "a"(other[0]));
                if (NFILL > 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r1) : "a"(other[1]));
                if (NFILL > 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r2) : "a"(other[2]));
                if (NFILL > 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r3) : "a"(other[3]));
                if (NFILL > 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r4) : "a"(other[4]));
                if (NFILL > 5) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r5) : "a"(other[5]));
            } else if (MODE == 2) {   // plain VALU on VGPRs
                if (NFILL > 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r0) : "v"(r6), "v"(r7));
                if (NFILL > 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r1) : "v"(r6), "v"(r7));
                if (NFILL > 2) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(r2) : "v"(r6));
                if (NFILL > 3) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(r3) : "v"(r7));
                if (NFILL > 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r4) : "v"(r6), "v"(r7));
                if (NFILL > 5) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(r5) : "v"(r6));
            } else if (MODE == 3) {   // AGPR reads of the accumulator the OTHER mfma chain owns
                if (NFILL > 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r0) : "a"(acc1[0]));
                if (NFILL > 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r1) : "a"(acc1[1]));
                if (NFILL > 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r2) : "a"(acc1[2]));
                if (NFILL > 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r3) : "a"(acc1[3]));
            }
        }
    }
    long t1 = clock64();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + other[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8.0f);
}

template <int MODE, int NFILL>
void run(const char *name, float *d) {
    hipLaunchKernelGGL((k<MODE, NFILL>), dim3(256), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NFILL>), dim3(256), dim3(256), 0, 0, d, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-28s fill=%d  %.1f ns/MFMA wall, clock64 ticks/MFMA %.2f\n", name, NFILL, ms * 1e6 / (20000 * 8.0), h);
}
int main() {
    float *d; hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>("mfma only", d);
    run<1, 2>("agpr read (untouched regs)", d); run<1, 4>("agpr read (untouched regs)", d); run<1, 6>("agpr read (untouched regs)", d);
    run<2, 2>("vgpr valu", d); run<2, 4>("vgpr valu", d); run<2, 6>("vgpr valu", d);
    run<3, 2>("agpr read (idle acc)", d); run<3, 4>("agpr read (idle acc)", d);
    return 0;
}
