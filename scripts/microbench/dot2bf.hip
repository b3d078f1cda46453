// is x - float(bf16(x)) through v_dot2c_f32_bf16 exact?  (split2 with 4 VALU ops a pair instead of 6)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *o, const float *in, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = in[2 * i], b = in[2 * i + 1];
    bf16x2 hi = __builtin_convertvector((f32x2){a, b}, bf16x2);
    // as compile-time constants (-1, 0) becomes the INLINE constant -1.0, which the instruction reads as f32 bits: (0, -1)
    unsigned c0 = 0x0000BF80u, c1 = 0xBF800000u;
    asm volatile("" : "+s"(c0), "+s"(c1));
    bf16x2 m0 = __builtin_bit_cast(bf16x2, c0), m1 = __builtin_bit_cast(bf16x2, c1);
    o[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(hi, m0, a, false);
    o[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(hi, m1, b, false);
}
static float bf16_rne(float x) { unsigned u; memcpy(&u, &x, 4); unsigned r = u + 0x7fffu + ((u >> 16) & 1u); r &= 0xffff0000u; float y; memcpy(&y, &r, 4); return y; }
int main() {
    const int n = 1 << 20;
    float *h = (float *)malloc(n * 4), *r = (float *)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float m = (float)rand() / RAND_MAX * 2 - 1;
        int e = rand() % 40 - 30;
        h[i] = ldexpf(m, e);
        if (i % 97 == 0) h[i] = 0.0f;
    }
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, o, d, n);
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
        float want = h[i] - bf16_rne(h[i]);
        if (memcmp(&want, &r[i], 4) != 0 && !(want == 0.0f && r[i] == 0.0f)) { if (bad < 5) printf("x=%a want %a got %a\n", h[i], want, r[i]); ++bad; double e = fabs((double)want - r[i]) / fmax(fabs((double)h[i]), 1e-40); if (e > worst) worst = e; }
    }
    printf("mismatches %ld of %d, worst relative to x %.3e\n", bad, n, worst);
    return 0;
}
