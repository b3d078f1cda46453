// ds_read_b64_tr_b16 on an image of 64-byte rows (row = point: [j = 0: 16 slots][j = 1: 16 slots] of 16 bits): lane
// (j = (lane >> 4) & 1, s = lane >> 5, i = lane & 15) must receive slot 16 j + i of points 8 s .. 8 s + 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *o) {
    __shared__ __attribute__((aligned(1024))) unsigned short img[32 * 32];
    for (int i = threadIdx.x; i < 1024; i += 64) img[i] = (unsigned short)((i >> 5) * 256 + (i & 31));
    __syncthreads();
    const int lane = threadIdx.x, j = (lane >> 4) & 1, s = lane >> 5, i = lane & 15, q = i >> 2, p = i & 3;
    const char *a = reinterpret_cast<const char *>(img) + (8 * s + q) * 64 + j * 32 + p * 8;
    v4i16 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)a);
    v4i16 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)(a + 256));
    for (int e = 0; e < 4; ++e) { o[lane * 8 + e] = (unsigned short)v0[e]; o[lane * 8 + 4 + e] = (unsigned short)v1[e]; }
}
int main() {
    unsigned short *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
            const int j = (lane >> 4) & 1, s = lane >> 5, i = lane & 15;
            const int want = (8 * s + e) * 256 + 16 * j + i;
            if (h[lane * 8 + e] != want) { if (bad < 8) printf("lane %d e %d: got point %d slot %d, want point %d slot %d\n", lane, e, h[lane*8+e] >> 8, h[lane*8+e] & 255, want >> 8, want & 255); ++bad; }
        }
    printf("mismatches %d of 512\n", bad);
    return 0;
}
