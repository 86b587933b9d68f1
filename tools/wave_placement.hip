// wave_placement.hip -- which SIMD each wave of a 512-thread (8-wave) workgroup lands on (HW_REG_HW_ID bits [5:4]), one
// workgroup per CU with the whole LDS.  Decides how roles (loader / consumer) must be dealt to waves so that every SIMD gets one
// of each.  build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/wave_placement tools/wave_placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(512) k(uint32_t *out)
{
    extern __shared__ unsigned char smem[];
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 1000) smem[0] = 1;
}
int main()
{
    uint32_t *d, h[64 * 8];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 160 * 1024, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < 64; ++b) {
        if (b < 12) printf("wg %2d (cu %2u se %u):", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
        for (int w = 0; w < 8; ++w) {
            const int simd = (h[b * 8 + w] >> 4) & 3;
            if (b < 12) printf(" w%d->simd%d", w, simd);
            hist[w][simd]++;
        }
        if (b < 12) printf("\n");
    }
    int same_w_w4 = 0, distinct03 = 0;
    for (int b = 0; b < 64; ++b) {
        int s[8];
        for (int w = 0; w < 8; ++w) s[w] = (h[b * 8 + w] >> 4) & 3;
        same_w_w4 += (s[0] == s[4]) + (s[1] == s[5]) + (s[2] == s[6]) + (s[3] == s[7]) == 4;
        distinct03 += ((1 << s[0]) | (1 << s[1]) | (1 << s[2]) | (1 << s[3])) == 15;
    }
    printf("of 64 workgroups: waves w and w+4 share a SIMD in %d; waves 0-3 on four distinct SIMDs in %d\n", same_w_w4, distinct03);
    return 0;
}
