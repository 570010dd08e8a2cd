// Which SIMD does wave w of a 1024-thread workgroup run on?  (HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/wave_simd.hip -o tools/micro/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d, h[4 * 16];
    hipMalloc(&d, sizeof(h));
    probe<<<4, 1024, 0, 0>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 16; ++w) printf(" w%d:simd%u/slot%u", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15);
        printf("  (cu %u)\n", (h[b * 16] >> 8) & 15);
    }
    return 0;
}
