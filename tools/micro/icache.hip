// Instruction-cache capacity / miss cost: straight-line code of N x 4 KB executed in a loop by one wave per CU.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/icache.hip -o tools/micro/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define BLOCK4K asm volatile(".rept 1024\n\tv_add_f32 %0, %0, %0\n\t.endr" : "+v"(x));
#define B(n) if (nblocks > n) { BLOCK4K }
__global__ void walk(float *out, unsigned long long *cyc, int nblocks, int passes)
{
    float x = threadIdx.x;
    unsigned long long t0 = 0;
    for (int p = 0; p < passes + 1; ++p) {
        if (p == 1) t0 = __builtin_amdgcn_s_memtime();      // pass 0 warms whatever can be warmed
        B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7) B(8) B(9) B(10) B(11) B(12) B(13) B(14) B(15)
        B(16) B(17) B(18) B(19) B(20) B(21) B(22) B(23) B(24) B(25) B(26) B(27)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main()
{
    float *o; unsigned long long *c, h[512];
    hipMalloc(&o, 512 * 1024 * 4); hipMalloc(&c, sizeof(h));
    const int passes = 8;
    for (int waves = 1; waves <= 16; waves *= 4)
        for (int nb : {1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28}) {
            walk<<<256, 64 * waves, 0, 0>>>(o, c, nb, passes);
            hipDeviceSynchronize();
            hipMemcpy(h, c, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
            printf("waves/CU %2d  code %3d KB: %.2f cycles per instruction (wave 0 of each workgroup, mean over 256)\n", waves, nb * 4,
                   s / 256 / passes / (nb * 1024.0));
        }
    return 0;
}
