// Diagnostic (not on the product path): a kernel with the FOOTPRINT of RCCL's collective kernel on gfx950, to
// measure what a collective that is resident on k CUs does to the one-workgroup-per-CU loss launch.
// Footprint taken from the kernel descriptors of torch's librccl.so (rcclGenericKernel<1|2|4, *>, gfx950 code
// object): 256 threads, 19 744 B of LDS, 261 - 280 VGPRs (one wave per SIMD).  The loss kernel's workgroup
// (1024 threads, 68 -> 72 VGPRs per wave, four waves per SIMD, 122 KB of LDS) needs 288 of a SIMD's 512
// registers per lane: 288 + 280 > 512, so the two can NOT share a CU -- a resident collective workgroup takes
// a whole CU away from the launch.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/coresident.hip -o ctc_amd/lib/libcoresident.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void occupant_kernel(unsigned long long ticks, int *sink)
{
    __shared__ int lds[19744 / 4];
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a23, 0" ::: "v255", "a23");   // 256 + 24 = 280 registers per lane, like rcclGenericKernel<4, *>
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}

// k workgroups that stay resident for `usec` microseconds
extern "C" int coresident_launch(int k, double usec, void *sink, void *stream)
{
    hipLaunchKernelGGL(occupant_kernel, dim3(k), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (unsigned long long)(usec * 100.0), static_cast<int *>(sink));
    return (int)hipGetLastError();
}
