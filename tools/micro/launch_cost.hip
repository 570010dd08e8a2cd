// Microbenchmark (diagnostic): what a dispatch of B workgroups costs on gfx950 before it does any
// work, as a function of workgroup size and dynamic LDS -- back-to-back launches, HIP events.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o tools/micro/launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel(float *p) { if (p == nullptr) p[threadIdx.x] = 0.f; }
// touches global memory once per wave (a load whose result is stored), like a real kernel's first access
__global__ void touch_kernel(float *p) { if ((threadIdx.x & 63) == 0) p[blockIdx.x * 16 + (threadIdx.x >> 6)] += 1.f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    float *buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(touch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int grids[] = {256, 2048};
    const int threads[] = {64, 256, 512, 1024};
    const int lds[] = {0, 64 * 1024, 134 * 1024};
    for (int touch = 0; touch < 2; ++touch)
        for (int g : grids)
            for (int t : threads)
                for (int l : lds) {
                    const int n = 400;
                    for (int r = 0; r < 2; ++r) {
                        CK(hipEventRecord(e0, s));
                        for (int i = 0; i < n; ++i) {
                            if (touch) hipLaunchKernelGGL(touch_kernel, dim3(g), dim3(t), l, s, buf);
                            else hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(t), l, s, buf);
                        }
                        CK(hipEventRecord(e1, s));
                        CK(hipEventSynchronize(e1));
                    }
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("%s grid %4d threads %4d lds %6d: %6.2f us per launch (back-to-back)\n", touch ? "touch" : "empty", g, t,
                           l, ms * 1000.f / n);
                }
    return 0;
}
