// Microbenchmark (diagnostic, not part of the product): what the matrix instructions of binary.hip / binary_flow.hpp cost
// on gfx950, one wave issuing them back to back -- independent accumulators and one dependent chain -- and what a wave of
// plain VALU work on the SAME SIMD sees while they run (waves w and w + 4 of a workgroup share a SIMD).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 256;

template <int KIND, bool DEP>
__device__ __forceinline__ void matrix_loop(f32x4 (&acc)[4], float a, float b, u32x4 ha, u32x4 hb)
{
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 &c = acc[DEP ? 0 : i];
            if (KIND == 0) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
            if (KIND == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ha), __builtin_bit_cast(bf16x8, hb), c, 0, 0, 0);
            if (KIND == 2) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        }
    }
}

// mode 0: wave 0 issues matrix instructions alone; mode 1: wave 4 (same SIMD) runs a VALU loop beside it; mode 2: the VALU loop alone
template <int KIND, bool DEP>
__global__ void bench(unsigned long long *out, float *sink, int mode)
{
    const int w = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + a;
    u32x4 ha = {threadIdx.x, 1, 2, 3}, hb = {5, 6, threadIdx.x, 7};
    __syncthreads();
    if (w == 0 && mode != 2) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        matrix_loop<KIND, DEP>(acc, a, b, ha, hb);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[0] = t1 - t0;
        sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (w == 4 && mode != 0) {
        float v0 = a, v1 = b, v2 = a + 2, v3 = b + 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 4 * kIters; ++it) {
            v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
            v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[1] = t1 - t0;
        sink[threadIdx.x] = v0 + v1 + v2 + v3;
    }
}

template <int KIND, bool DEP>
static void run(const char *name, unsigned long long *out, float *sink)
{
    unsigned long long h[2];
    double alone = 0, beside = 0, valu_alone = 0, valu_beside = 0;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(out, 0, 16);
            hipLaunchKernelGGL((bench<KIND, DEP>), dim3(1), dim3(512), 0, 0, out, sink, mode);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        if (mode == 0) alone = h[0] / (4.0 * kIters);
        if (mode == 1) { beside = h[0] / (4.0 * kIters); valu_beside = h[1] / (16.0 * kIters); }
        if (mode == 2) valu_alone = h[1] / (16.0 * kIters);
    }
    printf("%-34s %6.1f cycles each alone, %6.1f with a VALU wave on the SIMD (whose fma costs %.2f instead of %.2f cycles)\n",
           name, alone, beside, valu_beside, valu_alone);
}

int main()
{
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4096);
    run<0, false>("v_mfma_f32_16x16x4_f32, 4 accs", out, sink);
    run<0, true>("v_mfma_f32_16x16x4_f32, dependent", out, sink);
    run<1, false>("v_mfma_f32_16x16x32_bf16, 4 accs", out, sink);
    run<1, true>("v_mfma_f32_16x16x32_bf16, dependent", out, sink);
    run<2, false>("v_mfma_f32_4x4x1_16b_f32, 4 accs", out, sink);
    run<2, true>("v_mfma_f32_4x4x1_16b_f32, dependent", out, sink);
    return 0;
}
