// Microbenchmark (diagnostic): the hand-scheduled K / M chain loops of noblank_km.hpp alone -- one workgroup, one wave
// running a loop over cells that are already there, the other waves absent or spinning on LDS.
//   hipcc --offload-arch=gfx950 -O2 -I ctc_amd/csrc tools/micro/km_probe.hip -o tools/micro/km_probe && tools/micro/km_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "noblank_km_asm.hpp"

__global__ __launch_bounds__(1024) void probe(unsigned long long *out, int which, int nb, int others)
{
    extern __shared__ float4 raw[];
    float *lds = reinterpret_cast<float *>(raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 36000; i += blockDim.x) lds[i] = (i & 1) ? -3.5f : 1.25f;   // (pm, e) cells everywhere
    __syncthreads();
    if (w != 0) {
        if (others == 0) return;
        float acc = 0.f;                                         // LDS traffic like a worker's tile work
        volatile float *t = lds + 20000 + w * 640 + lane * 4;
        for (int r = 0; r < others; ++r) { t[0] = acc; acc += t[1] + t[64]; }
        if (acc == 12345.f) out[7] = 1;
        return;
    }
    const int lrow = (lane >= 1 && lane <= 20) ? lane - 1 : 20;
    const int TPc = 166;
    unsigned ea = (unsigned)((lrow * TPc + 8) * 8), oa = (unsigned)(30000 * 4 / 4 * 0 + (21 * TPc * 8) + (lrow * TPc + 8) * 8);
    unsigned ka = oa, ma = (unsigned)(2 * 21 * TPc * 8 + (lrow * 172 + 8) * 4);
    const unsigned pa = 140000;
    const unsigned long long live = (4ull << 20) - 1;
    int pc = 0, nbs = nb;
    float kf = lane == 1 ? 0.f : -16777216.f, m = lane == 1 ? 1.f : 0.f;
    int kp = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (which == 0)
        asm volatile(CTC_KM_K_FWD : [ea] "+v"(ea), [oa] "+v"(oa), [kf] "+v"(kf), [pc] "+v"(pc), [nb] "+s"(nbs) : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
    else
        asm volatile(CTC_KM_M_FWD : [ka] "+v"(ka), [ea] "+v"(ea), [oa] "+v"(ma), [m] "+v"(m), [kp] "+v"(kp), [pc] "+v"(pc), [nb] "+s"(nbs)
                     : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(kf + m); }
}

int main()
{
    unsigned long long *out, h[2];
    hipMalloc(&out, 256);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *names[2] = {"K loop", "M loop"};
    for (int which = 0; which < 2; ++which)
        for (int waves : {1, 16})
            for (int others : {0, 4000})
                for (int nb : {1, 4, 8}) {
                    if (waves == 1 && others) continue;
                    for (int r = 0; r < 3; ++r) {
                        hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 150 * 1024, 0, out, which, nb, others);
                        hipDeviceSynchronize();
                    }
                    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
                    printf("%s, %2d waves%s, %d double blocks: %6llu cycles = %5.1f per step\n", names[which], waves,
                           others ? " (15 on LDS)" : (waves > 1 ? " (15 gone)" : ""), nb, h[0], (double)h[0] / (16.0 * nb));
                }
    return 0;
}
