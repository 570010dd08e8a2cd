// Microbenchmark (diagnostic): single-wave issue cadence of the instructions the K / M chain loops of
// noblank_km.hpp are made of (round 4).  One wave, 256 repetitions between two s_memtime reads.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate2.hip -o tools/micro/issue_rate2 && tools/micro/issue_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define DPPS "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

#define BENCH(NAME, NINSTR, BODY)                                                          \
    __global__ void NAME(unsigned long long *out, float *sink)                            \
    {                                                                                      \
        float v0 = threadIdx.x, v1 = 1.0001f, v2 = 3.f, v3 = 4.f, v4 = 5.f, v5 = 6.f, v6 = 7.f, v7 = 8.f; \
        int i0 = threadIdx.x, i1 = 3, i2 = 5, i3 = 7;                                      \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
        for (int it = 0; it < 4; ++it) {                                                   \
            asm volatile(REP64(BODY)                                                       \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), \
                           "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)                          \
                         :: "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29"); \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = 256ull * NINSTR; } \
        sink[threadIdx.x & 63] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + i0 + i1 + i2 + i3; \
    }

BENCH(dep_add, 1, "v_add_f32 %0, %0, %1\n\t")
BENCH(indep_add, 4, "v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %1\n\t")
BENCH(indep_cvt_flr, 4, "v_cvt_flr_i32_f32 %8, %0\n\tv_cvt_flr_i32_f32 %9, %1\n\tv_cvt_flr_i32_f32 %10, %2\n\tv_cvt_flr_i32_f32 %11, %3\n\t")
BENCH(indep_cvt_i32, 4, "v_cvt_i32_f32 %8, %0\n\tv_cvt_i32_f32 %9, %1\n\tv_cvt_i32_f32 %10, %2\n\tv_cvt_i32_f32 %11, %3\n\t")
BENCH(indep_floor, 4, "v_floor_f32 %4, %0\n\tv_floor_f32 %5, %1\n\tv_floor_f32 %6, %2\n\tv_floor_f32 %7, %3\n\t")
BENCH(indep_ldexp, 4, "v_ldexp_f32 %4, %0, %9\n\tv_ldexp_f32 %5, %1, %9\n\tv_ldexp_f32 %6, %2, %9\n\tv_ldexp_f32 %7, %3, %9\n\t")
BENCH(indep_ashr, 4, "v_ashrrev_i32 %8, 8, %9\n\tv_ashrrev_i32 %10, 8, %11\n\tv_ashrrev_i32 v20, 8, %9\n\tv_ashrrev_i32 v21, 8, %11\n\t")
BENCH(indep_addi_clamp, 4, "v_add_i32 v20, %8, %9 clamp\n\tv_add_i32 v21, %9, %10 clamp\n\tv_add_i32 v22, %10, %11 clamp\n\tv_add_i32 v23, %11, %8 clamp\n\t")
BENCH(indep_sub_dpp, 4, "v_sub_u32_dpp v20, %8, %9 " DPPS "\n\tv_sub_u32_dpp v21, %9, %10 " DPPS "\n\tv_sub_u32_dpp v22, %10, %11 " DPPS "\n\tv_sub_u32_dpp v23, %11, %8 " DPPS "\n\t")
BENCH(indep_fmac_dpp, 4, "v_fmac_f32_dpp v20, %0, %1 " DPPS "\n\tv_fmac_f32_dpp v21, %1, %2 " DPPS "\n\tv_fmac_f32_dpp v22, %2, %3 " DPPS "\n\tv_fmac_f32_dpp v23, %3, %0 " DPPS "\n\t")
BENCH(indep_maxf_dpp, 4, "v_max_f32_dpp v20, %0, %0 " DPPS "\n\tv_max_f32_dpp v21, %1, %1 " DPPS "\n\tv_max_f32_dpp v22, %2, %2 " DPPS "\n\tv_max_f32_dpp v23, %3, %3 " DPPS "\n\t")
BENCH(indep_maxi_dpp, 4, "v_max_i32_dpp v20, %8, %8 " DPPS "\n\tv_max_i32_dpp v21, %9, %9 " DPPS "\n\tv_max_i32_dpp v22, %10, %10 " DPPS "\n\tv_max_i32_dpp v23, %11, %11 " DPPS "\n\t")
BENCH(indep_pk_fma, 4, "v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[20:21]\n\tv_pk_fma_f32 v[26:27], v[22:23], v[24:25], v[26:27]\n\tv_pk_fma_f32 v[28:29], v[22:23], v[24:25], v[28:29]\n\tv_pk_mul_f32 v[22:23], v[22:23], v[24:25]\n\t")
// the float K step of noblank_km (first version): max_dpp, add, 2 x cvt_flr, sub
BENCH(k_step_float, 5, "v_max_f32_dpp v20, %0, %0 " DPPS "\n\tv_cvt_flr_i32_f32 %9, %1\n\tv_add_f32 %0, v20, %1\n\tv_cvt_flr_i32_f32 %8, %0\n\tv_sub_u32 %10, %8, %9\n\t")
// the fixed-point K step: max_i32_dpp, saturating add, two shifts, sub
BENCH(k_step_fixed, 5, "v_max_i32_dpp v20, %8, %8 " DPPS "\n\tv_ashrrev_i32 v21, 8, %9\n\tv_add_i32 %8, v20, %9 clamp\n\tv_ashrrev_i32 v22, 8, %8\n\tv_sub_u32 %10, v22, v21\n\t")
// the M step: sub, sub_dpp, 2 x ldexp, mul, fmac_dpp
BENCH(m_step, 6, "v_sub_u32 v20, %8, %9\n\tv_sub_u32_dpp v21, %8, %9 " DPPS "\n\tv_ldexp_f32 v20, %1, v20\n\tv_ldexp_f32 v21, %1, v21\n\tv_mul_f32 v22, %0, v20\n\tv_fmac_f32_dpp v22, %0, v21 " DPPS "\n\tv_mov_b32 %0, v22\n\t")
// the r16 chain step (8 VALU)
BENCH(r16_step, 8, "v_max_i32_dpp v20, %8, %8 " DPPS "\n\tv_add_u32 v23, v20, %9\n\tv_sub_u32 v21, %8, v20\n\tv_sub_u32_dpp v22, %8, v20 " DPPS "\n\tv_ldexp_f32 v21, %1, v21\n\tv_ldexp_f32 v22, %1, v22\n\tv_mul_f32 v24, %0, v21\n\tv_fmac_f32_dpp v24, %0, v22 " DPPS "\n\tv_mov_b32 %0, v24\n\tv_mov_b32 %8, v23\n\t")

typedef void (*kern_t)(unsigned long long *, float *);
struct Entry { const char *name; kern_t k; };
#define E(n) {#n, n}
int main()
{
    unsigned long long *out;
    float *sink;
    if (hipMalloc(&out, 256) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    std::vector<Entry> es = {E(dep_add), E(indep_add), E(indep_cvt_flr), E(indep_cvt_i32), E(indep_floor), E(indep_ldexp), E(indep_ashr),
                             E(indep_addi_clamp), E(indep_sub_dpp), E(indep_fmac_dpp), E(indep_maxf_dpp), E(indep_maxi_dpp), E(indep_pk_fma),
                             E(k_step_float), E(k_step_fixed), E(m_step), E(r16_step)};
    for (auto &e : es) {
        unsigned long long h[2] = {0, 0};
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, out, sink);
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        }
        printf("%-22s %8llu cycles / %5llu instr = %6.2f cycles per instr\n", e.name, h[0], h[1], (double)h[0] / h[1]);
    }
    return 0;
}
