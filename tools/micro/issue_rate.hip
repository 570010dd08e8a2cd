// Microbenchmark (diagnostic, not part of the product): single-wave issue cadence of the
// instruction kinds the lattice chains are made of, on gfx950.  One wave, 256 repetitions of
// each pattern between two s_memtime reads; prints shader-clock cycles per instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define BENCH(NAME, NINSTR, BODY)                                                          \
    __global__ void NAME(unsigned long long *out, float *sink, float *lds_dummy)         \
    {                                                                                      \
        extern __shared__ float lds_dyn[];                                                 \
        float *lds = lds_dyn + out[3] / 4;                  /* byte offset of the working set */ \
        if (out[2] != 0 && threadIdx.x >= 64) return;      /* solo mode: the other waves leave */ \
        ((int *)lds)[threadIdx.x & 63] = (threadIdx.x * 4) % 256; lds[(threadIdx.x & 63) + 64] = 1.f;                      \
        float v0 = threadIdx.x, v1 = 1.0001f, v2 = 3.f, v3 = 4.f, v4 = 5.f, v5 = 6.f, v6 = 7.f, v7 = 8.f; \
        int i0 = threadIdx.x, i1 = 3, i2 = 5, i3 = (threadIdx.x & 63) < 20 ? (threadIdx.x & 63) * 8 : 1024;                                      \
        int addr = (threadIdx.x & 63) * (out[4] ? (int)out[4] : 8) + (int)out[3]; if (out[5] && (threadIdx.x & 63) >= out[5]) return;                                                        \
        __syncthreads();                                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
        for (int it = 0; it < 4; ++it) {                                                   \
            asm volatile(REP64(BODY)                                                       \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), \
                           "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(addr)              \
                         :: "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s20");                  \
        }                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = 256ull * NINSTR; }              \
        sink[threadIdx.x & 63] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + i0 + i1 + i2 + i3 + addr; \
    }

// operands: %0..%7 floats v0..v7, %8..%11 ints, %12 addr
BENCH(dep_add, 1, "v_add_f32 %0, %0, %1\n\t")
BENCH(indep_add, 4, "v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %1\n\t")
BENCH(dep_mul, 1, "v_mul_f32 %0, %0, %1\n\t")
BENCH(dep_ldexp, 1, "v_ldexp_f32 %0, %0, %9\n\t")
BENCH(indep_ldexp, 4, "v_ldexp_f32 %0, %0, %9\n\tv_ldexp_f32 %2, %2, %9\n\tv_ldexp_f32 %3, %3, %9\n\tv_ldexp_f32 %4, %4, %9\n\t")
BENCH(dep_maxi, 1, "v_max_i32 %8, %8, %9\n\t")
BENCH(dep_maxi_dpp, 1, "v_max_i32_dpp %8, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
BENCH(dep_mov_dpp, 1, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
BENCH(dep_mov_dpp_rowshr, 1, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
BENCH(dep_add_dpp_rowshr, 1, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
BENCH(dep_add_dpp_waveshr, 1, "v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
BENCH(dep_exp, 1, "v_exp_f32 %0, %0\n\t")
BENCH(indep_exp, 4, "v_exp_f32 %0, %1\n\tv_exp_f32 %2, %1\n\tv_exp_f32 %3, %1\n\tv_exp_f32 %4, %1\n\t")
BENCH(dep_log, 1, "v_log_f32 %0, %0\n\t")
BENCH(dep_frexp, 2, "v_frexp_exp_i32_f32 %8, %0\n\tv_frexp_mant_f32 %0, %0\n\t")
BENCH(dep_fma, 1, "v_fma_f32 %0, %0, %1, %2\n\t")
BENCH(dep_cndmask, 1, "v_cndmask_b32 %0, %0, %1, vcc\n\t")
BENCH(lds_read_dep, 1, "ds_read_b32 %12, %12\n\ts_waitcnt lgkmcnt(0)\n\t")
BENCH(lds_rw_indep, 2, "ds_read_b64 v[20:21], %12\n\tds_write_b64 %12, v[22:23] offset:2048\n\t")
BENCH(salu_dep, 1, "s_add_u32 s20, s20, 1\n\t")
BENCH(salu_valu_mix, 2, "s_add_u32 s20, s20, 1\n\tv_add_f32 %0, %0, %1\n\t")
// the extended-range chain step as compiled (11 VALU), no LDS
BENCH(xr_step, 11,
      "v_max_i32_dpp %10, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_sub_u32 %11, %8, %10\n\t"
      "v_sub_u32_dpp %8, %8, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %0, %0, %11\n\t"
      "v_ldexp_f32 %3, %2, %8\n\t"
      "v_add_f32 %0, %0, %3\n\t"
      "v_add_u32 %12, %12, %9\n\t"
      "v_mul_f32 %0, %1, %0\n\t"
      "v_add_u32 %8, %10, %9\n\t"
      "v_add_u32 %12, %12, %9\n\t")
// the split exponent / mantissa step (noblank_r16.hpp, round 2): compiler order (9 VALU, every
// instruction right behind the one it depends on), the same hand-interleaved (exponent side of step
// j+1 between the mantissa instructions of step j), and with the DPP read fused into v_fmac
BENCH(xr2_seq, 9,
      "v_max_i32_dpp v20, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_sub_u32 v21, %8, v20\n\t"
      "v_ldexp_f32 v21, %1, v21\n\t"
      "v_sub_u32_dpp v22, %8, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 v22, %1, v22\n\t"
      "v_mov_b32_dpp v23, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %0, %0, v21\n\t"
      "v_fmac_f32 %0, v23, v22\n\t"
      "v_add_u32 %8, v20, %9\n\t")
BENCH(xr2_pipe, 18,
      "v_max_i32_dpp v20, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v24, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 v25, %0, %2\n\t"
      "v_sub_u32 v21, %8, v20\n\t"
      "v_add_u32 %10, v20, %9\n\t"
      "v_sub_u32_dpp v22, %8, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %4, %1, v21\n\t"
      "v_fmac_f32 v25, v24, %3\n\t"
      "v_ldexp_f32 %5, %1, v22\n\t"
      "v_max_i32_dpp v20, %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v24, v25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %0, v25, %4\n\t"
      "v_sub_u32 v21, %10, v20\n\t"
      "v_add_u32 %8, v20, %9\n\t"
      "v_sub_u32_dpp v22, %10, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %2, %1, v21\n\t"
      "v_fmac_f32 %0, v24, %5\n\t"
      "v_ldexp_f32 %3, %1, v22\n\t")
BENCH(xr2_pipe_fused, 16,
      "v_max_i32_dpp v20, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 v25, %0, %2\n\t"
      "v_sub_u32 v21, %8, v20\n\t"
      "v_add_u32 %10, v20, %9\n\t"
      "v_sub_u32_dpp v22, %8, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %4, %1, v21\n\t"
      "v_fmac_f32_dpp v25, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %5, %1, v22\n\t"
      "v_max_i32_dpp v20, %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %0, v25, %4\n\t"
      "v_sub_u32 v21, %10, v20\n\t"
      "v_add_u32 %8, v20, %9\n\t"
      "v_sub_u32_dpp v22, %10, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %2, %1, v21\n\t"
      "v_fmac_f32_dpp %0, v25, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %3, %1, v22\n\t")
// the same with the LDS traffic of a pair of steps: one 16-byte read, one 16-byte write
BENCH(xr2_pipe_fused_lds, 18,
      "ds_read_b128 v[26:29], %12\n\t"
      "v_max_i32_dpp v20, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 v25, %0, %2\n\t"
      "v_sub_u32 v21, %8, v20\n\t"
      "v_add_u32 %10, v20, %9\n\t"
      "v_sub_u32_dpp v22, %8, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %4, %1, v21\n\t"
      "v_fmac_f32_dpp v25, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %5, %1, v22\n\t"
      "v_max_i32_dpp v20, %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %0, v25, %4\n\t"
      "v_sub_u32 v21, %10, v20\n\t"
      "v_add_u32 %8, v20, %9\n\t"
      "v_sub_u32_dpp v22, %10, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %2, %1, v21\n\t"
      "v_fmac_f32_dpp %0, v25, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_ldexp_f32 %3, %1, v22\n\t"
      "ds_write_b128 %12, v[20:23] offset:2048\n\t")

// the log-domain step (8 dependent ops incl. exp/log)
BENCH(log_step, 8,
      "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_sub_f32 %3, %0, %2\n\t"
      "v_mul_f32 %3, |%3|, %1\n\t"
      "v_exp_f32 %3, %3\n\t"
      "v_max_f32 %4, %0, %2\n\t"
      "v_add_f32 %3, 1.0, %3\n\t"
      "v_log_f32 %3, %3\n\t"
      "v_fma_f32 %0, %3, %1, %4\n\t")
// the common-scale linear step
BENCH(lin_step, 2,
      "v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %0, %0, %1\n\t")

// the fast-pass (double, common scale) chain step: product, two DPP moves, fma
BENCH(f64_step, 4,
      "v_mul_f64 v[20:21], v[22:23], v[20:21]\n\t"
      "v_mov_b32_dpp %2, v21 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %3, v20 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[20:21], v[22:23], v[20:21], v[20:21]\n\t")
BENCH(dep_mul_f64, 1, "v_mul_f64 v[20:21], v[22:23], v[20:21]\n\t")
BENCH(dep_fma_f64, 1, "v_fma_f64 v[20:21], v[22:23], v[20:21], v[20:21]\n\t")
// the same with its LDS traffic: one 8-byte read, one 8-byte write, two address bumps
BENCH(f64_step_lds, 8,
      "v_mul_f64 v[20:21], v[22:23], v[20:21]\n\t"
      "ds_read_b64 v[22:23], %12\n\t"
      "v_add_u32 %12, %12, %9\n\t"
      "v_mov_b32_dpp %2, v21 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %3, v20 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[20:21], v[22:23], v[20:21], v[20:21]\n\t"
      "ds_write_b64 %12, v[20:21] offset:2048\n\t"
      "s_waitcnt lgkmcnt(1)\n\t")
BENCH(lds_rw_b32, 2, "ds_read_b32 v20, %12\n\tds_write_b32 %12, v22 offset:2048\n\t")
BENCH(lds_rw_b128, 2, "ds_read_b128 v[20:23], %12\n\tds_write_b128 %12, v[20:23] offset:2048\n\t")

// chain-like LDS traffic: 20 lanes with their own cells, 44 idle lanes all on ONE spare address
BENCH(lds_w_idle_same, 1, "ds_write_b64 %11, v[22:23] offset:2048\n\t")
BENCH(lds_r_idle_same, 1, "ds_read_b64 v[20:21], %11\n\t")
BENCH(lds_w_distinct, 1, "ds_write_b64 %12, v[22:23] offset:2048\n\t")
BENCH(lds_r_distinct, 1, "ds_read_b64 v[20:21], %12\n\t")

// the fast chain's step exactly as compiled (two steps, state ping-pong), with and without LDS
BENCH(f64_chain_real, 14,
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_mul_f64 v[20:21], v[26:27], v[24:25]\n\t"
      "ds_write_b64 %12, v[20:21] offset:2048\n\t"
      "ds_read_b64 v[28:29], %12 offset:8\n\t"
      "v_mov_b32_dpp v21, v21 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v20, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[20:21], v[26:27], v[24:25], v[20:21]\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_mul_f64 v[24:25], v[28:29], v[20:21]\n\t"
      "ds_write_b64 %12, v[24:25] offset:2056\n\t"
      "ds_read_b64 v[26:27], %12 offset:16\n\t"
      "v_mov_b32_dpp v25, v25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v24, v24 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[24:25], v[28:29], v[20:21], v[24:25]\n\t"
)
BENCH(f64_chain_nowait, 12,
      "v_mul_f64 v[20:21], v[26:27], v[24:25]\n\t"
      "ds_write_b64 %12, v[20:21] offset:2048\n\t"
      "ds_read_b64 v[28:29], %12 offset:8\n\t"
      "v_mov_b32_dpp v21, v21 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v20, v20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[20:21], v[26:27], v[24:25], v[20:21]\n\t"
      "v_mul_f64 v[24:25], v[28:29], v[20:21]\n\t"
      "ds_write_b64 %12, v[24:25] offset:2056\n\t"
      "ds_read_b64 v[26:27], %12 offset:16\n\t"
      "v_mov_b32_dpp v25, v25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp v24, v24 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f64 v[24:25], v[28:29], v[20:21], v[24:25]\n\t"
)

// LDS float atomics (no return): distinct addresses / all lanes on one address
BENCH(lds_add_f32_distinct, 1, "ds_add_f32 %12, v22 offset:2048\n\t")
BENCH(lds_add_f32_same, 1, "ds_add_f32 %11, v22 offset:2048\n\t")

typedef void (*kern_t)(unsigned long long *, float *, float *);
struct Entry { const char *name; kern_t k; int waves; int grid; int solo; int ldsb; int off; int pitch; int lanes; };
#define E(n) {#n, n, 1, 1, 0, 16384, 0}
#define EW(n, w) {#n " x" #w " waves", n, w, 1, 0, 16384, 0}
#define EG(n, w, g, so) {#n " " #w "w grid" #g " solo" #so, n, w, g, so, 16384, 0}
#define EL(n, ldsb, off) {#n " lds" #ldsb " off" #off, n, 1, 1, 0, ldsb, off}

int main()
{
    unsigned long long *out;
    float *sink;
    if (hipMalloc(&out, 256) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    std::vector<Entry> es = {E(dep_add), E(indep_add), E(dep_mul), E(dep_ldexp), E(indep_ldexp), E(dep_maxi), E(dep_maxi_dpp),
                             E(dep_mov_dpp), E(dep_mov_dpp_rowshr), E(dep_add_dpp_rowshr), E(dep_add_dpp_waveshr), E(dep_exp),
                             E(indep_exp), E(dep_log), E(dep_frexp), E(dep_fma), E(dep_cndmask), E(lds_read_dep), E(lds_rw_indep),
                             E(salu_dep), E(salu_valu_mix), E(xr_step), E(xr2_seq), E(xr2_pipe), E(xr2_pipe_fused), E(log_step), E(lin_step), E(f64_step), E(dep_mul_f64), E(dep_fma_f64),
                             E(f64_step_lds), E(lds_rw_b32), E(lds_rw_b128), E(lds_w_idle_same), E(lds_r_idle_same),
                             E(lds_w_distinct), E(lds_r_distinct), EW(lds_w_distinct, 2), EW(lds_w_distinct, 4),
                             EW(lds_r_distinct, 2), EW(lds_rw_indep, 2), EW(lds_rw_indep, 4), EW(lds_rw_indep, 16), EW(dep_add, 2), EW(dep_add, 8),
                             EW(f64_step, 2), EW(lds_rw_b32, 2), EW(lds_rw_b32, 4), E(f64_chain_real), E(f64_chain_nowait),
                             EW(f64_chain_real, 2), EW(f64_chain_real, 16), EG(f64_chain_real, 1, 256, 0), EG(f64_chain_real, 16, 1, 1),
                             EG(f64_chain_real, 16, 256, 1), EG(f64_chain_real, 2, 256, 0), EG(dep_add, 1, 256, 0), EG(dep_add, 16, 256, 1), EG(f64_step, 16, 256, 1),
                             EL(f64_chain_real, 143360, 0), EL(f64_chain_real, 143360, 65536), EL(f64_chain_real, 143360, 131072),
                             EL(lds_r_distinct, 143360, 0), EL(lds_r_distinct, 143360, 131072), EL(lds_w_distinct, 143360, 131072), EL(lds_read_dep, 143360, 0),
                             {"f64_chain_real pitch1416 lanes20", f64_chain_real, 1, 1, 0, 143360, 0, 1416, 20},
                             {"f64_chain_real pitch1416 lanes64", f64_chain_real, 1, 1, 0, 143360, 0, 1416, 64},
                             {"f64_chain_real pitch8 lanes20", f64_chain_real, 1, 1, 0, 143360, 0, 8, 20},
                             {"lds_w_distinct pitch1416 lanes20", lds_w_distinct, 1, 1, 0, 143360, 0, 1416, 20},
                             {"lds_r_distinct pitch1416 lanes20", lds_r_distinct, 1, 1, 0, 143360, 0, 1416, 20},
                             {"lds_rw_indep pitch1416 lanes20", lds_rw_indep, 1, 1, 0, 143360, 0, 1416, 20},
                             {"xr2_pipe_fused_lds pitch16 lanes64", xr2_pipe_fused_lds, 1, 1, 0, 143360, 0, 16, 64},
                             {"xr2_pipe_fused_lds pitch1392 lanes20", xr2_pipe_fused_lds, 1, 1, 0, 143360, 0, 1392, 20},
                             {"xr2_pipe_fused_lds pitch1392 lanes20 16 waves", xr2_pipe_fused_lds, 16, 1, 0, 143360, 0, 1392, 20},
                             E(lds_add_f32_distinct), E(lds_add_f32_same),
                             {"lds_add_f32_distinct lanes20", lds_add_f32_distinct, 1, 1, 0, 16384, 0, 8, 20},
                             {"lds_add_f32_distinct 16 waves", lds_add_f32_distinct, 16, 1, 0, 16384, 0, 8, 0}};
    for (auto &e : es) {
        unsigned long long h[2] = {0, 0};
        for (int r = 0; r < 3; ++r) {
            unsigned long long cfg[4] = {(unsigned long long)e.solo, (unsigned long long)e.off, (unsigned long long)e.pitch, (unsigned long long)e.lanes};
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 4;
            if (hipMemcpy(out + 2, cfg, 32, hipMemcpyHostToDevice) != hipSuccess) return 3;
            hipLaunchKernelGGL(e.k, dim3(e.grid), dim3(64 * e.waves), e.ldsb, 0, out, sink, sink);
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        }
        printf("%-30s %8llu cycles / %5llu instr = %6.2f cycles per instr\n", e.name, h[0], h[1], (double)h[0] / h[1]);
    }
    return 0;
}
