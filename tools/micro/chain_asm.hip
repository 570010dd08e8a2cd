// Microbenchmark (diagnostic): one block of 16 chain steps of noblank_r16.hpp exactly as hipcc compiled it
// (no LDS traffic), in a loop, one wave: cycles per step.  Variants: with the compiler's s_nop, without.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v16","v18","memory"
#define INIT "v_mov_b32 v2, 1.5\n\tv_mov_b32 v3, -3\n\tv_mov_b32 v4, 1.25\n\tv_mov_b32 v5, -2\n\tv_mov_b32 v6, 1.75\n\tv_mov_b32 v7, -4\n\tv_mov_b32 v8, 1.5\n\tv_mov_b32 v9, -3\n\tv_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 0x20000000\n\t"
__global__ void k_as_compiled(unsigned long long *out, int iters, int lanes, int alive, int prio)
{
    extern __shared__ float dyn_lds[];
    if ((int)(threadIdx.x >> 6) >= alive || (int)(threadIdx.x & 63) >= lanes) return;
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
    if (iters < 0) dyn_lds[threadIdx.x] = 1.f;
    const int data_mode = prio >> 1;
    asm volatile(INIT ::: CLOB);
    if (data_mode == 1 && (threadIdx.x & 63) >= 20)          // idle lanes: no mass, zero emissions
        asm volatile("v_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                     "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0" ::: CLOB);
    if (data_mode == 2 && (threadIdx.x & 63) >= 20)          // idle lanes: no mass, emissions like the others
        asm volatile("v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0" ::: CLOB);
    if (data_mode == 3)                                      // every lane its own exponent
        asm volatile("v_mad_u32_u24 v11, %0, 7, v11" :: "v"(threadIdx.x & 63) : CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_frexp_exp_i32_f32_e32 v12, v10\n\t"
        "v_add_u32_e32 v11, v12, v11\n\t"
        "v_frexp_mant_f32_e32 v10, v10\n\t"
        "v_max_i32_dpp v12, v11, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v11, v12\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v11, v11, v12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_e32 v13, v10, v13\n\t"
        "v_ldexp_f32 v11, v8, v11\n\t"
        "v_fmac_f32_dpp v13, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v12, v9\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v18, v13, v12\n\t"
        "v_fmac_f32_dpp v18, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v4, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v13, v11, v5\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v12, v18, v12\n\t"
        "v_max_i32_dpp v11, v13, v13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp v12, v18, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v10, v13, v11\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_sub_u32_dpp v18, v13, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_e32 v10, v12, v10\n\t"
        "v_add_u32_e32 v11, v11, v3\n\t"
        "v_ldexp_f32 v18, v2, v18\n\t"
        "v_fmac_f32_dpp v10, v12, v18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "ds_write_b128 v16, v[10:13]\n\t"
        "v_add_u32_e32 v16, 0xffffff80, v16\n\t"
        ::: CLOB);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_no_nop(unsigned long long *out, int iters, int lanes, int alive, int prio)
{
    extern __shared__ float dyn_lds[];
    if ((int)(threadIdx.x >> 6) >= alive || (int)(threadIdx.x & 63) >= lanes) return;
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
    if (iters < 0) dyn_lds[threadIdx.x] = 1.f;
    const int data_mode = prio >> 1;
    asm volatile(INIT ::: CLOB);
    if (data_mode == 1 && (threadIdx.x & 63) >= 20)          // idle lanes: no mass, zero emissions
        asm volatile("v_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                     "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0" ::: CLOB);
    if (data_mode == 2 && (threadIdx.x & 63) >= 20)          // idle lanes: no mass, emissions like the others
        asm volatile("v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0" ::: CLOB);
    if (data_mode == 3)                                      // every lane its own exponent
        asm volatile("v_mad_u32_u24 v11, %0, 7, v11" :: "v"(threadIdx.x & 63) : CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_frexp_exp_i32_f32_e32 v12, v10\n\t"
        "v_add_u32_e32 v11, v12, v11\n\t"
        "v_frexp_mant_f32_e32 v10, v10\n\t"
        "v_max_i32_dpp v12, v11, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v11, v12\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v11, v11, v12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_e32 v13, v10, v13\n\t"
        "v_ldexp_f32 v11, v8, v11\n\t"
        "v_fmac_f32_dpp v13, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v12, v9\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v4, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v5\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v2, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_mul_f32_e32 v12, v13, v12\n\t"
        "v_fmac_f32_dpp v12, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v3\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v13, v10, v11\n\t"
        "v_ldexp_f32 v13, v8, v13\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v8, v10\n\t"
        "v_mul_f32_e32 v13, v12, v13\n\t"
        "v_fmac_f32_dpp v13, v12, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v9\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v6, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ldexp_f32 v10, v6, v10\n\t"
        "v_mul_f32_e32 v18, v13, v12\n\t"
        "v_fmac_f32_dpp v18, v13, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v10, v11, v7\n\t"
        "v_max_i32_dpp v11, v10, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v12, v10, v11\n\t"
        "v_ldexp_f32 v12, v4, v12\n\t"
        "v_sub_u32_dpp v10, v10, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32_e32 v13, v11, v5\n\t"
        "v_ldexp_f32 v10, v4, v10\n\t"
        "v_mul_f32_e32 v12, v18, v12\n\t"
        "v_max_i32_dpp v11, v13, v13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp v12, v18, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_sub_u32_e32 v10, v13, v11\n\t"
        "v_ldexp_f32 v10, v2, v10\n\t"
        "v_sub_u32_dpp v18, v13, v11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_e32 v10, v12, v10\n\t"
        "v_add_u32_e32 v11, v11, v3\n\t"
        "v_ldexp_f32 v18, v2, v18\n\t"
        "v_fmac_f32_dpp v10, v12, v18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "ds_write_b128 v16, v[10:13]\n\t"
        "v_add_u32_e32 v16, 0xffffff80, v16\n\t"
        ::: CLOB);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main()
{
    unsigned long long *out, h;
    hipMalloc(&out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_no_nop), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_as_compiled), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int threads, lds, alive, prio, lanes, grid; } cfgs[] = {
        {64, 0, 1, 0, 64, 1}, {64, 0, 1, 0, 20, 1},
        {64, 0, 1, 2, 64, 1}, {64, 0, 1, 4, 64, 1}, {64, 0, 1, 6, 64, 1}};
    for (auto &c : cfgs)
        for (int v = 0; v < 2; ++v) {
            for (int r = 0; r < 3; ++r) {
                if (v == 0) hipLaunchKernelGGL(k_as_compiled, dim3(c.grid), dim3(c.threads), c.lds, 0, out, 64, c.lanes, c.alive, c.prio);
                else hipLaunchKernelGGL(k_no_nop, dim3(c.grid), dim3(c.threads), c.lds, 0, out, 64, c.lanes, c.alive, c.prio);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("%s threads %4d lds %6d alive %2d prio/data %d lanes %2d grid %3d: %.1f cycles per step\n", v ? "without s_nop" : "as compiled  ",
                   c.threads, c.lds, c.alive, c.prio, c.lanes, c.grid, (double)h / (64 * 16));
        }
    return 0;
}
