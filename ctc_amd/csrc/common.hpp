// Shared device helpers for the gfx950 CTC kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ctc {

constexpr int kWave = 64;
// zero_padding sentinel of the reference (NoBlankCTC.py:25): keeping it (instead of
// -inf) keeps every intermediate finite and the float32 roundings identical.
constexpr float kNeg = -10000000000000.0f;
constexpr float kInfeasible = 1.0e12f;   // nll above this <=> no alignment exists

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// lane i <- lane i-1 (lane 0 keeps `fill`): DPP wave_shr:1, one VALU op, no LDS.
__device__ __forceinline__ float wave_shr1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// lane i <- lane i+1 (lane 63 keeps `fill`): DPP wave_shl:1.
__device__ __forceinline__ float wave_shl1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// _logsumexp over two values (NoBlankCTC.py:16-19): max + log(exp(a-max)+exp(b-max))
// == max + log(1 + exp(min-max)); v_exp_f32 / v_log_f32 based.
__device__ __forceinline__ float lse2(float a, float b)
{
    float m = fmaxf(a, b);
    float d = fminf(a, b) - m;
    return m + __logf(1.0f + __expf(d));
}

__device__ __forceinline__ int64_t load_label(const void *p, int is64, int64_t i)
{
    return is64 ? static_cast<const int64_t *>(p)[i] : (int64_t) static_cast<const int32_t *>(p)[i];
}

// Deterministic batch reduction by the LAST workgroup to finish its nll (in-launch
// arrival ticket).  Caller: exactly one wave per workgroup, after lane 0 has the
// sample's value.  nll is stored write-through (sc1), drained, then the ticket is
// drawn; the workgroup drawing B-1 reads every nll with sc1 loads in a fixed order.
template <typename F>
__device__ __forceinline__ void publish_and_reduce(float value, int b, int B, float *nll,
                                                   float *loss, float loss_scale,
                                                   unsigned *counter, F per_sample)
{
    int last = 0;
    if (lane_id() == 0) {
        __hip_atomic_store(&nll[b], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (ticket == (unsigned)(B - 1));
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    float s = 0.f;
    for (int i = lane_id(); i < B; i += kWave)
        s += per_sample(__hip_atomic_load(&nll[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), i);
    s = wave_sum(s);
    if (lane_id() == 0) {
        loss[0] = s * loss_scale;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace ctc
