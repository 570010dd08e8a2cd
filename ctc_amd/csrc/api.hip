// C-ABI entry points that are not tied to one kernel file (include/ctc_amd.h).
#include "common.hpp"
#include "launch.hpp"

namespace ctc {
size_t noblank_extra_workspace(int T, int B, int C, int S);   // noblank.hip
int blank_sync_ints(int T, int B);                              // blank.hip
}

extern "C" int ctc_amd_abi_version(void) { return CTC_AMD_ABI_VERSION; }

extern "C" const char *ctc_amd_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case CTC_AMD_ERR_BAD_ARGUMENT: return "ctc_amd: bad argument (null pointer or non-positive size)";
        case CTC_AMD_ERR_UNSUPPORTED_SHAPE: return "ctc_amd: shape not supported by the gfx950 kernels";
        case CTC_AMD_ERR_CODE_OVERFLOW:
            return "ctc_amd: more than 64 classes in the reference's int32 row codes (the reference raises OverflowError "
                   "at 2**64): pass exact_rows";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "ctc_amd: unknown error";
    }
}

extern "C" size_t ctc_amd_workspace_bytes(int variant, int T, int B, int C, int S)
{
    // [0,256): header -- arrival counter / packed batch sum of the in-launch reduction, status word, diagnostics
    size_t bytes = 256;
    if (variant == CTC_AMD_BLANK) {
        // emissions, alpha, beta lattices [B][T][NSP] fp32 + three [B][NSP] int state tables + [B] int2 lengths
        // + the hand-off counters of the fused schedule; NSP = 2S+1 padded to 64*K states (K = 2, 4, 8
        // states per lane), see blank.hip
        const size_t ns = 2 * (size_t)S + 1;
        const size_t nsp = ns <= 128 ? 128 : (ns <= 256 ? 256 : 512);
        bytes += 3 * (size_t)B * (size_t)T * nsp * sizeof(float) + 3 * (size_t)B * nsp * sizeof(int) +
                 (2 * (((size_t)B + 63) & ~(size_t)63) + (size_t)ctc::blank_sync_ints(T, B)) * sizeof(int);
    }
    // no-blank / binary: the list of samples whose nll does not fit the packed batch sum (common.hpp)
    if (variant == CTC_AMD_NOBLANK || variant == CTC_AMD_BINARY) bytes += ctc::acc_list_bytes(B);
    if (variant == CTC_AMD_NOBLANK) bytes += ctc::noblank_extra_workspace(T, B, C, S);   // 0 while T x S fits in LDS
    return bytes;
}

extern "C" int ctc_amd_workspace_status(void *workspace, int clear, void *stream, unsigned *status_host)
{
    if (!workspace || !status_host) return CTC_AMD_ERR_BAD_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned *word = static_cast<unsigned *>(workspace) + 2;
    hipError_t e = hipMemcpyAsync(status_host, word, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && clear && *status_host != 0) e = hipMemsetAsync(word, 0, sizeof(unsigned), s);
    return (int)e;
}

// ---- collective gate (include/ctc_amd.h) ---------------------------------------------
namespace ctc {
__global__ __launch_bounds__(64) void collective_gate_kernel(unsigned *counter, unsigned need, unsigned long long ticks)
{
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(counter + kGateWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // loss launches count from now on
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    for (;;) {
        unsigned v = lane < 16 ? __hip_atomic_load(arrival_shard(counter, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) v += __shfl_xor(v, sh, 64);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= need || __builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
}  // namespace ctc

extern "C" int ctc_amd_collective_gate(void *workspace, int B, int timeout_us, void *stream)
{
    if (!workspace || B < 1 || timeout_us < 0) return CTC_AMD_ERR_BAD_ARGUMENT;
    const int cus = ctc::device_cus();
    const unsigned need = (unsigned)(cus > 0 && B > cus ? cus : B);       // one round of workgroups fills the chip
    hipLaunchKernelGGL(ctc::collective_gate_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<unsigned *>(workspace), need, (unsigned long long)timeout_us * 100ull);
    return (int)hipGetLastError();
}
