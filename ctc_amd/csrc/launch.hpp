// Host-side launch helper for the C-ABI entry points (no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "../../include/ctc_amd.h"

namespace ctc {

constexpr size_t kMaxLds = 160 * 1024;        // LDS per CU on gfx950
constexpr size_t kDefaultDynLds = 64 * 1024;  // above this the attribute must be raised
constexpr int kMaxDevices = 64;               // per-device caches below are indexed by the HIP device ordinal

// ordinal of the calling thread's current device, or -1 (then nothing is cached)
inline int current_device()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}

// compute units of the current device (cached per device: a process may drive several GPUs)
inline int device_cus()
{
    static std::atomic<int> cus[kMaxDevices];
    const int dev = current_device();
    if (dev < 0) return 0;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Diagnostic switches (phase stamps, forced kernel choices) exist only in builds made with
// -DCTC_AMD_DIAGNOSTICS (python -m ctc_amd.build --diag -> libctc_amd_diag.so, used by tools/); the
// product library reads none of them.
inline int diag_env(const char *name, int fallback = 0)
{
#ifdef CTC_AMD_DIAGNOSTICS
    const char *v = getenv(name);
    return v ? (v[0] ? atoi(v) : 1) : fallback;
#else
    (void)name;
    return fallback;
#endif
}

template <auto kern, typename... Args>
inline int launch(dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args)
{
    if (smem > kDefaultDynLds) {
        // the attribute is sticky per kernel AND per device: one flag per (instantiation, device)
        static std::atomic<bool> granted[kMaxDevices];
        const int dev = current_device();
        if (dev < 0 || !granted[dev].load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0) granted[dev].store(true, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, args...);
    return (int)hipGetLastError();
}

}  // namespace ctc
