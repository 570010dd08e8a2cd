// Host-side launch helper for the C-ABI entry points (no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/ctc_amd.h"

namespace ctc {

constexpr size_t kMaxLds = 160 * 1024;        // LDS per CU on gfx950
constexpr size_t kDefaultDynLds = 64 * 1024;  // above this the attribute must be raised

template <auto kern, typename... Args>
inline int launch(dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args)
{
    if (smem > kDefaultDynLds) {
        // one high-water mark per kernel instantiation; the attribute is sticky
        static std::atomic<size_t> granted{0};
        if (granted.load(std::memory_order_relaxed) < smem) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
            if (e != hipSuccess) return (int)e;
            granted.store(kMaxLds, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, args...);
    return (int)hipGetLastError();
}

}  // namespace ctc
