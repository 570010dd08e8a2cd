// C-ABI entry points that are not tied to one kernel file (include/ctc_amd.h).
#include "launch.hpp"

extern "C" int ctc_amd_abi_version(void) { return CTC_AMD_ABI_VERSION; }

extern "C" const char *ctc_amd_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case CTC_AMD_ERR_BAD_ARGUMENT: return "ctc_amd: bad argument (null pointer or non-positive size)";
        case CTC_AMD_ERR_UNSUPPORTED_SHAPE: return "ctc_amd: shape not supported by the gfx950 kernels";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "ctc_amd: unknown error";
    }
}

extern "C" size_t ctc_amd_workspace_bytes(int variant, int T, int B, int C, int S)
{
    // [0,256): arrival counter of the in-launch batch reduction (+ padding)
    size_t bytes = 256;
    if (variant == CTC_AMD_BLANK) {
        // alpha and beta lattices [B][T][2S+1] fp32
        bytes += 2 * (size_t)B * (size_t)T * (size_t)(2 * S + 1) * sizeof(float);
    }
    (void)C;
    return bytes;
}
