// placeholder until the best-path kernel lands
#include "launch.hpp"
extern "C" int ctc_amd_noblank_best_path(const float *, int64_t, int64_t, const void *, int, const int64_t *,
                                         const int64_t *, int, int, int, int, int32_t *, float *, void *, void *)
{
    return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
}
