// placeholder until the binary kernel lands
#include "launch.hpp"
extern "C" int ctc_amd_binary_loss_grad(const float *, int64_t, int64_t, const float *, const int64_t *,
                                        const int64_t *, int, int, int, int, float, float, float *,
                                        float *, float *, void *, void *)
{
    return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
}
