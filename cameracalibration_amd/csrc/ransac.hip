// ransac.hip -- placeholder until the batched RANSAC scorer lands.
#include "ccal_internal.hpp"

extern "C" int ccal_ransac_plane(ccal_ctx* ctx, int, const double*, int, float, uint64_t, float*,
                                 int64_t*)
{
    if (!ctx) return CCAL_ERR_ARG;
    ctx->err = "ransac_plane: not implemented";
    return CCAL_ERR_STATE;
}
