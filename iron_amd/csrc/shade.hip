// placeholder until the shading kernels land (replaced in the next commit)
#include "iron_common.h"
extern "C" int iron_sdf_get_all(const iron_net_t*, const float*, int64_t, float*, float*, float*, void*) { return IRON_ERR_UNSUPPORTED; }
extern "C" int iron_render_forward(const iron_net_t*, const float*, const float*, const float*, const float*, int64_t,
                                   float*, void*) { return IRON_ERR_UNSUPPORTED; }
extern "C" size_t iron_shade_workspace_bytes(int64_t) { return 0; }
extern "C" int iron_shade_ggx(const iron_shade_nets*, float, int32_t, const float*, const float*, const float*,
                              const float*, const float*, const uint8_t*, int64_t, const iron_shade_out*, void*,
                              size_t, void*) { return IRON_ERR_UNSUPPORTED; }
