// placeholder until the tracer kernels land (replaced in the next commit)
#include "iron_common.h"
extern "C" size_t iron_trace_workspace_bytes(int64_t, const iron_trace_params*) { return 0; }
extern "C" int iron_trace(const iron_net_t*, const iron_trace_params*, const float*, const float*, const float*,
                          const float*, const float*, const uint8_t*, int64_t, uint8_t*, float*, float*, float*,
                          iron_trace_stats*, void*, size_t, void*) { return IRON_ERR_UNSUPPORTED; }
extern "C" int iron_trace_phase(int32_t, const iron_net_t*, const iron_trace_params*, const float*, const float*,
                                const float*, const float*, const float*, const uint8_t*, const int64_t*, int64_t,
                                int32_t*, int64_t, uint8_t*, float*, float*, float*, iron_trace_stats*, void*, size_t,
                                void*) { return IRON_ERR_UNSUPPORTED; }
