// Kernels on the h2 (split-fp16, LDS-ring) MLP core.
#include <stdlib.h>
#include "mlp_h2.h"
#include "h2_setup.h"

namespace iron {

#ifndef IRON_FAST_SOFTPLUS
#define IRON_FAST_SOFTPLUS 1
#endif
constexpr bool kFastActH = IRON_FAST_SOFTPLUS != 0;

#if IRON_H2_STAMP
__device__ unsigned long long g_h2_stamps[4 * kStampSteps * 8];
#endif

// x [n,3] -> out[n]: 4 waves x 32 points per pass
__global__ __launch_bounds__(256, 1) void k_sdf_values_h2(H2StreamDev s, H2Meta m, const float* __restrict__ x, int64_t n,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring ring;
    h2_setup(s, lds, ring);
    const int64_t n_groups = (n + 127) / 128;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t idx = g * 128 + wave * 32 + (lane & 31);
        const bool ok = idx < n;
        const int64_t src = ok ? idx : (n - 1);
        const float px = x[src * 3 + 0], py = x[src * 3 + 1], pz = x[src * 3 + 2];
        f32x16 hf[kHidTiles];
        sdf_hidden_stack_h2<kFastActH>(ring, lds, m.n_hidden_layers, m.skip_layer, m.scale, px, py, pz, lane, hf);
        const float v = (row_dot_lds(lds + kLdsRows, hf, lane >> 5) + m.b_last) / m.scale;
        if (ok && lane < 32) out[idx] = v;
    }
    ring.drain();
#if IRON_H2_STAMP
    if (blockIdx.x == 0) {
        const unsigned long long* st = reinterpret_cast<const unsigned long long*>(lds + kLdsStamp);
        for (int i = threadIdx.x; i < 4 * kStampSteps * 8; i += 256) g_h2_stamps[i] = st[i];
    }
#endif
}

#if IRON_H2_STAMP
extern "C" int iron_debug_h2_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_stamps), sizeof(unsigned long long) * 4 * kStampSteps * 8) == hipSuccess ? 0 : -1;
}
#endif

bool use_h2_core() {
    static int v = -1;
    if (v < 0) {
        // default: the split-fp16 "h2" core; IRON_MLP_CORE=f32 selects the exact-fp32 MFMA core (mlp_core.h)
        const char* e = getenv("IRON_MLP_CORE");
        v = (e && e[0] == 'f') ? 0 : 1;
    }
    return v == 1;
}

int launch_sdf_values_h2(const iron_net* net, const float* x, int64_t n, float* out, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_sdf_values_h2, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total + (IRON_H2_STAMP ? kLdsStampBytes : 0)));
        attr = true;
    }
    H2Meta m;
    m.n_hidden_layers = net->sdf.n_hidden_layers; m.skip_layer = net->sdf.skip_layer; m.scale = net->sdf.scale; m.b_last = net->sdf.b_last;
    const int64_t groups = (n + 127) / 128;
    const int64_t cus = cu_budget();
    const unsigned grid = (unsigned)(groups < cus ? groups : cus);
    hipLaunchKernelGGL(k_sdf_values_h2, dim3(grid), dim3(256), kLdsH2Total + (IRON_H2_STAMP ? kLdsStampBytes : 0), st, net->h2_trace, m, x, n, out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

}  // namespace iron
