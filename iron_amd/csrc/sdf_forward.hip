// Batched SDF queries: SDFNetwork.forward / .sdf (models/fields.py:82-104).
// This is the pure "8x256 SDF MLP" kernel the MFMA roofline sub-target is measured on.
#include "mlp_core.h"
#include "h2_setup.h"

namespace iron {

#ifndef IRON_FAST_SOFTPLUS
#define IRON_FAST_SOFTPLUS 0
#endif
constexpr bool kFastAct = IRON_FAST_SOFTPLUS != 0;

// x [n,3] -> out[n] (sdf only).  One wave per 32-point tile, grid-stride over tiles.
__global__ __launch_bounds__(64, 1) void k_sdf_values(SdfNetDev net, const float* __restrict__ x, int64_t n,
                                                     float* __restrict__ out, int out_stride) {
    const int lane = threadIdx.x;
    WStream ws;
    ws.init(net.blob, net.blob_bytes, lane);
    const int64_t n_tiles = (n + kTile - 1) / kTile;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t idx = tile * kTile + (lane & 31);
        const bool ok = idx < n;
        const int64_t src = ok ? idx : (n - 1);
        const float px = x[src * 3 + 0], py = x[src * 3 + 1], pz = x[src * 3 + 2];
        const float s = sdf_eval<kFastAct>(net, ws, px, py, pz, lane);
        if (ok && lane < 32) out[idx * out_stride] = s;
    }
}

// x [n,3] -> out[n, 257]: sdf + 256 feature columns (last linear layer, rows 1..256, no activation)
__global__ __launch_bounds__(64, 1) void k_sdf_full(SdfNetDev net, const float* __restrict__ x, int64_t n,
                                                   float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    WStream ws;
    ws.init(net.blob, net.blob_bytes, lane);
    const int64_t n_tiles = (n + kTile - 1) / kTile;
    const int ld = kHidden + 1;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t idx = tile * kTile + (lane & 31);
        const bool ok = idx < n;
        const int64_t src = ok ? idx : (n - 1);
        const float px = x[src * 3 + 0], py = x[src * 3 + 1], pz = x[src * 3 + 2];
        f32x16 h[kHidTiles];
        sdf_hidden_stack<kFastAct>(net, ws, px, py, pz, lane, h);
        const float s = (row_dot(ws, net.w_last, h) + net.b_last) / net.scale;
        if (ok && lane < 32) out[idx * ld] = s;
        f32x16 o[kHidTiles];
        WQueue wq;
        wq.prime(ws, net.w_feat);
        hidden_layer<IdentityAct, 1>(ws, net.w_feat, net.b_feat, false, 0u, nullptr, wq, h, o, IdentityAct());
        if (ok) {
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[idx * ld + 1 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half] = o[t][r];
        }
    }
}

}  // namespace iron

namespace iron {
int launch_sdf_values_h2(const iron_net* net, const float* x, int64_t n, float* out, hipStream_t st);
int launch_sdf_values_w16(const iron_net* net, const float* x, int64_t n, float* out, hipStream_t st);
bool use_w16_core();
}

using namespace iron;

static int grid_for_tiles(int64_t n_tiles) {
    // single-wave workgroups, one wave per SIMD: 256 CUs x 4 resident waves; x2 to smooth the tail
    const int64_t cap = 256 * 4 * 2;
    return (int)(n_tiles < cap ? n_tiles : cap);
}

extern "C" int iron_sdf_forward(const iron_net_t* net, const float* x, int64_t n, float* out, int32_t out_cols,
                                void* stream) {
    if (!net || net->desc.kind != IRON_NET_SDF || n < 0 || (n > 0 && (!x || !out))) return IRON_ERR_BAD_ARG;
    if (out_cols != 1 && out_cols != net->desc.d_out) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_tiles = (n + kTile - 1) / kTile;
    if (((uintptr_t)x & 3) || ((uintptr_t)out & 3)) return IRON_ERR_BAD_ARG;
    { const int rce = envelope_begin(net); if (rce != IRON_OK) return rce; }
    ProfScope ps(IRON_PROF_SDF_FORWARD, st);
    if (out_cols == 1 && use_w16_core() && net->w16_blob && !net->h2_disabled) return launch_sdf_values_w16(net, x, n, out, st);
    if (out_cols == 1 && h2_sdf_usable(net)) {
        const int rc = launch_sdf_values_h2(net, x, n, out, st);
        envelope_scan(net, out, n, nullptr, 1, st);
        return rc;
    }
    if (out_cols == 1) {
        hipLaunchKernelGGL(k_sdf_values, dim3(grid_for_tiles(n_tiles)), dim3(64), 0, st, net->sdf, x, n, out, 1);
    } else {
        if (!net->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_sdf_full, dim3(grid_for_tiles(n_tiles)), dim3(64), 0, st, net->sdf, x, n, out);
    }
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}
