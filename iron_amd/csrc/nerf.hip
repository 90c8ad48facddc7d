// NeRF background field of the stage-1 renderer (models/fields.py:243-327, use_viewdirs=True).
// k_nerf_h2: the default, on the split-fp16 h2 core (mlp_h2.h: four waves = 128 points share one weight stream through the LDS ring).
// k_nerf: the exact-fp32 MFMA core (mlp_core.h): one wave = 32 points, weights streamed per wave from L2 (IRON_MLP_CORE=f32, or a
// network the h2 stream is not built for).
#include "iron_common.h"
#include "mlp_core.h"
#include "h2_setup.h"

namespace iron {

// quads of head slots: 4-D points with LP levels -> 2 + 4 LP slots; view dirs with LV levels -> 2 + 3 LV slots
template <int LP, int LV>
struct NerfCfg {
    static constexpr int kQ4 = (2 + 4 * LP + 3) / 4;
    static constexpr int kQV = (head_slots(LV) + 3) / 4;
};

template <int LP, int LV>
__global__ __launch_bounds__(64, 1) void k_nerf(NerfNetDev net, const float* __restrict__ pts4, const float* __restrict__ views,
                                                int n, float* __restrict__ alpha, float* __restrict__ rgb) {
    using Cfg = NerfCfg<LP, LV>;
    constexpr int Q4 = Cfg::kQ4, QV = Cfg::kQV;
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    WStream ws;
    ws.init(net.blob, net.blob_bytes, lane);
    const int n_tiles = (n + kTile - 1) / kTile;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < n;
        float p[4] = {0.f, 0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 1.f};
        if (ok) {
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = pts4[4 * (size_t)li + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = views[3 * (size_t)li + c];
        }
        float head[4 * Q4];
#pragma unroll
        for (int i = 0; i < 4 * Q4; ++i) head[i] = 0.0f;
        head_fill4<LP>(p[0], p[1], p[2], p[3], half, head);

        // layer 0: PE(points) -> 256, relu
        f32x16 h[kHidTiles];
#pragma unroll
        for (int pr = 0; pr < kPairs; ++pr) {
            f32x16 a0 = load_half_tile(ws, net.bias, 2 * pr);
            f32x16 a1 = load_half_tile(ws, net.bias, 2 * pr + 1);
            dense_head_pair<Q4>(ws, net.w_head0, pr, head, a0, a1);
            h[2 * pr] = relu_tile(a0);
            h[2 * pr + 1] = relu_tile(a1);
        }
        // layers 1 .. D-1 (the input is concatenated again in front of layer skip_after + 1), one weight FIFO
        WQueue wq;
        wq.prime(ws, net.w_hid);
        int blk = 0;
        for (int l = 1; l < net.n_layers; ++l) {
            const uint32_t wb = net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u);
            const uint32_t bb = net.bias + (uint32_t)l * (kF4PerBiasLayer * 16u);
            f32x16 o[kHidTiles];
            hidden_layer<ReluAct, Q4>(ws, wb, bb, l == net.skip_after + 1 && net.skip_after >= 0, net.w_head_skip, head, wq, h, o, ReluAct());
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) h[t] = o[t];
        }
        // alpha = alpha_linear(h)
        const float a_out = row_dot(ws, net.w_alpha, h) + net.b_alpha;
        // feature = feature_linear(h) (no activation)
        {
            const uint32_t wb = net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u);
            const uint32_t bb = net.bias + (uint32_t)net.n_layers * (kF4PerBiasLayer * 16u);
            f32x16 o[kHidTiles];
            hidden_layer<IdentityAct, Q4>(ws, wb, bb, false, net.w_head0, head, wq, h, o, IdentityAct());
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) h[t] = o[t];
        }
        // views_linears[0]: [feature | PE(view)] -> 128, relu (tiles 0..3)
        float hv[4 * QV];
#pragma unroll
        for (int i = 0; i < 4 * QV; ++i) hv[i] = 0.0f;
        head_fill<LV>(v[0], v[1], v[2], half, hv);
        f32x16 o[kHidTiles];
        {
            const uint32_t wb = net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u);
            const uint32_t bb = net.bias + (uint32_t)(net.n_layers + 1) * (kF4PerBiasLayer * 16u);
            hidden_layer<ReluAct, QV, 2>(ws, wb, bb, true, net.w_head_view, hv, wq, h, o, ReluAct());
        }
        const float a_res = a_out;
        float c_out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) c_out[c] = row_dot_n<4>(ws, net.w_rgb + (uint32_t)c * (kF4PerBiasLayer * 16u), o) + net.b_rgb[c];
        if (ok && lane < 32) {
            if (alpha) alpha[li] = a_res;
            if (rgb) { rgb[3 * (size_t)li] = c_out[0]; rgb[3 * (size_t)li + 1] = c_out[1]; rgb[3 * (size_t)li + 2] = c_out[2]; }
        }
    }
}

// ---- h2 core ----------------------------------------------------------------------------------------------------------------
// Stream order and side blocks: pack_h2.hip build_h2_nerf (D = 8, skip after layer 4, PE-10 points, PE-4 views).
constexpr int kLdsNerfExt = kLdsH2Total;            // [views bias 1 KiB][row rgb 2 1 KiB]
constexpr int kLdsNerfTotal = kLdsH2Total + 2048;

__global__ __launch_bounds__(256, 1) void k_nerf_h2(H2StreamDev hs, NerfNetDev net, const float* __restrict__ pts4,
                                                    const float* __restrict__ views, int n, float* __restrict__ alpha, float* __restrict__ rgb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(hs.base + hs.rows_off + kLdsRowsBytes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds + kLdsNerfExt);
        for (int i = threadIdx.x; i < 2048 / 4; i += 256) dst[i] = src[i];
    }
    Ring ring;
    h2_setup(hs, lds, ring);   // (its barrier also publishes the extension block)
    const int n_tiles = (n + kTile - 1) / kTile;
    const int n_groups = (n_tiles + 3) / 4;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int tile = g * 4 + wave;
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < n;
        float p[4] = {0.f, 0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 1.f};
        if (ok) {
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = pts4[4 * (size_t)li + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = views[3 * (size_t)li + c];
        }
        HeadFrag hd, hd2;
        {
            float head[2 * kHeadSlots];
#pragma unroll
            for (int i = 0; i < 2 * kHeadSlots; ++i) head[i] = 0.0f;
            head_fill4<10>(p[0], p[1], p[2], p[3], half, head);
            split_head(head, hd);
            split_head(head + kHeadSlots, hd2);
        }
        TileFrag X[kHidTiles], Y[kHidTiles];
        f32x16 hf[kHidTiles];
        const char* bias = lds + kLdsBias;
        // layer 0: the two head slots only, relu
#pragma unroll
        for (int to = 0; to < kHidTiles; ++to) {
            f32x16 a_hi = zero16(), a_lo = zero16();
            ring.sync();
            const RingStep s0 = ring.step();
            step_head(s0.rd, bias, s0.wr, s0.src, s0.hidden, wave, lane, to, true, hd, a_hi, a_lo);
            ring.sync();
            const RingStep s1 = ring.step();
            step_head(s1.rd, bias, s1.wr, s1.src, s1.hidden, wave, lane, to, false, hd2, a_hi, a_lo);
            split_tile(relu_tile_nan(h2_combine(a_hi, a_lo)), X[to]);
        }
        f32x16 c_hi, c_lo;
        h2_hidden_layer<true, 0, false, 1, false, true>(ring, bias + 1 * 1024, hd, lane, X, Y, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 2 * 1024, hd, lane, Y, X, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 3 * 1024, hd, lane, X, Y, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 4 * 1024, hd, lane, Y, X, hf, c_hi, c_lo);
        h2_hidden_layer<true, 2, false, 1, true, true>(ring, bias + 5 * 1024, hd, lane, X, Y, hf, c_hi, c_lo, &hd2);   // input = [x | h]
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 6 * 1024, hd, lane, Y, X, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, true, 1, true, false>(ring, bias + 7 * 1024, hd, lane, X, Y, hf, c_hi, c_lo);
        // alpha = alpha_linear(h)
        const float a_out = row_dot_lds(lds + kLdsRows, hf, half) + net.b_alpha;
        // feature = feature_linear(h): no activation (neither DEFERs nor CARRYs: its neighbours are relu layers)
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) split_tile(hf[t], X[t]);
        h2_hidden_layer<true, 0, false, 2, false, false>(ring, bias + 8 * 1024, hd, lane, X, Y, hf, c_hi, c_lo);
        // views_linears[0]: [feature | PE(view)] -> 128, relu (output tiles 0..3)
        HeadFrag hv;
        {
            float head[kHeadSlots];
#pragma unroll
            for (int i = 0; i < kHeadSlots; ++i) head[i] = 0.0f;
            head_fill<4>(v[0], v[1], v[2], half, head);
            split_head(head, hv);
        }
        h2_hidden_layer<true, 1, true, 1, false, false, false, 4>(ring, lds + kLdsNerfExt, hv, lane, Y, X, hf, c_hi, c_lo);
#pragma unroll
        for (int t = 4; t < kHidTiles; ++t) hf[t] = zero16();
        float c_out[3];
        c_out[0] = row_dot_lds(lds + kLdsRows + 1024, hf, half) + net.b_rgb[0];
        c_out[1] = row_dot_lds(lds + kLdsRows + 2048, hf, half) + net.b_rgb[1];
        c_out[2] = row_dot_lds(lds + kLdsNerfExt + 1024, hf, half) + net.b_rgb[2];
        if (ok && lane < 32) {
            if (alpha) alpha[li] = a_out;
            if (rgb) { rgb[3 * (size_t)li] = c_out[0]; rgb[3 * (size_t)li + 1] = c_out[1]; rgb[3 * (size_t)li + 2] = c_out[2]; }
        }
    }
    ring.drain();
}

}  // namespace iron

using namespace iron;

extern "C" int iron_nerf_forward(const iron_net_t* nerf, const float* pts4, const float* view_dirs, int64_t n, float* alpha, float* rgb,
                                 void* stream) {
    if (!nerf || nerf->desc.kind != IRON_NET_NERF || n < 0) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!pts4 || !view_dirs || (!alpha && !rgb)) return IRON_ERR_BAD_ARG;
    const NerfNetDev& r = nerf->nerf;
    const int64_t tiles = (n + kTile - 1) / kTile;
    const int cus = cu_budget();
    const int64_t waves = (int64_t)cus * 4;
    const unsigned grid = (unsigned)(tiles < waves ? tiles : waves);
    hipStream_t st = (hipStream_t)stream;
    { const int rce = envelope_begin(nerf); if (rce != IRON_OK) return rce; }
    if (h2_enabled(nerf) && r.levels == 10 && r.levels_view == 4 && r.n_layers == 8 && r.skip_after == 4) {
        static bool attr = false;
        if (!attr) {
            IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_nerf_h2, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsNerfTotal));
            attr = true;
        }
        const int64_t groups = (tiles + 3) / 4;
        hipLaunchKernelGGL(k_nerf_h2, dim3((unsigned)(groups < cus ? groups : cus)), dim3(256), kLdsNerfTotal, st, nerf->h2_trace, r, pts4,
                           view_dirs, (int)n, alpha, rgb);
        IRON_HIP_TRY(hipGetLastError());
        envelope_scan(nerf, alpha, n, nullptr, 1, st);
        envelope_scan(nerf, rgb, n, nullptr, 3, st);
        return IRON_OK;
    }
    if (r.levels == 10 && r.levels_view == 4) {  // confs/womask_iron.conf: model.nerf
        hipLaunchKernelGGL((k_nerf<10, 4>), dim3(grid), dim3(64), 0, st, r, pts4, view_dirs, (int)n, alpha, rgb);
    } else {
        return IRON_ERR_UNSUPPORTED;
    }
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}
