// Shading: render_normal_and_color (models/raytracer.py:593-662) with the driver's GGX render_fn
// (render_surface.py:117-156), plus the standalone SDFNetwork.get_all and RenderingNetwork.forward.
//
//   k_compact     convergent rays -> hit list (ballot + one atomic per wave)
//   k_sdf_grad    get_all (models/fields.py:120-137): a 4-wave workgroup owns 32 hit points.  Wave 0
//                 runs the value pass; waves 1..3 run the FORWARD-MODE tangent of the same network
//                 for d/dx, d/dy, d/dz (replacing autograd.grad): identical MFMA stream without bias,
//                 with the layer's softplus derivative sigma(100 z) handed over by wave 0 through LDS
//                 (double-buffered, one barrier per layer).  Features leave in the register-tile
//                 layout so the material kernels reload them as B operands without a transpose.
//   k_material    RenderingNetwork.forward (models/fields.py:203-239): one wave per 32 hits and net.
//   k_ggx_shade   normalise, get_materials post-ops (models/rendering_func.py:5-16), GGX, scatter.
#include "mlp_h2.h"
#include "h2_setup.h"
#include "ggx_core.h"
#include "shade_args.h"

namespace iron {

#ifndef IRON_FAST_SOFTPLUS
#define IRON_FAST_SOFTPLUS 1
#endif
constexpr bool kFastActS = IRON_FAST_SOFTPLUS != 0;

// softplus(beta=100) and its derivative from one exponential.  torch backward (softplus_backward):
// grad * e/(e+1) with e = exp(100 z) below the threshold, grad above it.
template <bool FAST>
__device__ __forceinline__ void softplus100_both(float z, float& h, float& s) {
    if constexpr (FAST) {
        // u = exp(-|100 z|) never overflows: softplus = max(z, 0) + log2(1 + u) ln2 / 100 (mlp_core.h), and
        // sigmoid(100 z) = (z >= 0 ? 1 : u) / (1 + u) (rounds to exactly 1 above the reference's threshold 100 z = 20)
        const float u = __builtin_amdgcn_exp2f(__builtin_fabsf(z) * -144.26950408889634f);
        const float w = 1.0f + u;
        h = __builtin_fmaf(__builtin_amdgcn_logf(w), 0.0069314718055994531f, fmaxf(z, 0.0f));
        s = (z >= 0.0f ? 1.0f : u) * __builtin_amdgcn_rcpf(w);
    } else {
        const float t = z * 100.0f;
        const float e = expf(t);
        const float l = log1pf(e) / 100.0f;
        const float sg = e / (e + 1.0f);
        const bool lin = t > 20.0f;
        h = lin ? z : l;
        s = lin ? 1.0f : sg;
    }
}

// d(head slots)/d(component c) for one vec3 source with LEVELS (the tangent of head_fill)
template <int LEVELS>
__device__ __forceinline__ void head_fill_tangent(float vx, float vy, float vz, int c, int half, float* slots) {
    slots[0] = (half ? (c == 1) : (c == 0)) ? 1.0f : 0.0f;
    slots[1] = (!half && c == 2) ? 1.0f : 0.0f;
    const float v = c == 0 ? vx : (c == 1 ? vy : vz);
#pragma unroll
    for (int k = 0; k < LEVELS; ++k) {
        const float f = (float)(1 << k);
        float s, co;
        sincosf(v * f, &s, &co);
        const float d = half ? -(f * s) : (f * co);  // d sin = f cos, d cos = -f sin
        slots[2 + 3 * k + 0] = c == 0 ? d : 0.0f;
        slots[2 + 3 * k + 1] = c == 1 ? d : 0.0f;
        slots[2 + 3 * k + 2] = c == 2 ? d : 0.0f;
    }
}

__device__ __forceinline__ f32x16 zero_tile() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    return v;
}

constexpr int kSBufFloats = kHidTiles * 16 * 64;  // one layer's sigma'(z) for 32 points: 32 KiB

__global__ __launch_bounds__(256, 1) void k_sdf_grad(SdfNetDev net, GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x kSBufFloats
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_value = wave == 0;
    const int axis = wave - 1;
    WStream ws;
    ws.init(net.blob, net.blob_bytes, lane);
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int src = ok ? (a.list ? a.list[li] : li) : 0;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (ok) { px = a.x[3 * (size_t)src]; py = a.x[3 * (size_t)src + 1]; pz = a.x[3 * (size_t)src + 2]; }
        const float sx = px * net.scale, sy = py * net.scale, sz = pz * net.scale;

        float head[4 * kSdfHeadQuads];
        if (is_value) head_fill<kSdfPeLevels>(sx, sy, sz, half, head);
        else head_fill_tangent<kSdfPeLevels>(sx, sy, sz, axis, half, head);

        f32x16 h[kHidTiles];
        WQueue wq;
        wq.prime(ws, net.w_hid);
        for (int l = 0; l < net.n_hidden_layers; ++l) {
            float* sbuf = lds + (l & 1) * kSBufFloats;
            const uint32_t wb = net.w_hid + (uint32_t)(l > 0 ? l - 1 : 0) * (kF4PerHidLayer * 16u);
            const uint32_t bb = net.bias + (uint32_t)l * (kF4PerBiasLayer * 16u);
            const bool with_head = (l == 0) || (l == net.skip_layer);
            const uint32_t hb = (l == 0) ? net.w_pe0 : net.w_pe_skip;
            f32x16 o[kHidTiles];
#define IRON_GPAIR(P)                                                                               \
    {                                                                                               \
        f32x16 a0 = is_value ? load_half_tile(ws, bb, 2 * P) : zero_tile();                         \
        f32x16 a1 = is_value ? load_half_tile(ws, bb, 2 * P + 1) : zero_tile();                     \
        if (with_head) dense_head_pair<kSdfHeadQuads>(ws, hb, P, head, a0, a1);                     \
        if (l > 0) dense_hidden_pair<P>(ws, wb, wq, h, a0, a1);                                     \
        if (is_value) {                                                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                        \
                float hv, sv;                                                                       \
                softplus100_both<kFastActS>(a0[r], hv, sv);                                         \
                a0[r] = hv;                                                                         \
                sbuf[((2 * P) * 16 + r) * 64 + lane] = sv;                                          \
                softplus100_both<kFastActS>(a1[r], hv, sv);                                         \
                a1[r] = hv;                                                                         \
                sbuf[((2 * P + 1) * 16 + r) * 64 + lane] = sv;                                      \
            }                                                                                       \
        }                                                                                           \
        o[2 * P] = a0;                                                                              \
        o[2 * P + 1] = a1;                                                                          \
    }
            IRON_GPAIR(0) IRON_GPAIR(1) IRON_GPAIR(2) IRON_GPAIR(3)
#undef IRON_GPAIR
            __syncthreads();
            if (!is_value) {
#pragma unroll
                for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= sbuf[(t * 16 + r) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) h[t] = o[t];
        }

        if (is_value) {
            const float s = (row_dot(ws, net.w_last, h) + net.b_last) / net.scale;
            if (ok && lane < 32 && a.sdf_out) a.sdf_out[li] = s;
            if ((a.feat_packed || a.feat_rows) && net.w_feat) {
                f32x16 o[kHidTiles];
                WQueue wf;
                wf.prime(ws, net.w_feat);
                hidden_layer<IdentityAct, 1>(ws, net.w_feat, net.b_feat, false, 0u, nullptr, wf, h, o, IdentityAct());
                if (a.feat_packed) {
                    float* dst = a.feat_packed + (size_t)tile * kSBufFloats;
#pragma unroll
                    for (int t = 0; t < kHidTiles; ++t) feat_store_tile(dst, t, lane, o[t]);
                }
                if (a.feat_rows && ok) {
#pragma unroll
                    for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            a.feat_rows[(size_t)li * kHidden + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half] = o[t][r];
                }
            }
        } else {
            const float g = row_dot(ws, net.w_last, h);  // d out0 / d x_c  (scale cancels: (1/s) * d/d(s x) * s)
            if (ok && lane < 32 && a.grad_out) a.grad_out[3 * (size_t)li + axis] = g;
        }
        __syncthreads();  // the next tile's layer 0 reuses sbuf[0]
    }
}

// ---- get_all on the h2 core ---------------------------------------------------------------------------
// Same roles as k_sdf_grad (wave 0 = value, waves 1..3 = forward-mode tangents d/dx, d/dy, d/dz of the same
// 32 points), but all four waves walk ONE weight stream through the LDS ring (mlp_h2.h) in lock step.  Per output
// tile: MFMAs (value adds the bias) -> the value wave's epilogue writes sigma'(z) of the tile to a 4 KiB LDS
// buffer -> barrier -> the tangent waves scale their tile by it.  The feature rows of the last layer are 8 more
// ring slots that only the value wave multiplies.
constexpr int kLdsSbuf = kLdsH2Total;               // 2 x [16 regs][64 lanes] f32 = 2 x 4 KiB (double buffer, tile parity)
constexpr int kLdsGradTotal = kLdsH2Total + 8192;

#if defined(IRON_GRAD_VARIANT) && (IRON_GRAD_VARIANT & 1)   // timing experiment (garbage results): no activation in the value wave
#define IRON_GRAD_ACT(Z, H, S) { H = (Z); S = 1.0f; }
#else
#define IRON_GRAD_ACT(Z, H, S) softplus100_both<kFastActS>(Z, H, S)
#endif

__device__ __forceinline__ void lds_publish_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// One evaluation of value (wave 0) or tangent d/d axis (waves 1-3) of the SDF network at this lane's point on the h2
// ring (the trace stream: 72 slots).  Returns the sdf (value wave) or the gradient component (tangent waves); `out` is
// left holding the last hidden layer as f32 bit patterns (for the feature rows).  All four waves call it together.
__device__ __forceinline__ float sdf_value_or_tangent_h2(Ring& ring, char* lds, float* sbuf, const H2Meta& m, float px, float py,
                                                         float pz, bool is_value, int axis, int wave, int lane,
                                                         TileFrag (&in)[kHidTiles], TileFrag (&out)[kHidTiles]) {
    const int half = lane >> 5;
    const float sx = px * m.scale, sy = py * m.scale, sz = pz * m.scale;
    float head[kHeadSlots];
#pragma unroll
    for (int i = 0; i < kHeadSlots; ++i) head[i] = 0.0f;
    if (is_value) head_fill<kSdfPeLevels>(sx, sy, sz, half, head);
    else head_fill_tangent<kSdfPeLevels>(sx, sy, sz, axis, half, head);
    HeadFrag hd;
    split_head(head, hd);
    TileFrag dummy_out;
    f32x16 dummy_hf;
    f32x16 p_hi = zero16(), p_lo = zero16();   // tangent waves: the tile whose sigma' is still on its way
    for (int l = 0; l < m.n_hidden_layers; ++l) {
        const bool last = (l == m.n_hidden_layers - 1);
        const bool with_head = (l == 0) || (l == m.skip_layer);
        const char* bias = lds + kLdsBias + l * 1024;
        if (l > 0) {
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) in[t] = out[t];
        }
#define IRON_GTILE(TO)                                                                                                  \
    {                                                                                                                   \
        f32x16 a_hi = zero16(), a_lo = zero16();                                                                        \
        if (with_head) {                                                                                                \
            ring.sync();                                                                                                \
            const RingStep sh = ring.step();                                                                            \
            step_head(sh.rd, bias, sh.wr, sh.src, sh.hidden, wave, lane, TO, false, hd, a_hi, a_lo);                    \
        }                                                                                                               \
        if (l > 0) {                                                                                                    \
            ring.sync();                                                                                                \
            const RingStep st = ring.step();                                                                            \
            step_hidden<kFastActS, 0>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, false, in, a_hi, a_lo,     \
                                      a_hi, a_lo, dummy_out, dummy_hf);                                                 \
        }                                                                                                               \
        if (is_value) {                                                                                                 \
            /* value wave: activation of tile TO now; sigma'(z) goes to sbuf[TO & 1] for the tangent waves, which pick */ \
            /* it up one ring step later (behind that step's barrier)                                                  */ \
            f32x16 zt = h2_combine(a_hi, a_lo);                                                                         \
            const f32x16 bt = lds_half_tile(bias, TO, half);                                                            \
            float* sb = sbuf + ((TO) & 1) * (16 * 64);                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
                float hv, sv;                                                                                           \
                IRON_GRAD_ACT(zt[r] + bt[r], hv, sv);                                                                   \
                zt[r] = hv;                                                                                             \
                sb[r * 64 + lane] = sv;                                                                                 \
            }                                                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
            if (last) out[TO] = __builtin_bit_cast(TileFrag, zt);                                                       \
            else split_tile(zt, out[TO]);                                                                               \
        } else {                                                                                                        \
            /* tangent waves: finish tile TO - 1 (its sigma' was published during the previous step), keep TO pending  */ \
            if ((TO) > 0) IRON_GTANGENT_FINISH((TO) - 1)                                                                \
            p_hi = a_hi;                                                                                                \
            p_lo = a_lo;                                                                                                \
        }                                                                                                               \
    }
/* the last hidden layer's f32 tile travels in the SAME 16 registers its split fragments would use (hf materialises after the */
/* loop): a separate hf[8] live through the runtime layer loop costs 128 VGPRs -> 100+ spilled                                   */
#define IRON_GTANGENT_FINISH(T)                                                                                         \
    {                                                                                                                   \
        f32x16 zt = h2_combine(p_hi, p_lo);                                                                             \
        const float* sb = sbuf + ((T) & 1) * (16 * 64);                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) zt[r] *= sb[r * 64 + lane];                                      \
        if (last) out[T] = __builtin_bit_cast(TileFrag, zt);                                                            \
        else split_tile(zt, out[T]);                                                                                    \
    }
        IRON_GTILE(0) IRON_GTILE(1) IRON_GTILE(2) IRON_GTILE(3) IRON_GTILE(4) IRON_GTILE(5) IRON_GTILE(6) IRON_GTILE(7)
        lds_publish_barrier();   // sigma' of tile 7 is out; (also orders this layer's last sbuf reads before the next layer's writes)
        if (!is_value) IRON_GTANGENT_FINISH(7)
#undef IRON_GTANGENT_FINISH
#undef IRON_GTILE
    }
    f32x16 hf[kHidTiles];  // last hidden layer in f32 (value: h7, tangent: d h7)
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) hf[t] = __builtin_bit_cast(f32x16, out[t]);
    const float d = row_dot_lds(lds + kLdsRows, hf, half);
    return is_value ? (d + m.b_last) / m.scale : d;
}

__global__ __launch_bounds__(256, 1) void k_sdf_grad_h2(H2StreamDev hs, H2Meta m, GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_value = wave == 0;
    const int axis = wave - 1;
    Ring ring;
    h2_setup(hs, lds, ring);
    float* sbuf = reinterpret_cast<float*>(lds + kLdsSbuf);
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;
    const bool want_feat = (a.feat_packed != nullptr) || (a.feat_rows != nullptr);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int src = ok ? (a.list ? a.list[li] : li) : 0;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (ok) { px = a.x[3 * (size_t)src]; py = a.x[3 * (size_t)src + 1]; pz = a.x[3 * (size_t)src + 2]; }
        TileFrag in[kHidTiles], out[kHidTiles];
        const float res = sdf_value_or_tangent_h2(ring, lds, sbuf, m, px, py, pz, is_value, axis, wave, lane, in, out);
        if (is_value) {
            if (ok && lane < 32 && a.sdf_out) a.sdf_out[li] = res;
        } else {
            if (ok && lane < 32 && a.grad_out) a.grad_out[3 * (size_t)li + axis] = res;
        }
        // feature rows: 8 more ring slots (every wave steps the ring; only the value wave multiplies)
        if (want_feat) {
            TileFrag dummy_out;
            f32x16 dummy_hf;
            if (is_value) {
#pragma unroll
                for (int t = 0; t < kHidTiles; ++t) split_tile(__builtin_bit_cast(f32x16, out[t]), in[t]);
            }
            const char* fb = lds + kLdsBias + m.n_hidden_layers * 1024;
            float* dst = a.feat_packed ? a.feat_packed + (size_t)tile * kSBufFloats : nullptr;
#define IRON_FTILE(TO)                                                                                                  \
    {                                                                                                                   \
        ring.sync();                                                                                                    \
        const RingStep st = ring.step();                                                                                \
        f32x16 a_hi = zero16(), a_lo = zero16();                                                                        \
        if (is_value) {                                                                                                 \
            step_hidden<kFastActS, 0>(st.rd, fb, st.wr, st.src, st.hidden, wave, lane, TO, true, in, a_hi, a_lo,        \
                                      a_hi, a_lo, dummy_out, dummy_hf);                                                 \
            const f32x16 o = h2_combine(a_hi, a_lo);                                                                    \
            if (dst) feat_store_tile(dst, TO, lane, o);                                                                \
            if (a.feat_rows && ok) {                                                                                    \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                          \
                    a.feat_rows[(size_t)li * kHidden + 32 * (TO) + (r & 3) + 8 * (r >> 2) + 4 * half] = o[r];           \
            }                                                                                                           \
        } else {                                                                                                        \
            dma_issue(st.src, st.wr, st.hidden, wave);                                                                  \
        }                                                                                                               \
    }
            IRON_FTILE(0) IRON_FTILE(1) IRON_FTILE(2) IRON_FTILE(3) IRON_FTILE(4) IRON_FTILE(5) IRON_FTILE(6) IRON_FTILE(7)
#undef IRON_FTILE
        }
    }
    ring.drain();
}

// locate_edge_points' walk (models/raytracer.py:421-478) for 32 candidates per workgroup, all <= max_step + 1
// evaluations inside ONE launch: every candidate is an independent state machine (a found point never moves again),
// so the tile loops until all of its points are found or the step budget is spent.  Value + three tangent waves as in
// k_sdf_grad_h2; s, g are exchanged through LDS and every wave repeats the (cheap) update so that all four agree on
// the new positions and on the loop exit.
struct WalkArgs {
    const float* start;   // [n,3]
    int n;
    float cam[3];
    int max_step;
    float step_size, dot_threshold;
    float* points;        // [n,3] final positions
    uint8_t* found;       // [n]
};

__global__ __launch_bounds__(256, 1) void k_edge_walk_h2(H2StreamDev hs, H2Meta m, WalkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_value = wave == 0;
    const int axis = wave - 1;
    Ring ring;
    h2_setup(hs, lds, ring);
    float* sbuf = reinterpret_cast<float*>(lds + kLdsSbuf);
    float* xch = reinterpret_cast<float*>(lds + kLdsGradTotal);  // [4][32]: s, gx, gy, gz of the tile's points
    const int n_tiles = (a.n + kTile - 1) / kTile;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < a.n;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (ok) { px = a.start[3 * (size_t)li]; py = a.start[3 * (size_t)li + 1]; pz = a.start[3 * (size_t)li + 2]; }
        bool found = !ok;  // padding lanes never keep the tile alive
        for (int it = 0;; ++it) {
            TileFrag in[kHidTiles], out[kHidTiles];
            const float res = sdf_value_or_tangent_h2(ring, lds, sbuf, m, px, py, pz, is_value, axis, wave, lane, in, out);
            if (lane < 32) xch[wave * 32 + lane] = res;
            lds_publish_barrier();
            const float s = xch[lane & 31], gx = xch[32 + (lane & 31)], gy = xch[64 + (lane & 31)], gz = xch[96 + (lane & 31)];
            // raytracer.py:449-459
            float vx = a.cam[0] - px, vy = a.cam[1] - py, vz = a.cam[2] - pz;
            const float vn = sqrtf((vx * vx + vy * vy) + vz * vz) + 1e-10f;
            vx /= vn; vy /= vn; vz /= vn;
            const float gn = sqrtf((gx * gx + gy * gy) + gz * gz) + 1e-10f;
            const float nx = gx / gn, ny = gy / gn, nz = gz / gn;
            const float dot = (nx * vx + ny * vy) + nz * vz;
            const bool moving = !found && (fabsf(dot) > a.dot_threshold);  // a NaN dot counts as found, as in the reference
            found = !moving;
            const bool any_moving = __ballot(moving) != 0ull;  // identical in the four waves (same inputs, same arithmetic)
            __syncthreads();                                  // xch is rewritten by the next evaluation
            if (it >= a.max_step || !any_moving) break;
            if (moving) {  // raytracer.py:468-474
                float wx = nx - vx / dot, wy = ny - vy / dot, wz = nz - vz / dot;
                const float wn = sqrtf((wx * wx + wy * wy) + wz * wz) + 1e-10f;
                wx = wx / wn - s * nx; wy = wy / wn - s * ny; wz = wz / wn - s * nz;
                px += a.step_size * wx; py += a.step_size * wy; pz += a.step_size * wz;
            }
        }
        if (ok && wave == 0 && lane < 32) {
            a.points[3 * (size_t)li] = px; a.points[3 * (size_t)li + 1] = py; a.points[3 * (size_t)li + 2] = pz;
            a.found[li] = found ? 1 : 0;
        }
    }
    ring.drain();
}

// ---- material networks ------------------------------------------------------------------------------
struct MatArgs {
    const float* points;    // [*,3]
    const float* normals;   // [*,3]   (un-normalised gradient when normalise != 0)
    const float* view;      // [*,3] or null (view = -normal when neg_normal_view != 0)
    const float* feat_rows; // [*,256] row-major or null
    const float* feat_packed;  // [tiles][8][4][64][4] (mlp_core.h: feat_load_tile) or null
    const int* list;        // index into points (hit list) or null
    const int* count_ptr;
    int count;
    int normalise;          // n = g / (|g| + 1e-10)   (render_surface.py:127)
    int neg_normal_view;    // view_dirs = -normals    (rendering_func.py:7)
    int list_order_aux;     // normals/view/feat_rows are indexed by list position (1) or by point index (0)
    float* out;             // [count, d_out] in list order
};

// LP: PE levels on points, LV: PE levels on view dirs (HAS_VIEW), HAS_NRM: normals present
// the 256 features of this lane's point as register tiles (B-operand layout), from the packed per-tile buffer the
// gradient kernel writes or from row-major rows
__device__ __forceinline__ void load_feature_tiles(const MatArgs& a, int tile, int ai, bool ok, int lane, int half,
                                                   f32x16 (&h)[kHidTiles]) {
    if (a.feat_packed) {
        const float* src = a.feat_packed + (size_t)tile * kSBufFloats;
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) h[t] = feat_load_tile(src, t, lane);
    } else {
        const float4* row = reinterpret_cast<const float4*>(a.feat_rows + (size_t)ai * kHidden);
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // features 32t + 8q + 4*half + 0..3 are contiguous
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = row[(32 * t + 8 * q + 4 * half) >> 2];
                h[t][4 * q] = v.x; h[t][4 * q + 1] = v.y; h[t][4 * q + 2] = v.z; h[t][4 * q + 3] = v.w;
            }
    }
}

template <int LP, int LV, bool HAS_VIEW, bool HAS_NRM>
struct HeadCfg {
    static constexpr int kSlots = head_slots(LP) + (HAS_VIEW ? head_slots(LV) : 0) + (HAS_NRM ? 2 : 0);
    static constexpr int kQuads = kSlots <= 20 ? 5 : (kSlots <= 24 ? 6 : (kSlots + 3) / 4);  // as create_render picks nq
    static_assert(kSlots <= 48, "head too wide");
};

// HAS_SKIP: the net has a skip connection at a hidden layer (compiled in only for the instance that needs it: the extra
// layer variants in the runtime layer loop cost the skip-free instances ~250 spilled VGPRs otherwise)
template <int LP, int LV, bool HAS_VIEW, bool HAS_NRM, bool HAS_SKIP = false>
__global__ __launch_bounds__(64, 1) void k_material(RenderNetDev net, MatArgs a) {
    using Cfg = HeadCfg<LP, LV, HAS_VIEW, HAS_NRM>;
    constexpr int NQ = Cfg::kQuads;
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    WStream ws;
    ws.init(net.blob, net.blob_bytes, lane);
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int pi = ok ? (a.list ? a.list[li] : li) : 0;
        const int ai = ok ? (a.list_order_aux ? li : pi) : 0;
        float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 1.f, vx = 0.f, vy = 0.f, vz = 0.f;
        if (ok) {
            px = a.points[3 * (size_t)pi]; py = a.points[3 * (size_t)pi + 1]; pz = a.points[3 * (size_t)pi + 2];
            if (a.normals) { nx = a.normals[3 * (size_t)ai]; ny = a.normals[3 * (size_t)ai + 1]; nz = a.normals[3 * (size_t)ai + 2]; }
            if (a.normalise) {
                const float nn = sqrtf((nx * nx + ny * ny) + nz * nz) + 1e-10f;
                nx = nx / nn; ny = ny / nn; nz = nz / nn;
            }
            if (a.neg_normal_view) { vx = -nx; vy = -ny; vz = -nz; }
            else if (a.view) { vx = a.view[3 * (size_t)ai]; vy = a.view[3 * (size_t)ai + 1]; vz = a.view[3 * (size_t)ai + 2]; }
        }
        float head[4 * NQ];
#pragma unroll
        for (int i = 0; i < 4 * NQ; ++i) head[i] = 0.0f;
        int base = 0;
        head_fill<LP>(px, py, pz, half, head + base);
        base += head_slots(LP);
        if constexpr (HAS_VIEW) { head_fill<LV>(vx, vy, vz, half, head + base); base += head_slots(LV); }
        if constexpr (HAS_NRM) { head_fill<0>(nx, ny, nz, half, head + base); base += 2; }

        // features -> register tiles (B-operand layout)
        f32x16 h[kHidTiles];
        load_feature_tiles(a, tile, ai, ok, lane, half, h);

        // layer 0 (head + features) and the hidden layers share one linear weight stream
        WQueue wq;
        wq.prime(ws, net.w_feat0);
        for (int l = 0, blk = 0; l < net.n_hidden_layers; ++l) {
            const uint32_t bb = net.bias + (uint32_t)l * (kF4PerBiasLayer * 16u);
            f32x16 o[kHidTiles];
            if (l == 0) {
                hidden_layer<ReluAct, NQ>(ws, net.w_feat0, bb, true, net.w_head0, head, wq, h, o, ReluAct());
            } else if (HAS_SKIP && l == net.skip_layer) {
                // x = cat([x, rendering_input]) / sqrt(2) (fields.py:222-223), the 1/sqrt(2) folded into the weights:
                // first the x and head columns into the pre-activation sums, then the features (read again) on top
                hidden_layer<IdentityAct, NQ>(ws, net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u), bb, true, net.w_head_skip, head,
                                              wq, h, o, IdentityAct());
                load_feature_tiles(a, tile, ai, ok, lane, half, h);
                hidden_layer_accumulate<ReluAct>(ws, net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u), wq, h, o, ReluAct());
            } else {
                hidden_layer<ReluAct, NQ>(ws, net.w_hid + (uint32_t)(blk++) * (kF4PerHidLayer * 16u), bb, false, net.w_head0, head, wq, h, o,
                                          ReluAct());
            }
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) h[t] = o[t];
        }
        for (int c = 0; c < net.d_out; ++c) {
            float v = row_dot(ws, net.w_last + (uint32_t)c * (kF4PerBiasLayer * 16u), h) + net.b_last[c];
            v = net.output_scale * (v + net.output_bias);  // fields.py:235
            if (net.squeeze_out) v = net.squeeze_out_scale * (1.0f / (1.0f + expf(-v)));  // fields.py:236-237
            if (ok && lane < 32) a.out[(size_t)li * net.d_out + c] = v;
        }
    }
}

// RenderingNetwork.forward on the h2 core: a workgroup = 4 waves = 4 tiles of 32 hits of the SAME network, sharing its
// weight stream through the LDS ring.  Layer 0 = head product + feature product, relu layers, 1..3 output rows.
template <int LP, int LV, bool HAS_VIEW, bool HAS_NRM>
__global__ __launch_bounds__(256, 1) void k_material_h2(H2StreamDev hs, RenderNetDev net, MatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring ring;
    h2_setup(hs, lds, ring);
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;
    const int n_groups = (n_tiles + 3) / 4;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int tile = g * 4 + wave;
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int pi = ok ? (a.list ? a.list[li] : li) : 0;
        const int ai = ok ? (a.list_order_aux ? li : pi) : 0;
        float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 1.f, vx = 0.f, vy = 0.f, vz = 0.f;
        if (ok) {
            px = a.points[3 * (size_t)pi]; py = a.points[3 * (size_t)pi + 1]; pz = a.points[3 * (size_t)pi + 2];
            if (a.normals) { nx = a.normals[3 * (size_t)ai]; ny = a.normals[3 * (size_t)ai + 1]; nz = a.normals[3 * (size_t)ai + 2]; }
            if (a.normalise) {
                const float nn = sqrtf((nx * nx + ny * ny) + nz * nz) + 1e-10f;
                nx = nx / nn; ny = ny / nn; nz = nz / nn;
            }
            if (a.neg_normal_view) { vx = -nx; vy = -ny; vz = -nz; }
            else if (a.view) { vx = a.view[3 * (size_t)ai]; vy = a.view[3 * (size_t)ai + 1]; vz = a.view[3 * (size_t)ai + 2]; }
        }
        float head[kHeadSlots];
#pragma unroll
        for (int i = 0; i < kHeadSlots; ++i) head[i] = 0.0f;
        int base = 0;
        head_fill<LP>(px, py, pz, half, head + base);
        base += head_slots(LP);
        if constexpr (HAS_VIEW) { head_fill<LV>(vx, vy, vz, half, head + base); base += head_slots(LV); }
        if constexpr (HAS_NRM) { head_fill<0>(nx, ny, nz, half, head + base); base += 2; }
        HeadFrag hd;
        split_head(head, hd);

        TileFrag in[kHidTiles], out[kHidTiles];
        f32x16 hf[kHidTiles];
        {   // features -> split fragments
            const bool tile_ok = tile < n_tiles;
            const float* src = (a.feat_packed && tile_ok) ? a.feat_packed + (size_t)tile * kSBufFloats : nullptr;
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                f32x16 v = zero16();
                if (src) {
                    v = feat_load_tile(src, t, lane);
                } else if (a.feat_rows && ok) {
                    const float4* row = reinterpret_cast<const float4*>(a.feat_rows + (size_t)ai * kHidden);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w4 = row[(32 * t + 8 * q + 4 * half) >> 2];
                        v[4 * q] = w4.x; v[4 * q + 1] = w4.y; v[4 * q + 2] = w4.z; v[4 * q + 3] = w4.w;
                    }
                }
                split_tile(v, out[t]);
            }
        }
        // `out` holds the features; the two sets then alternate as layer input / output (no copies; the launcher admits an
        // even layer count): l0 (features + head) out -> in, middle pairs in -> out -> in, last layer in -> hf
        f32x16 c_hi, c_lo;
        h2_hidden_layer<true, true, false, 1, false, true>(ring, lds + kLdsBias, hd, lane, out, in, hf, c_hi, c_lo);
        for (int l = 1; l + 1 < net.n_hidden_layers - 1; l += 2) {
            h2_hidden_layer<true, false, false, 1, true, true>(ring, lds + kLdsBias + l * 1024, hd, lane, in, out, hf, c_hi, c_lo);
            h2_hidden_layer<true, false, false, 1, true, true>(ring, lds + kLdsBias + (l + 1) * 1024, hd, lane, out, in, hf, c_hi, c_lo);
        }
        h2_hidden_layer<true, false, true, 1, true, false>(ring, lds + kLdsBias + (net.n_hidden_layers - 1) * 1024, hd, lane, in, out, hf,
                                                           c_hi, c_lo);
        for (int c = 0; c < net.d_out; ++c) {
            float v = row_dot_lds(lds + kLdsRows + c * 1024, hf, half) + net.b_last[c];
            v = net.output_scale * (v + net.output_bias);  // fields.py:235
            if (net.squeeze_out) v = net.squeeze_out_scale * (1.0f / (1.0f + expf(-v)));  // fields.py:236-237
            if (ok && lane < 32) a.out[(size_t)li * net.d_out + c] = v;
        }
    }
    ring.drain();
}

// RenderingNetwork.forward on the h2 core for a net with a WIDE head (25..48 slots: two head ring slots) and a SKIP connection at
// a hidden layer (models/fields.py:222-223) -- the stage-1 colour net of confs/womask_iron.conf (PE-10 points, PE-4 view, normals;
// 8 layers, skip_in = [4]).  The skip layer's input is [x | head inputs | features] / sqrt(2); its feature product is taken FIRST,
// while the features are the live input set, and its partial sums wait in `scratch` ([block][wave][8 tiles][16][64] f32, written and
// read back by the same lane) until the skip layer starts its accumulators from them: no third activation set in registers.
// Stream order: pack_h2.hip build_h2_render.
template <int LP, int LV, bool HAS_VIEW, bool HAS_NRM>
__global__ __launch_bounds__(256, 1) void k_material_h2_skip(H2StreamDev hs, RenderNetDev net, MatArgs a, float* __restrict__ scratch) {
    static_assert(HeadCfg<LP, LV, HAS_VIEW, HAS_NRM>::kSlots > kHeadSlots && HeadCfg<LP, LV, HAS_VIEW, HAS_NRM>::kSlots <= 2 * kHeadSlots, "two head slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring ring;
    h2_setup(hs, lds, ring);
    float* part = scratch + ((size_t)blockIdx.x * 4 + wave) * (kHidTiles * 16 * 64);
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;
    const int n_groups = (n_tiles + 3) / 4;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int tile = g * 4 + wave;
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int pi = ok ? (a.list ? a.list[li] : li) : 0;
        const int ai = ok ? (a.list_order_aux ? li : pi) : 0;
        float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 1.f, vx = 0.f, vy = 0.f, vz = 0.f;
        if (ok) {
            px = a.points[3 * (size_t)pi]; py = a.points[3 * (size_t)pi + 1]; pz = a.points[3 * (size_t)pi + 2];
            if (a.normals) { nx = a.normals[3 * (size_t)ai]; ny = a.normals[3 * (size_t)ai + 1]; nz = a.normals[3 * (size_t)ai + 2]; }
            if (a.normalise) {
                const float nn = sqrtf((nx * nx + ny * ny) + nz * nz) + 1e-10f;
                nx = nx / nn; ny = ny / nn; nz = nz / nn;
            }
            if (a.neg_normal_view) { vx = -nx; vy = -ny; vz = -nz; }
            else if (a.view) { vx = a.view[3 * (size_t)ai]; vy = a.view[3 * (size_t)ai + 1]; vz = a.view[3 * (size_t)ai + 2]; }
        }
        HeadFrag hd, hd2;
        {
            float head[2 * kHeadSlots];
#pragma unroll
            for (int i = 0; i < 2 * kHeadSlots; ++i) head[i] = 0.0f;
            int base = 0;
            head_fill<LP>(px, py, pz, half, head + base);
            base += head_slots(LP);
            if constexpr (HAS_VIEW) { head_fill<LV>(vx, vy, vz, half, head + base); base += head_slots(LV); }
            if constexpr (HAS_NRM) { head_fill<0>(nx, ny, nz, half, head + base); base += 2; }
            split_head(head, hd);
            split_head(head + kHeadSlots, hd2);
        }
        TileFrag in[kHidTiles], out[kHidTiles];
        f32x16 hf[kHidTiles];
        {   // features -> split fragments
            const bool tile_ok = tile < n_tiles;
            const float* src = (a.feat_packed && tile_ok) ? a.feat_packed + (size_t)tile * kSBufFloats : nullptr;
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                f32x16 v = zero16();
                if (src) {
                    v = feat_load_tile(src, t, lane);
                } else if (a.feat_rows && ok) {
                    const float4* row = reinterpret_cast<const float4*>(a.feat_rows + (size_t)ai * kHidden);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w4 = row[(32 * t + 8 * q + 4 * half) >> 2];
                        v[4 * q] = w4.x; v[4 * q + 1] = w4.y; v[4 * q + 2] = w4.z; v[4 * q + 3] = w4.w;
                    }
                }
                split_tile(v, out[t]);
            }
        }
        // the skip layer's feature product (8 hidden slots, no bias, no activation) -> scratch
        {
            TileFrag dummy_out;
            f32x16 dummy_hf;
#define IRON_PTILE(TO)                                                                                                   \
    {                                                                                                                    \
        ring.sync();                                                                                                     \
        const RingStep st = ring.step();                                                                                 \
        f32x16 a_hi = zero16(), a_lo = zero16();                                                                         \
        step_hidden<true, 0, 1>(st.rd, lds + kLdsBias, st.wr, st.src, st.hidden, wave, lane, TO, false, out, a_hi, a_lo, \
                                a_hi, a_lo, dummy_out, dummy_hf, st.rec);                                                \
        const f32x16 z = h2_combine(a_hi, a_lo);                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) part[((TO) * 16 + r) * 64 + lane] = z[r];                         \
    }
            IRON_PTILE(0) IRON_PTILE(1) IRON_PTILE(2) IRON_PTILE(3) IRON_PTILE(4) IRON_PTILE(5) IRON_PTILE(6) IRON_PTILE(7)
#undef IRON_PTILE
        }
        // `out` holds the features; the sets alternate as layer input / output.  The launcher admits 8 relu layers with the
        // skip at layer 4:  l0 out->in, l1 in->out, l2 out->in, l3 in->out, l4 (skip) out->in, l5 in->out, l6 out->in, l7 in->hf
        f32x16 c_hi, c_lo;
        const char* bias = lds + kLdsBias;
        h2_hidden_layer<true, 2, false, 1, false, true>(ring, bias, hd, lane, out, in, hf, c_hi, c_lo, &hd2);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 1 * 1024, hd, lane, in, out, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 2 * 1024, hd, lane, out, in, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 3 * 1024, hd, lane, in, out, hf, c_hi, c_lo);
        h2_hidden_layer<true, 2, false, 1, true, true, true>(ring, bias + 4 * 1024, hd, lane, out, in, hf, c_hi, c_lo, &hd2, part);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 5 * 1024, hd, lane, in, out, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, false, 1, true, true>(ring, bias + 6 * 1024, hd, lane, out, in, hf, c_hi, c_lo);
        h2_hidden_layer<true, 0, true, 1, true, false>(ring, bias + 7 * 1024, hd, lane, in, out, hf, c_hi, c_lo);
        for (int c = 0; c < net.d_out; ++c) {
            float v = row_dot_lds(lds + kLdsRows + c * 1024, hf, half) + net.b_last[c];
            v = net.output_scale * (v + net.output_bias);  // fields.py:235
            if (net.squeeze_out) v = net.squeeze_out_scale * (1.0f / (1.0f + expf(-v)));  // fields.py:236-237
            if (ok && lane < 32) a.out[(size_t)li * net.d_out + c] = v;
        }
    }
    ring.drain();
}

// ---- hit list + final pointwise stage ---------------------------------------------------------------
__global__ void k_compact(const uint8_t* __restrict__ conv, int n, int* __restrict__ count, int* __restrict__ list) {
    const int stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    const int n_round = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool hit = i < n && conv[i] != 0;
        const unsigned long long m = __ballot(hit);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
        base = __shfl(base, 0, 64);
        if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

struct ShadeArgs {
    const int* list;
    const int* count_ptr;
    const float* ray_o;
    const float* ray_d;
    const float* points;
    const float* grad;   // [hits,3] list order
    const float* raw_kd; // [hits,3]
    const float* raw_ks; // [hits,3]
    const float* raw_r;  // [hits,1]
    const float* tab_trans;
    const float* tab_diff;
    float light;
    int is_metal;
    iron_shade_out out;
};

__global__ void k_ggx_shade(ShadeArgs a) {
    const int count = *a.count_ptr;
    const int stride = gridDim.x * blockDim.x;
    for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < count; li += stride) {
        const int ray = a.list[li];
        const float g[3] = {a.grad[3 * (size_t)li], a.grad[3 * (size_t)li + 1], a.grad[3 * (size_t)li + 2]};
        const float nn = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) + 1e-10f;  // render_surface.py:127
        const float nrm[3] = {g[0] / nn, g[1] / nn, g[2] / nn};
        const float p[3] = {a.points[3 * (size_t)ray], a.points[3 * (size_t)ray + 1], a.points[3 * (size_t)ray + 2]};
        const float o[3] = {a.ray_o[3 * (size_t)ray], a.ray_o[3 * (size_t)ray + 1], a.ray_o[3 * (size_t)ray + 2]};
        const float v[3] = {-a.ray_d[3 * (size_t)ray], -a.ray_d[3 * (size_t)ray + 1], -a.ray_d[3 * (size_t)ray + 2]};
        const float e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
        const float dist = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        // get_materials post-ops (rendering_func.py:7-11)
        float kd[3], ks[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            kd[c] = fabsf(a.raw_kd[3 * (size_t)li + c]);
            ks[c] = fabsf(a.raw_ks[3 * (size_t)li + c]);
        }
        if (!a.is_metal) {
            const float m = ((ks[0] + ks[1]) + ks[2]) / 3.0f;
            ks[0] = ks[1] = ks[2] = m;
        }
        const float rough = fabsf(a.raw_r[li]) + 0.01f;
        GgxOut r;
        ggx_colocated_point(a.light, dist, nrm, v, kd, ks, rough, a.tab_trans, a.tab_diff, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.out.color) a.out.color[3 * (size_t)ray + c] = r.rgb[c];
            if (a.out.diffuse_color) a.out.diffuse_color[3 * (size_t)ray + c] = r.diffuse[c];
            if (a.out.specular_color) a.out.specular_color[3 * (size_t)ray + c] = r.specular[c];
            if (a.out.diffuse_albedo) a.out.diffuse_albedo[3 * (size_t)ray + c] = kd[c];
            if (a.out.specular_albedo) a.out.specular_albedo[3 * (size_t)ray + c] = ks[c];
            if (a.out.normal) a.out.normal[3 * (size_t)ray + c] = nrm[c];
        }
        if (a.out.specular_roughness) a.out.specular_roughness[ray] = rough;
    }
}

// composite render_fn (render_surface.py:159-234) on the compacted hits: get_materials_comp post-ops (.abs()),
// CompositeRenderer.forward, scatter to the full-size maps
struct CompShadeArgs {
    const int* list;
    const int* count_ptr;
    const float *ray_o, *ray_d, *points, *grad;
    const float* raw[8];  // kd[3], ks[3], roughness, metallic, dielectric, metallic_eta, metallic_k, dielectric_eta
    const float *tab_trans, *tab_diff;
    float light;
    iron_shade_comp_out out;
};

__global__ void k_composite_shade(CompShadeArgs a) {
    const int count = *a.count_ptr;
    const int stride = gridDim.x * blockDim.x;
    for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < count; li += stride) {
        const int ray = a.list[li];
        const float g[3] = {a.grad[3 * (size_t)li], a.grad[3 * (size_t)li + 1], a.grad[3 * (size_t)li + 2]};
        const float nn = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) + 1e-10f;  // render_surface.py:185
        const float nrm[3] = {g[0] / nn, g[1] / nn, g[2] / nn};
        const float p[3] = {a.points[3 * (size_t)ray], a.points[3 * (size_t)ray + 1], a.points[3 * (size_t)ray + 2]};
        const float o[3] = {a.ray_o[3 * (size_t)ray], a.ray_o[3 * (size_t)ray + 1], a.ray_o[3 * (size_t)ray + 2]};
        const float v[3] = {-a.ray_d[3 * (size_t)ray], -a.ray_d[3 * (size_t)ray + 1], -a.ray_d[3 * (size_t)ray + 2]};
        const float e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
        const float dist = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        float kd[3], ks[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            kd[c] = fabsf(a.raw[0][3 * (size_t)li + c]);
            ks[c] = fabsf(a.raw[1][3 * (size_t)li + c]);
        }
        const float rough = fabsf(a.raw[2][li]), metallic = fabsf(a.raw[3][li]), dielectric = fabsf(a.raw[4][li]);
        const float m_eta = fabsf(a.raw[5][li]), m_k = fabsf(a.raw[6][li]), d_eta = fabsf(a.raw[7][li]);
        CompositeOut r;
        composite_point(a.light / (dist * dist + 1e-10f), nrm, v, kd, ks, rough, m_eta, m_k, d_eta, a.tab_trans, a.tab_diff, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t q = 3 * (size_t)ray + c;
            if (a.out.color) a.out.color[q] = r.rgb[c];
            if (a.out.diffuse_color) a.out.diffuse_color[q] = r.rgb[c];
            if (a.out.specular_color) a.out.specular_color[q] = r.specular[c];
            if (a.out.metallic_rgb) a.out.metallic_rgb[q] = r.metallic[c];
            if (a.out.dielectric_rgb) a.out.dielectric_rgb[q] = r.dielectric[c];
            if (a.out.diffuse_albedo) a.out.diffuse_albedo[q] = kd[c];
            if (a.out.specular_albedo) a.out.specular_albedo[q] = ks[c];
            if (a.out.normal) a.out.normal[q] = nrm[c];
        }
        if (a.out.specular_roughness) a.out.specular_roughness[ray] = rough;
        if (a.out.metallic) a.out.metallic[ray] = metallic;
        if (a.out.dielectric) a.out.dielectric[ray] = dielectric;
        if (a.out.metallic_eta) a.out.metallic_eta[ray] = m_eta;
        if (a.out.metallic_k) a.out.metallic_k[ray] = m_k;
        if (a.out.dielectric_eta) a.out.dielectric_eta[ray] = d_eta;
    }
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct ShadeLayout {
    size_t count, list, grad, kd, ks, rr, feat, park, park_bytes, total;
};
static ShadeLayout shade_layout(int64_t n) {
    ShadeLayout L;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const size_t tiles = (nn + kTile - 1) / kTile;
    size_t o = 0;
    L.count = o; o += 256;
    L.list = o; o += al256(sizeof(int) * nn);
    L.grad = o; o += al256(sizeof(float) * 3 * nn);
    L.kd = o; o += al256(sizeof(float) * 3 * nn);
    L.ks = o; o += al256(sizeof(float) * 3 * nn);
    L.rr = o; o += al256(sizeof(float) * nn);
    L.feat = o; o += al256(sizeof(float) * kSBufFloats * tiles);
    L.park = o; L.park_bytes = getall_rev_park_bytes(n); o += L.park_bytes;   // tape of the reverse-mode get_all (getall_rev.hip)
    L.total = o;
    return L;
}

static int cu_count() { return cu_budget(); }

// park / park_bytes: the tape of the reverse-mode kernel (getall_rev.hip), taken when the network has a reverse stream and the
// caller provided the workspace; otherwise the forward-mode kernels below
static int launch_sdf_grad(const iron_net* sdf, const GradArgs& a, int64_t max_tiles, hipStream_t st, void* park = nullptr,
                           size_t park_bytes = 0) {
    if (park && getall_rev_usable(sdf)) {
        if ((a.feat_packed || a.feat_rows) && !sdf->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
        return launch_sdf_getall_rev(sdf, a, max_tiles, park, park_bytes, st);
    }
    if (h2_sdf_usable(sdf)) {
        static bool attr2 = false;
        if (!attr2) {
            IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_sdf_grad_h2, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsGradTotal));
            attr2 = true;
        }
        const bool want_feat = a.feat_packed || a.feat_rows;
        if (want_feat && !sdf->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
        H2Meta m;
        m.n_hidden_layers = sdf->sdf.n_hidden_layers; m.skip_layer = sdf->sdf.skip_layer; m.scale = sdf->sdf.scale; m.b_last = sdf->sdf.b_last;
        const int cus = cu_count();
        const unsigned grid = (unsigned)(max_tiles < cus ? (max_tiles > 0 ? max_tiles : 1) : cus);
        ProfScope ps(IRON_PROF_SDF_GRAD, st);
        // the stream must match what the kernel walks: with features the 80-slot sequence, else the 72-slot one
        hipLaunchKernelGGL(k_sdf_grad_h2, dim3(grid), dim3(256), kLdsGradTotal, st, want_feat ? sdf->h2_full : sdf->h2_trace, m, a);
        IRON_HIP_TRY(hipGetLastError());
        return IRON_OK;
    }
    const size_t lds_bytes = 2 * kSBufFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_sdf_grad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    const int cus = cu_count();
    const unsigned grid = (unsigned)(max_tiles < cus ? (max_tiles > 0 ? max_tiles : 1) : cus);
    ProfScope ps(IRON_PROF_SDF_GRAD, st);
    hipLaunchKernelGGL(k_sdf_grad, dim3(grid), dim3(256), lds_bytes, st, sdf->sdf, a);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

// dispatch on the head configuration the kernels are instantiated for
static int launch_material(const iron_net* net, const MatArgs& a, int64_t max_tiles, hipStream_t st) {
    const RenderNetDev& r = net->rnd;
    const int waves = cu_count() * 4;
    const unsigned grid = (unsigned)(max_tiles < waves ? (max_tiles > 0 ? max_tiles : 1) : waves);
    const iron_net_desc& d = net->desc;
    const int lp = d.multires > 0 ? d.multires : 0;
    const int lv = d.multires_view > 0 ? d.multires_view : 0;
    ProfScope ps(IRON_PROF_MATERIAL, st);
    if (h2_enabled(net) && net->h2_scratch && r.skip_layer == 4 && r.n_hidden_layers == 8 && d.mode == IRON_MODE_IDR &&
        lp == 10 && lv == 4) {   // the stage-1 colour net (confs/womask_iron.conf)
        static bool attr4 = false;
        if (!attr4) {
            IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_material_h2_skip<10, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total));
            attr4 = true;
        }
        const int64_t groups = (max_tiles + 3) / 4;
        const int64_t cap = (int64_t)(net->h2_scratch_floats / (4 * kHidTiles * 16 * 64));   // blocks the scratch was sized for
        const unsigned g2 = (unsigned)(groups < cap ? (groups > 0 ? groups : 1) : cap);
        hipLaunchKernelGGL((k_material_h2_skip<10, 4, true, true>), dim3(g2), dim3(256), kLdsH2Total, st, net->h2_trace, r, a,
                           (float*)net->h2_scratch);
        IRON_HIP_TRY(hipGetLastError());
        return IRON_OK;
    }
    if (h2_enabled(net) && r.skip_layer == -1 && r.n_hidden_layers >= 2 && r.n_hidden_layers % 2 == 0) {
        static bool attr3 = false;
        if (!attr3) {
            (void)hipFuncSetAttribute((const void*)k_material_h2<0, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total);
            (void)hipFuncSetAttribute((const void*)k_material_h2<6, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total);
            (void)hipFuncSetAttribute((const void*)k_material_h2<6, 0, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total);
            attr3 = true;
        }
        const int64_t groups = (max_tiles + 3) / 4;
        const int cus = cu_count();
        const unsigned g2 = (unsigned)(groups < cus ? (groups > 0 ? groups : 1) : cus);
        if (d.mode == IRON_MODE_IDR && lp == 0 && lv == 4) {
            hipLaunchKernelGGL((k_material_h2<0, 4, true, true>), dim3(g2), dim3(256), kLdsH2Total, st, net->h2_trace, r, a);
        } else if (d.mode == IRON_MODE_NO_VIEW_DIR && lp == 6) {
            hipLaunchKernelGGL((k_material_h2<6, 0, false, true>), dim3(g2), dim3(256), kLdsH2Total, st, net->h2_trace, r, a);
        } else if (d.mode == IRON_MODE_POINTS_ONLY && lp == 6) {  // comp2's env_light_network (network_conf.py:367-378)
            hipLaunchKernelGGL((k_material_h2<6, 0, false, false>), dim3(g2), dim3(256), kLdsH2Total, st, net->h2_trace, r, a);
        } else {
            return IRON_ERR_UNSUPPORTED;
        }
        IRON_HIP_TRY(hipGetLastError());
        return IRON_OK;
    }
    if (r.skip_layer != -1 && !(d.mode == IRON_MODE_IDR && lp == 10 && lv == 4)) return IRON_ERR_UNSUPPORTED;
    if (d.mode == IRON_MODE_IDR && lp == 0 && lv == 4) {
        hipLaunchKernelGGL((k_material<0, 4, true, true>), dim3(grid), dim3(64), 0, st, r, a);
    } else if (d.mode == IRON_MODE_NO_VIEW_DIR && lp == 6) {
        hipLaunchKernelGGL((k_material<6, 0, false, true>), dim3(grid), dim3(64), 0, st, r, a);
    } else if (d.mode == IRON_MODE_POINTS_ONLY && lp == 6) {
        hipLaunchKernelGGL((k_material<6, 0, false, false>), dim3(grid), dim3(64), 0, st, r, a);
    } else if (d.mode == IRON_MODE_IDR && lp == 10 && lv == 4) {  // the stage-1 colour net (confs/womask_iron.conf)
        hipLaunchKernelGGL((k_material<10, 4, true, true, true>), dim3(grid), dim3(64), 0, st, r, a);
    } else {
        return IRON_ERR_UNSUPPORTED;
    }
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

}  // namespace iron

using namespace iron;

extern "C" size_t iron_sdf_get_all_workspace_bytes(const iron_net_t* sdf, int64_t n) {
    if (!sdf || sdf->desc.kind != IRON_NET_SDF || n <= 0 || !getall_rev_usable(sdf)) return 0;
    return getall_rev_park_bytes(n);
}

extern "C" int iron_sdf_get_all(const iron_net_t* sdf, const float* x, int64_t n, float* sdf_out, float* feature,
                                float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sdf || sdf->desc.kind != IRON_NET_SDF || n < 0 || (n > 0 && !x)) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (feature && !sdf->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
    if (n == 0) return IRON_OK;
    { const int rce = envelope_begin(sdf); if (rce != IRON_OK) return rce; }
    GradArgs a;
    a.x = x; a.list = nullptr; a.count_ptr = nullptr; a.count = (int)n;
    a.feat_packed = nullptr; a.sdf_out = sdf_out; a.grad_out = grad; a.feat_rows = feature;
    const int rc = launch_sdf_grad(sdf, a, (n + kTile - 1) / kTile, (hipStream_t)stream, workspace, workspace_bytes);
    // the envelope guard (envelope.hip): sdf and gradient carry every hidden activation's overflow; the feature rows are guarded where
    // they are split (the material networks' outputs)
    envelope_scan(sdf, sdf_out, n, nullptr, 1, (hipStream_t)stream);
    envelope_scan(sdf, grad, n, nullptr, 3, (hipStream_t)stream);
    return rc;
}

extern "C" int iron_edge_walk(const iron_net_t* sdf, const float* start, int64_t n, const float* cam_origin3, int32_t max_step,
                              float step_size, float dot_threshold, float* points, uint8_t* found, void* stream) {
    if (!sdf || sdf->desc.kind != IRON_NET_SDF || n < 0 || max_step < 0 || !cam_origin3) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!start || !points || !found) return IRON_ERR_BAD_ARG;
    { const int rce = envelope_begin(sdf); if (rce != IRON_OK) return rce; }
    if (!h2_sdf_usable(sdf)) return IRON_ERR_UNSUPPORTED;  // the caller then walks with iron_sdf_get_all launches
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_edge_walk_h2, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsGradTotal + 512));
        attr = true;
    }
    H2Meta m;
    m.n_hidden_layers = sdf->sdf.n_hidden_layers; m.skip_layer = sdf->sdf.skip_layer; m.scale = sdf->sdf.scale; m.b_last = sdf->sdf.b_last;
    WalkArgs a;
    a.start = start; a.n = (int)n; a.cam[0] = cam_origin3[0]; a.cam[1] = cam_origin3[1]; a.cam[2] = cam_origin3[2];
    a.max_step = max_step; a.step_size = step_size; a.dot_threshold = dot_threshold; a.points = points; a.found = found;
    const int64_t tiles = (n + kTile - 1) / kTile;
    const int cus = cu_count();
    ProfScope ps(IRON_PROF_SDF_GRAD, st);
    hipLaunchKernelGGL(k_edge_walk_h2, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(256), kLdsGradTotal + 512, st, sdf->h2_trace, m, a);
    IRON_HIP_TRY(hipGetLastError());
    envelope_scan(sdf, points, n, nullptr, 3, st);
    return IRON_OK;
}

extern "C" int iron_render_forward(const iron_net_t* net, const float* points, const float* normals,
                                   const float* view_dirs, const float* features, int64_t n, float* out, void* stream) {
    if (!net || net->desc.kind != IRON_NET_RENDER || n < 0) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!points || !features || !out) return IRON_ERR_BAD_ARG;
    if (((uintptr_t)features & 15) != 0) return IRON_ERR_BAD_ARG;
    const int mode = net->desc.mode;
    if ((mode == IRON_MODE_IDR || mode == IRON_MODE_NO_VIEW_DIR) && !normals) return IRON_ERR_BAD_ARG;
    if ((mode == IRON_MODE_IDR || mode == IRON_MODE_NO_NORMAL) && !view_dirs) return IRON_ERR_BAD_ARG;
    { const int rce = envelope_begin(net); if (rce != IRON_OK) return rce; }
    MatArgs a;
    a.points = points; a.normals = normals; a.view = view_dirs; a.feat_rows = features; a.feat_packed = nullptr;
    a.list = nullptr; a.count_ptr = nullptr; a.count = (int)n; a.normalise = 0; a.neg_normal_view = 0;
    a.list_order_aux = 1; a.out = out;
    const int rc = launch_material(net, a, (n + kTile - 1) / kTile, (hipStream_t)stream);
    envelope_scan(net, out, n, nullptr, net->desc.d_out, (hipStream_t)stream);
    return rc;
}

// composite: the GGX layout followed by the five extra scalar maps
static size_t comp_extra_off(int64_t n, int i) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return shade_layout(n).total + (size_t)i * al256(sizeof(float) * nn);
}

extern "C" size_t iron_shade_composite_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return comp_extra_off(n, 5);
}

extern "C" int iron_shade_composite(const iron_shade_comp_nets* nets, float light, const float* tab_trans,
                                    const float* tab_diff_trans, const float* ray_o, const float* ray_d, const float* points,
                                    const uint8_t* conv, int64_t n, const iron_shade_comp_out* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!nets || !out) return IRON_ERR_BAD_ARG;
    const iron_net_t* mats[8] = {nets->diffuse_albedo, nets->specular_albedo, nets->specular_roughness, nets->metallic,
                                 nets->dielectric, nets->metallic_eta, nets->metallic_k, nets->dielectric_eta};
    if (!nets->sdf) return IRON_ERR_BAD_ARG;
    for (int i = 0; i < 8; ++i) {
        if (!mats[i]) return IRON_ERR_BAD_ARG;
        if (mats[i]->desc.kind != IRON_NET_RENDER || mats[i]->desc.d_out != (i < 2 ? 3 : 1)) return IRON_ERR_UNSUPPORTED;
    }
    if (nets->sdf->desc.kind != IRON_NET_SDF || !nets->sdf->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
    if (n < 0 || n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!tab_trans || !tab_diff_trans || !ray_o || !ray_d || !points || !conv || !workspace) return IRON_ERR_BAD_ARG;
    { int rce = envelope_begin(nets->sdf); for (int i = 0; i < 8 && rce == IRON_OK; ++i) rce = envelope_begin(mats[i]); if (rce != IRON_OK) return rce; }
    const ShadeLayout L = shade_layout(n);
    if (workspace_bytes < comp_extra_off(n, 5)) return IRON_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    int* count = (int*)(base + L.count);
    int* list = (int*)(base + L.list);
    float* grad = (float*)(base + L.grad);
    float* feat = (float*)(base + L.feat);
    float* raw[8] = {(float*)(base + L.kd), (float*)(base + L.ks), (float*)(base + L.rr), nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 5; ++i) raw[3 + i] = (float*)(base + comp_extra_off(n, i));

    // non-hit pixels are zero in every output (render_surface.py:161-182)
    IRON_HIP_TRY(hipMemsetAsync(count, 0, 256, st));
    float* outs3[8] = {out->color, out->diffuse_color, out->specular_color, out->diffuse_albedo, out->specular_albedo, out->normal,
                       out->metallic_rgb, out->dielectric_rgb};
    for (float* p : outs3)
        if (p) IRON_HIP_TRY(hipMemsetAsync(p, 0, sizeof(float) * 3 * (size_t)n, st));
    float* outs1[6] = {out->specular_roughness, out->metallic_eta, out->metallic_k, out->dielectric_eta, out->metallic, out->dielectric};
    for (float* p : outs1)
        if (p) IRON_HIP_TRY(hipMemsetAsync(p, 0, sizeof(float) * (size_t)n, st));
    {
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(k_compact, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, st, conv, (int)n, count, list);
    }
    const int64_t max_tiles = (n + kTile - 1) / kTile;
    GradArgs ga;
    ga.x = points; ga.list = list; ga.count_ptr = count; ga.count = 0;
    ga.feat_packed = feat; ga.sdf_out = nullptr; ga.grad_out = grad; ga.feat_rows = nullptr;
    int rc = launch_sdf_grad(nets->sdf, ga, max_tiles, st, base + L.park, L.park_bytes);
    if (rc != IRON_OK) return rc;
    MatArgs ma;
    ma.points = points; ma.normals = grad; ma.view = nullptr; ma.feat_rows = nullptr; ma.feat_packed = feat;
    ma.list = list; ma.count_ptr = count; ma.count = 0; ma.normalise = 1; ma.list_order_aux = 1;
    envelope_scan(nets->sdf, grad, n, count, 3, st);
    for (int i = 0; i < 8; ++i) {
        ma.neg_normal_view = (i == 0) ? 1 : 0;  // the diffuse head sees view := -normal (rendering_func.py:22)
        ma.out = raw[i];
        rc = launch_material(mats[i], ma, max_tiles, st);
        if (rc != IRON_OK) return rc;
        envelope_scan(mats[i], raw[i], n, count, mats[i]->desc.d_out, st);
    }
    CompShadeArgs sa;
    sa.list = list; sa.count_ptr = count; sa.ray_o = ray_o; sa.ray_d = ray_d; sa.points = points; sa.grad = grad;
    for (int i = 0; i < 8; ++i) sa.raw[i] = raw[i];
    sa.tab_trans = tab_trans; sa.tab_diff = tab_diff_trans; sa.light = light; sa.out = *out;
    {
        ProfScope ps(IRON_PROF_GGX, st);
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(k_composite_shade, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, st, sa);
    }
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" size_t iron_shade_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return shade_layout(n).total;
}

extern "C" int iron_shade_ggx(const iron_shade_nets* nets, float light, int32_t is_metal, const float* tab_trans,
                              const float* tab_diff_trans, const float* ray_o, const float* ray_d, const float* points,
                              const uint8_t* conv, int64_t n, const iron_shade_out* out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!nets || !nets->sdf || !nets->diffuse_albedo || !nets->specular_albedo || !nets->specular_roughness || !out)
        return IRON_ERR_BAD_ARG;
    if (nets->sdf->desc.kind != IRON_NET_SDF || !nets->sdf->sdf.w_feat) return IRON_ERR_UNSUPPORTED;
    if (nets->diffuse_albedo->desc.d_out != 3 || nets->specular_albedo->desc.d_out != 3 ||
        nets->specular_roughness->desc.d_out != 1)
        return IRON_ERR_UNSUPPORTED;
    if (n < 0 || n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!tab_trans || !tab_diff_trans || !ray_o || !ray_d || !points || !conv || !workspace) return IRON_ERR_BAD_ARG;
    {
        int rce = envelope_begin(nets->sdf);
        if (rce == IRON_OK) rce = envelope_begin(nets->diffuse_albedo);
        if (rce == IRON_OK) rce = envelope_begin(nets->specular_albedo);
        if (rce == IRON_OK) rce = envelope_begin(nets->specular_roughness);
        if (rce != IRON_OK) return rce;
    }
    const ShadeLayout L = shade_layout(n);
    if (workspace_bytes < L.total) return IRON_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    int* count = (int*)(base + L.count);
    int* list = (int*)(base + L.list);
    float* grad = (float*)(base + L.grad);
    float* kd = (float*)(base + L.kd);
    float* ks = (float*)(base + L.ks);
    float* rr = (float*)(base + L.rr);
    float* feat = (float*)(base + L.feat);

    // non-hit pixels are zero in every output (render_surface.py:119-125)
    IRON_HIP_TRY(hipMemsetAsync(count, 0, 256, st));
    float* outs3[6] = {out->color, out->diffuse_color, out->specular_color, out->diffuse_albedo, out->specular_albedo, out->normal};
    for (float* p : outs3)
        if (p) IRON_HIP_TRY(hipMemsetAsync(p, 0, sizeof(float) * 3 * (size_t)n, st));
    if (out->specular_roughness) IRON_HIP_TRY(hipMemsetAsync(out->specular_roughness, 0, sizeof(float) * (size_t)n, st));

    {
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(k_compact, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, st, conv, (int)n, count, list);
    }
    const int64_t max_tiles = (n + kTile - 1) / kTile;
    GradArgs ga;
    ga.x = points; ga.list = list; ga.count_ptr = count; ga.count = 0;
    ga.feat_packed = feat; ga.sdf_out = nullptr; ga.grad_out = grad; ga.feat_rows = nullptr;
    int rc = launch_sdf_grad(nets->sdf, ga, max_tiles, st, base + L.park, L.park_bytes);
    if (rc != IRON_OK) return rc;

    MatArgs ma;
    ma.points = points; ma.normals = grad; ma.view = nullptr; ma.feat_rows = nullptr; ma.feat_packed = feat;
    ma.list = list; ma.count_ptr = count; ma.count = 0; ma.normalise = 1; ma.list_order_aux = 1;
    ma.neg_normal_view = 1; ma.out = kd;
    rc = launch_material(nets->diffuse_albedo, ma, max_tiles, st);
    if (rc != IRON_OK) return rc;
    ma.neg_normal_view = 0; ma.out = ks;
    rc = launch_material(nets->specular_albedo, ma, max_tiles, st);
    if (rc != IRON_OK) return rc;
    ma.out = rr;
    rc = launch_material(nets->specular_roughness, ma, max_tiles, st);
    if (rc != IRON_OK) return rc;
    // envelope guard (envelope.hip): the values each network returned for the *count hits
    envelope_scan(nets->sdf, grad, n, count, 3, st);
    envelope_scan(nets->diffuse_albedo, kd, n, count, 3, st);
    envelope_scan(nets->specular_albedo, ks, n, count, 3, st);
    envelope_scan(nets->specular_roughness, rr, n, count, 1, st);

    ShadeArgs sa;
    sa.list = list; sa.count_ptr = count; sa.ray_o = ray_o; sa.ray_d = ray_d; sa.points = points; sa.grad = grad;
    sa.raw_kd = kd; sa.raw_ks = ks; sa.raw_r = rr; sa.tab_trans = tab_trans; sa.tab_diff = tab_diff_trans;
    sa.light = light; sa.is_metal = is_metal; sa.out = *out;
    {
        ProfScope ps(IRON_PROF_GGX, st);
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(k_ggx_shade, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, st, sa);
    }
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}
