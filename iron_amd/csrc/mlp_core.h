// Register-resident fp32-MFMA MLP core for gfx950 (see iron_common.h for the tile geometry).
//
// One wave64 owns 32 points.  Every dense layer is  Z^T[out, point] = W[out, in] * H^T[in, point]
// issued as v_mfma_f32_32x32x2_f32 with A = a packed weight fragment streamed from L2 (one
// buffer_load_dwordx4 per lane feeds four MFMAs; SGPR descriptor + scalar offsets, so the stream
// costs one address VGPR) and B = the previous layer's accumulator registers used in place.
// Bias enters as the accumulator's initial value; the activation is applied on the accumulator
// registers, which then are the next layer's B operands.  No LDS, no barriers, no inter-wave state.
#pragma once
#include "iron_common.h"

namespace iron {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Packed feature tiles (what get_all hands to the material kernels through the shading workspace): per 32-point tile
// [8 feature tiles][4 pieces][64 lanes][4 f32] -- element (t, r, lane) at ((t * 4 + r / 4) * 64 + lane) * 4 + r % 4 -- so that a lane's 16
// registers of a tile are four 16-byte pieces and a wave's piece is 1 KiB contiguous (round 3: was [t][r][lane], sixteen dword accesses).
template <bool NT = false>
__device__ __forceinline__ void feat_store_tile(float* __restrict__ tile_base, int t, int lane, const f32x16& o) {
    f32x4* p = reinterpret_cast<f32x4*>(tile_base) + (size_t)(t * 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
        v[0] = o[4 * q]; v[1] = o[4 * q + 1]; v[2] = o[4 * q + 2]; v[3] = o[4 * q + 3];
        if (NT) __builtin_nontemporal_store(v, p + q * 64);   // read once, by a later kernel
        else p[q * 64] = v;
    }
}
__device__ __forceinline__ f32x16 feat_load_tile(const float* __restrict__ tile_base, int t, int lane) {
    const float4* p = reinterpret_cast<const float4*>(tile_base) + (size_t)(t * 4) * 64 + lane;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 x = p[q * 64];
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
    }
    return v;
}

#ifndef IRON_WQ_DEPTH
#define IRON_WQ_DEPTH 4  // weight fragments (pairs of 1-KiB loads) kept in flight ahead of the MFMAs
#endif

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Packed-weight stream of one network: buffer descriptor over the blob + this lane's byte offsets.
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int v_lane;  // lane * 16      (fragment loads: 64 lanes x float4 = 1 KiB)
    int v_half;  // (lane>>5) * 64 (per-half rows of 16 floats: bias / output rows)
    __device__ __forceinline__ void init(const void* blob, uint32_t bytes, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(blob), 0, (int)bytes, 0x00020000);
        v_lane = lane * 16;
        v_half = (lane >> 5) * 64;
    }
    // s_off: wave-uniform byte offset
    __device__ __forceinline__ f32x4 frag(uint32_t s_off) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, v_lane, (int)s_off, 0));
    }
    __device__ __forceinline__ f32x4 half_row(uint32_t s_off) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, v_half, (int)s_off, 0));
    }
};

// nn.Softplus(beta=100), torch threshold 20 (models/fields.py:80): x if 100x > 20 else log1p(exp(100x))/100
template <bool FAST>
__device__ __forceinline__ float softplus100(float z) {
    if constexpr (FAST) {
        // max(z, 0) + log2(1 + 2^(-|100 z| log2 e)) * ln2 / 100 through v_exp_f32 / v_log_f32: the same function as the
        // thresholded form (above 100 z = 20 the second term is < 2.1e-11), never overflows, needs no select; abs err
        // <~ 1e-9 on outputs of magnitude <= 0.2
        const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(z) * -144.26950408889634f);
        return __builtin_fmaf(__builtin_amdgcn_logf(1.0f + e), 0.0069314718055994531f, fmaxf(z, 0.0f));
    } else {
        const float t = z * 100.0f;
        const float s = log1pf(expf(t)) / 100.0f;
        return t > 20.0f ? z : s;
    }
}

// d softplus100 / dz = sigmoid(100 z) (1 above the threshold)
__device__ __forceinline__ float softplus100_grad(float z) {
    const float t = z * 100.0f;
    const float s = 1.0f / (1.0f + expf(-t));
    return t > 20.0f ? 1.0f : s;
}

// 16 floats of this lane-half for (tile): packed [tile][2][16]; base = byte offset of the array
__device__ __forceinline__ f32x16 load_half_tile(const WStream& ws, uint32_t base, int tile) {
    const uint32_t o = base + (uint32_t)tile * 128u;
    const f32x4 b0 = ws.half_row(o), b1 = ws.half_row(o + 16), b2 = ws.half_row(o + 32), b3 = ws.half_row(o + 48);
    f32x16 v;
    v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w;
    v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    v[8] = b2.x; v[9] = b2.y; v[10] = b2.z; v[11] = b2.w;
    v[12] = b3.x; v[13] = b3.y; v[14] = b3.z; v[15] = b3.w;
    return v;
}

// Head slots of one vec3 source: slot0 = (x|y), slot1 = (z|0), then (sin|cos)(2^k v_c).
// models/embedder.py:27-36: the argument is v * 2^k (exact in fp32), then sin / cos.
template <int LEVELS>
__device__ __forceinline__ void head_fill(float vx, float vy, float vz, int half, float* slots) {
    slots[0] = half ? vy : vx;
    slots[1] = half ? 0.0f : vz;
#pragma unroll
    for (int k = 0; k < LEVELS; ++k) {
        const float f = (float)(1 << k);
        float s, c;
        sincosf(vx * f, &s, &c); slots[2 + 3 * k + 0] = half ? c : s;
        sincosf(vy * f, &s, &c); slots[2 + 3 * k + 1] = half ? c : s;
        sincosf(vz * f, &s, &c); slots[2 + 3 * k + 2] = half ? c : s;
    }
}

// 4-component source (NeRF background points): slot0 = (x|y), slot1 = (z|w), then (sin|cos)(2^k v_c) at 2 + 4k + c
template <int LEVELS>
__device__ __forceinline__ void head_fill4(float vx, float vy, float vz, float vw, int half, float* slots) {
    slots[0] = half ? vy : vx;
    slots[1] = half ? vw : vz;
    const float v[4] = {vx, vy, vz, vw};
#pragma unroll
    for (int k = 0; k < LEVELS; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s, cs;
            sincosf(v[c] * f, &s, &cs);
            slots[2 + 4 * k + c] = half ? cs : s;
        }
    }
}

__device__ __forceinline__ void mfma_quad(const f32x4& a0, const f32x4& a1, float b0, float b1, float b2, float b3,
                                          f32x16& acc0, f32x16& acc1) {
    acc0 = mfma32(a0.x, b0, acc0);
    acc1 = mfma32(a1.x, b0, acc1);
    acc0 = mfma32(a0.y, b1, acc0);
    acc1 = mfma32(a1.y, b1, acc1);
    acc0 = mfma32(a0.z, b2, acc0);
    acc1 = mfma32(a1.z, b2, acc1);
    acc0 = mfma32(a0.w, b3, acc0);
    acc1 = mfma32(a1.w, b3, acc1);
}

// acc{0,1} += W_head[pair] * head   (NQ quads of head slots; packed [pair][NQ][2][64] float4)
template <int NQ>
__device__ __forceinline__ void dense_head_pair(const WStream& ws, uint32_t base, int pair, const float* head,
                                                f32x16& acc0, f32x16& acc1) {
    const uint32_t o = base + (uint32_t)pair * (NQ * 2048u);
    // fragments are fetched in chunks of <= 6 quads (a 12-quad head -- the stage-1 colour net's PE-10 + PE-4 + normal
    // inputs -- would otherwise hold 96 VGPRs of weights at once)
    constexpr int kChunk = NQ <= 6 ? NQ : 6;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += kChunk) {
        f32x4 a0[kChunk], a1[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < NQ) {
                a0[q] = ws.frag(o + (q0 + q) * 2048u);
                a1[q] = ws.frag(o + (q0 + q) * 2048u + 1024u);
            }
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q0 + q < NQ)
                mfma_quad(a0[q], a1[q], head[4 * (q0 + q)], head[4 * (q0 + q) + 1], head[4 * (q0 + q) + 2], head[4 * (q0 + q) + 3],
                          acc0, acc1);
        }
    }
}

// Weight FIFO over one 256x256 layer: 128 steps of (two 1-KiB fragments -> 8 MFMAs), in the packed
// order [pair][in-tile][quad].  `D` steps are kept in flight ahead of the MFMAs that consume them.
struct WQueue {
    static constexpr int D = IRON_WQ_DEPTH;
    f32x4 a0[D], a1[D];
    __device__ __forceinline__ void prime(const WStream& ws, uint32_t base) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a0[d] = ws.frag(base + d * 2048u);
            a1[d] = ws.frag(base + d * 2048u + 1024u);
        }
    }
};

#ifndef IRON_SCHED_MASK
#define IRON_SCHED_MASK 0x6  // VALU|SALU may cross a step boundary; MFMA and VMEM may not
#endif

// acc{0,1} += W[pair] * in  for a 256-wide input held as 8 register tiles; PAIR is compile-time so
// that FIFO slots are static registers.  The stream is linear in memory across pairs AND across
// consecutive hidden layers, so the FIFO simply keeps prefetching step g+D: at the end of a layer
// it is already primed for the next one (the blob is padded, and buffer loads are range-checked).
// sched_barrier pins each step's two loads in front of its eight MFMAs: without it hipcc sinks
// every load to just before its first use and the L2 latency is exposed once per step.
template <int PAIR>
__device__ __forceinline__ void dense_hidden_pair(const WStream& ws, uint32_t base, WQueue& wq,
                                                  const f32x16 (&in)[kHidTiles], f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        constexpr int D = WQueue::D;
        const int g = PAIR * 32 + s;
        const int slot = g % D;
        const f32x4 x0 = wq.a0[slot], x1 = wq.a1[slot];
        wq.a0[slot] = ws.frag(base + (uint32_t)(g + D) * 2048u);
        wq.a1[slot] = ws.frag(base + (uint32_t)(g + D) * 2048u + 1024u);
        const int ti = s >> 2, q = s & 3;
        mfma_quad(x0, x1, in[ti][4 * q], in[ti][4 * q + 1], in[ti][4 * q + 2], in[ti][4 * q + 3], acc0, acc1);
        __builtin_amdgcn_sched_barrier(IRON_SCHED_MASK);
    }
}

template <bool FAST>
__device__ __forceinline__ f32x16 softplus_tile(f32x16 z) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = softplus100<FAST>(z[i]);
    return r;
}

__device__ __forceinline__ f32x16 relu_tile(f32x16 z) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = fmaxf(z[i], 0.0f);
    return r;
}

// dot of this lane's 128 resident features with a packed output row ([8][2][16] floats), summed
// over the two lane halves -> every lane of a point holds the full 256-term dot product.
__device__ __forceinline__ float row_dot(const WStream& ws, uint32_t base, const f32x16 (&h)[kHidTiles]) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) {
        const f32x16 w = load_half_tile(ws, base, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(w[r], h[t][r], s);
    }
    return s + __shfl_xor(s, 32, 64);
}

// the same over the first NT tiles only (a layer narrower than 256)
template <int NT>
__device__ __forceinline__ float row_dot_n(const WStream& ws, uint32_t base, const f32x16 (&h)[kHidTiles]) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x16 w = load_half_tile(ws, base, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(w[r], h[t][r], s);
    }
    return s + __shfl_xor(s, 32, 64);
}

constexpr int kSdfPeLevels = 6;
constexpr int kSdfHeadQuads = head_slots(kSdfPeLevels) / 4;  // 20 slots -> 5 quads

// one 256 -> 256 layer (+ optional head product), activation ACT applied, result in `out`
// NPAIRS < 4: only the first 64 * NPAIRS output features exist (NeRF's 128-wide view layer); the weight FIFO still walks
// the block linearly, so such a layer must be the LAST consumer of the stream.
template <class Act, int NQ, int NPAIRS = kPairs>
__device__ __forceinline__ void hidden_layer(const WStream& ws, uint32_t w_base, uint32_t b_base, bool with_head,
                                             uint32_t head_base, const float* head, WQueue& wq,
                                             const f32x16 (&in)[kHidTiles], f32x16 (&out)[kHidTiles], Act act) {
#define IRON_PAIR(P)                                                                  \
    if constexpr (P < NPAIRS) {                                                       \
        f32x16 a0 = load_half_tile(ws, b_base, 2 * P);                                \
        f32x16 a1 = load_half_tile(ws, b_base, 2 * P + 1);                            \
        if (with_head) dense_head_pair<NQ>(ws, head_base, P, head, a0, a1);           \
        dense_hidden_pair<P>(ws, w_base, wq, in, a0, a1);                             \
        out[2 * P] = act(a0);                                                         \
        out[2 * P + 1] = act(a1);                                                     \
    }
    IRON_PAIR(0) IRON_PAIR(1) IRON_PAIR(2) IRON_PAIR(3)
#undef IRON_PAIR
}

// second half of a skip layer: out (the pre-activation partial sums of the first pass) += W * in, then the activation
template <class Act>
__device__ __forceinline__ void hidden_layer_accumulate(const WStream& ws, uint32_t w_base, WQueue& wq,
                                                        const f32x16 (&in)[kHidTiles], f32x16 (&out)[kHidTiles], Act act) {
#define IRON_PAIR(P)                                                                  \
    {                                                                                 \
        f32x16 a0 = out[2 * P];                                                       \
        f32x16 a1 = out[2 * P + 1];                                                   \
        dense_hidden_pair<P>(ws, w_base, wq, in, a0, a1);                             \
        out[2 * P] = act(a0);                                                         \
        out[2 * P + 1] = act(a1);                                                     \
    }
    IRON_PAIR(0) IRON_PAIR(1) IRON_PAIR(2) IRON_PAIR(3)
#undef IRON_PAIR
}

template <bool FAST>
struct SoftplusAct {
    __device__ __forceinline__ f32x16 operator()(const f32x16& z) const { return softplus_tile<FAST>(z); }
};
struct IdentityAct {
    __device__ __forceinline__ f32x16 operator()(const f32x16& z) const { return z; }
};
struct ReluAct {
    __device__ __forceinline__ f32x16 operator()(const f32x16& z) const { return relu_tile(z); }
};

// SDFNetwork hidden stack (models/fields.py:82-97): PE -> 8 softplus layers, skip-concat at
// `skip_layer` folded into a second head product.  On return `h` holds the last hidden activation.
template <bool FAST>
__device__ __forceinline__ void sdf_hidden_stack(const SdfNetDev& n, const WStream& ws, float x, float y, float z,
                                                 int lane, f32x16 (&h)[kHidTiles]) {
    const int half = lane >> 5;
    float pe[4 * kSdfHeadQuads];
    head_fill<kSdfPeLevels>(x * n.scale, y * n.scale, z * n.scale, half, pe);

    // layer 0: 39 -> 256 on the PE slots
#pragma unroll
    for (int p = 0; p < kPairs; ++p) {
        f32x16 a0 = load_half_tile(ws, n.bias, 2 * p);
        f32x16 a1 = load_half_tile(ws, n.bias, 2 * p + 1);
        dense_head_pair<kSdfHeadQuads>(ws, n.w_pe0, p, pe, a0, a1);
        h[2 * p] = softplus_tile<FAST>(a0);
        h[2 * p + 1] = softplus_tile<FAST>(a1);
    }
    // layers 1 .. n_hidden-1: 256 -> 256 (+ PE at the skip layer); one weight FIFO runs through all of them
    WQueue wq;
    wq.prime(ws, n.w_hid);
    for (int l = 1; l < n.n_hidden_layers; ++l) {
        const uint32_t w = n.w_hid + (uint32_t)(l - 1) * (kF4PerHidLayer * 16u);
        const uint32_t b = n.bias + (uint32_t)l * (kF4PerBiasLayer * 16u);
        f32x16 o[kHidTiles];
        hidden_layer<SoftplusAct<FAST>, kSdfHeadQuads>(ws, w, b, l == n.skip_layer, n.w_pe_skip, pe, wq, h, o,
                                                       SoftplusAct<FAST>());
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) h[t] = o[t];
    }
}

// signed distance of the point on lane&31 (both lane halves return the same value)
template <bool FAST>
__device__ __forceinline__ float sdf_eval(const SdfNetDev& n, const WStream& ws, float x, float y, float z, int lane) {
    f32x16 h[kHidTiles];
    sdf_hidden_stack<FAST>(n, ws, x, y, z, lane, h);
    const float s = row_dot(ws, n.w_last, h) + n.b_last;
    return s / n.scale;
}

}  // namespace iron
