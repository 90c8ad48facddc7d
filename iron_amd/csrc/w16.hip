// "w16" MLP core for gfx950: the split-fp16 arithmetic of mlp_h2.h (fp32 operand = two fp16 pieces, three MFMA products) on
// v_mfma_f32_16x16x32_f16 with EIGHT waves per workgroup, two per SIMD, 16 points per wave.
//
// Why a second core.  In mlp_h2.h a wave holds 32 points x 256 features twice (input and output set, split fp16: 256 registers)
// and therefore runs alone on its SIMD; every cycle it spends issuing LDS-DMA (~490 per ring step), waiting at the step's barrier,
// or issuing VALU at the lone-wave rate of 4 cycles per instruction is a cycle the matrix pipe idles: 55 % MFMA busy, by the stamps
// of DESIGN.md 3.1b a ceiling of ~60 % for that design.  With 16 points per wave the two activation sets are 128 registers, a wave
// fits 256, and a SIMD holds two waves that are deliberately OUT OF PHASE: between two ring barriers the early wave (0..3) runs
// [48 MFMAs of tile t] [epilogue of tile t], the late wave (4..7) [epilogue of tile t-1] [48 MFMAs of tile t] -- matrix work of
// one beside vector work of the other, on the same weight slot (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
//
// Layout.  MFMA 16x16x32: A = weights [16 out x 32 k] (lane l: row l % 16, k = 8 (l / 16) + i), B = activations [32 k x 16 points]
// (lane l: point l % 16, k = 8 (l / 16) + i), C/D: lane l holds point l % 16, rows 4 (l / 16) + r.  A lane-group g = l / 16 thus owns
// rows 4g..4g+3 of every 16-feature C tile; two C tiles (a slot's 32 output features) are exactly one B fragment of the next layer
// (k-step s of layer l+1 = output slot s of layer l; element i < 4 from the first tile, i >= 4 from the second), with the weights'
// k order permuted to match at pack time: input feature of (k-step s, group g, element i) = 32 s + 16 (i >= 4) + 4 g + (i & 3).
// The ring (4 x 32 KiB, LDS-DMA, three slots ahead) and the slot sizes (hidden 32 KiB = 8 k-steps x 2 tiles x {hi, lo} x 1 KiB,
// head 8 KiB = 2 k-steps of the 39 -> 64 padded PE) are those of mlp_h2.h; each of the 8 waves issues 4 of a slot's 32 DMA pieces.
#include <string.h>
#include <vector>
#include "iron_common.h"
#include "mlp_core.h"
#include "pack_common.h"
#include "h2_setup.h"
#include "lds_dma.h"

namespace iron {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kW16Lds = kRingBytes + 8 * 1024 /* bias [8][256] */ + 1024 /* last row */ + 256 /* flags */;
constexpr int kW16LdsBias = kRingBytes, kW16LdsRow = kRingBytes + 8 * 1024, kW16LdsMisc = kW16LdsRow + 1024;
constexpr int kW16Ahead = 3;

// ---- packing ------------------------------------------------------------------------------------------------------------------
// hidden slot of output tile `to` (32 rows): fragments [ks 0..7][u 0..1][piece] x [lane 64][8 halves]
__global__ void k_pack_w16_hidden(_Float16* __restrict__ dst, PackSrc s, int to, int cols_valid) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks, u, lane, i)
    if (e >= 8 * 2 * 64 * 8) return;
    const int i = e & 7, lane = (e >> 3) & 63, u = (e >> 9) & 1, ks = e >> 10;
    const int row = 32 * to + 16 * u + (lane & 15), g = lane >> 4;
    const int col = 32 * ks + 16 * (i >> 2) + 4 * g + (i & 3);
    float w = 0.0f;
    if (row < s.rows_valid && col < cols_valid) w = s.w[(size_t)(s.row_off + row) * s.ld + col] * s.scale[s.row_off + row] * s.mul;
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
    _Float16* f = dst + ((size_t)((ks * 2 + u) * 2) * 64 + lane) * 8 + i;
    f[0] = hi;
    f[64 * 8] = lo;
}

// head slot: fragments [ks 0..1][u][piece]; PE column p = 32 ks + 8 g + i (p < pe_cols), read at column col_off + p
__global__ void k_pack_w16_head(_Float16* __restrict__ dst, PackSrc s, int to, int col_off, int pe_cols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * 2 * 64 * 8) return;
    const int i = e & 7, lane = (e >> 3) & 63, u = (e >> 9) & 1, ks = e >> 10;
    const int row = 32 * to + 16 * u + (lane & 15), g = lane >> 4;
    const int p = 32 * ks + 8 * g + i;
    float w = 0.0f;
    if (row < s.rows_valid && p < pe_cols) w = s.w[(size_t)(s.row_off + row) * s.ld + col_off + p] * s.scale[s.row_off + row] * s.mul;
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
    _Float16* f = dst + ((size_t)((ks * 2 + u) * 2) * 64 + lane) * 8 + i;
    f[0] = hi;
    f[64 * 8] = lo;
}

__global__ void k_pack_w16_vec(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ scale, int n_valid) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 256) dst[e] = e < n_valid ? src[e] * (scale ? scale[0] : 1.0f) : 0.0f;
}

bool use_w16_core();

int build_w16_sdf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st) {
    const iron_net_desc& d = net->desc;
    const int nl = d.n_linear, skip = d.skip_layer, pe = pe_width(d.multires);
    if (!use_w16_core()) return IRON_OK;                                             // the prototype core packs only when it is selected
    if (d.d_hidden != kHidden || nl != 9 || skip != 4 || pe > 64) return IRON_OK;   // the 8 x 256 skip-4 SDF network only
    std::vector<uint32_t> table;
    size_t off = 0;
    auto add = [&](int kind) { table.push_back((uint32_t)off); table.push_back((uint32_t)kind); off += kind ? kSlotBytes : 8192; };
    for (int to = 0; to < 8; ++to) add(0);
    for (int l = 1; l <= nl - 2; ++l)
        for (int to = 0; to < 8; ++to) { if (l == skip) add(0); add(1); }
    const uint32_t n_slots = (uint32_t)(table.size() / 2);
    const size_t table_off = (off + 255) & ~(size_t)255;
    const size_t bias_off = table_off + 1024;
    const size_t rows_off = bias_off + 8 * 1024;
    const size_t total = rows_off + 1024 + 65536;
    IRON_HIP_TRY(hipMalloc(&net->w16_blob, total));
    IRON_HIP_TRY(hipMemsetAsync(net->w16_blob, 0, total, st));
    char* base = (char*)net->w16_blob;
    size_t q = 0;
    auto slot_ptr = [&](size_t idx) { return (_Float16*)(base + table[2 * idx]); };
    for (int to = 0; to < 8; ++to, ++q)
        hipLaunchKernelGGL(k_pack_w16_head, dim3(8), dim3(256), 0, st, slot_ptr(q), make_pack_src(L[0], scale_base + soff[0], kHidden, 0, 1.0f), to, 0, pe);
    for (int l = 1; l <= nl - 2; ++l) {
        const bool is_skip = (l == skip);
        const float mul = is_skip ? kInvSqrt2 : 1.0f;
        const int cols_valid = is_skip ? kHidden - pe : kHidden;
        for (int to = 0; to < 8; ++to) {
            if (is_skip) {
                hipLaunchKernelGGL(k_pack_w16_head, dim3(8), dim3(256), 0, st, slot_ptr(q), make_pack_src(L[l], scale_base + soff[l], L[l].out_dim, 0, mul), to, kHidden - pe, pe);
                ++q;
            }
            hipLaunchKernelGGL(k_pack_w16_hidden, dim3(32), dim3(256), 0, st, slot_ptr(q), make_pack_src(L[l], scale_base + soff[l], L[l].out_dim, 0, mul), to, cols_valid);
            ++q;
        }
    }
    for (int l = 0; l <= nl - 2; ++l)
        hipLaunchKernelGGL(k_pack_w16_vec, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)l * 1024), L[l].bias, (const float*)nullptr, L[l].out_dim);
    hipLaunchKernelGGL(k_pack_w16_vec, dim3(1), dim3(256), 0, st, (float*)(base + rows_off), L[nl - 1].weight_v, scale_base + soff[nl - 1], kHidden);
    IRON_HIP_TRY(hipGetLastError());
    IRON_HIP_TRY(hipStreamSynchronize(st));
    H2StreamDev s;
    s.base = base; s.table_off = (uint32_t)table_off; s.n_slots = n_slots; s.bias_off = (uint32_t)bias_off;
    s.rows_off = (uint32_t)rows_off; s.n_bias_layers = (uint32_t)(nl - 1);
    for (int i = 0; i < 4; ++i) s.kind_mask[i] = 0;
    for (size_t k = 0; k < table.size() / 2; ++k)
        if (table[2 * k + 1]) s.kind_mask[k >> 5] |= 1u << (k & 31);
    net->w16_trace = s;
    return IRON_OK;
}

// ---- device side ----------------------------------------------------------------------------------------------------------------
struct W16Ring {
    const char* gbase;
    char* lds;
    unsigned long long mask_lo, mask_hi;
    int n_slots, q_issue, b_issue, b_take;
    uint32_t off_issue;
    int wave, lane;

    __device__ __forceinline__ bool kind_of(int q) const { return ((q < 64 ? mask_lo : mask_hi) >> (q & 63)) & 1ull; }
    // this wave's 4 of the slot's 32 DMA pieces
    __device__ __forceinline__ void issue() {
#ifdef IRON_W16_VARIANT
        if (IRON_W16_VARIANT & 1) return;   // timing experiment: no DMA (results are garbage)
#endif
        const bool hidden = kind_of(q_issue);
        const uint32_t wr_lds = lds_addr_of(lds) + b_issue * kSlotBytes;
        const char* src = gbase + off_issue + (hidden ? lane * 16 : lane * 4);
        if (hidden) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = wave + 8 * i;
                lds_dma16(src + f * 1024, wr_lds + f * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = wave + 8 * i;
                lds_dma4(src + f * 256, wr_lds + f * 256);
            }
        }
        off_issue += hidden ? (uint32_t)kSlotBytes : 8192u;
        if (++q_issue == n_slots) { q_issue = 0; off_issue = 0; }
        b_issue = (b_issue + 1) & 3;
    }
    // wait for this wave's pieces of the slot about to be consumed (the 8 of the two younger slots stay in flight), then meet
    __device__ __forceinline__ const char* sync_take() {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int bt = b_take;
        asm volatile("" : "+s"(bt));
        const char* rd = lds + bt * kSlotBytes;
        b_take = (b_take + 1) & 3;
        return rd;
    }
    __device__ __forceinline__ void drain() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
};

__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct W16Frag {   // one B fragment (32 input features x this lane's point), both pieces
    u32x4 h, l;
};

// combine, activation, split of one slot's two C tiles -> one B fragment of the next layer
template <bool FAST>
__device__ __forceinline__ void w16_epilogue(const f32x4 (&ahi)[2], const f32x4 (&alo)[2], W16Frag& out) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = fmaf(alo[u][r], kLoInv, ahi[u][r]);
#ifdef IRON_W16_VARIANT
            if (IRON_W16_VARIANT & 2) { v[4 * u + r] = z; continue; }   // timing experiment: no activation
#endif
            constexpr float kC1 = 144.26950408889634f, kC2 = 0.0069314718055994531f;
            const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(z) * kC1);
            v[4 * u + r] = fmaf(__builtin_amdgcn_logf(1.0f + e), kC2, fmaxf(z, 0.0f));
        }
    split8(v, *reinterpret_cast<half8*>(&out.h), *reinterpret_cast<half8*>(&out.l));
}

// A fragment pair (hi, lo) of group q = ks * 2 + u of the slot
__device__ __forceinline__ void w16_frag(const char* __restrict__ rd, int q, int lane, half8& wh, half8& wl) {
    wh = *reinterpret_cast<const half8*>(rd + (q * 2) * 1024 + lane * 16);
    wl = *reinterpret_cast<const half8*>(rd + (q * 2 + 1) * 1024 + lane * 16);
}

// the three products of one group; the two updates of acc_lo are kept one MFMA apart
#if defined(IRON_W16_VARIANT) && (IRON_W16_VARIANT & 4)   // timing experiment: one MFMA of three
#define W16_GROUP(WH, WL, BH, BL, U)                       \
    alo[U] = mfma16(WH, BL, alo[U]);                       \
    asm volatile("" :: "v"(WL), "v"(BH));
#else
#define W16_GROUP(WH, WL, BH, BL, U)                       \
    alo[U] = mfma16(WH, BL, alo[U]);                       \
    ahi[U] = mfma16(WH, BH, ahi[U]);                       \
    alo[U] = mfma16(WL, BH, alo[U]);
#endif

// the slot's 48 (hidden) MFMAs: acc[u] (+)= W[32 rows of the slot, :] * in.  16 groups of 3 MFMAs; the A fragments of group q + 2
// are requested before the MFMAs of group q (hipcc, left alone, reads a group's two fragments and waits lgkmcnt(0) right in front
// of its MFMAs: one exposed LDS latency per 48 cycles of matrix work).  sched_barrier pins the order.
__device__ __forceinline__ void w16_mma_hidden(const char* __restrict__ rd, int lane, const W16Frag (&in)[8], f32x4 (&ahi)[2], f32x4 (&alo)[2]) {
    half8 h0, l0, h1, l1, h2, l2;
    w16_frag(rd, 0, lane, h0, l0);
    w16_frag(rd, 1, lane, h1, l1);
#pragma unroll
    for (int q = 0; q < 16; q += 3) {
        {
            if (q + 2 < 16) w16_frag(rd, q + 2, lane, h2, l2);
            __builtin_amdgcn_sched_barrier(0);
            const half8 bh = __builtin_bit_cast(half8, in[q >> 1].h), bl = __builtin_bit_cast(half8, in[q >> 1].l);
            if ((q & 1) == 0) { W16_GROUP(h0, l0, bh, bl, 0) } else { W16_GROUP(h0, l0, bh, bl, 1) }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 1 < 16) {
            if (q + 3 < 16) w16_frag(rd, q + 3, lane, h0, l0);
            __builtin_amdgcn_sched_barrier(0);
            const half8 bh = __builtin_bit_cast(half8, in[(q + 1) >> 1].h), bl = __builtin_bit_cast(half8, in[(q + 1) >> 1].l);
            if (((q + 1) & 1) == 0) { W16_GROUP(h1, l1, bh, bl, 0) } else { W16_GROUP(h1, l1, bh, bl, 1) }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 2 < 16) {
            if (q + 4 < 16) w16_frag(rd, q + 4, lane, h1, l1);
            __builtin_amdgcn_sched_barrier(0);
            const half8 bh = __builtin_bit_cast(half8, in[(q + 2) >> 1].h), bl = __builtin_bit_cast(half8, in[(q + 2) >> 1].l);
            if (((q + 2) & 1) == 0) { W16_GROUP(h2, l2, bh, bl, 0) } else { W16_GROUP(h2, l2, bh, bl, 1) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void w16_mma_head(const char* __restrict__ rd, int lane, const W16Frag (&pe)[2], f32x4 (&ahi)[2], f32x4 (&alo)[2]) {
    half8 h0, l0, h1, l1, h2, l2, h3, l3;
    w16_frag(rd, 0, lane, h0, l0);
    w16_frag(rd, 1, lane, h1, l1);
    w16_frag(rd, 2, lane, h2, l2);
    w16_frag(rd, 3, lane, h3, l3);
    __builtin_amdgcn_sched_barrier(0);
    const half8 b0h = __builtin_bit_cast(half8, pe[0].h), b0l = __builtin_bit_cast(half8, pe[0].l);
    const half8 b1h = __builtin_bit_cast(half8, pe[1].h), b1l = __builtin_bit_cast(half8, pe[1].l);
    W16_GROUP(h0, l0, b0h, b0l, 0)
    W16_GROUP(h1, l1, b0h, b0l, 1)
    W16_GROUP(h2, l2, b1h, b1l, 0)
    W16_GROUP(h3, l3, b1h, b1l, 1)
}

// bias of the slot's rows owned by this lane-group: C tile u, rows 4g..4g+3 = features 32 to + 16 u + 4 g + r
__device__ __forceinline__ void w16_bias(const char* __restrict__ lds, int layer, int to, int g, f32x4 (&ahi)[2], f32x4 (&alo)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        ahi[u] = *reinterpret_cast<const f32x4*>(lds + kW16LdsBias + layer * 1024 + (32 * to + 16 * u + 4 * g) * 4);
        alo[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// PE-6 of this lane's point into its two head fragments: element i of k-step s = PE column 32 s + 8 g + i
__device__ __forceinline__ void w16_pe(float x, float y, float z, int g, W16Frag (&pe)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 32 * s + 8 * g + i;
            float val = 0.0f;
            if (p < 3) val = p == 0 ? x : (p == 1 ? y : z);
            else if (p < 39) {
                const int q = p - 3, k = q / 6, r = q % 6, c = r % 3;
                const float a = (c == 0 ? x : (c == 1 ? y : z)) * (float)(1 << k);
                val = r < 3 ? sinf(a) : cosf(a);
            }
            v[i] = val;
        }
        split8(v, *reinterpret_cast<half8*>(&pe[s].h), *reinterpret_cast<half8*>(&pe[s].l));
    }
}

// One SDF evaluation (8 softplus layers, skip at 4, then the 256 -> 1 row) of this wave's 16 points; all 8 waves call it together.
// LATE = false: [MFMA t][epilogue t];  LATE = true: [epilogue t-1][MFMA t] (the other wave of the SIMD).
template <bool FAST, bool LATE>
__device__ __forceinline__ float w16_sdf_eval(W16Ring& ring, const char* lds, float scale, float b_last, float x, float y, float z) {
    const int lane = ring.lane, g = lane >> 4;
    W16Frag pe[2];
    w16_pe(x * scale, y * scale, z * scale, g, pe);
    W16Frag X[8], Y[8];
    f32x4 ahi[2], alo[2];     // accumulators of the slot in flight
    f32x4 phi[2], plo[2];     // LATE: the previous slot's, awaiting their epilogue
    float dot = 0.0f;
    // ---- layer 0: head only
#pragma unroll
    for (int to = 0; to < 8; ++to) {
        const char* rd = ring.sync_take();
        if (LATE) { if (to > 0) w16_epilogue<FAST>(phi, plo, X[to - 1]); ring.issue(); }
        w16_bias(lds, 0, to, g, ahi, alo);
        w16_mma_head(rd, lane, pe, ahi, alo);
        if (LATE) { phi[0] = ahi[0]; phi[1] = ahi[1]; plo[0] = alo[0]; plo[1] = alo[1]; }
        else { w16_epilogue<FAST>(ahi, alo, X[to]); ring.issue(); }
    }
    // ---- layers 1..6 (ping-pong X -> Y -> X), the skip layer 4 with a head slot in front of every hidden slot
#define W16_LAYER(LAYER, IN, OUT, PREV_OUT, HEAD)                                                        \
    _Pragma("unroll") for (int to = 0; to < 8; ++to) {                                                  \
        const char* rd = ring.sync_take();                                                               \
        if (LATE) { if (to > 0) w16_epilogue<FAST>(phi, plo, OUT[to - 1]); else w16_epilogue<FAST>(phi, plo, PREV_OUT[7]); ring.issue(); } \
        w16_bias(lds, LAYER, to, g, ahi, alo);                                                           \
        if (HEAD) {                                                                                      \
            w16_mma_head(rd, lane, pe, ahi, alo);                                                        \
            if (!LATE) ring.issue();                                                                     \
            rd = ring.sync_take();                                                                       \
            if (LATE) ring.issue();                                                                      \
        }                                                                                                \
        w16_mma_hidden(rd, lane, IN, ahi, alo);                                                          \
        if (LATE) { phi[0] = ahi[0]; phi[1] = ahi[1]; plo[0] = alo[0]; plo[1] = alo[1]; }                \
        else { w16_epilogue<FAST>(ahi, alo, OUT[to]); ring.issue(); }                                    \
    }
    W16_LAYER(1, X, Y, X, false)
    W16_LAYER(2, Y, X, Y, false)
    W16_LAYER(3, X, Y, X, false)
    W16_LAYER(4, Y, X, Y, true)
    W16_LAYER(5, X, Y, X, false)
    W16_LAYER(6, Y, X, Y, false)
#undef W16_LAYER
    // ---- layer 7: the activation goes straight into the dot with the output row
    const float* wrow = reinterpret_cast<const float*>(lds + kW16LdsRow);
    auto finish = [&](const f32x4 (&hi)[2], const f32x4 (&lo)[2], int to) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 32 * to + 16 * u + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = fmaf(lo[u][r], kLoInv, hi[u][r]);
                constexpr float kC1 = 144.26950408889634f, kC2 = 0.0069314718055994531f;
                const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(zz) * kC1);
                dot = fmaf(fmaf(__builtin_amdgcn_logf(1.0f + e), kC2, fmaxf(zz, 0.0f)), w[r], dot);
            }
        }
    };
#pragma unroll
    for (int to = 0; to < 8; ++to) {
        const char* rd = ring.sync_take();
        if (LATE) { if (to > 0) finish(phi, plo, to - 1); else w16_epilogue<FAST>(phi, plo, X[7]); ring.issue(); }
        w16_bias(lds, 7, to, g, ahi, alo);
        w16_mma_hidden(rd, lane, X, ahi, alo);
        if (LATE) { phi[0] = ahi[0]; phi[1] = ahi[1]; plo[0] = alo[0]; plo[1] = alo[1]; }
        else { finish(ahi, alo, to); ring.issue(); }
    }
    if (LATE) finish(phi, plo, 7);
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    return (dot + b_last) / scale;
}

__global__ __launch_bounds__(512, 2) void k_sdf_values_w16(H2StreamDev s, H2Meta m, const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // biases and the output row -> LDS
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s.base + s.bias_off);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds + kW16LdsBias);
        for (int i = tid; i < (9 * 1024) / 4; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    W16Ring ring;
    ring.gbase = s.base; ring.lds = lds; ring.n_slots = (int)s.n_slots;
    ring.mask_lo = (unsigned long long)s.kind_mask[0] | ((unsigned long long)s.kind_mask[1] << 32);
    ring.mask_hi = (unsigned long long)s.kind_mask[2] | ((unsigned long long)s.kind_mask[3] << 32);
    ring.q_issue = 0; ring.off_issue = 0; ring.b_issue = 0; ring.b_take = 0; ring.wave = wave; ring.lane = lane;
    for (int i = 0; i < kW16Ahead; ++i) ring.issue();
    const int64_t n_groups = (n + 127) / 128;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t idx = grp * 128 + wave * 16 + (lane & 15);
        const bool ok = idx < n;
        const int64_t srcp = ok ? idx : (n - 1);
        const float px = x[srcp * 3 + 0], py = x[srcp * 3 + 1], pz = x[srcp * 3 + 2];
        float v;
        if (wave < 4) v = w16_sdf_eval<true, false>(ring, lds, m.scale, m.b_last, px, py, pz);
        else v = w16_sdf_eval<true, true>(ring, lds, m.scale, m.b_last, px, py, pz);
        if (ok && lane < 16) out[idx] = v;
    }
    ring.drain();
}

bool use_w16_core() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("IRON_MLP_CORE");
        v = (e && e[0] == 'w') ? 1 : 0;
    }
    return v == 1;
}

int launch_sdf_values_w16(const iron_net* net, const float* x, int64_t n, float* out, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_sdf_values_w16, hipFuncAttributeMaxDynamicSharedMemorySize, kW16Lds));
        attr = true;
    }
    H2Meta m;
    m.n_hidden_layers = net->sdf.n_hidden_layers; m.skip_layer = net->sdf.skip_layer; m.scale = net->sdf.scale; m.b_last = net->sdf.b_last;
    const int64_t groups = (n + 127) / 128;
    const unsigned grid = (unsigned)(groups < 256 ? groups : 256);
    hipLaunchKernelGGL(k_sdf_values_w16, dim3(grid), dim3(512), kW16Lds, st, net->w16_trace, m, x, n, out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

}  // namespace iron
