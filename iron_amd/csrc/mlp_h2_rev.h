// Reverse-mode input gradient of the SDF network on the h2 core (mlp_h2.h): the pieces get_all needs on top of the
// forward evaluation.
//
// SDFNetwork.get_all (models/fields.py:120-137) = forward (sdf + 256 features) + d sdf / d x.  The forward-mode form
// (shade.hip: k_sdf_grad_h2) spends four wave passes per point -- value + three tangents, all four waves of a workgroup on
// the SAME 32 points.  Here every wave owns its own 32 points (128 per workgroup and weight slot, as in the tracer) and the
// gradient is one reverse sweep:
//     forward   z_l = W_l h_{l-1} + b_l,  h_l = softplus_100(z_l)            -- parks P_l = +-(1 + exp(-|100 z_l|)) per unit
//     reverse   g_7 = w_last;  d_l = sigma'(z_l) * g_l;  g_{l-1} = W_l^T d_l   -- sigma'(z) = sigmoid(100 z) from P_l
//     input     grad_x = J_PE^T (W_0^T d_0 + W_4[:, 217:]^T d_4)
// The transposed layers are ordinary ring slots of a second half of the weight stream (pack_h2.hip: build_h2_sdf_rev), so
// the reverse sweep is the same MFMA pipeline with a different epilogue (times sigma' instead of softplus).
// The tape: 8 layers x 256 units x 4 B = 8 KiB per point, 1 MiB per workgroup pass, written and read back by the same lane
// in fully coalesced 16-byte pieces (each block is reused for every pass of the workgroup, so the footprint is one MiB per
// resident workgroup).  sigma' is parked as P = copysign(1 + u, z), u = exp(-|100 z|) -- one v_bfi in the forward
// epilogue -- and recovered as r = 1 / P, sigma' = r > 0 ? r : 1 + r (absolute error <= 1 ulp of 1).
#pragma once
#include "mlp_h2.h"

namespace iron {

constexpr int kParkTileBytes = 4096;                      // one wave's tile of 32 points x 32 units: [4 pieces][64 lanes][4 f32]
constexpr int kParkLayerBytes = kHidTiles * 4 * kParkTileBytes;   // [tile][wave]
constexpr int kParkLayers = 8;
constexpr int kParkBytesPerWg = kParkLayers * kParkLayerBytes;    // 1 MiB

struct ParkBuf {
    __amdgpu_buffer_rsrc_t rsrc;   // this workgroup's block
    int voff;                      // lane * 16
};

__device__ __forceinline__ int park_off(int layer, int tile, int wave) { return ((layer * kHidTiles + tile) * 4 + wave) * kParkTileBytes; }

#ifndef IRON_REV_TAPE_AUX
#define IRON_REV_TAPE_AUX 2   // cache policy bits of the tape's buffer stores / loads: 2 = nt.  The tape is written once and read once, 80+ ring
                              // steps later, by the same lane: with the default policy its 1 MiB per workgroup pass washes the weight stream's
                              // slots out of L2 (4.61 ms per 524 288 points; nt 4.08; sc0 alone 4.61)
#endif
#ifndef IRON_REV_ABL
#define IRON_REV_ABL 0   // timing ablations (garbage results): 1 no tape stores, 2 no tape loads, 4 no feature stores, 8 no PE recompute
#endif

__device__ __forceinline__ void park_store_piece(const ParkBuf& pb, int off, int piece, float a, float b, float c, float d) {
#if IRON_REV_ABL & 1
    asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d));
    return;
#endif
    u32x4 v;
    v[0] = __builtin_bit_cast(unsigned, a); v[1] = __builtin_bit_cast(unsigned, b);
    v[2] = __builtin_bit_cast(unsigned, c); v[3] = __builtin_bit_cast(unsigned, d);
    __builtin_amdgcn_raw_buffer_store_b128(v, pb.rsrc, pb.voff + piece * 1024, off, IRON_REV_TAPE_AUX);
}

__device__ __forceinline__ void park_store_tile(const ParkBuf& pb, int off, const f32x16& p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) park_store_piece(pb, off, q, p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
}

__device__ __forceinline__ f32x16 park_load_tile(const ParkBuf& pb, int off) {
    f32x16 p;
#if IRON_REV_ABL & 2
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = 1.5f + (float)off; asm volatile("" : "+v"(p[i])); }
    return p;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // (bit_cast of the builtin's result, as in mlp_core.h: assigning it to an ext_vector_type makes hipcc 7.2 load ONE dword and splat it)
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pb.rsrc, pb.voff + q * 1024, off, IRON_REV_TAPE_AUX));
#pragma unroll
        for (int i = 0; i < 4; ++i) p[4 * q + i] = v[i];
    }
    return p;
}

// sigma'(z) = sigmoid(100 z) from the parked P = copysign(1 + exp(-|100 z|), z)
__device__ __forceinline__ float sigma_from_park(float P) {
    const float r = __builtin_amdgcn_rcpf(P);
    return r > 0.0f ? r : 1.0f + r;
}

// softplus_100 of a tile (the v_exp / v_log form of mlp_core.h) together with the parked value; exposed (non-pipelined) form
__device__ __forceinline__ void softplus_park_tile(const f32x16& z, f32x16& h, f32x16& P) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float u = __builtin_amdgcn_exp2f(__builtin_fabsf(z[i]) * -144.26950408889634f);
        const float w = 1.0f + u;
        h[i] = __builtin_fmaf(__builtin_amdgcn_logf(w), 0.0069314718055994531f, relu_med3(z[i]));
        P[i] = __builtin_copysignf(w, z[i]);
    }
}

// MODE of a step / layer of the reverse-mode kernel
constexpr int kModeFwd = 0;    // softplus_100 epilogue that also parks P
constexpr int kModeBwd = 1;    // epilogue = times sigma' (read back from the tape)
constexpr int kModePlain = 2;  // no epilogue (feature rows, PE rows): the caller takes the accumulators

struct EpiStateX {
    float z[16], e[16], rr[16];
    unsigned hpb[8];
    u32x4 oh[2], ol[2];
};

// The staged epilogue of mlp_h2.h (epi_stage) with the two variants of this kernel; one stage per k-step, three parts per
// stage (the three MFMA gaps of the k-step).
//   kModeFwd: ... 4: w = 1 + u   5: P = copysign(w, z)   6,7: log   8: relu   9: fma   10: cvt hi   11: residual   12: scale
//             13: cvt lo;  the four 16-byte stores of P leave in stages 6..9 (part 1)
//   kModeBwd: 2,3: r = 1 / P   4: t = 1 + r   5,6: s = r > 0 ? r : t   7: z *= s   8: cvt hi   9: residual   10: scale   11: cvt lo
// EPI: 1 = split fragments out (next layer's B operand), 2 = f32 tile out (kModeFwd, last hidden layer).
template <int EPI, int MODE>
__device__ __forceinline__ void epi_stage_x(EpiStateX& st, int ks, int part, const f32x16& p_hi, const f32x16& p_lo, const f32x16& sp,
                                            const ParkBuf& pb, int off_pending) {
    constexpr float kC1 = 144.26950408889634f;            // 100 * log2(e)
    constexpr float kC2 = 0.0069314718055994531f;         // ln(2) / 100
    const int a16 = epi_lo(16, part), b16 = epi_hi(16, part), a8 = epi_lo(8, part), b8 = epi_hi(8, part);
    if (ks == 0) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = fmaf(p_lo[i], kLoInv, p_hi[i]); pin1(st.z[i]); } }
    constexpr int kSplit0 = MODE == kModeFwd ? 10 : 8;    // first stage of the fp16 split
    if constexpr (MODE == kModeFwd) {
        if (ks == 1) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.e[i] = __builtin_fabsf(st.z[i]) * -kC1; pin1(st.e[i]); } }
        if (ks == 2) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = __builtin_amdgcn_exp2f(st.e[i]); pin1(st.e[i]); } }
        if (ks == 3) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = __builtin_amdgcn_exp2f(st.e[i]); pin1(st.e[i]); } }
        if (ks == 4) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.e[i] = 1.0f + st.e[i]; pin1(st.e[i]); } }
        if (ks == 5) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.rr[i] = __builtin_copysignf(st.e[i], st.z[i]); pin1(st.rr[i]); } }
        if (ks >= 6 && ks <= 9 && part == 1) {
            const int q = ks - 6;
            park_store_piece(pb, off_pending, q, st.rr[4 * q], st.rr[4 * q + 1], st.rr[4 * q + 2], st.rr[4 * q + 3]);
        }
        if (ks == 6) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = __builtin_amdgcn_logf(st.e[i]); pin1(st.e[i]); } }
        if (ks == 7) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = __builtin_amdgcn_logf(st.e[i]); pin1(st.e[i]); } }
        if (ks == 8) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = relu_med3(st.z[i]); pin1(st.z[i]); } }
        if (ks == 9) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = __builtin_fmaf(st.e[i], kC2, st.z[i]); pin1(st.z[i]); } }
    } else if constexpr (MODE == kModeBwd) {
        if (ks == 2) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = __builtin_amdgcn_rcpf(sp[i]); pin1(st.e[i]); } }
        if (ks == 3) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = __builtin_amdgcn_rcpf(sp[i]); pin1(st.e[i]); } }
        if (ks == 4) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.rr[i] = 1.0f + st.e[i]; pin1(st.rr[i]); } }
        if (ks == 5) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = st.e[i] > 0.0f ? st.e[i] : st.rr[i]; pin1(st.e[i]); } }
        if (ks == 6) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = st.e[i] > 0.0f ? st.e[i] : st.rr[i]; pin1(st.e[i]); } }
        if (ks == 7) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = st.z[i] * st.e[i]; pin1(st.z[i]); } }
    }
    if constexpr (EPI == 1) {
        if (ks == kSplit0) {
            _Pragma("unroll") for (int q = a8; q < b8; ++q) {
                st.hpb[q] = __builtin_bit_cast(unsigned, cvt_pk_rn(st.z[2 * q], st.z[2 * q + 1]));
                pin1u(st.hpb[q]);
            }
        }
        if (ks == kSplit0 + 1) {
            _Pragma("unroll") for (int i = a16; i < b16; ++i) {
                st.rr[i] = (i & 1) ? residual_hi(st.hpb[i >> 1], st.z[i]) : residual_lo(st.hpb[i >> 1], st.z[i]);
                pin1(st.rr[i]);
            }
        }
        if (ks == kSplit0 + 2) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.rr[i] = st.rr[i] * kLoScale; pin1(st.rr[i]); } }
        if (ks == kSplit0 + 3) {
            _Pragma("unroll") for (int q = a8; q < b8; ++q) {
                unsigned lp = __builtin_bit_cast(unsigned, cvt_pk_rn(st.rr[2 * q], st.rr[2 * q + 1]));
                pin1u(lp);
                st.oh[q >> 2][q & 3] = st.hpb[q];
                st.ol[q >> 2][q & 3] = lp;
            }
        }
    }
}

// One ring step on a hidden slot (mlp_h2.h: step_hidden) for the reverse-mode kernel: 48 MFMAs on `in`, the pending tile's
// epilogue (EPI != 0) between them.  kModeBwd: `sp` holds the pending tile's parked values on entry (consumed by k-step 3) and is
// refilled from the tape with THIS step's tile at k-step 4 (`off_cur` >= 0), a whole ring step before it is needed.
template <int MODE, int EPI>
__device__ __forceinline__ void step_hidden_x(const char* __restrict__ rd, const char* __restrict__ bias, char* __restrict__ wr,
                                              const RingSrc& src, bool src_hidden, int wave, int lane, int tile, bool add_bias,
                                              TileFrag (&in)[kHidTiles], f32x16& acc_hi, f32x16& acc_lo,
                                              const f32x16& p_hi, const f32x16& p_lo, TileFrag& out_prev, f32x16& hf_prev,
                                              f32x16& sp, const ParkBuf& pb, int off_pending, int off_cur) {
    static_assert(MODE != kModePlain || EPI == 0, "a plain step has no epilogue");
    static_assert(EPI != 2 || MODE == kModeFwd, "f32 tiles leave the forward only");
    if (add_bias) acc_hi = lds_half_tile(bias, tile, lane >> 5);
    half8 fhs[2], fls[2];
    fhs[0] = lds_frag(rd, 0, lane);
    fls[0] = lds_frag(rd, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    dma_issue(src, wr, src_hidden, wave);
    __builtin_amdgcn_sched_barrier(0);
    EpiStateX es;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + 1 < 16) {
            fhs[(ks + 1) & 1] = lds_frag(rd, 2 * (ks + 1), lane);
            fls[(ks + 1) & 1] = lds_frag(rd, 2 * (ks + 1) + 1, lane);
        }
        const half8 fh = fhs[ks & 1], fl = fls[ks & 1];
        const int ti = ks >> 1, s = ks & 1;
        if constexpr (MODE == kModeBwd) {
            if (ks == 4 && off_cur >= 0) sp = park_load_tile(pb, off_cur);
        }
        acc_hi = mfma_h(fh, in[ti].h[s], acc_hi);
        if constexpr (EPI != 0) { epi_stage_x<EPI, MODE>(es, ks, 0, p_hi, p_lo, sp, pb, off_pending); __builtin_amdgcn_sched_barrier(0); }
        acc_lo = mfma_h(fh, in[ti].l[s], acc_lo);
        if constexpr (EPI != 0) { epi_stage_x<EPI, MODE>(es, ks, 1, p_hi, p_lo, sp, pb, off_pending); __builtin_amdgcn_sched_barrier(0); }
        acc_lo = mfma_h(fl, in[ti].h[s], acc_lo);
        if constexpr (EPI != 0) { epi_stage_x<EPI, MODE>(es, ks, 2, p_hi, p_lo, sp, pb, off_pending); }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EPI == 1) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            asm volatile("" : "+a"(es.oh[s2]), "+a"(es.ol[s2]));
            out_prev.h[s2] = __builtin_bit_cast(half8, es.oh[s2]);
            out_prev.l[s2] = __builtin_bit_cast(half8, es.ol[s2]);
        }
    }
    if constexpr (EPI == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) hf_prev[i] = es.z[i];
        asm volatile("" : "+v"(hf_prev));
    }
}

// One 256 -> 256 layer of the reverse-mode kernel on the ring (mlp_h2.h: h2_hidden_layer without the tile deferral: the last
// tile's epilogue is exposed at the layer boundary).
//   kModeFwd: out = softplus_100(W in + b), P parked as tape layer `lp`; HEAD: the layer also has a head product (skip layer);
//             LAST: the result is delivered as f32 tiles in `hf`.
//   kModeBwd: out = sigma'(tape layer lp) * (W^T in)   (no bias)
template <int MODE, bool HEAD, bool LAST>
__device__ __forceinline__ void h2_layer_x(Ring& ring, const char* bias, const HeadFrag& hd, int lane, TileFrag (&in)[kHidTiles],
                                           TileFrag (&out)[kHidTiles], f32x16 (&hf)[kHidTiles], const ParkBuf& pb, int lp) {
    static_assert(MODE == kModeFwd || (!HEAD && !LAST), "reverse layers are plain products");
    const int wave = ring.wave;
    f32x16 acc[2][2];
    f32x16 sp = zero16();
    TileFrag dummy_out;
    f32x16 dummy_hf;
#define IRON_X_TILE(TO)                                                                                                      \
    {                                                                                                                        \
        constexpr int P = (TO) & 1, Q = P ^ 1;                                                                               \
        acc[P][0] = zero16();                                                                                                \
        acc[P][1] = zero16();                                                                                                \
        if constexpr (HEAD) {                                                                                                \
            ring.sync();                                                                                                     \
            const RingStep sh = ring.step();                                                                                 \
            step_head(sh.rd, bias, sh.wr, sh.src, sh.hidden, wave, lane, TO, true, hd, acc[P][0], acc[P][1]);                \
        }                                                                                                                    \
        ring.sync();                                                                                                         \
        const RingStep st = ring.step();                                                                                     \
        const int off_cur = MODE == kModeBwd ? park_off(lp, TO, wave) : -1;                                                  \
        const int off_pen = park_off(lp, (TO) > 0 ? (TO) - 1 : 0, wave);                                                     \
        constexpr bool kBias = MODE == kModeFwd && !HEAD;                                                                    \
        if constexpr ((TO) == 0)                                                                                             \
            step_hidden_x<MODE, 0>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, kBias, in, acc[P][0], acc[P][1],   \
                                   acc[Q][0], acc[Q][1], dummy_out, dummy_hf, sp, pb, off_pen, off_cur);                     \
        else if constexpr (LAST)                                                                                             \
            step_hidden_x<MODE, 2>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, kBias, in, acc[P][0], acc[P][1],   \
                                   acc[Q][0], acc[Q][1], dummy_out, hf[(TO) > 0 ? (TO) - 1 : 0], sp, pb, off_pen, off_cur);  \
        else                                                                                                                 \
            step_hidden_x<MODE, 1>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, kBias, in, acc[P][0], acc[P][1],   \
                                   acc[Q][0], acc[Q][1], out[(TO) > 0 ? (TO) - 1 : 0], dummy_hf, sp, pb, off_pen, off_cur);  \
    }
    IRON_X_TILE(0) IRON_X_TILE(1) IRON_X_TILE(2) IRON_X_TILE(3) IRON_X_TILE(4) IRON_X_TILE(5) IRON_X_TILE(6) IRON_X_TILE(7)
#undef IRON_X_TILE
    // the last tile's epilogue, exposed
    f32x16 z = h2_combine(acc[1][0], acc[1][1]);
    if constexpr (MODE == kModeFwd) {
        f32x16 h, P;
        softplus_park_tile(z, h, P);
        park_store_tile(pb, park_off(lp, kHidTiles - 1, wave), P);
        if constexpr (LAST) hf[kHidTiles - 1] = h;
        else split_tile(h, out[kHidTiles - 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] *= sigma_from_park(sp[i]);
        split_tile(z, out[kHidTiles - 1]);
    }
}

// One plain ring step: acc = W[tile, :] in  (feature rows of the last layer; transposed PE rows of the reverse sweep)
__device__ __forceinline__ f32x16 h2_plain_tile(Ring& ring, const char* bias, int lane, int tile, bool add_bias, TileFrag (&in)[kHidTiles],
                                                const ParkBuf& pb) {
    f32x16 a_hi = zero16(), a_lo = zero16(), sp = zero16();
    TileFrag dummy_out;
    f32x16 dummy_hf;
    ring.sync();
    const RingStep st = ring.step();
    step_hidden_x<kModePlain, 0>(st.rd, bias, st.wr, st.src, st.hidden, ring.wave, lane, tile, add_bias, in, a_hi, a_lo, a_hi, a_lo,
                                 dummy_out, dummy_hf, sp, pb, 0, -1);
    return h2_combine(a_hi, a_lo);
}

}  // namespace iron
