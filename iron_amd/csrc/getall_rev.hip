// SDFNetwork.get_all (models/fields.py:120-137) on the h2 core as forward + ONE reverse sweep (mlp_h2_rev.h):
// a workgroup = 4 waves x 32 points walks the 140-slot stream [72 hidden-stack slots | 8 feature-row slots | 60 transposed slots]
// built by pack_h2.hip (build_h2_sdf_rev).  Per 128 points: 140 ring steps, against 4 x 80 for the forward-mode kernel
// (shade.hip: k_sdf_grad_h2, which stays as the fallback and for the silhouette walk).
#include <stdlib.h>
#include "mlp_h2_rev.h"
#include "h2_setup.h"
#include "shade_args.h"

namespace iron {

// d sdf / d(x, y, z) of this lane-half from the accumulators of a transposed PE tile: row (r, half) of tile T is head slot
// 16 T + r in the ROLE of the other half (the row of sin(2^k v_c) sits in the half that holds cos(2^k v_c), which is its
// derivative up to the factor 2^k, and vice versa with a minus sign); pack_h2.hip: k_pack_h2_pe_T.
template <int T>
__device__ __forceinline__ void pe_contract(const f32x16& g, const float* pe, int half, float& gx, float& gy, float& gz) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int s = 16 * T + r;
        if (s == 0) {
            if (half) gy += g[r]; else gx += g[r];
        } else if (s == 1) {
            if (!half) gz += g[r];
        } else if (s < 2 + 3 * kSdfPeLevels) {
            const int k = (s - 2) / 3, c = (s - 2) % 3;
            const float f = (float)(1 << k);
            const float v = (half ? f : -f) * pe[s] * g[r];
            if (c == 0) gx += v; else if (c == 1) gy += v; else gz += v;
        }
    }
}

#ifndef IRON_REV_FEAT_NT
#define IRON_REV_FEAT_NT 0
#endif
#ifdef IRON_REV_DEBUG   // diagnostic build (tools/diag_getall_rev.py): feat_rows receives d_l = d sdf / d z_l of layer `dbg` instead of the features
#define IRON_REV_DUMP(L, BUF)                                                                                              \
    if (dbg == (L) && a.feat_rows && ok) {                                                                                 \
        _Pragma("unroll") for (int t_ = 0; t_ < kHidTiles; ++t_)                                                           \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                            \
                const int s_ = r_ >> 3, j_ = r_ & 7;                                                                       \
                const float v_ = (float)BUF[t_].h[s_][j_] + (float)BUF[t_].l[s_][j_] * kLoInv;                             \
                a.feat_rows[(size_t)li * kHidden + 32 * t_ + (r_ & 3) + 8 * (r_ >> 2) + 4 * half] = v_;                    \
            }                                                                                                              \
    }
#define IRON_REV_DBG_PARAM , int dbg
#else
#define IRON_REV_DUMP(L, BUF)
#define IRON_REV_DBG_PARAM
#endif

__global__ __launch_bounds__(256, 1) void k_sdf_getall_rev_h2(H2StreamDev hs, H2Meta m, GradArgs a, char* __restrict__ park IRON_REV_DBG_PARAM) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds = smem;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ring ring;
    h2_setup(hs, lds, ring);
    ParkBuf pb;
    pb.rsrc = __builtin_amdgcn_make_buffer_rsrc(park + (size_t)blockIdx.x * kParkBytesPerWg, 0, kParkBytesPerWg, 0x00020000);
    pb.voff = lane * 16;
    const int count = a.count_ptr ? *a.count_ptr : a.count;
    const int n_tiles = (count + kTile - 1) / kTile;
    const int n_groups = (n_tiles + 3) / 4;
    const bool want_feat = (a.feat_packed != nullptr) || (a.feat_rows != nullptr);
    const char* bias = lds + kLdsBias;

    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int tile = 4 * g + wave;
        const int li = tile * kTile + (lane & 31);
        const bool ok = li < count;
        const int src = ok ? (a.list ? a.list[li] : li) : 0;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (ok) { px = a.x[3 * (size_t)src]; py = a.x[3 * (size_t)src + 1]; pz = a.x[3 * (size_t)src + 2]; }
        const float sx = px * m.scale, sy = py * m.scale, sz = pz * m.scale;

        TileFrag X[kHidTiles], Y[kHidTiles];
        f32x16 hf[kHidTiles];
        HeadFrag hd;
        {
            float pe[kHeadSlots];
#pragma unroll
            for (int i = 0; i < kHeadSlots; ++i) pe[i] = 0.0f;
            head_fill<kSdfPeLevels>(sx, sy, sz, half, pe);
            split_head(pe, hd);
        }
        // ---- forward, parking sigma' ----------------------------------------------------------------------------------
        // layer 0: head only (9 MFMAs per tile: epilogue in place)
#pragma unroll
        for (int to = 0; to < kHidTiles; ++to) {
            ring.sync();
            const RingStep st = ring.step();
            f32x16 a_hi = zero16(), a_lo = zero16();
            step_head(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, to, true, hd, a_hi, a_lo);
            f32x16 h, P;
            softplus_park_tile(h2_combine(a_hi, a_lo), h, P);
            park_store_tile(pb, park_off(0, to, wave), P);
            split_tile(h, X[to]);
        }
        // layers (1,2), (3,4), (5,6): X -> Y -> X; layer 4 is the skip layer; then layer 7 -> f32 tiles
        for (int l = 1; l < 7; l += 2) {
            h2_layer_x<kModeFwd, false, false>(ring, bias + l * 1024, hd, lane, X, Y, hf, pb, l);
            if (l + 1 == 4) h2_layer_x<kModeFwd, true, false>(ring, bias + (l + 1) * 1024, hd, lane, Y, X, hf, pb, l + 1);
            else h2_layer_x<kModeFwd, false, false>(ring, bias + (l + 1) * 1024, hd, lane, Y, X, hf, pb, l + 1);
        }
        h2_layer_x<kModeFwd, false, true>(ring, bias + 7 * 1024, hd, lane, X, Y, hf, pb, 7);
        {
            const float s = (row_dot_lds(lds + kLdsRows, hf, half) + m.b_last) / m.scale;
            if (ok && lane < 32 && a.sdf_out) a.sdf_out[li] = s;
        }
        // ---- feature rows: 8 plain slots (always walked: the stream is one fixed sequence) --------------------------------
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) split_tile(hf[t], X[t]);
        {
            const char* fb = bias + 8 * 1024;
            float* dst = (a.feat_packed && tile < n_tiles) ? a.feat_packed + (size_t)tile * (kHidTiles * 16 * 64) : nullptr;
#pragma unroll
            for (int to = 0; to < kHidTiles; ++to) {
                const f32x16 o = h2_plain_tile(ring, fb, lane, to, true, X, pb);
                if (want_feat && !(IRON_REV_ABL & 4)) {
                    if (dst) feat_store_tile<IRON_REV_FEAT_NT != 0>(dst, to, lane, o);
#ifdef IRON_REV_DEBUG
                    if (a.feat_rows && ok && dbg < 0) {
#else
                    if (a.feat_rows && ok) {
#endif
#pragma unroll
                        for (int r = 0; r < 16; ++r) a.feat_rows[(size_t)li * kHidden + 32 * to + (r & 3) + 8 * (r >> 2) + 4 * half] = o[r];
                    }
                }
            }
        }
        // ---- reverse sweep ------------------------------------------------------------------------------------------------
        // d_7 = sigma'(z_7) * w_last  (the last layer's row 0 sits in LDS in register-tile order)
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            const f32x16 P = park_load_tile(pb, park_off(7, t, wave));
            f32x16 d = lds_half_tile(lds + kLdsRows, t, half);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] *= sigma_from_park(P[i]);
            split_tile(d, Y[t]);
        }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        // the PE values the Jacobian contraction needs, once for both PE stages (20 registers live through the sweep; a second
        // head_fill -- 18 sincosf -- cost 3.7 % of the kernel)
        float pe[kHeadSlots];
#pragma unroll
        for (int i = 0; i < kHeadSlots; ++i) pe[i] = 0.0f;
        if (!(IRON_REV_ABL & 8)) head_fill<kSdfPeLevels>(sx, sy, sz, half, pe);
        IRON_REV_DUMP(7, Y)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, Y, X, hf, pb, 6);   // W_7^T -> d_6
        IRON_REV_DUMP(6, X)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, X, Y, hf, pb, 5);   // W_6^T -> d_5
        IRON_REV_DUMP(5, Y)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, Y, X, hf, pb, 4);   // W_5^T -> d_4
        IRON_REV_DUMP(4, X)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, X, Y, hf, pb, 3);   // W_4[:, :217]^T -> d_3
        IRON_REV_DUMP(3, Y)
        {
            const f32x16 g0 = h2_plain_tile(ring, bias, lane, 0, false, X, pb);      // W_4[:, 217:]^T d_4: the skip's PE rows
            pe_contract<0>(g0, pe, half, gx, gy, gz);
            const f32x16 g1 = h2_plain_tile(ring, bias, lane, 1, false, X, pb);
            pe_contract<1>(g1, pe, half, gx, gy, gz);
        }
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, Y, X, hf, pb, 2);   // W_3^T -> d_2
        IRON_REV_DUMP(2, X)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, X, Y, hf, pb, 1);   // W_2^T -> d_1
        IRON_REV_DUMP(1, Y)
        h2_layer_x<kModeBwd, false, false>(ring, bias, hd, lane, Y, X, hf, pb, 0);   // W_1^T -> d_0
        IRON_REV_DUMP(0, X)
        {
            const f32x16 g0 = h2_plain_tile(ring, bias, lane, 0, false, X, pb);      // W_0^T d_0
            pe_contract<0>(g0, pe, half, gx, gy, gz);
            const f32x16 g1 = h2_plain_tile(ring, bias, lane, 1, false, X, pb);
            pe_contract<1>(g1, pe, half, gx, gy, gz);
        }
        gx += __shfl_xor(gx, 32, 64);
        gy += __shfl_xor(gy, 32, 64);
        gz += __shfl_xor(gz, 32, 64);
        if (ok && lane < 32 && a.grad_out) {
            a.grad_out[3 * (size_t)li] = gx; a.grad_out[3 * (size_t)li + 1] = gy; a.grad_out[3 * (size_t)li + 2] = gz;
        }
        // the tape block is reused by the next pass: this wave's own stores and loads are ordered by the memory system
        // (same lane, same address), nothing is shared between waves
    }
    ring.drain();
}

size_t getall_rev_park_bytes(int64_t n_points) {
    const int64_t groups = (n_points + 127) / 128;
    const int64_t cus = cu_total();
    return (size_t)(groups < cus ? (groups > 0 ? groups : 1) : cus) * kParkBytesPerWg;
}

bool getall_rev_usable(const iron_net* sdf) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("IRON_GETALL");   // "fwd" keeps the forward-mode (tangent) kernel
        on = (e && e[0] == 'f') ? 0 : 1;
    }
    return on == 1 && h2_sdf_usable(sdf) && sdf->h2_rev_blob != nullptr;
}

int launch_sdf_getall_rev(const iron_net* sdf, const GradArgs& a, int64_t max_tiles, void* park, size_t park_bytes, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        IRON_HIP_TRY(hipFuncSetAttribute((const void*)k_sdf_getall_rev_h2, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsH2Total));
        attr = true;
    }
    H2Meta m;
    m.n_hidden_layers = sdf->sdf.n_hidden_layers; m.skip_layer = sdf->sdf.skip_layer; m.scale = sdf->sdf.scale; m.b_last = sdf->sdf.b_last;
    const int64_t groups = (max_tiles + 3) / 4;
    const int64_t cus = cu_budget();
    int64_t grid = groups < cus ? (groups > 0 ? groups : 1) : cus;
    const int64_t cap = (int64_t)(park_bytes / kParkBytesPerWg);
    if (!park || cap < 1) return IRON_ERR_WORKSPACE;
    if (grid > cap) grid = cap;
    if (((uintptr_t)park & 15) != 0) return IRON_ERR_BAD_ARG;
    ProfScope ps(IRON_PROF_SDF_GRAD, st);
#ifdef IRON_REV_DEBUG
    const char* de = getenv("IRON_REV_DEBUG_LAYER");
    hipLaunchKernelGGL(k_sdf_getall_rev_h2, dim3((unsigned)grid), dim3(256), kLdsH2Total, st, sdf->h2_rev, m, a, (char*)park, de ? atoi(de) : -1);
#else
    hipLaunchKernelGGL(k_sdf_getall_rev_h2, dim3((unsigned)grid), dim3(256), kLdsH2Total, st, sdf->h2_rev, m, a, (char*)park);
#endif
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

}  // namespace iron
