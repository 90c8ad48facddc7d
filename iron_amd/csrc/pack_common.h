// Shared declarations of the weight packers (pack.hip, pack_h2.hip).
#pragma once
#include "iron_common.h"

namespace iron {

struct PackSrc {
    const float* w;      // [rows, ld] row-major weight_v
    const float* scale;  // per-row fold factor
    int ld;
    int rows_valid;      // rows >= this are zero padding
    int row_off;         // first source row
    float mul;           // extra factor (1/sqrt(2) for the skip layer)
};

struct HeadSrcs {
    int n;
    int slot_base[3];
    int levels[3];
    int col_off[3];
};

const float kInvSqrt2 = 1.0f / 1.41421356237309504880f;  // activations / np.sqrt(2) folded into W

inline PackSrc make_pack_src(const iron_linear& l, const float* scale, int rows_valid, int row_off, float mul) {
    PackSrc s;
    s.w = l.weight_v; s.scale = scale; s.ld = l.in_dim; s.rows_valid = rows_valid; s.row_off = row_off; s.mul = mul;
    return s;
}

__global__ void k_pack_bias(float* __restrict__ dst, const float* __restrict__ bias, int row_off, int rows_valid);
__global__ void k_pack_row(float* __restrict__ dst, PackSrc s, int row, int col_off, int cols_valid);

int build_h2_sdf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st);
int build_w16_sdf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st);
int build_h2_render(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, const HeadSrcs& hs,
                    int head_w, hipStream_t st);

int build_h2_nerf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st);

}  // namespace iron
