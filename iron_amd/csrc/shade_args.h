// Argument blocks shared by the get_all kernels (shade.hip, getall_rev.hip).
#pragma once
#include "iron_common.h"

namespace iron {

struct GradArgs {
    const float* x;        // [*,3] point source
    const int* list;       // hit list (indices into x) or null = identity
    const int* count_ptr;  // device count or null
    int count;             // used when count_ptr == null
    float* feat_packed;    // [tiles][8][4][64][4] (mlp_core.h: feat_store_tile) or null
    float* sdf_out;        // [count] (list order) or null
    float* grad_out;       // [count,3] (list order) or null
    float* feat_rows;      // [count,256] row-major or null
};

// reverse-mode get_all (getall_rev.hip): forward + one reverse sweep, 128 points per workgroup; needs a tape workspace
bool getall_rev_usable(const iron_net* sdf);
size_t getall_rev_park_bytes(int64_t n_points);
int launch_sdf_getall_rev(const iron_net* sdf, const GradArgs& a, int64_t max_tiles, void* park, size_t park_bytes, hipStream_t st);

}  // namespace iron
