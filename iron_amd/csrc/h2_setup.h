// Kernel-side setup shared by every kernel on the h2 core.
#pragma once
#include "mlp_h2.h"

namespace iron {

struct H2Meta {
    int n_hidden_layers;
    int skip_layer;
    float scale;
    float b_last;
};

bool use_h2_core();   // (also declared in iron_common.h)

// The SDF stack on the h2 core (sdf_hidden_stack_h2) is written for the reference's 8 x 256 network with the skip at
// layer 4 (models/network_conf.py:31-44) and needs the h2 stream (absent when a folded weight overflows fp16); any
// other SDF network runs on the exact-fp32 core.
inline bool h2_sdf_usable(const iron_net* net) {
    return h2_enabled(net) && net->sdf.n_hidden_layers == 8 && net->sdf.skip_layer == 4;
}

// biases / output rows -> LDS, then start the weight ring.  Called once per kernel by all 256 threads.
__device__ __forceinline__ void h2_setup(const H2StreamDev& s, char* lds, Ring& ring) {
    const int tid = threadIdx.x;
    const uint32_t* src_b = reinterpret_cast<const uint32_t*>(s.base + s.bias_off);
    uint32_t* dst_b = reinterpret_cast<uint32_t*>(lds + kLdsBias);
    for (int i = tid; i < (kLdsBiasBytes + kLdsRowsBytes) / 4; i += 256) dst_b[i] = src_b[i];  // bias and rows are adjacent
    __syncthreads();
    ring_start(ring, s, lds, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
}

}  // namespace iron
