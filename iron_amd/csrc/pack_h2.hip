// Packing for the h2 core (mlp_h2.h): folded fp32 weights -> split fp16 A-fragments of
// v_mfma_f32_32x32x16_f16, laid out as the linear slot sequence the LDS ring streams.
#include <string.h>
#include <vector>
#include "mlp_h2.h"
#include "pack_common.h"

namespace iron {

// element j of lane (i, h) in the fragment of k-step ks (input tile ti = ks>>1, sub-step s = ks&1) multiplies
// input feature 32*ti + 16*s + 8*(j>>2) + 4*h + (j&3): the order in which the previous layer's accumulator
// registers 8s..8s+7 are handed over as the B operand.
// set when a folded weight does not fit fp16 (|w| >= 65504, or not finite): the net then keeps only its fp32 pack and
// every kernel runs it on the exact-fp32 core
__device__ int g_h2_weight_overflow;

__device__ __forceinline__ void store_split(_Float16* dst_hi, _Float16* dst_lo, float w) {
    if (!(fabsf(w) < 65504.0f)) atomicOr(&g_h2_weight_overflow, 1);
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * kLoScale);
    *dst_hi = hi;
    *dst_lo = lo;
}

// dst: one hidden slot (32 KiB): fragments [ks 0..15][piece 0..1][lane 64][j 8] of fp16
// (all slot packers: blockIdx.y = tile offset within a run of equally spaced slots; dst / to are the run's first slot / tile)
__global__ void k_pack_h2_hidden(_Float16* __restrict__ dst, size_t stride, PackSrc s, int to, int col_off, int cols_valid) {
    dst = (_Float16*)((char*)dst + blockIdx.y * stride);
    to += blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks, lane, j)
    if (e >= 16 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = e >> 9;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * to + i;
    const int col = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    float w = 0.0f;
    if (row < s.rows_valid && col < cols_valid)
        w = s.w[(size_t)(s.row_off + row) * s.ld + col_off + col] * s.scale[s.row_off + row] * s.mul;
    store_split(dst + ((size_t)(2 * ks) * 64 + lane) * 8 + j, dst + ((size_t)(2 * ks + 1) * 64 + lane) * 8 + j, w);
}

// dst: one head slot (8 KiB): fragments [ks 0..2][piece][lane][j] + 2 KiB of zero padding
// slot_off: first head slot of this ring slot (a head wider than 24 slots is streamed as two ring slots: 0 and 24)
__global__ void k_pack_h2_head(_Float16* __restrict__ dst, size_t stride, PackSrc s, HeadSrcs hs, int to, int slot_off) {
    dst = (_Float16*)((char*)dst + blockIdx.y * stride);
    to += blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks 0..3, lane, j)
    if (e >= 4 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = e >> 9;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * to + i;
    float w = 0.0f;
    if (ks < kHeadKSteps) {
        const int slot = slot_off + 8 * ks + j;
        for (int k = 0; k < hs.n; ++k) {
            const int local = slot - hs.slot_base[k];
            if (local >= 0 && local < head_slots(hs.levels[k])) {
                const int col = head_slot_column(local, h, hs.levels[k]);
                if (col >= 0 && row < s.rows_valid)
                    w = s.w[(size_t)(s.row_off + row) * s.ld + hs.col_off[k] + col] * s.scale[s.row_off + row] * s.mul;
            }
        }
    }
    store_split(dst + ((size_t)(2 * ks) * 64 + lane) * 8 + j, dst + ((size_t)(2 * ks + 1) * 64 + lane) * 8 + j, w);
}

// Transposed hidden slot (reverse-mode get_all, mlp_h2_rev.h): output row 32*to + i of the slot is ORIGINAL COLUMN col_off + row of the
// layer, the k index runs over the layer's original rows (its output units, in the register order their sigma' * g tiles have):
//   dst[ks][piece][lane = (i, h)][j] = W[o][col_off + 32 to + i] * scale[o] * mul,   o = 32 (ks>>1) + 16 (ks&1) + 8 (j>>2) + 4 h + (j&3)
__global__ void k_pack_h2_hidden_T(_Float16* __restrict__ dst, size_t stride, PackSrc s, int to, int col_off, int cols_valid) {
    dst = (_Float16*)((char*)dst + blockIdx.y * stride);
    to += blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks, lane, j)
    if (e >= 16 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = e >> 9;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * to + i;
    const int o = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    float w = 0.0f;
    if (row < cols_valid && o < s.rows_valid)
        w = s.w[(size_t)(s.row_off + o) * s.ld + col_off + row] * s.scale[s.row_off + o] * s.mul;
    store_split(dst + ((size_t)(2 * ks) * 64 + lane) * 8 + j, dst + ((size_t)(2 * ks + 1) * 64 + lane) * 8 + j, w);
}

// Transposed PE rows: output row i of tile T lands in accumulator register r = (i&3) + 4 (i>>3) of lane-half hh = (i>>2)&1 and stands
// for head slot 16 T + r in the role of the OTHER half -- the row of sin(2^k v_c) goes to the half whose head slot holds cos(2^k v_c)
// and vice versa, so that each lane contracts its rows with the PE values it already has (getall_rev.hip: pe_contract).
__global__ void k_pack_h2_pe_T(_Float16* __restrict__ dst, size_t stride, PackSrc s, int T, int col_off, int levels) {
    dst = (_Float16*)((char*)dst + blockIdx.y * stride);
    T += blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks, lane, j)
    if (e >= 16 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = e >> 9;
    const int i = lane & 31, h = lane >> 5;
    const int r = (i & 3) + 4 * (i >> 3), hh = (i >> 2) & 1;
    const int slot = 16 * T + r;
    const int col = slot < 2 ? head_slot_column(slot, hh, levels) : (slot < head_slots(levels) ? head_slot_column(slot, 1 - hh, levels) : -1);
    const int o = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    float w = 0.0f;
    if (col >= 0 && o < s.rows_valid)
        w = s.w[(size_t)(s.row_off + o) * s.ld + col_off + col] * s.scale[s.row_off + o] * s.mul;
    store_split(dst + ((size_t)(2 * ks) * 64 + lane) * 8 + j, dst + ((size_t)(2 * ks + 1) * 64 + lane) * 8 + j, w);
}

// head slot of a 4-component source (NeRF background points, mlp_core.h head_fill4): slot 0 = (x|y), 1 = (z|w), then
// (sin|cos)(2^k v_c) at slot 2 + 4k + c; embedding columns: 4 raw, then per level [sin x4 | cos x4]
__global__ void k_pack_h2_head4(_Float16* __restrict__ dst, size_t stride, PackSrc s, int levels, int col_off, int to, int slot_off) {
    dst = (_Float16*)((char*)dst + blockIdx.y * stride);
    to += blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (ks 0..3, lane, j)
    if (e >= 4 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = e >> 9;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * to + i;
    float w = 0.0f;
    if (ks < kHeadKSteps) {
        const int slot = slot_off + 8 * ks + j;
        int col = -1;
        if (slot == 0) col = h;
        else if (slot == 1) col = 2 + h;
        else if (slot < 2 + 4 * levels) col = 4 + 8 * ((slot - 2) / 4) + 4 * h + ((slot - 2) % 4);
        if (col >= 0 && row < s.rows_valid) w = s.w[(size_t)(s.row_off + row) * s.ld + col_off + col] * s.scale[s.row_off + row] * s.mul;
    }
    store_split(dst + ((size_t)(2 * ks) * 64 + lane) * 8 + j, dst + ((size_t)(2 * ks + 1) * 64 + lane) * 8 + j, w);
}

}  // namespace iron

using namespace iron;

namespace iron {

static int h2_overflow_reset(hipStream_t st) {
    const int zero = 0;
    IRON_HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_h2_weight_overflow), &zero, sizeof(int), 0, hipMemcpyHostToDevice, st));
    return IRON_OK;
}

// after the pack kernels have run (stream synchronised): drop the h2 stream if a weight overflowed fp16
static int h2_overflow_check(iron_net* net) {
    int flag = 0;
    IRON_HIP_TRY(hipMemcpyFromSymbol(&flag, HIP_SYMBOL(g_h2_weight_overflow), sizeof(int), 0, hipMemcpyDeviceToHost));
    if (flag) {
        (void)hipFree(net->h2_blob);
        net->h2_blob = nullptr;
    }
    return IRON_OK;
}

// Builds the h2 streams of an SDF network next to its fp32 pack.  Slot sequence (= memory order):
//   layer 0: 8 head slots; layers 1..n-2: per output tile [head slot if skip layer] hidden slot;
//   then (full stream only) the 8 hidden slots of the feature rows of the last layer;
//   then (reverse stream only: the reference's 8 x 256 network with its feature rows) the transposed layers of the reverse sweep
//   (mlp_h2_rev.h): for l = n-2 .. 1: 8 transposed hidden slots of W_l [+ 2 PE-row slots behind the skip layer's], then the 2
//   PE-row slots of W_0.
// The forward-only streams live in h2_blob (h2_trace / h2_full), the reverse stream is a second blob (h2_rev_blob) holding the
// same forward slots followed by the transposed ones, so that one ring walks it end to end.
static int pack_h2_sdf_blob(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, bool with_rev, void** blob_out,
                            H2StreamDev* s_out, uint32_t* n_trace_out, uint32_t* n_full_out, hipStream_t st) {
    const iron_net_desc& d = net->desc;
    const int nl = d.n_linear;
    const int skip = d.skip_layer;
    const int pe = pe_width(d.multires);
    const bool has_feat = d.d_out == kHidden + 1;
    const int head_bytes = 8192, hid_bytes = kSlotBytes;
    std::vector<uint32_t> table;  // {offset, kind} pairs
    size_t off = 0;
    auto add = [&](int kind) { table.push_back((uint32_t)off); table.push_back((uint32_t)kind); off += kind ? hid_bytes : head_bytes; };
    for (int to = 0; to < kHidTiles; ++to) add(0);
    for (int l = 1; l <= nl - 2; ++l)
        for (int to = 0; to < kHidTiles; ++to) { if (l == skip) add(0); add(1); }
    const uint32_t n_trace = (uint32_t)(table.size() / 2);
    if (has_feat) for (int to = 0; to < kHidTiles; ++to) add(1);
    const uint32_t n_full = (uint32_t)(table.size() / 2);
    if (n_full > 127) return IRON_ERR_UNSUPPORTED;
    if (with_rev) {
        for (int l = nl - 2; l >= 1; --l) {
            for (int to = 0; to < kHidTiles; ++to) add(1);
            if (l == skip) { add(1); add(1); }
        }
        add(1); add(1);
    }
    const uint32_t n_all = (uint32_t)(table.size() / 2);
    if (n_all > 250) return IRON_ERR_UNSUPPORTED;
    const size_t data_bytes = off;
    const size_t table_off = (data_bytes + 255) & ~(size_t)255;
    const size_t bias_off = table_off + 2048;
    const size_t rows_off = bias_off + kLdsBiasBytes;
    const size_t total = rows_off + kLdsRowsBytes + 65536;  // tail padding: the ring may prefetch past the end
    void* blob = nullptr;
    IRON_HIP_TRY(hipMalloc(&blob, total));
    *blob_out = blob;   // the caller owns it from here (freed with the net also when a later step fails)
    IRON_HIP_TRY(hipMemsetAsync(blob, 0, total, st));
    char* base = (char*)blob;
    IRON_HIP_TRY(hipMemcpyAsync(base + table_off, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    // (`table` lives until the synchronisation at the end of this function)

    HeadSrcs hs;
    memset(&hs, 0, sizeof(hs));
    hs.n = 1; hs.slot_base[0] = 0; hs.levels[0] = d.multires; hs.col_off[0] = 0;
    size_t q = 0;
    auto slot_ptr = [&](size_t idx) { return (_Float16*)(base + table[2 * idx]); };
    // one launch per run of equally spaced slots (blockIdx.y = tile): a training step re-packs every network, and one launch per slot
    // (224 per step at C3) cost ~1 ms per step of 3-4 us launches
    const size_t kHeadB = 8192, kHidB = kSlotBytes;
    hipLaunchKernelGGL(k_pack_h2_head, dim3(8, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHeadB, make_pack_src(L[0], scale_base + soff[0], kHidden, 0, 1.0f), hs, 0, 0);
    q += kHidTiles;
    for (int l = 1; l <= nl - 2; ++l) {
        const bool is_skip = (l == skip);
        const float mul = is_skip ? kInvSqrt2 : 1.0f;
        const int cols_valid = is_skip ? kHidden - pe : kHidden;
        const PackSrc ps = make_pack_src(L[l], scale_base + soff[l], L[l].out_dim, 0, mul);
        if (is_skip) {   // per tile: [head slot][hidden slot]
            HeadSrcs h2 = hs;
            h2.col_off[0] = kHidden - pe;
            hipLaunchKernelGGL(k_pack_h2_head, dim3(8, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHeadB + kHidB, ps, h2, 0, 0);
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q + 1), kHeadB + kHidB, ps, 0, 0, cols_valid);
            q += 2 * kHidTiles;
        } else {
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHidB, ps, 0, 0, cols_valid);
            q += kHidTiles;
        }
    }
    const iron_linear& last = L[nl - 1];
    if (has_feat) {
        hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHidB, make_pack_src(last, scale_base + soff[nl - 1], kHidden, 1, 1.0f), 0, 0, kHidden);
        q += kHidTiles;
    }
    if (with_rev) {
        for (int l = nl - 2; l >= 1; --l) {
            const bool is_skip = (l == skip);
            const float mul = is_skip ? kInvSqrt2 : 1.0f;
            const PackSrc ps = make_pack_src(L[l], scale_base + soff[l], L[l].out_dim, 0, mul);
            const int cols_valid = is_skip ? kHidden - pe : kHidden;   // the layer's inputs that come from the previous layer
            hipLaunchKernelGGL(k_pack_h2_hidden_T, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHidB, ps, 0, 0, cols_valid);
            q += kHidTiles;
            if (is_skip) {
                hipLaunchKernelGGL(k_pack_h2_pe_T, dim3(32, 2), dim3(256), 0, st, slot_ptr(q), kHidB, ps, 0, kHidden - pe, d.multires);
                q += 2;
            }
        }
        const PackSrc p0 = make_pack_src(L[0], scale_base + soff[0], kHidden, 0, 1.0f);
        hipLaunchKernelGGL(k_pack_h2_pe_T, dim3(32, 2), dim3(256), 0, st, slot_ptr(q), kHidB, p0, 0, 0, d.multires);
        q += 2;
    }
    if (q != table.size() / 2) return IRON_ERR_UNSUPPORTED;   // the launches above and the slot table must describe the same sequence
    // f32 side blocks: biases of layers 0..nl-2 (+ feature bias as block nl-1), last-layer row 0
    for (int l = 0; l <= nl - 2; ++l)
        hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)l * 1024), L[l].bias, 0, L[l].out_dim);
    if (has_feat)
        hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)(nl - 1) * 1024), last.bias, 1, kHidden);
    hipLaunchKernelGGL(k_pack_row, dim3(1), dim3(256), 0, st, (float*)(base + rows_off), make_pack_src(last, scale_base + soff[nl - 1], 1, 0, 1.0f), 0, 0, kHidden);
    IRON_HIP_TRY(hipGetLastError());
    IRON_HIP_TRY(hipStreamSynchronize(st));

    H2StreamDev s;
    s.base = base; s.table_off = (uint32_t)table_off; s.n_slots = n_all; s.bias_off = (uint32_t)bias_off;
    s.rows_off = (uint32_t)rows_off; s.n_bias_layers = (uint32_t)nl;
    for (int i = 0; i < 4; ++i) s.kind_mask[i] = 0;
    for (size_t k = 0; k < table.size() / 2; ++k) {
        if (k >= 128) { if (!table[2 * k + 1]) return IRON_ERR_UNSUPPORTED; continue; }   // beyond the mask: hidden slots only
        if (table[2 * k + 1]) s.kind_mask[k >> 5] |= 1u << (k & 31);
    }
    *s_out = s;
    *n_trace_out = n_trace;
    *n_full_out = n_full;
    return IRON_OK;
}

int build_h2_sdf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st) {
    const iron_net_desc& d = net->desc;
    { const int rc0 = h2_overflow_reset(st); if (rc0 != IRON_OK) return rc0; }
    H2StreamDev s;
    uint32_t n_trace = 0, n_full = 0;
    int rc = pack_h2_sdf_blob(net, L, scale_base, soff, false, &net->h2_blob, &s, &n_trace, &n_full, st);
    if (rc != IRON_OK) return rc;
    s.n_slots = n_trace;
    net->h2_trace = s;
    s.n_slots = n_full;
    net->h2_full = s;
    // the reverse stream: the network shape getall_rev.hip is written for (8 hidden layers of 256, skip at 4, PE-6, feature rows)
    if (d.n_linear == 9 && d.skip_layer == 4 && d.multires == kSdfPeLevels && d.d_out == kHidden + 1) {
        H2StreamDev r;
        rc = pack_h2_sdf_blob(net, L, scale_base, soff, true, &net->h2_rev_blob, &r, &n_trace, &n_full, st);
        if (rc != IRON_OK) return rc;
        net->h2_rev = r;
    }
    rc = h2_overflow_check(net);
    if (rc == IRON_OK && !net->h2_blob && net->h2_rev_blob) { (void)hipFree(net->h2_rev_blob); net->h2_rev_blob = nullptr; }
    return rc;
}

// h2 stream of a material network.  Sequence (= memory order):
//   [skip nets only: 8 hidden slots = the FEATURE columns of the skip layer (x 1/sqrt(2)); the kernel multiplies them while the
//    features are still its live input and parks the partial sums in net->h2_scratch until the skip layer]
//   layer 0: per output tile [head slot(s)][hidden slot = feature part];
//   layers 1..n-2: per output tile [skip layer: head slot(s) of its head columns] hidden slot (skip layer: its x columns).
// A head of up to 24 slots is one ring slot, up to 48 (the stage-1 colour net: PE-10 points + PE-4 view + normals) two.
// Side blocks: biases of layers 0..n-2, rows 0..d_out-1 of the last layer.
int build_h2_render(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, const HeadSrcs& hs,
                    int head_w, hipStream_t st) {
    const iron_net_desc& d = net->desc;
    const int nl = d.n_linear;
    const int skip = d.skip_layer;
    int head_slots_total = 0;
    for (int k = 0; k < hs.n; ++k) head_slots_total += head_slots(hs.levels[k]);
    if (head_slots_total > 2 * kHeadSlots) return IRON_OK;   // fp32 pack only
    const int n_head = head_slots_total > kHeadSlots ? 2 : 1;
    // kernels exist for (one head slot, no skip) and (two head slots, skip at a hidden layer): shade.hip launch_material
    if ((n_head == 2) != (skip != -1)) return IRON_OK;
    std::vector<uint32_t> table;
    size_t off = 0;
    auto add = [&](int kind) { table.push_back((uint32_t)off); table.push_back((uint32_t)kind); off += kind ? kSlotBytes : 8192; };
    if (skip != -1) for (int to = 0; to < kHidTiles; ++to) add(1);
    for (int to = 0; to < kHidTiles; ++to) { for (int h = 0; h < n_head; ++h) add(0); add(1); }
    for (int l = 1; l <= nl - 2; ++l)
        for (int to = 0; to < kHidTiles; ++to) {
            if (l == skip) for (int h = 0; h < n_head; ++h) add(0);
            add(1);
        }
    const uint32_t n_slots = (uint32_t)(table.size() / 2);
    if (n_slots > 127 || nl - 1 > 8) return IRON_ERR_UNSUPPORTED;
    const size_t table_off = (off + 255) & ~(size_t)255;
    const size_t bias_off = table_off + 1024;
    const size_t rows_off = bias_off + kLdsBiasBytes;
    const size_t total = rows_off + kLdsRowsBytes + 65536;
    IRON_HIP_TRY(hipMalloc(&net->h2_blob, total));
    IRON_HIP_TRY(hipMemsetAsync(net->h2_blob, 0, total, st));
    { const int rc0 = h2_overflow_reset(st); if (rc0 != IRON_OK) return rc0; }
    char* base = (char*)net->h2_blob;
    size_t q = 0;
    auto slot_ptr = [&](size_t idx) { return (_Float16*)(base + table[2 * idx]); };
    // one launch per run of equally spaced slots (blockIdx.y = tile), as in pack_h2_sdf_blob
    const size_t kHeadB = 8192, kHidB = kSlotBytes;
    if (skip != -1) {   // feature columns of the skip layer: [x 256 | head inputs | features 256] / sqrt(2)
        const PackSrc ps = make_pack_src(L[skip], scale_base + soff[skip], kHidden, 0, kInvSqrt2);
        hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHidB, ps, 0, kHidden + head_w, kHidden);
        q += kHidTiles;
    }
    {   // layer 0, per tile: [head slot(s)][hidden slot = feature part]
        const PackSrc ps = make_pack_src(L[0], scale_base + soff[0], kHidden, 0, 1.0f);
        const size_t stride = (size_t)n_head * kHeadB + kHidB;
        for (int h = 0; h < n_head; ++h)
            hipLaunchKernelGGL(k_pack_h2_head, dim3(8, kHidTiles), dim3(256), 0, st, slot_ptr(q + h), stride, ps, hs, 0, h * kHeadSlots);
        hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q + n_head), stride, ps, 0, head_w, kHidden);
        q += (size_t)(n_head + 1) * kHidTiles;
    }
    for (int l = 1; l <= nl - 2; ++l) {
        const bool is_skip = (l == skip);
        const PackSrc ps = make_pack_src(L[l], scale_base + soff[l], kHidden, 0, is_skip ? kInvSqrt2 : 1.0f);
        if (is_skip) {   // per tile: [head slot(s) of the head columns][hidden slot = the x columns]
            HeadSrcs hs2 = hs;
            for (int k = 0; k < hs2.n; ++k) hs2.col_off[k] += kHidden;
            const size_t stride = (size_t)n_head * kHeadB + kHidB;
            for (int h = 0; h < n_head; ++h)
                hipLaunchKernelGGL(k_pack_h2_head, dim3(8, kHidTiles), dim3(256), 0, st, slot_ptr(q + h), stride, ps, hs2, 0, h * kHeadSlots);
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q + n_head), stride, ps, 0, 0, kHidden);
            q += (size_t)(n_head + 1) * kHidTiles;
        } else {
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32, kHidTiles), dim3(256), 0, st, slot_ptr(q), kHidB, ps, 0, 0, kHidden);
            q += kHidTiles;
        }
    }
    if (q != table.size() / 2) return IRON_ERR_UNSUPPORTED;
    for (int l = 0; l <= nl - 2; ++l)
        hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)l * 1024), L[l].bias, 0, kHidden);
    const iron_linear& last = L[nl - 1];
    for (int o = 0; o < d.d_out; ++o)
        hipLaunchKernelGGL(k_pack_row, dim3(1), dim3(256), 0, st, (float*)(base + rows_off + (size_t)o * 1024), make_pack_src(last, scale_base + soff[nl - 1], d.d_out, 0, 1.0f), o, 0, kHidden);
    IRON_HIP_TRY(hipGetLastError());
    IRON_HIP_TRY(hipStreamSynchronize(st));
    H2StreamDev s;
    s.base = base; s.table_off = (uint32_t)table_off; s.n_slots = n_slots; s.bias_off = (uint32_t)bias_off;
    s.rows_off = (uint32_t)rows_off; s.n_bias_layers = (uint32_t)(nl - 1);
    for (int i = 0; i < 4; ++i) s.kind_mask[i] = 0;
    for (size_t k = 0; k < table.size() / 2; ++k)
        if (table[2 * k + 1]) s.kind_mask[k >> 5] |= 1u << (k & 31);
    net->h2_trace = s;
    net->h2_full = s;
    if (skip != -1) {   // partial sums of the skip layer: one 32 KiB tile set per wave of a full grid
        hipDeviceProp_t prop;
        IRON_HIP_TRY(hipGetDeviceProperties(&prop, net->device));
        net->h2_scratch_floats = (size_t)prop.multiProcessorCount * 4 * kHidTiles * 16 * 64;
        IRON_HIP_TRY(hipMalloc(&net->h2_scratch, net->h2_scratch_floats * sizeof(float)));
    }
    const int rc = h2_overflow_check(net);
    if (!net->h2_blob && net->h2_scratch) { (void)hipFree(net->h2_scratch); net->h2_scratch = nullptr; }
    return rc;
}

// h2 stream of the NeRF background field (models/fields.py:243-327; layers as create_nerf orders them: pts_linears 0..D-1, alpha,
// feature, views, rgb) for D = 8, skip after layer 4, PE-10 points (42 head slots = two ring slots), PE-4 view dirs (one).  Sequence:
//   layer 0: per tile [head A][head B];  layers 1..7: per tile [layer 5: head A, head B of the re-concatenated input] hidden;
//   feature_linear: 8 hidden;  views_linears[0] (128 rows): per tile 0..3 [head = PE(view) columns][hidden = feature columns].
// Side blocks: biases of layers 0..7 and of feature_linear (9 blocks), rows alpha, rgb 0, rgb 1; the extension block behind them
// holds the views bias and row rgb 2 (k_nerf_h2 copies it behind the standard LDS map).
int build_h2_nerf(iron_net* net, const iron_linear* L, const float* scale_base, const size_t* soff, hipStream_t st) {
    const iron_net_desc& d = net->desc;
    const int D = d.n_linear - 4;
    const int lp = d.multires > 0 ? d.multires : 0, lv = d.multires_view > 0 ? d.multires_view : 0;
    if (D != 8 || d.skip_layer != 4 || lp != 10 || lv != 4) return IRON_OK;   // fp32 pack only
    const int in_p = 4 + 8 * lp;
    std::vector<uint32_t> table;
    size_t off = 0;
    auto add = [&](int kind) { table.push_back((uint32_t)off); table.push_back((uint32_t)kind); off += kind ? kSlotBytes : 8192; };
    for (int to = 0; to < kHidTiles; ++to) { add(0); add(0); }
    for (int l = 1; l < D; ++l)
        for (int to = 0; to < kHidTiles; ++to) {
            if (l == d.skip_layer + 1) { add(0); add(0); }
            add(1);
        }
    for (int to = 0; to < kHidTiles; ++to) add(1);
    for (int to = 0; to < kHidTiles / 2; ++to) { add(0); add(1); }
    const uint32_t n_slots = (uint32_t)(table.size() / 2);
    if (n_slots > 127) return IRON_ERR_UNSUPPORTED;
    const size_t table_off = (off + 255) & ~(size_t)255;
    const size_t bias_off = table_off + 1024;
    const size_t rows_off = bias_off + kLdsBiasBytes;
    const size_t ext_off = rows_off + kLdsRowsBytes;
    const size_t total = ext_off + 2048 + 65536;
    IRON_HIP_TRY(hipMalloc(&net->h2_blob, total));
    IRON_HIP_TRY(hipMemsetAsync(net->h2_blob, 0, total, st));
    { const int rc0 = h2_overflow_reset(st); if (rc0 != IRON_OK) return rc0; }
    char* base = (char*)net->h2_blob;
    size_t q = 0;
    auto slot_ptr = [&](size_t idx) { return (_Float16*)(base + table[2 * idx]); };
    auto heads4 = [&](const PackSrc& ps, int to) {
        for (int h = 0; h < 2; ++h, ++q)
            hipLaunchKernelGGL(k_pack_h2_head4, dim3(8), dim3(256), 0, st, slot_ptr(q), (size_t)0, ps, lp, 0, to, h * kHeadSlots);
    };
    for (int to = 0; to < kHidTiles; ++to) heads4(make_pack_src(L[0], scale_base + soff[0], kHidden, 0, 1.0f), to);
    for (int l = 1; l < D; ++l) {
        const bool skip_in = (l == d.skip_layer + 1);   // input = [x (in_p) | h (256)]
        const PackSrc ps = make_pack_src(L[l], scale_base + soff[l], kHidden, 0, 1.0f);
        for (int to = 0; to < kHidTiles; ++to) {
            if (skip_in) heads4(ps, to);
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32), dim3(256), 0, st, slot_ptr(q), (size_t)0, ps, to, skip_in ? in_p : 0, kHidden);
            ++q;
        }
    }
    {
        const PackSrc ps = make_pack_src(L[D + 1], scale_base + soff[D + 1], kHidden, 0, 1.0f);
        for (int to = 0; to < kHidTiles; ++to, ++q)
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32), dim3(256), 0, st, slot_ptr(q), (size_t)0, ps, to, 0, kHidden);
    }
    {
        const PackSrc ps = make_pack_src(L[D + 2], scale_base + soff[D + 2], kHidden / 2, 0, 1.0f);
        HeadSrcs hv;
        memset(&hv, 0, sizeof(hv));
        hv.n = 1; hv.slot_base[0] = 0; hv.levels[0] = lv; hv.col_off[0] = kHidden;
        for (int to = 0; to < kHidTiles / 2; ++to) {
            hipLaunchKernelGGL(k_pack_h2_head, dim3(8), dim3(256), 0, st, slot_ptr(q), (size_t)0, ps, hv, to, 0);
            ++q;
            hipLaunchKernelGGL(k_pack_h2_hidden, dim3(32), dim3(256), 0, st, slot_ptr(q), (size_t)0, ps, to, 0, kHidden);
            ++q;
        }
    }
    for (int l = 0; l < D; ++l)
        hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)l * 1024), L[l].bias, 0, kHidden);
    hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + bias_off + (size_t)D * 1024), L[D + 1].bias, 0, kHidden);
    hipLaunchKernelGGL(k_pack_bias, dim3(1), dim3(256), 0, st, (float*)(base + ext_off), L[D + 2].bias, 0, kHidden / 2);
    hipLaunchKernelGGL(k_pack_row, dim3(1), dim3(256), 0, st, (float*)(base + rows_off), make_pack_src(L[D], scale_base + soff[D], 1, 0, 1.0f), 0, 0, kHidden);
    for (int o = 0; o < 3; ++o)
        hipLaunchKernelGGL(k_pack_row, dim3(1), dim3(256), 0, st, (float*)(base + (o < 2 ? rows_off + (size_t)(o + 1) * 1024 : ext_off + 1024)),
                           make_pack_src(L[D + 3], scale_base + soff[D + 3], 3, 0, 1.0f), o, 0, kHidden / 2);
    IRON_HIP_TRY(hipGetLastError());
    IRON_HIP_TRY(hipMemcpyAsync(base + table_off, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    IRON_HIP_TRY(hipStreamSynchronize(st));
    H2StreamDev s;
    s.base = base; s.table_off = (uint32_t)table_off; s.n_slots = n_slots; s.bias_off = (uint32_t)bias_off;
    s.rows_off = (uint32_t)rows_off; s.n_bias_layers = (uint32_t)(D + 1);
    for (int i = 0; i < 4; ++i) s.kind_mask[i] = 0;
    for (size_t k = 0; k < table.size() / 2; ++k)
        if (table[2 * k + 1]) s.kind_mask[k >> 5] |= 1u << (k & 31);
    net->h2_trace = s;
    net->h2_full = s;
    return h2_overflow_check(net);
}

}  // namespace iron
