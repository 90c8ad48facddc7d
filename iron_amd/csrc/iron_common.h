// Shared host/device declarations for libiron_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/iron_hip.h"

namespace iron {

// ---- fixed geometry of the MFMA path --------------------------------------------------------
// One wave evaluates a tile of 32 points with v_mfma_f32_32x32x2_f32: points live on lanes
// (lane & 31), features in registers.  A 32-feature tile is 16 registers per lane; feature f of
// the tile sits in register r = (f&3) + 4*(f>>3) of lane-half h = (f>>2)&1  (the MFMA C/D map
// row = (r&3) + 8*(r>>2) + 4*h), which is also exactly the B-operand slot order when the tile is
// fed back as the next layer's input, so activations never leave registers.
constexpr int kTile = 32;
constexpr int kHidden = 256;
constexpr int kHidTiles = kHidden / kTile;  // 8
constexpr int kPairs = kHidTiles / 2;       // output tiles are produced two at a time

// float4 counts of the packed blobs
constexpr int kF4PerHidLayer = kPairs * kHidTiles * 4 * 2 * 64;  // 16384 float4 = 256 KiB
constexpr int kF4PerBiasLayer = kHidTiles * 2 * 4;               // 64 float4

// Head ("non-hidden") inputs: vec3 sources with optional NeRF positional encoding.  A source with
// L levels occupies 2 + 3L register slots: slot0 = (x | y), slot1 = (z | 0), then per (level k,
// component c) one slot holding (sin | cos)(2^k * v_c) in lane-half (0 | 1).
__host__ __device__ constexpr int head_slots(int levels) { return 2 + 3 * (levels > 0 ? levels : 0); }
__host__ __device__ constexpr int pe_width(int levels) { return 3 + 6 * (levels > 0 ? levels : 0); }

// canonical (reference) column of head slot `s`, lane-half h, for a source with `levels`;
// -1 = unused slot.  Reference order (models/embedder.py:27-36): [v, sin(2^0 v), cos(2^0 v), ...].
__host__ __device__ inline int head_slot_column(int s, int h, int levels) {
    if (s == 0) return h;                 // x | y
    if (s == 1) return h ? -1 : 2;        // z | -
    int idx = s - 2;
    int k = idx / 3, c = idx % 3;
    if (k >= levels) return -1;
    return 3 + 6 * k + 3 * h + c;
}

// Packed arrays are addressed as byte offsets into one blob through a buffer resource (SGPR
// descriptor + one per-lane VGPR offset + scalar offsets), so weight streaming costs no address VGPRs.
struct SdfNetDev {
    const void* blob;
    uint32_t blob_bytes;
    uint32_t w_pe0;      // layer 0 on PE slots           [pairs][NQ][2][64] float4
    uint32_t w_hid;      // hidden layers 1..n-2          [L][pairs][8][4][2][64]
    uint32_t w_pe_skip;  // skip layer, PE part           [pairs][NQ][2][64]
    uint32_t bias;       // layers 0..n-2                 [L+1][8][2][4]
    uint32_t w_last;     // last layer row 0              [8][2][4]
    uint32_t w_feat;     // last layer rows 1..256        [pairs][8][4][2][64]  (0 = absent)
    uint32_t b_feat;     //                               [8][2][4]
    float b_last;
    float scale;
    int n_hidden_layers;      // number of softplus layers (8)
    int skip_layer;           // 4
};

// Material net: head (<= 12 quads of slots) + 256 features -> 256 x (n-1) -> d_out (<=3)
struct RenderNetDev {
    const void* blob;
    uint32_t blob_bytes;
    uint32_t w_head0;    // layer 0, head part            [pairs][NQ][2][64]
    uint32_t w_feat0;    // layer 0, feature part         [pairs][8][4][2][64]
    uint32_t w_hid;      // layers 1..n-2                 [L][pairs][8][4][2][64]
    uint32_t bias;       // layers 0..n-2                 [L+1][8][2][4]
    uint32_t w_last;     // last layer rows 0..d_out-1    [3][8][2][4]
    float b_last[3];
    int d_out;
    int n_hidden_layers;      // number of relu layers (4)
    int head_quads;           // NQ
    // head sources, in slot order
    int n_src;
    int src_kind[3];          // 0 points, 1 view_dirs, 2 normals
    int src_levels[3];
    // skip connection at hidden layer `skip_layer` (models/fields.py:222-223: x = cat([x, rendering_input]) / sqrt(2)),
    // -1 = none.  Its weights are stored as hidden block (x part), head block (w_head_skip) and a second hidden block
    // (feature part); the hidden blocks stay one contiguous stream [.., skip x, skip features, ..].
    int skip_layer;
    uint32_t w_head_skip;
    int squeeze_out;
    float squeeze_out_scale, output_bias, output_scale;
};

// NeRF background field (models/fields.py:243-327, use_viewdirs=True): 4-D points with PE-L (head of 2 + 4 L slots:
// (x|y), (z|w), then per (level, component) one (sin|cos) slot), D relu layers of 256 with the input concatenated again
// in front of layer skip+1, alpha row, a linear feature layer, [feature, PE(view)] -> 128 relu, 3 rgb rows.
struct NerfNetDev {
    const void* blob;
    uint32_t blob_bytes;
    uint32_t w_head0;      // layer 0 on the point head            [pairs][NQ4][2][64]
    uint32_t w_head_skip;  // layer skip+1, point-head columns     [pairs][NQ4][2][64]
    uint32_t w_hid;        // layers 1..D-1, feature layer, view layer (rows 0..127): contiguous hidden blocks
    uint32_t w_head_view;  // view layer, PE(view) columns         [pairs][NQV][2][64]
    uint32_t bias;         // layers 0..D-1, feature, view         [D+2][8][2][4]
    uint32_t w_alpha;      // alpha row                            [8][2][4]
    uint32_t w_rgb;        // rgb rows over the 128 view features  [3][8][2][4]
    float b_alpha, b_rgb[3];
    int n_layers;          // D
    int skip_after;        // index i with h = cat([x, h]) after layer i (4); -1 none
    int levels, levels_view;
};

// Packed weight stream of the h2 core (mlp_h2.h): the slot sequence the LDS ring walks.
struct H2StreamDev {
    const char* base;     // packed stream (global)
    uint32_t table_off;   // byte offset (from base) of the slot table: uint2 {byte offset, kind(0 head | 1 hidden)}
    uint32_t n_slots;     // sequence length
    uint32_t bias_off;    // f32 bias blocks [layer][8][2][16]
    uint32_t rows_off;    // f32 last-layer rows [3][8][2][16]
    uint32_t n_bias_layers;
    uint32_t kind_mask[4]; // bit q: slot q of the sequence is a hidden (32 KiB) slot, else a head (8 KiB) slot; slots >= 128 are hidden slots
};

}  // namespace iron

struct iron_net {
    iron_net_desc desc;
    void* blob;           // device allocation holding every packed array
    size_t blob_bytes;
    iron::SdfNetDev sdf;
    iron::RenderNetDev rnd;
    iron::NerfNetDev nerf;
    void* h2_blob;        // h2 (split-fp16) stream, SDF nets
    iron::H2StreamDev h2_trace;  // hidden stack only
    iron::H2StreamDev h2_full;   // + the feature rows of the last layer
    void* h2_rev_blob;    // stream of the reverse-mode get_all (getall_rev.hip): h2_full's slots + the transposed layers
    iron::H2StreamDev h2_rev;
    void* h2_scratch;     // material nets with a skip layer on the h2 core: partial sums parked between layer 0 and the skip layer
    size_t h2_scratch_floats;
    void* w16_blob;       // stream of the 8-wave w16 core (w16.hip), SDF nets of the reference shape
    iron::H2StreamDev w16_trace;
    int device;
    // numeric envelope of the h2 core (envelope.hip): a flag word in pinned host memory the device writes through, and the
    // sticky status it leads to -- the one mutable part of a handle
    int* flag_host;
    int* flag_dev;
    int overflow_seen;    // a call on the h2 core returned non-finite values
    int h2_disabled;      // the network runs on the exact-fp32 core from now on (iron_net_force_exact, or after an overflow)
};

namespace iron {
extern thread_local int g_last_hip_error;
inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return IRON_ERR_HIP;
}
}  // namespace iron

namespace iron {
bool use_h2_core();
// may this network run on the h2 core at all?  (IRON_MLP_CORE, an h2 stream exists, no envelope overflow / force_exact)
inline bool h2_enabled(const iron_net* net) { return use_h2_core() && net->h2_blob && !net->h2_disabled; }
int envelope_create(iron_net* net);
void envelope_destroy(iron_net* net);
int envelope_begin(const iron_net* net);   // at the start of an entry point: acts on a flag raised by an earlier call
void envelope_scan(const iron_net* net, const float* p, int64_t n_rows, const int* count_ptr, int width, hipStream_t st);
int cu_total();    // CUs of the current device
int cu_budget();   // CUs the launches of this moment may fill (iron_set_cu_limit; profile.hip)
// hipEvent pair around one kernel launch when profiling is enabled (profile.hip)
void prof_begin(int kind, hipStream_t st);
void prof_end(int kind, hipStream_t st);
struct ProfScope {
    int kind; hipStream_t st;
    ProfScope(int k, hipStream_t s) : kind(k), st(s) { prof_begin(kind, st); }
    ~ProfScope() { prof_end(kind, st); }
};
}  // namespace iron

#define IRON_HIP_TRY(expr)                                  \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return ::iron::hip_fail(_e);  \
    } while (0)
