// "h2" MLP core for gfx950: fp32-accurate dense layers on the f16 matrix pipe.
//
// Every fp32 operand is split into two fp16 pieces, x = xh + xl * 2^-11 (xh = f16(x), xl = f16((x - xh) * 2^11):
// 22 mantissa bits, the low piece scaled so that it stays in fp16's normal range), and a product is taken as
//      w*x ~= wh*xh  +  2^-11 * (wh*xl + wl*xh)
// i.e. three v_mfma_f32_32x32x16_f16 per 16-deep k-step (the dropped wl*xl term is 2^-22 relative) with fp32
// accumulation in two accumulators.  That is 3 x 32 cycles for 32x32x16 MACs against 8 x 64 cycles of
// v_mfma_f32_32x32x2_f32: 5.3x the matrix rate of the exact-fp32 core (mlp_core.h) at ~fp32 accuracy
// (measured against the reference in tests/ and DESIGN.md).
//
// At that rate weight fragments can no longer be streamed per wave from L2 (the chip's L2 bandwidth would be
// the bound), so a workgroup = 4 waves (one per SIMD, 32 points each) shares ONE weight stream through an
// LDS ring: 4 slots x 32 KiB filled by LDS-DMA (global_load_lds), three slots in flight ahead of the
// consumers, one raw s_barrier + one counted s_waitcnt vmcnt per slot, and the stream keeps running across
// evaluations (the slot sequence of a network is periodic), so there is no pipeline fill per evaluation.
// Activations stay in registers exactly as in mlp_core.h: points on lanes, the MFMA C/D register order of one
// layer is the B-operand k order of the next (weights are permuted at pack time).
#pragma once
#include "iron_common.h"
#include "mlp_core.h"
#include "lds_dma.h"

// build-time switches (tools/variants.py A/B arms; IRON_H2_NO_DMA / IRON_H2_NO_BARRIER are timing experiments only)
#ifndef IRON_H2_STAMP
#define IRON_H2_STAMP 0   // diagnostic build: s_memtime stamps of one evaluation's ring steps (tools/stamps.py)
#endif
#ifndef IRON_H2_ASM_DMA
#define IRON_H2_ASM_DMA 0      // 1: LDS-DMA from inline assembly (lds_dma.h: exact lgkmcnt waits); measured level with the builtin
#endif
#ifndef IRON_H2_FRAG_AHEAD
#define IRON_H2_FRAG_AHEAD 1   // k-steps of A fragments in flight in step_hidden (2 with ASM_DMA 1: no gain)
#endif
#ifndef IRON_H2_RING_AHEAD
#define IRON_H2_RING_AHEAD 3
#endif
#ifndef IRON_H2_SPREAD
#define IRON_H2_SPREAD 1       // round 3: stage 0 of the staged epilogue pinned into its three gaps (its 16 fma otherwise all land behind the
                               // step's first MFMA), and the finished fragments parked in AGPRs under k-steps 13 / 14 instead of behind the last MFMA
#endif
#ifndef IRON_H2_ABL
#define IRON_H2_ABL 0          // timing ablations (garbage results): 1 = half of the A-fragment LDS reads, 2 = no staged epilogue VALU
#endif
#ifndef IRON_H2_ROT_ISSUE
#define IRON_H2_ROT_ISSUE 0    // experiment: ONE wave (in rotation) issues all 32 LDS-DMA pieces of a slot instead of 8 per wave
#endif

namespace iron {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int kRingSlots = 4;
constexpr int kSlotBytes = 32768;
constexpr int kRingBytes = kRingSlots * kSlotBytes;
constexpr int kRingAhead = IRON_H2_RING_AHEAD;            // slots in flight ahead of the one being consumed
constexpr int kLoadsPerSlot = 8;         // LDS-DMA instructions per wave and slot (head: 256 B each, hidden: 1 KiB each)
constexpr float kLoScale = 2048.0f;
constexpr float kLoInv = 1.0f / 2048.0f;
constexpr int kHeadSlots = 24;           // head inputs: 3 k-steps of 8 (lane-half 0 | 1 share a slot: sin | cos)
constexpr int kHeadKSteps = 3;

// LDS map of an h2 kernel (one dynamic array; byte offsets)
constexpr int kLdsRing = 0;
constexpr int kLdsBias = kRingBytes;                    // [9 layers][8 tiles][2 halves][16] f32 = 9 KiB
constexpr int kLdsBiasBytes = 9 * 1024;
constexpr int kLdsRows = kLdsBias + kLdsBiasBytes;      // [3 rows][8][2][16] f32 = 3 KiB (last-layer rows)
constexpr int kLdsRowsBytes = 3 * 1024;
constexpr int kLdsTable = kLdsRows + kLdsRowsBytes;     // slot table: uint2 {offset, kind} x 128
constexpr int kLdsTableBytes = 1024;
constexpr int kLdsMisc = kLdsTable + kLdsTableBytes;    // per-wave flags etc.
constexpr int kLdsMiscBytes = 256;
constexpr int kLdsH2Total = kLdsMisc + kLdsMiscBytes;   // 144 640 B

// diagnostic stamps: 8 x u64 per ring step and wave, staged in LDS behind the normal map (block 0 only)
constexpr int kLdsStamp = kLdsH2Total;
constexpr int kStampSteps = 72;
constexpr int kStampFirst = 144;  // record steps [144, 216): the third evaluation of the 72-slot sequence
constexpr int kLdsStampBytes = 4 * kStampSteps * 8 * 8;
__device__ __forceinline__ void h2_stamp(unsigned long long* rec, int k) {
#if IRON_H2_STAMP
    if (rec) {
        const unsigned long long t = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) rec[k] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ f32x16 mfma_h(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// fp32 pair -> packed (hi, scaled lo) fp16 pairs, as raw dwords.  Both conversions round to nearest
// (v_cvt_pk_f16_f32 on gfx950); x - hi is exact in fp32, so the pair carries 22 mantissa bits.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    f16x2 h;
    h[0] = (_Float16)x0;
    h[1] = (_Float16)x1;
    f16x2 l;
    l[0] = (_Float16)((x0 - (float)h[0]) * kLoScale);
    l[1] = (_Float16)((x1 - (float)h[1]) * kLoScale);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// one 32-feature register tile (f32, MFMA C/D order) -> the two k-step B fragments (regs 8s..8s+7), hi and lo
struct TileFrag {
    half8 h[2];
    half8 l[2];
};

__device__ __forceinline__ void split8(const float* v, half8& h, half8& l) {
    u32x4 hh, ll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b;
        split2(v[2 * i], v[2 * i + 1], a, b);
        hh[i] = a;
        ll[i] = b;
    }
    h = __builtin_bit_cast(half8, hh);
    l = __builtin_bit_cast(half8, ll);
}

__device__ __forceinline__ void split_tile(const f32x16& v, TileFrag& t) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float tmp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i] = v[8 * s + i];
        split8(tmp, t.h[s], t.l[s]);
    }
}

struct HeadFrag {
    half8 h[kHeadKSteps];
    half8 l[kHeadKSteps];
};

__device__ __forceinline__ void split_head(const float* slots /*[24]*/, HeadFrag& f) {
#pragma unroll
    for (int s = 0; s < kHeadKSteps; ++s) split8(slots + 8 * s, f.h[s], f.l[s]);
}

// ------------------------------------------------------------------------------------------------------
// The shared weight ring.  All four waves of the workgroup step through the slot sequence together.
// Slot q of the (periodic) sequence lives in ring buffer q % 4.  Per step:
//     ring.sync()      s_waitcnt vmcnt(16) (my 8 loads of this slot are the oldest outstanding; the 16 loads of
//                      the two younger slots stay in flight)  +  s_barrier (everyone's loads of this slot have
//                      landed AND everyone is done reading the previous slot)
//     ring.step(...)   hands out (read pointer, DMA destination, DMA source) and advances
// The DMA refill and the LDS reads of one step are issued from ONE function whose pointers are __restrict__:
// hipcc otherwise orders every ds_read behind ALL pending LDS-DMA (it cannot see that they touch different ring
// buffers) with an s_waitcnt vmcnt(0), which drains the prefetch.
// ------------------------------------------------------------------------------------------------------
// source of one slot refill: the weight stream as a buffer resource + the slot's (uniform) byte offset in it
struct RingSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    const char* gbase;
    uint32_t off;
    int turn;           // IRON_H2_ROT_ISSUE: the wave that issues this refill
};

struct RingStep {
    const char* rd;     // LDS: the slot to consume
    char* wr;           // LDS: buffer to refill (the one the previous slot occupied)
    RingSrc src;        // global: source of the refill
    bool hidden;        // kind of the slot being refilled
    unsigned long long* rec;  // diagnostic stamp record of this step (null unless IRON_H2_STAMP records it)
};

struct Ring {
    __amdgpu_buffer_rsrc_t rsrc;
    const char* gbase;
    char* lds;
    unsigned long long mask_lo, mask_hi;  // bit q = 1: slot q is a hidden slot (32 KiB), 0: head slot (8 KiB)
    int n_slots;
    int q_issue;          // sequence index of the next slot to issue
    uint32_t off_issue;   // its byte offset in the stream
    int b_issue;          // ring buffer it goes to
    int b_take;           // ring buffer of the next slot to consume
    int wave, lane;
    unsigned long long* stamps;  // diagnostic (IRON_H2_STAMP)
    int n_step;
    int n_issue, n_sync;         // IRON_H2_ROT_ISSUE: refills issued / slots consumed so far

    __device__ __forceinline__ unsigned long long* cur_rec() const {
#if IRON_H2_STAMP
        return (stamps && n_step >= kStampFirst && n_step < kStampFirst + kStampSteps) ? stamps + (n_step - kStampFirst) * 8 : nullptr;
#else
        return nullptr;
#endif
    }
    __device__ __forceinline__ bool kind_of(int q) const {
        if (q >= 128) return true;   // sequences longer than the mask (getall_rev.hip) end in hidden slots only
        const unsigned long long w = q < 64 ? mask_lo : mask_hi;
        return (w >> (q & 63)) & 1ull;
    }
    __device__ __forceinline__ RingStep step() {
        RingStep s;
        s.rec = cur_rec();
        ++n_step;
        h2_stamp(s.rec, 0);  // after the barrier
        s.hidden = kind_of(q_issue);
        s.src.rsrc = rsrc;
        s.src.gbase = gbase;
        s.src.off = off_issue;
#if IRON_H2_ROT_ISSUE
        s.src.turn = n_issue & 3;
        ++n_issue;
#else
        s.src.turn = 0;
#endif
        // The ring position is compile-time periodic after unrolling; hide that, or hipcc folds it into per-read
        // absolute LDS addresses (> 16-bit immediates: one v_add per ds_read, plus AGPR parking of the CSE'd sums).
        // As an opaque scalar the slot base is ONE v_add per step and every fragment read uses an immediate offset.
        int bt = b_take;
        asm volatile("" : "+s"(bt));
        s.wr = lds + kLdsRing + b_issue * kSlotBytes;
        s.rd = lds + kLdsRing + bt * kSlotBytes;
        off_issue += s.hidden ? (uint32_t)kSlotBytes : 8192u;
        if (++q_issue == n_slots) { q_issue = 0; off_issue = 0; }
        b_issue = (b_issue + 1) & (kRingSlots - 1);
        b_take = (b_take + 1) & (kRingSlots - 1);
        return s;
    }
    __device__ __forceinline__ void sync() {
        h2_stamp(cur_rec(), 5);  // arrival at the step boundary
#if IRON_H2_ROT_ISSUE
        // the slot about to be consumed was issued whole by wave (n_sync & 3), which has issued nothing since
        if ((n_sync & 3) == wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ++n_sync;
#elif IRON_H2_RING_AHEAD == 3
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#elif IRON_H2_RING_AHEAD == 2
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
#error "ring depth"
#endif
        h2_stamp(cur_rec(), 6);  // own loads of this slot have landed
#ifndef IRON_H2_NO_BARRIER  // timing experiment only
        __builtin_amdgcn_s_barrier();
#endif
    }
    // before the kernel ends: no DMA may still be writing this workgroup's LDS
    __device__ __forceinline__ void drain() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
};

// this wave's 8 LDS-DMA instructions of one slot (global_load_lds; the buffer-addressed form, raw_ptr_buffer_load_lds, measured no faster).
// The CU's texture-address path moves 64 B/clk, so the 32 KiB of a hidden slot hold all four waves ~500 cycles: a wave
// does not run ahead of its LDS-DMA instructions (issuing them one by one behind MFMAs, or staggered over the waves,
// costs 90-170 cycles apiece instead of ~60 in a burst; both measured).
__device__ __forceinline__ void dma_issue(const RingSrc& src, char* __restrict__ wr, bool hidden, int wave) {
#ifdef IRON_H2_NO_DMA  // timing experiment only: results are garbage
    return;
#endif
    const int lane = threadIdx.x & 63;
#if IRON_H2_ROT_ISSUE
    if (src.turn != wave) return;
    {
        const char* gsrc = src.gbase + src.off + (hidden ? lane * 16 : lane * 4);
        if (hidden) {
#pragma unroll
            for (int f = 0; f < 4 * kLoadsPerSlot; ++f)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + f * 1024),
                                                 (__attribute__((address_space(3))) void*)(wr + f * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int f = 0; f < 4 * kLoadsPerSlot; ++f)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + f * 256),
                                                 (__attribute__((address_space(3))) void*)(wr + f * 256), 4, 0, 0);
        }
    }
    return;
#endif
#if IRON_H2_ASM_DMA
    // uniform base in SGPRs + a constant per-lane offset (lds_dma.h): no vector instruction per piece
    // running scalar pointers, opaque to the optimiser: hoisted out of the unrolled evaluation as loop invariants, the 16 piece
    // offsets (wave + 4 i) * {1024, 256} overflow the SGPR file and come back per step as v_readlane of a spill register
    // (readfirstlane: in the tracer kernels the ring state travels through code the uniformity analysis gives up on)
    int w = __builtin_amdgcn_readfirstlane(wave);
    asm volatile("" : "+s"(w));
    const unsigned long long gp = (unsigned long long)(src.gbase + src.off);
    const char* gstep = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gp >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gp));
    if (hidden) {
        const uint32_t lo = (uint32_t)lane * 16u;
        const char* g = gstep + w * 1024;
        uint32_t l = lds_addr_of(wr) + w * 1024;
#pragma unroll
        for (int i = 0; i < kLoadsPerSlot; ++i) {
            lds_dma16_s(g, lo, l);
            g += 4096; l += 4096;
            asm volatile("" : "+s"(g), "+s"(l));
        }
    } else {
        const uint32_t lo = (uint32_t)lane * 4u;
        const char* g = gstep + w * 256;
        uint32_t l = lds_addr_of(wr) + w * 256;
#pragma unroll
        for (int i = 0; i < kLoadsPerSlot; ++i) {
            lds_dma4_s(g, lo, l);
            g += 1024; l += 1024;
            asm volatile("" : "+s"(g), "+s"(l));
        }
    }
#else
    const char* gsrc = src.gbase + src.off + (hidden ? lane * 16 : lane * 4);
    if (hidden) {
#pragma unroll
        for (int i = 0; i < kLoadsPerSlot; ++i) {
            const int f = wave + 4 * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + f * 1024),
                                             (__attribute__((address_space(3))) void*)(wr + f * 1024), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < kLoadsPerSlot; ++i) {
            const int f = wave + 4 * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + f * 256),
                                             (__attribute__((address_space(3))) void*)(wr + f * 256), 4, 0, 0);
        }
    }
#endif
}

__device__ __forceinline__ void ring_start(Ring& r, const H2StreamDev& s, char* lds_base, int wave, int lane) {
    // the stream blob ends with the bias / row tables; 1 MiB of slack keeps num_records valid for every offset used
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(s.base), 0, 0x7fffffff, 0x00020000);
    r.gbase = s.base; r.lds = lds_base; r.n_slots = (int)s.n_slots;
    r.mask_lo = (unsigned long long)s.kind_mask[0] | ((unsigned long long)s.kind_mask[1] << 32);
    r.mask_hi = (unsigned long long)s.kind_mask[2] | ((unsigned long long)s.kind_mask[3] << 32);
    r.q_issue = 0; r.off_issue = 0; r.b_issue = 0; r.b_take = 0; r.wave = wave; r.lane = lane;
    r.n_step = -kRingAhead;  // the start-up steps below are not ring steps
    r.n_issue = 0; r.n_sync = 0;
#if IRON_H2_STAMP
    r.stamps = blockIdx.x == 0 ? reinterpret_cast<unsigned long long*>(lds_base + kLdsStamp) + wave * kStampSteps * 8 : nullptr;
#else
    r.stamps = nullptr;
#endif
    for (int i = 0; i < kRingAhead; ++i) {
        RingStep st = r.step();
        dma_issue(st.src, st.wr, st.hidden, wave);
    }
    r.b_take = 0;  // nothing consumed yet
}

// bias / output-row tiles live in LDS as f32 [tile][half][16]
__device__ __forceinline__ f32x16 lds_half_tile(const char* __restrict__ lds_block, int tile, int half) {
    const float4* p = reinterpret_cast<const float4*>(lds_block + (tile * 2 + half) * 64);
    const float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    f32x16 v;
    v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w;
    v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    v[8] = b2.x; v[9] = b2.y; v[10] = b2.z; v[11] = b2.w;
    v[12] = b3.x; v[13] = b3.y; v[14] = b3.z; v[15] = b3.w;
    return v;
}

__device__ __forceinline__ half8 lds_frag(const char* __restrict__ slot, int frag, int lane) {
    return *reinterpret_cast<const half8*>(slot + frag * 1024 + lane * 16);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    return v;
}

__device__ __forceinline__ f32x16 h2_combine(const f32x16& hi, const f32x16& lo) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = fmaf(lo[i], kLoInv, hi[i]);
    return z;
}

// One ring step on a head slot: fragments [k-step 0..2][piece hi, lo] (+ padding)
__device__ __forceinline__ void step_head(const char* __restrict__ rd, const char* __restrict__ bias, char* __restrict__ wr,
                                          const RingSrc& src, bool src_hidden, int wave, int lane, int tile, bool add_bias,
                                          const HeadFrag& hd, f32x16& acc_hi, f32x16& acc_lo) {
    dma_issue(src, wr, src_hidden, wave);
    if (add_bias) acc_hi = lds_half_tile(bias, tile, lane >> 5);
#pragma unroll
    for (int ks = 0; ks < kHeadKSteps; ++ks) {
        const half8 wh = lds_frag(rd, 2 * ks, lane);
        const half8 wl = lds_frag(rd, 2 * ks + 1, lane);
        acc_hi = mfma_h(wh, hd.h[ks], acc_hi);
        acc_lo = mfma_h(wh, hd.l[ks], acc_lo);
        acc_lo = mfma_h(wl, hd.h[ks], acc_lo);
    }
}

// (x0, x1) -> packed fp16 pair, round to nearest: v_cvt_pk_f16_f32
__device__ __forceinline__ f16x2 cvt_pk_rn(float x0, float x1) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v;
    v[0] = x0;
    v[1] = x1;
    return __builtin_convertvector(v, f16x2);
}

// max(x, 0) as one v_med3_f32 (fmaxf would add a canonicalising v_max x, x, x in IEEE mode)
__device__ __forceinline__ float relu_med3(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f); }

// relu that keeps a NaN a NaN (of either sign): 0.5 * (z + |z|), exact for every finite z.  v_med3_f32 / v_max_f32 return the OTHER
// operand for a NaN input, so a hidden pre-activation that became inf - inf after an fp16 overflow (envelope.hip) would turn into a
// plausible 0; this way it reaches the output and raises the flag.  (-inf gives NaN as well: it only arises behind an overflow.)
__device__ __forceinline__ float relu_twice(float z) { return z + __builtin_fabsf(z); }
__device__ __forceinline__ float relu_halve(float t) { return 0.5f * t; }
__device__ __forceinline__ f32x16 relu_tile_nan(f32x16 z) {
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = relu_halve(relu_twice(z[i]));
    return z;
}

// x - f32(h) with the fp16 operand read in place (low / high half of a packed pair): v_fma_mix_f32, exact
__device__ __forceinline__ float residual_lo(unsigned hpair, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
    return r;
}
__device__ __forceinline__ float residual_hi(unsigned hpair, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
    return r;
}

// epilogue of one output tile: z = hi + lo * 2^-11, activation, split into next-layer B fragments
template <bool FAST>
__device__ __forceinline__ void h2_epilogue_split(const f32x16& hi, const f32x16& lo, TileFrag& out) {
    split_tile(softplus_tile<FAST>(h2_combine(hi, lo)), out);
}

// ---- the staged epilogue -------------------------------------------------------------------------------
// softplus_100(z) = max(z, 0) + log2(1 + 2^(-|z| * 100 log2 e)) * ln2 / 100: one multiply (source modifiers carry the
// -|.|), no overflow for any z, no select.  Split: ONE packed conversion per pair for the hi halves, the residual z - hi
// by v_fma_mix (fp16 source read in place), the 2^11 scale, the second packed conversion.  12 issue slots per element.
// Pipeline state of one pending output tile.  Stage `ks` of the 16-stage pipeline is cut into three parts, one per
// MFMA gap of k-step ks, so that no gap carries more than ~6 VALU issue slots (an MFMA leaves 24 of its 32 cycles
// to the vector issue port; MI355X_MICROARCH.md).
struct EpiState {
    float z[16], e[16], rr[16];
    unsigned hpb[8];
    u32x4 oh[2], ol[2];
};

// Empty asm statements that take a value as a read-write operand: they are chained to the side-effect order (barriers,
// sched_barrier), so the producing instruction cannot float away from the place in the stream it was written at.
__device__ __forceinline__ void pin1(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin1u(unsigned& a) { asm volatile("" : "+v"(a)); }
// (The stages stay one element per instruction on purpose: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 halve the instruction count but a
// packed fp32 operation beside an MFMA costs 21 cycles of matrix throughput where a plain one costs 0.7-1.3 -- 54.7 cycles per MFMA
// with ONE v_pk_fma_f32 per gap against 35.3 with two v_fma_f32, tools/micro/mfma_gap_fill.hip, profiles/r03_mfma_gap_fill.txt --
// and hipcc 7.2 splits the packed forms it finds between MFMAs back into scalar ones for the same reason.)

// element range of part `part` (0..2) of an n-element stage: 16 -> 5/6/5, 8 -> 3/3/2
__device__ __forceinline__ constexpr int epi_lo(int n, int part) { return n == 16 ? (part == 0 ? 0 : part == 1 ? 5 : 11) : (part == 0 ? 0 : part == 1 ? 3 : 6); }
__device__ __forceinline__ constexpr int epi_hi(int n, int part) { return n == 16 ? (part == 0 ? 5 : part == 1 ? 11 : 16) : (part == 0 ? 3 : part == 1 ? 6 : 8); }

template <int EPI, int ACT>
__device__ __forceinline__ void epi_stage(EpiState& st, int ks, int part, const f32x16& p_hi, const f32x16& p_lo) {
    constexpr float kC1 = 144.26950408889634f;            // 100 * log2(e)
    constexpr float kC2 = 0.0069314718055994531f;         // ln(2) / 100
    const int a16 = epi_lo(16, part), b16 = epi_hi(16, part), a8 = epi_lo(8, part), b8 = epi_hi(8, part);
#if IRON_H2_SPREAD
    // the inputs are a finished accumulator tile: nothing keeps the compiler from computing all 16 fma behind the first MFMA, so the
    // operand is pinned to its part as well
    if (ks == 0) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { float lo = p_lo[i]; pin1(lo); st.z[i] = fmaf(lo, kLoInv, p_hi[i]); pin1(st.z[i]); } }
#else
    if (ks == 0) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = fmaf(p_lo[i], kLoInv, p_hi[i]); pin1(st.z[i]); } }
#endif
    if constexpr (ACT == 0) {
        if (ks == 1) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.e[i] = __builtin_fabsf(st.z[i]) * -kC1; pin1(st.e[i]); } }
        if (ks == 2) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = __builtin_amdgcn_exp2f(st.e[i]); pin1(st.e[i]); } }
        if (ks == 3) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = __builtin_amdgcn_exp2f(st.e[i]); pin1(st.e[i]); } }
        if (ks == 4) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.e[i] = 1.0f + st.e[i]; pin1(st.e[i]); } }
        if (ks == 5) { _Pragma("unroll") for (int i = a8; i < b8; ++i) { st.e[i] = __builtin_amdgcn_logf(st.e[i]); pin1(st.e[i]); } }
        if (ks == 6) { _Pragma("unroll") for (int i = 8 + a8; i < 8 + b8; ++i) { st.e[i] = __builtin_amdgcn_logf(st.e[i]); pin1(st.e[i]); } }
        if (ks == 7) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = relu_med3(st.z[i]); pin1(st.z[i]); } }
        if (ks == 8) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = __builtin_fmaf(st.e[i], kC2, st.z[i]); pin1(st.z[i]); } }
    } else if constexpr (ACT == 1) {   // relu, NaN-preserving (relu_twice / relu_halve): two stages of one op per element
        if (ks == 7) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.e[i] = relu_twice(st.z[i]); pin1(st.e[i]); } }
        if (ks == 8) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.z[i] = relu_halve(st.e[i]); pin1(st.z[i]); } }
    }   // ACT == 2: identity (a linear layer)
    if constexpr (EPI == 1) {
        if (ks == 9) {
            _Pragma("unroll") for (int q = a8; q < b8; ++q) {
                st.hpb[q] = __builtin_bit_cast(unsigned, cvt_pk_rn(st.z[2 * q], st.z[2 * q + 1]));
                pin1u(st.hpb[q]);
            }
        }
        if (ks == 10) {
            _Pragma("unroll") for (int i = a16; i < b16; ++i) {
                st.rr[i] = (i & 1) ? residual_hi(st.hpb[i >> 1], st.z[i]) : residual_lo(st.hpb[i >> 1], st.z[i]);
                pin1(st.rr[i]);
            }
        }
        if (ks == 11) { _Pragma("unroll") for (int i = a16; i < b16; ++i) { st.rr[i] = st.rr[i] * kLoScale; pin1(st.rr[i]); } }
        if (ks == 12) {
            _Pragma("unroll") for (int q = a8; q < b8; ++q) {
                unsigned lp = __builtin_bit_cast(unsigned, cvt_pk_rn(st.rr[2 * q], st.rr[2 * q + 1]));
                pin1u(lp);
                st.oh[q >> 2][q & 3] = st.hpb[q];
                st.ol[q >> 2][q & 3] = lp;
            }
        }
    }
}

// ---- the same epilogue, scheduled by MFMA gap (softplus networks, steps without CARRY) ---------------------
// What a vector instruction beside an MFMA costs in the micro-benchmark depends on how many share the gap (tools/micro/mfma_gap_fill.hip,
// cycles of matrix throughput per gap: 1..4 plain 1.3 / 2.0 / 2.7 / 3.3, but 5 -> 7.3 and 6 -> 9.3; 2 transcendentals 3.3, 3 -> 9.3).
// The stage-per-k-step pipeline above puts 5 / 6 / 5 plain or 3 / 3 / 2 transcendental instructions into the three gaps of a k-step
// and leaves three k-steps empty (~270 cycles per step by that table); a tile's 128 plain + 32 transcendental instructions are
// exactly 32 gaps of four + 16 gaps of two (~160 by the table).  Measured on the frame (C1, one box, interleaved runs, ms):
//   stage per k-step (epi_stage)                                                      52.96 / 52.96   sphere 19.86  sampler 26.10
//   the stages in order, four elements (two transcendentals) per gap, all 48 gaps     52.64 / 52.64          19.71          25.92   <- default
//   the tile's 8 element pairs one after the other, six gaps each (dependent
//   instructions share a gap; built, removed)                                         53.57 against 52.85 on its box: slower
// -0.6 %, a sixth of what the table promises: the ring step is not the micro-benchmark's steady state.
// CARRY steps keep the stage-per-k-step form (their fragments are due before k-steps 14 / 15).
#ifndef IRON_H2_EPI_GAPS
#define IRON_H2_EPI_GAPS 1
#endif
template <int EPI>
__device__ __forceinline__ void epi_gap(EpiState& st, int g, const f32x16& p_hi, const f32x16& p_lo) {
    constexpr float kC1 = 144.26950408889634f;            // 100 * log2(e)
    constexpr float kC2 = 0.0069314718055994531f;         // ln(2) / 100
    // gaps 0-3 combine | 4-7 -|z| c1 | 8-15 exp2 | 16-19 1 + e | 20-27 log2 | 28-31 max(z, 0) | 32-35 fma | 36-37 hi pairs | 38-41 residual |
    // 42-45 scale | 46-47 lo pairs
    if (g < 4) { _Pragma("unroll") for (int i = 4 * g; i < 4 * g + 4; ++i) { float lo = p_lo[i]; pin1(lo); st.z[i] = fmaf(lo, kLoInv, p_hi[i]); pin1(st.z[i]); } }
    else if (g < 8) { _Pragma("unroll") for (int i = 4 * (g - 4); i < 4 * (g - 4) + 4; ++i) { st.e[i] = __builtin_fabsf(st.z[i]) * -kC1; pin1(st.e[i]); } }
    else if (g < 16) { _Pragma("unroll") for (int i = 2 * (g - 8); i < 2 * (g - 8) + 2; ++i) { st.e[i] = __builtin_amdgcn_exp2f(st.e[i]); pin1(st.e[i]); } }
    else if (g < 20) { _Pragma("unroll") for (int i = 4 * (g - 16); i < 4 * (g - 16) + 4; ++i) { st.e[i] = 1.0f + st.e[i]; pin1(st.e[i]); } }
    else if (g < 28) { _Pragma("unroll") for (int i = 2 * (g - 20); i < 2 * (g - 20) + 2; ++i) { st.e[i] = __builtin_amdgcn_logf(st.e[i]); pin1(st.e[i]); } }
    else if (g < 32) { _Pragma("unroll") for (int i = 4 * (g - 28); i < 4 * (g - 28) + 4; ++i) { st.z[i] = relu_med3(st.z[i]); pin1(st.z[i]); } }
    else if (g < 36) { _Pragma("unroll") for (int i = 4 * (g - 32); i < 4 * (g - 32) + 4; ++i) { st.z[i] = __builtin_fmaf(st.e[i], kC2, st.z[i]); pin1(st.z[i]); } }
    else if constexpr (EPI == 1) {
        if (g < 38) {
            _Pragma("unroll") for (int q = 4 * (g - 36); q < 4 * (g - 36) + 4; ++q) {
                st.hpb[q] = __builtin_bit_cast(unsigned, cvt_pk_rn(st.z[2 * q], st.z[2 * q + 1]));
                pin1u(st.hpb[q]);
            }
        } else if (g < 42) {
            _Pragma("unroll") for (int i = 4 * (g - 38); i < 4 * (g - 38) + 4; ++i) {
                st.rr[i] = (i & 1) ? residual_hi(st.hpb[i >> 1], st.z[i]) : residual_lo(st.hpb[i >> 1], st.z[i]);
                pin1(st.rr[i]);
            }
        } else if (g < 46) { _Pragma("unroll") for (int i = 4 * (g - 42); i < 4 * (g - 42) + 4; ++i) { st.rr[i] = st.rr[i] * kLoScale; pin1(st.rr[i]); } }
        else {
            _Pragma("unroll") for (int q = 4 * (g - 46); q < 4 * (g - 46) + 4; ++q) {
                unsigned lp = __builtin_bit_cast(unsigned, cvt_pk_rn(st.rr[2 * q], st.rr[2 * q + 1]));
                pin1u(lp);
                st.oh[q >> 2][q & 3] = st.hpb[q];
                st.ol[q >> 2][q & 3] = lp;
            }
        }
    }
}

// One ring step on a hidden slot, fused with the epilogue of the previous output tile:
//   refill `wr`;  acc += W[tile,:] * in  (48 MFMAs, fragment reads one k-step ahead);  meanwhile (VALU) the pending
//   accumulators `p_hi/p_lo` of tile-1 go through combine + softplus (+ fp16 split) into out_prev / hf_prev.
// EPI: 0 = nothing pending, 1 = pending tile -> split fragments, 2 = pending tile -> f32 tile (last layer).
// rd / bias / wr are distinct LDS regions (see the note above): __restrict__ is what lets the reads proceed
// while the refill is in flight.
// ACT: 0 = softplus(beta=100) (SDF net), 1 = relu (material nets), 2 = identity (NeRF feature_linear).
// CARRY (with EPI = 1): the pending tile is tile 7 of the PREVIOUS layer, i.e. this layer's own input in[7].  Its
// fragments leave the pipeline behind k-step 12 and are first multiplied in k-step 14, so the last tile of a layer
// gets the same hidden epilogue as the other seven instead of an exposed one at the layer boundary.
template <bool FAST, int EPI, int ACT = 0, bool CARRY = false>
__device__ __forceinline__ void step_hidden(const char* __restrict__ rd, const char* __restrict__ bias, char* __restrict__ wr,
                                            const RingSrc& src, bool src_hidden, int wave, int lane, int tile, bool add_bias,
                                            TileFrag (&in)[kHidTiles], f32x16& acc_hi, f32x16& acc_lo,
                                            const f32x16& p_hi, const f32x16& p_lo, TileFrag& out_prev, f32x16& hf_prev,
                                            unsigned long long* rec = nullptr) {
    if (add_bias) acc_hi = lds_half_tile(bias, tile, lane >> 5);
    // A fragments: kFragAhead k-steps in flight (static register sets f*[ks % (kFragAhead + 1)] after unrolling)
    constexpr int kFragAhead = IRON_H2_FRAG_AHEAD, kSets = kFragAhead + 1;
    half8 fhs[kSets], fls[kSets];
#pragma unroll
    for (int a = 0; a < kFragAhead; ++a) { fhs[a] = lds_frag(rd, 2 * a, lane); fls[a] = lds_frag(rd, 2 * a + 1, lane); }
    __builtin_amdgcn_sched_barrier(0);
    dma_issue(src, wr, src_hidden, wave);
    __builtin_amdgcn_sched_barrier(0);
    h2_stamp(rec, 1);  // refill issued
    h2_stamp(rec, 7);  // calibration: what a stamp itself costs at this point
    // The previous tile's epilogue runs as a 16-stage software pipeline, one stage per k-step (epi_stage), each stage
    // cut into three parts that sit in the three MFMA gaps of its k-step; a stage consumes what the previous one
    // produced a whole k-step earlier, so no dependent VALU chain ever stalls the in-order wave.  sched_barrier(0)
    // keeps hipcc from re-clustering (left alone it issues the 48 MFMAs first and the ~200 VALU ops afterwards).
    static_assert(FAST || EPI == 0, "the staged epilogue implements the v_exp/v_log softplus");
    EpiState es;
#if IRON_H2_ABL & 2
    // (timing only) the skipped epilogue's outputs are still defined values, so that nothing downstream folds away
#pragma unroll
    for (int i = 0; i < 16; ++i) es.z[i] = p_hi[i];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            es.oh[q][i] = __builtin_bit_cast(unsigned, p_hi[4 * q + i]) & 0x3bff3bffu;   // finite fp16 pairs
            es.ol[q][i] = __builtin_bit_cast(unsigned, p_lo[4 * q + i]) & 0x3bff3bffu;
        }
#endif
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + kFragAhead < 16 && (!(IRON_H2_ABL & 1) || ((ks + kFragAhead) & 1) == 0)) {  // fragments of k-step ks + kFragAhead: in flight while this and the next steps' MFMAs run
            fhs[(ks + kFragAhead) % kSets] = lds_frag(rd, 2 * (ks + kFragAhead), lane);
            fls[(ks + kFragAhead) % kSets] = lds_frag(rd, 2 * (ks + kFragAhead) + 1, lane);
        }
        const half8 fh = fhs[ks % kSets], fl = fls[ks % kSets];
        const int ti = ks >> 1, s = ks & 1;
        constexpr bool kByGap = IRON_H2_EPI_GAPS != 0 && ACT == 0 && !CARRY;   // softplus tiles: the gap-scheduled form (epi_gap)
        acc_hi = mfma_h(fh, in[ti].h[s], acc_hi);
        if constexpr (EPI != 0 && !(IRON_H2_ABL & 2)) {
            if constexpr (kByGap) epi_gap<EPI>(es, 3 * ks, p_hi, p_lo); else epi_stage<EPI, ACT>(es, ks, 0, p_hi, p_lo);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_lo = mfma_h(fh, in[ti].l[s], acc_lo);
        if constexpr (EPI != 0 && !(IRON_H2_ABL & 2)) {
            if constexpr (kByGap) epi_gap<EPI>(es, 3 * ks + 1, p_hi, p_lo); else epi_stage<EPI, ACT>(es, ks, 1, p_hi, p_lo);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_lo = mfma_h(fl, in[ti].h[s], acc_lo);
        if constexpr (EPI != 0 && !(IRON_H2_ABL & 2)) {
            if constexpr (kByGap) epi_gap<EPI>(es, 3 * ks + 2, p_hi, p_lo); else epi_stage<EPI, ACT>(es, ks, 2, p_hi, p_lo);
        }
        if constexpr (CARRY) {
            static_assert(!CARRY || EPI == 1, "a carried tile ends as split fragments");
            if (ks == 12) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    asm volatile("" : "+a"(es.oh[s2]), "+a"(es.ol[s2]));
                    in[kHidTiles - 1].h[s2] = __builtin_bit_cast(half8, es.oh[s2]);
                    in[kHidTiles - 1].l[s2] = __builtin_bit_cast(half8, es.ol[s2]);
                }
            }
        }
#if IRON_H2_SPREAD
        if constexpr (EPI == 1 && !CARRY) {   // finished fragments are parked in the AGPR file under the MFMAs of k-steps 13 / 14
            if constexpr (!kByGap) {
                if (ks == 13) asm volatile("" : "+a"(es.oh[0]), "+a"(es.ol[0]));
                if (ks == 14) asm volatile("" : "+a"(es.oh[1]), "+a"(es.ol[1]));
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) h2_stamp(rec, 2);
        if (ks == 7) h2_stamp(rec, 3);
        if (ks == 15) h2_stamp(rec, 4);
    }
    if constexpr (EPI == 1 && !CARRY) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // pin to this step (LLVM would otherwise sink the epilogue to the end of the layer) and park the finished
            // fragments in the AGPR file
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

// One 256 -> 256 layer on the ring.  HEAD: the layer also has a head product (skip layer); LAST: the result is
// delivered as f32 tiles in `hf` instead of split fragments in `out`.
// CARRY: in[7] is still pending as the accumulators (c_hi, c_lo) the previous layer DEFERred; its epilogue runs under
// this layer's first tile (step_hidden<CARRY>).  DEFER: this layer's tile 7 is handed on the same way (out[7] is then
// not written here).
// HEAD = 2: the head product is two ring slots (hd, *hd2: head slots 0..23 and 24..47).  INIT: every tile's pre-activation sum
// starts from bias + init[tile][16][64] (f32, this lane's column: partial sums another product left in global memory).
// NT: number of output tiles (8; 4 for a 128-wide layer: its slots' rows 128..255 are never streamed).
// A CARRYing layer finishes the previous layer's last tile with ITS OWN ACT: only layers of equal ACT may be chained by DEFER/CARRY.
template <bool FAST, int HEAD, bool LAST, int ACT = 0, bool CARRY = false, bool DEFER = false, bool INIT = false, int NT = kHidTiles>
__device__ __forceinline__ void h2_hidden_layer(Ring& ring, const char* bias, const HeadFrag& hd, int lane,
                                                TileFrag (&in)[kHidTiles], TileFrag (&out)[kHidTiles],
                                                f32x16 (&hf)[kHidTiles], f32x16& c_hi, f32x16& c_lo,
                                                const HeadFrag* hd2 = nullptr, const float* __restrict__ init = nullptr) {
    static_assert(!INIT || HEAD > 0, "the partial sums are added on top of the head product");
    static_assert(!(LAST && DEFER), "the last layer ends in f32 tiles");
    const int wave = ring.wave;
    f32x16 acc[2][2];  // [tile parity][hi, lo]: the epilogue of tile t-1 overlaps the MFMAs of tile t
    TileFrag dummy_out;
    f32x16 dummy_hf;
#define IRON_H2_TILE(TO)                                                                                                   \
    if constexpr ((TO) < NT) {                                                                                             \
        constexpr int P = (TO) & 1, Q = P ^ 1;                                                                             \
        acc[P][0] = zero16();                                                                                              \
        acc[P][1] = zero16();                                                                                              \
        f32x16 part;                                                                                                       \
        if constexpr (INIT) {                                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) part[r] = init[((TO) * 16 + r) * 64 + lane];                    \
        }                                                                                                                  \
        if constexpr (HEAD >= 1) {                                                                                         \
            ring.sync();                                                                                                   \
            const RingStep sh = ring.step();                                                                               \
            step_head(sh.rd, bias, sh.wr, sh.src, sh.hidden, wave, lane, TO, true, hd, acc[P][0], acc[P][1]);              \
        }                                                                                                                  \
        if constexpr (HEAD >= 2) {                                                                                         \
            ring.sync();                                                                                                   \
            const RingStep sh = ring.step();                                                                               \
            step_head(sh.rd, bias, sh.wr, sh.src, sh.hidden, wave, lane, TO, false, *hd2, acc[P][0], acc[P][1]);           \
        }                                                                                                                  \
        if constexpr (INIT) {                                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[P][0][r] += part[r];                                        \
        }                                                                                                                  \
        ring.sync();                                                                                                       \
        const RingStep st = ring.step();                                                                                   \
        if constexpr ((TO) == 0 && CARRY)                                                                                  \
            step_hidden<FAST, 1, ACT, true>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, !HEAD, in, acc[P][0],   \
                                            acc[P][1], c_hi, c_lo, dummy_out, dummy_hf, st.rec);                           \
        else if constexpr ((TO) == 0)                                                                                      \
            step_hidden<FAST, 0, ACT>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, !HEAD, in, acc[P][0], acc[P][1],   \
                                 acc[Q][0], acc[Q][1], dummy_out, dummy_hf, st.rec);                                       \
        else if constexpr (LAST)                                                                                           \
            step_hidden<FAST, 2, ACT>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, !HEAD, in, acc[P][0], acc[P][1],   \
                                 acc[Q][0], acc[Q][1], dummy_out, hf[(TO) > 0 ? (TO) - 1 : 0], st.rec);                    \
        else                                                                                                               \
            step_hidden<FAST, 1, ACT>(st.rd, bias, st.wr, st.src, st.hidden, wave, lane, TO, !HEAD, in, acc[P][0], acc[P][1],   \
                                 acc[Q][0], acc[Q][1], out[(TO) > 0 ? (TO) - 1 : 0], dummy_hf, st.rec);                    \
    }
    IRON_H2_TILE(0) IRON_H2_TILE(1) IRON_H2_TILE(2) IRON_H2_TILE(3)
    IRON_H2_TILE(4) IRON_H2_TILE(5) IRON_H2_TILE(6) IRON_H2_TILE(7)
#undef IRON_H2_TILE
    static_assert(NT % 2 == 0 && NT >= 2 && NT <= kHidTiles, "the last tile sits in accumulator set 1");
    if constexpr (DEFER) {
        c_hi = acc[1][0];
        c_lo = acc[1][1];
    } else if constexpr (ACT == 0) {
        if constexpr (LAST) hf[NT - 1] = softplus_tile<FAST>(h2_combine(acc[1][0], acc[1][1]));
        else h2_epilogue_split<FAST>(acc[1][0], acc[1][1], out[NT - 1]);
    } else if constexpr (ACT == 1) {
        if constexpr (LAST) hf[NT - 1] = relu_tile_nan(h2_combine(acc[1][0], acc[1][1]));
        else split_tile(relu_tile_nan(h2_combine(acc[1][0], acc[1][1])), out[NT - 1]);
    } else {
        if constexpr (LAST) hf[NT - 1] = h2_combine(acc[1][0], acc[1][1]);
        else split_tile(h2_combine(acc[1][0], acc[1][1]), out[NT - 1]);
    }
}

// dot of this lane's 128 resident f32 features with a row held in LDS ([8][2][16] f32), summed over the halves
__device__ __forceinline__ float row_dot_lds(const char* __restrict__ row, const f32x16 (&h)[kHidTiles], int half) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) {
        const f32x16 w = lds_half_tile(row, t, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(w[r], h[t][r], s);
    }
    return s + __shfl_xor(s, 32, 64);
}

// SDFNetwork hidden stack on the h2 core.  All four waves of the workgroup must call it together.
// On return hf (f32 tiles) holds the last hidden activation.
// DEFER_TILES: hand every layer's last tile to the next layer's first step (see h2_hidden_layer); false keeps the
// exposed epilogue at the layer boundary (an escape hatch: hipcc 7.2's vgpr-form MFMA pass has crashed on k_sampler
// with the deferral in one revision of this file).
template <bool FAST, bool DEFER_TILES = true>
__device__ __forceinline__ void sdf_hidden_stack_h2(Ring& ring, const char* lds, int n_hidden_layers, int skip_layer,
                                                    float scale, float x, float y, float z, int lane,
                                                    f32x16 (&hf)[kHidTiles]) {
    const int half = lane >> 5;
    const int wave = ring.wave;
    float pe[kHeadSlots];
#pragma unroll
    for (int i = 0; i < kHeadSlots; ++i) pe[i] = 0.0f;
    head_fill<kSdfPeLevels>(x * scale, y * scale, z * scale, half, pe);
    HeadFrag hd;
    split_head(pe, hd);

    // Two activation sets X, Y alternate as input and output (no 128-register copy at a layer boundary: a copying layer
    // loop costs ~280 register moves per layer).  The launchers admit n_hidden_layers = 8, skip_layer = 4 only: an even
    // number of middle layers, the skip layer second in its pair.
    TileFrag X[kHidTiles], Y[kHidTiles];
    // layer 0: head only (9 MFMAs per tile: epilogue in place)
#pragma unroll
    for (int to = 0; to < kHidTiles; ++to) {
        ring.sync();
        const RingStep st = ring.step();
        f32x16 a_hi = zero16(), a_lo = zero16();
        step_head(st.rd, lds + kLdsBias, st.wr, st.src, st.hidden, wave, lane, to, true, hd, a_hi, a_lo);
        h2_epilogue_split<FAST>(a_hi, a_lo, X[to]);
    }
    // layers (1,2), (3,4), (5,6): X -> Y -> X; every layer but the last may defer its tile 7 (DEFER_TILES)
    f32x16 c_hi, c_lo;
    constexpr bool D = DEFER_TILES;
    for (int l = 1; l + 1 < n_hidden_layers - 1; l += 2) {
        const char* bias_a = lds + kLdsBias + l * 1024;
        const char* bias_b = bias_a + 1024;
        if (l == 1) h2_hidden_layer<FAST, false, false, 0, false, D>(ring, bias_a, hd, lane, X, Y, hf, c_hi, c_lo);
        else h2_hidden_layer<FAST, false, false, 0, D, D>(ring, bias_a, hd, lane, X, Y, hf, c_hi, c_lo);
        if (l + 1 == skip_layer) h2_hidden_layer<FAST, true, false, 0, D, D>(ring, bias_b, hd, lane, Y, X, hf, c_hi, c_lo);
        else h2_hidden_layer<FAST, false, false, 0, D, D>(ring, bias_b, hd, lane, Y, X, hf, c_hi, c_lo);
    }
    h2_hidden_layer<FAST, false, true, 0, D, false>(ring, lds + kLdsBias + (n_hidden_layers - 1) * 1024, hd, lane, X, Y, hf, c_hi, c_lo);
}

}  // namespace iron
