// Sphere tracer: RayTracer.forward = sphere_tracing + ray_sampler + rootfind
// (models/raytracer.py:45-220) as four persistent kernels on one stream, no host sync.
//
//   k_sphere   one wave owns 32 ray slots.  Every pass evaluates the SDF MLP for all 32 slots, then
//              each slot steps or retires; retired slots are refilled from a global ray queue
//              (ballot -> rank among free slots -> one atomicAdd per wave), so MFMA tiles stay full
//              although rays finish after 1..17 evaluations.  Rays still unfinished after the
//              iteration cap are appended to the sampler list.
//   k_sampler  one wave per listed ray: the 128 dense samples are evaluated 32 at a time in march
//              order and the search stops at the first block that holds a negative sample (the
//              reference evaluates all 128 and then takes the first sign change: same result).
//   k_bisect_a 32 bracketed rays per wave, each bisected until ITS interval is <= 2*threshold;
//              records the per-ray count and atomicMax-es it into the ray's chunk.
//   k_bisect_b the reference loops while ANY ray of the call is unfinished and updates ALL rays, so
//              every ray runs the chunk-wide maximum count: finish the remaining iterations, then
//              the final mid-point evaluation.
//
// Tails (an experiment kept behind IRON_TRACE_SPLIT = 2..4, default 1 = off): a persistent kernel ends on its slowest rays'
// chains of dependent evaluations (up to 17 / 16 / 10 of ~90 us each) with part of the chip idle.  The split form cuts the
// rays into independent parts whose kernel chains run on separate streams (the caller's + library-owned low-priority side
// streams, forked and joined with events, so the call stays ordered on the caller's stream) so that one part's workgroups
// could take the CUs another part's draining kernel frees.  Rays are independent and the chunk-global bisection count is an
// atomicMax table shared by the parts, so the result does not depend on the split (tests/test_gpu_trace.py).  Measured on an
// 80 k-ray tile shard (tools/shard_step_time.py, round 3): the two parts' kernels share the CUs evenly instead of one
// filling in behind the other -- every kernel takes twice as long and the step 8.75 ms against 8.64 unsplit (unequal parts
// 65/35/80 %: 9.1-9.3) -- stream priorities do not order workgroup dispatch between two resident persistent grids.
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include "mlp_h2.h"
#include "ggx_core.h"
#include "h2_setup.h"

namespace iron {

constexpr int kMaxTraceSplits = 4;

#ifndef IRON_FAST_SOFTPLUS
#define IRON_FAST_SOFTPLUS 1
#endif
constexpr bool kFastActT = IRON_FAST_SOFTPLUS != 0;

struct TraceCounters {  // zeroed at the start of every call
    int q_head;         // sphere-trace ray queue
    int n_sampler;      // rays appended for dense sampling
    int sampler_head;
    int n_root;         // rays with a sign-change bracket
    int root_head_a;
    int root_head_b;
    int n_cont;         // k_sampler: continuation items published
    int n_sampler_done; // k_sampler: listed rays whose sampling has ended
    long long n_evals;
    long long n_sphere_conv;
    long long n_evals_sphere;
    long long sampler_abort;   // k_sampler workgroups that left the queue protocol by the poll bound (0 unless the protocol is broken)
};

struct TraceWs {
    TraceCounters* cnt;
    int* sampler_list;  // [n]
    int* root_list;     // [n]
    float* root_lo;     // [n] by list position
    float* root_hi;
    float* root_flo;
    float* root_fhi;
    int* root_k;        // iterations done in phase A
    int* chunk_iters;   // [n_chunks]
    int* chunk_roots;   // [n_chunks] bisected rays per chunk (for the reference-equivalent eval count)
    int n_chunks;
    unsigned long long* cont;  // k_sampler's continuation items, [cont_cap] (zeroed at the start of a call)
    int cont_cap;
};

struct TraceArgs {
    const float* ray_o;
    const float* ray_d;
    const float* near;
    const float* far;
    const uint8_t* work;
    const int64_t* ray_index;  // may be null
    const float* lin;
    uint8_t* conv;
    float* points;
    float* sdf;
    float* dist;
    int ray0;   // first ray of this part (rays [ray0, ray0 + n) are queued; ray ids stay global)
    int n;
    int n_steps;
    int iters;
    float thr;
    long long chunk;
};

__device__ __forceinline__ int lane_rank(unsigned mask, int j) { return __popc(mask & ((1u << j) - 1u)); }

// ---- evaluation back ends ---------------------------------------------------------------------------------
// The tracer kernels are written once as per-wave state machines around three calls:
//   be.any(p)   workgroup-wide OR of a wave-uniform predicate (decides whether another evaluation pass runs)
//   be.eval()   SDF of the point on lane&31 -- collective over the workgroup
//   be.finish()
// BackendF32: one wave per workgroup, exact-fp32 MFMA core (mlp_core.h), weights streamed per wave from L2.
// BackendH2 : four waves per workgroup in lock step on the split-fp16 core (mlp_h2.h) sharing the LDS ring.
struct BackendF32 {
    static constexpr int kThreads = 64;
    SdfNetDev net;
    WStream ws;
    int lane;
    __device__ __forceinline__ void init(const SdfNetDev& n, const H2StreamDev&, const H2Meta&) {
        net = n;
        lane = threadIdx.x;
        ws.init(net.blob, net.blob_bytes, lane);
    }
    __device__ __forceinline__ bool any(bool p) { return p; }
    __device__ __forceinline__ float eval(float x, float y, float z) { return sdf_eval<kFastActT>(net, ws, x, y, z, lane); }
    __device__ __forceinline__ void finish() {}
    // per-lane state across an evaluation: this core leaves the registers for it
    __device__ __forceinline__ void park(int, float) {}
    __device__ __forceinline__ void park(int, int) {}
    __device__ __forceinline__ float unpark(int, float v) { return v; }
    __device__ __forceinline__ int unpark(int, int v) { return v; }
};

// The h2 evaluation wants all 512 registers of a lane, so whatever a tracer kernel keeps per ray across it (origin, direction, depths,
// counters: 15-17 values) was spilled to scratch by the compiler and re-read after every evaluation: 300+ MB of scratch writes per
// launch reached HBM (round 2's counters: 37x the kernel's algorithmic bytes).  The kernels now put that state into LDS themselves
// ([field][256 threads] behind the ring's map: conflict-free ds_write_b32 / ds_read_b32) and take it back after the evaluation.
constexpr int kParkFields = 17;
constexpr int kLdsPark = kLdsH2Total;
constexpr int kLdsTraceTotal = kLdsPark + kParkFields * 256 * 4;   // 162 048 B of the CU's 163 840
static_assert(kLdsTraceTotal <= 160 * 1024, "LDS of a tracer workgroup");

template <bool DEFER_TILES>
struct BackendH2T {
    static constexpr int kThreads = 256;
    Ring ring;
    char* lds;
    H2Meta m;
    int lane, wave, parity;
    __device__ __forceinline__ void init(const SdfNetDev&, const H2StreamDev& s, const H2Meta& meta) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        lds = smem;
        m = meta;
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        parity = 0;
        h2_setup(s, lds, ring);
    }
    __device__ __forceinline__ bool any(bool p) {
        volatile int* f = reinterpret_cast<volatile int*>(lds + kLdsMisc) + parity * 4;
        if (lane == 0) f[wave] = p ? 1 : 0;
        __syncthreads();
        const int r = f[0] | f[1] | f[2] | f[3];
        parity ^= 1;
        return r != 0;
    }
    __device__ __forceinline__ float eval(float x, float y, float z) {
        f32x16 hf[kHidTiles];
        sdf_hidden_stack_h2<kFastActT, DEFER_TILES>(ring, lds, m.n_hidden_layers, m.skip_layer, m.scale, x, y, z, lane, hf);
        return (row_dot_lds(lds + kLdsRows, hf, lane >> 5) + m.b_last) / m.scale;
    }
    __device__ __forceinline__ void finish() { ring.drain(); }
    __device__ __forceinline__ void park(int i, float v) { reinterpret_cast<float*>(lds + kLdsPark)[i * 256 + threadIdx.x] = v; }
    __device__ __forceinline__ void park(int i, int v) { reinterpret_cast<int*>(lds + kLdsPark)[i * 256 + threadIdx.x] = v; }
    __device__ __forceinline__ float unpark(int i, float) { return reinterpret_cast<const float*>(lds + kLdsPark)[i * 256 + threadIdx.x]; }
    __device__ __forceinline__ int unpark(int i, int) { return reinterpret_cast<const int*>(lds + kLdsPark)[i * 256 + threadIdx.x]; }
};

#ifndef IRON_TRACE_DEFER
#define IRON_TRACE_DEFER 1
#endif
typedef BackendH2T<IRON_TRACE_DEFER != 0> BackendH2;
#ifndef IRON_SAMPLER_DEFER
#define IRON_SAMPLER_DEFER 1  // see sdf_hidden_stack_h2 (0: the fallback if hipcc's vgpr-form pass crashes on k_sampler again)
#endif
typedef BackendH2T<IRON_SAMPLER_DEFER != 0> BackendH2Sampler;

#define IRON_TRACE_KERNEL_ARGS SdfNetDev net, H2StreamDev hs, H2Meta hm, TraceArgs a, TraceWs w

template <class BE>
__global__ __launch_bounds__(BE::kThreads, 1) void k_sphere(IRON_TRACE_KERNEL_ARGS) {
    BE be;
    be.init(net, hs, hm);
    const int lane = be.lane;
    const int j = lane & 31;

    bool active = false, unf = false, work = false, exhausted = false;
    int ray = 0, steps = 0;
    float px = 0.f, py = 0.f, pz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, t = 0.f, far = 0.f;
    long long evals = 0, nconv = 0;

    for (;;) {
        // ---- refill free slots from the queue
        const unsigned act = (unsigned)__ballot(active);
        const int nfree = 32 - __popc(act);
        if (nfree > 0 && !exhausted) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&w.cnt->q_head, nfree);
            base = __shfl(base, 0, 64);
            int avail = a.n - base;
            avail = avail < 0 ? 0 : (avail > nfree ? nfree : avail);
            if (base + nfree >= a.n) exhausted = true;
            const int rank = lane_rank(~act, j);
            if (!active && rank < avail) {
                ray = a.ray0 + base + rank;
                const float ox = a.ray_o[3 * (size_t)ray], oy = a.ray_o[3 * (size_t)ray + 1], oz = a.ray_o[3 * (size_t)ray + 2];
                dx = a.ray_d[3 * (size_t)ray]; dy = a.ray_d[3 * (size_t)ray + 1]; dz = a.ray_d[3 * (size_t)ray + 2];
                t = a.near[ray];
                far = a.far[ray];
                work = a.work[ray] != 0;
                unf = work;
                px = ox + dx * t; py = oy + dy * t; pz = oz + dz * t;  // raytracer.py:110
                steps = 0;
                active = true;
            }
        }
        const unsigned act2 = (unsigned)__ballot(active);
        if (!be.any(act2 != 0u)) break;
        evals += __popc(act2);

        {   // the ray state waits in LDS while the evaluation has the registers
            const int flags = (active ? 1 : 0) | (unf ? 2 : 0) | (work ? 4 : 0) | (exhausted ? 8 : 0);
            be.park(0, flags); be.park(1, ray); be.park(2, steps);
            be.park(3, px); be.park(4, py); be.park(5, pz); be.park(6, dx); be.park(7, dy); be.park(8, dz); be.park(9, t); be.park(10, far);
            be.park(11, (int)(unsigned)evals); be.park(12, (int)(evals >> 32)); be.park(13, (int)(unsigned)nconv); be.park(14, (int)(nconv >> 32));
        }
        const float s = be.eval(px, py, pz);
        {
            const int flags = be.unpark(0, (active ? 1 : 0) | (unf ? 2 : 0) | (work ? 4 : 0) | (exhausted ? 8 : 0));
            active = flags & 1; unf = flags & 2; work = flags & 4; exhausted = flags & 8;
            ray = be.unpark(1, ray); steps = be.unpark(2, steps);
            px = be.unpark(3, px); py = be.unpark(4, py); pz = be.unpark(5, pz); dx = be.unpark(6, dx); dy = be.unpark(7, dy); dz = be.unpark(8, dz);
            t = be.unpark(9, t); far = be.unpark(10, far);
            evals = (long long)(((unsigned long long)(unsigned)be.unpark(12, (int)(evals >> 32)) << 32) | (unsigned)be.unpark(11, (int)(unsigned)evals));
            nconv = (long long)(((unsigned long long)(unsigned)be.unpark(14, (int)(nconv >> 32)) << 32) | (unsigned)be.unpark(13, (int)(unsigned)nconv));
        }

        bool retire = false, to_sampler = false;
        if (active) {
            unf = unf && (fabsf(s) > a.thr) && (t < far);  // raytracer.py:114-116
            if (!unf || steps == a.iters) {
                retire = true;
                to_sampler = unf;
            } else {  // raytracer.py:123-125 (separate mul / add, p accumulates)
                t += s;
                px += dx * s; py += dy * s; pz += dz * s;
                ++steps;
            }
        }
        const unsigned samp = (unsigned)__ballot(to_sampler);
        int sbase = 0;
        if (samp) {
            if (lane == 0) sbase = atomicAdd(&w.cnt->n_sampler, __popc(samp));
            sbase = __shfl(sbase, 0, 64);
        }
        if (retire) {
            if (lane < 32) {
                const bool conv = work && !unf && (fabsf(s) <= a.thr) && (t < far);  // raytracer.py:128-133
                a.conv[ray] = conv ? 1 : 0;
                a.points[3 * (size_t)ray] = px; a.points[3 * (size_t)ray + 1] = py; a.points[3 * (size_t)ray + 2] = pz;
                a.sdf[ray] = s;
                a.dist[ray] = t;
                if (to_sampler) w.sampler_list[sbase + lane_rank(samp, j)] = ray;
                if (conv) ++nconv;
            }
            active = false;
            px = py = pz = 0.f;
        }
    }
    be.finish();
    long long c = nconv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&w.cnt->n_evals, (unsigned long long)evals);
        atomicAdd((unsigned long long*)&w.cnt->n_evals_sphere, (unsigned long long)evals);
        atomicAdd((unsigned long long*)&w.cnt->n_sphere_conv, (unsigned long long)c);
    }
}

// Samples of one ray evaluated per pass.  A wave's 32 points per pass are 32 / kSamplerBlock rays ("slots"), each marching
// through its n_steps samples kSamplerBlock at a time and leaving at the first block that holds a negative sample; a finished
// slot is refilled at once.  The reference evaluates all n_steps samples of every listed ray
// (raytracer.py:153-166); the finer the block, the fewer evaluations behind the first sign change are executed (block 32:
// 77 of 128 on average at 800x800 S0; block 8: 65) -- the samples that ARE evaluated, and what is made of them, are the same.
//
// Work items and the drain (round 3).  A listed ray needs 1 .. n_steps / kSamplerBlock passes, which nobody knows in advance: when the
// list ran out, every workgroup sat on a few long rays (up to 16 passes of 84 us each at 4 slots per wave) while the rest of its
// lanes -- and, a little later, most of the chip -- had nothing to pull.  Now a ray is marched kSamplerSeg blocks at a time: a slot
// that reaches a segment's end without a negative sample PUBLISHES the rest of the ray as a new item (ray, next block, f of the last
// sample: one 64-bit word) behind the list and takes the next ticket like any free slot.  Items = the list's rays (tickets below
// n_list) followed by the continuations in publication order; a ticket beyond what has been published is held and polled once per
// pass.  The march order of a ray's samples, the samples evaluated and the outcome are those of the one-slot-per-ray form; only which
// slot of which workgroup evaluates a segment changes.  The kernel ends when every listed ray has ended (n_sampler_done == n_list);
// no workgroup waits on another's arrival -- every published item has exactly one ticket, held by a running wave or not yet drawn --
// so a grid that is not fully resident cannot deadlock.  Relaxed agent-scope atomics on the 64-bit word are all the ordering it needs
// (everything else an item refers to was written before the kernel started).
#ifndef IRON_SAMPLER_BLOCK
#define IRON_SAMPLER_BLOCK 8
#endif
#ifndef IRON_SAMPLER_SEG
#define IRON_SAMPLER_SEG 4   // blocks per work item; 0: a slot keeps its ray to the end (the round-2 form)
#endif
constexpr int kSamplerBlock = IRON_SAMPLER_BLOCK;
constexpr int kSamplerSlots = 32 / kSamplerBlock;
constexpr int kSamplerSeg = IRON_SAMPLER_SEG;
static_assert(kSamplerBlock == 4 || kSamplerBlock == 8 || kSamplerBlock == 16 || kSamplerBlock == 32, "sampler block");
constexpr int kContRayBits = 24, kContBlkBits = 8;   // item word: [ray + 1 : 24][next block : 8][f of the previous sample : 32]

// continuation items a call of n rays can publish (0: the kernel keeps every ray in its slot)
static inline int64_t sampler_cont_cap(int64_t n, int n_steps) {
    if (kSamplerSeg <= 0) return 0;
    const int64_t blocks = (n_steps + kSamplerBlock - 1) / kSamplerBlock;
    if (n + 2 >= (1ll << kContRayBits) || blocks >= (1ll << kContBlkBits)) return 0;
    const int64_t segs = (blocks + kSamplerSeg - 1) / kSamplerSeg;
    if (n * (segs - 1) * 8 > (256ll << 20)) return 0;   // (unusual step counts on a large call: not worth more than 256 MiB of workspace)
    return n * (segs - 1);
}

__device__ __forceinline__ float sample_depth(float smin, float lin, float width) { return smin + lin * width; }   // raytracer.py:147-149

template <class BE>
__global__ __launch_bounds__(BE::kThreads, 1) void k_sampler(IRON_TRACE_KERNEL_ARGS) {
    BE be;
    be.init(net, hs, hm);
    const int lane = be.lane;
    const int j = lane & 31;
    const int slot = j / kSamplerBlock, s_in = j % kSamplerBlock;   // this lane's ray slot and its sample within the slot's block
    const int slot_lane0 = slot * kSamplerBlock;
    const unsigned slot_bits = (kSamplerBlock == 32 ? 0xffffffffu : ((1u << kSamplerBlock) - 1u)) << slot_lane0;
    const int n_list = w.cnt->n_sampler;
    const bool dyn = w.cont_cap > 0;
    const long long n_tickets = (long long)n_list + (dyn ? (long long)w.cont_cap : 0ll);
    long long evals = 0;
    // per slot (uniform over its lanes): a ray, or a ticket for the next item, or nothing more to draw
    bool has_ray = false, retired = false, publish = false;
    int ticket = -1;
    int ray = 0, blk = 0;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, smin = 0.f, width = 0.f, prev_z = 0.f, prev_f = 0.f;
    unsigned idle_polls = 0;
    for (;;) {
        // ---- publish the continuations of the last pass and draw tickets for the free slots: one atomicAdd each per wave
        unsigned pub_slots = 0, need_slots = 0;
        {
            const unsigned pub = (unsigned)__ballot(publish), need = (unsigned)__ballot(!has_ray && ticket < 0 && !retired);
#pragma unroll
            for (int q = 0; q < kSamplerSlots; ++q) {
                pub_slots |= ((pub >> (q * kSamplerBlock)) & 1u) << q;
                need_slots |= ((need >> (q * kSamplerBlock)) & 1u) << q;
            }
        }
        if (pub_slots | need_slots) {
            int v = 0;
            if (lane == 0 && need_slots) v = atomicAdd(&w.cnt->sampler_head, __popc(need_slots));
            if (lane == 1 && pub_slots) v = atomicAdd(&w.cnt->n_cont, __popc(pub_slots));
            const int tbase = __shfl(v, 0, 64), pbase = __shfl(v, 1, 64);
            if (publish) {
                const int c = pbase + __popc(pub_slots & ((1u << slot) - 1u));
                if (lane == slot_lane0 && c < w.cont_cap) {
                    const unsigned long long item = (unsigned long long)(unsigned)(ray + 1) | ((unsigned long long)(unsigned)blk << kContRayBits) |
                                                    ((unsigned long long)__float_as_uint(prev_f) << 32);
                    __hip_atomic_store(&w.cont[c], item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                publish = false;
            }
            if (!has_ray && ticket < 0 && !retired) {
                const long long t = (long long)tbase + __popc(need_slots & ((1u << slot) - 1u));
                if (t >= n_tickets || tbase < 0) retired = true;   // nothing beyond: every ray publishes at most segs - 1 items
                else ticket = (int)t;
            }
        }
        // ---- a held ticket: the list's ray, or the continuation item once it is there
        if (!has_ray && ticket >= 0) {
            bool got = false;
            if (ticket < n_list) {
                ray = w.sampler_list[ticket];
                blk = 0;
                prev_f = 0.f;
                got = true;
            } else {
                const unsigned long long item = __hip_atomic_load(&w.cont[ticket - n_list], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (item != 0ull) {
                    ray = (int)(item & ((1u << kContRayBits) - 1u)) - 1;
                    blk = (int)((item >> kContRayBits) & ((1u << kContBlkBits) - 1u));
                    prev_f = __uint_as_float((unsigned)(item >> 32));
                    got = true;
                }
            }
            if (got) {
                ox = a.ray_o[3 * (size_t)ray]; oy = a.ray_o[3 * (size_t)ray + 1]; oz = a.ray_o[3 * (size_t)ray + 2];
                dx = a.ray_d[3 * (size_t)ray]; dy = a.ray_d[3 * (size_t)ray + 1]; dz = a.ray_d[3 * (size_t)ray + 2];
                const float t = a.dist[ray], s0 = a.sdf[ray];
                const bool pos = s0 > 0.0f;  // raytracer.py:59-65: sample [t, far] if sdf > 0 else [near, t]
                smin = pos ? t : a.near[ray];
                const float smax = pos ? a.far[ray] : t;
                width = smax - smin;
                prev_z = blk > 0 ? sample_depth(smin, a.lin[blk * kSamplerBlock - 1], width) : 0.f;   // the sample before the item's first
                has_ray = true;
                ticket = -1;
            }
        }
        if (!be.any(__ballot(has_ray) != 0ull)) {
            // no ray in the workgroup: done when no slot holds a ticket that can still be served
            const bool all_ended = __hip_atomic_load(&w.cnt->n_sampler_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_list;
            const bool waiting = __ballot(ticket >= 0 || publish) != 0ull && !all_ended;
            if (!be.any(waiting)) break;
            if (++idle_polls > (1u << 22)) {   // the bound: a few seconds; never reached unless the protocol is broken -- reported in the stats
                if (threadIdx.x == 0) atomicAdd((unsigned long long*)&w.cnt->sampler_abort, 1ull);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        const int idx = blk * kSamplerBlock + s_in;
        const bool in_range = has_ray && idx < a.n_steps;
        const float z = sample_depth(smin, a.lin[in_range ? idx : a.n_steps - 1], width);
        const float qx = has_ray ? ox + dx * z : 0.f, qy = has_ray ? oy + dy * z : 0.f, qz = has_ray ? oz + dz * z : 0.f;  // :150
        {   // the slot's state waits in LDS while the evaluation has the registers
            const int flags = (has_ray ? 1 : 0) | (retired ? 2 : 0) | (publish ? 4 : 0) | (in_range ? 8 : 0);
            be.park(0, flags); be.park(1, ray); be.park(2, blk); be.park(3, ticket);
            be.park(4, ox); be.park(5, oy); be.park(6, oz); be.park(7, dx); be.park(8, dy); be.park(9, dz);
            be.park(10, smin); be.park(11, width); be.park(12, prev_z); be.park(13, prev_f); be.park(14, z);
            be.park(15, (int)(unsigned)evals); be.park(16, (int)(evals >> 32));
        }
        const float f = be.eval(qx, qy, qz);
        const int flags_back = be.unpark(0, (has_ray ? 1 : 0) | (retired ? 2 : 0) | (publish ? 4 : 0) | (in_range ? 8 : 0));
        has_ray = flags_back & 1; retired = flags_back & 2; publish = flags_back & 4;
        const bool in_range_b = flags_back & 8;
        ray = be.unpark(1, ray); blk = be.unpark(2, blk); ticket = be.unpark(3, ticket);
        ox = be.unpark(4, ox); oy = be.unpark(5, oy); oz = be.unpark(6, oz); dx = be.unpark(7, dx); dy = be.unpark(8, dy); dz = be.unpark(9, dz);
        smin = be.unpark(10, smin); width = be.unpark(11, width); prev_z = be.unpark(12, prev_z); prev_f = be.unpark(13, prev_f);
        const float zb = be.unpark(14, z);
        evals = (long long)(((unsigned long long)(unsigned)be.unpark(16, (int)(evals >> 32)) << 32) | (unsigned)be.unpark(15, (int)(unsigned)evals));
        idle_polls = 0;
        evals += __popc((unsigned)__ballot(in_range_b));   // lanes 32..63 mirror 0..31: the low word counts every point once
        const unsigned neg_all = (unsigned)__ballot(in_range_b && f < 0.0f);  // sign(f) == -1 (raytracer.py:162-166)
        // the slot's neighbours' values, fetched by every lane (shuffles are wave-wide operations)
        const unsigned neg = neg_all & slot_bits;
        const int first = neg ? (__ffs(neg) - 1) : slot_lane0;          // wave lane of the slot's first negative sample
        const float z_first = __shfl(zb, first, 64), f_first = __shfl(f, first, 64);
        const float z_before = __shfl(zb, first > slot_lane0 ? first - 1 : slot_lane0, 64);
        const float f_before = __shfl(f, first > slot_lane0 ? first - 1 : slot_lane0, 64);
        const float z_last = __shfl(zb, slot_lane0 + kSamplerBlock - 1, 64), f_last = __shfl(f, slot_lane0 + kSamplerBlock - 1, 64);
        // outcome of this block for the slot (straight-line: every lane of the slot computes the same)
        const int gidx = blk * kSamplerBlock + (first - slot_lane0);          // index of the first negative sample, if any
        const bool found_neg = has_ray && neg != 0u;
        const bool root = found_neg && gidx >= 1;                              // raytracer.py:167
        const bool done = has_ray && (found_neg || (blk + 1) * kSamplerBlock >= a.n_steps);
        const float z_lo = first > slot_lane0 ? z_before : prev_z, f_lo = first > slot_lane0 ? f_before : prev_f;
        if (done && lane == slot_lane0) {   // the slot's first lane (lower half of the wave) writes the outcome
            if (root) {
                const int pos_l = atomicAdd(&w.cnt->n_root, 1);
                w.root_list[pos_l] = ray;
                w.root_lo[pos_l] = z_lo; w.root_hi[pos_l] = z_first;
                w.root_flo[pos_l] = f_lo; w.root_fhi[pos_l] = f_first;
            } else {  // raytracer.py:158-160, 75-78: sampled rays without a root get zeros
                a.conv[ray] = 0;
                a.points[3 * (size_t)ray] = 0.f; a.points[3 * (size_t)ray + 1] = 0.f; a.points[3 * (size_t)ray + 2] = 0.f;
                a.sdf[ray] = 0.f;
                a.dist[ray] = 0.f;
            }
            atomicAdd(&w.cnt->n_sampler_done, 1);
        }
        prev_z = z_last;
        prev_f = f_last;
        ++blk;
        // the end of a work item: the ray goes back to the queue (published at the top of the next pass), the slot draws the next ticket
        const bool hand_over = has_ray && !done && dyn && (blk % (kSamplerSeg > 0 ? kSamplerSeg : 1)) == 0;
        publish = hand_over;
        if (done || hand_over) has_ray = false;
    }
    be.finish();
    if (lane == 0) atomicAdd((unsigned long long*)&w.cnt->n_evals, (unsigned long long)evals);
}

__device__ __forceinline__ long long ray_chunk(const TraceArgs& a, int ray) {
    const long long gi = a.ray_index ? a.ray_index[ray] : (long long)ray;
    return a.chunk > 0 ? gi / a.chunk : 0;
}

// rootfind, per-ray part (raytracer.py:199-217)
template <class BE>
__global__ __launch_bounds__(BE::kThreads, 1) void k_bisect_a(IRON_TRACE_KERNEL_ARGS) {
    BE be;
    be.init(net, hs, hm);
    const int lane = be.lane;
    const int j = lane & 31;
    const int n_root = w.cnt->n_root;
    const float thr2 = 2.0f * a.thr;
    long long evals = 0;
    bool has_batch = false, exhausted = false, valid = false, work = false;
    int li = 0, ray = 0, k = 0;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, lo = 0.f, hi = 0.f, mid = 0.f;
    auto store = [&]() {
        if (valid && lane < 32) {
            w.root_lo[li] = lo; w.root_hi[li] = hi;
            w.root_k[li] = k;
            const long long ch = ray_chunk(a, ray);
            atomicAdd(&w.chunk_roots[ch], 1);
            if (k > 0) atomicMax(&w.chunk_iters[ch], k);
        }
    };
    for (;;) {
        while (!has_batch && !exhausted) {  // wave-local: batches that need no evaluation are finished here
            int base = 0;
            if (lane == 0) base = atomicAdd(&w.cnt->root_head_a, 32);
            base = __shfl(base, 0, 64);
            if (base >= n_root) { exhausted = true; break; }
            li = base + j;
            valid = li < n_root;
            ray = valid ? w.root_list[li] : 0;
            ox = oy = oz = dx = dy = dz = lo = hi = 0.f;
            work = false;
            if (valid) {
                ox = a.ray_o[3 * (size_t)ray]; oy = a.ray_o[3 * (size_t)ray + 1]; oz = a.ray_o[3 * (size_t)ray + 2];
                dx = a.ray_d[3 * (size_t)ray]; dy = a.ray_d[3 * (size_t)ray + 1]; dz = a.ray_d[3 * (size_t)ray + 2];
                lo = w.root_lo[li]; hi = w.root_hi[li];
                work = (w.root_flo[li] > 0.0f) && (w.root_fhi[li] < 0.0f);
            }
            mid = (lo + hi) / 2.0f;
            k = 0;
            if (__ballot(work) != 0ull) has_batch = true;
            else store();
        }
        if (!be.any(has_batch)) break;
        evals += has_batch ? __popc((unsigned)__ballot(work)) : 0;
        {   // the batch's state waits in LDS while the evaluation has the registers
            const int flags = (has_batch ? 1 : 0) | (exhausted ? 2 : 0) | (valid ? 4 : 0) | (work ? 8 : 0);
            be.park(0, flags); be.park(1, li); be.park(2, ray); be.park(3, k);
            be.park(4, ox); be.park(5, oy); be.park(6, oz); be.park(7, dx); be.park(8, dy); be.park(9, dz);
            be.park(10, lo); be.park(11, hi); be.park(12, mid); be.park(13, (int)(unsigned)evals); be.park(14, (int)(evals >> 32));
        }
        const float f = be.eval(ox + dx * mid, oy + dy * mid, oz + dz * mid);
        {
            const int flags = be.unpark(0, (has_batch ? 1 : 0) | (exhausted ? 2 : 0) | (valid ? 4 : 0) | (work ? 8 : 0));
            has_batch = flags & 1; exhausted = flags & 2; valid = flags & 4; work = flags & 8;
            li = be.unpark(1, li); ray = be.unpark(2, ray); k = be.unpark(3, k);
            ox = be.unpark(4, ox); oy = be.unpark(5, oy); oz = be.unpark(6, oz); dx = be.unpark(7, dx); dy = be.unpark(8, dy); dz = be.unpark(9, dz);
            lo = be.unpark(10, lo); hi = be.unpark(11, hi); mid = be.unpark(12, mid);
            evals = (long long)(((unsigned long long)(unsigned)be.unpark(14, (int)(evals >> 32)) << 32) | (unsigned)be.unpark(13, (int)(unsigned)evals));
        }
        if (has_batch) {
            if (work) {
                if (f > 0.0f) lo = mid; else hi = mid;
                mid = (lo + hi) / 2.0f;
                ++k;
                work = ((hi - lo) > thr2) && (k < 64);  // k < 64: exit bound for non-finite intervals
            }
            if (__ballot(work) == 0ull) {
                store();
                has_batch = false;
                ox = oy = oz = dx = dy = dz = mid = 0.f;
            }
        }
    }
    be.finish();
    if (lane == 0) atomicAdd((unsigned long long*)&w.cnt->n_evals, (unsigned long long)evals);
}

// rootfind, chunk-global remainder + final evaluation (raytracer.py:204-219)
template <class BE>
__global__ __launch_bounds__(BE::kThreads, 1) void k_bisect_b(IRON_TRACE_KERNEL_ARGS) {
    BE be;
    be.init(net, hs, hm);
    const int lane = be.lane;
    const int j = lane & 31;
    const int n_root = w.cnt->n_root;
    long long evals = 0;
    bool has_batch = false, exhausted = false, valid = false;
    int li = 0, ray = 0, remaining = 0;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, lo = 0.f, hi = 0.f, mid = 0.f;
    for (;;) {
        if (!has_batch && !exhausted) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&w.cnt->root_head_b, 32);
            base = __shfl(base, 0, 64);
            if (base >= n_root) {
                exhausted = true;
            } else {
                li = base + j;
                valid = li < n_root;
                ray = valid ? w.root_list[li] : 0;
                ox = oy = oz = dx = dy = dz = lo = hi = 0.f;
                remaining = 0;
                if (valid) {
                    ox = a.ray_o[3 * (size_t)ray]; oy = a.ray_o[3 * (size_t)ray + 1]; oz = a.ray_o[3 * (size_t)ray + 2];
                    dx = a.ray_d[3 * (size_t)ray]; dy = a.ray_d[3 * (size_t)ray + 1]; dz = a.ray_d[3 * (size_t)ray + 2];
                    lo = w.root_lo[li]; hi = w.root_hi[li];
                    remaining = w.chunk_iters[ray_chunk(a, ray)] - w.root_k[li];
                }
                mid = (lo + hi) / 2.0f;
                has_batch = true;
            }
        }
        if (!be.any(has_batch)) break;
        const unsigned rem = has_batch ? (unsigned)__ballot(remaining > 0) : 0u;
        float qx = ox + dx * mid, qy = oy + dy * mid, qz = oz + dz * mid;
        {   // the batch's state waits in LDS while the evaluation has the registers
            const int flags = (has_batch ? 1 : 0) | (exhausted ? 2 : 0) | (valid ? 4 : 0);
            be.park(0, flags); be.park(1, li); be.park(2, ray); be.park(3, remaining);
            be.park(4, ox); be.park(5, oy); be.park(6, oz); be.park(7, dx); be.park(8, dy); be.park(9, dz);
            be.park(10, lo); be.park(11, hi); be.park(12, mid); be.park(13, (int)(unsigned)evals); be.park(14, (int)(evals >> 32));
            be.park(15, (int)rem);
        }
        const float f = be.eval(qx, qy, qz);
        {
            const int flags = be.unpark(0, (has_batch ? 1 : 0) | (exhausted ? 2 : 0) | (valid ? 4 : 0));
            has_batch = flags & 1; exhausted = flags & 2; valid = flags & 4;
            li = be.unpark(1, li); ray = be.unpark(2, ray); remaining = be.unpark(3, remaining);
            ox = be.unpark(4, ox); oy = be.unpark(5, oy); oz = be.unpark(6, oz); dx = be.unpark(7, dx); dy = be.unpark(8, dy); dz = be.unpark(9, dz);
            lo = be.unpark(10, lo); hi = be.unpark(11, hi); mid = be.unpark(12, mid);
            evals = (long long)(((unsigned long long)(unsigned)be.unpark(14, (int)(evals >> 32)) << 32) | (unsigned)be.unpark(13, (int)(unsigned)evals));
            qx = ox + dx * mid; qy = oy + dy * mid; qz = oz + dz * mid;   // (the same expressions: the evaluated point)
        }
        const unsigned rem_b = (unsigned)be.unpark(15, (int)rem);
        if (has_batch) {
            if (rem_b) {
                evals += __popc(rem_b);
                if (remaining > 0) {
                    if (f > 0.0f) lo = mid; else hi = mid;
                    mid = (lo + hi) / 2.0f;
                    --remaining;
                }
            } else {  // this was the final mid-point evaluation (raytracer.py:218-219)
                evals += __popc((unsigned)__ballot(valid));
                if (valid && lane < 32) {  // raytracer.py:75-78: the sampler's mask overwrites convergent
                    a.conv[ray] = 1;
                    a.points[3 * (size_t)ray] = qx; a.points[3 * (size_t)ray + 1] = qy; a.points[3 * (size_t)ray + 2] = qz;
                    a.sdf[ray] = f;
                    a.dist[ray] = mid;
                }
                has_batch = false;
                ox = oy = oz = dx = dy = dz = mid = 0.f;
            }
        }
    }
    be.finish();
    if (lane == 0) atomicAdd((unsigned long long*)&w.cnt->n_evals, (unsigned long long)evals);
}

// `w.cnt` = the first part's counters; the parts' counter blocks are kCntStride bytes apart
constexpr size_t kCntStride = 256;
__global__ void k_trace_stats(TraceWs w, int n_parts, int n_steps, iron_trace_stats* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long n_evals = 0, n_sphere_conv = 0, n_sampler = 0, n_root = 0, n_evals_sphere = 0, n_abort = 0;
        for (int p = 0; p < n_parts; ++p) {
            const TraceCounters* c = (const TraceCounters*)((const char*)w.cnt + (size_t)p * kCntStride);
            n_evals += c->n_evals; n_sphere_conv += c->n_sphere_conv; n_sampler += c->n_sampler; n_root += c->n_root;
            n_evals_sphere += c->n_evals_sphere;
            n_abort += c->sampler_abort;
        }
        out->n_evals = n_evals;
        out->n_sphere_conv = n_sphere_conv;
        out->n_sampler = n_sampler;
        out->n_bisect = n_root;
        out->n_conv = n_sphere_conv + n_root;
        long long e = n_evals_sphere + n_sampler * n_steps;
        for (int c = 0; c < w.n_chunks; ++c) e += (long long)w.chunk_roots[c] * (w.chunk_iters[c] + 1);
        out->n_evals_ref = e;
        out->n_evals_sphere = n_evals_sphere;
        out->reserved = n_abort;   // k_sampler workgroups that gave up polling (iron_hip.h)
    }
}

// ---- single stages (iron_trace_stage: RayTracer.sphere_tracing / ray_sampler / rootfind as callable methods) ---------------------
__global__ void k_stage_mark_list(const int* __restrict__ list, const int* __restrict__ count, uint8_t* __restrict__ mask) {
    const int n = *count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) mask[list[i]] = 1;
}

// ray_sampler (raytracer.py:142-197) takes every ray of the call with its own interval [min_dis, max_dis]: the list is the
// identity and the per-ray state the dense sampler reads is set so that it samples exactly that interval (sdf > 0: [dist, far])
__global__ void k_stage_sampler_init(TraceArgs a, TraceWs w, const float* __restrict__ min_dis) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        w.sampler_list[i] = i;
        a.dist[i] = min_dis[i];
        a.sdf[i] = 1.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) w.cnt->n_sampler = a.n;
}

// rootfind (raytracer.py:199-220) on caller-given brackets
__global__ void k_stage_root_init(TraceWs w, int n, const float* __restrict__ f_low, const float* __restrict__ f_high,
                                  const float* __restrict__ d_low, const float* __restrict__ d_high) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        w.root_list[i] = i;
        w.root_lo[i] = d_low[i]; w.root_hi[i] = d_high[i];
        w.root_flo[i] = f_low[i]; w.root_fhi[i] = f_high[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) w.cnt->n_root = n;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
    size_t cnt, sampler_list, root_list, lo, hi, flo, fhi, k, chunk_iters, chunk_roots, cont, total;
    int64_t n_chunks, cont_cap;
};

static WsLayout ws_layout(int64_t n, const iron_trace_params* p) {
    WsLayout L;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    L.n_chunks = (p && p->chunk > 0) ? (n + p->chunk - 1) / p->chunk : 1;
    if (L.n_chunks < 1) L.n_chunks = 1;
    size_t o = 0;
    static_assert(sizeof(TraceCounters) <= kCntStride, "counter block");
    L.cnt = o; o += kCntStride * kMaxTraceSplits;
    L.chunk_iters = o; o += align256(sizeof(int) * (size_t)L.n_chunks);
    L.chunk_roots = o; o += align256(sizeof(int) * (size_t)L.n_chunks);
    L.sampler_list = o; o += align256(sizeof(int) * nn);
    L.root_list = o; o += align256(sizeof(int) * nn);
    L.lo = o; o += align256(sizeof(float) * nn);
    L.hi = o; o += align256(sizeof(float) * nn);
    L.flo = o; o += align256(sizeof(float) * nn);
    L.fhi = o; o += align256(sizeof(float) * nn);
    L.k = o; o += align256(sizeof(int) * nn);
    L.cont_cap = sampler_cont_cap(n, p ? p->n_steps : 128);   // k_sampler's continuation items
    L.cont = o; o += align256(sizeof(unsigned long long) * (size_t)(L.cont_cap > 0 ? L.cont_cap : 1));
    L.total = o;
    return L;
}

// which: 0 sphere, 1 sampler, 2 bisect_a, 3 bisect_b; `units` = wave-sized work items available
static void launch_trace_kernel(int which, bool h2, const iron_net* sdf, const TraceArgs& a, const TraceWs& w, int64_t units,
                                hipStream_t st);

static int resident_waves() {
    // single-wave workgroups, one wave per SIMD (the kernels need > 256 registers per lane); iron_set_cu_limit narrows it
    return cu_budget() * 4;
}

static void launch_trace_kernel(int which, bool h2, const iron_net* sdf, const TraceArgs& a, const TraceWs& w, int64_t units,
                                hipStream_t st) {
    H2Meta m;
    m.n_hidden_layers = sdf->sdf.n_hidden_layers; m.skip_layer = sdf->sdf.skip_layer; m.scale = sdf->sdf.scale; m.b_last = sdf->sdf.b_last;
    if (units < 1) units = 1;
    if (h2) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_sphere<BackendH2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTraceTotal);
            (void)hipFuncSetAttribute((const void*)k_sampler<BackendH2Sampler>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTraceTotal);
            (void)hipFuncSetAttribute((const void*)k_bisect_a<BackendH2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTraceTotal);
            (void)hipFuncSetAttribute((const void*)k_bisect_b<BackendH2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTraceTotal);
            attr = true;
        }
        const int64_t wgs = (units + 3) / 4;
        const int cus = resident_waves() / 4;
        const dim3 grid((unsigned)(wgs < cus ? wgs : cus)), block(256);
        switch (which) {
            case 0: hipLaunchKernelGGL(k_sphere<BackendH2>, grid, block, kLdsTraceTotal, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            case 1: hipLaunchKernelGGL(k_sampler<BackendH2Sampler>, grid, block, kLdsTraceTotal, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            case 2: hipLaunchKernelGGL(k_bisect_a<BackendH2>, grid, block, kLdsTraceTotal, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            default: hipLaunchKernelGGL(k_bisect_b<BackendH2>, grid, block, kLdsTraceTotal, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
        }
    } else {
        const int waves = resident_waves();
        const dim3 grid((unsigned)(units < waves ? units : waves)), block(64);
        switch (which) {
            case 0: hipLaunchKernelGGL(k_sphere<BackendF32>, grid, block, 0, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            case 1: hipLaunchKernelGGL(k_sampler<BackendF32>, grid, block, 0, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            case 2: hipLaunchKernelGGL(k_bisect_a<BackendF32>, grid, block, 0, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
            default: hipLaunchKernelGGL(k_bisect_b<BackendF32>, grid, block, 0, st, sdf->sdf, sdf->h2_trace, m, a, w); break;
        }
    }
}

}  // namespace iron

using namespace iron;

extern "C" size_t iron_trace_workspace_bytes(int64_t n, const iron_trace_params* p) {
    if (n < 0) return 0;
    // + room for a whole-image chunk table (multi-rank form): 64 Ki chunks
    return ws_layout(n, p).total + align256(sizeof(int) * 65536);
}

// ---- side streams of the split form -------------------------------------------------------------------------------------
namespace iron {
namespace {
struct SideStreams {
    bool ready = false;
    hipStream_t s[kMaxTraceSplits - 1];
    hipEvent_t fork, join[kMaxTraceSplits - 1];
};
std::mutex g_side_mu;           // serialises the enqueue of split calls (streams and events are shared per device)
SideStreams g_side[64];

// IRON_TRACE_SPLIT = k (2..4) runs a call as k parts; unset / 0 / 1 = one part (the measured optimum, see the head of this file)
std::atomic<int> g_trace_split{0};   // iron_set_trace_split; 0 = the environment's / default
int trace_splits(int64_t n) {
    static int from_env = -1;
    if (from_env < 0) {
        const char* e = getenv("IRON_TRACE_SPLIT");
        from_env = e ? atoi(e) : 0;
        if (from_env < 0) from_env = 0;
    }
    int k = g_trace_split.load(std::memory_order_relaxed);
    if (k <= 0) k = from_env;
    if (k < 1) k = 1;
    if (k > kMaxTraceSplits) k = kMaxTraceSplits;
    while (k > 1 && n < (int64_t)k * 256) --k;   // a part is at least two workgroup-passes of rays
    return k;
}

int side_streams(SideStreams** out) {
    int dev = 0;
    IRON_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return IRON_ERR_UNSUPPORTED;
    SideStreams& S = g_side[dev];
    if (!S.ready) {
        int least = 0, greatest = 0;
        IRON_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // lowest priority: a side part's workgroups take the CUs the caller-stream part leaves idle, not the other way round
        for (int i = 0; i < kMaxTraceSplits - 1; ++i) {
            IRON_HIP_TRY(hipStreamCreateWithPriority(&S.s[i], hipStreamNonBlocking, least));
            IRON_HIP_TRY(hipEventCreateWithFlags(&S.join[i], hipEventDisableTiming));
        }
        IRON_HIP_TRY(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
        S.ready = true;
    }
    *out = &S;
    return IRON_OK;
}
}  // namespace
}  // namespace iron

extern "C" int32_t iron_set_trace_split(int32_t parts) {
    return g_trace_split.exchange(parts > 0 ? (parts > kMaxTraceSplits ? kMaxTraceSplits : parts) : 0, std::memory_order_relaxed);
}

extern "C" int iron_trace_phase(int32_t phase, const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps,
                                const float* ray_o, const float* ray_d, const float* near, const float* far,
                                const uint8_t* work, const int64_t* ray_index, int64_t n, int32_t* chunk_iters,
                                int64_t n_chunks, uint8_t* conv, float* points, float* sdf_out, float* dist,
                                iron_trace_stats* stats, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sdf || sdf->desc.kind != IRON_NET_SDF || !p || n < 0 || (phase != 0 && phase != 1)) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (p->n_steps < 2 || p->n_steps > 4096 || p->sphere_tracing_iters < 0 || !(p->sdf_threshold > 0.0f)) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!lin_steps || !ray_o || !ray_d || !near || !far || !work || !conv || !points || !sdf_out || !dist || !workspace)
        return IRON_ERR_BAD_ARG;
    const WsLayout L = ws_layout(n, p);
    if (workspace_bytes < L.total) return IRON_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return IRON_ERR_BAD_ARG;
    if (chunk_iters && (n_chunks < 1 || n_chunks > 65536)) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    TraceWs w0;
    w0.cnt = (TraceCounters*)(base + L.cnt);
    w0.sampler_list = (int*)(base + L.sampler_list);
    w0.root_list = (int*)(base + L.root_list);
    w0.root_lo = (float*)(base + L.lo);
    w0.root_hi = (float*)(base + L.hi);
    w0.root_flo = (float*)(base + L.flo);
    w0.root_fhi = (float*)(base + L.fhi);
    w0.root_k = (int*)(base + L.k);
    w0.chunk_iters = chunk_iters ? chunk_iters : (int*)(base + L.chunk_iters);
    w0.chunk_roots = (int*)(base + L.chunk_roots);
    w0.n_chunks = (int)(chunk_iters ? n_chunks : L.n_chunks);
    w0.cont = (unsigned long long*)(base + L.cont);
    w0.cont_cap = (int)L.cont_cap;
    if (chunk_iters && n_chunks > L.n_chunks) {
        // multi-rank: the chunk table covers the whole image, not just this rank's rays
        if (workspace_bytes < L.total + align256(sizeof(int) * (size_t)n_chunks)) return IRON_ERR_WORKSPACE;
        w0.chunk_roots = (int*)(base + L.total);
    }
    TraceArgs a0;
    a0.ray_o = ray_o; a0.ray_d = ray_d; a0.near = near; a0.far = far; a0.work = work; a0.ray_index = ray_index;
    a0.lin = lin_steps; a0.conv = conv; a0.points = points; a0.sdf = sdf_out; a0.dist = dist;
    a0.ray0 = 0; a0.n = (int)n; a0.n_steps = p->n_steps; a0.iters = p->sphere_tracing_iters; a0.thr = p->sdf_threshold;
    a0.chunk = p->chunk > 0 ? p->chunk : 0;
    if (phase == 0) { const int rce = envelope_begin(sdf); if (rce != IRON_OK) return rce; }
    const bool h2 = h2_sdf_usable(sdf);

    // the parts: rays [b_k, b_{k+1}), own counters, own stretch [b_k, ..) of every list (a part lists at most its own rays);
    // both phases of a call see the same n, hence the same split
    const int parts = trace_splits(n);
    int64_t bnd[kMaxTraceSplits + 1];
    for (int k = 0; k <= parts; ++k) bnd[k] = k == parts ? n : ((n * k / parts) + 31) / 32 * 32;
    if (parts == 2) {   // experiment: IRON_TRACE_SPLIT_FRAC = percent of the rays in the caller-stream part
        static int frac = -1;
        if (frac < 0) { const char* e = getenv("IRON_TRACE_SPLIT_FRAC"); frac = e ? atoi(e) : 50; if (frac < 5 || frac > 95) frac = 50; }
        bnd[1] = ((n * frac / 100) + 31) / 32 * 32;
    }
    SideStreams* S = nullptr;
    std::unique_lock<std::mutex> lock(g_side_mu, std::defer_lock);
    if (parts > 1) {
        lock.lock();
        const int rc = side_streams(&S);
        if (rc != IRON_OK) return rc;
    }
    if (phase == 0) {
        IRON_HIP_TRY(hipMemsetAsync(base + L.cnt, 0, kCntStride * kMaxTraceSplits, st));
        if (chunk_iters) IRON_HIP_TRY(hipMemsetAsync(chunk_iters, 0, sizeof(int) * (size_t)n_chunks, st));
        else IRON_HIP_TRY(hipMemsetAsync(base + L.chunk_iters, 0, align256(sizeof(int) * (size_t)L.n_chunks), st));
        IRON_HIP_TRY(hipMemsetAsync(w0.chunk_roots, 0, sizeof(int) * (size_t)w0.n_chunks, st));
        if (L.cont_cap > 0) IRON_HIP_TRY(hipMemsetAsync(base + L.cont, 0, sizeof(unsigned long long) * (size_t)L.cont_cap, st));
    }
    if (parts > 1) IRON_HIP_TRY(hipEventRecord(S->fork, st));
    // side parts first: their launches are queued before the caller-stream part occupies the chip
    for (int k = parts - 1; k >= 0; --k) {
        hipStream_t sk = k == 0 ? st : S->s[k - 1];
        if (k > 0) IRON_HIP_TRY(hipStreamWaitEvent(sk, S->fork, 0));
        TraceWs w = w0;
        TraceArgs a = a0;
        const int64_t b0 = bnd[k], nk = bnd[k + 1] - bnd[k];
        if (nk <= 0) continue;
        w.cnt = (TraceCounters*)(base + L.cnt + (size_t)k * kCntStride);
        w.sampler_list += b0; w.root_list += b0; w.root_lo += b0; w.root_hi += b0; w.root_flo += b0; w.root_fhi += b0; w.root_k += b0;
        a.ray0 = (int)b0; a.n = (int)nk;
        if (L.cont_cap > 0) { const int64_t per_ray = L.cont_cap / n; w.cont += b0 * per_ray; w.cont_cap = (int)(nk * per_ray); }
        const int64_t tiles = (nk + 31) / 32;
        if (phase == 0) {
            {
                ProfScope ps(IRON_PROF_SPHERE, sk);
                launch_trace_kernel(0, h2, sdf, a, w, tiles, sk);
            }
            {
                ProfScope ps(IRON_PROF_SAMPLER, sk);
                launch_trace_kernel(1, h2, sdf, a, w, (nk + kSamplerSlots - 1) / kSamplerSlots, sk);
            }
            {
                ProfScope ps(IRON_PROF_BISECT_A, sk);
                launch_trace_kernel(2, h2, sdf, a, w, tiles, sk);
            }
        } else {
            ProfScope ps(IRON_PROF_BISECT_B, sk);
            launch_trace_kernel(3, h2, sdf, a, w, tiles, sk);
        }
        if (k > 0) IRON_HIP_TRY(hipEventRecord(S->join[k - 1], sk));
    }
    for (int k = 1; k < parts; ++k)   // join: everything behind this call on the caller's stream sees all parts finished
        if (bnd[k + 1] > bnd[k]) IRON_HIP_TRY(hipStreamWaitEvent(st, S->join[k - 1], 0));
    if (phase == 1 && h2) envelope_scan(sdf, sdf_out, n, nullptr, 1, st);   // envelope guard (envelope.hip): every ray's last sdf value
    if (phase == 1 && stats) hipLaunchKernelGGL(k_trace_stats, dim3(1), dim3(64), 0, st, w0, parts, p->n_steps, stats);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_trace_stage(int32_t stage, const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps,
                                const float* ray_o, const float* ray_d, const float* in0, const float* in1, const float* in2,
                                const float* in3, const uint8_t* work, int64_t n, uint8_t* mask_out, uint8_t* unfinished_out,
                                float* points, float* sdf_out, float* dist, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sdf || sdf->desc.kind != IRON_NET_SDF || !p || n < 0 || stage < 0 || stage > 2) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL - 64) return IRON_ERR_BAD_ARG;
    if (p->n_steps < 2 || p->n_steps > 4096 || p->sphere_tracing_iters < 0 || !(p->sdf_threshold > 0.0f)) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!lin_steps || !ray_o || !ray_d || !in0 || !in1 || !mask_out || !points || !sdf_out || !dist || !workspace) return IRON_ERR_BAD_ARG;
    if (stage == 0 && (!work || !unfinished_out)) return IRON_ERR_BAD_ARG;
    if (stage == 2 && (!in2 || !in3)) return IRON_ERR_BAD_ARG;
    iron_trace_params q = *p;
    q.chunk = 0;   // one reference call = one chunk
    const WsLayout L = ws_layout(n, &q);
    if (workspace_bytes < L.total) return IRON_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return IRON_ERR_BAD_ARG;
    { const int rce = envelope_begin(sdf); if (rce != IRON_OK) return rce; }
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    TraceWs w;
    w.cnt = (TraceCounters*)(base + L.cnt);
    w.sampler_list = (int*)(base + L.sampler_list);
    w.root_list = (int*)(base + L.root_list);
    w.root_lo = (float*)(base + L.lo); w.root_hi = (float*)(base + L.hi);
    w.root_flo = (float*)(base + L.flo); w.root_fhi = (float*)(base + L.fhi);
    w.root_k = (int*)(base + L.k);
    w.chunk_iters = (int*)(base + L.chunk_iters);
    w.chunk_roots = (int*)(base + L.chunk_roots);
    w.n_chunks = 1;
    w.cont = (unsigned long long*)(base + L.cont);
    w.cont_cap = (int)L.cont_cap;
    TraceArgs a;
    a.ray_o = ray_o; a.ray_d = ray_d; a.work = work; a.ray_index = nullptr; a.lin = lin_steps;
    a.conv = mask_out; a.points = points; a.sdf = sdf_out; a.dist = dist;
    a.ray0 = 0; a.n = (int)n; a.n_steps = p->n_steps; a.iters = p->sphere_tracing_iters; a.thr = p->sdf_threshold; a.chunk = 0;
    a.near = in0; a.far = in1;
    const bool h2 = h2_sdf_usable(sdf);
    IRON_HIP_TRY(hipMemsetAsync(base + L.cnt, 0, kCntStride * kMaxTraceSplits, st));
    IRON_HIP_TRY(hipMemsetAsync(base + L.chunk_iters, 0, align256(sizeof(int)), st));
    IRON_HIP_TRY(hipMemsetAsync(base + L.chunk_roots, 0, align256(sizeof(int)), st));
    if (L.cont_cap > 0) IRON_HIP_TRY(hipMemsetAsync(base + L.cont, 0, sizeof(unsigned long long) * (size_t)L.cont_cap, st));
    const int64_t tiles = (n + 31) / 32;
    const unsigned gb = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    if (stage == 0) {            // sphere_tracing (raytracer.py:105-140): in0 = min_dis, in1 = max_dis
        IRON_HIP_TRY(hipMemsetAsync(unfinished_out, 0, (size_t)n, st));
        launch_trace_kernel(0, h2, sdf, a, w, tiles, st);
        hipLaunchKernelGGL(k_stage_mark_list, dim3(gb), dim3(256), 0, st, w.sampler_list, &w.cnt->n_sampler, unfinished_out);
    } else if (stage == 1) {     // ray_sampler (:142-197): in0 = min_dis, in1 = max_dis, every ray sampled on its own interval
        hipLaunchKernelGGL(k_stage_sampler_init, dim3(gb), dim3(256), 0, st, a, w, in0);
        launch_trace_kernel(1, h2, sdf, a, w, (n + kSamplerSlots - 1) / kSamplerSlots, st);
        launch_trace_kernel(2, h2, sdf, a, w, tiles, st);
        launch_trace_kernel(3, h2, sdf, a, w, tiles, st);
    } else {                     // rootfind (:199-220): in0 = f_low, in1 = f_high, in2 = d_low, in3 = d_high
        hipLaunchKernelGGL(k_stage_root_init, dim3(gb), dim3(256), 0, st, w, (int)n, in0, in1, in2, in3);
        launch_trace_kernel(2, h2, sdf, a, w, tiles, st);
        launch_trace_kernel(3, h2, sdf, a, w, tiles, st);
    }
    if (h2) envelope_scan(sdf, sdf_out, n, nullptr, 1, st);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_trace(const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps, const float* ray_o,
                          const float* ray_d, const float* near, const float* far, const uint8_t* work, int64_t n,
                          uint8_t* conv, float* points, float* sdf_out, float* dist, iron_trace_stats* stats,
                          void* workspace, size_t workspace_bytes, void* stream) {
    int rc = iron_trace_phase(0, sdf, p, lin_steps, ray_o, ray_d, near, far, work, nullptr, n, nullptr, 0, conv, points,
                              sdf_out, dist, stats, workspace, workspace_bytes, stream);
    if (rc != IRON_OK) return rc;
    return iron_trace_phase(1, sdf, p, lin_steps, ray_o, ray_d, near, far, work, nullptr, n, nullptr, 0, conv, points,
                            sdf_out, dist, stats, workspace, workspace_bytes, stream);
}
