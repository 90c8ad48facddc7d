// Numeric envelope of the h2 (split-fp16) core.
//
// The h2 kernels carry every fp32 operand as two fp16 pieces, so an ACTIVATION or feature of magnitude >= 65 504 (or a
// non-finite one) leaves fp16's range where the reference's plain fp32 arithmetic (models/fields.py:82-98, 203-239) would
// still be exact.  (Weights are checked at pack time: pack_h2.hip.)  Such an operand becomes +-inf in the split and the
// value that leaves the network is non-finite -- or the correct saturated limit where an activation maps -inf to 0 -- so
// the guard sits where outputs are written, not in the epilogue: behind every entry point that ran a network on the h2
// core a small kernel scans the outputs of the call; a non-finite value raises the network's flag (a word in pinned host
// memory the device writes through).  The NEXT entry on that handle sees the flag and moves the network to the exact-fp32
// MFMA core for good (or returns IRON_ERR_RANGE with IRON_H2_OVERFLOW=error); iron_net_numeric_status() reports it.  The
// call that overflowed has returned non-finite values -- loud, not plausible-looking -- and the callers that can afford a
// synchronisation re-run it (iron_amd/fields.py: IRON_H2_OVERFLOW=rerun).
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "iron_common.h"

namespace iron {

__global__ void k_scan_nonfinite(const float* __restrict__ p, long long n_rows, const int* __restrict__ count_ptr, int width,
                                 int* __restrict__ flag_host) {
    const long long rows = count_ptr ? (long long)*count_ptr : n_rows;
    const long long n = rows * width;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad |= !(fabsf(p[i]) <= 3.0e38f);   // inf and NaN
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) {
        *reinterpret_cast<volatile int*>(flag_host) = 1;
        __threadfence_system();
    }
}

static int overflow_mode() {   // 0 = move the network to the exact core (default), 1 = IRON_ERR_RANGE
    static int m = -1;
    if (m < 0) {
        const char* e = getenv("IRON_H2_OVERFLOW");
        m = (e && e[0] == 'e') ? 1 : 0;
    }
    return m;
}

// Flag words come from one pinned, device-mapped page pool per process: a training loop re-packs its networks every step, and a
// hipHostMalloc / hipHostFree pair per handle would put a host-side map / unmap (and the implicit synchronisation of the free) into
// every step.
namespace {
std::mutex g_pool_mu;
int* g_pool_host = nullptr;
int* g_pool_dev = nullptr;
std::vector<int> g_pool_free;
constexpr int kPoolSlots = 4096, kSlotInts = 16;   // 64 bytes per flag: one cache line each
}  // namespace

int envelope_create(iron_net* net) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool_host) {
        void* h = nullptr;
        IRON_HIP_TRY(hipHostMalloc(&h, (size_t)kPoolSlots * kSlotInts * sizeof(int), hipHostMallocMapped));
        void* d = nullptr;
        IRON_HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
        memset(h, 0, (size_t)kPoolSlots * kSlotInts * sizeof(int));
        g_pool_host = (int*)h;
        g_pool_dev = (int*)d;
        g_pool_free.reserve(kPoolSlots);
        for (int i = kPoolSlots - 1; i >= 0; --i) g_pool_free.push_back(i);
    }
    if (g_pool_free.empty()) { net->flag_host = net->flag_dev = nullptr; return IRON_OK; }   // > 4096 live handles: unguarded, not an error
    const int slot = g_pool_free.back();
    g_pool_free.pop_back();
    net->flag_host = g_pool_host + (size_t)slot * kSlotInts;
    net->flag_dev = g_pool_dev + (size_t)slot * kSlotInts;
    *(volatile int*)net->flag_host = 0;
    return IRON_OK;
}

void envelope_destroy(iron_net* net) {
    if (net->flag_host) {
        // a scan kernel of this handle may still be in flight: its (rare) write lands in a slot that is zeroed again when it is handed out
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool_free.push_back((int)((net->flag_host - g_pool_host) / kSlotInts));
    }
    net->flag_host = net->flag_dev = nullptr;
}

int envelope_begin(const iron_net* cnet) {
    iron_net* net = const_cast<iron_net*>(cnet);   // the sticky status is the one mutable part of a handle
    if (!net || !net->flag_host) return IRON_OK;
    if (*(volatile int*)net->flag_host) {
        net->overflow_seen = 1;
        if (overflow_mode() == 1) return IRON_ERR_RANGE;
        net->h2_disabled = 1;
        *(volatile int*)net->flag_host = 0;
    }
    return IRON_OK;
}

void envelope_scan(const iron_net* net, const float* p, int64_t n_rows, const int* count_ptr, int width, hipStream_t st) {
    if (!net || !net->flag_dev || !p || n_rows <= 0 || !h2_enabled(net)) return;   // exact-core results need no guard
    const long long total = (long long)n_rows * width;
    long long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_scan_nonfinite, dim3((unsigned)blocks), dim3(256), 0, st, p, (long long)n_rows, count_ptr, width, net->flag_dev);
}

}  // namespace iron

using namespace iron;

extern "C" int iron_net_numeric_status(const iron_net_t* net, int32_t* status_out, void* stream) {
    if (!net || !status_out) return IRON_ERR_BAD_ARG;
    IRON_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    iron_net* n = const_cast<iron_net*>(net);
    int s = 0;
    if (n->flag_host && *(volatile int*)n->flag_host) { n->overflow_seen = 1; s |= 4; }
    if (n->overflow_seen) s |= 1;
    if (n->h2_disabled) s |= 2;
    *status_out = s;
    return IRON_OK;
}

extern "C" int iron_net_force_exact(iron_net_t* net, int32_t on) {
    if (!net) return IRON_ERR_BAD_ARG;
    net->h2_disabled = on ? 1 : 0;
    if (!on) { net->overflow_seen = 0; if (net->flag_host) *(volatile int*)net->flag_host = 0; }
    return IRON_OK;
}
