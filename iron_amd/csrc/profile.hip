// Per-kernel device timing with hipEvents on the caller's stream (diagnostics for bench.py).
#include <atomic>
#include <mutex>
#include <vector>
#include "iron_common.h"

namespace iron {
namespace {
struct Pending { int kind; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_open[IRON_PROF_KINDS];
}  // namespace

void prof_begin(int kind, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_open[kind].push_back(e);
}

void prof_end(int kind, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || g_open[kind].empty()) return;
    hipEvent_t a = g_open[kind].back();
    g_open[kind].pop_back();
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
    (void)hipEventRecord(b, st);
    g_pending.push_back({kind, a, b});
}
}  // namespace iron

using namespace iron;

extern "C" int iron_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return IRON_OK;
}

extern "C" int iron_profile_read(double* ms, int64_t* launches) {
    if (!ms || !launches) return IRON_ERR_BAD_ARG;
    std::vector<Pending> todo;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        todo.swap(g_pending);
    }
    int rc = IRON_OK;
    for (auto& p : todo) {
        float t = 0.0f;
        hipError_t e = hipEventSynchronize(p.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, p.a, p.b);
        if (e == hipSuccess) { ms[p.kind] += (double)t; launches[p.kind] += 1; }
        else rc = hip_fail(e);
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    return rc;
}

// ---- CU budget -------------------------------------------------------------------------------------------------------------------
// The persistent kernels size their grids to the chip (one workgroup per CU: the h2 kernels' LDS ring fills a CU).  A caller that
// wants two launch sequences to run side by side on two streams (render_camera: shading of the hits beside the silhouette pass, which
// is a chain of short latency-bound launches) gives each a share of the CUs for the duration of its launches.
namespace iron {
namespace {
std::atomic<int> g_cu_limit{0};
}
int cu_total() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount;
    return cached;
}
int cu_budget() {
    const int total = cu_total(), lim = g_cu_limit.load(std::memory_order_relaxed);
    return (lim > 0 && lim < total) ? lim : total;
}
}  // namespace iron

extern "C" int32_t iron_set_cu_limit(int32_t n_cus) {
    iron::g_cu_limit.store(n_cus > 0 ? n_cus : 0, std::memory_order_relaxed);
    return iron::cu_total();
}
