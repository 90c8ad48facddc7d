// Per-kernel device timing with hipEvents on the caller's stream (diagnostics for bench.py).
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
