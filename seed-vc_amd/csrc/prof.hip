// Optional per-kernel-class launch timing with HIP events on the launch stream (bench.py roofline leg).
#include <vector>

#include "../../include/seedvc_hip.h"
#include "common.h"

namespace svc {
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_cur = nullptr;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
bool prof_enabled() { return g_on; }
void prof_begin(int cls, hipStream_t st) {
    g_cur = get_event();
    (void)hipEventRecord(g_cur, st);
}
void prof_end(int cls, double flops, double bytes, hipStream_t st) {
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, st);
    g_recs.push_back({g_cur, e, cls, flops, bytes});
    g_cur = nullptr;
}
}  // namespace svc

extern "C" {
int svc_prof_enable(int on) {
    svc::g_on = on != 0;
    return 0;
}
// out[cls*4 + {0,1,2,3}] = {launches, total ms, total algorithmic flops, total algorithmic bytes}; clears the records
int svc_prof_collect(double* out, int n_cls) {
    using namespace svc;
    for (int i = 0; i < n_cls * 4; ++i) out[i] = 0.0;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("prof: event sync failed"); return 1; }
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (r.cls < n_cls) {
            out[r.cls * 4 + 0] += 1.0;
            out[r.cls * 4 + 1] += ms;
            out[r.cls * 4 + 2] += r.flops;
            out[r.cls * 4 + 3] += r.bytes;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}
}
