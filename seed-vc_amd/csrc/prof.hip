// Optional per-kernel-class launch timing with HIP events on the launch stream (bench.py roofline leg).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/seedvc_hip.h"
#include "common.h"

namespace svc {
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; unsigned long long tag; };
// Several host threads may drive handles on different streams (pipeline.Lanes): the record list and the event pool are
// shared under a mutex, the open begin-event belongs to the calling thread.
bool g_on = false;
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local hipEvent_t g_cur = nullptr;
hipEvent_t get_event() {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    }
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
void prof_end(int cls, double flops, double bytes, hipStream_t st, unsigned long long tag) {
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, st);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_recs.push_back({g_cur, e, cls, flops, bytes, tag});
    }
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
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < n_cls * 4; ++i) out[i] = 0.0;
    // SVC_PROF_DUMP=<file>: also append one line per (class, shape tag) -- tuning aid, see tools/shape_report.py
    const char* dump = getenv("SVC_PROF_DUMP");
    struct Agg { double n = 0, ms = 0, fl = 0, by = 0; };
    std::map<std::pair<int, unsigned long long>, Agg> agg;
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
        if (dump) {
            Agg& a = agg[{r.cls, r.tag}];
            a.n += 1; a.ms += ms; a.fl += r.flops; a.by += r.bytes;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    if (dump) {
        if (FILE* f = fopen(dump, "a")) {
            for (auto& kv : agg) {
                const unsigned long long t = kv.first.second;
                fprintf(f, "%d,%llu,%llu,%llu,%llu,%.0f,%.4f,%.4e,%.4e\n", kv.first.first, t >> 40, (t >> 20) & 0xFFFFF,
                        (t >> 4) & 0xFFFF, t & 0xF, kv.second.n, kv.second.ms, kv.second.fl, kv.second.by);
            }
            fclose(f);
        }
    }
    return 0;
}
}
