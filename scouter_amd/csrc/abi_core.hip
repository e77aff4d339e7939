// ABI plumbing: version, thread-local error text, optional per-kernel-class hipEvent timing.
#include "common.h"

#include <stdarg.h>
#include <mutex>
#include <vector>

#define SCOUTER_ABI_VERSION 1

static thread_local char g_err[512] = "";

void sc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sc_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SC_ERR_LAUNCH;
    }
    return SC_OK;
}

extern "C" int scouter_abi_version(void) { return SCOUTER_ABI_VERSION; }
extern "C" const char* scouter_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------------------
// profiling: when enabled every launch of a kernel class is bracketed by two hipEvents on ITS stream
// ---------------------------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; const char* name; double flops, bytes; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ScProfScope::ScProfScope(const char* name, hipStream_t st, double flops, double bytes) : stream(st), slot(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, st);
    slot = (int)g_prof.size();
    g_prof.push_back(r);
}
ScProfScope::~ScProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_prof[slot].b, stream);
}

extern "C" void scouter_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

// Synchronises the recorded events and writes one line per kernel name into `buf`:
//   name <tab> launches <tab> total_ms <tab> algorithmic_flops <tab> algorithmic_bytes \n
// then clears the log.  Returns the number of bytes written (truncated to cap-1), or -1 on a bad buffer.
extern "C" int scouter_prof_collect(char* buf, int cap) {
    if (!buf || cap <= 0) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Agg { const char* name; double n, ms, fl, by; };
    std::vector<Agg> aggs;
    for (auto& r : g_prof) {
        float ms = 0.f;
        hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg* a = nullptr;
            for (auto& x : aggs) if (strcmp(x.name, r.name) == 0) { a = &x; break; }
            if (!a) { aggs.push_back(Agg{r.name, 0, 0, 0, 0}); a = &aggs.back(); }
            a->n += 1; a->ms += ms; a->fl += r.flops; a->by += r.bytes;
        }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    g_prof.clear();
    int off = 0;
    buf[0] = 0;
    for (auto& a : aggs) {
        int w = snprintf(buf + off, cap - off, "%s\t%.0f\t%.6f\t%.6e\t%.6e\n", a.name, a.n, a.ms, a.fl, a.by);
        if (w < 0 || w >= cap - off) break;
        off += w;
    }
    return off;
}
