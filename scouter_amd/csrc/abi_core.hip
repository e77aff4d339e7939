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
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ScProfScope::ScProfScope(int cls_, hipStream_t st, double flops, double bytes) : cls(cls_), stream(st), slot(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.cls = cls_; r.flops = flops; r.bytes = bytes;
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

// Synchronises the recorded events, adds {launch count, total ms, algorithmic flops, algorithmic bytes} per class
// into out[SC_PROF_NCLASS][4] and clears the log.  Returns the number of classes.
extern "C" int scouter_prof_collect(double* out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < SC_PROF_NCLASS * 4; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        float ms = 0.f;
        hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            out[r.cls * 4 + 0] += 1.0;
            out[r.cls * 4 + 1] += ms;
            out[r.cls * 4 + 2] += r.flops;
            out[r.cls * 4 + 3] += r.bytes;
        }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    g_prof.clear();
    return SC_PROF_NCLASS;
}
