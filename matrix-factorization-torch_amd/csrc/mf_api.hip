// mf_api.hip -- error plumbing of libmf_hip.so (C ABI in include/mf_hip.h).
#include <cstdarg>
#include <cstdio>

#include "mf_common.h"

static thread_local char g_err[512] = "";

int mf_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mf_set_error(MF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MF_OK;
}

extern "C" const char* mf_last_error(void) { return g_err; }
extern "C" int mf_version(void) { return 100; }

// ---------------------------------------------------------------- kernel timing --
// Optional HIP-event timing of the dominant kernels, recorded on the stream the
// kernel is launched on (bench.py's roofline leg).  Off by default: no event is
// created or recorded unless mf_timing_enable(1) was called.
#include <map>
#include <string>
#include <vector>

struct TimedSpan {
    hipEvent_t start, stop;
};
static bool g_timing = false;
static std::map<std::string, std::vector<TimedSpan>> g_spans;

bool mf_timing_on() { return g_timing; }

void mf_timing_begin(const char* name, hipStream_t s) {
    TimedSpan sp;
    if (hipEventCreate(&sp.start) != hipSuccess || hipEventCreate(&sp.stop) != hipSuccess) return;
    (void)hipEventRecord(sp.start, s);
    g_spans[name].push_back(sp);
}

void mf_timing_end(const char* name, hipStream_t s) {
    auto it = g_spans.find(name);
    if (it == g_spans.end() || it->second.empty()) return;
    (void)hipEventRecord(it->second.back().stop, s);
}

extern "C" void mf_timing_enable(int on) { g_timing = on != 0; }

extern "C" void mf_timing_reset(void) {
    for (auto& kv : g_spans)
        for (auto& sp : kv.second) {
            (void)hipEventDestroy(sp.start);
            (void)hipEventDestroy(sp.stop);
        }
    g_spans.clear();
}

// Blocks until the recorded spans of `name` have completed; returns their count.
extern "C" int64_t mf_timing_get(const char* name, double* total_ms) {
    double tot = 0.0;
    int64_t n = 0;
    auto it = g_spans.find(name);
    if (it != g_spans.end())
        for (auto& sp : it->second) {
            float ms = 0.f;
            if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
                tot += ms;
                ++n;
            }
        }
    if (total_ms) *total_ms = tot;
    return n;
}
