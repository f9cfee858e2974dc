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
// created or recorded unless mf_timing_enable(k >= 1) was called (k: time every k-th launch of each
// name -- an event pair costs a few microseconds of stream time around the kernel it brackets).
#include <map>
#include <string>
#include <vector>

struct TimedSpan {
    hipEvent_t start, stop;
};
struct TimedName {
    std::vector<TimedSpan> spans;
    long long calls = 0;
    bool open = false;             // the current launch is being timed
};
static int g_every = 0;            // 0: off; k: every k-th launch of each name is timed
static std::map<std::string, TimedName> g_spans;

bool mf_timing_on() { return g_every > 0; }

void mf_timing_begin(const char* name, hipStream_t s) {
    TimedName& tn = g_spans[name];
    tn.open = (tn.calls++ % g_every) == 0;
    if (!tn.open) return;
    TimedSpan sp;
    if (hipEventCreate(&sp.start) != hipSuccess || hipEventCreate(&sp.stop) != hipSuccess) {
        tn.open = false;
        return;
    }
    (void)hipEventRecord(sp.start, s);
    tn.spans.push_back(sp);
}

void mf_timing_end(const char* name, hipStream_t s) {
    auto it = g_spans.find(name);
    if (it == g_spans.end() || !it->second.open) return;
    (void)hipEventRecord(it->second.spans.back().stop, s);
    it->second.open = false;
}

extern "C" void mf_timing_enable(int every) { g_every = every > 0 ? every : 0; }

extern "C" void mf_timing_reset(void) {
    for (auto& kv : g_spans)
        for (auto& sp : kv.second.spans) {
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
        for (auto& sp : it->second.spans) {
            float ms = 0.f;
            if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
                tot += ms;
                ++n;
            }
        }
    if (total_ms) *total_ms = tot;
    return n;
}
