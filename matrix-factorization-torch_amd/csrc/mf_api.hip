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
