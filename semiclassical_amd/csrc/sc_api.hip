// Host-side plumbing of the C-ABI: error reporting and version.
#include "sc_common.h"

static thread_local char g_error[512] = "";

int sc_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int sc_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SC_OK;
}

extern "C" int sc_version(void) { return 1; }
extern "C" const char *sc_last_error(void) { return g_error; }
