// Host-side plumbing of the C-ABI: error reporting and version.
#include "sc_common.h"
#include <string.h>

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

extern "C" int sc_version(void) { return 2; }
extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" int sc_tuning_build(void) {
#ifdef SC_TUNING
    return 1;
#else
    return 0;
#endif
}
extern "C" int sc_struct_size(const char *name) {
    if (!name) return -1;
#define SC_SIZE_OF(T) if (!strcmp(name, #T)) return (int)sizeof(T);
    SC_SIZE_OF(sc_potential) SC_SIZE_OF(sc_state) SC_SIZE_OF(sc_hk_consts) SC_SIZE_OF(sc_overlap_consts)
    SC_SIZE_OF(sc_nac_consts) SC_SIZE_OF(sc_wm_consts) SC_SIZE_OF(sc_gdml_model) SC_SIZE_OF(sc_dense_scratch)
    SC_SIZE_OF(sc_multi_scratch)
#undef SC_SIZE_OF
    return -1;
}
extern "C" const char *sc_last_error(void) { return g_error; }
