// Error reporting + version for the C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int mm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mm_fail(MM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MM_OK;
}

extern "C" {
const char* mm_last_error(void) { return g_err; }
int mm_abi_version(void) { return 5; }
}
