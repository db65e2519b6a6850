// Version / error plumbing of the C ABI (include/mpnn_amd.h).
#include <stdarg.h>

#include "common.h"

namespace mpnn {
static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mpnn

extern "C" int mpnn_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* mpnn_last_error_string(void) { return mpnn::g_err; }
