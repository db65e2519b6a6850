// Version / error plumbing of the C ABI (include/mpnn_amd.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mpnn {
static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const Switches& switches() {
    static const Switches sw = [] {
        Switches s;
        const char* m = getenv("MPNN_GRU_MATH");
        s.math_fp32 = m && !strcmp(m, "fp32");
        s.da64_direct = getenv("MPNN_DA64_DIRECT") != nullptr;
        s.gru128_sliced = getenv("MPNN_GRU128_SLICED") != nullptr;
        s.gru128_sliced_dx = getenv("MPNN_GRU128_SLICED_DX") != nullptr;
        s.gru256_narrow = getenv("MPNN_GRU256_NARROW") != nullptr;
        s.gru_bwd_uniform = getenv("MPNN_GRU_BWD_UNIFORM") != nullptr;
        s.gru_bwd_fp32tile = getenv("MPNN_GRU_BWD_FP32TILE") != nullptr;
        s.gru_fwd_bf16 = getenv("MPNN_GRU_FWD_BF16") != nullptr;
        s.gru_bwd_bf16 = getenv("MPNN_GRU_BWD_BF16") != nullptr;
        s.gru_dx_slice64 = getenv("MPNN_GRU_DX_SLICE64") != nullptr;
        s.gru_dx_insplit = getenv("MPNN_GRU_DX_INSPLIT") != nullptr;
        const char* v = getenv("MPNN_SEGSUM_VARIANT");
        s.segsum_variant = v ? atoi(v) : 3;
        return s;
    }();
    return sw;
}
}  // namespace mpnn

extern "C" int mpnn_version(void) { return 200; /* 0.2.0 */ }

extern "C" int mpnn_init(void) {
    (void)mpnn::switches();
    return MPNN_OK;
}

extern "C" const char* mpnn_last_error_string(void) { return mpnn::g_err; }
