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
        const char* b = getenv("MPNN_GRU_BWD");
        s.gru_bwd_pieces = b && !strcmp(b, "pieces");
        return s;
    }();
    return sw;
}
}  // namespace mpnn

// 0.3.0: mpnn_gru_update_f32 takes a workspace, mpnn_message_aggregate_wide_f32 added, mpnn_message_aggregate_bwd_da_f32
//        removed.  0.3.1: mpnn_gru_update_norm_f32 / _norm_bwd_f32 / _norm_supported, mpnn_norm_fold_f32,
//        mpnn_norm_bwd_consts_f32, mpnn_norm_bwd_sums_f32 added; the H = 128 / 256 backward workspace holds one scale per atom.
// 0.4.0: the fused norm's backward sums run against the norm's raw input: mpnn_gru_update_norm_bwd_f32 takes h_raw,
//        mpnn_norm_bwd_sums_f32 takes y_raw, mpnn_norm_bwd_consts_f32 lost its bias argument.
extern "C" int mpnn_version(void) { return 400; }

extern "C" int mpnn_init(void) {
    (void)mpnn::switches();
    return MPNN_OK;
}

extern "C" const char* mpnn_last_error_string(void) { return mpnn::g_err; }
