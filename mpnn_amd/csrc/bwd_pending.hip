// TEMPORARY: backward entry points not written yet -- they fail loudly (no silent fallback).
// Replaced by real kernels in edge_message_bwd.hip / gru_bwd.hip.
#include "common.h"
using namespace mpnn;

extern "C" int mpnn_edge_message_bwd_f32(const float*, const float*, const int32_t*, const int32_t*, const int32_t*,
                                         const float*, const float*, float*, float*, int64_t, int64_t, int, int, int,
                                         void*) {
    set_error("mpnn_edge_message_bwd_f32: not implemented in this build");
    return MPNN_EINVAL;
}
extern "C" size_t mpnn_gru_bwd_workspace_bytes(int64_t V, int H) { return (size_t)V * 6 * H * sizeof(float); }
extern "C" int mpnn_gru_update_bwd_f32(const float*, const float*, const float*, const float*, const float*,
                                       const float*, const float*, float*, float*, float*, float*, float*, float*,
                                       void*, size_t, int64_t, int, void*) {
    set_error("mpnn_gru_update_bwd_f32: not implemented in this build");
    return MPNN_EINVAL;
}
