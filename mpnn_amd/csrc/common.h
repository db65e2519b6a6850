// Shared host/device helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mpnn_amd.h"

namespace mpnn {

constexpr int kWave = 64;
constexpr int kNumXcd = 8;     // MI355X: 8 XCDs, private L2 each; blocks b and b+8 share an XCD

void set_error(const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MPNN_ELAUNCH;
    }
    return MPNN_OK;
}

// One-time opt-in of a kernel (family) to more than 64 KB of dynamic LDS.  Used as the initialiser of a function-local
// `static const hipError_t` (C++11: initialised exactly once, thread-safely); every launch checks the cached result.
struct LdsOptIn {
    hipError_t err = hipSuccess;
    void operator()(const void* kernel, hipFuncAttribute attr, int bytes) {
        const hipError_t e = hipFuncSetAttribute(kernel, attr, bytes);
        if (err == hipSuccess) err = e;
    }
};
inline int lds_opt_in_failed(hipError_t e) {
    set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    return MPNN_ELAUNCH;
}

// Environment switches (A/B alternates of the default kernels; README "Switches"), read ONCE per process by
// mpnn_init() or by the first call that needs one (thread-safe static initialisation in capi.hip).
struct Switches {
    bool math_fp32;          // MPNN_GRU_MATH=fp32: dense contractions on the fp32 matrix pipe
    bool da64_direct;        // MPNN_DA64_DIRECT: fp32 register-direct dA kernel at width 64
    bool gru128_sliced;      // MPNN_GRU128_SLICED: resident-slice GRU forward at width 128
    bool gru128_sliced_dx;   // MPNN_GRU128_SLICED_DX: resident-slice dm/dh kernel at width 128
    bool gru256_narrow;      // MPNN_GRU256_NARROW: 32-feature wave tiles in the streamed width-256 GRU
    bool gru_bwd_uniform;    // MPNN_GRU_BWD_UNIFORM: all-waves-identical GRU backward at width 64
    bool gru_bwd_fp32tile;   // MPNN_GRU_BWD_FP32TILE: fp32 LDS tile in the GRU backward at width 64
    bool gru_fwd_bf16;       // MPNN_GRU_FWD_BF16: GRU forward (64 / 128 / 256) on three bf16 pieces instead of two row-guarded fp16 pieces
    bool gru_bwd_bf16;       // MPNN_GRU_BWD_BF16: width-64 GRU backward on three bf16 pieces (gru_bwd_presplit.hip) instead of two fp16 pieces (gru_bwd_f16.hip)
    bool gru_dx_slice64;     // MPNN_GRU_DX_SLICE64: width-128/256 dm|dh on 64-column slices instead of 128-column ones
    bool gru_dx_insplit;     // MPNN_GRU_DX_INSPLIT: the 128-column dm|dh kernel splits its weight chunks itself (no pre-split workspace)
    int segsum_variant;      // MPNN_SEGSUM_VARIANT: 1 = one atom per lane group, 2 = cached loads/stores, 3 = default
};
const Switches& switches();

#define MPNN_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ::mpnn::set_error(__VA_ARGS__);  \
            return MPNN_EINVAL;              \
        }                                    \
    } while (0)

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32 accumulator tile of v_mfma_f32_32x32x2_f32: register `reg` of lane `lane`
// holds element (row, col) = ((reg&3) + 8*(reg>>2) + 4*(lane>>5), lane&31).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// v_readlane of a float: the builtin is typed int, so the value must travel as bits (a plain call would
// value-convert, i.e. truncate)
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

}  // namespace mpnn
