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

// Environment switch (README "Switches"), read ONCE per process by
// mpnn_init() or by the first call that needs one (thread-safe static initialisation in capi.hip).
struct Switches {
    bool math_fp32;          // MPNN_GRU_MATH=fp32: dense contractions on the fp32 matrix pipe (strict fp32 MFMA, no operand splits)
    bool gru_bwd_pieces;     // MPNN_GRU_BWD=pieces: the round-3 wide GRU backward (gate gradients through an HBM workspace), for A/B runs
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
