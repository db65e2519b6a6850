// 3-way bf16 operand splitting ("bf16x6") helpers shared by the GRU forward and backward kernels.
//
// x = h + m + l, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): three 8-bit pieces carry the whole
// 24-bit fp32 mantissa.  a*b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm (dropped terms
// <= 3*2^-24 relative); every partial product is exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16, so the result has fp32-GEMM accuracy at 6*32 matrix-pipe cycles per K=16
// instead of 8*64 for v_mfma_f32_32x32x2_f32.
#pragma once
#include "common.h"

namespace mpnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;      // exact
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;     // exact
    l = (__bf16)r2;
}

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& ph, bf16x8& pm, bf16x8& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 h, m, l;
        split3(x0[j], h, m, l);
        ph[j] = h; pm[j] = m; pl[j] = l;
        split3(x1[j], h, m, l);
        ph[4 + j] = h; pm[4 + j] = m; pl[4 + j] = l;
    }
}

// six partial products of one K=16 step into one accumulator, small terms first
__device__ __forceinline__ void mma6(f32x16& acc, const bf16x8& ah, const bf16x8& am, const bf16x8& al,
                                     const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// Two independent accumulators, issued alternately.  On gfx950 an MFMA that accumulates into the result of the
// previous one costs ~45 cycles even with two waves per SIMD interleaving their chains; with an independent MFMA in
// between the pipe runs at its 32 (tools/microbench/mfma_valu_overlap.hip: 7.6 vs 5.4 ms for the same MFMA count).
// mma6x2_a: both products share the A pieces (two column blocks); mma6x2_b: both share the B pieces (two row blocks).
__device__ __forceinline__ void mma6x2_a(f32x16& c0, f32x16& c1, const bf16x8& ah, const bf16x8& am, const bf16x8& al,
                                         const bf16x8& b0h, const bf16x8& b0m, const bf16x8& b0l, const bf16x8& b1h,
                                         const bf16x8& b1m, const bf16x8& b1l) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, c1, 0, 0, 0);
}

__device__ __forceinline__ void mma6x2_b(f32x16& c0, f32x16& c1, const bf16x8& a0h, const bf16x8& a0m, const bf16x8& a0l,
                                         const bf16x8& a1h, const bf16x8& a1m, const bf16x8& a1l, const bf16x8& bh,
                                         const bf16x8& bm, const bf16x8& bl) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bh, c1, 0, 0, 0);
}

// general form: two unrelated products
__device__ __forceinline__ void mma6x2(f32x16& c0, const bf16x8& a0h, const bf16x8& a0m, const bf16x8& a0l,
                                       const bf16x8& b0h, const bf16x8& b0m, const bf16x8& b0l, f32x16& c1,
                                       const bf16x8& a1h, const bf16x8& a1m, const bf16x8& a1l, const bf16x8& b1h,
                                       const bf16x8& b1m, const bf16x8& b1l) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, b0m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, b1m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b1h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, b1h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0m, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1m, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1h, c1, 0, 0, 0);
}

}  // namespace mpnn
