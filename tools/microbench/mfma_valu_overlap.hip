// Does VALU work of one wave overlap the MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// One 8-wave block per CU (two waves per SIMD).  mode 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA and
// waves 4-7 VALU (one of each per SIMD); 3: every wave alternates one MFMA with K VALU ops.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int K, int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[NACC] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v0 = threadIdx.x, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wv < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wv >= 4);
    for (int it = 0; it < iters; it += NACC) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {               // compile-time accumulator index
            if (do_mfma) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int j = 0; j < K; j += 4) {       // four independent fma chains
                    v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 1.2345e30f) out[0] = s;
}

template <int MODE, int K, int NACC>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, K, NACC>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, K, NACC>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 200000;
    printf("per iteration and wave: 1 MFMA 32x32x16 bf16 (32 pipe cycles) and/or K full-rate VALU fmas\n");
    printf("one accumulator (every MFMA depends on the previous one of its wave):\n");
    printf("  all-MFMA %.3f ms\n", run<0, 8, 1>(d, iters));
    printf("  K=8 : all-VALU %.3f ms | MFMA waves + VALU waves %.3f ms | interleaved in every wave %.3f ms\n", run<1, 8, 1>(d, iters), run<2, 8, 1>(d, iters), run<3, 8, 1>(d, iters));
    printf("  K=16: all-VALU %.3f ms | MFMA waves + VALU waves %.3f ms | interleaved in every wave %.3f ms\n", run<1, 16, 1>(d, iters), run<2, 16, 1>(d, iters), run<3, 16, 1>(d, iters));
    printf("four accumulators in rotation (a wave's consecutive MFMAs are independent):\n");
    printf("  all-MFMA %.3f ms\n", run<0, 8, 4>(d, iters));
    printf("  K=4 : all-VALU %.3f ms | MFMA waves + VALU waves %.3f ms | interleaved in every wave %.3f ms\n", run<1, 4, 4>(d, iters), run<2, 4, 4>(d, iters), run<3, 4, 4>(d, iters));
    printf("  K=8 : all-VALU %.3f ms | MFMA waves + VALU waves %.3f ms | interleaved in every wave %.3f ms\n", run<1, 8, 4>(d, iters), run<2, 8, 4>(d, iters), run<3, 8, 4>(d, iters));
    printf("  K=16: all-VALU %.3f ms | MFMA waves + VALU waves %.3f ms | interleaved in every wave %.3f ms\n", run<1, 16, 4>(d, iters), run<2, 16, 4>(d, iters), run<3, 16, 4>(d, iters));
    return 0;
}
