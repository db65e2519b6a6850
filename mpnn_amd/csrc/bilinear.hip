// BiLiniearEdgeNetwork (sic) message: the bond features of a pair ARE a (nf x nf x nf) tensor,
//   out[b,i,j,k] = sum_{a,c} afm[b,j,a] * T[b,i,j][a,k,c] * afm[b,i,c]
// replaces: mpnn_functions/message/bilinear_edge_network.py:25-37 (two batched matmuls over the dense padded batch).
// HBM-bound: every pair's nf^3 floats are read once (4*nf^3 bytes per pair against 2*nf^3 multiply-adds).  A thread owns one
// (pair, k) output: for each a it reads the nf contiguous floats T[a][k][:], so the threads of a pair walk T[a] as one
// contiguous run of nf*nf floats.
#include "common.h"

namespace mpnn {

template <int NF>
__global__ void __launch_bounds__(256) bilinear_message_kernel(const float* __restrict__ afm, const float* __restrict__ bfm,
                                                               float* __restrict__ out, int64_t pairs, int N) {
    const int64_t total = pairs * NF;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / NF;
        const int k = (int)(idx % NF);
        const int64_t bi = p / N;                        // b * N + i
        const int64_t bj = (bi / N) * N + p % N;         // b * N + j
        const float* T = bfm + p * (NF * NF * NF) + k * NF;
        float hi[NF], acc = 0.f;
#pragma unroll
        for (int c = 0; c < NF; ++c) hi[c] = afm[bi * NF + c];
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < NF; ++c) d += T[a * NF * NF + c] * hi[c];
            acc += afm[bj * NF + a] * d;
        }
        out[idx] = acc;
    }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_bilinear_message_f32(const float* afm, const float* bfm, float* out, int64_t B, int N, int nf,
                                         void* stream) {
    MPNN_REQUIRE(B >= 0 && N >= 0 && nf >= 1 && nf <= 8, "mpnn_bilinear_message_f32: nf=%d out of range (1..8)", nf);
    const int64_t pairs = B * (int64_t)N * N;
    if (pairs == 0) return MPNN_OK;
    MPNN_REQUIRE(afm && bfm && out, "mpnn_bilinear_message_f32: NULL buffer");
    int64_t g = ceil_div(pairs * nf, 256);
    if (g > 256 * 16) g = 256 * 16;
    hipStream_t s = (hipStream_t)stream;
#define MPNN_BIL(NFV) \
    case NFV: hipLaunchKernelGGL(bilinear_message_kernel<NFV>, dim3((unsigned)g), dim3(256), 0, s, afm, bfm, out, pairs, N); break;
    switch (nf) {
        MPNN_BIL(1) MPNN_BIL(2) MPNN_BIL(3) MPNN_BIL(4) MPNN_BIL(5) MPNN_BIL(6) MPNN_BIL(7) MPNN_BIL(8)
    }
#undef MPNN_BIL
    return launch_status("mpnn_bilinear_message_f32");
}
