// Typed edge message when there are MANY matrices and few edges per matrix (continuous bond features: every
// undirected bond has its own feature row, K ~ E/2, so a "type" is one or two edges).  There is no dense
// contraction left to put on the matrix cores: each A_k (mf x nf) is read once and multiplied by one or two
// vectors -- HBM-bound streaming of A, 4*mf*nf bytes per type.
//
// One wave per type.  A 64-column strip of four consecutive matrix rows is one coalesced 1-KB load (16 lanes x
// float4 per row); the same registers serve every edge of the type:
//   forward   msg[e][m]  = sum_n A[m][n] x_e[n]            16-lane shuffle reduction per row
//   dx        dx[e][n]   = sum_m A[m][n] dmsg_e[m]         accumulated per lane, 4-way reduction at the end
//   dA        dA[m][n]  += sum_e dmsg_e[m] x_e[n]          one read-modify-write pass, no atomics (one wave per type)
// x_e = gate[e] * h[src(e)] (gate optional).  Edges of a type are taken EJ = 2 at a time (both directions of a bond).
#include "common.h"

namespace mpnn {

constexpr int kEJ = 2;

// mode 0: forward, 1: dx, 2: dA
template <int MODE>
__global__ void __launch_bounds__(256) edge_pertype_kernel(const float* __restrict__ h, const float* A,
                                                           const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ type_ptr,
                                                           const float* __restrict__ gate,
                                                           const float* __restrict__ dmsg, float* __restrict__ out,
                                                           float* dA, int K, int nf, int mf) {
    const int lane = threadIdx.x & 63;
    const int cg = lane & 15, rs = lane >> 4;              // column group (4 floats), row inside a 4-row group
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (nf & 3) == 0;
    for (int64_t k = wave; k < K; k += nwaves) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        const float* Ak = A + k * (int64_t)mf * nf;
        float* dAk = MODE == 2 ? dA + k * (int64_t)mf * nf : nullptr;
        for (int p = tb; p < te; p += kEJ) {
            int e[kEJ], s[kEJ];
            bool ok[kEJ];
#pragma unroll
            for (int j = 0; j < kEJ; ++j) {
                ok[j] = p + j < te;
                e[j] = order[ok[j] ? p + j : tb];
                s[j] = src[e[j]];
            }
            for (int c0 = 0; c0 < nf; c0 += 64) {
                const int c = c0 + 4 * cg;
                // the edges' input vectors for this column strip
                f32x4 x[kEJ];
#pragma unroll
                for (int j = 0; j < kEJ; ++j) {
                    x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (MODE != 1 && ok[j]) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < nf) {
                                float v = h[(int64_t)s[j] * nf + c + u];
                                if (gate) v *= gate[(int64_t)e[j] * nf + c + u];
                                x[j][u] = v;
                            }
                    }
                }
                f32x4 dxa[kEJ];
#pragma unroll
                for (int j = 0; j < kEJ; ++j) dxa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int m0 = 0; m0 < mf; m0 += 4) {
                    const int m = m0 + rs;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    if (MODE != 2 && m < mf) {
                        const float* pa = Ak + (int64_t)m * nf + c;
                        if (vec && c + 3 < nf) a = *reinterpret_cast<const f32x4*>(pa);
                        else {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (c + u < nf) a[u] = pa[u];
                        }
                    }
                    if (MODE == 0) {
#pragma unroll
                        for (int j = 0; j < kEJ; ++j) {
                            float part = a.x * x[j].x + a.y * x[j].y + a.z * x[j].z + a.w * x[j].w;
                            part += __shfl_xor(part, 1);
                            part += __shfl_xor(part, 2);
                            part += __shfl_xor(part, 4);
                            part += __shfl_xor(part, 8);
                            if (cg == 0 && ok[j] && m < mf) {
                                float* o = out + (int64_t)e[j] * mf + m;
                                *o = c0 == 0 ? part : *o + part;           // column strips accumulate in order
                            }
                        }
                    } else if (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < kEJ; ++j) {
                            const float d = (ok[j] && m < mf) ? dmsg[(int64_t)e[j] * mf + m] : 0.f;
                            dxa[j] += a * d;
                        }
                    } else {
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < kEJ; ++j) {
                            const float d = (ok[j] && m < mf) ? dmsg[(int64_t)e[j] * mf + m] : 0.f;
                            g += x[j] * d;
                        }
                        if (m < mf) {
                            float* pd = dAk + (int64_t)m * nf + c;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (c + u < nf) pd[u] += g[u];
                        }
                    }
                }
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < kEJ; ++j) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float v = dxa[j][u];
                            v += __shfl_xor(v, 16);
                            v += __shfl_xor(v, 32);
                            if (rs == 0 && ok[j] && c + u < nf) out[(int64_t)e[j] * nf + c + u] = v;
                        }
                    }
                }
            }
        }
    }
}

// returns launch status; `mode` as above
int launch_edge_pertype(int mode, const float* h, const float* A, const int32_t* src, const int32_t* order,
                        const int32_t* type_ptr, const float* gate, const float* dmsg, float* out, float* dA, int K,
                        int nf, int mf, hipStream_t s) {
    int64_t blocks = ceil_div((int64_t)K, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(256);
    if (mode == 0)
        hipLaunchKernelGGL(edge_pertype_kernel<0>, grid, block, 0, s, h, A, src, order, type_ptr, gate, dmsg, out, dA, K,
                           nf, mf);
    else if (mode == 1)
        hipLaunchKernelGGL(edge_pertype_kernel<1>, grid, block, 0, s, h, A, src, order, type_ptr, gate, dmsg, out, dA, K,
                           nf, mf);
    else
        hipLaunchKernelGGL(edge_pertype_kernel<2>, grid, block, 0, s, h, A, src, order, type_ptr, gate, dmsg, out, dA, K,
                           nf, mf);
    return launch_status("mpnn_edge_message(per-type matvec)");
}

}  // namespace mpnn
