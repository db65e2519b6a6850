// Typed edge message on the fp32 matrix cores:  msg[e,:] = A[type(e)] . (gate[e] * h[src(e),:])
//
// Edges are visited in type-sorted order (`order`, `type_ptr`), so one 128-edge tile
// multiplies 128 gathered source rows by ONE (mf x nf) matrix: a true dense contraction,
// D[128 x mf] = X[128 x nf] . A_k^T, done with v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered
// fma chain).  Flops 2*nf*mf per edge; HBM traffic per edge = one gathered nf-row (mostly L2:
// a molecule's atoms are neighbours in memory) + one mf-row written to its destination-sorted
// slot e, so the aggregator that follows streams `msg` linearly.
//
// LDS images (both k-contiguous, row stride LD = KC+4 floats so ds_read_b128 is conflict-free):
//   Xs[r][k]  gathered rows, r = edge inside the tile
//   Bs[n][k]  A_k[n][k]      (A_k is stored (mf, nf) row-major == already "n-major, k-contiguous")
// Lane (r = lane&31, hi = lane>>5) of the MFMA supplies k = hi*KC/2 + s for step s, so a lane
// reads 4 consecutive k (one b128) per 4 MFMAs.
#include "common.h"

namespace mpnn {

constexpr int kTileEdges = 128;   // 4 waves x 32 rows
constexpr int kKC = 64;           // k-chunk staged per pass
constexpr int kLD = kKC + 4;      // padded LDS row stride (floats)

__device__ __forceinline__ f32x4 load_row4(const float* __restrict__ base, int64_t row, int ncols, int k, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* p = base + row * ncols + k;
    if (vec) {
        if (k < ncols) v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (k + 0 < ncols) v.x = p[0];
        if (k + 1 < ncols) v.y = p[1];
        if (k + 2 < ncols) v.z = p[2];
        if (k + 3 < ncols) v.w = p[3];
    }
    return v;
}

// block -> (type, first position inside `order`, rows) ; returns false for surplus blocks
__device__ __forceinline__ bool locate_tile(const int32_t* __restrict__ type_ptr, int K, int tile, int* type,
                                            int* start, int* rows) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
        const int b = type_ptr[k], e = type_ptr[k + 1];
        const int nt = (e - b + kTileEdges - 1) / kTileEdges;
        if (tile < acc + nt) {
            const int s = b + (tile - acc) * kTileEdges;
            *type = k;
            *start = s;
            *rows = min(kTileEdges, e - s);
            return true;
        }
        acc += nt;
    }
    return false;
}

template <int NB>   // NB = ceil(mf/32) column blocks per wave
__global__ void __launch_bounds__(256) edge_message_kernel(const float* __restrict__ h, const float* __restrict__ A,
                                                           const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ type_ptr,
                                                           const float* __restrict__ gate, float* __restrict__ msg,
                                                           int K, int nf, int mf) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);                 // [kTileEdges][kLD]
    float* Bs = Xs + kTileEdges * kLD;                              // [32*NB][kLD]
    int* s_eid = reinterpret_cast<int*>(Bs + 32 * NB * kLD);        // [kTileEdges]
    int* s_src = s_eid + kTileEdges;                                // [kTileEdges]

    int type, start, rows;
    if (!locate_tile(type_ptr, K, blockIdx.x, &type, &start, &rows)) return;   // uniform per block

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < kTileEdges) {
        int e = -1, s = 0;
        if (tid < rows) { e = order[start + tid]; s = src[e]; }
        s_eid[tid] = e;
        s_src[tid] = s;
    }
    __syncthreads();

    const bool vec = (nf & 3) == 0;
    const int c4 = tid & 15, r0 = tid >> 4;
    const float* Ak = A + (int64_t)type * mf * nf;

    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

    const int r = lane & 31, hi = lane >> 5;
    for (int kc0 = 0; kc0 < nf; kc0 += kKC) {
        if (kc0) __syncthreads();
        const int k = kc0 + 4 * c4;
        // gather the source rows (full 256-B lines per 16 lanes), optional feature gate
#pragma unroll
        for (int p = 0; p < kTileEdges / 16; ++p) {
            const int row = r0 + 16 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                v = load_row4(h, s_src[row], nf, k, vec);
                if (gate) v *= load_row4(gate, s_eid[row], nf, k, vec);
            }
            *reinterpret_cast<f32x4*>(Xs + row * kLD + 4 * c4) = v;
        }
#pragma unroll
        for (int p = 0; p < 2 * NB; ++p) {
            const int n = r0 + 16 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < mf) v = load_row4(Ak, n, nf, k, vec);
            *reinterpret_cast<f32x4*>(Bs + n * kLD + 4 * c4) = v;
        }
        __syncthreads();
        if (32 * wv < rows) {
            const float* xa = Xs + (32 * wv + r) * kLD + hi * (kKC / 2);
            const float* xb = Bs + r * kLD + hi * (kKC / 2);
#pragma unroll
            for (int kq = 0; kq < kKC / 8; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xa + 4 * kq);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(xb + 32 * n * kLD + 4 * kq);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[n], 0, 0, 0);
                }
            }
        }
    }
    // each accumulator register is one 128-B row segment per half-wave: dword stores, coalesced
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = 32 * n + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * wv + acc_row(i, lane);
            if (row < rows && col < mf) msg[(int64_t)s_eid[row] * mf + col] = acc[n][i];
        }
    }
}

static size_t message_lds_bytes(int nb) {
    return (size_t)(kTileEdges * kLD + 32 * nb * kLD) * sizeof(float) + 2 * kTileEdges * sizeof(int);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_edge_message_f32(const float* h, const float* A, const int32_t* src, const int32_t* order,
                                     const int32_t* type_ptr, const float* gate, float* msg, int64_t V, int64_t E,
                                     int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(E >= 0 && V >= 0 && K >= 0, "mpnn_edge_message_f32: negative size");
    MPNN_REQUIRE(nf > 0 && nf <= MPNN_MAX_FEATURES && mf > 0 && mf <= 256,
                 "mpnn_edge_message_f32: nf=%d mf=%d unsupported (nf<=%d, mf<=256)", nf, mf, MPNN_MAX_FEATURES);
    MPNN_REQUIRE(E < (1ll << 31) && V < (1ll << 31), "mpnn_edge_message_f32: int32 index overflow");
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && src && order && type_ptr && msg && K > 0, "mpnn_edge_message_f32: NULL buffer");
    MPNN_REQUIRE(K <= 4096, "mpnn_edge_message_f32: K=%d edge types; use the per-edge-matrix path above 4096", K);
    if ((nf & 3) == 0) {
        const uintptr_t al = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(A) |
                             reinterpret_cast<uintptr_t>(gate);
        MPNN_REQUIRE(al % 16 == 0, "mpnn_edge_message_f32: buffers must be 16-byte aligned");
    }
    const int nb = (mf + 31) / 32;
    const int64_t tiles = ceil_div(E, kTileEdges) + K;   // upper bound; surplus blocks exit at once
    const dim3 grid((unsigned)tiles), block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = message_lds_bytes(nb);
#define MPNN_MSG_CASE(NB)                                                                                            \
    case NB:                                                                                                         \
        if (lds > 48 * 1024)                                                                                         \
            (void)hipFuncSetAttribute((const void*)edge_message_kernel<NB>,                                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((edge_message_kernel<NB>), grid, block, lds, s, h, A, src, order, type_ptr, gate, msg, K, \
                           nf, mf);                                                                                  \
        break;
    switch (nb) {
        MPNN_MSG_CASE(1)
        MPNN_MSG_CASE(2)
        MPNN_MSG_CASE(3)
        MPNN_MSG_CASE(4)
        MPNN_MSG_CASE(5)
        MPNN_MSG_CASE(6)
        MPNN_MSG_CASE(7)
        MPNN_MSG_CASE(8)
    }
#undef MPNN_MSG_CASE
    return launch_status("mpnn_edge_message_f32");
}
