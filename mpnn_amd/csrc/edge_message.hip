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
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mpnn {

constexpr int kManyTypes = 4096;  // above this many matrices a "type" is a handful of edges: per-type matvec path
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

// ------------------------------------------------------------------------------------------------
// Fast path (nf in {32, 64, 128}): matrix resident in LDS per type, no barrier inside a type.
//
// A block walks the K types in order.  For type k it stages A_k once (LDS image [n][k], stride
// nf+4), then each of its 8 waves takes 32-edge tiles of that type on its own: the MFMA A-fragment
// of lane (r,hi) is the contiguous half-row h[src(e_r)][hi*nf/2 ...] gathered straight from global
// into registers (full 128-B segments, mostly L2 hits: a molecule's atoms are neighbouring rows),
// the next tile's rows and the tile-after-next's indices are requested before the current tile is
// multiplied.  Stores go to the destination-sorted slot e (one 128-B row segment per half-wave).
//
// BWD = false: msg[e] = A_k . (gate[e] * h[src(e)])       rows gathered by source atom, image = A_k
// BWD = true : dx[e]  = A_k^T . dmsg[e]                   rows indexed by edge id,     image = A_k^T
//              (`h` is dmsg, `mf` the OUTPUT width nf_x, A_k is (NF, mf) read transposed)
template <int NF, int NB, bool BWD, bool GATED>
__global__ void __launch_bounds__(256, GATED ? 4 : 5) edge_message_resident_kernel(
    const float* __restrict__ h, const float* __restrict__ A, const int32_t* __restrict__ src,
    const int32_t* __restrict__ order, const int32_t* __restrict__ type_ptr, const float* __restrict__ gate,
    float* __restrict__ msg, int K, int mf) {
    constexpr int LD = NF + 4;
    constexpr int NF4 = NF / 8;                         // float4 fragments per lane
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);     // [32*NB][LD]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;

    for (int k = 0; k < K; ++k) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        if (te == tb) continue;                         // uniform over the grid
        __syncthreads();                                // everyone is done with the previous matrix
        const float* Ak = A + (int64_t)k * mf * NF;
        if (!BWD) {
            for (int idx = tid; idx < 32 * NB * (NF / 4); idx += 256) {
                const int n = idx / (NF / 4), q = idx % (NF / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < mf) v = *reinterpret_cast<const f32x4*>(Ak + (int64_t)n * NF + 4 * q);
                *reinterpret_cast<f32x4*>(As + n * LD + 4 * q) = v;
            }
        } else {
            // A_k is (NF rows a) x (mf cols b); image As[b][a] = A_k[a][b]  (once per type per block)
            for (int idx = tid; idx < NF * 32 * NB; idx += 256) {
                const int a = idx / (32 * NB), b = idx % (32 * NB);
                As[b * LD + a] = (b < mf) ? Ak[(int64_t)a * mf + b] : 0.f;
            }
        }
        __syncthreads();

        const int tiles = (te - tb + 31) / 32;
        int t = gw;
        if (t >= tiles) continue;
        // lane r owns edge slot tb + 32*t + r of the type-sorted list (slots past the end re-use the
        // tile's first edge for loads and are skipped at the store)
        auto edge_of = [&](int tile) {
            const int pos = tb + 32 * tile + r;
            return order[pos < te ? pos : tb + 32 * tile];
        };
        auto load_rows = [&](int s_row, int e_row, f32x4 (&f)[NF4]) {
            const float* p = h + (int64_t)s_row * NF + hi * (NF / 2);
#pragma unroll
            for (int q = 0; q < NF4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
            if (GATED) {
                const float* g = gate + (int64_t)e_row * NF + hi * (NF / 2);
#pragma unroll
                for (int q = 0; q < NF4; ++q) f[q] *= *reinterpret_cast<const f32x4*>(g + 4 * q);
            }
        };
        auto src_of = [&](int e) { return BWD ? e : src[e]; };
        int e_cur = edge_of(t);
        int s_cur = src_of(e_cur);
        f32x4 f_cur[NF4];
        int e_nxt = e_cur, s_nxt = s_cur;
        if (t + nw < tiles) { e_nxt = edge_of(t + nw); s_nxt = src_of(e_nxt); }

        // No register prefetch of the next tile's rows: at <= 80 VGPRs six waves share a SIMD and hide each other's
        // gather latency; only the two-deep index chain (order -> src) is fetched ahead.
        for (; t < tiles; t += nw) {
            const bool has2 = t + 2 * nw < tiles;
            load_rows(s_cur, e_cur, f_cur);
            int e_nn = e_nxt;
            if (has2) e_nn = edge_of(t + 2 * nw);

            f32x16 acc[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
            const float* xb = As + r * LD + hi * (NF / 2);
            // B fragments double-buffered by hand; the scheduling barriers keep the compiler from hoisting all
            // NF4*NB LDS reads above the MFMAs (which costs ~100 registers and an occupancy step)
#pragma unroll
            for (int q = 0; q < NF4; ++q) {
                f32x4 bq[NB];
#pragma unroll
                for (int n = 0; n < NB; ++n) bq[n] = *reinterpret_cast<const f32x4*>(xb + 32 * n * LD + 4 * q);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_cur[q].x, bq[n].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_cur[q].y, bq[n].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_cur[q].z, bq[n].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_cur[q].w, bq[n].w, acc[n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            int s_nn = s_nxt;
            if (has2) s_nn = src_of(e_nn);                       // index of the tile after next

            const int rows = min(32, te - tb - 32 * t);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = acc_row(i, lane);                // = const(i) + 4*hi
                const int e_row = __shfl(e_cur, row);            // lane `row` holds that edge's id
                if (row < rows) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const int col = 32 * n + r;
                        if (col < mf) __builtin_nontemporal_store(acc[n][i], msg + (int64_t)e_row * mf + col);   // write-once stream
                    }
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            e_cur = e_nxt;
            s_cur = s_nxt;
            e_nxt = e_nn;
            s_nxt = s_nn;
        }
    }
}

template <int NF, int NB, bool BWD>
static int launch_message_resident(const float* h, const float* A, const int32_t* src, const int32_t* order,
                                   const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, int mf,
                                   hipStream_t s) {
    const size_t lds = (size_t)32 * NB * (NF + 4) * sizeof(float);
    int64_t blocks = gate ? 1024 : 1280;                    // 4-wave blocks: 5 per CU ungated (<= 96 VGPRs), 4 gated
    const int64_t need = ceil_div(ceil_div(E, 32) + K, 4);
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (gate)
        hipLaunchKernelGGL((edge_message_resident_kernel<NF, NB, BWD, true>), dim3((unsigned)blocks), dim3(256), lds, s, h,
                           A, src, order, type_ptr, gate, msg, K, mf);
    else
        hipLaunchKernelGGL((edge_message_resident_kernel<NF, NB, BWD, false>), dim3((unsigned)blocks), dim3(256), lds, s,
                           h, A, src, order, type_ptr, gate, msg, K, mf);
    return launch_status("mpnn_edge_message_f32(resident)");
}

static size_t message_lds_bytes(int nb) {
    return (size_t)(kTileEdges * kLD + 32 * nb * kLD) * sizeof(float) + 2 * kTileEdges * sizeof(int);
}

// dx[e] = A_type(e)^T dmsg[e] on the resident-matrix kernel; returns 1 when the shape has no fast path
// many matrices, few edges each (edge_pertype.hip)
int launch_edge_pertype(int mode, const float* h, const float* A, const int32_t* src, const int32_t* order,
                        const int32_t* type_ptr, const float* gate, const float* dmsg, float* out, float* dA, int K,
                        int nf, int mf, hipStream_t s);
// nf = mf = 128 on the bf16x6 pipe (edge_message128.hip)
int launch_message_split128(const float* h, const float* A, const int32_t* src, const int32_t* order,
                            const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s);
int launch_message_dx_split128(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                               float* dx, int64_t E, int K, hipStream_t s);
int launch_message_stream256(const float* h, const float* A, const int32_t* src, const int32_t* order,
                             const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s);
int launch_message_split64(const float* h, const float* A, const int32_t* src, const int32_t* order,
                           const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s);
int launch_message_dx_split64(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                              float* dx, int64_t E, int K, hipStream_t s);

static bool fp32_only() {
    const bool v = switches().math_fp32;
    return v;
}

int launch_message_dx_resident(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                               float* dx, int64_t E, int K, int nf, int mf, hipStream_t s) {
    if (K > 64) return 1;
    if (mf == 128 && nf == 128 && !fp32_only()) return launch_message_dx_split128(dmsg, A, order, type_ptr, dx, E, K, s);
    if (mf == 64 && nf == 64 && !fp32_only()) return launch_message_dx_split64(dmsg, A, order, type_ptr, dx, E, K, s);
    if (mf == 64 && nf == 64)
        return launch_message_resident<64, 2, true>(dmsg, A, nullptr, order, type_ptr, nullptr, dx, E, K, nf, s);
    if (mf == 32 && nf == 32)
        return launch_message_resident<32, 1, true>(dmsg, A, nullptr, order, type_ptr, nullptr, dx, E, K, nf, s);
    return 1;
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
    if (K > kManyTypes)   // continuous bond features: one or two edges per matrix, HBM-bound matvec (edge_pertype.hip)
        return launch_edge_pertype(0, h, A, src, order, type_ptr, gate, nullptr, msg, nullptr, K, nf, mf,
                                   (hipStream_t)stream);
    if ((nf & 3) == 0) {
        const uintptr_t al = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(A) |
                             reinterpret_cast<uintptr_t>(gate);
        MPNN_REQUIRE(al % 16 == 0, "mpnn_edge_message_f32: buffers must be 16-byte aligned");
    }
    const int nb = (mf + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (K <= 64) {      // resident-matrix fast path: the type loop is sequential, keep K small
        if (nf == 64 && mf == 64 && !fp32_only()) return launch_message_split64(h, A, src, order, type_ptr, gate, msg, E, K, st);
        if (nf == 64 && nb == 2) return launch_message_resident<64, 2, false>(h, A, src, order, type_ptr, gate, msg, E, K, mf, st);
        if (nf == 64 && nb == 1) return launch_message_resident<64, 1, false>(h, A, src, order, type_ptr, gate, msg, E, K, mf, st);
        if (nf == 32 && nb == 1) return launch_message_resident<32, 1, false>(h, A, src, order, type_ptr, gate, msg, E, K, mf, st);
        if (nf == 128 && mf == 128 && !fp32_only()) return launch_message_split128(h, A, src, order, type_ptr, gate, msg, E, K, st);
        if (nf == 256 && mf == 256 && !fp32_only()) return launch_message_stream256(h, A, src, order, type_ptr, gate, msg, E, K, st);
    }
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
