// Backward kernels of the edge message and the GRU update (fp32 MFMA).
//
// Two GEMM shapes cover every gradient:
//   rows_gemm      out[r,:] = X[r,:] . B          one 128-row tile per block, K in chunks of 64
//                  - dx[e]  = A_type(e)^T dmsg[e]     (typed tiles, B = A_k stored (mf,nf) = [k][n])
//                  - dm     = dgi W_ih^T, dh = dgh W_hh^T + z*dout   (B = W stored (H,3H) = [n][k])
//   tn_accumulate  C += X^T . Y  (reduction over rows; persistent blocks, one atomic flush each)
//                  - dA[k]  = sum_{e of type k} dmsg[e] (x) (gate[e]*h[src e])
//                  - dW_ih  = m^T dgi, dW_hh = h^T dgh, db = column sums of dgi / dgh
// plus the elementwise gate-gradient kernel of the GRU.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mpnn {

constexpr int kBT = 128;          // rows per tile (4 waves x 32)
constexpr int kBK = 64;           // contraction chunk
constexpr int kLDA = kBK + 4;     // row stride of k-contiguous LDS images

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p, int have, bool vec) {
    // 4 consecutive floats at p, the first `have` of them valid (have <= 0 -> zeros)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        if (have >= 4) v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (have > 0) v.x = p[0];
        if (have > 1) v.y = p[1];
        if (have > 2) v.z = p[2];
        if (have > 3) v.w = p[3];
    }
    return v;
}

__device__ __forceinline__ int count_tiles(const int32_t* __restrict__ type_ptr, int K) {
    int acc = 0;
    for (int k = 0; k < K; ++k) acc += (type_ptr[k + 1] - type_ptr[k] + kBT - 1) / kBT;
    return acc;
}

__device__ __forceinline__ bool find_tile(const int32_t* __restrict__ type_ptr, int K, int tile, int* type, int* start,
                                          int* rows) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
        const int b = type_ptr[k], e = type_ptr[k + 1];
        const int nt = (e - b + kBT - 1) / kBT;
        if (tile < acc + nt) {
            const int s = b + (tile - acc) * kBT;
            *type = k;
            *start = s;
            *rows = min(kBT, e - s);
            return true;
        }
        acc += nt;
    }
    return false;
}

// ------------------------------------------------------------------------------------ rows_gemm
// out[row, n] = sum_k X[row, k] * Bop[k, n] (+ add[row, n]),  n < N, k < Kdim
//   B_IS_NK: B stored [n][k] (ldb = row stride over n) else [k][n]
//   TYPED:   tile rows are edge ids order[start + r], B = Bbase + type * b_type_stride
template <int NB, bool B_IS_NK, bool TYPED>
__global__ void __launch_bounds__(256) rows_gemm_kernel(const float* __restrict__ X, int ldx,
                                                        const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ type_ptr, int K_types,
                                                        const float* __restrict__ Bbase, int64_t b_type_stride, int ldb,
                                                        const float* add, float* out, int ldo, int64_t R, int Kdim,
                                                        int N) {
    constexpr int LDB = 32 * NB + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);      // [kBT][kLDA]
    float* Bs = Xs + kBT * kLDA;                         // [kBK][LDB]
    int* s_row = reinterpret_cast<int*>(Bs + kBK * LDB); // [kBT]

    int type = 0, start, rows;
    if (TYPED) {
        if (!find_tile(type_ptr, K_types, blockIdx.x, &type, &start, &rows)) return;
    } else {
        const int64_t s = (int64_t)blockIdx.x * kBT;
        if (s >= R) return;
        start = (int)s;
        rows = (int)min((int64_t)kBT, R - s);
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < kBT) s_row[tid] = (tid < rows) ? (TYPED ? order[start + tid] : start + tid) : -1;
    __syncthreads();

    const float* B = Bbase + (int64_t)type * b_type_stride;
    const bool vecx = ((ldx & 3) == 0) && ((Kdim & 3) == 0);
    const bool vecb = ((ldb & 3) == 0) && (((B_IS_NK ? Kdim : N) & 3) == 0);
    const int c4 = tid & 15, r0 = tid >> 4;
    const int r = lane & 31, hi = lane >> 5;

    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

    for (int k0 = 0; k0 < Kdim; k0 += kBK) {
        if (k0) __syncthreads();
        const int k = k0 + 4 * c4;
#pragma unroll
        for (int p = 0; p < kBT / 16; ++p) {
            const int row = r0 + 16 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) v = ld4(X + (int64_t)s_row[row] * ldx + k, Kdim - k, vecx);
            *reinterpret_cast<f32x4*>(Xs + row * kLDA + 4 * c4) = v;
        }
        if (B_IS_NK) {
            // global [n][k] -> LDS [k][n]: read 4 consecutive k of one n, write them down a column
#pragma unroll
            for (int p = 0; p < 2 * NB; ++p) {
                const int n = r0 + 16 * p;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < N) v = ld4(B + (int64_t)n * ldb + k, Kdim - k, vecb);
                Bs[(4 * c4 + 0) * LDB + n] = v.x;
                Bs[(4 * c4 + 1) * LDB + n] = v.y;
                Bs[(4 * c4 + 2) * LDB + n] = v.z;
                Bs[(4 * c4 + 3) * LDB + n] = v.w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 2 * NB; ++p) {
                const int idx = tid + 256 * p;
                const int kk = idx / (8 * NB), q = idx % (8 * NB);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k0 + kk < Kdim) v = ld4(B + (int64_t)(k0 + kk) * ldb + 4 * q, N - 4 * q, vecb);
                *reinterpret_cast<f32x4*>(Bs + kk * LDB + 4 * q) = v;
            }
        }
        __syncthreads();
        if (32 * wv < rows) {
            const float* xa = Xs + (32 * wv + r) * kLDA + hi * (kBK / 2);
            const float* xb = Bs + hi * (kBK / 2) * LDB + r;
#pragma unroll
            for (int kq = 0; kq < kBK / 8; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xa + 4 * kq);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* bp = xb + (4 * kq + c) * LDB;
                    const float av = a[c];
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32 * n], acc[n], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = 32 * n + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * wv + acc_row(i, lane);
            if (row < rows && col < N) {
                const int64_t o = (int64_t)s_row[row] * ldo + col;
                out[o] = add ? acc[n][i] + add[o] : acc[n][i];
            }
        }
    }
}

static size_t rows_gemm_lds(int nb) { return (size_t)(kBT * kLDA + kBK * (32 * nb + 4)) * 4 + kBT * 4; }

template <bool B_IS_NK, bool TYPED>
static int launch_rows_gemm(const float* X, int ldx, const int32_t* order, const int32_t* type_ptr, int K_types,
                            const float* B, int64_t b_type_stride, int ldb, const float* add, float* out, int ldo,
                            int64_t R, int Kdim, int N, hipStream_t s, const char* what) {
    const int nb = (N + 31) / 32;
    const int64_t tiles = TYPED ? ceil_div(R, kBT) + K_types : ceil_div(R, kBT);
    const dim3 grid((unsigned)tiles), block(256);
    const size_t lds = rows_gemm_lds(nb);
#define MPNN_RG_CASE(NB)                                                                                               \
    case NB:                                                                                                           \
        if (lds > 48 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)rows_gemm_kernel<NB, B_IS_NK, TYPED>,                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((rows_gemm_kernel<NB, B_IS_NK, TYPED>), grid, block, lds, s, X, ldx, order, type_ptr,       \
                           K_types, B, b_type_stride, ldb, add, out, ldo, R, Kdim, N);                                 \
        break;
    switch (nb) {
        MPNN_RG_CASE(1)
        MPNN_RG_CASE(2)
        MPNN_RG_CASE(3)
        MPNN_RG_CASE(4)
        MPNN_RG_CASE(5)
        MPNN_RG_CASE(6)
        MPNN_RG_CASE(7)
        MPNN_RG_CASE(8)
        default:
            set_error("%s: N=%d wider than 256 columns", what, N);
            return MPNN_EINVAL;
    }
#undef MPNN_RG_CASE
    return launch_status(what);
}

// ------------------------------------------------------------------------------------ tn_accumulate
// C[type][a, b] += sum_rows X[xrow, a] * (gate[xrow, b] *) Y[yrow, b]      a < M, b < N
//   TYPED: rows of a tile are edges e = order[start+r]; xrow = e, yrow = src[e], gate row = e
//   else : xrow = yrow = start + r
// grid.y enumerates 64x64 blocks of C; grid.x blocks each own a contiguous range of row tiles and
// keep their partial C in registers until the type changes or the range ends (one atomic flush).
template <bool TYPED>
__global__ void __launch_bounds__(256) tn_accumulate_kernel(const float* __restrict__ X, int ldx, int M,
                                                            const float* __restrict__ Y, int ldy, int N,
                                                            const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ src,
                                                            const float* __restrict__ gate,
                                                            const int32_t* __restrict__ type_ptr, int K_types,
                                                            float* C, float* colsum, int64_t R) {
    __shared__ __attribute__((aligned(16))) float Xs[kBT * kLDA];
    __shared__ __attribute__((aligned(16))) float Ys[kBT * kLDA];
    __shared__ int s_x[kBT];
    __shared__ int s_y[kBT];

    const int nbc = (N + 63) / 64;
    const int a0 = (blockIdx.y / nbc) * 64, b0 = (blockIdx.y % nbc) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wa = wv >> 1, wb = wv & 1;
    const int c4 = tid & 15, r0 = tid >> 4;
    const int i = lane & 31, hi = lane >> 5;
    const bool vecx = ((ldx & 3) == 0) && ((M & 3) == 0);
    const bool vecy = ((ldy & 3) == 0) && ((N & 3) == 0);

    const int total = TYPED ? count_tiles(type_ptr, K_types) : (int)ceil_div(R, kBT);
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int t_lo = blockIdx.x * per, t_hi = min(total, t_lo + per);

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float cs = 0.f;
    int cur_type = -1;

    auto flush = [&](int type) {
        float* Ct = C + (int64_t)type * M * N;
        const int col = b0 + 32 * wb + i;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = a0 + 32 * wa + acc_row(q, lane);
            if (row < M && col < N) atomicAdd(Ct + (int64_t)row * N + col, acc[q]);
            acc[q] = 0.f;
        }
    };

    for (int t = t_lo; t < t_hi; ++t) {
        int type = 0, start, rows;
        if (TYPED) {
            find_tile(type_ptr, K_types, t, &type, &start, &rows);
        } else {
            start = t * kBT;
            rows = (int)min((int64_t)kBT, R - (int64_t)start);
        }
        if (type != cur_type) {
            if (cur_type >= 0) flush(cur_type);
            cur_type = type;
        }
        __syncthreads();   // previous tile's LDS reads are done
        if (tid < kBT) {
            int xr = -1, yr = 0;
            if (tid < rows) {
                xr = TYPED ? order[start + tid] : start + tid;
                yr = TYPED ? src[xr] : xr;
            }
            s_x[tid] = xr;
            s_y[tid] = yr;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kBT / 16; ++p) {
            const int row = r0 + 16 * p;
            f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vy = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                const int ka = a0 + 4 * c4, kb = b0 + 4 * c4;
                vx = ld4(X + (int64_t)s_x[row] * ldx + ka, M - ka, vecx);
                vy = ld4(Y + (int64_t)s_y[row] * ldy + kb, N - kb, vecy);
                if (gate) vy *= ld4(gate + (int64_t)s_x[row] * N + kb, N - kb, vecy);
            }
            *reinterpret_cast<f32x4*>(Xs + row * kLDA + 4 * c4) = vx;
            *reinterpret_cast<f32x4*>(Ys + row * kLDA + 4 * c4) = vy;
        }
        __syncthreads();
        const float* xa = Xs + hi * (kBT / 2) * kLDA + 32 * wa + i;
        const float* yb = Ys + hi * (kBT / 2) * kLDA + 32 * wb + i;
#pragma unroll 16
        for (int s = 0; s < kBT / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s * kLDA], yb[s * kLDA], acc, 0, 0, 0);
        if (colsum && a0 == 0 && tid < 64) {
            float part = 0.f;
            for (int rr = 0; rr < kBT; ++rr) part += Ys[rr * kLDA + tid];
            cs += part;
        }
    }
    if (cur_type >= 0) flush(cur_type);
    if (colsum && a0 == 0 && tid < 64 && b0 + tid < N && t_hi > t_lo) atomicAdd(colsum + b0 + tid, cs);
}

// ------------------------------------------------------------------------------------ dA, register-direct
// dA[k][a][b] += sum_{e of type k} dmsg[e][a] * (gate[e][b] *) h[src e][b]        (mf = nf = 64)
//
// For a transposed product the MFMA operand layouts ARE coalesced global loads: step s of a 32-edge
// tile needs A[i=a][k=row] = dmsg[row][a] and B[k=row][j=b] = x[row][b] with the feature index on the
// lane -- one dword load per lane, 128 contiguous bytes per half-wave -- so nothing is staged in LDS
// and there is no barrier per tile.  One wave owns the whole 64x64 output (four 32x32 accumulators)
// and therefore reads every dmsg / x element exactly once.  Waves take 32-edge tiles of the
// type-sorted list on their own; per type the block's 8 partial tiles are combined in LDS (ds_add)
// and leave through one float-atomic flush per block.
// With `dst` given the left operand is NOT a materialised dmsg: row e is  w[e] * dagg[dst[e]]  (the
// aggregator's backward folded into this kernel, saving the E x F write + read).
template <bool HAS_DST, bool HAS_W, bool GATED>
__global__ void __launch_bounds__(512, 4) edge_dA_direct64_kernel(const float* __restrict__ dmsg,
                                                                  const float* __restrict__ h,
                                                                  const int32_t* __restrict__ src,
                                                                  const int32_t* __restrict__ order,
                                                                  const int32_t* __restrict__ type_ptr,
                                                                  const float* __restrict__ gate, float* dA, int K,
                                                                  const int32_t* __restrict__ dst,
                                                                  const float* __restrict__ w) {
    constexpr int F = 64;
    __shared__ float red[F * F];                        // block-level partial of the current type
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;
    const int gw = blockIdx.x * 8 + wv, nw = gridDim.x * 8;

    for (int k = 0; k < K; ++k) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        if (te == tb) continue;                         // uniform over the grid
        for (int idx = tid; idx < F * F; idx += 512) red[idx] = 0.f;
        __syncthreads();

        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[q][z] = 0.f;
        const int tiles = (te - tb + 31) / 32;
        bool any = false;
        for (int t = gw; t < tiles; t += nw) {
            any = true;
            const int pos = tb + 32 * t + i;            // both lane halves hold the tile's 32 edge ids
            const bool ok = pos < te;
            const int e_l = order[ok ? pos : tb + 32 * t];
            const int s_l = src[e_l];
            const int d_l = HAS_DST ? dst[e_l] : e_l;   // row of the left operand
            const float w_l = HAS_W ? w[e_l] : 1.0f;
            const int rows = min(32, te - tb - 32 * t);
            // Two batches of 8 steps.  A batch issues ALL its loads before anything looks at a loaded value (the
            // flags are template parameters and the ragged-tile mask is applied afterwards): with run-time branches
            // around the gate / weight factors the compiler waited for memory once per step, 16 round trips a tile.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float a0[8], a1[8], b0[8], b1[8], g0[8], g1[8], wr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = 8 * half + u;         // this step multiplies rows s (lanes 0-31) and 16+s (32-63)
                    const int s_lo = __builtin_amdgcn_readlane(s_l, s), s_hi = __builtin_amdgcn_readlane(s_l, 16 + s);
                    const int d_lo = __builtin_amdgcn_readlane(d_l, s), d_hi = __builtin_amdgcn_readlane(d_l, 16 + s);
                    const float* pa = dmsg + (int64_t)(hi ? d_hi : d_lo) * F + i;
                    const float* pb = h + (int64_t)(hi ? s_hi : s_lo) * F + i;
                    a0[u] = pa[0];
                    a1[u] = pa[32];
                    b0[u] = pb[0];
                    b1[u] = pb[32];
                    if (GATED) {
                        const int e_lo = __builtin_amdgcn_readlane(e_l, s), e_hi = __builtin_amdgcn_readlane(e_l, 16 + s);
                        const float* pg = gate + (int64_t)(hi ? e_hi : e_lo) * F + i;
                        g0[u] = pg[0];
                        g1[u] = pg[32];
                    }
                    if (HAS_W) {
                        const float w_lo = readlane_f(w_l, s), w_hi = readlane_f(w_l, 16 + s);
                        wr[u] = hi ? w_hi : w_lo;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = 8 * half + u;
                    const bool live = (hi ? 16 + s : s) < rows;
                    float x0 = a0[u], x1 = a1[u], y0 = b0[u], y1 = b1[u];
                    if (HAS_W) { x0 *= wr[u]; x1 *= wr[u]; }
                    if (GATED) { y0 *= g0[u]; y1 *= g1[u]; }
                    if (!live) { x0 = 0.f; x1 = 0.f; }
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
                }
            }
        }
        if (any) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * (q & 1) + i;
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const int row = 32 * (q >> 1) + acc_row(z, lane);
                    atomicAdd(&red[row * F + col], acc[q][z]);          // LDS float add
                }
            }
        }
        __syncthreads();
        float* out = dA + (int64_t)k * F * F;
        for (int idx = tid; idx < F * F; idx += 512) {
            const float v = red[idx];
            if (v != 0.f) atomicAdd(out + idx, v);
        }
        __syncthreads();                                // red is re-zeroed for the next type
    }
}

// ------------------------------------------------------------------------------------ tn, register-direct, wide
// C[type] (32*MA x 32*NBT) += sum_rows A[arow, :]^T (x) B[brow, :] for outputs too wide for one wave.
// The block's 8 waves walk the SAME 32-row tiles (their operand loads hit the same L1 lines) and split the
// output: wave w owns a-block (w % MA) and the NBW b-blocks {bg + G*j}, bg = w / MA, G = 8 / MA, so the
// block covers all MA x (G*NBW) tiles and no two waves add to the same element (no LDS reduction).
// Operands are coalesced dword loads straight from global, as in edge_dA_direct64_kernel.
//   TYPED : rows are edges of the type-sorted list; arow = dst[e] (or e), brow = src[e], optional w / gate;
//           C is flushed per type.        untyped: arow = brow = row; optional column sums of B (bias grads).
template <int MA, int NBW, bool TYPED>
__global__ void __launch_bounds__(512) tn_direct_kernel(const float* __restrict__ Xa, int lda,
                                                        const float* __restrict__ Yb, int ldb,
                                                        const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ type_ptr, int K,
                                                        const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                        const float* __restrict__ w, const float* __restrict__ gate,
                                                        float* C, float* colsum, int64_t R) {
    constexpr int G = 8 / MA;
    constexpr int NBT = G * NBW;
    constexpr int M = 32 * MA, N = 32 * NBT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;
    const int ai = wv % MA, bg = wv / MA;

    f32x16 acc[NBW];
    float cs[NBW];
    const int ntypes = TYPED ? K : 1;
    for (int k = 0; k < ntypes; ++k) {
        int64_t tb = 0, te = R;
        if (TYPED) {
            tb = type_ptr[k];
            te = type_ptr[k + 1];
            if (te == tb) continue;
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            cs[j] = 0.f;
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[j][z] = 0.f;
        }
        const int64_t tiles = (te - tb + 31) / 32;
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t pos = tb + 32 * t + i;
            const bool ok = pos < te;
            const int64_t p0 = ok ? pos : tb + 32 * t;
            int a_l, b_l, e_l = 0;
            float w_l = 1.0f;
            if (TYPED) {
                e_l = order[p0];
                a_l = dst ? dst[e_l] : e_l;
                b_l = src[e_l];
                if (dst && w) w_l = w[e_l];
            } else {
                a_l = b_l = (int)p0;
            }
            const int rows = (int)min((int64_t)32, te - tb - 32 * t);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int a_r = hi ? __builtin_amdgcn_readlane(a_l, 16 + s) : __builtin_amdgcn_readlane(a_l, s);
                const int b_r = hi ? __builtin_amdgcn_readlane(b_l, 16 + s) : __builtin_amdgcn_readlane(b_l, s);
                const bool live = (hi ? 16 + s : s) < rows;
                float av = Xa[(int64_t)a_r * lda + 32 * ai + i];
                if (TYPED && dst && w) av *= hi ? readlane_f(w_l, 16 + s) : readlane_f(w_l, s);
                if (!live) av = 0.f;
                const float* pb = Yb + (int64_t)b_r * ldb + 32 * bg + i;
                const float* pg = nullptr;
                if (TYPED && gate) {
                    const int e_r = hi ? __builtin_amdgcn_readlane(e_l, 16 + s) : __builtin_amdgcn_readlane(e_l, s);
                    pg = gate + (int64_t)e_r * N + 32 * bg + i;
                }
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    float bv = pb[32 * G * j];
                    if (TYPED && gate) bv *= pg[32 * G * j];
                    if (!TYPED && live) cs[j] += bv;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
        float* Ct = C + (int64_t)k * M * N;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int col = 32 * (bg + G * j) + i;
#pragma unroll
            for (int z = 0; z < 16; ++z) {
                const float v = acc[j][z];
                if (v != 0.f) atomicAdd(Ct + (int64_t)(32 * ai + acc_row(z, lane)) * N + col, v);
            }
            if (!TYPED && colsum && ai == 0) {
                const float tot = cs[j] + __shfl_xor(cs[j], 32);
                if (hi == 0) atomicAdd(colsum + col, tot);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ GRU gate gradients
// From dout and the saved forward gates (r, z, n, gh_n) to the pre-activation gradients
//   ws[row] = [ dgi_r dgi_z dgi_n | dgh_r dgh_z dgh_n ]   (6H floats)   and   dh_direct = dout*mask*z.
// mask is 0/1 (as the reference's create_mask makes it): d sigma / d tanh use the masked gate values.
// COMPACT: ws[row] = [ dar daz dan dnh ] (4H floats; dgi = first 3 blocks, dgh = blocks 0, 1, 3) for the H = 128
// kernels of gru_bwd128.hip, which index the blocks themselves.
// NORM: dout is the gradient of norm(out); the gradient of out = this update's raw output y is formed here as
// dout * k1[col] + y * k2[col] + k4[col] with y = (1-z) n + z h of the saved gates (gru_bwd128_f16.hip has the algebra).
template <bool COMPACT, bool NORM = false>
__global__ void __launch_bounds__(256) gru_gate_grad_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                                            const float* __restrict__ mask,
                                                            const float* __restrict__ saved, float* __restrict__ ws,
                                                            float* __restrict__ dh, int64_t V, int H,
                                                            const float* __restrict__ kn = nullptr) {
    const int64_t total = V * H;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / H;
        const int col = (int)(idx - row * H);
        const float mk = mask ? mask[row] : 1.0f;
        const float* sv = saved + row * 4 * H + col;
        const float r = sv[0], z = sv[H], n = sv[2 * H], nh = sv[3 * H];
        float g = dout[idx] * mk;                       // through the final "* mask"
        if (NORM) {
            const float y = ((1.0f - z) * n + z * h[idx]) * mk;
            g = (dout[idx] * kn[col] + y * kn[H + col] + kn[2 * H + col]) * mk;
        }
        const float dn = g * (1.0f - z);
        const float dz = g * (h[idx] - n);
        const float dan = dn * mk * (1.0f - n * n);     // n = tanh(.)*mask
        const float dar = dan * nh * mk * r * (1.0f - r);
        const float daz = dz * mk * z * (1.0f - z);
        if (COMPACT) {
            float* w = ws + row * 4 * H + col;
            w[0] = dar;
            w[H] = daz;
            w[2 * H] = dan;
            w[3 * H] = dan * r;
        } else {
            float* w = ws + row * 6 * H + col;
            w[0] = dar;
            w[H] = daz;
            w[2 * H] = dan;
            w[3 * H] = dar;
            w[4 * H] = daz;
            w[5 * H] = dan * r;
        }
        dh[idx] = g * z;
    }
}

// resident-matrix dx path (edge_message.hip); returns 1 when the shape is not covered
int launch_message_dx_resident(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                               float* dx, int64_t E, int K, int nf, int mf, hipStream_t s);
// nf = mf = 128 weight gradient on the bf16x6 pipe (edge_da128.hip)
int launch_edge_da_att128(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const int32_t* order,
                          const int32_t* type_ptr, const float* z_atom, const float* q, const float* stats_by_atom, float* dA,
                          int64_t E, int K, hipStream_t s);
int launch_edge_da_split128(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                            const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                            int K, hipStream_t s);
int launch_edge_da_split256(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                            const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                            int K, hipStream_t s);
int launch_edge_da_split64(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                           const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                           int K, hipStream_t s);
static bool math_fp32_only() {
    const bool v = switches().math_fp32;
    return v;
}
int launch_edge_pertype(int mode, const float* h, const float* A, const int32_t* src, const int32_t* order,
                        const int32_t* type_ptr, const float* gate, const float* dmsg, float* out, float* dA, int K,
                        int nf, int mf, hipStream_t s);
int launch_message_dgate(const float* dagg, const float* A, const int32_t* dst, const float* w, const int32_t* order,
                         const int32_t* type_ptr, const float* hmul, const int32_t* hsrc, float* dgate, int64_t E, int K,
                         int F, hipStream_t s);
// fused H = 64 path (gru_bwd.hip)
int launch_gru_bwd_fused64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                           const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                           float* db_ih, float* db_hh, int64_t V, hipStream_t s);

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_edge_message_bwd_f32(const float* h, const float* A, const int32_t* src, const int32_t* order,
                                         const int32_t* type_ptr, const float* gate, const float* dmsg, float* dx,
                                         float* dA, int64_t V, int64_t E, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(E >= 0 && V >= 0 && K >= 0, "mpnn_edge_message_bwd_f32: negative size");
    MPNN_REQUIRE(nf > 0 && nf <= 256 && mf > 0 && mf <= MPNN_MAX_FEATURES,
                 "mpnn_edge_message_bwd_f32: nf=%d mf=%d unsupported (nf<=256)", nf, mf);
    MPNN_REQUIRE(E < (1ll << 31) && V < (1ll << 31), "mpnn_edge_message_bwd_f32: int32 index overflow");
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && src && order && type_ptr && dmsg && K > 0, "mpnn_edge_message_bwd_f32: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    int rc = MPNN_OK;
    if (K > 4096) {   // many matrices, few edges each: streaming matvec kernels (edge_pertype.hip)
        if (dx) rc = launch_edge_pertype(1, h, A, src, order, type_ptr, gate, dmsg, dx, nullptr, K, nf, mf, s);
        if (rc == MPNN_OK && dA) rc = launch_edge_pertype(2, h, A, src, order, type_ptr, gate, dmsg, nullptr, dA, K, nf, mf, s);
        return rc;
    }
    if (dx) {
        // dx[e, b] = sum_a dmsg[e, a] * A_k[a, b]  : X = dmsg (ld mf), B = A_k as [k=a][n=b]
        rc = launch_message_dx_resident(dmsg, A, order, type_ptr, dx, E, K, nf, mf, s);
        if (rc == 1) rc = launch_rows_gemm<false, true>(dmsg, mf, order, type_ptr, K, A, (int64_t)mf * nf, nf, nullptr, dx, nf, E, mf,
                                           nf, s, "mpnn_edge_message_bwd_f32(dx)");
        if (rc) return rc;
    }
    if (dA && mf == 64 && nf == 64 && K <= 64 && !math_fp32_only()) {
        rc = launch_edge_da_split64(dmsg, h, src, nullptr, nullptr, order, type_ptr, gate, dA, E, K, s);
    } else if (dA && mf == 64 && nf == 64 && K <= 64) {
        int64_t gx = 512;                               // 2 blocks of 8 waves per CU
        const int64_t need = ceil_div(ceil_div(E, 32) + K, 8);
        if (gx > need) gx = need;
        if (gate)
            hipLaunchKernelGGL((edge_dA_direct64_kernel<false, false, true>), dim3((unsigned)gx), dim3(512), 0, s, dmsg, h,
                               src, order, type_ptr, gate, dA, K, (const int32_t*)nullptr, (const float*)nullptr);
        else
            hipLaunchKernelGGL((edge_dA_direct64_kernel<false, false, false>), dim3((unsigned)gx), dim3(512), 0, s, dmsg,
                               h, src, order, type_ptr, gate, dA, K, (const int32_t*)nullptr, (const float*)nullptr);
        rc = launch_status("mpnn_edge_message_bwd_f32(dA direct)");
    } else if (dA && mf == 128 && nf == 128 && K <= 64 && !math_fp32_only()) {
        rc = launch_edge_da_split128(dmsg, h, src, nullptr, nullptr, order, type_ptr, gate, dA, E, K, s);
    } else if (dA && mf == 256 && nf == 256 && K <= 64 && !math_fp32_only()) {
        rc = launch_edge_da_split256(dmsg, h, src, nullptr, nullptr, order, type_ptr, gate, dA, E, K, s);
    } else if (dA && mf == 128 && nf == 128 && K <= 64) {
        int64_t gx = 512;
        const int64_t need = ceil_div(E, 32) + K;
        if (gx > need) gx = need;
        hipLaunchKernelGGL((tn_direct_kernel<4, 2, true>), dim3((unsigned)gx), dim3(512), 0, s, dmsg, mf, h, nf, order,
                           type_ptr, K, src, (const int32_t*)nullptr, (const float*)nullptr, gate, dA, (float*)nullptr, E);
        rc = launch_status("mpnn_edge_message_bwd_f32(dA direct 128)");
    } else if (dA) {
        const int pairs = (int)(ceil_div(mf, 64) * ceil_div(nf, 64));
        int64_t gx = ceil_div(E, kBT) + K;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL((tn_accumulate_kernel<true>), dim3((unsigned)gx, pairs), dim3(256), 0, s, dmsg, mf, mf, h, nf,
                           nf, order, src, gate, type_ptr, K, dA, (float*)nullptr, E);
        rc = launch_status("mpnn_edge_message_bwd_f32(dA)");
    }
    return rc;
}

namespace mpnn {
size_t gru_bwd_f16_workspace_bytes(int64_t V, int H);                      // gru_bwd128_f16.hip (H = 128, 256)
int launch_gru_bwd_f16_wide(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                            const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                            float* db_ih, float* db_hh, void* workspace, int64_t V, int H, const float* out_norm_k,
                            double* in_norm_sums, const float* in_norm_raw, hipStream_t s);
size_t gru_bwd_rc_workspace_bytes(int H);                                  // gru_bwd_rc.hip: no per-atom workspace
bool gru_bwd_rc_covers(int H);
int launch_gru_bwd_rc(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                      const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh, float* db_ih,
                      float* db_hh, void* workspace, int64_t V, int H, const float* out_norm_k, double* in_norm_sums,
                      const float* in_norm_raw, hipStream_t s);
// the wide backward of this width and process: gate gradients formed inside the two contractions (gru_bwd_rc.hip) unless
// the width has no such kernels yet or MPNN_GRU_BWD=pieces asks for round 3's three-kernel form
static inline bool gru_bwd_use_rc(int H) { return gru_bwd_rc_covers(H) && !switches().gru_bwd_pieces; }
static inline size_t gru_bwd_wide_workspace(int64_t V, int H) {
    return gru_bwd_use_rc(H) ? gru_bwd_rc_workspace_bytes(H) : gru_bwd_f16_workspace_bytes(V, H);
}
static inline int launch_gru_bwd_wide(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                                      const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                                      float* db_ih, float* db_hh, void* workspace, int64_t V, int H, const float* out_norm_k,
                                      double* in_norm_sums, const float* in_norm_raw, hipStream_t s) {
    if (gru_bwd_use_rc(H))
        return launch_gru_bwd_rc(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V, H,
                                 out_norm_k, in_norm_sums, in_norm_raw, s);
    return launch_gru_bwd_f16_wide(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V, H,
                                   out_norm_k, in_norm_sums, in_norm_raw, s);
}
}  // namespace mpnn

// the generic-width backward after the gate-gradient pass (ws = (V, 6H) pre-activation gradients, dh holds g * z):
// dm, dh as row GEMMs, dW / db as accumulating contractions over the atoms
namespace mpnn {
// widths up to 40, batches up to 8 k atoms: the whole backward as one vector-pipe kernel (gru_small.hip)
bool gru_small_covers(int H, int64_t V);
int launch_gru_bwd_small(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                         const float* W_hh, const float* saved, const float* out_norm_k, float* dm, float* dh, float* dW_ih,
                         float* dW_hh, float* db_ih, float* db_hh, int64_t V, int H, hipStream_t s);
}

static int gru_bwd_generic_tail(const float* m, const float* h, const float* W_ih, const float* W_hh, float* ws, float* dm,
                                float* dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, int64_t V, int H,
                                hipStream_t s) {
    int rc;
    // dm = dgi . W_ih^T : B given as [n = input feature][k = gate column], ld 3H
    rc = launch_rows_gemm<true, false>(ws, 6 * H, nullptr, nullptr, 0, W_ih, 0, 3 * H, nullptr, dm, H, V, 3 * H, H, s,
                                       "mpnn_gru_update_bwd_f32(dm)");
    if (rc) return rc;
    // dh = dgh . W_hh^T + dout*mask*z   (the direct term already sits in dh)
    rc = launch_rows_gemm<true, false>(ws + 3 * H, 6 * H, nullptr, nullptr, 0, W_hh, 0, 3 * H, dh, dh, H, V, 3 * H, H, s,
                                       "mpnn_gru_update_bwd_f32(dh)");
    if (rc) return rc;
    if (H == 128) {
        // dW (128 x 384) and db on the register-direct wide kernel: 8 waves = 4 a-blocks x 2 groups of 6 b-blocks
        int64_t gxd = 512;
        if (gxd > ceil_div(V, 32)) gxd = ceil_div(V, 32);
        hipLaunchKernelGGL((tn_direct_kernel<4, 6, false>), dim3((unsigned)gxd), dim3(512), 0, s, m, H, ws, 6 * H,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, 1, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, dW_ih, db_ih, V);
        hipLaunchKernelGGL((tn_direct_kernel<4, 6, false>), dim3((unsigned)gxd), dim3(512), 0, s, h, H, ws + 3 * H, 6 * H,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, 1, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, dW_hh, db_hh, V);
        return launch_status("mpnn_gru_update_bwd_f32(dW direct 128)");
    }
    const int pairs = (int)(ceil_div(H, 64) * ceil_div(3 * H, 64));
    int64_t gx = ceil_div(V, kBT);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL((tn_accumulate_kernel<false>), dim3((unsigned)gx, pairs), dim3(256), 0, s, m, H, H, ws, 6 * H,
                       3 * H, (const int32_t*)nullptr, (const int32_t*)nullptr, (const float*)nullptr,
                       (const int32_t*)nullptr, 1, dW_ih, db_ih, V);
    hipLaunchKernelGGL((tn_accumulate_kernel<false>), dim3((unsigned)gx, pairs), dim3(256), 0, s, h, H, H, ws + 3 * H,
                       6 * H, 3 * H, (const int32_t*)nullptr, (const int32_t*)nullptr, (const float*)nullptr,
                       (const int32_t*)nullptr, 1, dW_hh, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dW)");
}

extern "C" size_t mpnn_gru_bwd_workspace_bytes(int64_t V, int H) {
    if (V < 0 || H <= 0) return 0;
    if (H == 64) return 16;                                               // one kernel, gate gradients stay in LDS
    if ((H == 128 || H == 256) && !switches().math_fp32)                  // fp16 pieces per 32-atom tile, tile scales,
        return gru_bwd_wide_workspace(V, H);                              // pre-split weights of the dm | dh kernel (+ pieces, old form)
    return (size_t)V * 6 * H * sizeof(float);                            // generic widths: (dgi | dgh)
}

extern "C" int mpnn_gru_update_bwd_f32(const float* dout, const float* m, const float* h, const float* mask,
                                       const float* W_ih, const float* W_hh, const float* saved, float* dm, float* dh,
                                       float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* workspace,
                                       size_t workspace_bytes, int64_t V, int H, void* stream) {
    MPNN_REQUIRE(V >= 0 && H > 0 && H <= 256, "mpnn_gru_update_bwd_f32: V=%lld H=%d out of range (H<=256)",
                 (long long)V, H);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(dout && m && h && W_ih && W_hh && saved && dm && dh && dW_ih && dW_hh && db_ih && db_hh,
                 "mpnn_gru_update_bwd_f32: NULL buffer");
    if (!workspace || workspace_bytes < mpnn_gru_bwd_workspace_bytes(V, H)) {
        set_error("mpnn_gru_update_bwd_f32: workspace %zu < %zu", workspace_bytes, mpnn_gru_bwd_workspace_bytes(V, H));
        return MPNN_EWORKSPACE;
    }
    MPNN_REQUIRE(V * 6 * (int64_t)H < (1ll << 40), "mpnn_gru_update_bwd_f32: V too large");
    hipStream_t s = (hipStream_t)stream;
    if (H == 64)   // fused path: no (V,6H) workspace traffic
        return launch_gru_bwd_fused64(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, s);
    float* ws = (float*)workspace;
    int64_t g = ceil_div(V * H, 256);
    if (g > 256 * 16) g = 256 * 16;
    const bool fp32_only = switches().math_fp32;
    int rc;
    // hidden 128 / 256: gate gradients as fp16 pieces, split once (gru_bwd128_f16.hip); MPNN_GRU_MATH=fp32 and other
    // widths: elementwise gate gradients into a (V, 6H) workspace + generic fp32 contractions below
    if ((H == 128 || H == 256) && !fp32_only)
        return launch_gru_bwd_wide(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V, H,
                                   nullptr, nullptr, nullptr, s);
    if (gru_small_covers(H, V))
        return launch_gru_bwd_small(dout, m, h, mask, W_ih, W_hh, saved, nullptr, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, H, s);
    hipLaunchKernelGGL(gru_gate_grad_kernel<false>, dim3((unsigned)g), dim3(256), 0, s, dout, h, mask, saved, ws, dh, V,
                       H);
    rc = launch_status("mpnn_gru_update_bwd_f32(gates)");
    if (rc) return rc;
    return gru_bwd_generic_tail(m, h, W_ih, W_hh, ws, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, H, s);
}

extern "C" int mpnn_edge_message_agg_bwd_dgate_f32(const float* dagg, const float* A, const float* h, const int32_t* src,
                                                   const int32_t* dst, const float* w, const int32_t* order,
                                                   const int32_t* type_ptr, float* dgate, int64_t V, int64_t E, int K,
                                                   int nf, int mf, void* stream) {
    MPNN_REQUIRE(E >= 0 && V >= 0 && K >= 0, "mpnn_edge_message_agg_bwd_dgate_f32: negative size");
    MPNN_REQUIRE(nf == mf && (nf == 64 || nf == 128) && K <= 64,
                 "mpnn_edge_message_agg_bwd_dgate_f32: only nf = mf in {64, 128}, K <= 64 (got %d, %d, %d)", nf, mf, K);
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(dagg && A && h && src && dst && order && type_ptr && dgate && K > 0,
                 "mpnn_edge_message_agg_bwd_dgate_f32: NULL buffer");
    return launch_message_dgate(dagg, A, dst, w, order, type_ptr, h, src, dgate, E, K, nf, (hipStream_t)stream);
}

extern "C" int mpnn_edge_message_agg_bwd_da_f32(const float* dagg, const float* h, const int32_t* src,
                                                const int32_t* dst, const float* w, const int32_t* order,
                                                const int32_t* type_ptr, const float* gate, float* dA, int64_t V,
                                                int64_t E, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(E >= 0 && V >= 0 && K >= 0, "mpnn_edge_message_agg_bwd_da_f32: negative size");
    MPNN_REQUIRE(nf == mf && (nf == 64 || nf == 128 || nf == 256) && K <= 64,
                 "mpnn_edge_message_agg_bwd_da_f32: only nf = mf in {64, 128, 256}, K <= 64 (got %d, %d, %d)", nf, mf, K);
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(dagg && h && src && dst && order && type_ptr && dA && K > 0,
                 "mpnn_edge_message_agg_bwd_da_f32: NULL buffer");
    if (nf == 256) {
        MPNN_REQUIRE(!math_fp32_only(), "mpnn_edge_message_agg_bwd_da_f32: width 256 has no fp32-only path");
        return launch_edge_da_split256(dagg, h, src, dst, w, order, type_ptr, gate, dA, E, K, (hipStream_t)stream);
    }
    if (nf == 128 && !math_fp32_only())
        return launch_edge_da_split128(dagg, h, src, dst, w, order, type_ptr, gate, dA, E, K, (hipStream_t)stream);
    if (nf == 128) {
        int64_t gx = 512;
        const int64_t need = ceil_div(E, 32) + K;
        if (gx > need) gx = need;
        hipLaunchKernelGGL((tn_direct_kernel<4, 2, true>), dim3((unsigned)gx), dim3(512), 0, (hipStream_t)stream, dagg, mf,
                           h, nf, order, type_ptr, K, src, dst, w, gate, dA, (float*)nullptr, E);
        return launch_status("mpnn_edge_message_agg_bwd_da_f32(128)");
    }
    if (!math_fp32_only())
        return launch_edge_da_split64(dagg, h, src, dst, w, order, type_ptr, gate, dA, E, K, (hipStream_t)stream);
    int64_t gx = 512;
    const int64_t need = ceil_div(ceil_div(E, 32) + K, 8);
    if (gx > need) gx = need;
#define MPNN_DA64(W, G)                                                                                              \
    hipLaunchKernelGGL((edge_dA_direct64_kernel<true, W, G>), dim3((unsigned)gx), dim3(512), 0, (hipStream_t)stream, dagg, \
                       h, src, order, type_ptr, gate, dA, K, dst, w)
    if (w) { if (gate) MPNN_DA64(true, true); else MPNN_DA64(true, false); }
    else { if (gate) MPNN_DA64(false, true); else MPNN_DA64(false, false); }
#undef MPNN_DA64
    return launch_status("mpnn_edge_message_agg_bwd_da_f32");
}

extern "C" int mpnn_edge_message_agg_bwd_da_att_f32(const float* dagg, const float* h, const int32_t* src,
                                                    const int32_t* dst, const int32_t* order, const int32_t* type_ptr,
                                                    const float* z_atom, const float* q, const float* stats_by_atom,
                                                    float* dA, int64_t V, int64_t E, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(E >= 0 && V >= 0, "mpnn_edge_message_agg_bwd_da_att_f32: negative size");
    MPNN_REQUIRE(nf == mf && nf == 128 && K >= 1 && K <= 64 && !math_fp32_only(),
                 "mpnn_edge_message_agg_bwd_da_att_f32: nf = mf = 128, split math only (got %d, %d, K = %d)", nf, mf, K);
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(dagg && h && src && dst && order && type_ptr && z_atom && q && stats_by_atom && dA,
                 "mpnn_edge_message_agg_bwd_da_att_f32: NULL buffer");
    return launch_edge_da_att128(dagg, h, src, dst, order, type_ptr, z_atom, q, stats_by_atom, dA, E, K, (hipStream_t)stream);
}

int mpnn_gru_update_norm_supported(int H);

// the generic form keeps the (V, 6H) pre-activation gradients whatever the width (mpnn_gru_bwd_workspace_bytes is 16 bytes
// at H = 64, whose plain backward is one fused kernel)
extern "C" size_t mpnn_gru_norm_bwd_workspace_bytes(int64_t V, int H) {
    if (V < 0 || H <= 0) return 0;
    if (mpnn_gru_update_norm_supported(H) == 2) return gru_bwd_wide_workspace(V, H);
    return (size_t)V * 6 * H * sizeof(float);
}

extern "C" int mpnn_gru_update_norm_bwd_f32(const float* dout, const float* m, const float* h_norm, const float* mask,
                                            const float* W_ih, const float* W_hh, const float* saved,
                                            const float* out_norm_k, float* dm, float* dh_norm, float* dW_ih,
                                            float* dW_hh, float* db_ih, float* db_hh, double* in_norm_sums,
                                            const float* h_raw, void* workspace, size_t workspace_bytes, int64_t V, int H,
                                            void* stream) {
    const int kind = mpnn_gru_update_norm_supported(H);
    MPNN_REQUIRE(kind, "mpnn_gru_update_norm_bwd_f32: no fused-norm kernels at H=%d (1 <= H <= 256)", H);
    MPNN_REQUIRE(V >= 0, "mpnn_gru_update_norm_bwd_f32: V=%lld out of range", (long long)V);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(dout && m && h_norm && W_ih && W_hh && saved && dm && dh_norm && dW_ih && dW_hh && db_ih && db_hh,
                 "mpnn_gru_update_norm_bwd_f32: NULL buffer");
    MPNN_REQUIRE(!in_norm_sums || h_raw, "mpnn_gru_update_norm_bwd_f32: in_norm_sums needs h_raw (the norm's raw input)");
    if (!workspace || workspace_bytes < mpnn_gru_norm_bwd_workspace_bytes(V, H)) {
        set_error("mpnn_gru_update_norm_bwd_f32: workspace %zu < %zu", workspace_bytes, mpnn_gru_norm_bwd_workspace_bytes(V, H));
        return MPNN_EWORKSPACE;
    }
    MPNN_REQUIRE(V * 6 * (int64_t)H < (1ll << 40), "mpnn_gru_update_norm_bwd_f32: V too large");
    if (kind == 1) {
        // generic widths: the plain backward with the gate-gradient pass in its NORM form, then the two column sums of
        // dh_norm in a pass of their own (the generic dm / dh are library-style GEMM launches without an epilogue hook)
        hipStream_t s = (hipStream_t)stream;
        if (gru_small_covers(H, V)) {
            const int rc = launch_gru_bwd_small(dout, m, h_norm, mask, W_ih, W_hh, saved, out_norm_k, dm, dh_norm, dW_ih,
                                                dW_hh, db_ih, db_hh, V, H, s);
            if (rc) return rc;
            if (in_norm_sums) return mpnn_norm_bwd_sums_f32(dh_norm, h_raw, nullptr, in_norm_sums, V, H, stream);
            return MPNN_OK;
        }
        float* ws = (float*)workspace;
        int64_t g = ceil_div(V * H, 256);
        if (g > 256 * 16) g = 256 * 16;
        if (out_norm_k)
            hipLaunchKernelGGL((gru_gate_grad_kernel<false, true>), dim3((unsigned)g), dim3(256), 0, s, dout, h_norm, mask, saved,
                               ws, dh_norm, V, H, out_norm_k);
        else
            hipLaunchKernelGGL((gru_gate_grad_kernel<false, false>), dim3((unsigned)g), dim3(256), 0, s, dout, h_norm, mask,
                               saved, ws, dh_norm, V, H, (const float*)nullptr);
        int rc = launch_status("mpnn_gru_update_norm_bwd_f32(gates)");
        if (rc) return rc;
        rc = gru_bwd_generic_tail(m, h_norm, W_ih, W_hh, ws, dm, dh_norm, dW_ih, dW_hh, db_ih, db_hh, V, H, s);
        if (rc) return rc;
        if (in_norm_sums) return mpnn_norm_bwd_sums_f32(dh_norm, h_raw, nullptr, in_norm_sums, V, H, stream);
        return MPNN_OK;
    }
    return launch_gru_bwd_wide(dout, m, h_norm, mask, W_ih, W_hh, saved, dm, dh_norm, dW_ih, dW_hh, db_ih, db_hh, workspace,
                               V, H, out_norm_k, in_norm_sums, h_raw, (hipStream_t)stream);
}
