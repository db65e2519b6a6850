// Masked GRU node update, fused: two fp32-MFMA GEMMs + gate epilogue in one kernel.
//
//   gi = m W_ih + b_ih ; gh = h W_hh + b_hh            (W stored (H, 3H), gate order r,z,n)
//   r = sig(gi_r+gh_r)*mask ; z = sig(gi_z+gh_z)*mask ; n = tanh(gi_n + r*gh_n)*mask
//   out = ((1-z)*n + z*h) * mask
//
// 12*H^2 flops per atom against 12*H bytes: matrix-core bound for H >= 64 (fp32 MFMA, exact).
//
// Tiling: a block owns 128 atoms x 32 hidden columns of ALL gates, so a wave (32 atoms) keeps
// four 32x32 accumulators -- r, z, gi_n, gh_n for the same (atom, column) elements in the same
// lane/register -- and the gate math needs no cross-wave exchange.  r and z accumulate the m-
// and h-products into one accumulator (K = 2H), gi_n / gh_n stay apart because of r*gh_n.
// Blocks that share an atom tile (the H/32 column slices) are dealt to the same XCD so the
// tile's second read hits that XCD's L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mpnn {

constexpr int kRows = 128;
constexpr int kKC = 64;
constexpr int kLDX = kKC + 4;     // Xs row stride (floats): conflict-free ds_read_b128
constexpr int kLDB = 96;          // Bs row stride: 3 gates x 32 columns, k-major

// Gate math on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Absolute error
// of sigmoid <= ~2e-7 and of tanh <= ~5e-7 on the whole real line -- far inside the 1e-5 parity
// bar -- at ~5 VALU instructions instead of the ~40-60 of libm's expf/tanhf + IEEE division,
// which matters because the epilogue competes with the MFMA stream for issue slots.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2 / (1 + e^{2x}); exp2 saturates to +inf / 0 so the limits are exact +-1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}

// acc{0,1,2} += X[128 x KC] . W[k0:k0+KC, {0,H,2H}+c0 : +32]
__device__ __forceinline__ void gru_gemm_pass(const float* __restrict__ X, const float* __restrict__ W,
                                              float* Xs, float* Bs, int64_t i0, int64_t V, int H, int c0, bool vec,
                                              f32x16& acc0, f32x16& acc1, f32x16& acc2) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c4 = tid & 15, r0 = tid >> 4;
    const int r = lane & 31, hi = lane >> 5;
    for (int k0 = 0; k0 < H; k0 += kKC) {
        __syncthreads();   // previous pass finished reading Xs/Bs
        const int k = k0 + 4 * c4;
#pragma unroll
        for (int p = 0; p < kRows / 16; ++p) {
            const int row = r0 + 16 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i0 + row < V) {
                const float* src = X + (i0 + row) * H + k;
                if (vec) {
                    if (k < H) v = *reinterpret_cast<const f32x4*>(src);
                } else {
                    if (k + 0 < H) v.x = src[0];
                    if (k + 1 < H) v.y = src[1];
                    if (k + 2 < H) v.z = src[2];
                    if (k + 3 < H) v.w = src[3];
                }
            }
            *reinterpret_cast<f32x4*>(Xs + row * kLDX + 4 * c4) = v;
        }
#pragma unroll
        for (int p = 0; p < (kKC * 24) / 256; ++p) {
            const int idx = tid + 256 * p;
            const int kk = idx / 24, q = idx % 24, g = q >> 3, j4 = q & 7;
            const int col = c0 + 4 * j4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + kk < H) {
                const float* src = W + (int64_t)(k0 + kk) * 3 * H + g * H + col;
                if (vec) {
                    if (col < H) v = *reinterpret_cast<const f32x4*>(src);
                } else {
                    if (col + 0 < H) v.x = src[0];
                    if (col + 1 < H) v.y = src[1];
                    if (col + 2 < H) v.z = src[2];
                    if (col + 3 < H) v.w = src[3];
                }
            }
            *reinterpret_cast<f32x4*>(Bs + kk * kLDB + g * 32 + 4 * j4) = v;
        }
        __syncthreads();
        if (i0 + 32 * wv < V) {
            const float* xa = Xs + (32 * wv + r) * kLDX + hi * (kKC / 2);
            const float* xb = Bs + hi * (kKC / 2) * kLDB + r;
#pragma unroll
            for (int kq = 0; kq < kKC / 8; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xa + 4 * kq);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* bp = xb + (4 * kq + c) * kLDB;
                    const float av = a[c];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[64], acc2, 0, 0, 0);
                }
            }
        }
    }
}

// NORM: the masked batch norm in front of the `h` input fused in, as in gru_split.hip's wide kernel (same contract: `h` is
// the previous update's raw output, hn = (h * hs[col] + ht[col]) * mask is formed where h is read, W_hh / b_hh arrive with
// the map folded in; the column sums of out and out^2 go to `stats` (2H doubles), hn is written when `saved` is).
template <bool NORM>
__global__ void __launch_bounds__(256) gru_update_kernel(const float* __restrict__ m, const float* __restrict__ h,
                                                         const float* __restrict__ mask,
                                                         const float* __restrict__ W_ih, const float* __restrict__ W_hh,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         float* __restrict__ out, float* __restrict__ saved, int64_t V,
                                                         int H, int row_tiles, int col_slices, const float* __restrict__ hs,
                                                         const float* __restrict__ ht, float* __restrict__ hnorm,
                                                         double* stats) {
    __shared__ __attribute__((aligned(16))) float Xs[kRows * kLDX];
    __shared__ __attribute__((aligned(16))) float Bs[kKC * kLDB];

    // XCD-aware placement: blocks b and b+8 share an XCD (speed only, never correctness)
    const int xcd = blockIdx.x % kNumXcd, slot = blockIdx.x / kNumXcd;
    const int cs = slot % col_slices;
    const int rt = (slot / col_slices) * kNumXcd + xcd;
    if (rt >= row_tiles) return;   // uniform per block
    const int64_t i0 = (int64_t)rt * kRows;
    const int c0 = cs * 32;
    const bool vec = (H & 3) == 0;

    f32x16 acc_r, acc_z, acc_ni, acc_nh;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_r[i] = 0.f; acc_z[i] = 0.f; acc_ni[i] = 0.f; acc_nh[i] = 0.f; }

    gru_gemm_pass(m, W_ih, Xs, Bs, i0, V, H, c0, vec, acc_r, acc_z, acc_ni);
    gru_gemm_pass(h, W_hh, Xs, Bs, i0, V, H, c0, vec, acc_r, acc_z, acc_nh);

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = c0 + (lane & 31);
    if (col >= H) return;
    const float br = b_ih[col] + b_hh[col];
    const float bz = b_ih[H + col] + b_hh[H + col];
    const float bni = b_ih[2 * H + col], bnh = b_hh[2 * H + col];
    const float hsc = NORM ? hs[col] : 1.0f, hsh = NORM ? ht[col] : 0.0f;
    double sum1 = 0.0, sum2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t row = i0 + 32 * wv + acc_row(i, lane);
        if (row >= V) continue;
        const float mk = mask ? mask[row] : 1.0f;
        float hv = h[row * H + col];
        if (NORM) hv = fmaf(hv, hsc, hsh) * mk;
        const float r = sigmoidf_(acc_r[i] + br) * mk;
        const float z = sigmoidf_(acc_z[i] + bz) * mk;
        const float nh = acc_nh[i] + bnh;
        const float n = tanhf_(acc_ni[i] + bni + r * nh) * mk;
        const float o = ((1.0f - z) * n + z * hv) * mk;
        out[row * H + col] = o;
        if (NORM) {
            sum1 += (double)o;
            sum2 += (double)o * (double)o;
        }
        if (saved) {
            float* sv = saved + row * 4 * H + col;
            sv[0] = r;
            sv[H] = z;
            sv[2 * H] = n;
            sv[3 * H] = nh;
            if (NORM) hnorm[row * H + col] = hv;
        }
    }
    if (NORM) {                                            // the two lane halves hold the same column
        sum1 += __shfl_xor(sum1, 32);
        sum2 += __shfl_xor(sum2, 32);
        if (lane < 32) {
            atomicAdd(stats + col, sum1);
            atomicAdd(stats + H + col, sum2);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fast path (H a multiple of 32, weight slice fits LDS): persistent waves, no per-tile barrier.
//
// A block keeps the (H x 3*CS) slices of W_ih and W_hh for its CS = 32*NCS hidden columns resident
// in LDS for its whole life (k-major, so a B fragment is one conflict-free ds_read_b32).  Each of
// its 8 waves then walks 32-atom tiles on its own: the A fragments of v_mfma_f32_32x32x2_f32 are
// the lane's own contiguous half-row of m / h (lane (r,hi) holds X[row r][hi*H/2 ...]), loaded
// straight from global into registers -- no LDS image, no __syncthreads in the loop -- with the
// next operand block requested before the current one is multiplied.
template <int H, int NCS, int NW, bool HAS_MASK>   // NW waves per block (8 = two per SIMD)
__global__ void __launch_bounds__(64 * NW) gru_update_resident_kernel(
    const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask,
    const float* __restrict__ W_ih, const float* __restrict__ W_hh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ saved, int64_t V, int slices) {
    constexpr int CS = 32 * NCS;
    constexpr int LDW = 3 * CS;              // multiple of 32: lanes j..j+31 hit 32 distinct banks
    constexpr int NF4 = H / 8;               // float4 fragments per lane per operand
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Wi = reinterpret_cast<float*>(smem_raw);     // [H][LDW]
    float* Wh = Wi + H * LDW;                           // [H][LDW]

    const int slice = blockIdx.x % slices;              // neighbouring blocks share the atom range
    const int pblock = blockIdx.x / slices, pblocks = gridDim.x / slices;
    const int c0 = slice * CS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // one-time weight slice load: W[k][g*H + c0 + c] -> Ws[k][g*CS + c]
    for (int idx = tid; idx < H * (LDW / 4); idx += 64 * NW) {
        const int k = idx / (LDW / 4), q = idx % (LDW / 4);
        const int g = (4 * q) / CS, c = (4 * q) % CS;
        const int64_t srcoff = (int64_t)k * 3 * H + g * H + c0 + c;
        *reinterpret_cast<f32x4*>(Wi + k * LDW + 4 * q) = *reinterpret_cast<const f32x4*>(W_ih + srcoff);
        *reinterpret_cast<f32x4*>(Wh + k * LDW + 4 * q) = *reinterpret_cast<const f32x4*>(W_hh + srcoff);
    }
    __syncthreads();

    const int r = lane & 31, hi = lane >> 5;
    float br[NCS], bz[NCS], bni[NCS], bnh[NCS];
#pragma unroll
    for (int s = 0; s < NCS; ++s) {
        const int col = c0 + 32 * s + r;
        br[s] = b_ih[col] + b_hh[col];
        bz[s] = b_ih[H + col] + b_hh[H + col];
        bni[s] = b_ih[2 * H + col];
        bnh[s] = b_hh[2 * H + col];
    }
    const float* wi_lane = Wi + hi * (H / 2) * LDW + r;
    const float* wh_lane = Wh + hi * (H / 2) * LDW + r;

    const int64_t tiles = (V + 31) / 32;
    const int64_t stride = (int64_t)pblocks * NW;
    int64_t t = (int64_t)pblock * NW + wv;
    if (t >= tiles) return;

    f32x4 fa[NF4], fb[NF4];
    auto load_frags = [&](const float* __restrict__ X, int64_t tile, f32x4 (&f)[NF4]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;                       // clamp: loads stay in bounds, stores are skipped
        const float* p = X + row * H + hi * (H / 2);
#pragma unroll
        for (int q = 0; q < NF4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    load_frags(m, t, fa);
    for (; t < tiles; t += stride) {
        f32x16 acc_r[NCS], acc_z[NCS], acc_ni[NCS], acc_nh[NCS];
#pragma unroll
        for (int s = 0; s < NCS; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_r[s][i] = 0.f; acc_z[s][i] = 0.f; acc_ni[s][i] = 0.f; acc_nh[s][i] = 0.f; }

        load_frags(h, t, fb);                            // in flight while the m-products run
#pragma unroll
        for (int q = 0; q < NF4; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float av = fa[q][c];
                const float* bp = wi_lane + (4 * q + c) * LDW;
#pragma unroll
                for (int s = 0; s < NCS; ++s) {
                    acc_r[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32 * s], acc_r[s], 0, 0, 0);
                    acc_z[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[CS + 32 * s], acc_z[s], 0, 0, 0);
                    acc_ni[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * CS + 32 * s], acc_ni[s], 0, 0, 0);
                }
            }
        }
        if (t + stride < tiles) load_frags(m, t + stride, fa);   // next tile's m rows, behind the h-products
#pragma unroll
        for (int q = 0; q < NF4; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float av = fb[q][c];
                const float* bp = wh_lane + (4 * q + c) * LDW;
#pragma unroll
                for (int s = 0; s < NCS; ++s) {
                    acc_r[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32 * s], acc_r[s], 0, 0, 0);
                    acc_z[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[CS + 32 * s], acc_z[s], 0, 0, 0);
                    acc_nh[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * CS + 32 * s], acc_nh[s], 0, 0, 0);
                }
            }
        }
        // epilogue in 4 groups of 4 consecutive atoms: every load is unconditional (row clamped) and
        // issued before the group's math, only the stores are predicated -- no branch holds a load
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float mk4[4], hv4[4][NCS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t row = t * 32 + 8 * g + 4 * hi + u;
                if (row >= V) row = V - 1;
                mk4[u] = HAS_MASK ? mask[row] : 1.0f;
#pragma unroll
                for (int s = 0; s < NCS; ++s) hv4[u][s] = h[row * H + c0 + 32 * s + r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * g + u;
                const int64_t row = t * 32 + 8 * g + 4 * hi + u;
                const float mk = mk4[u];
#pragma unroll
                for (int s = 0; s < NCS; ++s) {
                    const int col = c0 + 32 * s + r;
                    const float rg = sigmoidf_(acc_r[s][i] + br[s]) * mk;
                    const float zg = sigmoidf_(acc_z[s][i] + bz[s]) * mk;
                    const float nh = acc_nh[s][i] + bnh[s];
                    const float ng = tanhf_(acc_ni[s][i] + bni[s] + rg * nh) * mk;
                    const float o = ((1.0f - zg) * ng + zg * hv4[u][s]) * mk;
                    if (row < V) {
                        out[row * H + col] = o;
                        if (saved) {
                            float* sv = saved + row * 4 * H + col;
                            sv[0] = rg;
                            sv[H] = zg;
                            sv[2 * H] = ng;
                            sv[3 * H] = nh;
                        }
                    }
                }
            }
        }
    }
}

template <int H, int NCS, int NW>
static int launch_gru_resident(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                               const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V,
                               hipStream_t s) {
    constexpr int CS = 32 * NCS;
    constexpr int slices = H / CS;
    const size_t lds = (size_t)2 * H * 3 * CS * sizeof(float);
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_update_resident_kernel<H, NCS, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_update_resident_kernel<H, NCS, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t tiles = (V + 31) / 32;
    int64_t pblocks = (256 + slices - 1) / slices;        // one block per CU (LDS-bound residency)
    if (pblocks * NW > tiles) pblocks = (tiles + NW - 1) / NW;
    if (pblocks < 1) pblocks = 1;
    if (mask)
        hipLaunchKernelGGL((gru_update_resident_kernel<H, NCS, NW, true>), dim3((unsigned)(pblocks * slices)),
                           dim3(64 * NW), lds, s, m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, slices);
    else
        hipLaunchKernelGGL((gru_update_resident_kernel<H, NCS, NW, false>), dim3((unsigned)(pblocks * slices)),
                           dim3(64 * NW), lds, s, m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, slices);
    return launch_status("mpnn_gru_update_f32(resident)");
}

// split-precision (bf16x6) forward, gru_split.hip; returns 1 when the width is not covered
int launch_gru_split(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                     const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, int H, void* workspace,
                     hipStream_t s);
size_t gru_fwd_workspace_bytes(int H);
int launch_gru_split_norm(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                          const float* b_ih, const float* b_hh, const float* hs, const float* ht, float* out,
                          float* saved, float* hnorm, double* stats, int64_t V, int H, void* workspace, hipStream_t s);

// widths up to 40 on the vector pipe (gru_small.hip)
bool gru_small_covers(int H, int64_t V);
int launch_gru_small(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                     const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, int H, const float* hs,
                     const float* ht, float* hnorm, double* stats, hipStream_t s);

}  // namespace mpnn

using namespace mpnn;

extern "C" size_t mpnn_gru_fwd_workspace_bytes(int64_t V, int H) {
    (void)V;
    if (H <= 0 || switches().math_fp32) return 0;
    return gru_fwd_workspace_bytes(H);
}

extern "C" int mpnn_gru_update_f32(const float* m, const float* h, const float* mask, const float* W_ih,
                                   const float* W_hh, const float* b_ih, const float* b_hh, float* out, float* saved,
                                   void* workspace, size_t workspace_bytes, int64_t V, int H, void* stream) {
    MPNN_REQUIRE(V >= 0 && H > 0 && H <= MPNN_MAX_FEATURES, "mpnn_gru_update_f32: V=%lld H=%d out of range",
                 (long long)V, H);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(m && h && W_ih && W_hh && b_ih && b_hh && out, "mpnn_gru_update_f32: NULL buffer");
    if ((H & 3) == 0) {
        const uintptr_t al = reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(h) |
                             reinterpret_cast<uintptr_t>(W_ih) | reinterpret_cast<uintptr_t>(W_hh);
        MPNN_REQUIRE(al % 16 == 0, "mpnn_gru_update_f32: buffers must be 16-byte aligned");
    }
    hipStream_t st = (hipStream_t)stream;
    // MPNN_GRU_MATH=fp32 keeps the GEMMs on v_mfma_f32_32x32x2_f32; default: bf16 pipe with 3-way operand
    // splitting (fp32-equivalent accuracy, see gru_split.hip)
    const bool fp32_only = switches().math_fp32;
    if (!fp32_only) {
        // the pre-split weight images of the streamed kernels live in the caller's workspace; without one (NULL or too
        // small) the kernels split their weight chunks themselves
        void* ws = (workspace && workspace_bytes >= gru_fwd_workspace_bytes(H) && gru_fwd_workspace_bytes(H)) ? workspace : nullptr;
        const int rc = launch_gru_split(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, H, ws, st);
        if (rc != 1) return rc;
    }
    if (H == 64) return launch_gru_resident<64, 2, 8>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, st);
    if (H == 128) return launch_gru_resident<128, 1, 4>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, st);
    if (H == 32) return launch_gru_resident<32, 1, 8>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, st);
    if (gru_small_covers(H, V))
        return launch_gru_small(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, H, nullptr, nullptr, nullptr, nullptr, st);
    const int64_t row_tiles = ceil_div(V, kRows);
    MPNN_REQUIRE(row_tiles < (1 << 24), "mpnn_gru_update_f32: V too large for one launch");
    const int col_slices = (H + 31) / 32;
    const int64_t blocks = ceil_div(row_tiles, kNumXcd) * kNumXcd * col_slices;
    hipLaunchKernelGGL(gru_update_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m, h, mask, W_ih,
                       W_hh, b_ih, b_hh, out, saved, V, H, (int)row_tiles, col_slices, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, (double*)nullptr);
    return launch_status("mpnn_gru_update_f32");
}

// 2: the wide split-precision kernels (H = 128 / 256); 1: the generic fp32 kernel (every other width, and every width
// under MPNN_GRU_MATH=fp32) -- the kernel that runs the plain update at those widths anyway, except H = 32 / 64 / 128,
// which have faster forms without the norm; 0: none
extern "C" int mpnn_gru_update_norm_supported(int H) {
    if (H <= 0 || H > 256) return 0;
    return (H == 128 || H == 256) && !switches().math_fp32 ? 2 : 1;
}

extern "C" int mpnn_gru_update_norm_f32(const float* m, const float* h_raw, const float* mask, const float* W_ih,
                                        const float* W_hh_folded, const float* b_ih, const float* b_hh_folded,
                                        const float* h_scale, const float* h_shift, float* out, float* saved,
                                        float* h_norm, double* out_moments, void* workspace, size_t workspace_bytes,
                                        int64_t V, int H, void* stream) {
    const int kind = mpnn_gru_update_norm_supported(H);
    MPNN_REQUIRE(kind, "mpnn_gru_update_norm_f32: no fused-norm kernel at H=%d (1 <= H <= 256)", H);
    MPNN_REQUIRE(V >= 0, "mpnn_gru_update_norm_f32: V=%lld out of range", (long long)V);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(m && h_raw && W_ih && W_hh_folded && b_ih && b_hh_folded && h_scale && h_shift && out && out_moments,
                 "mpnn_gru_update_norm_f32: NULL buffer");
    MPNN_REQUIRE(!saved || h_norm, "mpnn_gru_update_norm_f32: a training pass (saved != NULL) also needs h_norm");
    if (kind == 1 && gru_small_covers(H, V))                  // the Lipophilicity model's widths: vector-pipe kernel, no workspace
        return launch_gru_small(m, h_raw, mask, W_ih, W_hh_folded, b_ih, b_hh_folded, out, saved, V, H, h_scale, h_shift,
                                h_norm, out_moments, (hipStream_t)stream);
    if (kind == 1) {                                       // generic fp32 kernel: no workspace
        const int64_t row_tiles = ceil_div(V, kRows);
        MPNN_REQUIRE(row_tiles < (1 << 24), "mpnn_gru_update_norm_f32: V too large for one launch");
        const int col_slices = (H + 31) / 32;
        const int64_t blocks = ceil_div(row_tiles, kNumXcd) * kNumXcd * col_slices;
        hipLaunchKernelGGL(gru_update_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m, h_raw, mask,
                           W_ih, W_hh_folded, b_ih, b_hh_folded, out, saved, V, H, (int)row_tiles, col_slices, h_scale,
                           h_shift, h_norm, out_moments);
        return launch_status("mpnn_gru_update_norm_f32(generic)");
    }
    MPNN_REQUIRE(workspace && workspace_bytes >= gru_fwd_workspace_bytes(H),
                 "mpnn_gru_update_norm_f32: workspace of mpnn_gru_fwd_workspace_bytes(V, H) bytes required");
    const uintptr_t al = reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(h_raw) |
                         reinterpret_cast<uintptr_t>(W_ih) | reinterpret_cast<uintptr_t>(W_hh_folded) |
                         reinterpret_cast<uintptr_t>(workspace);
    MPNN_REQUIRE(al % 16 == 0 && reinterpret_cast<uintptr_t>(out_moments) % 8 == 0,
                 "mpnn_gru_update_norm_f32: buffers must be 16-byte aligned");
    const int rc = launch_gru_split_norm(m, h_raw, mask, W_ih, W_hh_folded, b_ih, b_hh_folded, h_scale, h_shift, out,
                                         saved, h_norm, out_moments, V, H, workspace, (hipStream_t)stream);
    MPNN_REQUIRE(rc != 1, "mpnn_gru_update_norm_f32: width %d not covered", H);
    return rc;
}
