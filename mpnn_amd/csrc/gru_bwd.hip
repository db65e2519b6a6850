// Fused backward of the masked GRU update for hidden width 64 (the headline configuration).
//
// Given dout and the saved forward gates (r, z, n, gh_n):
//   g   = dout*mask                      dn  = g*(1-z)        dz = g*(h-n)
//   dan = dn*mask*(1-n^2)                dar = dan*gh_n*mask*r*(1-r)
//   daz = dz*mask*z*(1-z)                dnh = dan*r
//   dgi = [dar daz dan]   dgh = [dar daz dnh]
//   dm  = dgi W_ih^T      dh  = dgh W_hh^T + g*z
//   dW_ih = m^T dgi       dW_hh = h^T dgh      db_ih = colsum(dgi)     db_hh = colsum(dgh)
//
// One fused kernel; the (V, 6H) pre-activation gradients are never materialised in HBM.  This file holds the strict
// fp32 form (MPNN_GRU_MATH=fp32: fp32 MFMA throughout, 3.0 ms on c2) and the dispatcher; the default is the same block
// structure on two fp16 pieces per operand (gru_bwd_f16.hip, 1.5 ms).  Rounds 1-2 also carried a three-bf16-piece form
// (2.5 -> 1.9 ms) and an all-waves-identical arrangement (2.8 ms); both were removed in round 3 (git history has them).
// History (measured, dropped): separate elementwise + 4 generic GEMM launches with a (V,6H) workspace
// 8.8 ms; a per-lane-row kernel computing gate gradients directly in fragment layout 5.3 ms (every
// 128-B line touched 16 B at a time, microseconds apart -> refetched from beyond L2).
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mpnn {

__device__ __forceinline__ void gate_grads4(const f32x4& dout, const f32x4& hv, const f32x4& r, const f32x4& z,
                                            const f32x4& n, const f32x4& nh, float mk, f32x4& dar, f32x4& daz,
                                            f32x4& dan, f32x4& dnh) {
    const f32x4 g = dout * mk;
    const f32x4 dn = g * (1.0f - z);
    const f32x4 dz = g * (hv - n);
    dan = dn * mk * (1.0f - n * n);
    dar = dan * nh * mk * r * (1.0f - r);
    daz = dz * mk * z * (1.0f - z);
    dnh = dan * r;
}

// ------------------------------------------------------------------------------------------ fused
// One kernel for the whole GRU backward at H = 64.  Every operand is read from HBM exactly once with
// full-line coalesced loads: the block's 512 threads compute the gate gradients of a 32-atom tile
// elementwise and park   G[row][ dar | daz | dan | dnh | g*z ]  and  X[row][ m | h ]   in a
// double-buffered LDS tile (one barrier per tile).  The 8 waves then split the two GEMM families:
//   waves 0-3  dW:  wave = (matrix, 32-row block of dW) x all six 32-column blocks, 6 accumulators
//              that live in registers for the block's whole life (flushed once with float atomics);
//   waves 4-7  dx:  wave = (dm | dh) x (32-column block); its 96-register B operand -- the slice of
//              W_ih / W_hh it multiplies by -- is loaded once and stays in registers, the A operand
//              comes from the LDS tile (ds_read_b128, row stride 5H+4 -> conflict-free).
// Both roles issue 96 MFMAs per tile, so the SIMD pairs stay balanced; 768 MFMAs per 32 atoms is the
// matrix-core floor of this backward (2x the forward).
template <int H, bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_bwd_fused_kernel(const float* __restrict__ dout, const float* __restrict__ m,
                                                            const float* __restrict__ h, const float* __restrict__ mask,
                                                            const float* __restrict__ W_ih,
                                                            const float* __restrict__ W_hh,
                                                            const float* __restrict__ saved, float* __restrict__ dm,
                                                            float* __restrict__ dh, float* dW_ih, float* dW_hh,
                                                            float* db_ih, float* db_hh, int64_t V) {
    static_assert(H == 64, "role split below is laid out for H = 64");
    constexpr int LDG = 5 * H + 4;        // dar | daz | dan | dnh | g*z   (+4: conflict-free b128 rows)
    constexpr int LDX = 2 * H;            // m | h
    constexpr int TILE_F = 32 * (LDG + LDX);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* buf = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;      // staging role: (row, 4 columns)
    const bool dw_role = wv < 4;                          // wave-uniform
    const int64_t tiles = (V + 31) / 32;

    // ---- role state ----
    // dW role: matrix + 32-row block; six column blocks
    const int mat = (wv >> 1) & 1, iblk = wv & 1;
    // ONE 96-register set per wave, used by role: dW waves keep their six 32x32 accumulators in it,
    // dx waves keep their B operand in it -- W[n = 32nb+i][g*H + hi*32 + 4q + c] at float (g*8+q)*4+c --
    // (a wave never changes role, so the two uses never coexist)
    f32x16 acc[6];
    // dx role: which product (0: dm with W_ih, 1: dh with W_hh), which 32-column block
    const int which = ((wv - 4) >> 1) & 1, nb = (wv - 4) & 1;
    if (dw_role) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    } else {
        const float* Wsrc = (which == 0 ? W_ih : W_hh) + (int64_t)(32 * nb + i) * 3 * H + hi * (H / 2);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < H / 8; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wsrc + g * H + 4 * q);
                const int f = (g * (H / 8) + q) * 4;
                acc[f / 16][f % 16 + 0] = w4.x;
                acc[f / 16][f % 16 + 1] = w4.y;
                acc[f / 16][f % 16 + 2] = w4.z;
                acc[f / 16][f % 16 + 3] = w4.w;
            }
    }
    float colsum = 0.f;                                   // threads 256..511: column (tid-256) of dar|daz|dan|dnh

    // staging is split (issue-early / write-late): the next tile's seven 16-byte loads are issued
    // BEFORE this tile's MFMAs and only consumed (gate math + LDS writes) after them, so their HBM
    // latency hides under the matrix work instead of stalling every wave at the top of the tile
    struct Staged { f32x4 v_do, vh, vm, v_r, v_z, v_n, v_nh; float mk; bool ok; };
    auto stage_load = [&](int64_t t) {
        Staged q;
        int64_t row = t * 32 + srow;
        q.ok = row < V;
        if (!q.ok) row = V - 1;                           // clamped: loads unconditional, result zeroed later
        q.mk = HAS_MASK ? mask[row] : 1.0f;
        q.v_do = *reinterpret_cast<const f32x4*>(dout + row * H + sc4);
        q.vh = *reinterpret_cast<const f32x4*>(h + row * H + sc4);
        q.vm = *reinterpret_cast<const f32x4*>(m + row * H + sc4);
        const float* sv = saved + row * 4 * H + sc4;
        q.v_r = *reinterpret_cast<const f32x4*>(sv);
        q.v_z = *reinterpret_cast<const f32x4*>(sv + H);
        q.v_n = *reinterpret_cast<const f32x4*>(sv + 2 * H);
        q.v_nh = *reinterpret_cast<const f32x4*>(sv + 3 * H);
        return q;
    };
    auto stage_write = [&](const Staged& q, float* G, float* X) {
        f32x4 dar, daz, dan, dnh;
        const float mk = q.ok ? q.mk : 0.0f;              // rows past V contribute exact zeros
        gate_grads4(q.v_do, q.vh, q.v_r, q.v_z, q.v_n, q.v_nh, mk, dar, daz, dan, dnh);
        const f32x4 gz = q.v_do * mk * q.v_z;
        const float live = q.ok ? 1.0f : 0.0f;
        float* g = G + srow * LDG + sc4;
        *reinterpret_cast<f32x4*>(g) = dar;
        *reinterpret_cast<f32x4*>(g + H) = daz;
        *reinterpret_cast<f32x4*>(g + 2 * H) = dan;
        *reinterpret_cast<f32x4*>(g + 3 * H) = dnh;
        *reinterpret_cast<f32x4*>(g + 4 * H) = gz;
        float* x = X + srow * LDX + sc4;
        *reinterpret_cast<f32x4*>(x) = q.vm * live;
        *reinterpret_cast<f32x4*>(x + H) = q.vh * live;
    };

    int64_t t = blockIdx.x;
    int cur = 0;
    if (t < tiles) {
        const Staged q0 = stage_load(t);
        stage_write(q0, buf, buf + 32 * LDG);
    }
    for (; t < tiles; t += gridDim.x) {
        __syncthreads();                                  // tile `cur` staged; the other buffer is free again
        float* G = buf + cur * TILE_F;
        float* X = G + 32 * LDG;
        const bool more = t + gridDim.x < tiles;
        Staged nxt;
        if (more) nxt = stage_load(t + gridDim.x);        // in flight during the MFMAs below
        if (dw_role) {
            // dW[mat][32*iblk + i'][32*jb + j'] += sum_rows X[row][mat*H + 32*iblk + i'] * Gm[row][col(jb) + j']
            const float* xa = X + hi * 16 * LDX + mat * H + 32 * iblk + i;
            const float* gb = G + hi * 16 * LDG + i;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = xa[s * LDX];
                const float* gs = gb + s * LDG;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[0], acc[0], 0, 0, 0);            // r, cols 0-31
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[32], acc[1], 0, 0, 0);           // r, cols 32-63
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[H], acc[2], 0, 0, 0);            // z
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[H + 32], acc[3], 0, 0, 0);
                const int noff = mat == 0 ? 2 * H : 3 * H;                                            // dan | dnh
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[noff], acc[4], 0, 0, 0);         // n
                acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[noff + 32], acc[5], 0, 0, 0);
            }
        } else {
            // dx tile (32 atoms x 32 columns): sum over k = (gate, column) of Gsel[row][k] * W[n][k]
            f32x16 d;
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] = 0.f;
            const float* ga = G + i * LDG + hi * (H / 2);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int seg = (g == 2 && which == 1) ? 3 : g;     // dh's n-gate operand is dnh
#pragma unroll
                for (int q = 0; q < H / 8; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ga + seg * H + 4 * q);
                    const int f = (g * (H / 8) + q) * 4;
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[f / 16][f % 16 + 0], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[f / 16][f % 16 + 1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc[f / 16][f % 16 + 2], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc[f / 16][f % 16 + 3], d, 0, 0, 0);
                }
            }
            float* outp = which == 0 ? dm : dh;
            const int col = 32 * nb + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rl = acc_row(q, lane);
                const int64_t row = t * 32 + rl;
                float v = d[q];
                if (which == 1) v += G[rl * LDG + 4 * H + col];     // + g*z
                if (row < V) outp[row * H + col] = v;
            }
            // bias gradients: column sums of dar | daz | dan | dnh
            const int c = tid - 256;
            float part = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) part += G[rr * LDG + c];
            colsum += part;
        }
        if (more) {
            float* Gn = buf + (cur ^ 1) * TILE_F;
            stage_write(nxt, Gn, Gn + 32 * LDG);
        }
        cur ^= 1;
    }
    if (dw_role) {
        float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = 32 * j + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * iblk + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, acc[j][q]);
            }
        }
    } else if (blockIdx.x < tiles) {
        const int c = tid - 256;
        const int seg = c / H, cc = c % H;                // 0 dar, 1 daz, 2 dan, 3 dnh
        if (seg < 2) {
            atomicAdd(db_ih + seg * H + cc, colsum);
            atomicAdd(db_hh + seg * H + cc, colsum);
        } else if (seg == 2) {
            atomicAdd(db_ih + 2 * H + cc, colsum);
        } else {
            atomicAdd(db_hh + 2 * H + cc, colsum);
        }
    }
}

int launch_gru_bwd_f16_64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                          const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                          float* db_ih, float* db_hh, int64_t V, hipStream_t s);       // gru_bwd_f16.hip

int launch_gru_bwd_fused64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                           const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                           float* db_ih, float* db_hh, int64_t V, hipStream_t s) {
    constexpr int H = 64;
    const int64_t tiles = (V + 31) / 32;
    const size_t lds = (size_t)2 * 32 * (5 * H + 4 + 2 * H) * sizeof(float);
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_fused_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_bwd_fused_kernel<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t blocks = 256;                                  // one 8-wave block per CU (116 KB of LDS)
    if (blocks > tiles) blocks = tiles;
    // default: gate gradients, m | h and weights as two fp16 pieces, split once at staging (gru_bwd_f16.hip);
    // MPNN_GRU_MATH=fp32: the kernel of this file on the fp32 matrix pipe
    if (!switches().math_fp32)
        return launch_gru_bwd_f16_64(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, s);
    if (mask)
        hipLaunchKernelGGL((gru_bwd_fused_kernel<H, true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    else
        hipLaunchKernelGGL((gru_bwd_fused_kernel<H, false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(fused)");
}

}  // namespace mpnn
