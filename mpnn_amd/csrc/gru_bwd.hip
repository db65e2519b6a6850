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
// One fused kernel; the (V, 6H) pre-activation gradients are never materialised in HBM.  Three
// arrangements of the same structure live here (all parity-tested):
//   gru_bwd_fused_split_kernel   default: dW waves + dx waves, bf16x6 GEMMs            (2.5 ms on c2)
//   gru_bwd_uniform_kernel       MPNN_GRU_BWD_UNIFORM=1: all waves identical           (2.8 ms)
//   gru_bwd_fused_kernel         MPNN_GRU_MATH=fp32: fp32 MFMA throughout              (3.0 ms)
// History (measured, dropped): separate elementwise + 4 generic GEMM launches with a (V,6H) workspace
// 8.8 ms; a per-lane-row kernel computing gate gradients directly in fragment layout 5.3 ms (every
// 128-B line touched 16 B at a time, microseconds apart -> refetched from beyond L2).
#include <stdlib.h>
#include <string.h>

#include "split_math.h"

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

// ------------------------------------------------------------------------------------------ fused, bf16x6
// Same structure as gru_bwd_fused_kernel (shared double-buffered fp32 LDS tile, dW waves + dx waves)
// with every GEMM on the bf16 pipe through 3-way operand splitting (split_math.h):
//   dx waves  B operand = their slice of W_ih / W_hh, split ONCE into 36 bf16x8 register fragments;
//             A operand = 8 consecutive k of one LDS row (two ds_read_b128), split before use.
//   dW waves  both operands are columns of the LDS tile (the contraction runs over atom rows):
//             8 ds_read_b32 down a column per fragment, split before use.
// 72 bf16 MFMAs (2,304 pipe cycles) per wave per tile instead of 96 fp32 MFMAs (6,144).
template <int H, bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_bwd_fused_split_kernel(
    const float* __restrict__ dout, const float* __restrict__ m, const float* __restrict__ h,
    const float* __restrict__ mask, const float* __restrict__ W_ih, const float* __restrict__ W_hh,
    const float* __restrict__ saved, float* __restrict__ dm, float* __restrict__ dh, float* dW_ih, float* dW_hh,
    float* db_ih, float* db_hh, int64_t V) {
    static_assert(H == 64, "role split below is laid out for H = 64");
    constexpr int LDG = 5 * H + 4;
    constexpr int LDX = 2 * H;
    constexpr int TILE_F = 32 * (LDG + LDX);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* buf = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: role branches become s_cbranch
    const int i = lane & 31, hi = lane >> 5;
    // staging is done by the dW waves only (threads 0..255, two rows each): the dx waves spend their
    // registers on the resident weight fragments instead
    const int srow = (tid & 255) >> 4, sc4 = (tid & 15) * 4;
    const bool dw_role = wv < 4;
    const int64_t tiles = (V + 31) / 32;
    const int mat = (wv >> 1) & 1, iblk = wv & 1;
    const int which = ((wv - 4) >> 1) & 1, nb = (wv - 4) & 1;

    // The two roles run SEPARATE tile loops (same number of block barriers in each): inside one shared loop every
    // register of one role is live through the other role's branch -- 144 weight registers through the dW code, the
    // 58 staging registers through the dx code -- and the allocator spills; a spill reload is a scratch access, which
    // shares vmcnt with the tile loads in flight and drains them before the first MFMA.
    struct Staged { f32x4 v_do, vh, vm, v_r, v_z, v_n, v_nh; float mk; bool ok; };
    auto stage_load = [&](int64_t t, int half) {
        Staged q;
        int64_t row = t * 32 + srow + 16 * half;
        q.ok = row < V;
        if (!q.ok) row = V - 1;
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
    auto stage_write = [&](const Staged& q, int half, float* G, float* X) {
        f32x4 dar, daz, dan, dnh;
        const float mk = q.ok ? q.mk : 0.0f;
        gate_grads4(q.v_do, q.vh, q.v_r, q.v_z, q.v_n, q.v_nh, mk, dar, daz, dan, dnh);
        const f32x4 gz = q.v_do * mk * q.v_z;
        const float live = q.ok ? 1.0f : 0.0f;
        const int lrow = srow + 16 * half;
        float* g = G + lrow * LDG + sc4;
        *reinterpret_cast<f32x4*>(g) = dar;
        *reinterpret_cast<f32x4*>(g + H) = daz;
        *reinterpret_cast<f32x4*>(g + 2 * H) = dan;
        *reinterpret_cast<f32x4*>(g + 3 * H) = dnh;
        *reinterpret_cast<f32x4*>(g + 4 * H) = gz;
        float* x = X + lrow * LDX + sc4;
        *reinterpret_cast<f32x4*>(x) = q.vm * live;
        *reinterpret_cast<f32x4*>(x + H) = q.vh * live;
    };
    // 8 consecutive rows of one LDS column (the K=16 fragment of a row-contraction), split in 3
    auto column_frag = [&](const float* base, int ld, bf16x8& ph, bf16x8& pm, bf16x8& pl) {
        const f32x4 x0 = {base[0], base[ld], base[2 * ld], base[3 * ld]};
        const f32x4 x1 = {base[4 * ld], base[5 * ld], base[6 * ld], base[7 * ld]};
        split8(x0, x1, ph, pm, pl);
    };

    const int64_t t0 = blockIdx.x, tstep = gridDim.x;
    if (dw_role) {
        // ---- dW waves: stage the next tile (registers), accumulate six 32x32 blocks of dW_ih | dW_hh ----
        f32x16 R[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) R[j][q] = 0.f;
        if (t0 < tiles) {
            const Staged q0 = stage_load(t0, 0), q1 = stage_load(t0, 1);
            stage_write(q0, 0, buf, buf + 32 * LDG);
            stage_write(q1, 1, buf, buf + 32 * LDG);
        }
        // wave (mat, jg): BOTH 32-row blocks of X_mat against three of the six 32-column gate blocks, so a step
        // splits 2 + 3 column fragments (it was 1 + 6 when a wave owned one row block and all six columns)
        const int noff = mat == 0 ? 2 * H : 3 * H;        // W_ih's n-gate column uses dan, W_hh's uses dnh
        const int jg = iblk;
        const int c0 = jg == 0 ? 0 : H + 32, c1 = jg == 0 ? 32 : noff, c2 = jg == 0 ? H : noff + 32;
        int cur = 0;
        for (int64_t t = t0; t < tiles; t += tstep) {
            __syncthreads();
            float* G = buf + cur * TILE_F;
            float* X = G + 32 * LDG;
            const bool more = t + tstep < tiles;
            // Staging is unconditional (past the end the tile index is clamped and the staged tile is never read):
            // under `if (more)` the compiler sinks the loads into that branch, i.e. behind the MFMAs they should cover
            const int64_t tn = more ? t + tstep : t;
            const Staged nx0 = stage_load(tn, 0), nx1 = stage_load(tn, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int row0 = 16 * st + 8 * hi;        // this lane half's 8 rows of the K=16 step
                bf16x8 a0h, a0m, a0l, a1h, a1m, a1l, bh, bm, bl;
                column_frag(X + row0 * LDX + mat * H + i, LDX, a0h, a0m, a0l);
                column_frag(X + row0 * LDX + mat * H + 32 + i, LDX, a1h, a1m, a1l);
                const float* gcol = G + row0 * LDG + i;
                column_frag(gcol + c0, LDG, bh, bm, bl);
                mma6(R[0], a0h, a0m, a0l, bh, bm, bl);
                mma6(R[3], a1h, a1m, a1l, bh, bm, bl);
                column_frag(gcol + c1, LDG, bh, bm, bl);
                mma6(R[1], a0h, a0m, a0l, bh, bm, bl);
                mma6(R[4], a1h, a1m, a1l, bh, bm, bl);
                column_frag(gcol + c2, LDG, bh, bm, bl);
                mma6(R[2], a0h, a0m, a0l, bh, bm, bl);
                mma6(R[5], a1h, a1m, a1l, bh, bm, bl);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float* Gn = buf + (cur ^ 1) * TILE_F;
                stage_write(nx0, 0, Gn, Gn + 32 * LDG);
                stage_write(nx1, 1, Gn, Gn + 32 * LDG);
            }
            cur ^= 1;
        }
        float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = 32 * (3 * iblk + j % 3) + i;      // R[a*3 + b]: row block a, column block 3*jg + b
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (j / 3) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, R[j][q]);
            }
        }
        return;
    }

    // ---- dx waves: 36 bf16x8 fragments (3 gates x 4 K-steps x 3 pieces) of their weight slice stay in registers ----
    f32x16 R[9];
    auto wfrag = [&](int g, int st, int piece) {
        const int p = (g * 4 + st) * 3 + piece;           // 0..35, four fragments per f32x16
        const f32x4 v = {R[p >> 2][(p & 3) * 4 + 0], R[p >> 2][(p & 3) * 4 + 1], R[p >> 2][(p & 3) * 4 + 2],
                         R[p >> 2][(p & 3) * 4 + 3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    {
        const float* Wsrc = (which == 0 ? W_ih : W_hh) + (int64_t)(32 * nb + i) * 3 * H + hi * (H / 2);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wsrc + g * H + 8 * st);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wsrc + g * H + 8 * st + 4);
                bf16x8 ph, pm, pl;
                split8(w0, w1, ph, pm, pl);
                const bf16x8 pc[3] = {ph, pm, pl};
#pragma unroll
                for (int piece = 0; piece < 3; ++piece) {
                    const int p = (g * 4 + st) * 3 + piece;
                    const f32x4 v = __builtin_bit_cast(f32x4, pc[piece]);
                    R[p >> 2][(p & 3) * 4 + 0] = v.x;
                    R[p >> 2][(p & 3) * 4 + 1] = v.y;
                    R[p >> 2][(p & 3) * 4 + 2] = v.z;
                    R[p >> 2][(p & 3) * 4 + 3] = v.w;
                }
            }
    }
    float colsum = 0.f;
    float* outp = which == 0 ? dm : dh;
    const int col = 32 * nb + i;
    const unsigned lane_off = (unsigned)(4 * hi * H + col);   // acc_row(q, lane) = 4*hi + (q&3) + 8*(q>>2)
    int cur = 0;
    for (int64_t t = t0; t < tiles; t += tstep) {
        __syncthreads();
        const float* G = buf + cur * TILE_F;
        f32x16 d;
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.f;
        const float* ga = G + i * LDG + hi * (H / 2);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int seg = (g == 2 && which == 1) ? 3 : g;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ga + seg * H + 8 * st);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(ga + seg * H + 8 * st + 4);
                bf16x8 ah, am, al;
                split8(a0, a1, ah, am, al);
                mma6(d, ah, am, al, wfrag(g, st, 0), wfrag(g, st, 1), wfrag(g, st, 2));
            }
        }
        if (which == 1) {                                 // scalar branch; the 16 LDS reads go out together
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] += G[acc_row(q, lane) * LDG + 4 * H + col];
        }
        float* ob = outp + t * 32 * H + lane_off;         // scalar tile base + lane offset; rows are immediates
        if (t * 32 + 32 <= V) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ob[((q & 3) + 8 * (q >> 2)) * H] = d[q];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (t * 32 + acc_row(q, lane) < V) ob[((q & 3) + 8 * (q >> 2)) * H] = d[q];
        }
        const int c = tid - 256;
        float part = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) part += G[rr * LDG + c];
        colsum += part;
        cur ^= 1;
    }
    if (blockIdx.x < tiles) {
        const int c = tid - 256;
        const int seg = c / H, cc = c % H;
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

// ------------------------------------------------------------------------------------------ fused, uniform waves
// Third arrangement of the same fused backward (shared double-buffered fp32 LDS tile): all 8 waves run
// the SAME program, so register allocation is not the union of two roles and every wave carries the
// same load.  Per 32-atom tile each wave
//   * advances its 3 of the 24 (32x32) tiles of dW_ih|dW_hh on v_mfma_f32_32x32x2_f32 (exact fp32; both
//     operands are single ds_read_b32 of the LDS tile, no splitting), 48 MFMAs;
//   * computes one 16-column unit of dm|dh (32 atoms x 16 columns, K = 192) on the bf16 pipe with 3-way
//     operand splitting (v_mfma_f32_16x16x32_bf16, split_math.h): its 72-register slice of W_ih / W_hh is
//     split once and stays in registers, the A rows come from the LDS tile, 72 MFMAs of 16 cycles.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma6_16(f32x4& acc, const bf16x8& ah, const bf16x8& am, const bf16x8& al,
                                        const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

template <int H, bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_bwd_uniform_kernel(
    const float* __restrict__ dout, const float* __restrict__ m, const float* __restrict__ h,
    const float* __restrict__ mask, const float* __restrict__ W_ih, const float* __restrict__ W_hh,
    const float* __restrict__ saved, float* __restrict__ dm, float* __restrict__ dh, float* dW_ih, float* dW_hh,
    float* db_ih, float* db_hh, int64_t V) {
    static_assert(H == 64, "tile ownership below is laid out for H = 64");
    constexpr int LDG = 5 * H + 4;        // dar | daz | dan | dnh | g*z
    constexpr int LDX = 2 * H;            // m | h
    constexpr int TILE_F = 32 * (LDG + LDX);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* buf = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;      // staging: (row, 4 columns)
    const int64_t tiles = (V + 31) / 32;

    // dW ownership (32x32x2 f32): matrix, 32-row block of dW, three 32-column blocks
    const int mat = wv >> 2, iblk = (wv >> 1) & 1, jbase = (wv & 1) * 3;
    const int i32 = lane & 31, hi = lane >> 5;
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    int goff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int jb = jbase + j;                          // 0..5: gate = jb/2, column half = jb%2
        goff[j] = (jb >> 1) * H + (jb & 1) * 32 + ((mat == 1 && (jb >> 1) == 2) ? H : 0);   // W_hh's n column: dnh
    }

    // dx ownership (16x16x32 bf16x6): product (0: dm via W_ih, 1: dh via W_hh), 16-column block
    const int which = wv >> 2, cb = wv & 3;
    const int r16 = lane & 15, kq = lane >> 4;
    bf16x8 wh[6], wm[6], wl[6];                            // B operand: W[16cb + r16][k = 32s + 8kq + j]
    {
        const float* Wsrc = (which == 0 ? W_ih : W_hh) + (int64_t)(16 * cb + r16) * 3 * H + 8 * kq;
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wsrc + 32 * s2);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wsrc + 32 * s2 + 4);
            split8(w0, w1, wh[s2], wm[s2], wl[s2]);
        }
    }
    float colsum = 0.f;                                    // threads 0..255: column tid of dar|daz|dan|dnh

    struct Staged { f32x4 v_do, vh, vm, v_r, v_z, v_n, v_nh; float mk; bool ok; };
    auto stage_load = [&](int64_t t) {
        Staged q;
        int64_t row = t * 32 + srow;
        q.ok = row < V;
        if (!q.ok) row = V - 1;
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
        const float mk = q.ok ? q.mk : 0.0f;
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
        __syncthreads();
        float* G = buf + cur * TILE_F;
        float* X = G + 32 * LDG;
        const bool more = t + gridDim.x < tiles;
        Staged nxt;
        if (more) nxt = stage_load(t + gridDim.x);         // in flight under the matrix work below

        // ---- dW: three fp32 accumulators, contraction over the tile's 32 atoms ----
        {
            const float* xa = X + hi * 16 * LDX + mat * H + 32 * iblk + i32;
            const float* gb = G + hi * 16 * LDG + i32;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = xa[s * LDX];
                const float* gs = gb + s * LDG;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[goff[0]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[goff[1]], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gs[goff[2]], acc[2], 0, 0, 0);
            }
        }
        // ---- dx: one 16-column unit of dm | dh for both 16-atom halves of the tile ----
        {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
            for (int s2 = 0; s2 < 6; ++s2) {
                const int g = s2 >> 1;                     // gate of this K=32 step
                const int seg = (g == 2 && which == 1) ? 3 : g;
                const float* ga = G + r16 * LDG + seg * H + (s2 & 1) * 32 + 8 * kq;
                bf16x8 ah, am, al;
                split8(*reinterpret_cast<const f32x4*>(ga), *reinterpret_cast<const f32x4*>(ga + 4), ah, am, al);
                mma6_16(d0, ah, am, al, wh[s2], wm[s2], wl[s2]);
                const float* gb2 = ga + 16 * LDG;
                split8(*reinterpret_cast<const f32x4*>(gb2), *reinterpret_cast<const f32x4*>(gb2 + 4), ah, am, al);
                mma6_16(d1, ah, am, al, wh[s2], wm[s2], wl[s2]);
            }
            // C/D of 16x16: col = lane&15, row = (lane>>4)*4 + reg
            float* outp = which == 0 ? dm : dh;
            const int col = 16 * cb + r16;
            if (which == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d0[q] += G[(4 * kq + q) * LDG + 4 * H + col];
                    d1[q] += G[(16 + 4 * kq + q) * LDG + 4 * H + col];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row0 = t * 32 + 4 * kq + q, row1 = row0 + 16;
                if (row0 < V) outp[row0 * H + col] = d0[q];
                if (row1 < V) outp[row1 * H + col] = d1[q];
            }
        }
        if (tid < 4 * H) {
            float part = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) part += G[rr * LDG + tid];
            colsum += part;
        }
        if (more) {
            float* Gn = buf + (cur ^ 1) * TILE_F;
            stage_write(nxt, Gn, Gn + 32 * LDG);
        }
        cur ^= 1;
    }
    float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = 32 * (jbase + j) + i32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 32 * iblk + acc_row(q, lane);
            atomicAdd(dW + (int64_t)row * 3 * H + col, acc[j][q]);
        }
    }
    if (tid < 4 * H && blockIdx.x < tiles) {
        const int seg = tid / H, cc = tid % H;            // 0 dar, 1 daz, 2 dan, 3 dnh
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

int launch_gru_bwd_presplit64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                              const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                              float* db_ih, float* db_hh, int64_t V, hipStream_t s);   // gru_bwd_presplit.hip
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
    const bool fp32_only = switches().math_fp32;
    const bool uniform = switches().gru_bwd_uniform;  // A/B: all-waves-identical arrangement
    if (!fp32_only && uniform) {
        static const hipError_t attr3 = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_uniform_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_bwd_uniform_kernel<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr3 != hipSuccess) return lds_opt_in_failed(attr3);
        if (mask)
            hipLaunchKernelGGL((gru_bwd_uniform_kernel<H, true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h,
                               mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
        else
            hipLaunchKernelGGL((gru_bwd_uniform_kernel<H, false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h,
                               mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
        return launch_status("mpnn_gru_update_bwd_f32(uniform)");
    }
    // default: gate gradients, m | h and weights as two fp16 pieces, split once at staging (gru_bwd_f16.hip);
    // MPNN_GRU_BWD_BF16=1: three bf16 pieces (gru_bwd_presplit.hip); MPNN_GRU_BWD_FP32TILE=1 keeps the fp32 tile whose
    // consumers split what they read
    const bool fp32_tile = switches().gru_bwd_fp32tile;
    if (!fp32_only && !fp32_tile && !switches().gru_bwd_bf16)
        return launch_gru_bwd_f16_64(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, s);
    if (!fp32_only && !fp32_tile)
        return launch_gru_bwd_presplit64(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, s);
    if (!fp32_only) {
        static const hipError_t attr2 = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_fused_split_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_bwd_fused_split_kernel<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr2 != hipSuccess) return lds_opt_in_failed(attr2);
        if (mask)
            hipLaunchKernelGGL((gru_bwd_fused_split_kernel<H, true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m,
                               h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
        else
            hipLaunchKernelGGL((gru_bwd_fused_split_kernel<H, false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m,
                               h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
        return launch_status("mpnn_gru_update_bwd_f32(fused bf16x6)");
    }
    if (mask)
        hipLaunchKernelGGL((gru_bwd_fused_kernel<H, true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    else
        hipLaunchKernelGGL((gru_bwd_fused_kernel<H, false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(fused)");
}

}  // namespace mpnn
