// GRU backward at H = 64 on two fp16 pieces per operand ("fp16x3": three MFMAs per product instead of the six of
// bf16x6).  Reference: mpnn_functions/update/gru_update.py:26-35 (autograd of it).
//
// Block structure of gru_bwd_presplit.hip: a double-buffered 32-atom LDS tile, waves 0-3 stage the next tile and own
// the dW_ih | dW_hh accumulators, waves 4-7 compute dm / dh with their weight slice resident in registers.  What
// changes with fp16 pieces:
//   * fp16 has 5 exponent bits, so every operand is range-guarded by an exact power-of-two scale: the gate gradients
//     of a ROW (atom) by sg_row (the row's largest magnitude lands in [2^14, 2^15)), a dx wave's weight slice by sw, the
//     row's m | h by sx_row.  x*s = hi + lo, hi = fp16(x*s), lo = fp16(x*s - hi); entries more than 2^16 below their
//     ROW's largest lose relative (not absolute) accuracy.  (Round 2 had one sg per tile: an atom whose gradients lay
//     1e6 below a tile-mate's kept ~15 bits in its dm / dh rows.)
//   * C (below) needs the tile's largest gate gradient before the first split, i.e. a maximum over what four waves
//     stage: one more block barrier per tile (the dx waves wait at that point anyway).
//   * dm / dh rows are per atom: the epilogue multiplies row by row by 1 / (sg_row * sw).  dW accumulates over all tiles
//     of a block in registers; for that sum sg_row * sx_row must be the same for every row of every tile.  The block
//     keeps C = min over the tiles so far of sg_tile * sx_opt (sg_tile = the scale of the tile's largest row, its
//     smallest) and scales a row's m | h by sx_row = C / sg_row <= sx_opt (never overflows; a tile whose gradients
//     are small next to earlier ones gets m | h pieces below their best precision, by exactly the factor its
//     contribution is small).  When C drops the accumulators are multiplied by the ratio (a power of two: exact).
//   * m | h are parked as pieces too (same [32][128 x 16 bit] image format as the gate pairs, same LDS footprint as
//     fp32 rows), so the dW waves fetch BOTH operands with transposed reads and do no vector work in their MFMA phase.
// LDS per tile: 2 pieces x 3 pair images (dar|daz, dan|dnh, m|h) x 8 KB + g*z in fp32 = 57,856 B; two tiles per block.
// Measured and dropped: all eight waves staging one row per thread (1.66 ms against 1.50 ms for this arrangement).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace mpnn {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int FH = 64;
constexpr int F_IMG = 32 * 256;                        // bytes of one [32 rows][128 x fp16] image
constexpr int F_P = 6 * F_IMG;                         // image (piece, pair) at (3 * piece + pair) * F_IMG
constexpr int F_LDZ = 68;
constexpr int F_GZ = 32 * F_LDZ * 4;
constexpr int F_TILE = F_P + F_GZ;                     // 57,856 bytes; two of them per block
constexpr int F_RED = 2 * F_TILE;                      // float red[8]: per staging wave max |gate gradient|, max |m|h|
constexpr int F_SCL = F_RED + 32;                      // float inv_sg[2][32]: per tile buffer, per row
constexpr int F_LDS = F_SCL + 256;

// byte offset of 16-byte chunk `ch` (8 columns) of row `row` inside an image (the swizzle of gru_bwd_presplit.hip)
__device__ __forceinline__ int f_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ void f_gate_grads4(const f32x4& dout, const f32x4& hv, const f32x4& r, const f32x4& z,
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

__device__ __forceinline__ h16x8 f_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(h16x8, v);
}

// power of two s with maxabs * s in [2^14, 2^15), and 1 / s.  s is clamped to [2^-46, 2^SMAX]: a smaller maximum keeps
// s = 2^SMAX (its pieces lose precision, not range).  SMAX = 90 for gate gradients (full precision down to maxima of
// 1e-23), 30 for weights and m | h, so that every product of two scales and its inverse stay normal floats.
template <int SMAX>
__device__ __forceinline__ void guard_scale(float maxabs, float& s, float& inv) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    e = e < 141 - SMAX ? 141 - SMAX : (e > 187 ? 187 : e);
    s = __int_as_float((268 - e) << 23);
    inv = __int_as_float((e - 14) << 23);
}

// maximum over the 16 lanes of a DPP row (every lane gets it): two quad permutes, then rotations by 4 and 8 within the row --
// no trip through the LDS crossbar (a __shfl_xor is a ds_bpermute), which the dx waves keep busy
__device__ __forceinline__ float row16_max(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true);         // row_ror:4
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true);         // row_ror:8
    return fmaxf(v, __int_as_float(x));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// three partial products of one K=16 step for two row blocks that share the B pieces, small terms first
__device__ __forceinline__ void mma3x2_b(f32x16& c0, f32x16& c1, const h16x8& a0h, const h16x8& a0l, const h16x8& a1h,
                                         const h16x8& a1l, const h16x8& bh, const h16x8& bl) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, c1, 0, 0, 0);
}
}  // namespace

template <bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_bwd_f16_kernel(
    const float* __restrict__ dout, const float* __restrict__ m, const float* __restrict__ h,
    const float* __restrict__ mask, const float* __restrict__ W_ih, const float* __restrict__ W_hh,
    const float* __restrict__ saved, float* __restrict__ dm, float* __restrict__ dh, float* dW_ih, float* dW_hh,
    float* db_ih, float* db_hh, int64_t V) {
    constexpr int H = FH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + F_RED);
    float* scl = reinterpret_cast<float*>(smem + F_SCL);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: role branches become s_cbranch
    const int i = lane & 31, hi = lane >> 5;
    const int64_t tiles = (V + 31) / 32;
    const int64_t t0 = blockIdx.x, tstep = gridDim.x;            // the launch keeps gridDim.x <= tiles

    if (wv < 4) {
        // ------------------------------------------------------------------ dW waves (+ staging of the next tile)
        const int srow = tid >> 4, sc4 = (tid & 15) * 4;  // two rows (srow, srow + 16), four columns of every segment
        const int mat = (wv >> 1) & 1, jg = wv & 1;

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
        float cs[16];                                     // column sums of what this thread stages: [segment][column]
#pragma unroll
        for (int k = 0; k < 16; ++k) cs[k] = 0.f;
        struct Grads { f32x4 s[6]; f32x4 gz; };           // dar, daz, dan, dnh, m, h of one row's four columns
        // rmx: largest gate-gradient magnitude of THIS row (the row's 16 staging lanes are neighbours in the wave): the gate
        // pieces carry one scale per row -- a row scale factors out of dm | dh, and in dW it is folded into the row's m | h
        // scale (sx_row = C / sg_row) -- so an atom keeps its 22 bits whatever its tile-mates' magnitudes are
        auto grads_of = [&](const Staged& q, float count, float& gmx, float& xmx, float& rmx) {
            Grads G;
            rmx = 0.f;
            const float mk = q.ok ? q.mk : 0.0f;          // rows past V contribute exact zeros
            f_gate_grads4(q.v_do, q.vh, q.v_r, q.v_z, q.v_n, q.v_nh, mk, G.s[0], G.s[1], G.s[2], G.s[3]);
            G.gz = q.v_do * mk * q.v_z;
            const float live = q.ok ? 1.0f : 0.0f;
            G.s[4] = q.vm * live;
            G.s[5] = q.vh * live;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[4 * s + j] = fmaf(G.s[s][j], count, cs[4 * s + j]);
                    rmx = fmaxf(rmx, fabsf(G.s[s][j]));
                }
            rmx = row16_max(rmx);                          // the row's 16 staging lanes are one DPP row
            gmx = fmaxf(gmx, rmx);
#pragma unroll
            for (int j = 0; j < 4; ++j) xmx = fmaxf(xmx, fmaxf(fabsf(G.s[4][j]), fabsf(G.s[5][j])));
            return G;
        };
        auto park = [&](const Grads& G, int half, char* T, float sg, float sx) {
            const int row = srow + 16 * half;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float sc = s < 4 ? sg : sx;
                h16x4 ph, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = G.s[s][j] * sc;
                    ph[j] = (_Float16)a;
                    pl[j] = (_Float16)(a - (float)ph[j]);
                }
                char* a = T + (s >> 1) * F_IMG + f_off(row, (s & 1) * 8 + (sc4 >> 3)) + (sc4 & 4) * 2;
                *reinterpret_cast<h16x4*>(a) = ph;
                *reinterpret_cast<h16x4*>(a + 3 * F_IMG) = pl;
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(T + F_P) + row * F_LDZ + sc4) = G.gz;
        };
        auto publish = [&](float gmx, float xmx) {
            gmx = wave_max(gmx);
            xmx = wave_max(xmx);
            if (lane == 0) {
                red[wv] = gmx;
                red[4 + wv] = xmx;
            }
        };
        float C_run = 3.0e38f;                            // min over staged tiles of sg * (best m|h scale)
        // after the barrier that follows publish(): C_run is updated from the tile's largest row (its scale is the smallest
        // of the tile, so C <= sg_row * sx_opt holds for every row); then a row's scales from its own maximum
        auto tile_scales = [&]() {
            const float gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const float xm = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
            float sgt, inv_sgt, sxo, inv_sxo;
            guard_scale<90>(gm, sgt, inv_sgt);
            guard_scale<30>(xm, sxo, inv_sxo);
            C_run = fminf(C_run, sgt * sxo);
        };
        auto row_scales = [&](float rmx, float& sg, float& sx, float& inv_sg) {
            guard_scale<90>(rmx, sg, inv_sg);
            sx = C_run * inv_sg;
        };

        // wave (mat, jg): BOTH 32-row blocks of X_mat against three of the six 32-column gate blocks.
        // block b of this wave = (segment, half): jg 0 -> (dar,0) (dar,1) (daz,0); jg 1 -> (daz,1) (n,0) (n,1) with
        // n = dan for W_ih, dnh for W_hh; pair = segment >> 1, cb = 32-column block inside the 128-column image
        const int ns = 2 + mat;
        const int bseg[3] = {jg == 0 ? 0 : 1, jg == 0 ? 0 : ns, jg == 0 ? 1 : ns};
        const int bhalf[3] = {jg == 0 ? 0 : 1, jg == 0 ? 1 : 0, jg == 0 ? 0 : 1};
        // transposed-read addressing: in its 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3 of the 4 x 16
        // block; groups 0/1 take columns 0-15 / 16-31, groups 2/3 the same columns 8 rows further (K half of the lane).
        const int g2 = lane >> 4, u = lane & 15, q4 = u >> 2, p4 = u & 3;
        auto tr_addr = [&](int pair, int cb, int j) {
            return pair * F_IMG + f_off(8 * (g2 >> 1) + 4 * j + q4, cb * 4 + 2 * (g2 & 1) + (p4 >> 1)) + 8 * (p4 & 1);
        };
        int LA[3][2], LX[2][2];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) LA[b][j] = tr_addr(bseg[b] >> 1, (bseg[b] & 1) * 2 + bhalf[b], j);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) LX[a][j] = tr_addr(2, mat * 2 + a, j);

        f32x16 R[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) R[j][q] = 0.f;
        {
            const Staged q0 = stage_load(t0, 0), q1 = stage_load(t0, 1);
            float gmx = 0.f, xmx = 0.f, r0, r1;
            const Grads G0 = grads_of(q0, 1.0f, gmx, xmx, r0), G1 = grads_of(q1, 1.0f, gmx, xmx, r1);
            publish(gmx, xmx);
            __syncthreads();
            tile_scales();
            float sg, sx, inv_sg;
            row_scales(r0, sg, sx, inv_sg);
            park(G0, 0, smem, sg, sx);
            if ((tid & 15) == 0) scl[srow] = inv_sg;
            row_scales(r1, sg, sx, inv_sg);
            park(G1, 1, smem, sg, sx);
            if ((tid & 15) == 0) scl[srow + 16] = inv_sg;
        }
        float C_acc = C_run;                              // sg * sx of everything summed into R so far
        float C_cur = C_run;                              // ... of the tile in buffer `cur`
        int cur = 0;
        for (int64_t t = t0; t < tiles; t += tstep) {
            __syncthreads();
            const char* T = smem + cur * F_TILE;
            // Staging is unconditional (past the end the tile index is clamped and the staged tile is never read):
            // under `if (more)` the compiler sinks the loads into that branch, i.e. behind the MFMAs they should cover
            const bool more = t + tstep < tiles;
            const int64_t tn = more ? t + tstep : t;
            const Staged nx0 = stage_load(tn, 0), nx1 = stage_load(tn, 1);
            if (__builtin_amdgcn_readfirstlane(__float_as_int(C_cur)) != __builtin_amdgcn_readfirstlane(__float_as_int(C_acc))) {
                const float ratio = C_cur / C_acc;        // < 1, a power of two
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) R[j][q] *= ratio;
                C_acc = C_cur;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char* Tb = T + 256 * 16 * st;       // rows +16: both swizzle terms unchanged
                const h16x8 a0h = f_tr8(Tb + LX[0][0], Tb + LX[0][1]);
                const h16x8 a0l = f_tr8(Tb + 3 * F_IMG + LX[0][0], Tb + 3 * F_IMG + LX[0][1]);
                const h16x8 a1h = f_tr8(Tb + LX[1][0], Tb + LX[1][1]);
                const h16x8 a1l = f_tr8(Tb + 3 * F_IMG + LX[1][0], Tb + 3 * F_IMG + LX[1][1]);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const h16x8 bh = f_tr8(Tb + LA[b][0], Tb + LA[b][1]);
                    const h16x8 bl = f_tr8(Tb + 3 * F_IMG + LA[b][0], Tb + 3 * F_IMG + LA[b][1]);
                    mma3x2_b(R[b], R[3 + b], a0h, a0l, a1h, a1l, bh, bl);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the vector work of the next tile runs beside the SIMD partner's (a dx wave's) 36 MFMAs: at equal priority
            // the partner wins the issue port and this stream crawls (measured on the wide kernels, gru_bwd_rc.hip)
            asm volatile("s_setprio 2" ::: "memory");
            {
                float gmx = 0.f, xmx = 0.f, r0, r1;
                const float count = more ? 1.0f : 0.0f;   // the clamped re-stage counts nothing
                const Grads G0 = grads_of(nx0, count, gmx, xmx, r0), G1 = grads_of(nx1, count, gmx, xmx, r1);
                publish(gmx, xmx);
                __syncthreads();
                tile_scales();
                char* Tn = smem + (cur ^ 1) * F_TILE;
                float sg, sx, inv_sg;
                row_scales(r0, sg, sx, inv_sg);
                park(G0, 0, Tn, sg, sx);
                if ((tid & 15) == 0) scl[(cur ^ 1) * 32 + srow] = inv_sg;
                row_scales(r1, sg, sx, inv_sg);
                park(G1, 1, Tn, sg, sx);
                if ((tid & 15) == 0) scl[(cur ^ 1) * 32 + srow + 16] = inv_sg;
                C_cur = C_run;
            }
            asm volatile("s_setprio 0" ::: "memory");
            cur ^= 1;
        }
        const float inv_C = 1.0f / C_acc;
        float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = 32 * (3 * jg + j % 3) + i;    // R[a*3 + b]: row block a, column block 3*jg + b
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (j / 3) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, R[j][q] * inv_C);
            }
        }
        // bias gradients: the 16 threads that stage the same columns sit 16 lanes apart in each of the four waves
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = cs[k];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            cs[k] = v;
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int s = k >> 2, cc = sc4 + (k & 3);     // segment 0 dar, 1 daz, 2 dan, 3 dnh
                if (s < 2) {
                    atomicAdd(db_ih + s * H + cc, cs[k]);
                    atomicAdd(db_hh + s * H + cc, cs[k]);
                } else if (s == 2) {
                    atomicAdd(db_ih + 2 * H + cc, cs[k]);
                } else {
                    atomicAdd(db_hh + 2 * H + cc, cs[k]);
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- dx waves
    // 24 fp16x8 fragments (3 gates x 4 K-steps x 2 pieces) of the wave's weight slice stay in registers, one scale
    const int which = ((wv - 4) >> 1) & 1, nb = (wv - 4) & 1;
    h16x8 Wh[12], Wl[12];
    float inv_sw;
    {
        const float* Wsrc = (which == 0 ? W_ih : W_hh) + (int64_t)(32 * nb + i) * 3 * H + hi * (H / 2);
        f32x4 w[24];
        float mx = 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                w[2 * (g * 4 + st)] = *reinterpret_cast<const f32x4*>(Wsrc + g * H + 8 * st);
                w[2 * (g * 4 + st) + 1] = *reinterpret_cast<const f32x4*>(Wsrc + g * H + 8 * st + 4);
            }
#pragma unroll
        for (int k = 0; k < 24; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(w[k][j]));
        float sw;
        guard_scale<30>(wave_max(mx), sw, inv_sw);
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = w[2 * p][j] * sw, b = w[2 * p + 1][j] * sw;
                Wh[p][j] = (_Float16)a;
                Wl[p][j] = (_Float16)(a - (float)Wh[p][j]);
                Wh[p][4 + j] = (_Float16)b;
                Wl[p][4 + j] = (_Float16)(b - (float)Wh[p][4 + j]);
            }
    }
    __syncthreads();                                      // pairs with the staging waves' first publish
    // row reads: lane (i, hi) takes columns hi*32 + 8*st .. +7 of gate segment g of row i (the K order of Wh / Wl);
    // g = 2 is dan (dm waves) or dnh (dh waves), the two halves of column pair 1
    int DA[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int e = g == 2 ? which : g;                 // 64-column half of the pair image
#pragma unroll
        for (int st = 0; st < 4; ++st) DA[g][st] = (g >> 1) * F_IMG + f_off(i, e * 8 + hi * 4 + st);
    }
    float* outp = which == 0 ? dm : dh;
    const int col = 32 * nb + i;
    const unsigned lane_off = (unsigned)(4 * hi * H + col);   // acc_row(q, lane) = 4*hi + (q&3) + 8*(q>>2)
    int cur = 0;
    for (int64_t t = t0; t < tiles; t += tstep) {
        __syncthreads();
        const char* T = smem + cur * F_TILE;
        const float* rs = scl + cur * 32;                  // 1 / (scale of the tile's row), per row
        // two accumulators (even / odd K steps) issued alternately
        f32x16 d, d1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { d[q] = 0.f; d1[q] = 0.f; }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int st = 0; st < 4; st += 2) {
                const char* a = T + DA[g][st];
                const char* b = T + DA[g][st + 1];
                const h16x8 ah = *reinterpret_cast<const h16x8*>(a);
                const h16x8 al = *reinterpret_cast<const h16x8*>(a + 3 * F_IMG);
                const h16x8 bh = *reinterpret_cast<const h16x8*>(b);
                const h16x8 bl = *reinterpret_cast<const h16x8*>(b + 3 * F_IMG);
                const int p = g * 4 + st;
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Wh[p], d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, Wh[p + 1], d1, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Wl[p], d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Wl[p + 1], d1, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Wh[p], d, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Wh[p + 1], d1, 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = (d[q] + d1[q]) * (rs[acc_row(q, lane)] * inv_sw);
        if (which == 1) {                                 // scalar branch; the 16 LDS reads go out together
            const float* GZ = reinterpret_cast<const float*>(T + F_P);
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] += GZ[acc_row(q, lane) * F_LDZ + col];
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
        __syncthreads();                                  // pairs with the staging waves' publish of the next tile
        cur ^= 1;
    }
}

int launch_gru_bwd_f16_64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                          const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                          float* db_ih, float* db_hh, int64_t V, hipStream_t s) {
    const int64_t tiles = (V + 31) / 32;
    const size_t lds = (size_t)F_LDS;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_f16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_bwd_f16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t blocks = 256;                                  // one 8-wave block per CU (113 KB of LDS)
    if (blocks > tiles) blocks = tiles;
    if (blocks < 1) return MPNN_OK;
    if (mask)
        hipLaunchKernelGGL((gru_bwd_f16_kernel<true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask, W_ih,
                           W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    else
        hipLaunchKernelGGL((gru_bwd_f16_kernel<false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask, W_ih,
                           W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(fp16x3 tile)");
}

}  // namespace mpnn
