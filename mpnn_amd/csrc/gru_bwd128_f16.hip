// GRU backward at H = 128 on two fp16 pieces per operand ("fp16x3": three MFMAs per fp32 product), gate gradients
// split ONCE.  Reference: mpnn_functions/update/gru_update.py:26-35 (autograd of it).
//
// At this width neither matrix's split images fit next to a tile (gru_bwd128.hip), so the work stays three kernels --
// gate gradients, dm | dh, dW -- but what travels between them is no longer fp32: the gate-gradient kernel writes each
// 32-atom tile's (dar daz dan dnh) as fp16 PIECES x*sg = hi + lo behind one power-of-two scale sg per tile (largest
// magnitude of the tile in [2^14, 2^15)), as 1 KB blocks of 16 columns:
//     pieces[tile][kstep 0..31][piece hi|lo][row 0..31][16 halves] = columns 16*kstep ... of the tile's rows
// (kstep >> 3 = segment dar, daz, dan, dnh; an MFMA A fragment = 16 bytes at row*32 + 16*(lane >> 5)).  Same bytes as the fp32 workspace (4H floats per atom), and
//   * the dm | dh kernel loads its A operand with one fully coalesced 1 KB request per (kstep, piece) and feeds it to
//     the matrix pipe untouched (no splitting, 3 MFMAs per product instead of 6); weights: two fp16 images of the
//     block's 32-column slice in LDS (96 KB) behind one scale per block;
//   * the dW kernel parks the same fragments row-major in LDS ([32 atoms][128 columns] images per segment, the swizzle
//     of gru_bwd_f16.hip) and reads COLUMNS with ds_read_b64_tr_b16; m | h are split by the kernel itself behind
//     sx = C / sg, C = running minimum of sg * (best scale of the tile's m | h), so that every tile's products carry
//     the same factor C and the register accumulators never need a per-tile fold (gru_bwd_f16.hip has the argument);
//   * the bias gradients (column sums of the gate gradients) are taken by the gate-gradient kernel, in fp32.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace mpnn {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int GH = 128;
constexpr int G_IMG = 32 * 256;                        // bytes of one [32 rows][128 x fp16] image


template <int SMAX>
__device__ __forceinline__ void g_guard_scale(float maxabs, float& s, float& inv) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    e = e < 141 - SMAX ? 141 - SMAX : (e > 187 ? 187 : e);
    s = __int_as_float((268 - e) << 23);
    inv = __int_as_float((e - 14) << 23);
}

// maxima by DPP (row of 16 lanes: two quad permutes, rotations by 4 and 8) and readlane: a __shfl_xor is a ds_bpermute, a
// trip through the LDS crossbar that the MFMA operand reads keep busy
__device__ __forceinline__ float g_row16_max(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true);
    return fmaxf(v, __int_as_float(x));
}
__device__ __forceinline__ float g_wave_max(float v) {
    v = g_row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

__device__ __forceinline__ void g_split8(const f32x4& x0, const f32x4& x1, float sc, h16x8& ph, h16x8& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * sc, b = x1[j] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
        ph[4 + j] = (_Float16)b;
        pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
    }
}

// Block barrier that orders LDS traffic only.  __syncthreads() carries a workgroup fence over ALL address spaces, which
// makes the compiler drain every outstanding global_load_lds copy (vmcnt(0)) -- the copies issued two tiles ahead are
// waited for explicitly (s_waitcnt vmcnt(0), in body() where the buffers change hands) where their buffer is needed.
__device__ __forceinline__ void g_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One wave copies 1 KB of global memory (lane L's 16 bytes at `src`) to LDS bytes [lds_dst + 16 L, +16) without touching
// registers.  Issued as inline assembly on purpose: behind the builtin the compiler drains EVERY outstanding copy
// (s_waitcnt vmcnt(0)) in front of the next LDS read that might alias it, which would undo the two-tile prefetch; the
// waits that matter are written out where the buffers change hands.
__device__ __forceinline__ void g_copy_to_lds(const char* src, const char* lds_dst) {
    typedef __attribute__((address_space(3))) const char lds_char;
    const unsigned dst = (unsigned)(uintptr_t)(lds_char*)lds_dst;
    unsigned keep;                                         // m0 is a reserved register: put back what was there
    // s_nop 0: one wait state between the SALU write of M0 and the LDS-DMA that reads it (the compiler pads nothing
    // inside an asm statement)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}

__device__ __forceinline__ h16x8 g_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(h16x8, v);
}
}  // namespace

// ------------------------------------------------------------------------------------------------ gate gradients
// Block = one 32-atom tile per iteration; thread = (row tid >> 4, column groups c16 + 16 j (8 columns each) of every
// segment, j < H / 128).  The tile's values stay in registers between the maximum and the split.
// NORM (the backward of the masked batch norm that follows this update, fused): `dout` is then the
// gradient of the NORMALISED state hn' = norm(y), y = this update's own output, and the gradient of y is
//     dy = dout * k1[col] + y * k2[col] + k4[col]                       (rows with mask 1; the others get none)
// with column constants the caller derives from the norm's statistics and from the column sums of dout and dout * hn'
// (which the dm | dh kernel of the FOLLOWING update took in its epilogue).  y is not read: it is (1-z) n + z h of the saved
// gates and the h this kernel reads anyway.  kn = k1 | k2 | k4, 3 H floats.
template <int H, bool HAS_MASK, bool NORM = false>
__global__ void __launch_bounds__(512) gru_gate_f16_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                                           const float* __restrict__ mask, const float* __restrict__ saved,
                                                           char* __restrict__ pieces, float* __restrict__ inv_scale,
                                                           float* __restrict__ dh, float* db_ih, float* db_hh, int64_t V,
                                                           const float* __restrict__ kn) {
    constexpr int NG = H / 128;                            // column groups per thread
    constexpr int TILE_BYTES = 32 * 4 * H * 4;
    constexpr int SEG_BYTES = (H / 16) * 2048;             // ksteps of one segment
    __shared__ float bsum[8][16][32 * NG + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int srow = tid >> 4, c16 = tid & 15;
    const int64_t tiles = (V + 31) / 32;
    float cs[32 * NG];                                     // column sums: [group][segment][column]
#pragma unroll
    for (int k = 0; k < 32 * NG; ++k) cs[k] = 0.f;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        int64_t row = t * 32 + srow;
        const bool ok = row < V;
        if (!ok) row = V - 1;
        const float mk = ok ? (HAS_MASK ? mask[row] : 1.0f) : 0.0f;
        unsigned kofs = 0;
        if (NORM) asm volatile("" : "+v"(kofs));
        f32x4 seg[NG][4][2], gz[NG][2];
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int c8 = 8 * (c16 + 16 * j);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t at = row * H + c8 + 4 * q;
                const f32x4 v_do = *reinterpret_cast<const f32x4*>(dout + at);
                const f32x4 vh = *reinterpret_cast<const f32x4*>(h + at);
                const float* sv = saved + row * 4 * H + c8 + 4 * q;
                const f32x4 r = *reinterpret_cast<const f32x4*>(sv);
                const f32x4 z = *reinterpret_cast<const f32x4*>(sv + H);
                const f32x4 n = *reinterpret_cast<const f32x4*>(sv + 2 * H);
                const f32x4 nh = *reinterpret_cast<const f32x4*>(sv + 3 * H);
                f32x4 g = v_do * mk;                       // through the final "* mask"
                if (NORM) {
                    // (column constants re-read per tile behind an opaque offset: as loop invariants they would take
                    // 24 registers per column group and spill at H = 256)
                    const float* kp = kn + kofs + c8 + 4 * q;
                    const f32x4 k1 = *reinterpret_cast<const f32x4*>(kp);
                    const f32x4 k2 = *reinterpret_cast<const f32x4*>(kp + H);
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(kp + 2 * H);
                    const f32x4 y = ((1.0f - z) * n + z * vh) * mk;
                    g = (v_do * k1 + y * k2 + k4) * mk;
                }
                const f32x4 dn = g * (1.0f - z);
                const f32x4 dz = g * (vh - n);
                const f32x4 dan = dn * mk * (1.0f - n * n);    // n = tanh(.)*mask
                seg[j][0][q] = dan * nh * mk * r * (1.0f - r);
                seg[j][1][q] = dz * mk * z * (1.0f - z);
                seg[j][2][q] = dan;
                seg[j][3][q] = dan * r;
                gz[j][q] = g * z;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cs[32 * j + 8 * s + 4 * q + u] += seg[j][s][q][u];
                        mx = fmaxf(mx, fabsf(seg[j][s][q][u]));
                    }
        }
        // one power-of-two scale per ROW (a row = one atom's 4H gate gradients, held by 16 neighbouring lanes): a row scale
        // factors out of dm | dh, whose rows are atoms, and the dW kernel folds it into that atom's m | h row -- so an atom
        // keeps its 22 bits whatever its tile-mates' magnitudes are (one scale per tile left an atom 1e6 below its
        // neighbours with ~15 bits), and the tile needs no block-wide maximum
        mx = g_row16_max(mx);
        float sg, inv_sg;
        g_guard_scale<90>(mx, sg, inv_sg);
        if (c16 == 0) inv_scale[t * 32 + srow] = inv_sg;
        // (the row scale lives in a vector register where the tile scale was wave-uniform: 130-134 registers instead of 126
        // at H = 128, three waves per SIMD instead of four, ~5 % on this HBM-bound kernel; forcing four spills 12-48 bytes)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            // kstep inside the segment = (c16 + 16 j) >> 1; inside its 1 KB block: row srow, 16-byte half c16 & 1
            char* base = pieces + t * (int64_t)TILE_BYTES + ((c16 + 16 * j) >> 1) * 2048 + srow * 32 + (c16 & 1) * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                h16x8 ph, pl;
                g_split8(seg[j][s][0], seg[j][s][1], sg, ph, pl);
                *reinterpret_cast<h16x8*>(base + s * SEG_BYTES) = ph;
                *reinterpret_cast<h16x8*>(base + s * SEG_BYTES + 1024) = pl;
            }
            if (ok) {
                const int c8 = 8 * (c16 + 16 * j);
                *reinterpret_cast<f32x4*>(dh + row * H + c8) = gz[j][0];
                *reinterpret_cast<f32x4*>(dh + row * H + c8 + 4) = gz[j][1];
            }
        }
    }
    // bias gradients: rows of one column group sit 16 lanes apart in a wave; then across the eight waves through LDS
#pragma unroll
    for (int k = 0; k < 32 * NG; ++k) {
        float v = cs[k];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        cs[k] = v;
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 32 * NG; ++k) bsum[wv][lane][k] = cs[k];
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * 32 * NG; idx += 512) {
        const int cg = idx / (32 * NG), k = idx % (32 * NG);   // lane group, (column group j, segment, column)
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += bsum[w][cg][k];
        const int j = k >> 5, s = (k >> 3) & 3, col = 8 * (cg + 16 * j) + (k & 7);
        if (s < 2) {
            atomicAdd(db_ih + s * H + col, v);
            atomicAdd(db_hh + s * H + col, v);
        } else if (s == 2) {
            atomicAdd(db_ih + 2 * H + col, v);
        } else {
            atomicAdd(db_hh + 2 * H + col, v);
        }
    }
}

// ------------------------------------------------------------------ dm | dh, weights streamed, 128-column slices
// A block owns 128 output features of dm and of dh -- a wave 32 rows x 128 features of both, eight accumulators -- so the
// piece workspace is read H / 128 times (a 64-column variant read it twice as often: 80 GB on c5, 15.8 ms; removed in
// round 3 together with a resident 32-column-slice variant).  What pays for the registers: the contraction is
// cut into 32-wide chunks (two K = 16 steps; A operand 16 registers per set instead of 32), and the third gate block is
// walked twice, once as (dan, W_ih) -> dm and once as (dnh, W_hh) -> dh, so that no second operand set is ever live.
// ---- pre-split weights of the 128-column dm | dh kernel: one workspace region per launch ----
// [0, 64): inverse weight scale of slice s at float s; then the LDS image of (slice, chunk ct) at 64 + (slice * NCT + ct) * 32 KB,
// [matrix][piece][128 output rows][32 k] exactly as gru_bwd_dx_deep_f16_kernel reads it, so that a chunk is copied
// global -> LDS verbatim (no vector work, no registers).  Chunks of the third gate block carry one matrix only.
template <int H>
__global__ void __launch_bounds__(512) gru_bwd_dx_presplit_kernel(const float* __restrict__ W_ih, const float* __restrict__ W_hh,
                                                                  char* __restrict__ ws) {
    constexpr int CPS = H / 32, NCT = 4 * CPS, IMGC = 128 * 64;
    __shared__ float redw[8];
    const int slice = blockIdx.x / NCT, ct = blockIdx.x % NCT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = 0.f;
    for (int idx = tid; idx < 2 * 128 * (3 * H / 4); idx += 512) {
        const int mat = idx / (128 * (3 * H / 4)), rem = idx % (128 * (3 * H / 4));
        const f32x4 w4 = *reinterpret_cast<const f32x4*>((mat ? W_hh : W_ih) + (int64_t)(128 * slice) * 3 * H + 4 * rem);
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(w4[u]));
    }
    mx = g_wave_max(mx);
    if (lane == 0) redw[wv] = mx;
    __syncthreads();
    mx = redw[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) mx = fmaxf(mx, redw[u]);
    float sw, inv_sw;
    g_guard_scale<30>(mx, sw, inv_sw);
    if (ct == 0 && tid == 0) reinterpret_cast<float*>(ws)[slice] = inv_sw;
    const int seg = ct < 2 * CPS ? ct / CPS : (ct < 3 * CPS ? 2 : 3), cc = ct % CPS;
    const int off = (seg == 3 ? 2 : seg) * H + 32 * cc;
    char* img = ws + 64 + (int64_t)(slice * NCT + ct) * (4 * IMGC);
    const int n = tid >> 2, o = tid & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (seg < 2 || seg - 2 == j) {
            const float* src = (j ? W_hh : W_ih) + (int64_t)(128 * slice + n) * 3 * H + 8 * o + off;
            h16x8 ph, pl;
            g_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), sw, ph, pl);
            char* base = img + j * 2 * IMGC + n * 64 + ((o ^ ((n >> 2) & 3)) << 4);
            *reinterpret_cast<h16x8*>(base) = ph;
            *reinterpret_cast<h16x8*>(base + IMGC) = pl;
        }
}

// ------------------------------------------------ dm | dh, loads two chunks ahead
// NORM: `h` entered the update as hn = norm(y_prev); the backward of that norm needs the column sums of dh and of
// dh * y_prev over all atoms (against the norm's RAW input: sum dh (y_prev - mean) then needs no division by the norm's
// weight).  They are taken here, where dh is final: `hn` = y_prev is read in the accumulator layout, the sums go to
// `sums` (2 H doubles, accumulated; per wave in LDS across its tiles, one atomic per column and block at the end).
// Rounds 2-3 had a form of this kernel that requested chunk c + 1's operands -- the block's weight image (global -> LDS
// copy) and the wave's row fragments -- at the start of chunk c and waited for them at its end.  Measured on it
// (DESIGN 3c): with BOTH served from nowhere / from L2 it took 2.08 instead of 3.0 ms at c4's size, with either one alone
// it did not move -- a chunk lasted as long as the slower of its two loads, not as long as its 48 MFMAs.  Here both run
// TWO chunks ahead: a ring of
// three weight images (96 KB) and three row-fragment register sets, the chunk loop unrolled by three so that every
// set has a fixed role per body.  vmcnt retires in order and the compiler counts only the loads it can see, so ALL
// loads of the loop are inline assembly (it inserts no wait of its own for them) and the one wait per chunk,
// "everything but what this chunk issued", is written out -- tied to the registers of the NEXT chunk's row set so that
// nothing that reads them can move above it.
template <int H, bool NORM>
__global__ void __launch_bounds__(512) gru_bwd_dx_deep_f16_kernel(const char* __restrict__ pieces,
                                                                  const float* __restrict__ inv_scale,
                                                                  float* __restrict__ dm, float* __restrict__ dh, int64_t V,
                                                                  const char* __restrict__ wws,
                                                                  const float* __restrict__ hn, double* sums) {
    constexpr int NS = H / 128, CPS = H / 32, NCT = 4 * CPS;
    constexpr int TILE_BYTES = 32 * 4 * H * 4;
    constexpr int IMGC = 128 * 64;             // one (matrix, piece) chunk image: 128 output rows x 32 k fp16
    constexpr int BUF = 4 * IMGC;              // 32 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];          // three chunk images
    __shared__ double stat_s[NORM ? 8 : 1][2][128];            // NORM: per wave, column sums of dh | dh * hn

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % NS;
    const int pblock = (jb / NS) * 8 + xcd, pblocks = gridDim.x / NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;

    const int64_t rounds_total = (V + 255) / 256;              // a round = 256 rows: every wave its own 32-row tile
    if (pblock >= rounds_total) return;
    const int64_t nrounds = (rounds_total - pblock + pblocks - 1) / pblocks;
    if (NORM)
        for (int i = tid; i < 8 * 2 * 128; i += 512) (&stat_s[0][0][0])[i] = 0.0;   // (published by the first chunk's barrier)
    const float inv_sw = reinterpret_cast<const float*>(wws)[slice];

    // chunk ct: gate blocks 0, 1 with both matrices, then block 2 as (dan, W_ih), then as (dnh, W_hh)
    auto chunk_seg = [](int ct) { return ct < 2 * CPS ? ct / CPS : (ct < 3 * CPS ? 2 : 3); };
    auto chunk_cc = [](int ct) { return ct % CPS; };
    auto bfrag = [&](const char* wb, int mat, int piece, int nb, int st) {
        const int n = 32 * nb + r;
        const int o = 2 * st + hi;
        return *reinterpret_cast<const h16x8*>(wb + (mat * 2 + piece) * IMGC + n * 64 + ((o ^ ((n >> 2) & 3)) << 4));
    };
    f32x16 d_m[4], d_h[4];                                 // 32 rows x 128 features of dm and of dh per wave
    auto product = [&](f32x16 (&d)[4], const char* wb, int mat, int st, const h16x8& ah, const h16x8& al) {
#pragma unroll
        for (int nb = 0; nb < 4; nb += 2) {
            const h16x8 w0h = bfrag(wb, mat, 0, nb, st), w0l = bfrag(wb, mat, 1, nb, st);
            const h16x8 w1h = bfrag(wb, mat, 0, nb + 1, st), w1l = bfrag(wb, mat, 1, nb + 1, st);
            d[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w0h, d[nb], 0, 0, 0);
            d[nb + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w1h, d[nb + 1], 0, 0, 0);
            d[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0l, d[nb], 0, 0, 0);
            d[nb + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1l, d[nb + 1], 0, 0, 0);
            d[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0h, d[nb], 0, 0, 0);
            d[nb + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1h, d[nb + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // this wave's share of chunk ct's weight image -> ring buffer `buf`: four 1 KB copies (two for the one-matrix chunks)
    auto w_issue = [&](int ct, int buf) {
        const int seg = chunk_seg(ct);
        const char* src = wws + 64 + (int64_t)(slice * NCT + ct) * BUF + lane * 16;
        const char* dst = smem + buf * BUF;
        const int first = seg == 3 ? 16 : 0, count = seg < 2 ? 32 : 16;
        for (int i = first + wv; i < first + count; i += 8) g_copy_to_lds(src + i * 1024, dst + i * 1024);
    };
    auto rows_ptr = [&](int64_t tile, int ct) {
        return pieces + tile * (int64_t)TILE_BYTES + (chunk_seg(ct) * (H / 16) + 2 * chunk_cc(ct)) * 2048 + r * 32 + hi * 16;
    };
    // a row set = (K step 0: hi, lo piece; K step 1: hi, lo piece) of the wave's 32 rows for one chunk
#define DX_ROWS(S0, S1, S2, S3, PTR)                                                                                      \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"                    \
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"             \
                 : "=&v"(S0), "=&v"(S1), "=&v"(S2), "=&v"(S3)                                                             \
                 : "v"(PTR)                                                                                               \
                 : "memory")
    // the wait itself carries no operands (nothing has to be moved into place in front of it); the empty statement behind
    // it re-defines the set, so every reader -- and every copy the register allocator may want -- comes after the wait
#define DX_LANDED(S0, S1, S2, S3) asm volatile("" : "+v"(S0), "+v"(S1), "+v"(S2), "+v"(S3))
#define DX_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

    const int64_t tiles = (V + 31) / 32;
    int64_t rd = 0;
    int64_t tile = (int64_t)pblock * 8 + wv;
    bool live_tile = tile < tiles;
    if (tile >= tiles) tile = tiles - 1;                   // a wave past the end repeats the last tile and stores nothing
    int64_t tile_next = nrounds > 1 ? (int64_t)(pblock + pblocks) * 8 + wv : tile;
    if (tile_next >= tiles) tile_next = tiles - 1;
    int ct = 0;

    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        // pass = (column blocks, rows): full tile 2 x 16 (NORM: 1 x 8, its hn values need the registers); ragged tile 1 x 4
        constexpr int RPP = FULL ? (NORM ? 8 : 16) : 4, NBP = FULL && !NORM ? 2 : 1, PASSES = 64 / (RPP * NBP);
        constexpr int PPB = 16 / RPP;                      // passes per column block (NBP == 1)
        unsigned ln = (unsigned)lane;                      // opaque: keeps the lane offsets from becoming loop invariants
        asm volatile("" : "+v"(ln));
        const unsigned lo = ((ln >> 5) << 2) * H + 128 * slice + (ln & 31u);
        float* dhb = dh + tile * 32 * H + lo;
        float* dmb = dm + tile * 32 * H + lo;
        const float* hnb = NORM ? hn + tile * 32 * H + lo : nullptr;
        const int64_t row0 = tile * 32 + 4 * hi;
        // inv_scale[row]: 1 / (scale of the row's gate gradients).  Accumulator entry i of a lane is row
        // 8 (i >> 2) + 4 hi + (i & 3): four consecutive rows per i >> 2, one 16-byte load each, requested with the pass's
        // other loads (per pass: nothing is kept across passes, and nothing across the K loop -- a staging of these 128
        // bytes in LDS at the start of a round put an s_waitcnt vmcnt(1) of the compiler's into the K loop, +15 %)
        const float* rsp = inv_scale + tile * 32 + ((ln >> 5) << 2);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int nb0 = NBP == 2 ? 2 * ps : ps / PPB, i0 = NBP == 2 ? 0 : RPP * (ps % PPB);
            float prev[NBP][RPP], hv[NBP][RPP];
            f32x4 un4[RPP / 4];
#pragma unroll
            for (int g4 = 0; g4 < RPP / 4; ++g4)
#ifdef MPNN_ABL_DX_NOSCALE      // timing experiment only: the rows' scales are not undone (wrong results)
                un4[g4] = f32x4{inv_sw, inv_sw, inv_sw, inv_sw};
#else
                un4[g4] = *reinterpret_cast<const f32x4*>(rsp + 8 * ((i0 >> 2) + g4)) * inv_sw;
#endif
#pragma unroll
            for (int b = 0; b < NBP; ++b)
#pragma unroll
                for (int q = 0; q < RPP; ++q) {
                    const int i = i0 + q, dr = 8 * (i >> 2) + (i & 3);
                    int ro = dr * H + 32 * (nb0 + b);
                    if (!FULL && row0 + dr >= V) ro = (int)(V - 1 - row0) * H + 32 * (nb0 + b);   // (clamped: value unused)
                    prev[b][q] = dhb[ro];
                    if (NORM) hv[b][q] = hnb[ro];
                }
            __builtin_amdgcn_sched_barrier(0);
            if (!FULL) {
                // Ragged tile: some of these values are used under `row < V` only.  A load that is not consumed on every
                // path stays "pending" in the compiler's wait-count bookkeeping when the chunk loop is re-entered, and it
                // then protects the register with an s_waitcnt vmcnt(1) INSIDE the K loop -- which also waits for the loop's
                // own two-chunks-ahead requests (+15 % on the whole kernel).  Consume them all here.
#pragma unroll
                for (int g4 = 0; g4 < RPP / 4; ++g4) asm volatile("" ::"v"(un4[g4]));
#pragma unroll
                for (int b = 0; b < NBP; ++b)
#pragma unroll
                    for (int q = 0; q < RPP; ++q) {
                        asm volatile("" ::"v"(prev[b][q]));
                        if (NORM) asm volatile("" ::"v"(hv[b][q]));
                    }
            }
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                const int nb = nb0 + b;
                float sum_d = 0.0f, sum_dh = 0.0f;
#pragma unroll
                for (int q = 0; q < RPP; ++q) {
                    const int i = i0 + q, dr = 8 * (i >> 2) + (i & 3);
                    if (FULL || row0 + dr < V) {
                        const float un = un4[q >> 2][q & 3];
                        const float dhv = d_h[nb][i] * un + prev[b][q];
                        dmb[dr * H + 32 * nb] = d_m[nb][i] * un;
                        dhb[dr * H + 32 * nb] = dhv;
                        if (NORM) {
                            sum_d += dhv;
                            sum_dh = fmaf(dhv, hv[b][q], sum_dh);
                        }
                    }
                }
                if (NORM) {
                    sum_d += __shfl_xor(sum_d, 32);
                    sum_dh += __shfl_xor(sum_dh, 32);
                    if (hi == 0) {
                        stat_s[wv][0][32 * nb + r] += (double)sum_d;
                        stat_s[wv][1][32 * nb + r] += (double)sum_dh;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One chunk.  X = row set of this chunk, Z = of the next one (in flight since the previous body), Y = the set that is
    // free (last chunk's) and receives chunk + 2; BI = ring buffer of this chunk's weight image.
#define DX_BODY(X0, X1, X2, X3, Y0, Y1, Y2, Y3, Z0, Z1, Z2, Z3, BI)                                                       \
    {                                                                                                                     \
        if (ct == 0) {                                                                                                    \
            _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                              \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) { d_m[nb][i] = 0.f; d_h[nb][i] = 0.f; }                    \
        }                                                                                                                 \
        g_barrier_lds(); /* every wave's share of this chunk's image has landed; last chunk's buffer is free */          \
        const int ct2 = ct + 2 >= NCT ? ct + 2 - NCT : ct + 2;                                                            \
        const char* rp = rows_ptr(ct + 2 >= NCT ? tile_next : tile, ct2);                                                 \
        DX_ROWS(Y0, Y1, Y2, Y3, rp);                                                                                      \
        w_issue(ct2, (BI + 2) % 3);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        {                                                                                                                 \
            const char* wb = smem + BI * BUF;                                                                             \
            const int seg = chunk_seg(ct);                                                                                \
            if (seg != 3) product(d_m, wb, 0, 0, X0, X1);                                                                 \
            if (seg != 2) product(d_h, wb, 1, 0, X0, X1);                                                                 \
            if (seg != 3) product(d_m, wb, 0, 1, X2, X3);                                                                 \
            if (seg != 2) product(d_h, wb, 1, 1, X2, X3);                                                                 \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        /* all but this chunk's requests (four row loads + four or two copies) have landed */                             \
        if (chunk_seg(ct2) < 2) DX_WAIT(8);                                                                               \
        else DX_WAIT(6);                                                                                                  \
        DX_LANDED(Z0, Z1, Z2, Z3);                                                                                        \
        if (ct == NCT - 1) {                                                                                              \
            if (live_tile) {                                                                                              \
                if (tile * 32 + 32 <= V) epilogue(std::true_type{});                                                      \
                else epilogue(std::false_type{});                                                                         \
            }                                                                                                             \
            ++rd;                                                                                                         \
            tile = tile_next;                                                                                             \
            live_tile = (int64_t)(pblock + rd * pblocks) * 8 + wv < tiles;                                                \
            tile_next = rd + 1 < nrounds ? (int64_t)(pblock + (rd + 1) * pblocks) * 8 + wv : tile;                        \
            if (tile_next >= tiles) tile_next = tiles - 1;                                                                \
            ct = 0;                                                                                                       \
        } else {                                                                                                          \
            ++ct;                                                                                                         \
        }                                                                                                                 \
    }

    h16x8 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    {
        w_issue(0, 0);
        w_issue(1, 1);
        const char* p0 = rows_ptr(tile, 0);
        const char* p1 = rows_ptr(tile, 1);
        DX_ROWS(a0, a1, a2, a3, p0);
        DX_ROWS(b0, b1, b2, b3, p1);
        DX_WAIT(0);
        DX_LANDED(a0, a1, a2, a3);
        DX_LANDED(b0, b1, b2, b3);
        c0 = a0; c1 = a0; c2 = a0; c3 = a0;                 // (defined before the first body overwrites it)
    }
    const int64_t total = nrounds * NCT;
#pragma unroll 1
    for (int64_t g = 0; g < total; g += 3) {
        DX_BODY(a0, a1, a2, a3, c0, c1, c2, c3, b0, b1, b2, b3, 0)
        if (g + 1 >= total) break;
        DX_BODY(b0, b1, b2, b3, a0, a1, a2, a3, c0, c1, c2, c3, 1)
        if (g + 2 >= total) break;
        DX_BODY(c0, c1, c2, c3, b0, b1, b2, b3, a0, a1, a2, a3, 2)
    }
#undef DX_BODY
#undef DX_ROWS
#undef DX_WAIT
#undef DX_LANDED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last two bodies' requests (clamped repeats) before the block ends
    if (NORM) {
        __syncthreads();
        if (tid < 256) {
            const int k = tid >> 7, cl = tid & 127;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += stat_s[w][k][cl];
            atomicAdd(sums + k * H + 128 * slice + cl, t);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------- dW
// H = 128: block type = matrix: dW_ih = m^T [dar daz dan], dW_hh = h^T [dar daz dnh] (128 x 384 each = 4 x 12 tiles of
//          32 x 32; wave = 2 a-tiles x 3 b-tiles); LDS images per buffer: three gate segments + X.
// H = 256: block type = 3 * matrix + gate: one 256 x 256 block of dW_matrix (8 x 8 tiles; wave = 2 a-tiles x 4 b-tiles);
//          LDS images per buffer: the two halves of the gate segment + the two halves of X.
// An image = 8 ksteps = 128 columns of the 32-atom tile, as the workspace has it ([kstep][32 rows][16 columns], 8 KB per
// piece); a buffer = 4 images x 2 pieces = 64 KB, double-buffered.  The gate pieces need no processing, so they go from
// global memory straight into LDS (global_load_lds_dwordx4: a wave copies 1 KB blocks verbatim,
// tools/microbench/lds_direct_load.hip) TWO tiles ahead, into the buffer the block has just finished reading; m | h rows
// travel through registers (they are split here) and are also fetched two tiles ahead.  Columns are read with
// ds_read_b64_tr_b16: a 16-lane group covers 4 rows x 16 columns = 128 contiguous bytes.
template <int H>
__global__ void __launch_bounds__(512) gru_bwd_dw_f16_kernel(const float* __restrict__ m, const float* __restrict__ h,
                                                             const char* __restrict__ pieces,
                                                             const float* __restrict__ inv_scale, float* dW_ih,
                                                             float* dW_hh, int64_t V) {
    constexpr int TILE_BYTES = 32 * 4 * H * 4;
    constexpr int KS = H / 16;                 // ksteps per segment
    constexpr int NXI = H / 128;               // X images (slots 4 - NXI .. 3); gate images in slots 0 .. 3 - NXI
    constexpr int NGB = (4 - NXI) * 16;        // 1 KB gate blocks per tile: 48 / 32
    constexpr int NB = H == 128 ? 3 : 4;       // b-tiles per wave
    constexpr int NACC = 2 * NB;
    constexpr int BUF = 8 * G_IMG;             // image (piece, slot) at (4 * piece + slot) * G_IMG
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    // The NY blocks that walk the SAME tiles (one per matrix, or per (matrix, gate)) share operands -- the gate pieces
    // between the two matrices, m | h between the gates -- so they are numbered onto one XCD, next to each other: launched
    // together and working at the same pace, all but the first find a tile's bytes in that XCD's L2 (PMC at width 256:
    // 12 H floats per atom from HBM when every block read for itself).  gridDim.x = NY * (tile streams, a multiple of 8).
    constexpr int NY = H == 128 ? 2 : 6;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int by = jb % NY;
    const int mat = H == 128 ? by : by / 3;
    const int gate = H == 128 ? 0 : by % 3;                    // H = 256: the gate block of dW this block owns
    const float* X = mat == 0 ? m : h;
    const int64_t tiles = (V + 31) / 32;
    const int64_t t0 = (jb / NY) * 8 + xcd, tstep = gridDim.x / NY;
    if (t0 >= tiles) return;                                // (block-uniform; only when the batch has fewer tiles than streams)

    // wave wv copies blocks (NGB / 8) wv ... of the tile's NGB (slot, kstep, piece) gate blocks
    auto issue_gates = [&](int64_t t, char* T) {
        const char* p = pieces + t * (int64_t)TILE_BYTES + lane * 16;
#pragma unroll
        for (int it = 0; it < NGB / 8; ++it) {
            const int b = (NGB / 8) * wv + it;
            const int slot = b >> 4, ks8 = (b >> 1) & 7, piece = b & 1;
            int ks;                                        // kstep of the workspace tile
            if (H == 128) ks = (slot == 2 ? 2 + mat : slot) * KS + ks8;
            else ks = (gate == 2 ? 2 + mat : gate) * KS + 8 * slot + ks8;
            g_copy_to_lds(p + (ks * 2 + piece) * 1024, T + (4 * piece + slot) * G_IMG + ks8 * 1024);
        }
    };
    const int srow = tid >> 4, c16 = tid & 15;
    const int x_dst = (4 - NXI) * G_IMG + (c16 >> 1) * 1024 + srow * 32 + (c16 & 1) * 16;
    struct XRows { f32x4 x[NXI][2]; float inv_sg, live; };     // consumed only in publish / park_x: nothing waits on the loads before
    auto load_x = [&](int64_t t) {
        XRows q;
        int64_t row = t * 32 + srow;
        const bool ok = row < V;
        if (!ok) row = V - 1;
        q.live = ok ? 1.0f : 0.0f;                         // rows past V count as zeros
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            q.x[j][0] = *reinterpret_cast<const f32x4*>(X + row * H + 128 * j + c16 * 8);
            q.x[j][1] = *reinterpret_cast<const f32x4*>(X + row * H + 128 * j + c16 * 8 + 4);
        }
        q.inv_sg = inv_scale[t * 32 + srow];               // this row's gate-gradient scale (inverse)
        return q;
    };
    auto publish = [&](const XRows& q, int par) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < NXI; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fabsf(q.x[j][0][u]), fabsf(q.x[j][1][u])));
        mx = g_wave_max(mx * q.live);
        const float iv = g_wave_max(q.inv_sg);             // largest inverse = the scale of the tile's largest row
        if (lane == 0) { red[8 * par + wv] = mx; red[16 + 8 * par + wv] = iv; }
    };
    float C_run = 3.0e38f;
    auto park_x = [&](const XRows& q, char* T, int par) {  // after the barrier that follows publish()
        float xm = red[8 * par], ivm = red[16 + 8 * par];
#pragma unroll
        for (int u = 1; u < 8; ++u) { xm = fmaxf(xm, red[8 * par + u]); ivm = fmaxf(ivm, red[16 + 8 * par + u]); }
        float sxo, inv_sxo;
        g_guard_scale<30>(xm, sxo, inv_sxo);
        // the gate pieces of row r carry s_r; its m | h row is split behind C / s_r, so every product of the tile carries C.
        // C <= (smallest s_r of the tile) * sxo keeps the largest-gradient row's m | h below the fp16 range; rows with
        // smaller gradients get m | h pieces below their best precision by exactly the factor their contribution is small
        const float sg = __int_as_float((254 - ((__float_as_int(ivm) >> 23) & 0xff)) << 23);
        C_run = fminf(C_run, sg * sxo);
        const float sx = C_run * q.inv_sg;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            h16x8 ph, pl;
            g_split8(q.x[j][0], q.x[j][1], sx * q.live, ph, pl);
            *reinterpret_cast<h16x8*>(T + j * G_IMG + x_dst) = ph;
            *reinterpret_cast<h16x8*>(T + (4 + j) * G_IMG + x_dst) = pl;
        }
    };

    // transposed reads: a 16-lane group takes rows 8*(g2>>1) + 4j + (0..3), columns 16*(g2&1) + (0..15) of 32-column
    // block cb = ksteps 2 cb, 2 cb + 1 of an image; lane 4q+p supplies row q, columns 4p..4p+3
    const int ag = H == 128 ? (wv & 1) : (wv & 3), bg = H == 128 ? (wv >> 1) : (wv >> 2);
    const int g2 = lane >> 4, u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3;
    auto tr_addr = [&](int slot, int cb, int j) {
        return slot * G_IMG + (2 * cb + (g2 & 1)) * 1024 + (8 * (g2 >> 1) + 4 * j + q4) * 32 + p4 * 8;
    };
    int LA[NB][2], LX[2][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) LA[b][j] = tr_addr((NB * bg + b) >> 2, (NB * bg + b) & 3, j);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) LX[a][j] = tr_addr((4 - NXI) + ((2 * ag + a) >> 2), (2 * ag + a) & 3, j);

    f32x16 R[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) R[j][q] = 0.f;

    XRows xp, xq;                                          // alternate: one is used while the other is reloaded (no copies)
    {
        issue_gates(t0, smem);
        const XRows xa = load_x(t0);
        const bool two = t0 + tstep < tiles;
        if (two) issue_gates(t0 + tstep, smem + BUF);
        xp = load_x(two ? t0 + tstep : t0);
        publish(xa, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's blocks of both tiles have landed
        __syncthreads();
        park_x(xa, smem, 0);
    }
    float C_acc = C_run, C_cur = C_run;
    int cur = 0;
    // one tile: `use` holds the m | h rows of tile t + 1 (requested one iteration ago), `reload` receives tile t + 2's
    auto body = [&](int64_t t, const XRows& use, XRows& reload) {
        g_barrier_lds();                                   // buffer `cur` is complete
        const char* T = smem + cur * BUF;
        const int64_t t2 = t + 2 * tstep;
        const bool has2 = t2 < tiles;
        reload = load_x(has2 ? t2 : t);                    // unconditional, clamped
        if (__builtin_amdgcn_readfirstlane(__float_as_int(C_cur)) != __builtin_amdgcn_readfirstlane(__float_as_int(C_acc))) {
            const float ratio = C_cur / C_acc;             // < 1, a power of two
#pragma unroll
            for (int j = 0; j < NACC; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) R[j][q] *= ratio;
            C_acc = C_cur;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const char* Tb = T + 512 * st;                 // rows +16
            const h16x8 a0h = g_tr8(Tb + LX[0][0], Tb + LX[0][1]);
            const h16x8 a0l = g_tr8(Tb + 4 * G_IMG + LX[0][0], Tb + 4 * G_IMG + LX[0][1]);
            const h16x8 a1h = g_tr8(Tb + LX[1][0], Tb + LX[1][1]);
            const h16x8 a1l = g_tr8(Tb + 4 * G_IMG + LX[1][0], Tb + 4 * G_IMG + LX[1][1]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const h16x8 bh = g_tr8(Tb + LA[b][0], Tb + LA[b][1]);
                const h16x8 bl = g_tr8(Tb + 4 * G_IMG + LA[b][0], Tb + 4 * G_IMG + LA[b][1]);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, R[NB + b], 0, 0, 0);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, R[NB + b], 0, 0, 0);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, R[NB + b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        publish(use, cur ^ 1);
        g_barrier_lds();                                   // every wave is done reading buffer `cur`
        park_x(use, smem + (cur ^ 1) * BUF, cur ^ 1);
        C_cur = C_run;
        __builtin_amdgcn_sched_barrier(0);
        // Everything in flight is due now: tile t + 1's gate copies (issued one iteration ago) and tile t + 2's m | h rows
        // (requested at the top).  Draining HERE, before the next copies go out, also keeps the compiler from parking its
        // own vmcnt(0) for the rows behind them (it does not see the copies).
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int j = 0; j < NXI; ++j) asm volatile("" ::"v"(reload.x[j][0]), "v"(reload.x[j][1]));
        asm volatile("" ::"v"(reload.inv_sg));
        if (has2) issue_gates(t2, smem + cur * BUF);       // into the buffer just read; lands during the next iteration
        cur ^= 1;
    };
    for (int64_t t = t0; t < tiles; t += 2 * tstep) {
        body(t, xp, xq);
        if (t + tstep < tiles) body(t + tstep, xq, xp);
    }
    const float inv_C = 1.0f / C_acc;
    float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = (H == 128 ? 0 : gate * H) + 32 * (NB * bg + b) + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (2 * ag + a) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, R[NB * a + b][q] * inv_C);
            }
        }
}

// pieces | one scale per row | pre-split weight images of the dm | dh kernel
static size_t gru_bwd_f16_dxw_bytes(int H) { return 64 + (size_t)(H / 128) * (4 * H / 32) * (4 * 128 * 64); }
size_t gru_bwd_f16_workspace_bytes(int64_t V, int H) {
    const int64_t tiles = (V + 31) / 32;
    return (size_t)tiles * (32 * 4 * H * 4) + (size_t)(tiles * 32) * sizeof(float) + gru_bwd_f16_dxw_bytes(H);
}

// out_norm_k != NULL: dout is the gradient of norm(out), the gate kernel turns it into the gradient of out (NORM there);
// in_norm_sums != NULL: h = hn = norm(y_prev), y_prev = in_norm_raw: the dm | dh kernel also takes the column sums the
// backward of THAT norm needs
template <int H>
static int launch_gru_bwd_f16_t(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                                const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                                float* db_ih, float* db_hh, void* workspace, int64_t V, const float* out_norm_k,
                                double* in_norm_sums, const float* in_norm_raw, hipStream_t s) {
    const int64_t tiles = (V + 31) / 32;
    char* pieces = (char*)workspace;
    float* inv_scale = (float*)(pieces + (size_t)tiles * (32 * 4 * H * 4));
    char* dxw = (char*)(inv_scale + tiles * 32);                         // (one scale per row) pre-split weights of the dm | dh kernel
    const size_t lds_dw = (size_t)2 * 8 * G_IMG + 128;       // two double-buffered tiles + the per-wave maxima of two tiles
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_dx_deep_f16_kernel<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 128 * 64);
        opt_in_((const void*)gru_bwd_dx_deep_f16_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 128 * 64);
        opt_in_((const void*)gru_bwd_dw_f16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);

    int64_t gg = H == 128 ? 1024 : 512;
    if (gg > tiles) gg = tiles;
    if (mask && out_norm_k)
        hipLaunchKernelGGL((gru_gate_f16_kernel<H, true, true>), dim3((unsigned)gg), dim3(512), 0, s, dout, h, mask, saved,
                           pieces, inv_scale, dh, db_ih, db_hh, V, out_norm_k);
    else if (out_norm_k)
        hipLaunchKernelGGL((gru_gate_f16_kernel<H, false, true>), dim3((unsigned)gg), dim3(512), 0, s, dout, h, mask, saved,
                           pieces, inv_scale, dh, db_ih, db_hh, V, out_norm_k);
    else if (mask)
        hipLaunchKernelGGL((gru_gate_f16_kernel<H, true>), dim3((unsigned)gg), dim3(512), 0, s, dout, h, mask, saved, pieces,
                           inv_scale, dh, db_ih, db_hh, V, out_norm_k);
    else
        hipLaunchKernelGGL((gru_gate_f16_kernel<H, false>), dim3((unsigned)gg), dim3(512), 0, s, dout, h, mask, saved, pieces,
                           inv_scale, dh, db_ih, db_hh, V, out_norm_k);
    int rc = launch_status("mpnn_gru_update_bwd_f32(gates, fp16 pieces)");
    if (rc) return rc;

    {   // dm | dh: streamed pre-split weights, 128-column slices
        constexpr int NS = H / 128;
        const int64_t rounds = (V + 255) / 256;
        int64_t pblocks = 256 / NS;                          // x NS slices = one block per CU
        if (pblocks > rounds) pblocks = rounds;
        pblocks = (pblocks + 7) / 8 * 8;
        hipLaunchKernelGGL(gru_bwd_dx_presplit_kernel<H>, dim3((unsigned)(NS * 4 * (H / 32))), dim3(512), 0, s, W_ih, W_hh, dxw);
        if (in_norm_sums)
            hipLaunchKernelGGL((gru_bwd_dx_deep_f16_kernel<H, true>), dim3((unsigned)(pblocks * NS)), dim3(512),
                               (size_t)3 * 4 * 128 * 64, s, pieces, inv_scale, dm, dh, V, (const char*)dxw, in_norm_raw, in_norm_sums);
        else
            hipLaunchKernelGGL((gru_bwd_dx_deep_f16_kernel<H, false>), dim3((unsigned)(pblocks * NS)), dim3(512),
                               (size_t)3 * 4 * 128 * 64, s, pieces, inv_scale, dm, dh, V, (const char*)dxw,
                               (const float*)nullptr, (double*)nullptr);
    }
    rc = launch_status("mpnn_gru_update_bwd_f32(dx, fp16 pieces)");
    if (rc) return rc;

    constexpr int NY = H == 128 ? 2 : 6;                    // block types: matrix, or (matrix, gate)
    int64_t gx = (256 / NY) / 8 * 8;                        // tile streams: one block per CU (128 KB of LDS), whole XCD groups
    while (gx > 8 && gx - 8 >= tiles) gx -= 8;
    hipLaunchKernelGGL(gru_bwd_dw_f16_kernel<H>, dim3((unsigned)(gx * NY)), dim3(512), lds_dw, s, m, h, pieces, inv_scale,
                       dW_ih, dW_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dW, fp16 pieces)");
}

int launch_gru_bwd_f16_wide(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                            const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                            float* db_ih, float* db_hh, void* workspace, int64_t V, int H, const float* out_norm_k,
                            double* in_norm_sums, const float* in_norm_raw, hipStream_t s) {
    if (H == 128)
        return launch_gru_bwd_f16_t<128>(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V,
                                         out_norm_k, in_norm_sums, in_norm_raw, s);
    return launch_gru_bwd_f16_t<256>(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V,
                                     out_norm_k, in_norm_sums, in_norm_raw, s);
}

}  // namespace mpnn
