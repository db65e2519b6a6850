// GRU backward at H = 64, gate gradients split ONCE (reference: mpnn_functions/update/gru_update.py:26-35, autograd of it).
//
// Same block structure as gru_bwd_fused_split_kernel (gru_bwd.hip): a double-buffered 32-atom LDS tile, four dW
// waves that also stage the next tile, four dx waves with their weight slice resident in registers, one block
// barrier per tile.  That kernel keeps the gate gradients in LDS as fp32 and every consumer splits what it reads
// into three bf16 pieces: the four dx waves each split the same rows (4x redundant for dar / daz), the dW waves the
// same columns twice -- 22 eight-element splits of ~44 VALU instructions per SIMD per tile, which is what holds the
// matrix pipe at ~47 % busy (in-kernel cycle stamps: VALU issue, not loads or LDS, is the critical resource).
// Here the staging threads split each gate gradient once and park the PIECES:
//   P   3 pieces x 2 column pairs (dar|daz, dan|dnh) of [32 rows][128 x bf16] images, 256-byte rows, 16-byte chunks
//       XOR-swizzled so that both kinds of read below are conflict-free
//   GZ  [32][68] fp32   dout * mask * z, added to dh in the epilogue
//   X   [32][128] fp32  m | h rows (the dW waves' A operand, still split by its two consumers)
// dx waves read their A fragments as rows (three ds_read_b128 per K step, no VALU at all); dW waves need COLUMNS of
// the same images -- eight consecutive atoms of one gate column per lane -- and get them with gfx950's transposed
// LDS read (ds_read_b64_tr_b16: per 16 lanes a 4-row x 16-column block delivered column-major), two per piece.
// The bias gradients are column sums of the gate gradients: the staging threads own fixed columns, so they add
// what they stage into 16 registers and reduce once at the end (the fp32 tile kernel re-read the tile for this).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "split_math.h"

namespace mpnn {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int PS_H = 64;
constexpr int PS_IMG = 32 * 256;                       // bytes of one [32 rows][128 x bf16] image
constexpr int PS_P = 6 * PS_IMG;                       // image (piece, pair) at (2 * piece + pair) * PS_IMG
constexpr int PS_LDZ = 68;
constexpr int PS_GZ = 32 * PS_LDZ * 4;
constexpr int PS_LDX = 2 * PS_H;
constexpr int PS_X = 32 * PS_LDX * 4;
constexpr int PS_TILE = PS_P + PS_GZ + PS_X;           // 74,240 bytes; two of them per block

// byte offset of 16-byte chunk `ch` (8 columns) of row `row` inside an image
__device__ __forceinline__ int ps_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ void ps_gate_grads4(const f32x4& dout, const f32x4& hv, const f32x4& r, const f32x4& z,
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

__device__ __forceinline__ bf16x8 ps_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

template <bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_bwd_presplit64_kernel(
    const float* __restrict__ dout, const float* __restrict__ m, const float* __restrict__ h,
    const float* __restrict__ mask, const float* __restrict__ W_ih, const float* __restrict__ W_hh,
    const float* __restrict__ saved, float* __restrict__ dm, float* __restrict__ dh, float* dW_ih, float* dW_hh,
    float* db_ih, float* db_hh, int64_t V) {
    constexpr int H = PS_H;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: role branches become s_cbranch
    const int i = lane & 31, hi = lane >> 5;
    const int64_t tiles = (V + 31) / 32;
    const int64_t t0 = blockIdx.x, tstep = gridDim.x;

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
        // chunk of this thread's four columns inside a 64-column segment half, pre-swizzled for its two rows
        // (rows srow and srow + 16 share the swizzle term)
        auto stage_write = [&](const Staged& q, int half, char* T, float count) {   // count: 1, or 0 for the clamped re-stage
            f32x4 sg[4];
            const float mk = q.ok ? q.mk : 0.0f;          // rows past V contribute exact zeros
            ps_gate_grads4(q.v_do, q.vh, q.v_r, q.v_z, q.v_n, q.v_nh, mk, sg[0], sg[1], sg[2], sg[3]);
            const f32x4 gz = q.v_do * mk * q.v_z;
            const float live = q.ok ? 1.0f : 0.0f;
            const int row = srow + 16 * half;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x4 ph, pm, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[4 * s + j] = fmaf(sg[s][j], count, cs[4 * s + j]);
                    __bf16 a, b, c;
                    split3(sg[s][j], a, b, c);
                    ph[j] = a; pm[j] = b; pl[j] = c;
                }
                char* a = T + (s >> 1) * PS_IMG + ps_off(row, (s & 1) * 8 + (sc4 >> 3)) + (sc4 & 4) * 2;
                *reinterpret_cast<bf16x4*>(a) = ph;
                *reinterpret_cast<bf16x4*>(a + 2 * PS_IMG) = pm;
                *reinterpret_cast<bf16x4*>(a + 4 * PS_IMG) = pl;
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(T + PS_P) + row * PS_LDZ + sc4) = gz;
            float* x = reinterpret_cast<float*>(T + PS_P + PS_GZ) + row * PS_LDX + sc4;
            *reinterpret_cast<f32x4*>(x) = q.vm * live;
            *reinterpret_cast<f32x4*>(x + H) = q.vh * live;
        };
        // 8 consecutive rows of one fp32 LDS column (the K=16 fragment of a row-contraction), split in 3
        auto column_frag = [&](const float* base, int ld, bf16x8& ph, bf16x8& pm, bf16x8& pl) {
            const f32x4 x0 = {base[0], base[ld], base[2 * ld], base[3 * ld]};
            const f32x4 x1 = {base[4 * ld], base[5 * ld], base[6 * ld], base[7 * ld]};
            split8(x0, x1, ph, pm, pl);
        };

        // wave (mat, jg): BOTH 32-row blocks of X_mat against three of the six 32-column gate blocks.
        // block b of this wave = (segment, half): jg 0 -> (dar,0) (dar,1) (daz,0); jg 1 -> (daz,1) (n,0) (n,1) with
        // n = dan for W_ih, dnh for W_hh; pair = segment >> 1, cb = 32-column block inside the 128-column image
        const int ns = 2 + mat;
        const int bseg[3] = {jg == 0 ? 0 : 1, jg == 0 ? 0 : ns, jg == 0 ? 1 : ns};
        const int bhalf[3] = {jg == 0 ? 0 : 1, jg == 0 ? 1 : 0, jg == 0 ? 0 : 1};
        // transposed-read addressing: in its 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3 of the 4 x 16
        // block; groups 0/1 take columns 0-15 / 16-31, groups 2/3 the same columns 8 rows further (K half of the lane).
        // Lane part of the address of read j (rows +4j) of column block cb; the rest (K step, piece) is immediates.
        const int g2 = lane >> 4, u = lane & 15, q4 = u >> 2, p4 = u & 3;
        int LA[3][2];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int cb = (bseg[b] & 1) * 2 + bhalf[b];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                LA[b][j] = (bseg[b] >> 1) * PS_IMG + ps_off(8 * (g2 >> 1) + 4 * j + q4, cb * 4 + 2 * (g2 & 1) + (p4 >> 1)) +
                           8 * (p4 & 1);
        }

        f32x16 R[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) R[j][q] = 0.f;
        if (t0 < tiles) {
            const Staged q0 = stage_load(t0, 0), q1 = stage_load(t0, 1);
            stage_write(q0, 0, smem, 1.0f);
            stage_write(q1, 1, smem, 1.0f);
        }
        int cur = 0;
        for (int64_t t = t0; t < tiles; t += tstep) {
            __syncthreads();
            const char* T = smem + cur * PS_TILE;
            const float* X = reinterpret_cast<const float*>(T + PS_P + PS_GZ);
            // Staging is unconditional (past the end the tile index is clamped and the staged tile is never read):
            // under `if (more)` the compiler sinks the loads into that branch, i.e. behind the MFMAs they should cover
            const bool more = t + tstep < tiles;
            const int64_t tn = more ? t + tstep : t;
            const Staged nx0 = stage_load(tn, 0), nx1 = stage_load(tn, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int row0 = 16 * st + 8 * hi;        // this lane half's 8 rows of the K=16 step
                bf16x8 a0h, a0m, a0l, a1h, a1m, a1l;
                column_frag(X + row0 * PS_LDX + mat * H + i, PS_LDX, a0h, a0m, a0l);
                column_frag(X + row0 * PS_LDX + mat * H + 32 + i, PS_LDX, a1h, a1m, a1l);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    // rows 16 st + 4 j + ... of the swizzle: 16 st leaves both swizzle terms unchanged
                    const char* c0 = T + 256 * 16 * st + LA[b][0];
                    const char* c1 = T + 256 * 16 * st + LA[b][1];
                    const bf16x8 bh = ps_tr8(c0, c1);
                    const bf16x8 bm = ps_tr8(c0 + 2 * PS_IMG, c1 + 2 * PS_IMG);
                    const bf16x8 bl = ps_tr8(c0 + 4 * PS_IMG, c1 + 4 * PS_IMG);
                    mma6x2_b(R[b], R[3 + b], a0h, a0m, a0l, a1h, a1m, a1l, bh, bm, bl);   // independent neighbours
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                char* Tn = smem + (cur ^ 1) * PS_TILE;
                stage_write(nx0, 0, Tn, more ? 1.0f : 0.0f);
                stage_write(nx1, 1, Tn, more ? 1.0f : 0.0f);
            }
            cur ^= 1;
        }
        float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = 32 * (3 * jg + j % 3) + i;    // R[a*3 + b]: row block a, column block 3*jg + b
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (j / 3) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, R[j][q]);
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
        if (lane < 16 && t0 < tiles) {
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
    // 36 bf16x8 fragments (3 gates x 4 K-steps x 3 pieces) of the wave's weight slice stay in registers
    const int which = ((wv - 4) >> 1) & 1, nb = (wv - 4) & 1;
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
    // row reads: lane (i, hi) takes columns hi*32 + 8*st .. +7 of gate segment g of row i (the K order of wfrag);
    // g = 2 is dan (dm waves) or dnh (dh waves), the two halves of column pair 1
    int DA[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int e = g == 2 ? which : g;                 // 64-column half of the pair image
#pragma unroll
        for (int st = 0; st < 4; ++st) DA[g][st] = (g >> 1) * PS_IMG + ps_off(i, e * 8 + hi * 4 + st);
    }
    float* outp = which == 0 ? dm : dh;
    const int col = 32 * nb + i;
    const unsigned lane_off = (unsigned)(4 * hi * H + col);   // acc_row(q, lane) = 4*hi + (q&3) + 8*(q>>2)
    int cur = 0;
    for (int64_t t = t0; t < tiles; t += tstep) {
        __syncthreads();
        const char* T = smem + cur * PS_TILE;
        // two accumulators (even / odd K steps) issued alternately: a single one is a chain of 72 dependent MFMAs
        f32x16 d, d1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { d[q] = 0.f; d1[q] = 0.f; }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int st = 0; st < 4; st += 2) {
                const char* a = T + DA[g][st];
                const char* b = T + DA[g][st + 1];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(a + 2 * PS_IMG);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(a + 4 * PS_IMG);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(b + 2 * PS_IMG);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b + 4 * PS_IMG);
                mma6x2(d, ah, am, al, wfrag(g, st, 0), wfrag(g, st, 1), wfrag(g, st, 2), d1, bh, bm, bl,
                       wfrag(g, st + 1, 0), wfrag(g, st + 1, 1), wfrag(g, st + 1, 2));
            }
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] += d1[q];
        if (which == 1) {                                 // scalar branch; the 16 LDS reads go out together
            const float* GZ = reinterpret_cast<const float*>(T + PS_P);
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] += GZ[acc_row(q, lane) * PS_LDZ + col];
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
        cur ^= 1;
    }
}

int launch_gru_bwd_presplit64(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                              const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                              float* db_ih, float* db_hh, int64_t V, hipStream_t s) {
    const int64_t tiles = (V + 31) / 32;
    const size_t lds = (size_t)2 * PS_TILE;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_presplit64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)gru_bwd_presplit64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t blocks = 256;                                  // one 8-wave block per CU (145 KB of LDS)
    if (blocks > tiles) blocks = tiles;
    if (mask)
        hipLaunchKernelGGL((gru_bwd_presplit64_kernel<true>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    else
        hipLaunchKernelGGL((gru_bwd_presplit64_kernel<false>), dim3((unsigned)blocks), dim3(512), lds, s, dout, m, h, mask,
                           W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(pre-split tile)");
}

}  // namespace mpnn
