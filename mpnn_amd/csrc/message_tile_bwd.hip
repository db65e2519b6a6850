// Weight gradient of the fused typed message + neighbour sum at nf = mf = 64, on the forward kernel's tile plan:
//   dA[k] += sum_{e of type k} dagg[dst e] (x) h[src e]                       (64 x 64 per bond type)
// replaces: the autograd of mpnn_functions/message/edge_network.py:50-51 (edge_embed.bmm(...)) with respect to the
//           edge matrices, i.e. of edge_network.py:40,52 composed with message_aggregators/adjacent_message_agg.py:18.
// No (E, mf) message gradient exists in HBM; dagg and h are each read ONCE (mpnn_edge_message_agg_bwd_da_f32, which this
// kernel replaces on molecule batches, gathers both per edge from HBM: 2.69 GB of traffic for 1.6 GB of rows on c2).
//
// Structure (one persistent 8-wave block per CU, tiles / blocks / row-tiles exactly as in message_tile.hip): the tile's
// dagg rows and h rows are staged once into LDS as fp16 image pairs.  A row-tile is "the rank-th incoming type-k edge of
// each of a block's 16 atoms", so its contribution is  D_blk^T [64 x 16] . X_rt [16 x 64]  with D_blk = the block's own
// dagg rows (no gather list needed: row m is atom m of the block) and X_rt = the slots' source rows (an empty slot reads
// the zero row and contributes nothing).  The contraction runs over the ROWS of both operands, so both fragments are
// read with gfx950's transposed LDS read (ds_read_b64_tr_b16: per 16 lanes a 4-row x 16-column block delivered
// column-major; every lane supplies its own row address, so gathered rows cost nothing extra) and feed
// v_mfma_f32_32x32x16_f16.  Wave w owns output quadrant w & 3 (32 x 32 of the 64 x 64) of ALL bond types for the blocks
// 4 (w >> 2) .. + 3 of every tile: its accumulators live in registers for the whole kernel and leave through one float
// atomic per element per wave at the end (128-byte segments, 64 MB in all).
//
// Math ("fp16x3", as the forward): one power-of-two scale sd per tile for the dagg rows; the h rows are split behind
// C / sd with C = the running minimum over the block's tiles of sd * (best scale of the tile's h rows), so that every
// tile's products carry the same factor C and the kernel-long accumulators need no per-tile fold (gru_bwd_f16.hip has the
// argument; when C drops they are multiplied by the ratio, a power of two).  x*s = hi + lo with hi = fp16(x*s),
// lo = fp16(x*s - hi); three MFMAs per product (hi*hi, hi*lo, lo*hi).  Error: normwise ~2^-22 of (tile max |dagg|)
// (tile max |h|) per term -- entries more than 2^18 below their tile's maximum lose relative (not absolute) accuracy,
// which a sum over all atoms does not see.  Two blocks are resident per CU (75 KB of LDS, 128 VGPRs).
#include "common.h"

namespace mpnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int MB_TV = 128;
constexpr int MB_F = 64;
constexpr int MB_KMAX = 4;
constexpr int MB_RTMAX = 16;
constexpr int MB_HT = (MB_TV + 1) * MB_F * 2;    // one fp16 piece of a tile image + the zero row

__host__ __device__ constexpr int mb_lds_bytes() {
    return 4 * MB_HT + 8 * 16 * MB_RTMAX * 4 + MB_TV * 4 + (8 * MB_KMAX + 4) * 4 + 128;
}

// power of two s with maxabs * s in [2^14, 2^15), and 1 / s; s is clamped to [2^-46, 2^SMAX] (gru_bwd_f16.hip)
template <int SMAX>
__device__ __forceinline__ void mb_guard_scale(float maxabs, float& scale, float& inv) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    e = e < 141 - SMAX ? 141 - SMAX : (e > 187 ? 187 : e);
    scale = __int_as_float((268 - e) << 23);
    inv = __int_as_float((e - 14) << 23);
}

// 8 consecutive ROWS (k index of the MFMA) of one column, as the two transposed reads deliver them
__device__ __forceinline__ f16x8 mb_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) message_sum_tile_bwd_kernel(
    const float* __restrict__ dagg, const float* __restrict__ h, const int32_t* __restrict__ tile_rec,
    const int32_t* __restrict__ tile_atom, const int32_t* __restrict__ tile_rtk, const int32_t* __restrict__ slots,
    float* __restrict__ dA, int num_tiles, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (wv >> 1) & 1, fb = wv & 1, bhalf = wv >> 2;          // my output quadrant; my four blocks of every tile
    auto swz = [](int n) { return (n >> 1) & 7; };

    const int D_OFF = 0, X_OFF = 2 * MB_HT;
    const int SL_OFF = 4 * MB_HT;                       // slot words, one region per block
    const int AT_OFF = SL_OFF + 8 * 16 * MB_RTMAX * 4;  // atom id of every (block, row) of the tile
    const int RK_OFF = AT_OFF + MB_TV * 4;              // first row-tile of every (block, type) + end
    const int RED_OFF = RK_OFF + (8 * MB_KMAX + 4) * 4; // partial maxima: 8 for dagg, 8 for h
    int* slw_all = reinterpret_cast<int*>(smem + SL_OFF);
    int* atoms = reinterpret_cast<int*>(smem + AT_OFF);
    int* rtk = reinterpret_cast<int*>(smem + RK_OFF);
    float* red = reinterpret_cast<float*>(smem + RED_OFF);

    if (tid < 16) {                                     // the zero rows of the X image (both pieces)
        const f16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        *reinterpret_cast<f16x4*>(smem + X_OFF + MB_TV * 128 + 8 * tid) = z;
        *reinterpret_cast<f16x4*>(smem + X_OFF + MB_HT + MB_TV * 128 + 8 * tid) = z;
    }

    struct Rec { int a0, n, r0, r1; };                  // first atom, atoms, row-tiles of block wv: [r0, r1)
    auto load_rec = [&](int t) {
        const int32_t* p = tile_rec + 16 * (int64_t)t;
        Rec r;
        r.a0 = p[0];
        r.n = p[1];
        r.r0 = p[2 + wv];
        r.r1 = p[3 + wv];
        return r;
    };
    const int srow = tid >> 4, sc4 = tid & 15;
    f32x4 sd[4], sx[4];
    int slotreg[MB_RTMAX / 4];
    int atomreg = -1, rtkreg = 0;
    const int nrk = 8 * K + 1;
    auto stage_load = [&](const Rec& r, int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = srow + 32 * j;
            const int rr = row < r.n ? row : 0;
            sd[j] = *reinterpret_cast<const f32x4*>(dagg + (int64_t)(r.a0 + rr) * MB_F + 4 * sc4);
            sx[j] = *reinterpret_cast<const f32x4*>(h + (int64_t)(r.a0 + rr) * MB_F + 4 * sc4);
        }
        const int nw = 16 * (r.r1 - r.r0);
#pragma unroll
        for (int j = 0; j < MB_RTMAX / 4; ++j) {
            if (64 * j < nw) {
                const int i = 64 * j + lane;
                slotreg[j] = slots[(int64_t)16 * r.r0 + (i < nw ? i : 0)];
            }
        }
        if (tid < MB_TV) atomreg = tile_atom[(int64_t)t * MB_TV + tid];
        if (tid < nrk) rtkreg = tile_rtk[(int64_t)t * nrk + tid];
    };
    auto stage_max = [&]() {
        float md = 0.f, mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                md = fmaxf(md, fabsf(sd[j][u]));
                mx = fmaxf(mx, fabsf(sx[j][u]));
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            md = fmaxf(md, __shfl_xor(md, o));
            mx = fmaxf(mx, __shfl_xor(mx, o));
        }
        if (lane == 0) {
            red[wv] = md;
            red[8 + wv] = mx;
        }
    };
    float C_run = 3.0e38f, C_acc = 3.0e38f;             // min over the tiles so far of scale_d * (best scale_x); of `tot`
    auto stage_write = [&](const Rec& r) {
        float md = 0.f, mx = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            md = fmaxf(md, red[u]);
            mx = fmaxf(mx, red[8 + u]);
        }
        float scd, invd, scx, invx;
        mb_guard_scale<90>(md, scd, invd);
        mb_guard_scale<30>(mx, scx, invx);
        // every tile's products must carry the same factor: the h rows are split behind C / scale_d <= their best scale,
        // C = the running minimum of scale_d * scale_x (gru_bwd_f16.hip has the argument); no per-tile accumulators
        C_run = fminf(C_run, scd * scx);
        scx = C_run * invd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = srow + 32 * j;
            f16x4 dh, dl, xh, xl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = sd[j][u] * scd, b = sx[j][u] * scx;
                dh[u] = (_Float16)a;
                dl[u] = (_Float16)(a - (float)dh[u]);
                xh[u] = (_Float16)b;
                xl[u] = (_Float16)(b - (float)xh[u]);
            }
            const int off = row * 128 + (((sc4 >> 1) ^ swz(row)) << 4) + ((sc4 & 1) << 3);
            *reinterpret_cast<f16x4*>(smem + D_OFF + off) = dh;
            *reinterpret_cast<f16x4*>(smem + D_OFF + MB_HT + off) = dl;
            *reinterpret_cast<f16x4*>(smem + X_OFF + off) = xh;
            *reinterpret_cast<f16x4*>(smem + X_OFF + MB_HT + off) = xl;
        }
        const int nw = 16 * (r.r1 - r.r0);
#pragma unroll
        for (int j = 0; j < MB_RTMAX / 4; ++j)
            if (64 * j < nw) slw_all[wv * 16 * MB_RTMAX + 64 * j + lane] = slotreg[j];
        if (tid < MB_TV) atoms[tid] = atomreg;
        if (tid < nrk) rtk[tid] = rtkreg;
    };

    // transposed-read geometry of this lane (see the header): 16-lane group gg, position 4 q + p inside it
    const int gg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int mrow = 8 * (gg >> 1) + q;                 // + 4 rd: the row (of the 16) whose address I supply for read rd
    const int dcol = 32 * nb + 16 * (gg & 1) + 4 * p;   // dagg column of my address (A operand: rows n of the quadrant)
    const int xcol = 32 * fb + 16 * (gg & 1) + 4 * p;   // h column of my address (B operand: columns f of the quadrant)
    auto img_off = [&](int row, int col) { return row * 128 + ((((col >> 3)) ^ swz(row)) << 4) + (((col >> 2) & 1) << 3); };

    f32x16 tot[MB_KMAX];
#pragma unroll
    for (int k = 0; k < MB_KMAX; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) tot[k][i] = 0.f;

    const int G = gridDim.x;
    for (int t = blockIdx.x; t < num_tiles; t += G) {
        // Two of these blocks are resident per CU (75 KB of LDS, 128 VGPRs each): one block's loads and staging sit under
        // the other's row-tile loop, which does more than a register prefetch of the next tile did (0.70 ms with it and
        // one block per CU)
        const Rec cur = load_rec(t);
        stage_load(cur, t);
        stage_max();
        __syncthreads();
        stage_write(cur);
        if (__builtin_amdgcn_readfirstlane(__float_as_int(C_run)) != __builtin_amdgcn_readfirstlane(__float_as_int(C_acc))) {
            const float ratio = C_run / C_acc;          // < 1, a power of two (0 before the first tile: tot is zero)
#pragma unroll
            for (int k = 0; k < MB_KMAX; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) tot[k][i] *= ratio;
            C_acc = C_run;
        }
        __syncthreads();
        // Block by block; inside a block its row-tiles in memory order (type-major), the type loop unrolled so that each
        // type has its own static accumulator.  The h fragments of row-tile i + 1 and the slot words of row-tile i + 2
        // are requested before the MFMAs of row-tile i, across the type boundaries (they do not depend on the type).
        // (A flat walk over all four blocks driven by a scalar state machine was measured at 0.90 ms: its bookkeeping
        // cost more than it hid.)
#pragma unroll 1
        for (int b = 0; b < 4; ++b) {
            const int blk = 4 * bhalf + b;
            const int rb0 = __builtin_amdgcn_readfirstlane(rtk[blk * K]);
            const int nblk = __builtin_amdgcn_readfirstlane(rtk[blk * K + K]) - rb0;
            if (nblk == 0) continue;                                   // block without row-tiles (wave-uniform)
            // A operand: the block's own dagg rows, transposed (rows of A = dagg columns of my quadrant)
            f16x8 ah, al;
            {
                const int at0 = atoms[16 * blk + mrow], at1 = atoms[16 * blk + mrow + 4];
                const int r0 = at0 >= 0 ? at0 - cur.a0 : 0, r1 = at1 >= 0 ? at1 - cur.a0 : 0;
                const int o0 = img_off(r0, dcol), o1 = img_off(r1, dcol);
                ah = mb_tr8(smem + D_OFF + o0, smem + D_OFF + o1);
                al = mb_tr8(smem + D_OFF + MB_HT + o0, smem + D_OFF + MB_HT + o1);
            }
            const int* slw = slw_all + blk * 16 * MB_RTMAX;
            auto words = [&](int i, int& w0, int& w1) {
                const int ii = i < nblk ? i : 0;
                w0 = slw[16 * ii + mrow];
                w1 = slw[16 * ii + mrow + 4];
            };
            auto frags = [&](int w0, int w1, f16x8& bh, f16x8& bl) {
                const int o0 = img_off(w0 & 0xff, xcol), o1 = img_off(w1 & 0xff, xcol);
                bh = mb_tr8(smem + X_OFF + o0, smem + X_OFF + o1);
                bl = mb_tr8(smem + X_OFF + MB_HT + o0, smem + X_OFF + MB_HT + o1);
            };
            int wa0, wa1, wb0, wb1;
            f16x8 bh, bl, bh_n, bl_n;
            words(0, wa0, wa1);
            frags(wa0, wa1, bh, bl);
            words(1, wa0, wa1);                                        // (wa: words of the next row-tile)
            int i = 0;
#pragma unroll
            for (int k = 0; k < MB_KMAX; ++k) {
                if (k < K) {
                    // scalar bounds: the transposed reads need every lane active
                    const int i1 = __builtin_amdgcn_readfirstlane(rtk[blk * K + k + 1]) - rb0;
                    for (; i < i1; ++i) {
                        words(i + 2, wb0, wb1);
                        frags(wa0, wa1, bh_n, bl_n);                   // row-tile i + 1 (a harmless re-read at the end)
                        __builtin_amdgcn_sched_barrier(0);
                        tot[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, tot[k], 0, 0, 0);
                        tot[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, tot[k], 0, 0, 0);
                        tot[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, tot[k], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bh = bh_n;
                        bl = bl_n;
                        wa0 = wb0;
                        wa1 = wb1;
                    }
                }
            }
        }
        __syncthreads();                                // every wave is done with this tile's LDS
    }
    // ---- my quadrant of every type -> dA (accumulated across blocks with float atomics; 128-byte segments)
    const float inv_C = C_acc < 1.0e38f ? 1.0f / C_acc : 0.f;
    const int col = lane & 31;
#pragma unroll
    for (int k = 0; k < MB_KMAX; ++k) {
        if (k < K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                atomicAdd(dA + ((int64_t)k * MB_F + 32 * nb + row) * MB_F + 32 * fb + col, tot[k][i] * inv_C);
            }
        }
    }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_message_aggregate_bwd_da_f32(const float* dagg, const float* h, const int32_t* tile_rec,
                                                 const int32_t* tile_atom, const int32_t* tile_rtk, const int32_t* slots,
                                                 float* dA, int64_t V, int64_t num_tiles, int K, int nf, int mf,
                                                 void* stream) {
    MPNN_REQUIRE(nf == MB_F && mf == MB_F, "mpnn_message_aggregate_bwd_da_f32: nf = mf = 64 only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(K >= 1 && K <= MB_KMAX, "mpnn_message_aggregate_bwd_da_f32: 1 <= K <= %d bond types (got %d)", MB_KMAX, K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 24), "mpnn_message_aggregate_bwd_da_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(dagg && h && tile_rec && tile_atom && tile_rtk && slots && dA, "mpnn_message_aggregate_bwd_da_f32: NULL buffer");
    static const hipError_t attr = [] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)message_sum_tile_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mb_lds_bytes());
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    int64_t blocks = 512;                                  // two resident blocks per CU
    if (blocks > num_tiles) blocks = num_tiles;
    hipLaunchKernelGGL(message_sum_tile_bwd_kernel, dim3((unsigned)blocks), dim3(512), mb_lds_bytes(), (hipStream_t)stream, dagg, h,
                       tile_rec, tile_atom, tile_rtk, slots, dA, (int)num_tiles, K);
    return launch_status("mpnn_message_aggregate_bwd_da_f32");
}
