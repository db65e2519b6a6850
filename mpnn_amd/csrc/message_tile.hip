// Typed edge message FUSED with the neighbour sum (EdgeNetwork -> AdjMsgAgg as one kernel) at nf = mf = 64:
//   out[i] = sum_{e in row i} A[type e] . h[src e]
// replaces: mpnn_functions/message/edge_network.py:40,52 (per-pair product) composed with
//           mpnn_functions/message_aggregators/adjacent_message_agg.py:18 (the neighbour sum), i.e. what
//           edge_network.py:50-51 computes as one bmm.  No (E, mf) message tensor exists in HBM.
//
// Structure (one persistent 8-wave block per CU):
//   * a block walks MOLECULE-ALIGNED tiles of at most 128 atoms (edges never leave a molecule, so every source row of a
//     tile's edges is one of the tile's own rows).  The tile's h rows are read from HBM once, coalesced, and parked in
//     LDS as two fp16 images (hi / scaled lo, see "Math"); the K bond-type matrices stay resident in LDS as fp16 image
//     pairs for the whole kernel;
//   * the tile plan (graph.py::TilePlan) sorts the tile's atoms by their per-type in-degree and deals them in blocks of
//     16; a ROW-TILE is "the rank-th incoming type-k edge of each atom of one block", so row m of its contraction
//     [16 x 64] . A_k^T [64 x 64] (v_mfma_f32_16x16x32_f16) IS destination atom m: successive row-tiles of a block -- any
//     type, any rank -- accumulate into the same registers and those registers are the output rows.  Nothing is scattered,
//     summed across lanes or split again; an atom without a rank-th type-k edge reads a row of zeros.  (Measured on the
//     way here: per-edge row-tiles + LDS float atomics 3.5 ms, + rank-ordered LDS read-modify-write 0.64 ms, + a second
//     MFMA with a one-hot incidence operand 0.67 ms, all bound by the vector instructions around the MFMAs.)
//   * wave PAIR p owns blocks 2p and 2p+1 of the tile (the plan labels the blocks so that these are the p-th heaviest
//     and the p-th lightest: with one block per wave the heaviest block's wave ran 6.7k cycles per tile and the lightest
//     1.6k); the two waves of a pair run the same row-tiles and split the 64 output features.  A operands are gathered from the LDS image by source row (ds_read_b128,
//     no conversion in the loop), the type's matrix fragments stay in registers across its row-tiles and the next
//     type's are fetched behind the last MFMA of the current one; slot words run two row-tiles ahead, gathered
//     fragments one, in two register sets used alternately (no copies);
//   * sum order inside an output row: types ascending, edge order within a type (accumulator order) -- deterministic;
//   * at the end of a tile the accumulators are scaled back, transposed through LDS and stored as 128-byte half rows
//     at their atoms' places.  The next tile's h rows, slot words and atom list are in registers by then.
//
// Math ("fp16x3"): the tile's h rows share one power-of-two scale s_h and the K matrices one scale s_A (max |x| lands
// in [2^14, 2^15): nothing overflows fp16, and entries down to 2^-28 of the largest keep their full precision); a scaled
// operand is split as x*s = hi + lo * 2^-11 with hi = fp16(x*s), lo = fp16((x*s - hi) * 2^11): 22 significant bits, both
// pieces normal fp16 numbers.  A product uses three MFMAs, hi*hi into one accumulator and hi*lo + lo*hi into a second
// one folded in with 2^-11 at the end; every partial product is exact in the fp32 accumulators.  Dropped: lo*lo
// (<= 2^-22 relative).  The scales are undone, exactly, when the accumulators are stored.
#include "common.h"

namespace mpnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int MT_TV = 128;          // atoms per tile (upper bound); LDS image row MT_TV is all zeros
constexpr int MT_F = 64;
constexpr int MT_OSTR = 68;         // floats per row of the out transposition tile
constexpr int MT_KMAX = 4;          // bond-type matrices resident in LDS (16 KB each as an fp16 image pair)
constexpr int MT_RTMAX = 16;        // row-tiles per block (their slot words are parked in LDS)
constexpr int MT_IMG = MT_F * MT_F * 2;          // one fp16 piece of one matrix: 8 KB
constexpr int MT_HT = (MT_TV + 1) * MT_F * 2;    // one fp16 piece of the h tile + the zero row

__host__ __device__ constexpr int mt_lds_bytes(int K) {
    return K * 2 * MT_IMG + 2 * MT_HT + 8 * 16 * MT_OSTR * 4 + 8 * 16 * MT_RTMAX * 4 + MT_TV * 4 + 256;
}

// power-of-two scale that puts `maxabs` into [2^14, 2^15), and its inverse
__device__ __forceinline__ void pow2_scale(float maxabs, float& scale, float& inv) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    e = e < 20 ? 20 : e;
    scale = __int_as_float((268 - e) << 23);
    inv = __int_as_float((e - 14) << 23);
}

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

#ifdef MT_STAMP   // diagnostic build only (-DMT_STAMP): per-phase cycle sums of one block, read by tools/_stamp.py
__device__ unsigned long long g_mt_stamps[16];
#define MT_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#else
#define MT_T(var)
#endif

__global__ void __launch_bounds__(512, 2) message_sum_tile_kernel(
    const float* __restrict__ h, const float* __restrict__ A, const int32_t* __restrict__ tile_rec,
    const int32_t* __restrict__ tile_atom, const int32_t* __restrict__ slots, float* __restrict__ out, int num_tiles,
    int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv >> 1, half = wv & 1;               // my pair's blocks are 2 pr and 2 pr + 1; my 32 output features
    const int q16 = lane & 15, g = lane >> 4;
    auto swz = [](int n) { return (n >> 1) & 7; };

    const int HT_OFF = K * 2 * MT_IMG;
    const int OW_OFF = HT_OFF + 2 * MT_HT;               // out transposition tiles, one per wave
    const int SL_OFF = OW_OFF + 8 * 16 * MT_OSTR * 4;    // slot words, one region per wave
    const int AT_OFF = SL_OFF + 8 * 16 * MT_RTMAX * 4;   // atom id of every sorted position of the tile
    const int RED_OFF = AT_OFF + MT_TV * 4;              // partial maxima (8 floats for A, 8 for the tile)
    float* outw = reinterpret_cast<float*>(smem + OW_OFF) + pr * (32 * MT_OSTR);     // the pair's two blocks: 32 rows
    int* slw_mine = reinterpret_cast<int*>(smem + SL_OFF) + wv * (16 * MT_RTMAX);     // block wv: I park its slot words
    int* slw_pair = reinterpret_cast<int*>(smem + SL_OFF) + 2 * pr * (16 * MT_RTMAX);
    int* atoms = reinterpret_cast<int*>(smem + AT_OFF);
    float* red = reinterpret_cast<float*>(smem + RED_OFF);

    // ---------------------------------------------------------------- matrices -> resident fp16 image pairs
    // image[n][kk] = A_k[n][kk] (row n = output feature, kk = input feature contiguous): an MFMA operand fragment for
    // lane (feature n, k-group) is one 16-byte read; 16-byte chunks XOR-swizzled by the row.  One scale for all K.
    float a_inv;
    {
        f32x4 a4[MT_KMAX][2];
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < MT_KMAX; ++k) {
            if (k < K) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    a4[k][j] = *reinterpret_cast<const f32x4*>(A + (int64_t)k * MT_F * MT_F + 4 * (tid + 512 * j));
#pragma unroll
                    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(a4[k][j][u]));
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) red[wv] = mx;
        // the zero row of the h image (both pieces): 128 bytes each
        if (tid < 16) {
            const f16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<f16x4*>(smem + HT_OFF + MT_TV * 128 + 8 * tid) = z;
            *reinterpret_cast<f16x4*>(smem + HT_OFF + MT_HT + MT_TV * 128 + 8 * tid) = z;
        }
        __syncthreads();
        mx = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, red[u]);
        float sc;
        pow2_scale(mx, sc, a_inv);
#pragma unroll
        for (int k = 0; k < MT_KMAX; ++k) {
            if (k < K) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int flat4 = tid + 512 * j;             // float4 index inside the 64 x 64 matrix
                    const int n = flat4 >> 4, c4 = flat4 & 15;
                    f16x4 hi, lo;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        _Float16 a, b;
                        split_f16(a4[k][j][u] * sc, a, b);
                        hi[u] = a;
                        lo[u] = b;
                    }
                    const int off = k * 2 * MT_IMG + n * 128 + (((c4 >> 1) ^ swz(n)) << 4) + ((c4 & 1) << 3);
                    *reinterpret_cast<f16x4*>(smem + off) = hi;
                    *reinterpret_cast<f16x4*>(smem + off + MT_IMG) = lo;
                }
            }
        }
    }

    // ---------------------------------------------------------------- tile loop
    // tile record: (first atom, atoms, first row-tile of block 0..7, end, ...) -- one scalar load per tile, fetched two
    // tiles ahead; h rows, the blocks' slot words and the atom list are fetched ONE tile ahead into registers and
    // parked in LDS at the top of their tile, so nothing in the row-tile loop waits on global memory.
    struct Rec { int a0, n, r0, r1, r2; };   // row-tiles of block 2 pr: [r0, r1), of block 2 pr + 1: [r1, r2)
    auto load_rec = [&](int t) {
        const int32_t* p = tile_rec + 16 * (int64_t)t;
        Rec r;
        r.a0 = p[0];
        r.n = p[1];
        r.r0 = p[2 + 2 * pr];
        r.r1 = p[3 + 2 * pr];
        r.r2 = p[4 + 2 * pr];
        return r;
    };
    // staging thread (tid, j): float4 number tid + 512 j of the tile = row (tid >> 4) + 32 j, columns 4 (tid & 15) ..
    const int srow = tid >> 4, sc4 = tid & 15;
    f32x4 stage[4];
    int slotreg[MT_RTMAX / 4];
    int atomreg = -1;
    auto stage_load = [&](const Rec& r, int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = srow + 32 * j;
            const int rr = row < r.n ? row : 0;                      // clamped: loads stay unconditional
            stage[j] = *reinterpret_cast<const f32x4*>(h + (int64_t)(r.a0 + rr) * MT_F + 4 * sc4);
        }
        const int rb = half ? r.r1 : r.r0;                            // I fetch block wv's slot words
        const int nw = 16 * ((half ? r.r2 : r.r1) - rb);
#pragma unroll
        for (int j = 0; j < MT_RTMAX / 4; ++j) {
            if (64 * j < nw) {                                        // wave-uniform
                const int i = 64 * j + lane;
                slotreg[j] = slots[(int64_t)16 * rb + (i < nw ? i : 0)];
            }
        }
        if (tid < MT_TV) atomreg = tile_atom[(int64_t)t * MT_TV + tid];
    };
    auto stage_max = [&]() {                                          // this wave's part of max |h| over the tile
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(stage[j][u]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) red[8 + wv] = mx;
    };
    auto stage_write = [&](const Rec& r) -> float {                   // returns the tile's inverse scale
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, red[8 + u]);
        float sc, inv;
        pow2_scale(mx, sc, inv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = srow + 32 * j;
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                _Float16 a, b;
                split_f16(stage[j][u] * sc, a, b);
                hi[u] = a;
                lo[u] = b;
            }
            const int off = HT_OFF + row * 128 + (((sc4 >> 1) ^ swz(row)) << 4) + ((sc4 & 1) << 3);
            *reinterpret_cast<f16x4*>(smem + off) = hi;
            *reinterpret_cast<f16x4*>(smem + off + MT_HT) = lo;
        }
        const int nw = 16 * (half ? r.r2 - r.r1 : r.r1 - r.r0);
#pragma unroll
        for (int j = 0; j < MT_RTMAX / 4; ++j)
            if (64 * j < nw) slw_mine[64 * j + lane] = slotreg[j];
        if (tid < MT_TV) atoms[tid] = atomreg;
        return inv;
    };

    const int G = gridDim.x;
    int t = blockIdx.x;
    Rec cur = load_rec(t < num_tiles ? t : 0);
    Rec nxt = load_rec(t + G < num_tiles ? t + G : 0);
    if (t < num_tiles) stage_load(cur, t);
    for (; t < num_tiles; t += G) {
        MT_T(t0);
        stage_max();
        MT_T(t0b);
        __syncthreads();                    // every wave is done with the previous tile's LDS data; maxima are in
        MT_T(t1);
        const float h_inv = stage_write(cur);
        MT_T(t1b);
        __syncthreads();
        MT_T(t2);
        const Rec nn = load_rec(t + 2 * G < num_tiles ? t + 2 * G : 0);
        if (t + G < num_tiles) stage_load(nxt, t + G);           // in flight during this tile's contractions
        MT_T(t3);

        // slot word (graph.py::TilePlan): source row | valid << 14 | type << 16; an empty slot names the zero row.
        // Lane (q16, g) gathers k-group g of the source row of slot q16 = destination atom q16 of the block.
        auto gather = [&](int word, f16x8 (&ah)[2], f16x8 (&al)[2]) {
            const int src = word & 0xff;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = HT_OFF + src * 128 + (((4 * s + g) ^ swz(src)) << 4);
                ah[s] = *reinterpret_cast<const f16x8*>(smem + off);
                al[s] = *reinterpret_cast<const f16x8*>(smem + off + MT_HT);
            }
        };
        f16x8 bh[2][2], bl[2][2];                                     // the running type's matrix, my two column tiles
        int cur_k = -1;
        auto load_matrix = [&](int k) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int nn2 = 16 * (2 * half + c) + q16;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = k * 2 * MT_IMG + nn2 * 128 + (((4 * s + g) ^ swz(nn2)) << 4);
                    bh[c][s] = *reinterpret_cast<const f16x8*>(smem + off);
                    bl[c][s] = *reinterpret_cast<const f16x8*>(smem + off + MT_IMG);
                }
            }
        };
        // one block: its row-tiles (type-major) accumulate into chh / cx = [16 atoms x my 32 features], hi*hi and cross
        // terms.  Slot words run two row-tiles ahead, gathered fragments one, in two register sets used alternately.
        auto run_block = [&](const int* slw, int nrt, f32x4 (&chh)[2], f32x4 (&cx)[2]) {
            auto my_word = [&](int i) { return slw[16 * (i < nrt ? i : 0) + q16]; };
            auto contract = [&](const f16x8 (&ah)[2], const f16x8 (&al)[2]) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        chh[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bh[c][s], chh[c], 0, 0, 0);
                        cx[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bl[c][s], cx[c], 0, 0, 0);
                        cx[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], bh[c][s], cx[c], 0, 0, 0);
                    }
            };
            // the next row-tile's matrix, if its type differs (wave-uniform), goes behind the last MFMA that read this one
            auto next_matrix = [&](int word, bool more) {
                const int k_n = __builtin_amdgcn_readfirstlane((word >> 16) & 15);
                if (k_n != cur_k && more) {
                    cur_k = k_n;
                    load_matrix(k_n);
                }
            };
            if (nrt <= 0) return;
            f16x8 fah[2], fal[2], fbh[2], fbl[2];
            const int w0 = my_word(0);
            int wn1 = my_word(1);
            next_matrix(w0, true);
            gather(w0, fah, fal);
            for (int i = 0; i < nrt; i += 2) {
                const int wn2 = my_word(i + 2);
                gather(wn1, fbh, fbl);                                // row-tile i + 1 (a harmless re-read at the end)
                __builtin_amdgcn_sched_barrier(0);
                contract(fah, fal);                                   // row-tile i
                __builtin_amdgcn_sched_barrier(0);
                next_matrix(wn1, i + 1 < nrt);
                if (i + 1 >= nrt) break;
                const int wn3 = my_word(i + 3);
                gather(wn2, fah, fal);                                // row-tile i + 2
                __builtin_amdgcn_sched_barrier(0);
                contract(fbh, fbl);                                   // row-tile i + 1
                __builtin_amdgcn_sched_barrier(0);
                next_matrix(wn2, i + 2 < nrt);
                wn1 = wn3;
            }
        };
        f32x4 chh0[2], cx0[2], chh1[2], cx1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            chh0[c] = cx0[c] = chh1[c] = cx1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        run_block(slw_pair, cur.r1 - cur.r0, chh0, cx0);
        run_block(slw_pair + 16 * MT_RTMAX, cur.r2 - cur.r1, chh1, cx1);
        MT_T(t4);
        // ---- my 32 columns of the pair's 32 out rows: accumulators (scales undone) -> the pair's LDS tile (transposition)
        // -> HBM as 128-byte half rows at the atoms' places.  (c, r) = out[atom of row 4 g + r][32 half + 16 c + q16]
        const float f = h_inv * a_inv, fx = f * (1.0f / 2048.0f);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                outw[(4 * g + r) * MT_OSTR + 32 * half + 16 * c + q16] = chh0[c][r] * f + cx0[c][r] * fx;
                outw[(16 + 4 * g + r) * MT_OSTR + 32 * half + 16 * c + q16] = chh1[c][r] * f + cx1[c][r] * fx;
            }
        const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = 8 * it + r8;                              // rows 0..15: block 2 pr, 16..31: block 2 pr + 1
            const int atom = atoms[32 * pr + row];
            const f32x4 v = *reinterpret_cast<const f32x4*>(outw + row * MT_OSTR + 32 * half + 4 * c8);
            if (atom >= 0)
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (int64_t)atom * MT_F + 32 * half + 4 * c8));
        }
        cur = nxt;
        nxt = nn;
#ifdef MT_STAMP
        MT_T(t5);
        if (blockIdx.x == 3 && lane == 0) {
            unsigned long long* q = g_mt_stamps + 8 * (wv == 0 ? 0 : 1);
            if (wv == 0 || wv == 7) {
                atomicAdd(q + 0, t0b - t0); atomicAdd(q + 1, t1 - t0b); atomicAdd(q + 2, t1b - t1); atomicAdd(q + 3, t2 - t1b);
                atomicAdd(q + 4, t3 - t2); atomicAdd(q + 5, t4 - t3); atomicAdd(q + 6, t5 - t4); atomicAdd(q + 7, 1ull);
            }
        }
#endif
    }
}

}  // namespace mpnn

using namespace mpnn;

/* Greedy packing of whole molecules into tiles of at most `max_atoms` atoms (host arrays; the plan is built once per
 * batch, next to the CSR).  Returns the number of tiles, or -1 when a molecule is larger than a tile. */
extern "C" int64_t mpnn_plan_tiles_host(const int32_t* graph_ptr, int64_t G, int max_atoms, int32_t* tile_ptr) {
    if (!graph_ptr || !tile_ptr || G < 0 || max_atoms <= 0) return -1;
    int64_t nt = 0;
    tile_ptr[0] = graph_ptr[0];
    int32_t start = graph_ptr[0];
    for (int64_t m = 0; m < G; ++m) {
        const int32_t sz = graph_ptr[m + 1] - graph_ptr[m];
        if (sz > max_atoms || sz < 0) return -1;
        if (graph_ptr[m + 1] - start > max_atoms) {               // molecule m opens a new tile
            tile_ptr[++nt] = graph_ptr[m];
            start = graph_ptr[m];
        }
    }
    if (graph_ptr[G] > start || nt == 0) tile_ptr[++nt] = graph_ptr[G];
    return nt;
}

#ifdef MT_STAMP
extern "C" int mpnn_debug_mt_stamps(unsigned long long* host16, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mt_stamps), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_mt_stamps), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int mpnn_message_aggregate_tile_atoms(void) { return MT_TV; }
extern "C" int mpnn_message_aggregate_max_types(void) { return MT_KMAX; }
extern "C" int mpnn_message_aggregate_max_row_tiles(void) { return MT_RTMAX; }

extern "C" int mpnn_message_aggregate_f32(const float* h, const float* A, const int32_t* tile_rec, const int32_t* tile_atom,
                                          const int32_t* slots, float* out, int64_t V, int64_t num_tiles, int K, int nf,
                                          int mf, void* stream) {
    MPNN_REQUIRE(nf == MT_F && mf == MT_F, "mpnn_message_aggregate_f32: nf = mf = 64 only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(K >= 1 && K <= MT_KMAX, "mpnn_message_aggregate_f32: 1 <= K <= %d bond types (got %d)", MT_KMAX, K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 24), "mpnn_message_aggregate_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && tile_rec && tile_atom && slots && out, "mpnn_message_aggregate_f32: NULL buffer");
    const size_t lds = (size_t)mt_lds_bytes(K);
    static const hipError_t attr = [] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)message_sum_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mt_lds_bytes(MT_KMAX));
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    int64_t blocks = 256;                                             // one block per CU
    if (blocks > num_tiles) blocks = num_tiles;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(message_sum_tile_kernel, dim3((unsigned)blocks), dim3(512), lds, s, h, A, tile_rec, tile_atom, slots,
                       out, (int)num_tiles, K);
    return launch_status("mpnn_message_aggregate_f32");
}
