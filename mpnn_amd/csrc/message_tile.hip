// Typed edge message FUSED with the neighbour sum (EdgeNetwork -> AdjMsgAgg as one kernel) at nf = mf = 64:
//   out[i] = sum_{e in row i} w[e] * A[type e] . h[src e]
// replaces: mpnn_functions/message/edge_network.py:40,52 (per-pair product) composed with
//           mpnn_functions/message_aggregators/adjacent_message_agg.py:18 (the neighbour sum), i.e. what
//           edge_network.py:50-51 computes as one bmm.  No (E, mf) message tensor exists in HBM.
//
// Structure (one persistent 4-wave block per CU):
//   * a block walks MOLECULE-ALIGNED tiles of at most 128 atoms (edges never leave a molecule, so every source row of a
//     tile's edges is one of the tile's own rows).  The tile's h rows are read from HBM once, coalesced, and parked in
//     LDS as two fp16 images (hi / scaled lo, see "Math"); the K bond-type matrices stay resident in LDS as fp16 image
//     pairs for the whole kernel;
//   * wave q owns sub-tile q of the tile (a quarter of its atoms) and that sub-tile's edges.  The tile plan
//     (graph.py::TilePlan) lists those edges grouped by bond type in ROW-TILES of 16 slots, padded per type; a row-tile
//     is one dense contraction [16 edges x 64] . A_k^T [64 x 64] on v_mfma_f32_16x16x32_f16, its A operand gathered
//     from the LDS image by source row (ds_read_b128, no conversion in the loop), its B operand (the type's matrix)
//     held in registers across the type's row-tiles;
//   * the 16 message rows of a row-tile are added to the wave's PRIVATE out tile in LDS (ds_add_f32 to row dst(e));
//     one wave issues all adds to a row in program order, so the sum order is fixed: types ascending, edge order within
//     a type (deterministic, unlike a cross-wave reduction);
//   * the wave then streams its out rows to HBM (whole 256-B rows) and re-zeroes them.  The next tile's h rows are
//     already in registers by then (issued before the current tile's contractions).
//
// Math ("fp16x3"): every fp32 operand x is scaled by a power of two s (per h row / per matrix: max |x| lands in
// [2^14, 2^15), so nothing overflows fp16 and the small end keeps 29 binades) and split as x*s = hi + lo * 2^-11 with
// hi = fp16(x*s), lo = fp16((x*s - hi) * 2^11): 22 significant bits, both pieces normal fp16 numbers.  A product uses
// three MFMAs, hi*hi into one accumulator and hi*lo + lo*hi into a second one that is folded in with 2^-11; every partial
// product is exact in the fp32 accumulator.  Dropped: lo*lo (<= 2^-22 relative).  The scales are undone, exactly, in the
// epilogue factor of each row.
#include "common.h"

namespace mpnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int MT_TV = 128;          // atoms per tile (upper bound)
constexpr int MT_F = 64;
constexpr int MT_OSTR = 80;         // floats per out row in LDS (16 of padding: adjacent rows start 16 banks apart)
constexpr int MT_OROWS = 33;        // 32 destination rows + one sink row for padding slots
constexpr int MT_KMAX = 5;
constexpr int MT_IMG = MT_F * MT_F * 2;          // one fp16 piece of one matrix: 8 KB
constexpr int MT_HT = MT_TV * MT_F * 2;          // one fp16 piece of the h tile: 16 KB
constexpr int MT_SLOT_SINK = 32 << 8;            // padding slot word: source row 0, destination = sink row

__host__ __device__ constexpr int mt_lds_bytes(int K) {
    return K * 2 * MT_IMG + 2 * MT_HT + MT_TV * 4 + 64 + 4 * MT_OROWS * MT_OSTR * 4 + 4 * 4 * MT_KMAX;
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

template <bool WEIGHTED>
__global__ void __launch_bounds__(256, 1) message_sum_tile_kernel(
    const float* __restrict__ h, const float* __restrict__ A, const float* __restrict__ w,
    const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ rt_ptr, const int32_t* __restrict__ rt_type,
    const int32_t* __restrict__ slots, const int32_t* __restrict__ slot_eid, float* __restrict__ out, int num_tiles,
    int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q16 = lane & 15, g = lane >> 4;
    auto swz = [](int n) { return (n >> 1) & 7; };

    const int HT_OFF = K * 2 * MT_IMG;
    const int RS_OFF = HT_OFF + 2 * MT_HT;               // float inv-scale per tile row
    const int AS_OFF = RS_OFF + MT_TV * 4;               // float inv-scale per type (16 floats reserved)
    const int OW_OFF = AS_OFF + 64;
    const int RED_OFF = OW_OFF + 4 * MT_OROWS * MT_OSTR * 4;   // [K][4 waves] partial maxima
    float* rowscale = reinterpret_cast<float*>(smem + RS_OFF);
    float* ascale = reinterpret_cast<float*>(smem + AS_OFF);
    float* outw = reinterpret_cast<float*>(smem + OW_OFF) + wv * (MT_OROWS * MT_OSTR);
    float* red = reinterpret_cast<float*>(smem + RED_OFF);

    // ---------------------------------------------------------------- matrices -> resident fp16 image pairs
    // image[n][kk] = A_k[n][kk] (row n = output feature, kk = input feature contiguous): the MFMA's B fragment
    // B[kk][n] for lane (column n, k-group) is one 16-byte read; 16-byte chunks XOR-swizzled by the row.
    {
        f32x4 a4[MT_KMAX][4];
#pragma unroll
        for (int k = 0; k < MT_KMAX; ++k) {
            if (k < K) {
                float mx = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a4[k][j] = *reinterpret_cast<const f32x4*>(A + (int64_t)k * MT_F * MT_F + 4 * (tid + 256 * j));
#pragma unroll
                    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(a4[k][j][u]));
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if (lane == 0) red[k * 4 + wv] = mx;
            }
        }
        // zero this wave's out tile while the partial maxima land
        for (int i = lane; i < MT_OROWS * MT_OSTR / 4; i += 64)
            reinterpret_cast<f32x4*>(outw)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MT_KMAX; ++k) {
            if (k < K) {
                const float mx = fmaxf(fmaxf(red[k * 4], red[k * 4 + 1]), fmaxf(red[k * 4 + 2], red[k * 4 + 3]));
                float sc, inv;
                pow2_scale(mx, sc, inv);
                if (tid == 0) ascale[k] = inv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int flat4 = tid + 256 * j;             // float4 index inside the 64 x 64 matrix
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
    // staging thread (tid, j): float4 number tid + 256 j of the tile = row (tid >> 4) + 16 j, columns 4 (tid & 15) ..
    const int srow = tid >> 4, sc4 = tid & 15;
    f32x4 stage[8];
    auto stage_load = [&](int t) {
        const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = srow + 16 * j;
            const int rr = row < n ? row : 0;                        // clamped: loads stay unconditional
            stage[j] = *reinterpret_cast<const f32x4*>(h + (int64_t)(a0 + rr) * MT_F + 4 * sc4);
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = srow + 16 * j;
            float mx = fmaxf(fmaxf(fabsf(stage[j][0]), fabsf(stage[j][1])), fmaxf(fabsf(stage[j][2]), fabsf(stage[j][3])));
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));     // the 16 lanes that hold this row
            float sc, inv;
            pow2_scale(mx, sc, inv);
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
            if (sc4 == 0) rowscale[row] = inv;
        }
    };

    int t = blockIdx.x;
    if (t < num_tiles) stage_load(t);
    for (; t < num_tiles; t += gridDim.x) {
        __syncthreads();                    // every wave is done gathering from the previous tile's image
        stage_write();
        __syncthreads();
        const int tn = t + gridDim.x;
        if (tn < num_tiles) stage_load(tn);  // in flight during this tile's contractions

        const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
        const int ss = (n + 3) >> 2;                                  // atoms per sub-tile (<= 32)
        const int rt0 = rt_ptr[4 * t + wv], rt1 = rt_ptr[4 * t + wv + 1];

        // slot words of row-tile `rt`: mine (slot q16: the source row I gather) and the four whose rows my accumulator
        // registers hold (slots 4g .. 4g+3: destination rows and source rows' scales)
        int sw_n = MT_SLOT_SINK;
        i32x4 sw4_n = {MT_SLOT_SINK, MT_SLOT_SINK, MT_SLOT_SINK, MT_SLOT_SINK};
        i32x4 eid4_n = {-1, -1, -1, -1};
        if (rt0 < rt1) {
            sw_n = slots[rt0 * 16 + q16];
            sw4_n = *reinterpret_cast<const i32x4*>(slots + rt0 * 16 + 4 * g);
            if (WEIGHTED) eid4_n = *reinterpret_cast<const i32x4*>(slot_eid + rt0 * 16 + 4 * g);
        }
        int cur_k = -1;
        f16x8 bh[4][2], bl[4][2];
        float ainv = 0.f;
        for (int rt = rt0; rt < rt1; ++rt) {
            const int sw = sw_n;
            const i32x4 sw4 = sw4_n;
            const i32x4 eid4 = eid4_n;
            const int rtn = rt + 1 < rt1 ? rt + 1 : rt;             // clamped prefetch of the next row-tile's words
            sw_n = slots[rtn * 16 + q16];
            sw4_n = *reinterpret_cast<const i32x4*>(slots + rtn * 16 + 4 * g);
            if (WEIGHTED) eid4_n = *reinterpret_cast<const i32x4*>(slot_eid + rtn * 16 + 4 * g);
            const int k = __builtin_amdgcn_readfirstlane(rt_type[rt]);
            if (k != cur_k) {                                          // wave-uniform: a few times per sub-tile
                cur_k = k;
                ainv = ascale[k];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int nn = 16 * ct + q16;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int off = k * 2 * MT_IMG + nn * 128 + (((4 * s + g) ^ swz(nn)) << 4);
                        bh[ct][s] = *reinterpret_cast<const f16x8*>(smem + off);
                        bl[ct][s] = *reinterpret_cast<const f16x8*>(smem + off + MT_IMG);
                    }
                }
            }
            // A fragments: my slot's source row, k-group g of each K = 32 step
            const int src = sw & 0xff;
            f16x8 ah[2], al[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = HT_OFF + src * 128 + (((4 * s + g) ^ swz(src)) << 4);
                ah[s] = *reinterpret_cast<const f16x8*>(smem + off);
                al[s] = *reinterpret_cast<const f16x8*>(smem + off + MT_HT);
            }
            // epilogue factors of my four accumulator rows: inverse scales of the source row and the matrix (x weight)
            float f[4];
            int drow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int word = sw4[r];
                drow[r] = (word >> 8) & 0x3f;
                f[r] = rowscale[word & 0xff] * ainv;
                if (WEIGHTED) {
                    const int e = eid4[r];
                    f[r] *= e >= 0 ? w[e] : 0.f;
                }
            }
            f32x4 chh[4], cx[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                chh[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                cx[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    chh[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bh[ct][s], chh[ct], 0, 0, 0);
                    cx[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], bl[ct][s], cx[ct], 0, 0, 0);
                    cx[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], bh[ct][s], cx[ct], 0, 0, 0);
                }
            }
            // accumulator register r of column tile ct = (slot 4g + r, feature 16 ct + q16)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* orow = outw + drow[r] * MT_OSTR + q16;
                const float fx = f[r] * (1.0f / 2048.0f);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const float v = chh[ct][r] * f[r] + cx[ct][r] * fx;
                    __hip_atomic_fetch_add(orow + 16 * ct, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        // ---- this wave's out rows -> HBM, then back to zero (sink row included).  LDS operations of one wave
        // complete in order, so the reads below see every add above.
        const int rows = min(ss, n - wv * ss);                       // may be <= 0 for a short tile's last waves
        const int64_t obase = (int64_t)(a0 + wv * ss) * MT_F;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 4 * it + g;
            f32x4* p = reinterpret_cast<f32x4*>(outw + row * MT_OSTR + 4 * q16);
            const f32x4 v = *p;
            *p = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + obase + (int64_t)row * MT_F + 4 * q16));
        }
        if (lane < 16) reinterpret_cast<f32x4*>(outw + 32 * MT_OSTR)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
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

extern "C" int mpnn_message_aggregate_tile_atoms(void) { return MT_TV; }
extern "C" int mpnn_message_aggregate_max_types(void) { return MT_KMAX; }

extern "C" int mpnn_message_aggregate_f32(const float* h, const float* A, const float* w, const int32_t* tile_ptr,
                                          const int32_t* rt_ptr, const int32_t* rt_type, const int32_t* slots,
                                          const int32_t* slot_eid, float* out, int64_t V, int64_t num_tiles, int K,
                                          int nf, int mf, void* stream) {
    MPNN_REQUIRE(nf == MT_F && mf == MT_F, "mpnn_message_aggregate_f32: nf = mf = 64 only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(K >= 1 && K <= MT_KMAX, "mpnn_message_aggregate_f32: 1 <= K <= %d bond types (got %d)", MT_KMAX, K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 30), "mpnn_message_aggregate_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && tile_ptr && rt_ptr && rt_type && slots && out, "mpnn_message_aggregate_f32: NULL buffer");
    MPNN_REQUIRE(!w || slot_eid, "mpnn_message_aggregate_f32: weights need the plan's slot_eid array");
    const size_t lds = (size_t)mt_lds_bytes(K);
    static const hipError_t attr = [] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)message_sum_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, mt_lds_bytes(MT_KMAX));
        opt_in_((const void*)message_sum_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, mt_lds_bytes(MT_KMAX));
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    int64_t blocks = 256;                                             // one block per CU
    if (blocks > num_tiles) blocks = num_tiles;
    hipStream_t s = (hipStream_t)stream;
    if (w)
        hipLaunchKernelGGL(message_sum_tile_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, h, A, w, tile_ptr, rt_ptr,
                           rt_type, slots, slot_eid, out, (int)num_tiles, K);
    else
        hipLaunchKernelGGL(message_sum_tile_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, h, A, w, tile_ptr,
                           rt_ptr, rt_type, slots, slot_eid, out, (int)num_tiles, K);
    return launch_status("mpnn_message_aggregate_f32");
}
