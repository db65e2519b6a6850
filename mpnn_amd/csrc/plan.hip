// Per-batch index work as purpose-made kernels: the CSR-derived arrays of graph.py::MolGraph (destination list, type order,
// transposed graph) and the work lists of the fused message + sum kernels (TilePlan / WidePlan).
//
// graph.py builds the same arrays from ~60 generic launches (three stable 6 M-key sorts among them): 7-11 ms per NEW batch at
// c2 ... c5, in a loop that -- like the reference's `for batch in DataLoader(...)` (test_lipo.py:157-165) -- sees a new batch
// every step.  A batch of molecules has structure those launches cannot use: atoms of a molecule are adjacent, an edge never
// leaves its molecule, and a tile of whole molecules (<= 256 atoms, a few hundred edges) fits one workgroup's LDS.  So every
// array is built tile by tile, one workgroup per tile, with two global steps in between (a prefix sum over the per-tile
// counts, done by the caller with one library scan).  Results are bit-identical to graph.py's (tests/test_plan_gpu.py); the
// torch builders stay as the CPU path and the fallback for batches that are not separate molecules.
#include <hip/hip_runtime.h>
#include "common.h"

namespace mpnn {
namespace {

constexpr int PL_TV = 256;          // atoms per index tile
constexpr int PL_EMAX = 6144;       // edges of one tile held in LDS (24 KB of ids)
constexpr int PL_KMAX = 16;         // bond types of the type-order kernel

// the pattern key of graph.py::_tile_layout: one field per type, rare types (high ids) lead, high counts first
__device__ __forceinline__ unsigned long long pattern_code(const int* cnt, int K) {
    const int bits = K <= 7 ? 8 : 7, top = (1 << bits) - 1;
    unsigned long long code = 0;
    for (int k = K - 1; k >= 0; --k) code = (code << bits) + (unsigned long long)(top - (cnt[k] < top ? cnt[k] : top));
    return code;
}

// -------------------------------------------------------------------------------------------- index arrays, per tile
// One workgroup per tile (molecule-aligned, <= 256 atoms; its in-edges are the contiguous range [row_ptr[a0], row_ptr[a0 + n])):
//   edge_dst[e]            destination atom of every edge;
//   t_row_ptr / t_eid      the same edges grouped by SOURCE atom, ascending edge id inside a group: every source of a tile's
//                          edges lies in the tile, so the tile's out-edges are the same range and t_row_ptr needs no global scan;
//   hist[t][k]             edges of type k in the tile (for the type order).
// flags[0] |= 1: an edge leaves its tile; flags[0] |= 2: a tile with more than PL_EMAX edges (the caller falls back).
__global__ void __launch_bounds__(256) index_tile_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                         const int32_t* __restrict__ edge_type, const int32_t* __restrict__ tile_ptr,
                                                         int K, int64_t V, int32_t* __restrict__ edge_dst,
                                                         int32_t* __restrict__ t_row_ptr, int32_t* __restrict__ t_eid,
                                                         int32_t* __restrict__ hist, int32_t* flags) {
    __shared__ int cnt_s[PL_TV + 1], start_s[PL_TV + 1], cur_s[PL_TV], hist_s[PL_KMAX];
    __shared__ int list_s[PL_EMAX];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    const int e0 = row_ptr[a0], e1 = row_ptr[a0 + n], ne = e1 - e0;
    cnt_s[tid] = 0;
    cur_s[tid] = 0;
    if (tid < PL_KMAX) hist_s[tid] = 0;
    if (tid == 0) cnt_s[PL_TV] = 0;
    __syncthreads();
    if (tid < n) {
        const int a = a0 + tid;
        for (int e = row_ptr[a]; e < row_ptr[a + 1]; ++e) edge_dst[e] = a;
    }
    bool bad = false;
    for (int e = e0 + tid; e < e1; e += 256) {
        const int j = col_idx[e] - a0;
        if (j < 0 || j >= n) { bad = true; continue; }
        atomicAdd(&cnt_s[j], 1);
        if (K <= PL_KMAX) atomicAdd(&hist_s[edge_type[e]], 1);
    }
    if (bad) atomicOr(flags, 1);
    if (ne > PL_EMAX && tid == 0) atomicOr(flags, 2);
    __syncthreads();
    if (tid == 0) {                                        // (<= 256 entries: a serial scan is a few hundred cycles)
        int s = 0;
        for (int j = 0; j < n; ++j) { start_s[j] = s; s += cnt_s[j]; }
        start_s[n] = s;
    }
    __syncthreads();
    if (tid < n) t_row_ptr[a0 + tid] = e0 + start_s[tid];
    if (a0 + n == V && tid == 0) t_row_ptr[V] = e1;
    if (K <= PL_KMAX && tid < K) hist[(int64_t)t * K + tid] = hist_s[tid];
    if (ne > PL_EMAX) return;                              // block-uniform
    for (int e = e0 + tid; e < e1; e += 256) {
        const int j = col_idx[e] - a0;
        if (j < 0 || j >= n) continue;
        list_s[start_s[j] + atomicAdd(&cur_s[j], 1)] = e;
    }
    __syncthreads();
    if (tid < n) {                                         // ascending edge id inside the group = stable in edge order
        const int s = start_s[tid], c = cnt_s[tid];
        for (int i = 1; i < c; ++i) {
            const int v = list_s[s + i];
            int p = i - 1;
            while (p >= 0 && list_s[s + p] > v) { list_s[s + p + 1] = list_s[s + p]; --p; }
            list_s[s + p + 1] = v;
        }
        for (int i = 0; i < c; ++i) t_eid[e0 + s + i] = list_s[s + i];
    }
}

// type order: edge ids stably sorted by type.  off[k * T + t] = position of the first type-k edge of tile t (an exclusive
// scan of hist in (type, tile) order, by the caller); one wave per tile ranks its edges by ballots.
__global__ void __launch_bounds__(64) order_fill_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_type,
                                                        const int32_t* __restrict__ tile_ptr, const int64_t* __restrict__ off,
                                                        int K, int64_t T, int32_t* __restrict__ order,
                                                        int32_t* __restrict__ type_ptr, int64_t E) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    const int e0 = row_ptr[a0], e1 = row_ptr[a0 + n];
    int64_t base[PL_KMAX];
#pragma unroll
    for (int k = 0; k < PL_KMAX; ++k) base[k] = k < K ? off[(int64_t)k * T + t] : 0;
    if (t == 0 && lane <= K) type_ptr[lane] = lane < K ? (int32_t)off[(int64_t)lane * T] : (int32_t)E;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int eb = e0; eb < e1; eb += 64) {
        const int e = eb + lane;
        const int ty = e < e1 ? edge_type[e] : -1;
#pragma unroll
        for (int k = 0; k < PL_KMAX; ++k) {
            if (k < K) {                                   // (wave-uniform)
                const unsigned long long m = __ballot(ty == k);
                if (ty == k) order[base[k] + __popcll(m & lt)] = e;
                base[k] += __popcll(m);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------- tile plan (width 64)
// graph.py::TilePlan.build, one workgroup per tile of <= 128 atoms.  Count pass: per-atom in-degree by type -> pattern key ->
// position in the tile's (key, atom) order -> block of 16 and row; a block's row-tile need per type = the largest count among
// its atoms; blocks relabelled so that blocks 2p / 2p + 1 are the p-th heaviest / p-th lightest.  Writes the need table
// (tile, block, type), tile_atom and every atom's (block << 4 | row).
__global__ void __launch_bounds__(128) tile_plan_count_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                              const int32_t* __restrict__ edge_type,
                                                              const int32_t* __restrict__ tile_ptr, int K, int32_t* flags,
                                                              int64_t* __restrict__ need_out, int32_t* __restrict__ tile_atom,
                                                              int32_t* __restrict__ atom_slot) {
    __shared__ unsigned long long code_s[128];
    __shared__ int need_s[8][8], load_s[8], newlab_s[8];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    int cnt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt[k] = 0;
    if (tid < n) {
        bool bad = false;
        for (int e = row_ptr[a0 + tid]; e < row_ptr[a0 + tid + 1]; ++e) {
            const int ty = edge_type[e], j = col_idx[e] - a0;
            bad |= j < 0 || j >= n;                        // an edge that leaves its tile: not a batch of separate molecules
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt[k] += ty == k;
        }
        if (bad) atomicOr(flags, 4);
    }
    const unsigned long long code = pattern_code(cnt, K);
    code_s[tid] = code;
    if (tid < 64) need_s[tid >> 3][tid & 7] = 0;
    tile_atom[(int64_t)t * 128 + tid] = -1;
    __syncthreads();
    int pos = 0;
    if (tid < n)
        for (int b = 0; b < n; ++b) pos += code_s[b] < code || (code_s[b] == code && b < tid);
    const int sblk = pos >> 4, row = pos & 15;
    if (tid < n)
        for (int k = 0; k < K; ++k) atomicMax(&need_s[sblk][k], cnt[k]);
    __syncthreads();
    if (tid < 8) {
        int s = 0;
        for (int k = 0; k < K; ++k) s += need_s[tid][k];
        load_s[tid] = s;
    }
    __syncthreads();
    if (tid < 8) {                                         // rank of block tid in the stable descending order of the loads
        int r = 0;
        for (int b = 0; b < 8; ++b) r += load_s[b] > load_s[tid] || (load_s[b] == load_s[tid] && b < tid);
        newlab_s[tid] = r < 4 ? 2 * r : 2 * (7 - r) + 1;
    }
    __syncthreads();
    if (tid < n) {
        const int blk = newlab_s[sblk];
        tile_atom[(int64_t)t * 128 + blk * 16 + row] = a0 + tid;
        atom_slot[a0 + tid] = blk * 16 + row;
    }
    if (tid < 8 * K) {
        const int b = tid / K, k = tid % K;
        need_out[((int64_t)t * 8 + newlab_s[b]) * K + k] = need_s[b][k];
    }
}

// Fill pass: rt_start = exclusive scan of the need table (+ the total).  Slot words as graph.py: (source - tile start) |
// valid << 14 | type << 16; an empty slot reads row `tv` (zeros) and carries its type.
__global__ void __launch_bounds__(128) tile_plan_fill_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                             const int32_t* __restrict__ edge_type, const int32_t* __restrict__ tile_ptr,
                                                             int K, const int64_t* __restrict__ rt_start,
                                                             const int32_t* __restrict__ atom_slot, int32_t* __restrict__ slots,
                                                             int32_t* __restrict__ slot_eid, int32_t* __restrict__ rt_ptr32, int64_t T) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    const int64_t g0 = (int64_t)t * 8 * K;
    if (tid < 8) rt_ptr32[(int64_t)t * 8 + tid] = (int32_t)rt_start[g0 + (int64_t)tid * K];
    if (t == T - 1 && tid == 0) rt_ptr32[T * 8] = (int32_t)rt_start[T * 8 * K];
    // empty pattern of the tile's row-tiles
    for (int g = 0; g < 8 * K; ++g) {
        const int64_t r0 = rt_start[g0 + g], r1 = rt_start[g0 + g + 1];
        const int k = g % K;
        for (int64_t w = r0 * 16 + tid; w < r1 * 16; w += 128) {
            slots[w] = 128 | (k << 16);
            slot_eid[w] = -1;
        }
    }
    __syncthreads();
    if (tid < n) {
        const int a = a0 + tid, sl = atom_slot[a], blk = sl >> 4, row = sl & 15;
        int rk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rk[k] = 0;
        for (int e = row_ptr[a]; e < row_ptr[a + 1]; ++e) {
            const int ty = edge_type[e];
            int r = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { if (ty == k) { r = rk[k]; rk[k] = r + 1; } }
            const int64_t w = (rt_start[g0 + (int64_t)blk * K + ty] + r) * 16 + row;
            slots[w] = (col_idx[e] - a0) | (1 << 14) | (ty << 16);
            slot_eid[w] = e;
        }
    }
}

// -------------------------------------------------------------------------------------------- wide plan (128 / 256)
// graph.py::WidePlan.build, one workgroup per tile of <= 256 atoms, blocks of 32 sorted atoms, no relabelling.
__global__ void __launch_bounds__(256) wide_plan_count_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                              const int32_t* __restrict__ edge_type,
                                                              const int32_t* __restrict__ tile_ptr, int K, int32_t* flags,
                                                              int64_t* __restrict__ need_out, int32_t* __restrict__ tile_atom,
                                                              int32_t* __restrict__ atom_slot) {
    __shared__ unsigned long long code_s[256];
    __shared__ int need_s[8][8];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    int cnt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt[k] = 0;
    if (tid < n) {
        bool bad = false;
        for (int e = row_ptr[a0 + tid]; e < row_ptr[a0 + tid + 1]; ++e) {
            const int ty = edge_type[e], j = col_idx[e] - a0;
            bad |= j < 0 || j >= n;                        // an edge that leaves its tile: not a batch of separate molecules
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt[k] += ty == k;
        }
        if (bad) atomicOr(flags, 4);
    }
    const unsigned long long code = pattern_code(cnt, K);
    code_s[tid] = code;
    if (tid < 64) need_s[tid >> 3][tid & 7] = 0;
    tile_atom[(int64_t)t * 256 + tid] = -1;
    __syncthreads();
    int pos = 0;
    if (tid < n)
        for (int b = 0; b < n; ++b) pos += code_s[b] < code || (code_s[b] == code && b < tid);
    const int blk = pos >> 5;
    if (tid < n) {
        for (int k = 0; k < K; ++k) atomicMax(&need_s[blk][k], cnt[k]);
        tile_atom[(int64_t)t * 256 + pos] = a0 + tid;
        atom_slot[a0 + tid] = pos;
    }
    __syncthreads();
    if (tid < 8 * K) need_out[(int64_t)t * 8 * K + tid] = need_s[tid / K][tid % K];
}

__global__ void __launch_bounds__(256) wide_plan_fill_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                             const int32_t* __restrict__ edge_type, const int32_t* __restrict__ tile_ptr,
                                                             int K, const int64_t* __restrict__ start,
                                                             const int32_t* __restrict__ atom_slot, int16_t* __restrict__ slots,
                                                             int32_t* __restrict__ slot_eid, int32_t* __restrict__ tile_rec,
                                                             int32_t* __restrict__ blk_off) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a0 = tile_ptr[t], n = tile_ptr[t + 1] - a0;
    const int per = 8 * K;
    const int64_t g0 = (int64_t)t * per;
    const int64_t row0 = start[g0], rows = start[g0 + per] - row0;
    if (tid <= per) blk_off[(int64_t)t * (per + 1) + tid] = (int32_t)(start[g0 + tid] - row0);
    if (tid == 0) {
        tile_rec[4 * t] = a0;
        tile_rec[4 * t + 1] = n;
        tile_rec[4 * t + 2] = (int32_t)row0;
        tile_rec[4 * t + 3] = (int32_t)rows;
    }
    for (int64_t w = row0 * 32 + tid; w < (row0 + rows) * 32; w += 256) {
        slots[w] = (int16_t)256;
        slot_eid[w] = -1;
    }
    __syncthreads();
    if (tid < n) {
        const int a = a0 + tid, pos = atom_slot[a], blk = pos >> 5, row = pos & 31;
        int rk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rk[k] = 0;
        for (int e = row_ptr[a]; e < row_ptr[a + 1]; ++e) {
            const int ty = edge_type[e];
            int r = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { if (ty == k) { r = rk[k]; rk[k] = r + 1; } }
            const int64_t w = (start[g0 + (int64_t)blk * K + ty] + r) * 32 + row;
            slots[w] = (int16_t)(col_idx[e] - a0);
            slot_eid[w] = e;
        }
    }
}

}  // namespace
}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_plan_index_tile_atoms(void) { return PL_TV; }
extern "C" int mpnn_plan_index_max_types(void) { return PL_KMAX; }

extern "C" int mpnn_plan_index_tiles(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                                     const int32_t* tile_ptr, int64_t V, int64_t E, int64_t T, int K, int32_t* edge_dst,
                                     int32_t* t_row_ptr, int32_t* t_eid, int32_t* hist, int32_t* flags, void* stream) {
    MPNN_REQUIRE(V >= 0 && E >= 0 && T >= 0 && K >= 1, "mpnn_plan_index_tiles: bad sizes");
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && col_idx && edge_type && tile_ptr && edge_dst && t_row_ptr && t_eid && hist && flags,
                 "mpnn_plan_index_tiles: NULL buffer");
    hipLaunchKernelGGL(index_tile_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, row_ptr, col_idx, edge_type,
                       tile_ptr, K, V, edge_dst, t_row_ptr, t_eid, hist, flags);
    return launch_status("mpnn_plan_index_tiles");
}

extern "C" int mpnn_plan_type_order(const int32_t* row_ptr, const int32_t* edge_type, const int32_t* tile_ptr,
                                    const int64_t* offsets, int64_t E, int64_t T, int K, int32_t* order, int32_t* type_ptr,
                                    void* stream) {
    MPNN_REQUIRE(E >= 0 && T >= 0 && K >= 1 && K <= PL_KMAX, "mpnn_plan_type_order: 1 <= K <= %d (got %d)", PL_KMAX, K);
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && edge_type && tile_ptr && offsets && order && type_ptr, "mpnn_plan_type_order: NULL buffer");
    hipLaunchKernelGGL(order_fill_kernel, dim3((unsigned)T), dim3(64), 0, (hipStream_t)stream, row_ptr, edge_type, tile_ptr,
                       offsets, K, T, order, type_ptr, E);
    return launch_status("mpnn_plan_type_order");
}

extern "C" int mpnn_tile_plan_count(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                                    const int32_t* tile_ptr, int64_t T, int K, int64_t* need, int32_t* tile_atom,
                                    int32_t* atom_slot, int32_t* flags, void* stream) {
    MPNN_REQUIRE(T >= 0 && K >= 1 && K <= 8, "mpnn_tile_plan_count: 1 <= K <= 8 (got %d)", K);
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && col_idx && edge_type && tile_ptr && need && tile_atom && atom_slot && flags,
                 "mpnn_tile_plan_count: NULL buffer");
    hipLaunchKernelGGL(tile_plan_count_kernel, dim3((unsigned)T), dim3(128), 0, (hipStream_t)stream, row_ptr, col_idx, edge_type,
                       tile_ptr, K, flags, need, tile_atom, atom_slot);
    return launch_status("mpnn_tile_plan_count");
}

extern "C" int mpnn_tile_plan_fill(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                                   const int32_t* tile_ptr, int64_t T, int K, const int64_t* rt_start, const int32_t* atom_slot,
                                   int32_t* slots, int32_t* slot_eid, int32_t* rt_ptr, void* stream) {
    MPNN_REQUIRE(T >= 0 && K >= 1 && K <= 8, "mpnn_tile_plan_fill: 1 <= K <= 8 (got %d)", K);
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && col_idx && edge_type && tile_ptr && rt_start && atom_slot && slots && slot_eid && rt_ptr,
                 "mpnn_tile_plan_fill: NULL buffer");
    hipLaunchKernelGGL(tile_plan_fill_kernel, dim3((unsigned)T), dim3(128), 0, (hipStream_t)stream, row_ptr, col_idx, edge_type,
                       tile_ptr, K, rt_start, atom_slot, slots, slot_eid, rt_ptr, T);
    return launch_status("mpnn_tile_plan_fill");
}

extern "C" int mpnn_wide_plan_count(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                                    const int32_t* tile_ptr, int64_t T, int K, int64_t* need, int32_t* tile_atom,
                                    int32_t* atom_slot, int32_t* flags, void* stream) {
    MPNN_REQUIRE(T >= 0 && K >= 1 && K <= 8, "mpnn_wide_plan_count: 1 <= K <= 8 (got %d)", K);
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && col_idx && edge_type && tile_ptr && need && tile_atom && atom_slot && flags,
                 "mpnn_wide_plan_count: NULL buffer");
    hipLaunchKernelGGL(wide_plan_count_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, row_ptr, col_idx, edge_type,
                       tile_ptr, K, flags, need, tile_atom, atom_slot);
    return launch_status("mpnn_wide_plan_count");
}

extern "C" int mpnn_wide_plan_fill(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                                   const int32_t* tile_ptr, int64_t T, int K, const int64_t* start, const int32_t* atom_slot,
                                   int16_t* slots, int32_t* slot_eid, int32_t* tile_rec, int32_t* blk_off, void* stream) {
    MPNN_REQUIRE(T >= 0 && K >= 1 && K <= 8, "mpnn_wide_plan_fill: 1 <= K <= 8 (got %d)", K);
    if (T == 0) return MPNN_OK;
    MPNN_REQUIRE(row_ptr && col_idx && edge_type && tile_ptr && start && atom_slot && slots && slot_eid && tile_rec && blk_off,
                 "mpnn_wide_plan_fill: NULL buffer");
    hipLaunchKernelGGL(wide_plan_fill_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, row_ptr, col_idx, edge_type,
                       tile_ptr, K, start, atom_slot, slots, slot_eid, tile_rec, blk_off);
    return launch_status("mpnn_wide_plan_fill");
}
