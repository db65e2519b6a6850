// Dense padded batch -> CSR by destination (mpnn_csr_count / mpnn_csr_fill).
//
// One wave64 walks one padded row (b,i): lane l tests pair (b,i,j0+l), a 64-bit ballot gives
// the row's edge count and each edge's rank, so column indices come out in increasing j --
// i.e. exactly adj.nonzero() order.  Integer work, HBM-bound: the row is read once per pass.
#include "common.h"

namespace mpnn {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanChunk = kScanThreads * kScanItems;

__device__ __forceinline__ bool pair_is_edge(const float* __restrict__ adj, const float* __restrict__ bfm,
                                             int64_t pair, int ef) {
    bool on = false;
    if (adj) on = adj[pair] != 0.0f;
    if (bfm && !on) {
        const float* p = bfm + pair * ef;
        for (int f = 0; f < ef; ++f) on |= (p[f] != 0.0f);
    }
    return on;
}

__global__ void __launch_bounds__(256) csr_count_kernel(const float* __restrict__ adj,
                                                        const float* __restrict__ bfm, int64_t rows,
                                                        int cols, int ef, int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = wave; row < rows; row += nwaves) {
        int cnt = 0;
        for (int j0 = 0; j0 < cols; j0 += 64) {
            const int j = j0 + lane;
            const bool on = (j < cols) && pair_is_edge(adj, bfm, row * cols + j, ef);
            cnt += __popcll(__ballot(on));
        }
        if (lane == 0) counts[row] = cnt;
    }
}

// exclusive prefix of `v` over the 256 threads of a block; *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    __shared__ int wave_sum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_sum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < w) base += wave_sum[i];
        tot += wave_sum[i];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) scan_chunk_sums_kernel(const int32_t* __restrict__ counts,
                                                                        int64_t n, int32_t* __restrict__ sums) {
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) s += counts[base + i];
    int total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one block: sums[0..nb) -> exclusive prefix, in place
__global__ void __launch_bounds__(kScanThreads) scan_chunk_offsets_kernel(int32_t* __restrict__ sums, int nb) {
    const int per = (nb + kScanThreads - 1) / kScanThreads;
    const int lo = threadIdx.x * per;
    int s = 0;
    for (int i = lo; i < lo + per && i < nb; ++i) s += sums[i];
    int total;
    int run = block_exclusive_scan(s, &total);
    for (int i = lo; i < lo + per && i < nb; ++i) {
        const int v = sums[i];
        sums[i] = run;
        run += v;
    }
}

// counts (at row_ptr+1) -> inclusive prefix in place; row_ptr[0] = 0
__global__ void __launch_bounds__(kScanThreads) scan_apply_kernel(int32_t* __restrict__ row_ptr, int64_t n,
                                                                   const int32_t* __restrict__ offsets) {
    int32_t* counts = row_ptr + 1;
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = (base + i < n) ? counts[base + i] : 0;
        s += v[i];
    }
    int total;
    int run = block_exclusive_scan(s, &total) + offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        run += v[i];
        if (base + i < n) counts[base + i] = run;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) row_ptr[0] = 0;
}

__global__ void __launch_bounds__(256) csr_fill_kernel(const float* __restrict__ adj, const float* __restrict__ bfm,
                                                       int64_t rows, int cols, int ef,
                                                       const int32_t* __restrict__ row_ptr,
                                                       int32_t* __restrict__ col_idx, float* __restrict__ edge_weight,
                                                       float* __restrict__ edge_feat) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t row = wave; row < rows; row += nwaves) {
        int64_t pos = row_ptr[row];
        const int64_t node0 = (row / cols) * cols;   // first padded atom of this molecule
        for (int j0 = 0; j0 < cols; j0 += 64) {
            const int j = j0 + lane;
            const int64_t pair = row * cols + j;
            const bool on = (j < cols) && pair_is_edge(adj, bfm, pair, ef);
            const unsigned long long m = __ballot(on);
            if (on) {
                const int64_t e = pos + __popcll(m & below);
                col_idx[e] = (int32_t)(node0 + j);
                if (edge_weight) edge_weight[e] = adj ? adj[pair] : 1.0f;
                if (edge_feat)
                    for (int f = 0; f < ef; ++f) edge_feat[e * ef + f] = bfm ? bfm[pair * ef + f] : 0.0f;
            }
            pos += __popcll(m);
        }
    }
}

static int row_grid(int64_t rows) {
    int64_t g = ceil_div(rows, 4);
    if (g > 256 * 32) g = 256 * 32;   // grid-stride beyond 8192 blocks
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mpnn

using namespace mpnn;

extern "C" size_t mpnn_csr_workspace_bytes(int64_t rows) {
    if (rows < 0) return 0;
    return (size_t)(ceil_div(rows, kScanChunk) + 1) * sizeof(int32_t);
}

extern "C" int mpnn_csr_count(const float* adj, const float* bfm, int64_t rows, int cols, int ef, int32_t* row_ptr,
                              void* workspace, size_t workspace_bytes, void* stream) {
    MPNN_REQUIRE(adj || bfm, "mpnn_csr_count: adj and bfm are both NULL");
    MPNN_REQUIRE(rows >= 0 && cols > 0 && row_ptr, "mpnn_csr_count: bad shape rows=%lld cols=%d", (long long)rows, cols);
    MPNN_REQUIRE(!bfm || ef > 0, "mpnn_csr_count: bfm given with ef=%d", ef);
    MPNN_REQUIRE(rows * (int64_t)cols < (1ll << 31), "mpnn_csr_count: more than 2^31 pairs");
    if (workspace_bytes < mpnn_csr_workspace_bytes(rows) || !workspace) {
        set_error("mpnn_csr_count: workspace %zu < %zu", workspace_bytes, mpnn_csr_workspace_bytes(rows));
        return MPNN_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int32_t* sums = (int32_t*)workspace;
    const int nb = (int)ceil_div(rows > 0 ? rows : 1, kScanChunk);
    if (rows > 0)
        hipLaunchKernelGGL(csr_count_kernel, dim3(row_grid(rows)), dim3(256), 0, s, adj, bfm, rows, cols, ef, row_ptr + 1);
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nb), dim3(kScanThreads), 0, s, row_ptr + 1, rows, sums);
    hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(kScanThreads), 0, s, sums, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kScanThreads), 0, s, row_ptr, rows, sums);
    return launch_status("mpnn_csr_count");
}

extern "C" int mpnn_csr_fill(const float* adj, const float* bfm, int64_t rows, int cols, int ef,
                             const int32_t* row_ptr, int32_t* col_idx, float* edge_weight, float* edge_feat,
                             void* stream) {
    MPNN_REQUIRE(adj || bfm, "mpnn_csr_fill: adj and bfm are both NULL");
    MPNN_REQUIRE(rows >= 0 && cols > 0 && row_ptr && col_idx, "mpnn_csr_fill: bad arguments");
    MPNN_REQUIRE(!edge_feat || (bfm && ef > 0), "mpnn_csr_fill: edge_feat requested without bfm");
    if (rows == 0) return MPNN_OK;
    hipLaunchKernelGGL(csr_fill_kernel, dim3(row_grid(rows)), dim3(256), 0, (hipStream_t)stream, adj, bfm, rows, cols,
                       ef, row_ptr, col_idx, edge_weight, edge_feat);
    return launch_status("mpnn_csr_fill");
}
