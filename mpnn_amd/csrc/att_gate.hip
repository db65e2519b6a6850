// Feature gate of AttEdgeNetwork (reference: mpnn_functions/message/att_edge_network.py:18-21).
//
// The reference concatenates [h_i, e_ij] for every pair, applies Linear(nf+ef -> nf) and a softmax over the FEATURE
// axis.  The Linear splits into an atom part z_atom[i] = W_h h_i + b (one row per atom) and a bond part
// q[k] = W_e e_k (one row per distinct bond-feature row), so the per-edge work is
//     gate[e, :] = softmax_f( z_atom[dst(e), :] + q[type(e), :] )
// -- a streaming kernel bound by writing E x F floats (edges are sorted by destination, so z_atom rows are read
// sequentially).  As library calls this was a row broadcast, a thin GEMM on E rows, an add and a softmax (~8 ms per
// MP step at E = 6 M, F = 128); the backward (softmax backward, per-row segmented sum, per-type column sums) never
// materialises d(logits).
#include "common.h"

namespace mpnn {

// LPR lanes x float4 cover one row of F floats (F <= 4 * LPR); 64 / LPR rows per wave pass.
template <int LPR>
__global__ void __launch_bounds__(256) att_gate_fwd_kernel(const float* __restrict__ z_atom, const float* __restrict__ q,
                                                           const int32_t* __restrict__ dst,
                                                           const int32_t* __restrict__ edge_type,
                                                           float* __restrict__ gate, int64_t E, int F) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, c = 4 * (lane % LPR);
    const bool live = c < F;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    // UN edges per lane group and pass, every load of the pass requested before the first is used: with one edge per
    // pass the kernel was a chain index -> logits -> store per wave (89 % of its wave cycles parked on loads)
    constexpr int UN = 4;
    for (int64_t e0 = wave * (RPW * UN); e0 < E; e0 += nwaves * (RPW * UN)) {
        int64_t e[UN];
        bool ok[UN];
        int d[UN], t[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            e[u] = e0 + u * RPW + sub;
            ok[u] = e[u] < E;
            const int64_t ec = ok[u] ? e[u] : E - 1;
            d[u] = dst[ec];
            t[u] = edge_type[ec];
        }
        f32x4 z[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            z[u] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            if (live) z[u] = *reinterpret_cast<const f32x4*>(z_atom + (int64_t)d[u] * F + c) +
                             *reinterpret_cast<const f32x4*>(q + (int64_t)t[u] * F + c);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float mx = fmaxf(fmaxf(z[u].x, z[u].y), fmaxf(z[u].z, z[u].w));
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            f32x4 p = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                p.x = __expf(z[u].x - mx);
                p.y = __expf(z[u].y - mx);
                p.z = __expf(z[u].z - mx);
                p.w = __expf(z[u].w - mx);
            }
            float sm = p.x + p.y + p.z + p.w;
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) sm += __shfl_xor(sm, o);
            const float inv = 1.0f / sm;
            if (ok[u] && live) __builtin_nontemporal_store(p * inv, reinterpret_cast<f32x4*>(gate + e[u] * F + c));
        }
    }
}

// dz = gate * (dgate - <gate, dgate>);  dz_atom[i] = sum over row i;  dq[type] += dz (LDS partials when K is small)
template <int LPR, int KREG>
__global__ void __launch_bounds__(256) att_gate_bwd_kernel(const float* __restrict__ gate, const float* __restrict__ dgate,
                                                           const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ edge_type,
                                                           float* __restrict__ dz_atom, float* dq, int64_t V, int K,
                                                           int F, int k_lds) {
    extern __shared__ float part[];                        // [k_lds][F] per-block partial of dq (k_lds = 0: none)
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, c = 4 * (lane % LPR);
    const bool live = c < F;
    for (int idx = threadIdx.x; idx < k_lds * F; idx += 256) part[idx] = 0.f;
    __syncthreads();
    // KREG > 0 (K <= KREG): a lane keeps its own partial of every type's dq columns in registers (predicated adds,
    // no atomics in the loop) and adds them to the LDS partial once at the end
    f32x4 qacc[KREG > 0 ? KREG : 1];
#pragma unroll
    for (int k = 0; k < (KREG > 0 ? KREG : 1); ++k) qacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t v0 = wave * RPW; v0 < V; v0 += nwaves * RPW) {
        const int64_t v = v0 + sub;
        const bool ok = v < V;
        const int beg = ok ? row_ptr[v] : 0, end = ok ? row_ptr[v + 1] : 0;
        int n = end - beg;
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) n = max(n, __shfl_xor(n, o));   // longest row of the wave pass
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // two edges of the row per step, all four loads requested before the first is used (the adds stay in edge order)
        for (int j = 0; j < n; j += 2) {
            bool has[2];
            int64_t e[2];
            f32x4 g[2], dg[2];
            int t[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                has[u] = beg + j + u < end;
                e[u] = has[u] ? beg + j + u : (beg < end ? beg : 0);
                g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                dg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (live && has[u]) {
                    g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gate + e[u] * F + c));
                    dg[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dgate + e[u] * F + c));
                }
                t[u] = (live && has[u]) ? edge_type[e[u]] : -1;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float dot = g[u].x * dg[u].x + g[u].y * dg[u].y + g[u].z * dg[u].z + g[u].w * dg[u].w;
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) dot += __shfl_xor(dot, o);
                const f32x4 dz = g[u] * (dg[u] - dot);
                acc += dz;
                if (KREG > 0) {
#pragma unroll
                    for (int k = 0; k < KREG; ++k)
                        if (t[u] == k) qacc[k] += dz;
                } else if (t[u] >= 0) {
                    if (t[u] < k_lds) {                    // LDS float adds (ds_add_f32), flushed once per block
                        float* p = part + t[u] * F + c;
                        atomicAdd(p, dz.x); atomicAdd(p + 1, dz.y); atomicAdd(p + 2, dz.z); atomicAdd(p + 3, dz.w);
                    } else {
                        float* p = dq + (int64_t)t[u] * F + c;
                        atomicAdd(p, dz.x); atomicAdd(p + 1, dz.y); atomicAdd(p + 2, dz.z); atomicAdd(p + 3, dz.w);
                    }
                }
            }
        }
        if (ok && live) *reinterpret_cast<f32x4*>(dz_atom + v * F + c) = acc;
    }
    if (KREG > 0 && live) {
#pragma unroll
        for (int k = 0; k < KREG; ++k)
            if (k < K) {
                float* p = part + k * F + c;
                atomicAdd(p, qacc[k].x); atomicAdd(p + 1, qacc[k].y); atomicAdd(p + 2, qacc[k].z); atomicAdd(p + 3, qacc[k].w);
            }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < k_lds * F; idx += 256) {
        const float x = part[idx];
        if (x != 0.f) atomicAdd(dq + idx, x);
    }
}

}  // namespace mpnn

using namespace mpnn;

static int lanes_per_row(int F) {
    int l = 1;
    while (4 * l < F) l <<= 1;
    return l;
}

extern "C" int mpnn_att_gate_f32(const float* z_atom, const float* q, const int32_t* dst, const int32_t* edge_type,
                                 float* gate, int64_t V, int64_t E, int K, int F, void* stream) {
    MPNN_REQUIRE(V >= 0 && E >= 0 && K >= 0, "mpnn_att_gate_f32: negative size");
    MPNN_REQUIRE(F > 0 && F <= 256 && (F & 3) == 0, "mpnn_att_gate_f32: F=%d unsupported (multiple of 4, <= 256)", F);
    if (E == 0) return MPNN_OK;
    MPNN_REQUIRE(z_atom && q && dst && edge_type && gate, "mpnn_att_gate_f32: NULL buffer");
    const int lpr = lanes_per_row(F);
    int64_t blocks = ceil_div(E, (int64_t)4 * (64 / lpr));
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MPNN_GATE_CASE(L)                                                                                         \
    case L:                                                                                                       \
        hipLaunchKernelGGL((att_gate_fwd_kernel<L>), grid, block, 0, s, z_atom, q, dst, edge_type, gate, E, F);   \
        break;
    switch (lpr) {
        MPNN_GATE_CASE(1) MPNN_GATE_CASE(2) MPNN_GATE_CASE(4) MPNN_GATE_CASE(8) MPNN_GATE_CASE(16) MPNN_GATE_CASE(32)
        MPNN_GATE_CASE(64)
    }
#undef MPNN_GATE_CASE
    return launch_status("mpnn_att_gate_f32");
}

extern "C" int mpnn_att_gate_bwd_f32(const float* gate, const float* dgate, const int32_t* row_ptr,
                                     const int32_t* edge_type, float* dz_atom, float* dq, int64_t V, int64_t E, int K,
                                     int F, void* stream) {
    MPNN_REQUIRE(V >= 0 && E >= 0 && K >= 0, "mpnn_att_gate_bwd_f32: negative size");
    MPNN_REQUIRE(F > 0 && F <= 256 && (F & 3) == 0, "mpnn_att_gate_bwd_f32: F=%d unsupported (multiple of 4, <= 256)", F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(gate && dgate && row_ptr && edge_type && dz_atom && dq, "mpnn_att_gate_bwd_f32: NULL buffer");
    const int lpr = lanes_per_row(F);
    const int k_lds = K <= 32 ? K : 0;                      // few types: per-block LDS partials of dq; many: global atomics
    int64_t blocks = ceil_div(V, (int64_t)4 * (64 / lpr));
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = (size_t)k_lds * F * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define MPNN_GATE_CASE(L)                                                                                          \
    case L:                                                                                                        \
        if (K <= 8)                                                                                                \
            hipLaunchKernelGGL((att_gate_bwd_kernel<L, 8>), grid, block, lds, s, gate, dgate, row_ptr, edge_type, dz_atom, \
                               dq, V, K, F, k_lds);                                                                \
        else                                                                                                       \
            hipLaunchKernelGGL((att_gate_bwd_kernel<L, 0>), grid, block, lds, s, gate, dgate, row_ptr, edge_type, dz_atom, \
                               dq, V, K, F, k_lds);                                                                \
        break;
    switch (lpr) {
        MPNN_GATE_CASE(1) MPNN_GATE_CASE(2) MPNN_GATE_CASE(4) MPNN_GATE_CASE(8) MPNN_GATE_CASE(16) MPNN_GATE_CASE(32)
        MPNN_GATE_CASE(64)
    }
#undef MPNN_GATE_CASE
    return launch_status("mpnn_att_gate_bwd_f32");
}
