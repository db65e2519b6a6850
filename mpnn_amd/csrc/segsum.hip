// Neighbour aggregation: segmented sum over destination-sorted edge rows.
//
//   out[i,:] = sum_{e in [row_ptr[i], row_ptr[i+1])} w[e] * msg[e,:]
//
// HBM-bound: every message row is read once, every output row written once.
// Algorithmic bytes per launch = 4*F*(E+V) + 4*(V+1) (+4*E with weights).
//
// Mapping: a row of F floats is covered by LPR lanes x VEC floats (16 lanes x float4 at
// F=64), so one wave64 reduces 64/LPR destination atoms side by side (the default
// segsum_pair_kernel: two adjacent atoms per lane group per pass, 5.42 TB/s on c2 vs 5.1;
// with nontemporal loads/stores -- every row is touched exactly once -- 5.93 TB/s).  Edges of one atom
// are contiguous (CSR by destination) and atoms of one wave are adjacent, so the wave's
// loads walk one contiguous span of `msg`.  The edge loop runs in predicated batches of 4 rows
// (all loads of a batch in flight together) with the adds kept in edge order: deterministic.
#include <stdlib.h>

#include "common.h"

namespace mpnn {

template <int VEC>
struct Row;
template <>
struct Row<4> {
    f32x4 v;
    __device__ __forceinline__ void zero() { v = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = v; }
    // streamed once: keep the rows out of the way of reused data in L2 / Infinity Cache
    __device__ __forceinline__ void load_nt(const float* p) { v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
    __device__ __forceinline__ void store_nt(float* p) const { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
    __device__ __forceinline__ void fma(const Row& o, float s) { v += o.v * s; }
    __device__ __forceinline__ void add(const Row& o) { v += o.v; }
    __device__ __forceinline__ void scale(float s) { v *= s; }
};
template <>
struct Row<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
    __device__ __forceinline__ void load_nt(const float* p) { v = __builtin_nontemporal_load(p); }
    __device__ __forceinline__ void store_nt(float* p) const { __builtin_nontemporal_store(v, p); }
    __device__ __forceinline__ void fma(const Row& o, float s) { v += o.v * s; }
    __device__ __forceinline__ void add(const Row& o) { v += o.v; }
    __device__ __forceinline__ void scale(float s) { v *= s; }
};

// GATHER: message row of edge e is x[idx[e]] (idx may be NULL => e)
template <int VEC, int LPR, bool GATHER>
__global__ void __launch_bounds__(256) segsum_kernel(const float* __restrict__ msg, const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                     float* __restrict__ out, int64_t V, int F) {
    constexpr int GPB = 256 / LPR;                  // destination atoms per block pass
    const int lig = threadIdx.x % LPR;              // lane inside the row group
    const int grp = threadIdx.x / LPR;
    for (int64_t i = (int64_t)blockIdx.x * GPB + grp; i < V; i += (int64_t)gridDim.x * GPB) {
        const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
        for (int c = lig * VEC; c < F; c += LPR * VEC) {
            Row<VEC> acc;
            acc.zero();
            // edges in batches of 4 with the loads predicated on the row's length: the (typical) atom of
            // degree <= 4 costs ONE memory round trip, all its rows in flight together; rows past the
            // end contribute an exact +0 so the sum keeps edge order and stays deterministic
            for (int e = e0; e < e1; e += 4) {
                const int n = e1 - e;
                Row<VEC> r0, r1, r2, r3;
                r1.zero(); r2.zero(); r3.zero();
                float w0 = 1.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
                int64_t s0 = e, s1 = e + 1, s2 = e + 2, s3 = e + 3;
                if (GATHER && idx) {
                    s0 = idx[e];
                    if (n > 1) s1 = idx[e + 1];
                    if (n > 2) s2 = idx[e + 2];
                    if (n > 3) s3 = idx[e + 3];
                }
                r0.load(msg + s0 * F + c);
                if (n > 1) r1.load(msg + s1 * F + c);
                if (n > 2) r2.load(msg + s2 * F + c);
                if (n > 3) r3.load(msg + s3 * F + c);
                if (w) {
                    w0 = w[e];
                    if (n > 1) w1 = w[e + 1];
                    if (n > 2) w2 = w[e + 2];
                    if (n > 3) w3 = w[e + 3];
                    acc.fma(r0, w0); acc.fma(r1, w1); acc.fma(r2, w2); acc.fma(r3, w3);
                } else {
                    acc.add(r0); acc.add(r1); acc.add(r2); acc.add(r3);
                }
            }
            acc.store(out + i * F + c);
        }
    }
}

// Variant 2: each lane group reduces TWO adjacent atoms per pass (their rows are one contiguous
// span [e0,e2)), up to 8 row loads in flight per group, and the next pass's three row_ptr values are
// requested before the current rows are summed (one dependent round trip per pass instead of two).
template <int VEC, int LPR, bool NT, bool HUB = (LPR >= 32)>
__global__ void __launch_bounds__(256) segsum_pair_kernel(const float* __restrict__ msg,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const float* __restrict__ w, float* __restrict__ out,
                                                          int64_t V, int F) {
    constexpr int GPB = 256 / LPR;
    const int lig = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    const int c = lig * VEC;
    const int64_t step = 2 * (int64_t)gridDim.x * GPB;
    int64_t i = 2 * ((int64_t)blockIdx.x * GPB + grp);
    if (i >= V) return;
    int e0 = row_ptr[i], e1 = row_ptr[i + 1], e2 = (i + 1 < V) ? row_ptr[i + 2] : e1;
    for (; i < V; i += step) {
        const int64_t in = i + step;
        int n0 = 0, n1 = 0, n2 = 0;
        if (in < V) {                                   // prefetch the next pass's row pointers
            n0 = row_ptr[in];
            n1 = row_ptr[in + 1];
            n2 = (in + 1 < V) ? row_ptr[in + 2] : n1;
        }
        if (c < F) {
            Row<VEC> accA, accB;
            accA.zero();
            accB.zero();
            int a = e0, b = e1;
            while (a < e1 || b < e2) {
                const int na = e1 - a, nb = e2 - b;
                // Skewed degrees (configs[4]): a hub row outlives its partner by many passes and would then run with four
                // loads in flight instead of eight.  At the wide widths (a lane group = half a wave or a whole one, so
                // the branch below is all but uniform) the finished atom's four load slots go to the survivor: its
                // rows a+4 .. a+7 ride in them and are added AFTER rows a .. a+3 -- same edge order, same result.
                const bool soloA = HUB && nb <= 0, soloB = HUB && na <= 0;
                const int sb = soloA ? a + 4 : b, cb = soloA ? na - 4 : nb;       // what the B slots load, and how many
                const int sa = soloB ? b + 4 : a, ca = soloB ? nb - 4 : na;       // (solo B: the A slots take b+4 .. b+7)
                Row<VEC> ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
                ra0.zero(); ra1.zero(); ra2.zero(); ra3.zero();
                rb0.zero(); rb1.zero(); rb2.zero(); rb3.zero();
                float wa0 = 0.f, wa1 = 0.f, wa2 = 0.f, wa3 = 0.f, wb0 = 0.f, wb1 = 0.f, wb2 = 0.f, wb3 = 0.f;
                if (ca > 0) { if (NT) ra0.load_nt(msg + (int64_t)(sa + 0) * F + c); else ra0.load(msg + (int64_t)(sa + 0) * F + c); }
                if (ca > 1) { if (NT) ra1.load_nt(msg + (int64_t)(sa + 1) * F + c); else ra1.load(msg + (int64_t)(sa + 1) * F + c); }
                if (ca > 2) { if (NT) ra2.load_nt(msg + (int64_t)(sa + 2) * F + c); else ra2.load(msg + (int64_t)(sa + 2) * F + c); }
                if (ca > 3) { if (NT) ra3.load_nt(msg + (int64_t)(sa + 3) * F + c); else ra3.load(msg + (int64_t)(sa + 3) * F + c); }
                if (cb > 0) { if (NT) rb0.load_nt(msg + (int64_t)(sb + 0) * F + c); else rb0.load(msg + (int64_t)(sb + 0) * F + c); }
                if (cb > 1) { if (NT) rb1.load_nt(msg + (int64_t)(sb + 1) * F + c); else rb1.load(msg + (int64_t)(sb + 1) * F + c); }
                if (cb > 2) { if (NT) rb2.load_nt(msg + (int64_t)(sb + 2) * F + c); else rb2.load(msg + (int64_t)(sb + 2) * F + c); }
                if (cb > 3) { if (NT) rb3.load_nt(msg + (int64_t)(sb + 3) * F + c); else rb3.load(msg + (int64_t)(sb + 3) * F + c); }
                if (w) {
                    if (ca > 0) wa0 = w[sa];
                    if (ca > 1) wa1 = w[sa + 1];
                    if (ca > 2) wa2 = w[sa + 2];
                    if (ca > 3) wa3 = w[sa + 3];
                    if (cb > 0) wb0 = w[sb];
                    if (cb > 1) wb1 = w[sb + 1];
                    if (cb > 2) wb2 = w[sb + 2];
                    if (cb > 3) wb3 = w[sb + 3];
                    ra0.scale(wa0); ra1.scale(wa1); ra2.scale(wa2); ra3.scale(wa3);
                    rb0.scale(wb0); rb1.scale(wb1); rb2.scale(wb2); rb3.scale(wb3);
                }
                if (soloA) {            // rows a .. a+7 of atom A, in order
                    accA.add(ra0); accA.add(ra1); accA.add(ra2); accA.add(ra3);
                    accA.add(rb0); accA.add(rb1); accA.add(rb2); accA.add(rb3);
                    a += 8;
                } else if (soloB) {     // rows b .. b+7 of atom B: the B slots hold b .. b+3, the A slots b+4 .. b+7
                    accB.add(rb0); accB.add(rb1); accB.add(rb2); accB.add(rb3);
                    accB.add(ra0); accB.add(ra1); accB.add(ra2); accB.add(ra3);
                    b += 8;
                } else {
                    accA.add(ra0); accA.add(ra1); accA.add(ra2); accA.add(ra3);
                    accB.add(rb0); accB.add(rb1); accB.add(rb2); accB.add(rb3);
                    a += 4;
                    b += 4;
                }
            }
            if (NT) {
                accA.store_nt(out + i * F + c);
                if (i + 1 < V) accB.store_nt(out + (i + 1) * F + c);
            } else {
                accA.store(out + i * F + c);
                if (i + 1 < V) accB.store(out + (i + 1) * F + c);
            }
        }
        e0 = n0;
        e1 = n1;
        e2 = n2;
    }
}

template <int VEC, int LPR>
__global__ void __launch_bounds__(256) segsum_bwd_kernel(const float* __restrict__ dout,
                                                         const int32_t* __restrict__ row_ptr,
                                                         const float* __restrict__ w, float* __restrict__ dmsg,
                                                         int64_t V, int F) {
    constexpr int GPB = 256 / LPR;
    const int lig = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    for (int64_t i = (int64_t)blockIdx.x * GPB + grp; i < V; i += (int64_t)gridDim.x * GPB) {
        const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
        for (int c = lig * VEC; c < F; c += LPR * VEC) {
            Row<VEC> g;
            g.load(dout + i * F + c);
            for (int e = e0; e < e1; ++e) {
                Row<VEC> r = g;
                if (w) r.scale(w[e]);
                r.store(dmsg + (int64_t)e * F + c);
            }
        }
    }
}

static int pick_lpr(int F, int vec) {
    int need = (F + vec - 1) / vec;   // lanes to cover a row in one pass
    int lpr = 4;
    while (lpr < need && lpr < 64) lpr <<= 1;
    return lpr;
}

static int grid_for(int64_t V, int lpr) {
    int64_t g = ceil_div(V, 256 / lpr);
    const int64_t cap = 256 * 16;   // 16 blocks of 4 waves per CU, grid-stride beyond
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <bool GATHER>
static int launch_segsum(const float* msg, const int32_t* row_ptr, const int32_t* idx, const float* w, float* out,
                         int64_t V, int F, hipStream_t s) {
    const bool v4 = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(msg) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int lpr = pick_lpr(F, v4 ? 4 : 1);
    const dim3 grid(grid_for(V, lpr)), block(256);
    if (!GATHER && v4 && lpr * 4 >= F) {                   // two atoms per lane group, nontemporal accesses
        const dim3 g2(grid_for((V + 1) / 2, lpr));
#define MPNN_PAIR(LPR) hipLaunchKernelGGL((segsum_pair_kernel<4, LPR, true>), g2, block, 0, s, msg, row_ptr, w, out, V, F);
        switch (lpr) {
            case 16: MPNN_PAIR(16) break;
            case 32: MPNN_PAIR(32) break;
            case 64: MPNN_PAIR(64) break;
            default: MPNN_PAIR(8) break;
        }
#undef MPNN_PAIR
        return launch_status("mpnn_segsum(pair)");
    }
#define MPNN_SEGSUM_CASE(VEC, LPR)                                                                              \
    hipLaunchKernelGGL((segsum_kernel<VEC, LPR, GATHER>), grid, block, 0, s, msg, row_ptr, idx, w, out, V, F); \
    break;
    if (v4) {
        switch (lpr) {
            case 4: MPNN_SEGSUM_CASE(4, 4)
            case 8: MPNN_SEGSUM_CASE(4, 8)
            case 16: MPNN_SEGSUM_CASE(4, 16)
            case 32: MPNN_SEGSUM_CASE(4, 32)
            default: MPNN_SEGSUM_CASE(4, 64)
        }
    } else {
        switch (lpr) {
            case 4: MPNN_SEGSUM_CASE(1, 4)
            case 8: MPNN_SEGSUM_CASE(1, 8)
            case 16: MPNN_SEGSUM_CASE(1, 16)
            case 32: MPNN_SEGSUM_CASE(1, 32)
            default: MPNN_SEGSUM_CASE(1, 64)
        }
    }
#undef MPNN_SEGSUM_CASE
    return launch_status("mpnn_segsum");
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_segsum_f32(const float* msg, const int32_t* row_ptr, const float* w, float* out, int64_t V, int F,
                               void* stream) {
    MPNN_REQUIRE(row_ptr && out && V >= 0, "mpnn_segsum_f32: bad arguments");
    MPNN_REQUIRE(F > 0 && F <= MPNN_MAX_FEATURES, "mpnn_segsum_f32: F=%d out of range", F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(msg, "mpnn_segsum_f32: msg is NULL");
    return launch_segsum<false>(msg, row_ptr, nullptr, w, out, V, F, (hipStream_t)stream);
}

extern "C" int mpnn_segsum_gather_f32(const float* x, const int32_t* row_ptr, const int32_t* idx, const float* w,
                                      float* out, int64_t V, int F, void* stream) {
    MPNN_REQUIRE(row_ptr && out && V >= 0, "mpnn_segsum_gather_f32: bad arguments");
    MPNN_REQUIRE(F > 0 && F <= MPNN_MAX_FEATURES, "mpnn_segsum_gather_f32: F=%d out of range", F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(x, "mpnn_segsum_gather_f32: x is NULL");
    return launch_segsum<true>(x, row_ptr, idx, w, out, V, F, (hipStream_t)stream);
}

extern "C" int mpnn_segsum_bwd_f32(const float* dout, const int32_t* row_ptr, const float* w, float* dmsg, int64_t V,
                                   int F, void* stream) {
    MPNN_REQUIRE(row_ptr && V >= 0, "mpnn_segsum_bwd_f32: bad arguments");
    MPNN_REQUIRE(F > 0 && F <= MPNN_MAX_FEATURES, "mpnn_segsum_bwd_f32: F=%d out of range", F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(dout && dmsg, "mpnn_segsum_bwd_f32: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dmsg)) % 16 == 0);
    const int lpr = pick_lpr(F, v4 ? 4 : 1);
    const dim3 grid(grid_for(V, lpr)), block(256);
#define MPNN_BWD_CASE(VEC, LPR) \
    hipLaunchKernelGGL((segsum_bwd_kernel<VEC, LPR>), grid, block, 0, s, dout, row_ptr, w, dmsg, V, F); \
    break;
    if (v4) {
        switch (lpr) {
            case 4: MPNN_BWD_CASE(4, 4)
            case 8: MPNN_BWD_CASE(4, 8)
            case 16: MPNN_BWD_CASE(4, 16)
            case 32: MPNN_BWD_CASE(4, 32)
            default: MPNN_BWD_CASE(4, 64)
        }
    } else {
        switch (lpr) {
            case 4: MPNN_BWD_CASE(1, 4)
            case 8: MPNN_BWD_CASE(1, 8)
            case 16: MPNN_BWD_CASE(1, 16)
            case 32: MPNN_BWD_CASE(1, 32)
            default: MPNN_BWD_CASE(1, 64)
        }
    }
#undef MPNN_BWD_CASE
    return launch_status("mpnn_segsum_bwd");
}
