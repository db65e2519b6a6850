// Typed edge message at nf = mf = 128 (and 64) on the bf16 matrix pipe with 3-way operand splitting (split_math.h).
//
// Same shape as edge_message_resident_kernel (edge_message.hip): persistent waves, the type's matrix resident in
// LDS, a lane gathers its own edge's contiguous half-row, no barrier inside a type.  Differences at this width:
//   * the matrix is kept as three bf16 images (96 KB) instead of one fp32 image: six K=16 MFMAs of 32 cycles per
//     512 fp32-MFMA cycles, and the fp32 image would not leave room for anything else;
//   * the half-row is 64 floats, so it moves through a ring of 16-float chunks fetched ahead (scheduling
//     barriers pin the order, as in gru_split.hip).
// BWD = false: msg[e] = A_k . (gate[e] * h[src(e)])       image[n][k] = A_k[n][k]
// BWD = true : dx[e]  = A_k^T . dmsg[e]                   image[n][k] = A_k[k][n], rows indexed by edge id
#include "split_math.h"

namespace mpnn {

// DGATE (with BWD): the gate gradient of message + adjacency-weighted sum in one pass,
//   out[e] = (A_k^T (w[e] * dagg[dst(e)])) * hmul[hsrc(e)]      -- rows of `h` (= dagg) gathered through `rowidx` (= dst),
// so neither d(msg) nor dx is ever written.
template <int F, int NW, bool BWD, bool GATED, bool DGATE = false>
__global__ void __launch_bounds__(64 * NW, (F == 64 && !GATED) ? 4 : 1) edge_message_split_kernel(
    const float* __restrict__ h, const float* __restrict__ A, const int32_t* __restrict__ src,
    const int32_t* __restrict__ order, const int32_t* __restrict__ type_ptr, const float* __restrict__ gate,
    float* __restrict__ msg, int K, const int32_t* __restrict__ rowidx = nullptr,
    const float* __restrict__ wrow = nullptr, const float* __restrict__ hmul = nullptr,
    const int32_t* __restrict__ hsrc = nullptr) {
    constexpr int ROWB = 2 * F, IMG = F * ROWB;
    constexpr int NCH = F / 32;                // 16-float chunks in a lane's half-row
    constexpr int NB = F / 32;                 // 32-column output blocks
    constexpr int SW = F == 128 ? 15 : 7;      // swizzle mask: rows are F/2 dwords apart
    constexpr int RD = (GATED || NCH < 4) ? 2 : 4;   // ring depth; chunks are fetched RD-1 ahead
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 pieces][F n][F k] bf16

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    // 16-byte chunks of an image row are XOR-swizzled by the row: H=128 rows start on one bank (16 positions),
    // H=64 rows alternate between the two halves of the banks (8 positions, keyed on row/2)
    auto swz = [](int n) { return F == 128 ? (n & 15) : ((n >> 1) & 7); };
    (void)SW;
    const int gw = blockIdx.x * NW + wv, nw = gridDim.x * NW;

    for (int k = 0; k < K; ++k) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        if (te == tb) continue;                         // uniform over the grid
        __syncthreads();                                // everyone is done with the previous matrix
        const float* Ak = A + (int64_t)k * F * F;
        for (int idx = tid; idx < F * (F / 4); idx += 64 * NW) {
            const int row = idx / (F / 4), c4 = 4 * (idx % (F / 4));
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Ak + (int64_t)row * F + c4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = BWD ? c4 + u : row, kk = BWD ? row : c4 + u;
                __bf16 ph, pm, pl;
                split3(w4[u], ph, pm, pl);
                const int off = n * ROWB + (((kk >> 3) ^ swz(n)) << 4) + ((kk & 7) << 1);
                *reinterpret_cast<__bf16*>(smem + off) = ph;
                *reinterpret_cast<__bf16*>(smem + IMG + off) = pm;
                *reinterpret_cast<__bf16*>(smem + 2 * IMG + off) = pl;
            }
        }
        __syncthreads();

        const int tiles = (te - tb + 31) / 32;
        int t = gw;
        if (t >= tiles) continue;
        auto edge_of = [&](int tile) {
            const int pos = tb + 32 * tile + r;
            return order[pos < te ? pos : tb + 32 * tile];
        };
        auto src_of = [&](int e) { return BWD ? (DGATE ? rowidx[e] : e) : src[e]; };
        // chunk j (0..NCH-1): 16 floats at 16*j of the lane half's F/2 floats of the row
        auto load_chunk = [&](int s_row, int j, f32x4 (&f)[4]) {
            const float* p = h + (int64_t)s_row * F + hi * (F / 2) + 16 * j;
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
        };
        auto load_gate = [&](int e_row, int j, f32x4 (&f)[4]) {
            const float* g = gate + (int64_t)e_row * F + hi * (F / 2) + 16 * j;
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(g + 4 * q);
        };
        auto bfrag = [&](int piece, int n, int chunk) {
            return *reinterpret_cast<const bf16x8*>(smem + piece * IMG + n * ROWB + ((chunk ^ swz(n)) << 4));
        };

        int e_cur = edge_of(t);
        int s_cur = src_of(e_cur);
        int e_nxt = e_cur, s_nxt = s_cur;
        if (t + nw < tiles) { e_nxt = edge_of(t + nw); s_nxt = src_of(e_nxt); }
        // DGATE epilogue operands of lane r's own edge (broadcast by shuffle later): fetched a tile ahead with the
        // other indices, so the epilogue only issues the independent hmul row loads
        int hs_cur = 0, hs_nxt = 0;
        float w_cur = 1.0f, w_nxt = 1.0f;
        if (DGATE) {
            hs_cur = hsrc[e_cur];
            hs_nxt = hsrc[e_nxt];
            if (wrow) { w_cur = wrow[e_cur]; w_nxt = wrow[e_nxt]; }
        }
        // Gated rows: the gate chunk travels in its own one-deep buffer and is multiplied in AFTER the MFMAs of the
        // running chunk are issued (a multiply at fetch time would wait for both loads before any MFMA goes out).
        f32x4 ring[RD][4], gbuf[4];
#pragma unroll
        for (int j = 0; j < RD - 1; ++j) {
            load_chunk(s_cur, j, ring[j]);
            if (GATED) {
                load_gate(e_cur, j, gbuf);
#pragma unroll
                for (int q = 0; q < 4; ++q) ring[j][q] *= gbuf[q];
            }
        }

        for (; t < tiles; t += nw) {
            const bool has2 = t + 2 * nw < tiles;
            int e_nn = e_nxt;
            if (has2) e_nn = edge_of(t + 2 * nw);
            f32x16 acc[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cn = c + RD - 1;                                 // chunk to fetch now
                if (cn < NCH) {
                    load_chunk(s_cur, cn, ring[cn % RD]);
                    if (GATED) load_gate(e_cur, cn, gbuf);
                } else {                                                   // next tile (or a harmless re-read)
                    load_chunk(s_nxt, cn - NCH, ring[cn % RD]);
                    if (GATED) load_gate(e_nxt, cn - NCH, gbuf);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 ah, am, al;
                    split8(ring[c % RD][2 * s2], ring[c % RD][2 * s2 + 1], ah, am, al);
                    const int chunk = (F / 16) * hi + 2 * c + s2;
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        mma6(acc[n], ah, am, al, bfrag(0, 32 * n + r, chunk), bfrag(1, 32 * n + r, chunk),
                             bfrag(2, 32 * n + r, chunk));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (GATED) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ring[cn % RD][q] *= gbuf[q];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            int s_nn = s_nxt;
            if (has2) s_nn = src_of(e_nn);

            const int rows = min(32, te - tb - 32 * t);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = acc_row(i, lane);
                const int e_row = __shfl(e_cur, row);
                const float wv = DGATE ? __shfl(w_cur, row) : 1.0f;      // shuffles stay outside the divergent branch
                const int hs_row = DGATE ? __shfl(hs_cur, row) : 0;
                if (row < rows) {
                    if (DGATE) {
                        const float* hm = hmul + (int64_t)hs_row * F + r;
#pragma unroll
                        for (int n = 0; n < NB; ++n)
                            __builtin_nontemporal_store(acc[n][i] * wv * hm[32 * n], msg + (int64_t)e_row * F + 32 * n + r);
                    } else {
#pragma unroll
                        for (int n = 0; n < NB; ++n)
                            __builtin_nontemporal_store(acc[n][i], msg + (int64_t)e_row * F + 32 * n + r);
                    }
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            e_cur = e_nxt;
            s_cur = s_nxt;
            e_nxt = e_nn;
            s_nxt = s_nn;
            if (DGATE) {
                hs_cur = hs_nxt;
                w_cur = w_nxt;
                hs_nxt = hsrc[e_nxt];
                if (wrow) w_nxt = wrow[e_nxt];
            }
        }
    }
}

template <int F, bool BWD>
static int launch_split(const float* h, const float* A, const int32_t* src, const int32_t* order,
                        const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s) {
    // F = 128: one 8-wave block per CU (96 KB of LDS).  F = 64: 24 KB images, 4-wave blocks, 4 per CU ungated (<= 128 VGPRs).
    constexpr int NW = F == 128 ? 8 : 4;
    const size_t lds = (size_t)3 * F * 2 * F;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)edge_message_split_kernel<F, NW, BWD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)edge_message_split_kernel<F, NW, BWD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t blocks = F == 128 ? 256 : (gate ? 512 : 1024);
    const int64_t need = ceil_div(ceil_div(E, 32) + K, NW);
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (gate)
        hipLaunchKernelGGL((edge_message_split_kernel<F, NW, BWD, true>), dim3((unsigned)blocks), dim3(64 * NW), lds, s, h,
                           A, src, order, type_ptr, gate, msg, K);
    else
        hipLaunchKernelGGL((edge_message_split_kernel<F, NW, BWD, false>), dim3((unsigned)blocks), dim3(64 * NW), lds, s,
                           h, A, src, order, type_ptr, gate, msg, K);
    return launch_status(BWD ? "mpnn_edge_message_bwd_f32(dx, bf16x6)" : "mpnn_edge_message_f32(bf16x6)");
}

int launch_message_split128(const float* h, const float* A, const int32_t* src, const int32_t* order,
                            const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s) {
    return launch_split<128, false>(h, A, src, order, type_ptr, gate, msg, E, K, s);
}
int launch_message_dx_split128(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                               float* dx, int64_t E, int K, hipStream_t s) {
    return launch_split<128, true>(dmsg, A, nullptr, order, type_ptr, nullptr, dx, E, K, s);
}
template <int F>
static int launch_dgate(const float* dagg, const float* A, const int32_t* dst, const float* w, const int32_t* order,
                        const int32_t* type_ptr, const float* hmul, const int32_t* hsrc, float* dgate, int64_t E, int K,
                        hipStream_t s) {
    constexpr int NW = F == 128 ? 8 : 4;
    const size_t lds = (size_t)3 * F * 2 * F;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)edge_message_split_kernel<F, NW, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t blocks = F == 128 ? 256 : 1024;
    const int64_t need = ceil_div(ceil_div(E, 32) + K, NW);
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((edge_message_split_kernel<F, NW, true, false, true>), dim3((unsigned)blocks), dim3(64 * NW), lds, s,
                       dagg, A, (const int32_t*)nullptr, order, type_ptr, (const float*)nullptr, dgate, K, dst, w, hmul,
                       hsrc);
    return launch_status("mpnn_edge_message_agg_bwd_dgate_f32");
}

int launch_message_dgate(const float* dagg, const float* A, const int32_t* dst, const float* w, const int32_t* order,
                         const int32_t* type_ptr, const float* hmul, const int32_t* hsrc, float* dgate, int64_t E, int K,
                         int F, hipStream_t s) {
    if (F == 128) return launch_dgate<128>(dagg, A, dst, w, order, type_ptr, hmul, hsrc, dgate, E, K, s);
    if (F == 64) return launch_dgate<64>(dagg, A, dst, w, order, type_ptr, hmul, hsrc, dgate, E, K, s);
    return 1;
}

int launch_message_split64(const float* h, const float* A, const int32_t* src, const int32_t* order,
                           const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s) {
    return launch_split<64, false>(h, A, src, order, type_ptr, gate, msg, E, K, s);
}
int launch_message_dx_split64(const float* dmsg, const float* A, const int32_t* order, const int32_t* type_ptr,
                              float* dx, int64_t E, int K, hipStream_t s) {
    return launch_split<64, true>(dmsg, A, nullptr, order, type_ptr, nullptr, dx, E, K, s);
}

}  // namespace mpnn

// ---------------------------------------------------------------------------------------------------------------
// nf = mf = 256: the three images of one A_k are 393 KB, so the matrix is STREAMED like the GRU weights at this width
// (gru_split.hip, gru_update_stream_kernel): a block owns a 64-feature output slice, the contraction is cut into four
// 64-wide chunks, all threads split the next chunk of the 64 matrix rows into a double-buffered LDS image while the
// eight waves (one 32-edge tile each, 64 output features) multiply the current one.  Types are walked in order;
// within a type a round is 256 edges.
namespace mpnn {

template <bool GATED>
__global__ void __launch_bounds__(512) edge_message_stream256_kernel(
    const float* __restrict__ h, const float* __restrict__ A, const int32_t* __restrict__ src,
    const int32_t* __restrict__ order, const int32_t* __restrict__ type_ptr, const float* __restrict__ gate,
    float* __restrict__ msg, int K) {
    constexpr int F = 256, NS = F / 64, KC = 64, NCT = F / KC;
    constexpr int IMGC = 64 * 2 * KC;          // one piece of a chunk image: 64 rows x 64 k bf16 = 8 KB
    constexpr int BUF = 3 * IMGC;              // 24 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % NS;
    const int pblock = (jb / NS) * 8 + xcd, pblocks = gridDim.x / NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;

    // staging unit = (matrix row n of the slice, k-octet of the chunk): 64 x 8 = 512 units, one per thread
    const int sn = tid / 8, so = tid % 8;
    const int wofs = (64 * slice + sn) * F + 8 * so;
    const int ldst = sn * 2 * KC + ((so ^ ((sn >> 1) & 7)) << 4);
    auto bfrag = [&](int buf, int piece, int nb, int st) {   // K step st (0..3) of the chunk, octet 4*hi + st
        const int n = 32 * nb + r;
        const int o = 4 * hi + st;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BUF + piece * IMGC + n * 2 * KC + ((o ^ ((n >> 1) & 7)) << 4));
    };

    int cur = 0;
    for (int k = 0; k < K; ++k) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        if (te == tb) continue;
        const int rounds_total = (te - tb + 255) / 256;
        if (pblock >= rounds_total) continue;                // block-uniform
        const int nrounds = (rounds_total - pblock + pblocks - 1) / pblocks;
        const float* Ak = A + (int64_t)k * F * F;

        f32x4 raw[2];
        auto stage_load = [&](int ct) {
            raw[0] = *reinterpret_cast<const f32x4*>(Ak + wofs + KC * ct);
            raw[1] = *reinterpret_cast<const f32x4*>(Ak + wofs + KC * ct + 4);
        };
        auto stage_write = [&](int buf) {
            bf16x8 ph, pm, pl;
            split8(raw[0], raw[1], ph, pm, pl);
            char* base = smem + buf * BUF + ldst;
            *reinterpret_cast<bf16x8*>(base) = ph;
            *reinterpret_cast<bf16x8*>(base + IMGC) = pm;
            *reinterpret_cast<bf16x8*>(base + 2 * IMGC) = pl;
        };
        // lane r owns edge slot tb + 32*tile + r; slots past the end reuse the type's first edge and are not stored
        auto edge_of = [&](int tile) {
            const int pos = tb + 32 * tile + r;
            return order[pos < te ? pos : tb];
        };
        // 32 floats (chunk ct, this lane half) of the edge's source row / of its gate row
        auto load_rows = [&](int s_row, int ct, f32x4 (&f)[8]) {
            const float* p = h + (int64_t)s_row * F + KC * ct + 32 * hi;
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
        };
        auto load_gate = [&](int e, int ct, f32x4 (&f)[8]) {
            const float* g = gate + (int64_t)e * F + KC * ct + 32 * hi;
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = *reinterpret_cast<const f32x4*>(g + 4 * q);
        };

        int tile = pblock * 8 + wv;
        int e_cur = edge_of(tile), s_cur = src[e_cur];
        f32x4 x0[8], x1[8], gb[8];
        f32x16 acc[2];
        __syncthreads();                                     // the previous type is done with both buffers
        stage_load(0);
        stage_write(cur);
        load_rows(s_cur, 0, x0);
        if (GATED) {
            load_gate(e_cur, 0, gb);
#pragma unroll
            for (int q = 0; q < 8; ++q) x0[q] *= gb[q];
        }

        // the gate chunk is multiplied in after the MFMAs of the running chunk are issued (see the 128 kernel)
        auto chunk = [&](int ct, int e_n, int s_n, f32x4 (&xc)[8], f32x4 (&xn)[8]) {
            __syncthreads();
            const int cn = (ct + 1) % NCT;
            stage_load(cn);
            if (cn == 0) {
                load_rows(s_n, 0, xn);
                if (GATED) load_gate(e_n, 0, gb);
            } else {
                load_rows(s_cur, cn, xn);
                if (GATED) load_gate(e_cur, cn, gb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 a_h, a_m, a_l;
                split8(xc[2 * st], xc[2 * st + 1], a_h, a_m, a_l);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    mma6(acc[nb], a_h, a_m, a_l, bfrag(cur, 0, nb, st), bfrag(cur, 1, nb, st), bfrag(cur, 2, nb, st));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (GATED) {
#pragma unroll
                for (int q = 0; q < 8; ++q) xn[q] *= gb[q];
            }
            stage_write(cur ^ 1);
            cur ^= 1;
        };

        for (int rd = 0; rd < nrounds; ++rd) {
            const int tile_next = rd + 1 < nrounds ? tile + pblocks * 8 : tile;
            const int e_n = edge_of(tile_next), s_n = src[e_n];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
            chunk(0, e_n, s_n, x0, x1);
            chunk(1, e_n, s_n, x1, x0);
            chunk(2, e_n, s_n, x0, x1);
            chunk(3, e_n, s_n, x1, x0);
            const int rows = min(32, te - tb - 32 * tile);       // may be <= 0 on the ragged last round
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = acc_row(i, lane);
                const int e_row = __shfl(e_cur, row);
                if (row < rows) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        __builtin_nontemporal_store(acc[nb][i], msg + (int64_t)e_row * F + 64 * slice + 32 * nb + r);
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            tile = tile_next;
            e_cur = e_n;
            s_cur = s_n;
        }
    }
}

int launch_message_stream256(const float* h, const float* A, const int32_t* src, const int32_t* order,
                             const int32_t* type_ptr, const float* gate, float* msg, int64_t E, int K, hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * 64 * 128;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)edge_message_stream256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        opt_in_((const void*)edge_message_stream256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const dim3 grid(256), block(512);                       // 64 row groups x 4 slices = one block per CU
    if (gate)
        hipLaunchKernelGGL((edge_message_stream256_kernel<true>), grid, block, lds, s, h, A, src, order, type_ptr, gate,
                           msg, K);
    else
        hipLaunchKernelGGL((edge_message_stream256_kernel<false>), grid, block, lds, s, h, A, src, order, type_ptr, gate,
                           msg, K);
    return launch_status("mpnn_edge_message_f32(bf16x6 256, streamed)");
}

}  // namespace mpnn
