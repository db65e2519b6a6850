// Weight gradient of the typed edge message followed by the neighbour sum, contracted over ATOMS instead of edges:
//   dA[k] = sum_i dagg[i] (x) X_ik,   X_ik = gate_ik * sum_{e in row i, type k} h[src e]
// replaces: the autograd of mpnn_functions/message/edge_network.py:40,52 composed with
//           message_aggregators/adjacent_message_agg.py:18 for the bond-type matrices (and, with the gate, of
//           att_edge_network.py:26-31).  The message is linear in h and the (attention) gate depends on the destination
//           atom and the bond type only, so the sum over an atom's type-k neighbours can be taken BEFORE the outer product:
//           the contraction runs over V atoms x (types present) instead of E edges, dagg rows are read once and in order,
//           and no per-edge row of dagg, h or gate is ever formed.  Widths 64 and 128, K <= 4 bond types.
//
// Structure (that of the wide GRU weight-gradient kernel, gru_bwd_rc.hip): a persistent 8-wave block walks 32-atom row
// blocks (natural atom order: no plan) and owns 64 columns of X (at width 128 two blocks share a row-block stream, one per
// column half; both read dagg).  Thread (row = tid >> 4, c16 = tid & 15) owns 4 X columns and F / 16 dagg columns of its
// atom: it requests the atom's CSR row three blocks ahead, its first three edges two blocks ahead and the rows they name
// (h[src], dagg, logits) one block ahead; sums the neighbour rows by type, applies the gate (evaluated from the logits and
// the forward's softmax statistics), and parks dagg and the X_k as two-piece fp16 images in LDS -- dagg behind its row's
// power-of-two scale sg_row, X behind sx_row = C / sg_row with C the running minimum over the block's row blocks of
// (smallest sg_row) x (best scale of the X values): every product carries C (gru_bwd_f16.hip has the argument), the
// accumulators are multiplied down when C drops.  Wave (k = wv >> 1, b = wv & 1) contracts dagg^T (all F rows, transposed
// LDS reads) with the 32-column half b of X_k: 2 x F / 32 x 3 MFMAs per row block; a type without edges in the row block
// is skipped.  Two barriers per row block.  Sum order: fixed per block; across blocks atomicAdd (as the per-edge kernel).
#include "common.h"

namespace mpnn {

struct DaAtt {
    const float* z_atom;     // (V, F) atom part of the gate logits (nullptr: no gate)
    const float* q;          // (K, F) bond part
    const float2* stats;     // (V, K): log2(e) * max_f, 1 / sum_f exp
};

namespace {
typedef _Float16 dh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dh16x4 __attribute__((ext_vector_type(4)));
typedef short ds16x4 __attribute__((ext_vector_type(4)));
constexpr float D_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int d_exp(float maxabs) {
    const int e = (__float_as_int(maxabs) >> 23) & 0xff;
    return e < 51 ? 51 : (e > 187 ? 187 : e);
}
__device__ __forceinline__ float d_pow2(int field) { return __int_as_float(field << 23); }
__device__ __forceinline__ float d_row16_max(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true);
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true);
    return fmaxf(v, __int_as_float(x));
}
__device__ __forceinline__ float d_wave_max(float v) {
    v = d_row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ void d_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ dh16x8 d_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) ds16x4 lds_s16x4;
    const ds16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const ds16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(dh16x8, v);
}
__device__ __forceinline__ void d_split4(const f32x4& x, float sc, dh16x4& ph, dh16x4& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = x[j] * sc;
        ph[j] = (_Float16)v;
        pl[j] = (_Float16)(v - (float)ph[j]);
    }
}
__device__ __forceinline__ float d_max4(float m, const f32x4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
}  // namespace

template <int F>
__host__ __device__ constexpr int da_rows_lds_bytes() { return 2 * (2 * F * 64 + 8 * 4096) + 256 + 1024; }

template <int F, bool ATT>
__global__ void __launch_bounds__(512) da_rows_kernel(const float* __restrict__ dagg, const float* __restrict__ h,
                                                      const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                      const int32_t* __restrict__ edge_type, DaAtt att, float* dA, int64_t V,
                                                      int64_t E, int K) {
    constexpr int NH = F / 64;                 // column halves of X = blocks per row-block stream
    constexpr int NA = F / 32;                 // 32-row tiles of dagg^T
    constexpr int DV = F / 64;                 // 4-vectors of dagg columns per thread
    constexpr int DIMG = F * 64;               // one dagg image piece: [F / 16 column groups][32 rows][16 columns] fp16
    constexpr int XIMG = 4096;                 // one X image piece: 64 columns
    constexpr int BUF = 2 * DIMG + 8 * XIMG;   // dagg hi | lo, then X[k][hi | lo]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 2 * BUF);      // [2 parities][max |X| x 8 | max 1 / sg x 8 | type masks x 8 | pad]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int half = NH == 2 ? (jb & 1) : 0;   // the two halves' blocks of a row-block stream sit on one XCD
    const int64_t blocks = (V + 31) / 32;
    const int64_t t0 = NH == 2 ? (int64_t)(jb >> 1) * 8 + xcd : (int64_t)blockIdx.x;
    const int64_t tstep = gridDim.x / NH;
    if (t0 >= blocks) return;

    const int srow = tid >> 4, c16 = tid & 15;
    const int xcol = 64 * half + 4 * c16;                       // this thread's X columns (of F)
    const int x_dst = (c16 >> 2) * 1024 + srow * 32 + (c16 & 3) * 8;
    const int d_dst = DV == 2 ? (c16 >> 1) * 1024 + srow * 32 + (c16 & 1) * 16 : x_dst;

    struct Csr { int rp0, rp1; bool ok; };                      // an atom's CSR row
    struct Idx { int src[3], ty[3], rp0, deg; };                 // ... and its first three edges
    struct Rows { f32x4 hv[3], d[DV], zz; float2 st[4]; int ty[3], rp0, deg; float live; };
    auto load_csr = [&](int64_t t) {
        const int64_t a = t * 32 + srow;
        Csr c;
        const bool ok = a < V;
        const int64_t ac = ok ? a : V - 1;
        c.rp0 = row_ptr[ac];
        c.rp1 = row_ptr[ac + 1];                                // (turned into a count in load_idx: nothing waits here)
        c.ok = ok;
        return c;
    };
    auto load_idx = [&](const Csr& c) {
        Idx x;
        x.rp0 = c.rp0;
        x.deg = c.ok ? c.rp1 - c.rp0 : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int64_t e = (int64_t)c.rp0 + q;
            e = e < E ? e : E - 1;
            x.src[q] = col_idx[e];
            x.ty[q] = edge_type[e];
        }
        return x;
    };
    auto load_rows = [&](int64_t t, const Idx& x) {
        Rows r;
        const int64_t a = t * 32 + srow;
        const bool ok = a < V;
        const int64_t ac = ok ? a : V - 1;
        r.live = ok ? 1.0f : 0.0f;
        r.rp0 = x.rp0;
        r.deg = x.deg;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            r.ty[q] = x.ty[q];
            r.hv[q] = *reinterpret_cast<const f32x4*>(h + (int64_t)x.src[q] * F + xcol);
        }
#pragma unroll
        for (int u = 0; u < DV; ++u) r.d[u] = *reinterpret_cast<const f32x4*>(dagg + ac * F + 4 * DV * c16 + 4 * u);
        if (ATT) {
            r.zz = *reinterpret_cast<const f32x4*>(att.z_atom + ac * F + xcol);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.st[k] = att.stats[ac * K + (k < K ? k : 0)];
        }
        return r;
    };
    float* const qs = red + 64;                                // ATT: log2(e) * q, the block's 64 columns of each type
    if (ATT) {
        if (tid < 256) qs[tid] = (tid >> 6) < K ? att.q[(tid >> 6) * F + 64 * half + (tid & 63)] * D_LOG2E : 0.f;
        __syncthreads();
    }

    // the atom's X values (this thread's 4 columns, per type) and which types its edges have
    auto form_x = [&](const Rows& r, f32x4 (&X)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        int has = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int ty = q < r.deg ? r.ty[q] : -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float w = ty == k ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) X[k][j] = fmaf(w, r.hv[q][j], X[k][j]);
            }
            has |= ty >= 0 ? 1 << ty : 0;
        }
        for (int q = 3; q < r.deg; ++q) {                       // a fourth, fifth ... edge: rare, fetched on the spot
            const int e = r.rp0 + q;
            const int ty = edge_type[e];
            const f32x4 v = *reinterpret_cast<const f32x4*>(h + (int64_t)col_idx[e] * F + xcol);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float w = ty == k ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) X[k][j] = fmaf(w, v[j], X[k][j]);
            }
            has |= 1 << ty;
        }
        if (ATT) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((has >> k) & 1) {
                    const f32x4 ql = *reinterpret_cast<const f32x4*>(qs + 64 * k + 4 * c16);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        X[k][j] *= __builtin_amdgcn_exp2f(fmaf(r.zz[j], D_LOG2E, ql[j] - r.st[k].x)) * r.st[k].y;
                }
        }
        return has;
    };
    // dagg pieces of a row block -> buffer T (row scale); returns the row's 1 / sg
    auto park_d = [&](const Rows& r, char* T) {
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < DV; ++u) mx = d_max4(mx, r.d[u]);
        mx = d_row16_max(mx * r.live);
        const int e = d_exp(mx);
        const float sg = d_pow2(268 - e) * r.live;
        if (DV == 2) {
            dh16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = r.d[0][j] * sg, b = r.d[1][j] * sg;
                ph[j] = (_Float16)a;
                pl[j] = (_Float16)(a - (float)ph[j]);
                ph[4 + j] = (_Float16)b;
                pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
            }
            *reinterpret_cast<dh16x8*>(T + d_dst) = ph;
            *reinterpret_cast<dh16x8*>(T + DIMG + d_dst) = pl;
        } else {
            dh16x4 ph, pl;
            d_split4(r.d[0], sg, ph, pl);
            *reinterpret_cast<dh16x4*>(T + d_dst) = ph;
            *reinterpret_cast<dh16x4*>(T + DIMG + d_dst) = pl;
        }
        return d_pow2(e - 14);
    };
    auto publish = [&](const f32x4 (&X)[4], float live, float inv_sg, int has, int par) {
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = d_max4(mx, X[k]);
        mx = d_wave_max(mx * live);
        const float iv = d_wave_max(inv_sg);
        int pm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) pm |= (__builtin_amdgcn_ballot_w64((has >> k) & 1) != 0ull) << k;
        if (lane == 0) {
            red[32 * par + wv] = mx;
            red[32 * par + 8 + wv] = iv;
            reinterpret_cast<int*>(red)[32 * par + 16 + wv] = pm;
        }
    };
    float C_run = 3.0e38f;
    // after the barrier that follows publish(): X pieces -> buffer T; returns the row block's type mask
    auto park_x = [&](const f32x4 (&X)[4], float live, float inv_sg, char* T, int par) {
        float xm = red[32 * par], ivm = red[32 * par + 8];
        int pm = reinterpret_cast<const int*>(red)[32 * par + 16];
#pragma unroll
        for (int u = 1; u < 8; ++u) {
            xm = fmaxf(xm, red[32 * par + u]);
            ivm = fmaxf(ivm, red[32 * par + 8 + u]);
            pm |= reinterpret_cast<const int*>(red)[32 * par + 16 + u];
        }
        int ex = (__float_as_int(xm) >> 23) & 0xff;
        ex = ex < 111 ? 111 : (ex > 187 ? 187 : ex);
        const float sxo = d_pow2(268 - ex);
        const float sgm = d_pow2(254 - ((__float_as_int(ivm) >> 23) & 0xff));
        C_run = fminf(C_run, sgm * sxo);
        const float sx = C_run * inv_sg * live;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((pm >> k) & 1) {                               // (block-uniform)
                dh16x4 ph, pl;
                d_split4(X[k], sx, ph, pl);
                *reinterpret_cast<dh16x4*>(T + 2 * DIMG + 2 * k * XIMG + x_dst) = ph;
                *reinterpret_cast<dh16x4*>(T + 2 * DIMG + (2 * k + 1) * XIMG + x_dst) = pl;
            }
        return pm;
    };

    // transposed reads (as gru_bwd_rc.hip): a 16-lane group takes rows 8 (g2 >> 1) + 4 j + (0..3), columns 16 (g2 & 1) +
    // (0..15) of 32-column block cb of an image; lane 4 q + p supplies row q, columns 4 p .. 4 p + 3
    const int kt = wv >> 1, bt = wv & 1;
    const int g2 = lane >> 4, u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3;
    auto tr_addr = [&](int cb, int j) { return (2 * cb + (g2 & 1)) * 1024 + (8 * (g2 >> 1) + 4 * j + q4) * 32 + p4 * 8; };
    const int LB0 = 2 * DIMG + 2 * kt * XIMG + tr_addr(bt, 0), LB1 = 2 * DIMG + 2 * kt * XIMG + tr_addr(bt, 1);
    const int LA0 = tr_addr(0, 0), LA1 = tr_addr(0, 1);        // a-tile a: + 2048 a

    f32x16 R[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) R[a][q] = 0.f;

    // ---- prologue: row block t0 in buffer 0; rows of t0 + step, edges of t0 + 2 steps, CSR rows of t0 + 3 steps in flight
    auto clampt = [&](int64_t t) { return t < blocks ? t : t0; };
    Rows nxt;
    Idx idx;
    Csr csr;
    int pm_cur;
    {
        const Rows first = load_rows(t0, load_idx(load_csr(t0)));
        nxt = load_rows(clampt(t0 + tstep), load_idx(load_csr(clampt(t0 + tstep))));
        idx = load_idx(load_csr(clampt(t0 + 2 * tstep)));
        csr = load_csr(clampt(t0 + 3 * tstep));
        f32x4 X[4];
        const int has = form_x(first, X);
        const float iv = park_d(first, smem);
        publish(X, first.live, iv, has, 0);
        __syncthreads();
        pm_cur = park_x(X, first.live, iv, smem, 0);
    }
    float C_acc = C_run, C_cur = C_run;
    int cur = 0;
#pragma unroll 1
    for (int64_t t = t0; t < blocks; t += tstep) {
        d_barrier();                                           // buffer `cur` is complete; the other one is free
        const char* T = smem + cur * BUF;
        char* Tn = smem + (cur ^ 1) * BUF;
        const bool has1 = t + tstep < blocks;
        // row block t + 1: X values and dagg pieces from the rows that arrived during the last contraction
        f32x4 X[4];
        float iv = 0.f, live1 = 0.f;
        int has = 0;
        if (has1) {
            has = form_x(nxt, X);
            iv = park_d(nxt, Tn);
            live1 = nxt.live;
        }
        // requests: rows of t + 2, edges of t + 3, CSR rows of t + 4 (unconditional, clamped)
        nxt = load_rows(clampt(t + 2 * tstep), idx);
        idx = load_idx(csr);
        csr = load_csr(clampt(t + 4 * tstep));
        if (__builtin_amdgcn_readfirstlane(__float_as_int(C_cur)) != __builtin_amdgcn_readfirstlane(__float_as_int(C_acc))) {
            const float ratio = C_cur / C_acc;                 // < 1, a power of two
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) R[a][q] *= ratio;
            C_acc = C_cur;
        }
        if (kt < K && ((pm_cur >> kt) & 1)) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char* Tb = T + 512 * st;                 // rows + 16
                const dh16x8 bh = d_tr8(Tb + LB0, Tb + LB1), bl = d_tr8(Tb + XIMG + LB0, Tb + XIMG + LB1);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const dh16x8 ah = d_tr8(Tb + 2048 * a + LA0, Tb + 2048 * a + LA1);
                    const dh16x8 al = d_tr8(Tb + DIMG + 2048 * a + LA0, Tb + DIMG + 2048 * a + LA1);
                    R[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, R[a], 0, 0, 0);
                    R[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, R[a], 0, 0, 0);
                    R[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, R[a], 0, 0, 0);
                }
            }
        }
        if (has1) {
            publish(X, live1, iv, has, cur ^ 1);
            d_barrier();                                       // the maxima of row block t + 1 are in LDS
            pm_cur = park_x(X, live1, iv, Tn, cur ^ 1);
            C_cur = C_run;
        }
        cur ^= 1;
    }
    if (kt < K) {
        const float inv_C = 1.0f / C_acc;
        float* out = dA + (int64_t)kt * F * F;
        const int col = 64 * half + 32 * bt + (lane & 31);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = R[a][q] * inv_C;
                if (v != 0.f) atomicAdd(out + (int64_t)(32 * a + acc_row(q, lane)) * F + col, v);
            }
    }
}

template <int F>
static int launch_da_rows(const float* dagg, const float* h, const int32_t* row_ptr, const int32_t* col_idx,
                          const int32_t* edge_type, const float* z_atom, const float* q, const float* stats, float* dA,
                          int64_t V, int64_t E, int K, hipStream_t s) {
    constexpr int lds = da_rows_lds_bytes<F>();
    static const hipError_t attr = [] {
        LdsOptIn opt_in_;
        opt_in_((const void*)da_rows_kernel<F, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        opt_in_((const void*)da_rows_kernel<F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    const int64_t blocks = (V + 31) / 32;
    int64_t gx = 256;                                          // one block per CU; a multiple of 16 (XCD x column half)
    const int64_t need = (blocks + 7) / 8 * 8 * (F / 64);
    if (gx > need) gx = need;
    const DaAtt att{z_atom, q, reinterpret_cast<const float2*>(stats)};
    if (z_atom)
        hipLaunchKernelGGL((da_rows_kernel<F, true>), dim3((unsigned)gx), dim3(512), lds, s, dagg, h, row_ptr, col_idx,
                           edge_type, att, dA, V, E, K);
    else
        hipLaunchKernelGGL((da_rows_kernel<F, false>), dim3((unsigned)gx), dim3(512), lds, s, dagg, h, row_ptr, col_idx,
                           edge_type, att, dA, V, E, K);
    return launch_status("mpnn_message_agg_bwd_da_rows_f32");
}

int launch_da_rows64(const float* dagg, const float* h, const int32_t* row_ptr, const int32_t* col_idx,
                     const int32_t* edge_type, float* dA, int64_t V, int64_t E, int K, hipStream_t s) {
    return launch_da_rows<64>(dagg, h, row_ptr, col_idx, edge_type, nullptr, nullptr, nullptr, dA, V, E, K, s);
}
int launch_da_rows128(const float* dagg, const float* h, const int32_t* row_ptr, const int32_t* col_idx,
                      const int32_t* edge_type, const float* z_atom, const float* q, const float* stats, float* dA,
                      int64_t V, int64_t E, int K, hipStream_t s) {
    return launch_da_rows<128>(dagg, h, row_ptr, col_idx, edge_type, z_atom, q, stats, dA, V, E, K, s);
}

}  // namespace mpnn

using namespace mpnn;

extern "C" int mpnn_message_agg_bwd_da_rows_f32(const float* dagg, const float* h, const int32_t* row_ptr,
                                                const int32_t* col_idx, const int32_t* edge_type, const float* z_atom,
                                                const float* q, const float* stats_by_atom, float* dA, int64_t V,
                                                int64_t E, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(V >= 0 && E >= 0, "mpnn_message_agg_bwd_da_rows_f32: negative size");
    MPNN_REQUIRE(nf == mf && (nf == 64 || nf == 128) && K >= 1 && K <= 4,
                 "mpnn_message_agg_bwd_da_rows_f32: nf = mf in {64, 128}, 1 <= K <= 4 (got %d, %d, K = %d)", nf, mf, K);
    MPNN_REQUIRE(!switches().math_fp32, "mpnn_message_agg_bwd_da_rows_f32: a split-precision kernel (MPNN_GRU_MATH=fp32 is set)");
    if (E == 0 || V == 0) return MPNN_OK;
    MPNN_REQUIRE(dagg && h && row_ptr && col_idx && edge_type && dA, "mpnn_message_agg_bwd_da_rows_f32: NULL buffer");
    MPNN_REQUIRE(!z_atom || (nf == 128 && q && stats_by_atom),
                 "mpnn_message_agg_bwd_da_rows_f32: the gated form exists at width 128 and needs q and the statistics");
    if (nf == 64) return launch_da_rows64(dagg, h, row_ptr, col_idx, edge_type, dA, V, E, K, (hipStream_t)stream);
    return launch_da_rows128(dagg, h, row_ptr, col_idx, edge_type, z_atom, q, stats_by_atom, dA, V, E, K, (hipStream_t)stream);
}
