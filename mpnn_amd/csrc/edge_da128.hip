// Weight gradient of the typed edge message at nf = mf = 128 on the bf16 matrix pipe (3-way operand splitting):
//   dA[k][a][b] += sum_{e in type k}  y[e][a] * x[e][b],   y[e] = w[e] * Y[arow(e)],  x[e] = gate[e] * h[src(e)]
// arow(e) = dst[e] (Y = d(aggregate), the fused message+aggregate backward) or e (Y = dmsg).
//
// The contraction runs over EDGES, so an MFMA fragment is 8 consecutive edges of one column.  As in
// gru_bwd_dw128_kernel a thread owns one of the 256 columns [y (128) | x (128)] of half of the step's 32 edges:
// its 16 gathered dword loads (coalesced across the wave: same row, consecutive columns) are two fragments, split
// once and parked in LDS as bf16 pieces ([piece][octet][column] 16-byte slots), double-buffered, one barrier per
// 32-edge step.  Wave (a, b-pair) owns 2 of the 4 x 4 output tiles; accumulators are flushed once per type.
#include "split_math.h"

namespace mpnn {

// ATT: the gate is AttEdgeNetwork's feature softmax, evaluated in flight instead of read from an (E, F) tensor:
//   gate[e][c] = exp2(log2e * (z_atom[dst e][c] + q[type e][c]) - stats[dst e][type e].x) * stats[dst e][type e].y
// (`gate` then points at z_atom; the statistics come from the forward's kernel, in atom order)
struct AttGateArgs {
    const float* q;          // (K, F) bond part of the logits
    const float2* stats;     // (V, K): log2e * max_f, 1 / sum_f exp
};

template <int F, bool HAS_DST, bool HAS_W, bool GATED, bool ATT = false>
__global__ void __launch_bounds__(F >= 128 ? 512 : 256) edge_da_split_kernel(
    const float* __restrict__ Y, const float* __restrict__ h, const int32_t* __restrict__ src,
    const int32_t* __restrict__ dst, const float* __restrict__ w, const int32_t* __restrict__ order,
    const int32_t* __restrict__ type_ptr, const float* __restrict__ gate, float* dA, int K, AttGateArgs att) {
    static_assert(!ATT || (GATED && HAS_DST), "the in-flight gate needs the destination list");
    constexpr int NC = 2 * F;                              // staged columns [y | x]
    constexpr int SLOT = NC * 16;                          // bytes of one (piece, octet) plane
    constexpr int NWV = F >= 128 ? 8 : 4;                  // waves per block
    constexpr int EPS = F == 256 ? 16 : 32;                // edges per step (F = 256: 512 columns x 16 edges)
    constexpr int NO = EPS / 8;                            // octets (MFMA fragments along the contraction) per step
    constexpr int NA = F / 32;                             // 32-wide output blocks per side
    constexpr int NT = NA * NA / NWV;                      // output tiles per wave (8 / 2 / 1 at F = 256 / 128 / 64)
    constexpr int BUF = 3 * NO * SLOT;                     // 3 pieces x NO octets
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;
    const int col = tid % NC;                              // staged column: < F = y, else x
    const int og = wv / (NC / 64);                         // which 16 of the step's 32 edges this thread stages
    const bool is_y = (wv * 64) % NC < F;                  // wave-uniform
    const int fcol = col % F;
    const int ta = wv % NA, tb2 = wv / NA;                 // output tiles (ta, NT*tb2 + b)

    f32x16 acc[NT];
    float raw0[16], raw1[16], aux0[16], aux1[16];
    float2 sta0 = {0.f, 0.f}, sta1 = {0.f, 0.f};           // ATT: softmax statistics of this lane's edge slot, per buffer
    for (int k = 0; k < K; ++k) {
        const int tb = type_ptr[k], te = type_ptr[k + 1];
        if (te == tb) continue;
        const int steps = (te - tb + EPS - 1) / EPS;
        const float qk = ATT ? att.q[k * F + fcol] * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;

        // Index pipeline.  Lane (l & 15) of a wave carries edge slot 16*og + (l & 15) of a step: `order` is read four
        // steps ahead, dst / src / w three steps ahead and the data rows two steps ahead, so an iteration never waits
        // for the dependent chain order -> row -> data (it used to: two scalar round trips before the data loads of
        // every step could even be issued).  Everything in the fetch is branch-free: flags are template parameters
        // and the y / x role only selects pointers.  Masking and the w / gate factors are applied when the step is
        // parked, so nothing waits at issue time.
        const int slot = 16 * og + (lane & 15);
        auto fetch_e = [&](int st) {                       // st may run past the type: clamped, masked at park()
            const int pos = tb + EPS * st + slot;
            return order[pos < te ? pos : tb];
        };
        auto fetch_r = [&](int e_l, int& r_l, float& w_l, int& d_l) {
            r_l = is_y ? (HAS_DST ? dst[e_l] : e_l) : src[e_l];
            w_l = (HAS_W && is_y) ? w[e_l] : 1.0f;
            d_l = (ATT && !is_y) ? dst[e_l] : 0;
        };
        auto load_raw = [&](float (&raw)[16], float (&aux)[16], float2& sta, int e_l, int r_l, float w_l, int d_l) {
            if (ATT && !is_y) sta = att.stats[(int64_t)d_l * K + k];
            if (is_y) {                                    // one wave-uniform branch around the whole batch
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    raw[u] = Y[(int64_t)__builtin_amdgcn_readlane(r_l, u) * F + fcol];
                    aux[u] = HAS_W ? readlane_f(w_l, u) : 1.0f;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    raw[u] = h[(int64_t)__builtin_amdgcn_readlane(r_l, u) * F + fcol];
                    aux[u] = ATT     ? gate[(int64_t)__builtin_amdgcn_readlane(d_l, u) * F + fcol]
                             : GATED ? gate[(int64_t)__builtin_amdgcn_readlane(e_l, u) * F + fcol]
                                     : 1.0f;
                }
            }
        };
        auto park = [&](int buf, int st, const float (&raw)[16], const float (&aux)[16], const float2& sta) {
            const int p0 = tb + EPS * st + 16 * og;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                f32x4 x0, x1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u0 = 8 * o + j, u1 = 8 * o + 4 + j;
                    float v0 = raw[u0], v1 = raw[u1];
                    if (ATT && !is_y) {
                        v0 *= __builtin_amdgcn_exp2f(fmaf(aux[u0], 1.4426950408889634f, qk - readlane_f(sta.x, u0))) * readlane_f(sta.y, u0);
                        v1 *= __builtin_amdgcn_exp2f(fmaf(aux[u1], 1.4426950408889634f, qk - readlane_f(sta.x, u1))) * readlane_f(sta.y, u1);
                    } else if (HAS_W || GATED) { v0 *= aux[u0]; v1 *= aux[u1]; }
                    x0[j] = p0 + u0 < te ? v0 : 0.f;
                    x1[j] = p0 + u1 < te ? v1 : 0.f;
                }
                bf16x8 ph, pm, pl;
                split8(x0, x1, ph, pm, pl);
                char* base = smem + buf * BUF + (2 * og + o) * SLOT + col * 16;
                *reinterpret_cast<bf16x8*>(base) = ph;
                *reinterpret_cast<bf16x8*>(base + NO * SLOT) = pm;
                *reinterpret_cast<bf16x8*>(base + 2 * NO * SLOT) = pl;
            }
        };
        auto frag = [&](int buf, int piece, int octet, int c) {
            return *reinterpret_cast<const bf16x8*>(smem + buf * BUF + (piece * NO + octet) * SLOT + c * 16);
        };

        int st = blockIdx.x;
        const int g = gridDim.x;
        int cur = 0;
        __syncthreads();                                    // the previous type's last buffer is no longer read
        int e2 = 0, e3 = 0, r2 = 0, d2 = 0;
        float w2 = 1.0f;
        if (st < steps) {
            int e0 = fetch_e(st), r0, d0;
            float w0;
            fetch_r(e0, r0, w0, d0);
            load_raw(raw0, aux0, sta0, e0, r0, w0, d0);
            park(0, st, raw0, aux0, sta0);
            int e1 = fetch_e(st + g), r1, d1;
            float w1;
            fetch_r(e1, r1, w1, d1);
            load_raw(raw0, aux0, sta0, e1, r1, w1, d1);     // step st+g, parked by the first iteration
            e2 = fetch_e(st + 2 * g);
            fetch_r(e2, r2, w2, d2);
            e3 = fetch_e(st + 3 * g);
        }
        // one step: fetch step st+2g into `rin`, multiply step st out of LDS, park step st+g (already in `rout`)
        auto step = [&](int st_now, float (&rout)[16], float (&aout)[16], float2& sout, float (&rin)[16], float (&ain)[16],
                        float2& sin) {
            __syncthreads();
            const int e4 = fetch_e(st_now + 4 * g);         // consumed two iterations from now
            int r3, d3;
            float w3;
            fetch_r(e3, r3, w3, d3);                        // consumed next iteration
            load_raw(rin, ain, sin, e2, r2, w2, d2);        // harmless clamped re-read past the last step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < EPS / 16; ++ks) {
                const int oc = 2 * ks + hi;
                const int ca = 32 * ta + i;
                const bf16x8 ah = frag(cur, 0, oc, ca), am = frag(cur, 1, oc, ca), al = frag(cur, 2, oc, ca);
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int cb = F + 32 * (NT * tb2 + b) + i;
                    mma6(acc[b], ah, am, al, frag(cur, 0, oc, cb), frag(cur, 1, oc, cb), frag(cur, 2, oc, cb));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st_now + g < steps) park(cur ^ 1, st_now + g, rout, aout, sout);
            cur ^= 1;
            e2 = e3;
            r2 = r3;
            w2 = w3;
            d2 = d3;
            e3 = e4;
        };
        for (; st < steps; st += 2 * g) {
            step(st, raw0, aux0, sta0, raw1, aux1, sta1);
            if (st + g < steps) step(st + g, raw1, aux1, sta1, raw0, aux0, sta0);
        }
        if ((int)blockIdx.x < steps) {
            float* out = dA + (int64_t)k * F * F;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int c = 32 * (NT * tb2 + b) + i;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float v = acc[b][q];
                    if (v != 0.f) atomicAdd(out + (int64_t)(32 * ta + acc_row(q, lane)) * F + c, v);
                }
            }
        }
    }
}

template <int F>
static int launch_edge_da_split(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                                const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                                int K, hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * (F == 256 ? 2 : 4) * (2 * F) * 16;   // 96 KB at F >= 128 (1 block / CU), 48 KB at F = 64 (3)
    int64_t gx = F >= 128 ? 256 : 768;
    const int64_t need = ceil_div(E, F == 256 ? 16 : 32) + K;
    if (gx > need) gx = need;
#define MPNN_DA(D, W, G)                                                                                            \
    do {                                                                                                            \
        static const hipError_t attr_done = [&] { LdsOptIn opt_in_; opt_in_((const void*)edge_da_split_kernel<F, D, W, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); return opt_in_.err; }();  /* once, thread-safe */ \
        if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);                                                                                                           \
        hipLaunchKernelGGL((edge_da_split_kernel<F, D, W, G>), dim3((unsigned)gx), dim3(F >= 128 ? 512 : 256), lds, s, Y, h, src, dst, \
                           w, order, type_ptr, gate, dA, K, AttGateArgs{nullptr, nullptr});                         \
    } while (0)
    const bool hw = dst && w;
    if (!dst) { if (gate) MPNN_DA(false, false, true); else MPNN_DA(false, false, false); }
    else if (!hw) { if (gate) MPNN_DA(true, false, true); else MPNN_DA(true, false, false); }
    else { if (gate) MPNN_DA(true, true, true); else MPNN_DA(true, true, false); }
#undef MPNN_DA
    return launch_status("mpnn_edge_message_bwd_f32(dA, bf16x6)");
}

// width 128 with AttEdgeNetwork's gate evaluated in flight (see AttGateArgs)
int launch_edge_da_att128(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const int32_t* order,
                          const int32_t* type_ptr, const float* z_atom, const float* q, const float* stats_by_atom, float* dA,
                          int64_t E, int K, hipStream_t s) {
    constexpr int F = 128;
    const size_t lds = (size_t)2 * 3 * 4 * (2 * F) * 16;
    static const hipError_t attr_done = [&] {
        LdsOptIn opt_in_;
        opt_in_((const void*)edge_da_split_kernel<F, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    int64_t gx = 256;
    const int64_t need = ceil_div(E, 32) + K;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((edge_da_split_kernel<F, true, false, true, true>), dim3((unsigned)gx), dim3(512), lds, s, Y, h, src, dst,
                       nullptr, order, type_ptr, z_atom, dA, K, AttGateArgs{q, reinterpret_cast<const float2*>(stats_by_atom)});
    return launch_status("mpnn_edge_message_agg_bwd_da_att_f32");
}

int launch_edge_da_split128(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                            const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                            int K, hipStream_t s) {
    return launch_edge_da_split<128>(Y, h, src, dst, w, order, type_ptr, gate, dA, E, K, s);
}
int launch_edge_da_split256(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                            const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                            int K, hipStream_t s) {
    return launch_edge_da_split<256>(Y, h, src, dst, w, order, type_ptr, gate, dA, E, K, s);
}
int launch_edge_da_split64(const float* Y, const float* h, const int32_t* src, const int32_t* dst, const float* w,
                           const int32_t* order, const int32_t* type_ptr, const float* gate, float* dA, int64_t E,
                           int K, hipStream_t s) {
    return launch_edge_da_split<64>(Y, h, src, dst, w, order, type_ptr, gate, dA, E, K, s);
}

}  // namespace mpnn
