// Typed edge message FUSED with the neighbour sum at nf = mf = 128 / 256 (and 64), molecules of up to 256 atoms:
//   out[i] = sum_{e in row i} A[type e] . h[src e]  =  sum_k A_k . S_k[i],   S_k[i] = sum_{e in row i, type k} h[src e]
// replaces: mpnn_functions/message/edge_network.py:40,52 (per-pair product) composed with
//           mpnn_functions/message_aggregators/adjacent_message_agg.py:18 (the neighbour sum), i.e. what
//           edge_network.py:50-51 computes as one bmm.  No (E, mf) message tensor exists in HBM: the h rows are read
//           once, the out rows written once.
//
// At these widths neither the bond-type matrices (K x F x F: 1 MB at F = 256) nor a tile's h rows (256 x 1 KB) fit in
// LDS, so the contraction dimension is cut into 32-wide chunks and the sum over neighbours is taken BEFORE the product
// (typed aggregate-then-contract: the message is linear in h, so A_k . sum = sum A_k .):
//   * a persistent 8-wave block walks molecule-aligned tiles of at most 256 atoms (graph.py::WidePlan); wave w owns
//     block w of the tile = 32 atoms sorted next to each other by their bond-type pattern, and keeps their 32 x F output
//     rows in MFMA accumulators for the whole tile;
//   * per 32-column chunk kc of the h rows (128 bytes per row = one line, copied global -> LDS without touching
//     registers, one chunk ahead) and per bond type k (the chunk of A_k as two fp16 pieces, 64 bytes per output column,
//     copied the same way from the pre-split workspace, one phase ahead): a wave whose block has type-k edges sums,
//     per atom, the chunk of its type-k neighbours' rows from the LDS image (slot words name the source rows; an atom
//     without a rank-th neighbour reads a row of zeros), splits the 32 sums into two fp16 pieces behind a per-atom
//     power-of-two scale, and contracts them with the A_k chunk: 3 x F / 32 x 2 v_mfma_f32_32x32x16_f16 per phase;
//   * a (block, type) pair without edges skips its phase's work; one block barrier per phase;
//   * sum order inside an output row: types ascending, within a type the CSR edge order -- deterministic, no atomics.
//
// Math ("fp16x3", as the GRU kernels): operands are split as x*s = hi + lo with hi = fp16(x*s), lo = fp16(x*s - hi);
// three MFMAs per product (lo*hi, hi*lo, hi*hi), every partial product exact in fp32; dropped: lo*lo (2^-22 relative).
// The matrices share one power-of-two scale (largest |A| in [2^14, 2^15)); every atom's sums have their own: a row
// scale factors out of the product, it is chosen from the first non-zero fragment with eight-fold headroom and lowered
// (accumulator entries multiplied by the ratio, a power of two) if a later fragment outgrows it.
#include <stdlib.h>

#include "common.h"

namespace mpnn {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#ifdef MB_STAMP   // diagnostic build only (-DMB_STAMP): cycle sums of the gated backward kernel, block 3, wave 0
__device__ unsigned long long g_mb_stamps[16];
#define MB_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define MB_ADD(i, v) do { if (blockIdx.x == 3 && tid == 0) g_mb_stamps[i] += (v); } while (0)
#else
#define MB_T(var)
#define MB_ADD(i, v)
#endif
#ifdef MW_STAMP   // diagnostic build only (-DMW_STAMP): cycle sums per phase part of block 3, waves 0 and 7
__device__ unsigned long long g_mw_stamps[32];
#define MW_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#else
#define MW_T(var)
#endif

namespace {
constexpr int MW_TV = 256;          // atoms per tile (upper bound); LDS image row MW_TV is all zeros
constexpr int MW_NB = 8;            // blocks of 32 atoms = waves
constexpr int MW_ROWS = 256;        // slot rows of a tile parked in LDS
constexpr int MW_KMAX = 8;
constexpr int MW_HB = (MW_TV + 1) * 128;       // one h chunk image: 257 rows x 128 bytes

template <int F>
__host__ __device__ constexpr int mw_lds_bytes() {
    return 2 * MW_HB + 2 * F * 128 + MW_ROWS * 64 + MW_TV * 4 + (MW_NB * MW_KMAX + 1) * 4 + 64;
}

__device__ __forceinline__ void mw_barrier() {       // orders LDS traffic only (the copies are waited for by hand)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One wave copies 1 KB: lane L's 16 bytes at `src` land at LDS byte lds_dst + 16 L (gru_bwd128_f16.hip has the why).
__device__ __forceinline__ void mw_copy(const char* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    typedef __attribute__((address_space(3))) const char lds_char;
    return (unsigned)(uintptr_t)(lds_char*)p;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ matrices -> pieces
// workspace: [0, 64): float scale inverse (and padding); then chunk (kc, k) at 64 + (kc * K + k) * F * 128 bytes:
// [piece hi | lo][output column n][32 halves] with the 16-byte slots of a column XOR-ed by (n >> 2) & 3 (the LDS image
// the GRU kernels read conflict-free); halves 8 o + j of slot o = A_k[n][32 kc + 8 o + j].
__global__ void __launch_bounds__(1024) mw_absmax_kernel(const float* __restrict__ A, int64_t n, float* __restrict__ ws) {
    __shared__ float red[16];
    float mx = 0.f;
    for (int64_t i = threadIdx.x; i < n / 4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + 4 * i);
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(v[u]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int u = 1; u < 16; ++u) mx = fmaxf(mx, red[u]);
        int e = (__float_as_int(mx) >> 23) & 0xff;
        e = e < 20 ? 20 : (e > 250 ? 250 : e);
        ws[0] = __int_as_float((268 - e) << 23);            // scale: largest |A| lands in [2^14, 2^15)
        ws[1] = __int_as_float((e - 14) << 23);             // its inverse
    }
}

template <int F>
__global__ void __launch_bounds__(256) mw_split_kernel(const float* __restrict__ A, char* __restrict__ ws, int K) {
    // one thread per (k, n, octet o): 8 consecutive input features of output column n
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_k = F * (F / 8);
    if (idx >= (int64_t)K * per_k) return;
    const int k = (int)(idx / per_k), rem = (int)(idx % per_k);
    const int n = rem / (F / 8), oc = rem % (F / 8);         // oc = global octet: chunk kc = oc >> 2, slot o = oc & 3
    const float sc = reinterpret_cast<const float*>(ws)[0];
    const float* p = A + ((int64_t)k * F + n) * F + 8 * oc;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4);
    h16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * sc, b = x1[j] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
        ph[4 + j] = (_Float16)b;
        pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
    }
    const int kc = oc >> 2, o = oc & 3;
    char* dst = ws + 64 + ((int64_t)kc * K + k) * (F * 128) + n * 64 + ((o ^ ((n >> 2) & 3)) << 4);
    *reinterpret_cast<h16x8*>(dst) = ph;
    *reinterpret_cast<h16x8*>(dst + F * 64) = pl;
}

// ---- the feature gate of AttEdgeNetwork inside the fused kernel (att_edge_network.py:18-21) ----
// gate[e, :] = softmax_f(z_atom[dst e, :] + q[type e, :]) depends on the DESTINATION atom and the bond type only, so it
// factorises like the message itself:  out[i] = sum_k A_k (g_ik * S_k[i]),  g_ik = softmax_f(z_atom[i] + q[k]).  The
// kernel multiplies its S_k fragment by g_ik before the split; what it needs per (atom, type) are the two softmax
// statistics, written here in the plan's sorted tile order: stats[(tile * 256 + position) * K + k] =
// (log2(e) * max_f(z + q), 1 / sum_f exp(z + q - max)).  A tile per block pass, 32 lanes per atom (F = 128).
constexpr float MW_LOG2E = 1.4426950408889634f;
// Eight lanes per atom, sixteen logits per lane (a 64-byte piece of the row), the K <= 4 types reduced side by side (four
// independent shuffle chains): 32 lanes x one type at a time was a chain of ten dependent cross-lane steps per (atom, type)
// and took 0.85 ms at c3's size against 0.3 ms of bytes.
__global__ void __launch_bounds__(512) mw_gate_stats_kernel(const float* __restrict__ z_atom, const float* __restrict__ q,
                                                            const int32_t* __restrict__ tile_rec,
                                                            const int32_t* __restrict__ tile_atom, float2* __restrict__ stats,
                                                            int num_tiles, int K) {
    constexpr int F = 128, KM = 4;
    __shared__ float qs[KM * F];
    for (int i = threadIdx.x; i < KM * F; i += 512) qs[i] = i < K * F ? q[i] : 0.f;
    __syncthreads();
    const int grp = threadIdx.x >> 3, l = threadIdx.x & 7;          // 64 atom groups of 8 lanes; lane l: columns 16 l ...
    for (int t = blockIdx.x; t < num_tiles; t += gridDim.x) {
        const int n = tile_rec[4 * t + 1];
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + grp;
            const bool ok = p < n;
            const int atom = ok ? tile_atom[(int64_t)t * MW_TV + p] : 0;
            const float* zr = z_atom + (int64_t)(atom < 0 ? 0 : atom) * F + 16 * l;
            f32x4 zv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) zv[j] = *reinterpret_cast<const f32x4*>(zr + 4 * j);
            float m[KM], e[KM];
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                float mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = zv[j] + *reinterpret_cast<const f32x4*>(qs + k * F + 16 * l + 4 * j);
                    mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                }
                m[k] = mx;
            }
#pragma unroll
            for (int o = 4; o >= 1; o >>= 1)
#pragma unroll
                for (int k = 0; k < KM; ++k) m[k] = fmaxf(m[k], __shfl_xor(m[k], o));
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = zv[j] + *reinterpret_cast<const f32x4*>(qs + k * F + 16 * l + 4 * j);
#pragma unroll
                    for (int u = 0; u < 4; ++u) sm += __builtin_amdgcn_exp2f((v[u] - m[k]) * MW_LOG2E);
                }
                e[k] = sm;
            }
#pragma unroll
            for (int o = 4; o >= 1; o >>= 1)
#pragma unroll
                for (int k = 0; k < KM; ++k) e[k] += __shfl_xor(e[k], o);
            if (ok && l < K) {
                const float mk = l == 0 ? m[0] : l == 1 ? m[1] : l == 2 ? m[2] : m[3];
                const float ek = l == 0 ? e[0] : l == 1 ? e[1] : l == 2 ? e[2] : e[3];
                stats[((int64_t)t * MW_TV + p) * K + l] = float2{mk * MW_LOG2E, 1.0f / ek};
            }
        }
    }
}

struct MwGate {
    const float* z_atom;   // (V, F) atom part of the gate logits
    const float* q;        // (K, F) bond part
    const float2* stats;   // from mw_gate_stats_kernel
};

// ------------------------------------------------------------------------------------------------------ the kernel
template <int F, bool GATED = false>
__global__ void __launch_bounds__(512) message_sum_wide_kernel(
    const float* __restrict__ h, const char* __restrict__ ws, const int32_t* __restrict__ tile_rec,
    const int32_t* __restrict__ tile_atom, const int32_t* __restrict__ blk_off, const int16_t* __restrict__ slots,
    float* __restrict__ out, int num_tiles, int K, int dbg, MwGate gt) {
    static_assert(!GATED || F == 128, "the gated form exists at width 128");
    constexpr int NKC = F / 32, CT = F / 32, ABUF = F * 128;
    __shared__ float qs_s[GATED ? 4 * F : 1];              // GATED: log2(e) * q, at most four bond types
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const HB = smem;                               // two h chunk images
    char* const AB = smem + 2 * MW_HB;                   // two matrix chunk images
    int16_t* const SL = reinterpret_cast<int16_t*>(AB + 2 * ABUF);
    int* const AT = reinterpret_cast<int*>(reinterpret_cast<char*>(SL) + MW_ROWS * 64);
    int* const OFF = AT + MW_TV;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const float a_inv = reinterpret_cast<const float*>(ws)[1];
    const char* const wsA = ws + 64;
    const int nphase = NKC * K;
    if (GATED)
        for (int i = tid; i < K * F; i += 512) qs_s[i] = gt.q[i] * MW_LOG2E;   // (published by the first tile's barriers)

    // the zero rows of the two h images: slot word MW_TV reads them
    if (tid < 16) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(HB + MW_TV * 128 + 16 * (tid & 7) + (tid >> 3) * MW_HB) = z;
    }

    // copies: an h chunk = 256 rows x 128 bytes = 32 wave-instructions (8 rows each), four per wave; lane = (row, 16-byte
    // slot c): the source is slot c ^ (row & 7) of the row's line, so a reader finds slot j at position j ^ (row & 7)
    // (rank, share): this wave copies pieces rank, rank + share, ... -- the waves whose block has no work in a phase take
    // the phase's copies off the others (in-kernel stamps: issuing its share and waiting for it cost the one wave that is
    // busy in EVERY phase 1,400 of its 4,600 cycles per phase, while the other waves sat at the barrier)
    auto copy_h = [&](int a0, int n, int kc, int buf, int rank, int share) {
        for (int row8 = rank; row8 < 32; row8 += share) {  // group of 8 rows
            const int row = 8 * row8 + (lane >> 3), c = lane & 7;
            const int rr = row < n ? row : n - 1;
            const char* src = reinterpret_cast<const char*>(h + (int64_t)(a0 + rr) * F + 32 * kc) + ((c ^ (row & 7)) << 4);
            mw_copy(src, lds_addr(HB + buf * MW_HB + row8 * 1024));
        }
    };
    // a matrix chunk = F x 128 bytes, contiguous in the workspace: F / 8 wave-instructions
    auto copy_a = [&](int phase, int buf, int rank, int share) {
        const char* src = wsA + (int64_t)phase * ABUF + lane * 16;
        for (int blk = rank; blk < F / 8; blk += share) mw_copy(src + blk * 1024, lds_addr(AB + buf * ABUF + blk * 1024));
    };

    // ---- per-wave pieces of a phase
    // S: per atom (lane r, both halves) the sum of its type-k neighbours' rows of the chunk in `hb`.  The first ranks are
    // straight-line code (slot words first, then every row read, then the adds: two LDS round trips in all; a missing
    // rank reads the row of zeros), further ranks -- hubs -- run in a loop with the slot word fetched one rank ahead.
    auto add_row = [&](const char* hb, int w, f32x4 (&s)[2][2]) {
        const char* row = hb + w * 128;
        const int sw = w & 7;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int c0 = 2 * (2 * st + hi);             // 16-byte slots of columns 16 st + 8 hi ...
            s[st][0] += *reinterpret_cast<const f32x4*>(row + ((c0 ^ sw) << 4));
            s[st][1] += *reinterpret_cast<const f32x4*>(row + (((c0 + 1) ^ sw) << 4));
        }
    };
    auto gather = [&](const char* hb, int base, int cnt, f32x4 (&s)[2][2]) {
#pragma unroll
        for (int st = 0; st < 2; ++st) s[st][0] = s[st][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int16_t* sl = SL + base * 32 + r;
        int w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ww = sl[32 * (q < cnt ? q : 0)];
            w[q] = q < cnt ? ww : MW_TV;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) add_row(hb, w[q], s);
        if (cnt > 2) {
#pragma unroll
            for (int q = 2; q < 4; ++q) add_row(hb, w[q], s);
            if (cnt > 4) {
                int wq = sl[32 * 4];
                for (int q = 4; q < cnt; ++q) {
                    const int wn = sl[32 * (q + 1 < cnt ? q + 1 : q)];
                    add_row(hb, wq, s);
                    wq = wn;
                }
            }
        }
    };

    int t = blockIdx.x;
    int a0 = 0, n = 1;
    if (t < num_tiles) {
        a0 = tile_rec[4 * t];
        n = tile_rec[4 * t + 1];
        copy_h(a0, n, 0, 0, wv, 8);                       // the first tile's first images; later tiles' are requested
        copy_a(0, 0, wv, 8);                              // during the previous tile's last chunk
    }
    for (; t < num_tiles; t += gridDim.x) {
        const int row0 = tile_rec[4 * t + 2], nrows = tile_rec[4 * t + 3];
        const int tn = t + (int)gridDim.x;
        const bool more = tn < num_tiles;
        const int a0n = more ? tile_rec[4 * tn] : a0, nn = more ? tile_rec[4 * tn + 1] : n;
        mw_barrier();                                     // every wave is done with the previous tile's index data
        // ---- the tile's index data: slot rows (64 bytes each), sorted-atom list, (block, type) offsets
        {
            // every load goes out before the first LDS store: as load -> store pairs (a loop over the slot rows, then the
            // two lists) the compiler waited for each load in turn, four memory round trips per tile with all waves idle
            const int4* sp = reinterpret_cast<const int4*>(slots + (int64_t)row0 * 32);
            const int nq = nrows * 4;                      // 16-byte quads of slot words: at most 1024 (256 slot rows)
            int4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            int at_v = 0, off_v = 0;
            if (tid < nq) q0 = sp[tid];
            if (tid + 512 < nq) q1 = sp[tid + 512];
            if (tid < MW_TV) at_v = tile_atom[(int64_t)t * MW_TV + tid];
            if (tid < MW_NB * K + 1) off_v = blk_off[(int64_t)t * (MW_NB * K + 1) + tid];
            __builtin_amdgcn_sched_barrier(0);
            if (tid < nq) reinterpret_cast<int4*>(SL)[tid] = q0;
            if (tid + 512 < nq) reinterpret_cast<int4*>(SL)[tid + 512] = q1;
            if (tid < MW_TV) AT[tid] = at_v;
            if (tid < MW_NB * K + 1) OFF[tid] = off_v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mw_barrier();
        // which bond types my block has (wave-uniform bit mask)
        int amask = 0;
        for (int k = 0; k < K; ++k)
            amask |= (__builtin_amdgcn_readfirstlane(OFF[wv * K + k + 1]) > __builtin_amdgcn_readfirstlane(OFF[wv * K + k])) << k;
        if (dbg & 8) amask = 0;
        // ... and which (block, type) pairs of the whole tile have work: bit b * K + k
        const unsigned long long pairs = __builtin_amdgcn_ballot_w64(lane < MW_NB * K && OFF[lane < MW_NB * K ? lane + 1 : 1] >
                                                                                       OFF[lane < MW_NB * K ? lane : 0]);

        // GATED: this lane's atom (block wv, row r): its logits row, its softmax statistics per bond type, and the chunk of
        // logits the current phases multiply with (`zc`) plus the next chunk's (`zn`, requested a chunk ahead)
        const float* zp = nullptr;
        float sm0 = 0.f, sm1 = 0.f, sm2 = 0.f, sm3 = 0.f, si0 = 0.f, si1 = 0.f, si2 = 0.f, si3 = 0.f;
        // (named registers: as arrays handed to lambdas they lived in scratch)
        f32x4 zc00, zc01, zc10, zc11, zn00, zn01, zn10, zn11;
#define MW_ZLOAD(KC)                                                                  \
    zn00 = *reinterpret_cast<const f32x4*>(zp + 32 * (KC));                           \
    zn01 = *reinterpret_cast<const f32x4*>(zp + 32 * (KC) + 4);                       \
    zn10 = *reinterpret_cast<const f32x4*>(zp + 32 * (KC) + 16);                      \
    zn11 = *reinterpret_cast<const f32x4*>(zp + 32 * (KC) + 20)
        if (GATED) {
            const int at = AT[32 * wv + r];
            zp = gt.z_atom + (int64_t)(at < 0 ? 0 : at) * F + 8 * hi;
            const float2* sp2 = gt.stats + ((int64_t)t * MW_TV + 32 * wv + r) * K;
            const float2 a = sp2[0], b = sp2[K > 1 ? 1 : 0], c2 = sp2[K > 2 ? 2 : 0], d = sp2[K > 3 ? 3 : 0];
            sm0 = a.x; si0 = a.y; sm1 = b.x; si1 = b.y; sm2 = c2.x; si2 = c2.y; sm3 = d.x; si3 = d.y;
            MW_ZLOAD(0);
        }
        // s *= softmax_f(z + q_k) on the chunk's columns of this lane: exp2(z L + (q_k L - max L)) / sum
        auto apply_gate = [&](int kc, int k, f32x4 (&s)[2][2]) {
            const float m = k == 0 ? sm0 : k == 1 ? sm1 : k == 2 ? sm2 : sm3;
            const float iv = k == 0 ? si0 : k == 1 ? si1 : k == 2 ? si2 : si3;
            const float* qp = qs_s + k * F + 32 * kc + 8 * hi;
            auto one = [&](f32x4& sv, const f32x4& zv, const float* qq) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qq);
#pragma unroll
                for (int j = 0; j < 4; ++j) sv[j] *= __builtin_amdgcn_exp2f(fmaf(zv[j], MW_LOG2E, qv[j] - m)) * iv;
            };
            one(s[0][0], zc00, qp);
            one(s[0][1], zc01, qp + 4);
            one(s[1][0], zc10, qp + 16);
            one(s[1][1], zc11, qp + 20);
        };

        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        float row_sc = 0.f, row_inv = 0.f;                // 0 = this atom has not seen a non-zero sum yet
        f32x4 s[2][2];                                    // sums of the phase that multiplies next
        bool have = false;                                // ... already gathered (during the previous phase's products)

        // range guard + split of the sums in `s`: one power-of-two scale per atom, lowered when a fragment outgrows it
        auto guard_split = [&](h16x8 (&ah)[2], h16x8 (&al)[2]) {
            float mx = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fabsf(s[st][0][u]), fabsf(s[st][1][u])));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const bool unset = row_sc == 0.f;
            const bool grow = unset ? mx > 0.f : mx * row_sc >= 32768.0f;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                int e = (__float_as_int(mx) >> 23) & 0xff;
                e = e < 40 ? 40 : (e > 240 ? 240 : e);
                const float ns = grow ? __int_as_float((265 - e) << 23) : row_sc;    // mx * ns in [2^11, 2^12)
                const float ni = grow ? __int_as_float((e - 11) << 23) : row_inv;
                if (__builtin_amdgcn_ballot_w64(grow && !unset) != 0) {
                    const float ratio = unset ? 1.0f : ns * row_inv;   // lane j (< 32): factor of the block's atom j
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int dr = 8 * (i >> 2) + (i & 3);
                        const float f_lo = readlane_f(ratio, dr), f_hi = readlane_f(ratio, 4 + dr);
                        const float f = hi ? f_hi : f_lo;
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[c][i] *= f;
                    }
                }
                row_sc = ns;
                row_inv = ni;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = s[st][0][j] * row_sc, b = s[st][1][j] * row_sc;
                    ah[st][j] = (_Float16)a;
                    al[st][j] = (_Float16)(a - (float)ah[st][j]);
                    ah[st][4 + j] = (_Float16)b;
                    al[st][4 + j] = (_Float16)(b - (float)ah[st][4 + j]);
                }
        };

        for (int ph = 0; ph < nphase; ++ph) {
            const int kc = ph / K, k = ph - kc * K;
            MW_T(s0);
            mw_barrier();                                 // this phase's images are complete, last phase's are free
            MW_T(s1);
            // who copies: the waves without work in this phase, or everybody when every block has some
            int idle = 0;
            for (int b = 0; b < MW_NB; ++b) idle |= (int)(((pairs >> (b * K + k)) & 1ull) ^ 1ull) << b;
            const bool i_copy = idle == 0 || ((idle >> wv) & 1);
            const int share = idle == 0 ? MW_NB : __builtin_popcount(idle);
            const int rank = idle == 0 ? wv : __builtin_popcount(idle & ((1 << wv) - 1));
            if (!(dbg & 4) && i_copy) {
                if (ph + 1 < nphase) copy_a(ph + 1, (ph + 1) & 1, rank, share);
                else if (more) copy_a(0, 0, rank, share);   // next tile, first phase (NKC is even: buffer 0 is free)
                if (k == 0) {
                    if (kc + 1 < NKC) copy_h(a0, n, kc + 1, (kc + 1) & 1, rank, share);
                    else if (more) copy_h(a0n, nn, 0, 0, rank, share);
                }
            }
            if (GATED && k == 0 && amask != 0) {          // a new chunk: its logits arrived during the last one
                zc00 = zn00; zc01 = zn01; zc10 = zn10; zc11 = zn11;
                if (kc + 1 < NKC) { MW_ZLOAD(kc + 1); }
            }
            if ((amask >> k) & 1) {
                const char* hb = HB + (kc & 1) * MW_HB;
                const char* ab = AB + (ph & 1) * ABUF;
                if (!have) {
                    const int b0 = __builtin_amdgcn_readfirstlane(OFF[wv * K + k]);
                    gather(hb, b0, __builtin_amdgcn_readfirstlane(OFF[wv * K + k + 1]) - b0, s);
                    if (GATED) apply_gate(kc, k, s);
                }
                MW_T(s2);
                h16x8 ah[2], al[2];
                guard_split(ah, al);
                MW_T(s3);
                // the next type of my block in this chunk: its sums are gathered while this phase's products run
                const int rest = amask >> (k + 1);
                have = rest != 0;
                if (have) {
                    const int kn = k + 1 + __builtin_ctz(rest);
                    const int bn = __builtin_amdgcn_readfirstlane(OFF[wv * K + kn]);
                    gather(hb, bn, __builtin_amdgcn_readfirstlane(OFF[wv * K + kn + 1]) - bn, s);
                    if (GATED) apply_gate(kc, kn, s);
                }
                if (!(dbg & 2)) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int o = 2 * st + hi;
#pragma unroll
                        for (int c = 0; c < CT; c += 2) {
                            const int n0 = 32 * c + r, n1 = n0 + 32;
                            const char* p0 = ab + n0 * 64 + ((o ^ ((n0 >> 2) & 3)) << 4);
                            const char* p1 = ab + n1 * 64 + ((o ^ ((n1 >> 2) & 3)) << 4);
                            const h16x8 b0h = *reinterpret_cast<const h16x8*>(p0), b0l = *reinterpret_cast<const h16x8*>(p0 + F * 64);
                            const h16x8 b1h = *reinterpret_cast<const h16x8*>(p1), b1l = *reinterpret_cast<const h16x8*>(p1 + F * 64);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], b0h, acc[c], 0, 0, 0);
                            acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], b1h, acc[c + 1], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], b0l, acc[c], 0, 0, 0);
                            acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], b1l, acc[c + 1], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], b0h, acc[c], 0, 0, 0);
                            acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], b1h, acc[c + 1], 0, 0, 0);
                        }
                    }
                } else {
                    asm volatile("" ::"v"(ah[0]), "v"(al[0]), "v"(ah[1]), "v"(al[1]));
                }
#ifdef MW_STAMP
                MW_T(s4);
                if (blockIdx.x == 3 && lane == 0 && (wv == 0 || wv == 7)) {
                    unsigned long long* q = g_mw_stamps + 16 * (wv == 7);
                    atomicAdd(q + 4 + 0, s2 - s1); atomicAdd(q + 4 + 1, s3 - s2); atomicAdd(q + 4 + 2, s4 - s3); atomicAdd(q + 4 + 3, 1ull);
                }
#endif
            }
            MW_T(s5);
            if (i_copy) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my copies for the next phase have landed
#ifdef MW_STAMP
            MW_T(s6);
            if (blockIdx.x == 3 && lane == 0 && (wv == 0 || wv == 7)) {
                unsigned long long* q = g_mw_stamps + 16 * (wv == 7);
                atomicAdd(q + 0, s1 - s0); atomicAdd(q + 1, s5 - s1); atomicAdd(q + 2, s6 - s5); atomicAdd(q + 3, 1ull);
            }
#endif
        }
        // ---- out rows: accumulator (row 8 (i >> 2) + (i & 3) + 4 hi, column 32 c + r) scaled back, at the atom's place
        int atom[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) atom[i] = AT[32 * wv + 8 * (i >> 2) + (i & 3) + 4 * hi];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int dr = 8 * (i >> 2) + (i & 3);
            const float u_lo = readlane_f(row_inv, dr), u_hi = readlane_f(row_inv, 4 + dr);
            const float un = (hi ? u_hi : u_lo) * a_inv;
            if (atom[i] >= 0) {
                float* o = out + (int64_t)atom[i] * F + r;
#pragma unroll
                for (int c = 0; c < CT; ++c) __builtin_nontemporal_store(acc[c][i] * un, o + 32 * c);
            }
        }
        a0 = a0n;
        n = nn;
    }
}

#undef MW_ZLOAD

// ------------------------------------------------------------------------------ gated message + sum: backward per (atom, type)
// att_edge_network.py:18-31 composed with adjacent_message_agg.py:18, differentiated for the gate logits.  With
//   out[i] = sum_k A_k X_ik,   X_ik = g_ik * S_ik,   g_ik = softmax_f(z_i + q_k),   S_ik = sum_{e in row i, type k} h[src e]
// the gradient of the logits l_ik = z_i + q_k of an (atom, type) pair with edges is
//   T_ik = A_k^T dout_i,   u_ik = X_ik * T_ik,   D_ik = sum_f u_ikf,   dl_ik = u_ik - g_ik D_ik
// and dz_i = sum_k dl_ik, dq_k = sum_i dl_ik.  No (E, F) gate or gate-gradient tensor exists: the kernel walks the forward's
// plan -- the same tiles, blocks, phases (32-column chunk kc of the h rows x bond type k), LDS images and copies -- and per
// phase forms S (from the h chunk image), g (from the atom's logits and the forward's softmax statistics) and the chunk of
// T^T = (A_k^T chunk) . dout^T on the matrix pipe: the chunk of A_k^T is the A operand (from the pre-split LDS image), the
// block's 32 dout rows are the B operand -- split ONCE per tile behind an exact per-atom power-of-two scale and held in
// registers (64) for the tile's 16 phases.  The accumulator lane (atom, half) then holds the T entries of ITS atom: S, g and
// the logits are fetched in that layout (16-byte pieces at columns 8 q + 4 half), and every later step is per-lane.
//   * D_ik spans all four chunks, so the first term sum_k u_ik is written to dz per chunk (16 registers, not 64) and the
//     second term is subtracted when the tile's phases are done (the rows are still in L2): g is re-evaluated, one exp each;
//   * dq: the atom sums of the first term are column sums of A_k * dA_k (dA_k = sum_i dout_i (x) X_ik is the weight gradient
//     the caller has anyway), so the kernel only reduces the second term over its block's atoms (through a per-wave LDS
//     transposition: 80 dependent DPP adds per (chunk, type) took 30 % of the kernel) into per-wave LDS accumulators,
//     written once per wave as dq_part rows; the caller adds them up (fixed order: deterministic);
//   * stats_atom: the forward's softmax statistics in ATOM order, for the weight-gradient kernel's in-flight gate.
constexpr int MB_DQ = MW_NB * 4 * 128 * 4;              // per-wave dq accumulators: [wave][type <= 4][128 columns] floats
constexpr int MB_STG = MW_NB * 2048;                    // per-wave staging: 16 rows x 128 bytes
__host__ __device__ constexpr int mb_lds_bytes() { return ((mw_lds_bytes<128>() + 15) & ~15) + MB_DQ + MB_STG; }

// A_k^T chunk images: chunk (kc, k) at 64 + (kc * K + k) * 16 KB: [piece hi | lo][nf column n of the chunk (32)][16 slots of
// 8 halves]; slot oc ^ (n & 15) holds A_k[mf = 8 oc + j][nf = 32 kc + n] * scale, j = 0..7 (an A fragment = 32 rows x 16
// bytes with the slots of 16 consecutive rows all different: conflict-free ds_read_b128)
__global__ void __launch_bounds__(256) mw_split_t_kernel(const float* __restrict__ A, char* __restrict__ ws, int K) {
    constexpr int F = 128;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_k = F * (F / 8);
    if (idx >= (int64_t)K * per_k) return;
    const int k = (int)(idx / per_k), rem = (int)(idx % per_k);
    const int oc = rem / F, col = rem % F;               // consecutive threads: consecutive nf columns of one mf octet
    const int kc = col >> 5, n = col & 31;
    const float sc = reinterpret_cast<const float*>(ws)[0];
    const float* p = A + ((int64_t)k * F + 8 * oc) * F + col;
    h16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = p[(int64_t)j * F] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
    }
    char* dst = ws + 64 + ((int64_t)kc * K + k) * (F * 128) + n * (2 * F) + ((oc ^ (n & 15)) << 4);
    *reinterpret_cast<h16x8*>(dst) = ph;
    *reinterpret_cast<h16x8*>(dst + F * 64) = pl;
}

__global__ void __launch_bounds__(512) att_message_bwd_tile_kernel(
    const float* __restrict__ h, const float* __restrict__ dout, const char* __restrict__ ws,
    const int32_t* __restrict__ tile_rec, const int32_t* __restrict__ tile_atom, const int32_t* __restrict__ blk_off,
    const int16_t* __restrict__ slots, MwGate gt, float* __restrict__ dz, float* __restrict__ dq_part,
    float2* __restrict__ stats_atom, int num_tiles, int K) {
    constexpr int F = 128, NKC = 4, ABUF = F * 128;
    __shared__ float qs_s[4 * F];                         // log2(e) * q
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const HB = smem;
    char* const AB = smem + 2 * MW_HB;
    int16_t* const SL = reinterpret_cast<int16_t*>(AB + 2 * ABUF);
    int* const AT = reinterpret_cast<int*>(reinterpret_cast<char*>(SL) + MW_ROWS * 64);
    int* const OFF = AT + MW_TV;
    float* const DQ = reinterpret_cast<float*>(smem + ((mw_lds_bytes<128>() + 15) & ~15));
    char* const STG = reinterpret_cast<char*>(DQ) + MB_DQ;    // per wave: 16 rows x 128 bytes, slot j of row p at j ^ (p & 7)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int cr = lane >> 3, cp = lane & 7;              // the "row" layout of a 128-byte chunk: row cr of 8, 16-byte piece cp
    const float a_inv = reinterpret_cast<const float*>(ws)[1];
    const char* const wsA = ws + 64;
    const int nphase = NKC * K;
    char* const st = STG + wv * 2048;
    for (int i = tid; i < 4 * F; i += 512) qs_s[i] = i < K * F ? gt.q[i] * MW_LOG2E : 0.f;
    for (int i = lane; i < 4 * F; i += 64) DQ[wv * 4 * F + i] = 0.f;
    if (tid < 16) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(HB + MW_TV * 128 + 16 * (tid & 7) + (tid >> 3) * MW_HB) = z;
    }
    auto copy_h = [&](int a0, int n, int kc, int buf, int rank, int share) {
        for (int row8 = rank; row8 < 32; row8 += share) {
            const int row = 8 * row8 + (lane >> 3), c = lane & 7;
            const int rr = row < n ? row : n - 1;
            const char* src = reinterpret_cast<const char*>(h + (int64_t)(a0 + rr) * F + 32 * kc) + ((c ^ (row & 7)) << 4);
            mw_copy(src, lds_addr(HB + buf * MW_HB + row8 * 1024));
        }
    };
    auto copy_a = [&](int phase, int buf, int rank, int share) {
        const char* src = wsA + (int64_t)phase * ABUF + lane * 16;
        for (int blk = rank; blk < F / 8; blk += share) mw_copy(src + blk * 1024, lds_addr(AB + buf * ABUF + blk * 1024));
    };
    // S in the accumulator's layout: lane (atom r, half) takes the 16-byte pieces 2 q + half of a source row (columns 8 q + 4 half ..)
    auto add_row = [&](const char* hb, int w, f32x4 (&s)[4]) {
        const char* row = hb + w * 128;
        const int sw = w & 7;
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += *reinterpret_cast<const f32x4*>(row + (((2 * q + hi) ^ sw) << 4));
    };
    auto gather = [&](const char* hb, int base, int cnt, f32x4 (&s)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int16_t* sl = SL + base * 32 + r;
        int w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ww = sl[32 * (q < cnt ? q : 0)];
            w[q] = q < cnt ? ww : MW_TV;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) add_row(hb, w[q], s);
        if (cnt > 2) {
#pragma unroll
            for (int q = 2; q < 4; ++q) add_row(hb, w[q], s);
            if (cnt > 4) {
                int wq = sl[32 * 4];
                for (int q = 4; q < cnt; ++q) {
                    const int wn = sl[32 * (q + 1 < cnt ? q + 1 : q)];
                    add_row(hb, wq, s);
                    wq = wn;
                }
            }
        }
    };
    // lanes of ONE wave exchanging data through LDS: the hardware runs a wave's LDS instructions in order, but the compiler
    // sees independent threads and may let the lanes that skip a divergent store run ahead to the load behind it
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // a 32-row x 128-byte chunk between the row layout (lane: row 8 j + cr, piece cp; a wave instruction covers eight whole
    // 128-byte lines) and the accumulator's layout (lane (atom r, half): pieces 2 q + half), 16 rows at a time through `st`.
    // As direct 16-byte accesses in the accumulator's layout every instruction touched 32 lines for 32 bytes each: the
    // kernel's six row streams made ~3,000 line requests per wave and tile, four times what the bytes need
    auto rows_to_lanes = [&](f32x4 (&v)[4]) {              // in place
        const f32x4 v2 = v[2], v3 = v[3];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            *reinterpret_cast<f32x4*>(st + cr * 128 + ((cp ^ cr) << 4)) = half ? v2 : v[0];
            *reinterpret_cast<f32x4*>(st + (8 + cr) * 128 + ((cp ^ cr) << 4)) = half ? v3 : v[1];
            wave_sync();
            if ((r >> 4) == half) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = *reinterpret_cast<const f32x4*>(st + (r & 15) * 128 + (((2 * q + hi) ^ (r & 7)) << 4));
            }
            wave_sync();
        }
    };
    auto lanes_to_rows = [&](const f32x4 (&v)[4], f32x4 (&o)[4]) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((r >> 4) == half) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(st + (r & 15) * 128 + (((2 * q + hi) ^ (r & 7)) << 4)) = v[q];
            }
            wave_sync();
#pragma unroll
            for (int j = 0; j < 2; ++j)
                o[2 * half + j] = *reinterpret_cast<const f32x4*>(st + (8 * j + cr) * 128 + ((cp ^ cr) << 4));
            wave_sync();
        }
    };

    // this lane's dout row (columns 16 s + 8 hi ..), split once per tile.  (Requested a tile ahead -- during the second-term
    // pass of the tile before -- the 64 registers in flight pushed loop invariants to scratch and bought nothing: the pass is
    // bound by the bytes it moves.)
    f32x4 x[16];
    auto load_dout = [&](int atom) {
        const float* dp = dout + (int64_t)(atom < 0 ? 0 : atom) * F + 8 * hi;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x[2 * s] = *reinterpret_cast<const f32x4*>(dp + 16 * s);
            x[2 * s + 1] = *reinterpret_cast<const f32x4*>(dp + 16 * s + 4);
        }
    };
    int t = blockIdx.x;
    int a0 = 0, n = 1;
    if (t < num_tiles) {
        a0 = tile_rec[4 * t];
        n = tile_rec[4 * t + 1];
        copy_h(a0, n, 0, 0, wv, 8);
        copy_a(0, 0, wv, 8);
    }
    for (; t < num_tiles; t += gridDim.x) {
        const int row0 = tile_rec[4 * t + 2], nrows = tile_rec[4 * t + 3];
        const int tn = t + (int)gridDim.x;
        const bool more = tn < num_tiles;
        const int a0n = more ? tile_rec[4 * tn] : a0, nn = more ? tile_rec[4 * tn + 1] : n;
        mw_barrier();
        {
            const int4* sp = reinterpret_cast<const int4*>(slots + (int64_t)row0 * 32);
            const int nq = nrows * 4;
            int4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            int at_v = 0, off_v = 0;
            if (tid < nq) q0 = sp[tid];
            if (tid + 512 < nq) q1 = sp[tid + 512];
            if (tid < MW_TV) at_v = tile_atom[(int64_t)t * MW_TV + tid];
            if (tid < MW_NB * K + 1) off_v = blk_off[(int64_t)t * (MW_NB * K + 1) + tid];
            __builtin_amdgcn_sched_barrier(0);
            if (tid < nq) reinterpret_cast<int4*>(SL)[tid] = q0;
            if (tid + 512 < nq) reinterpret_cast<int4*>(SL)[tid + 512] = q1;
            if (tid < MW_TV) AT[tid] = at_v;
            if (tid < MW_NB * K + 1) OFF[tid] = off_v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mw_barrier();
        int amask = 0;
        for (int k = 0; k < K; ++k)
            amask |= (__builtin_amdgcn_readfirstlane(OFF[wv * K + k + 1]) > __builtin_amdgcn_readfirstlane(OFF[wv * K + k])) << k;
        const unsigned long long pairs = __builtin_amdgcn_ballot_w64(lane < MW_NB * K && OFF[lane < MW_NB * K ? lane + 1 : 1] >
                                                                                       OFF[lane < MW_NB * K ? lane : 0]);
        // ---- this lane's atom: softmax statistics, dout row (split once, exact row scale)
        MB_T(t0);
        const int at = AT[32 * wv + r];
        const bool live = at >= 0;
        load_dout(at);
        const float2* sp2 = gt.stats + ((int64_t)t * MW_TV + 32 * wv + r) * K;
        float2 st0 = sp2[0], st1 = sp2[K > 1 ? 1 : 0], st2 = sp2[K > 2 ? 2 : 0], st3 = sp2[K > 3 ? 3 : 0];
        if (!live) st0 = st1 = st2 = st3 = float2{3.0e38f, 0.f};   // (positions past the tile's atoms were never written): gate 0
        if (live && hi == 0) {
            float2* so = stats_atom + (int64_t)at * K;
            so[0] = st0;
            if (K > 1) so[1] = st1;
            if (K > 2) so[2] = st2;
            if (K > 3) so[3] = st3;
        }
        h16x8 dh[8], dl[8];
        float tsc = 0.f;                                   // un-scale of an accumulator entry: 1 / (matrix scale * row scale)
        if (amask != 0) {
            float mx = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(x[s][j]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            int e = (__float_as_int(mx) >> 23) & 0xff;
            e = e < 20 ? 20 : (e > 250 ? 250 : e);
            const float dsc = live ? __int_as_float((267 - e) << 23) : 0.f;     // largest |dout| of the row lands in [2^13, 2^14)
            tsc = __int_as_float((e - 13) << 23) * a_inv;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = x[2 * s][j] * dsc, b = x[2 * s + 1][j] * dsc;
                    dh[s][j] = (_Float16)a;
                    dl[s][j] = (_Float16)(a - (float)dh[s][j]);
                    dh[s][4 + j] = (_Float16)b;
                    dl[s][4 + j] = (_Float16)(b - (float)dh[s][4 + j]);
                }
        }
        f32x4 zc[4];                                      // the chunk's logits: row layout until `zlanes`, then the accumulator's
        bool zlanes = false;
        f32x4 P[4];                                       // first term of dz, current chunk
#pragma unroll
        for (int q = 0; q < 4; ++q) P[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        float D0 = 0.f, D1 = 0.f, D2 = 0.f, D3 = 0.f;     // this lane's part of D_ik (its 64 of the atom's 128 columns)
        MB_T(t1);
        MB_ADD(0, t1 - t0);
        MB_ADD(15, 1);

        for (int ph = 0; ph < nphase; ++ph) {
            const int kc = ph / K, k = ph - kc * K;
            MB_T(p0);
            mw_barrier();
            MB_T(p1);
            int idle = 0;
            for (int b = 0; b < MW_NB; ++b) idle |= (int)(((pairs >> (b * K + k)) & 1ull) ^ 1ull) << b;
            const bool i_copy = idle == 0 || ((idle >> wv) & 1);
            const int share = idle == 0 ? MW_NB : __builtin_popcount(idle);
            const int rank = idle == 0 ? wv : __builtin_popcount(idle & ((1 << wv) - 1));
            if (i_copy) {
                if (ph + 1 < nphase) copy_a(ph + 1, (ph + 1) & 1, rank, share);
                else if (more) copy_a(0, 0, rank, share);
                if (k == 0) {
                    if (kc + 1 < NKC) copy_h(a0, n, kc + 1, (kc + 1) & 1, rank, share);
                    else if (more) copy_h(a0n, nn, 0, 0, rank, share);
                }
            }
            if (k == 0 && amask != 0) {                    // the chunk's logits, row layout; first used behind a phase's products
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ra = AT[32 * wv + 8 * j + cr];   // the atom of row 8 j + cr of the block
                    zc[j] = *reinterpret_cast<const f32x4*>(gt.z_atom + (int64_t)(ra < 0 ? 0 : ra) * F + 32 * kc + 4 * cp);
                }
                zlanes = false;
            }
            MB_T(p2);
            MB_ADD(1, p1 - p0);
            MB_ADD(2, p2 - p1);
            if ((amask >> k) & 1) {
                const char* hb = HB + (kc & 1) * MW_HB;
                const char* ab = AB + (ph & 1) * ABUF + r * (2 * F);
                // T^T chunk: rows = the chunk's 32 nf columns, columns = the block's atoms; contraction over mf in 8 steps
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const char* p = ab + (((2 * s + hi) ^ (r & 15)) << 4);
                    const h16x8 wh = *reinterpret_cast<const h16x8*>(p), wl = *reinterpret_cast<const h16x8*>(p + F * 64);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, dh[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, dl[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, dh[s], acc, 0, 0, 0);
                }
                MB_T(p3);
                if (!zlanes) {
                    rows_to_lanes(zc);
                    zlanes = true;
                }
                f32x4 s[4];
                const int b0 = __builtin_amdgcn_readfirstlane(OFF[wv * K + k]);
                gather(hb, b0, __builtin_amdgcn_readfirstlane(OFF[wv * K + k + 1]) - b0, s);
                MB_T(p4);
                const float m = k == 0 ? st0.x : k == 1 ? st1.x : k == 2 ? st2.x : st3.x;
                const float iv = (k == 0 ? st0.y : k == 1 ? st1.y : k == 2 ? st2.y : st3.y) * tsc;
                const float* qp = qs_s + k * F + 32 * kc + 4 * hi;
                float dsum = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xg = __builtin_amdgcn_exp2f(fmaf(zc[q][j], MW_LOG2E, qv[j] - m)) * iv * s[q][j];   // X * un-scale
                        const float u = xg * acc[4 * q + j];
                        dsum += u;
                        P[q][j] += u;
                    }
                }
                D0 += k == 0 ? dsum : 0.f;
                D1 += k == 1 ? dsum : 0.f;
                D2 += k == 2 ? dsum : 0.f;
                D3 += k == 3 ? dsum : 0.f;
                MB_T(p5);
                MB_ADD(3, p3 - p2);
                MB_ADD(4, p4 - p3);
                MB_ADD(5, p5 - p4);
                MB_ADD(14, 1);
            }
            MB_T(p6);
            if (k == K - 1) {                              // the chunk is done: first term of its dz columns, whole lines per store
                f32x4 pr[4];
                if (amask != 0) {
                    lanes_to_rows(P, pr);
#pragma unroll
                    for (int q = 0; q < 4; ++q) P[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) pr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ra = AT[32 * wv + 8 * j + cr];
                    if (ra >= 0) *reinterpret_cast<f32x4*>(dz + (int64_t)ra * F + 32 * kc + 4 * cp) = pr[j];
                }
            }
            MB_T(p7);
            if (i_copy) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MB_T(p8);
            MB_ADD(6, p7 - p6);
            MB_ADD(7, p8 - p7);
        }
        // ---- second term: dz -= sum_k g_k D_k and dq_k -= sum_atoms g_k D_k, in a whole-row layout: a wave instruction
        // covers two atoms' rows (lane: atom 2 it + half, columns 4 r ..), so every access is whole lines and the sum over
        // atoms runs down a lane's registers.  Per-atom scalars (max, D / sum) travel through LDS.
        MB_T(f0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my first-term stores are done: other lanes of the wave read them back
        MB_T(f0a);
        mw_barrier();                                      // every wave is done with the last chunk's h image (buffer 1): its
                                                           // first 32 KB are free until the next tile's first phase
        MB_T(f0b);
        MB_T(f0c);
        MB_ADD(9, f0a - f0);
        MB_ADD(10, f0b - f0a);
        MB_ADD(11, f0c - f0b);
        if (amask != 0) {
            float* const SC = reinterpret_cast<float*>(HB + MW_HB + wv * 4096);    // [32 atoms][max x 4 | D / sum x 4]
            D0 += __shfl_xor(D0, 32);
            D1 += __shfl_xor(D1, 32);
            D2 += __shfl_xor(D2, 32);
            D3 += __shfl_xor(D3, 32);
            if (hi == 0) {
                *reinterpret_cast<f32x4*>(SC + 8 * r) = f32x4{st0.x, st1.x, st2.x, st3.x};
                *reinterpret_cast<f32x4*>(SC + 8 * r + 4) =
                    live ? f32x4{st0.y * D0, st1.y * D1, st2.y * D2, st3.y * D3} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wave_sync();
            f32x4 ql[4], dqa[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ql[k] = *reinterpret_cast<const f32x4*>(qs_s + k * F + 4 * r);
                dqa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int a = 2 * it + hi;                 // the block's atom of this lane's row
                const int atom = AT[32 * wv + a];
                const int64_t off = (int64_t)(atom < 0 ? 0 : atom) * F + 4 * r;
                const f32x4 zz = *reinterpret_cast<const f32x4*>(gt.z_atom + off);
                f32x4 pp = *reinterpret_cast<const f32x4*>(dz + off);
                const f32x4 mm = *reinterpret_cast<const f32x4*>(SC + 8 * a), gd = *reinterpret_cast<const f32x4*>(SC + 8 * a + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!((amask >> k) & 1)) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = __builtin_amdgcn_exp2f(fmaf(zz[j], MW_LOG2E, ql[k][j] - mm[k])) * gd[k];
                        pp[j] -= v;
                        dqa[k][j] += v;
                    }
                }
                if (atom >= 0) *reinterpret_cast<f32x4*>(dz + off) = pp;
            }
            MB_T(f0d);
            MB_ADD(12, f0d - f0c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((amask >> k) & 1)) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) dqa[k][j] += __shfl_xor(dqa[k][j], 32);
                if (hi == 0) {
                    f32x4* d = reinterpret_cast<f32x4*>(DQ + (wv * 4 + k) * F + 4 * r);
                    *d = *d + dqa[k];
                }
            }
        }
        MB_T(f1);
        MB_ADD(8, f1 - f0);
        a0 = a0n;
        n = nn;
    }
    // ---- this wave's dq sums (second term; the caller subtracts them from the column sums of A * dA)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = lane; i < 4 * F; i += 64)
        if (i < K * F) dq_part[((int64_t)blockIdx.x * MW_NB + wv) * K * F + i] = DQ[wv * 4 * F + i];
}

// ------------------------------------------------------------------------------------------ width 64: resident matrices
// At nf = mf = 64 the K <= 4 matrices fit in LDS as fp16 piece pairs (16 KB per type) next to a whole 256-atom tile of h
// rows (64 KB as fp32), so nothing is streamed in phases: after the tile's rows are parked (they were fetched into
// registers during the previous tile, with the tile's slot rows and atom list) every wave walks the bond types of its
// 32-atom block at its own pace -- sum the neighbours' rows (all 64 columns), guard, split, 24 MFMAs -- and the only
// block barriers are the two around the parking.  Same plan (graph.py::WidePlan), same math as the kernel above.
constexpr int M64_K = 4;
constexpr int M64_AIMG = 8192;                  // one (type, piece) image: 64 output columns x 128 bytes
constexpr int M64_HT = (MW_TV + 1) * 256;
__host__ __device__ constexpr int m64_lds_bytes() {
    return M64_K * 2 * M64_AIMG + M64_HT + MW_ROWS * 64 + MW_TV * 4 + (MW_NB * M64_K + 1) * 4 + 64;
}

// workspace: scale | inverse, then [type][piece][column n][64 halves], the 16-byte slots of a column XOR-ed by (n >> 1) & 7
__global__ void __launch_bounds__(256) m64_split_kernel(const float* __restrict__ A, char* __restrict__ ws, int K) {
    const int idx = blockIdx.x * 256 + threadIdx.x;       // (k, n, octet o)
    if (idx >= K * 64 * 8) return;
    const int k = idx >> 9, n = (idx >> 3) & 63, o = idx & 7;
    const float sc = reinterpret_cast<const float*>(ws)[0];
    const float* p = A + ((int64_t)k * 64 + n) * 64 + 8 * o;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4);
    h16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * sc, b = x1[j] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
        ph[4 + j] = (_Float16)b;
        pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
    }
    char* dst = ws + 64 + (int64_t)(2 * k) * M64_AIMG + n * 128 + ((o ^ ((n >> 1) & 7)) << 4);
    *reinterpret_cast<h16x8*>(dst) = ph;
    *reinterpret_cast<h16x8*>(dst + M64_AIMG) = pl;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) message_sum_res64_kernel(
    const float* __restrict__ h, const char* __restrict__ ws, const int32_t* __restrict__ tile_rec,
    const int32_t* __restrict__ tile_atom, const int32_t* __restrict__ blk_off, const int16_t* __restrict__ slots,
    float* __restrict__ out, int num_tiles, int K) {
    constexpr int F = 64, CT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const AR = smem;                               // resident matrices
    char* const HT = smem + M64_K * 2 * M64_AIMG;        // the tile's h rows (fp32, 256 bytes each) + the zero row
    int16_t* const SL = reinterpret_cast<int16_t*>(HT + M64_HT);
    int* const AT = reinterpret_cast<int*>(reinterpret_cast<char*>(SL) + MW_ROWS * 64);
    int* const OFF = AT + MW_TV;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const float a_inv = reinterpret_cast<const float*>(ws)[1];

    // matrices: K x 16 KB, copied verbatim (2 K wave-instructions per wave); the zero row of the h image
    for (int i = wv; i < K * 16; i += 8) mw_copy(ws + 64 + i * 1024 + lane * 16, lds_addr(AR + i * 1024));
    if (tid < 16) *reinterpret_cast<f32x4*>(HT + MW_TV * 256 + 16 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers of the NEXT tile: thread = (row prow + 32 j, 16-byte slot) of the h rows, two slot-row quads,
    // one atom id, one offset word
    // The loads are inline assembly on purpose.  A tile's 32 out stores are issued AFTER the next tile's loads and BEFORE
    // they are used; vmcnt retires in order, so the right wait is "all but the youngest stores" -- but across the loop's
    // back edge the compiler waits vmcnt(<= 9) for its own loads, i.e. until the tile's stores have been acknowledged as
    // well (measured: tile time = stream time + compute time instead of their maximum).  Hidden from its bookkeeping,
    // the loads are waited for by hand in park(): vmcnt(16) -- at least 16 of the 32 stores that follow them stay in flight.
    const int prow = tid >> 4, pslot = tid & 15;
    f32x4 pf[8];
    int4 psl0 = {0, 0, 0, 0}, psl1 = psl0;                // (two named values: as an array they lived in scratch)
    int pat = -1, poff = 0;
    auto fetch = [&](int t) {
        const int a0 = tile_rec[4 * t], n = tile_rec[4 * t + 1], row0 = tile_rec[4 * t + 2], nrows = tile_rec[4 * t + 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = prow + 32 * j;
            const int rr = row < n ? row : n - 1;
            pf[j] = *reinterpret_cast<const f32x4*>(h + (int64_t)(a0 + rr) * F + 4 * pslot);
        }
        const int4* sp = reinterpret_cast<const int4*>(slots + (int64_t)row0 * 32);
        psl0 = sp[tid < nrows * 4 ? tid : 0];
        psl1 = sp[tid + 512 < nrows * 4 ? tid + 512 : 0];
        if (tid < MW_TV) pat = tile_atom[(int64_t)t * MW_TV + tid];
        if (tid < MW_NB * K + 1) poff = blk_off[(int64_t)t * (MW_NB * K + 1) + tid];
    };
    auto park = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = prow + 32 * j;
            *reinterpret_cast<f32x4*>(HT + row * 256 + ((pslot ^ (row & 15)) << 4)) = pf[j];
        }
        reinterpret_cast<int4*>(SL)[tid] = psl0;
        reinterpret_cast<int4*>(SL)[tid + 512] = psl1;
        if (tid < MW_TV) AT[tid] = pat;
        if (tid < MW_NB * K + 1) OFF[tid] = poff;
    };

    int t = blockIdx.x;
    if (t < num_tiles) fetch(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the matrices' copies)
    for (; t < num_tiles; t += gridDim.x) {
        mw_barrier();                                     // every wave is done with the previous tile's LDS data
        park();
        mw_barrier();
        if (t + (int)gridDim.x < num_tiles) fetch(t + (int)gridDim.x);    // in flight during this tile's work

        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        float row_sc = 0.f, row_inv = 0.f;                // 0 = this atom has not seen a non-zero sum yet
        for (int k = 0; k < K; ++k) {
            const int base = __builtin_amdgcn_readfirstlane(OFF[wv * K + k]);
            const int cnt = __builtin_amdgcn_readfirstlane(OFF[wv * K + k + 1]) - base;
            if (cnt <= 0) continue;
            // ---- S: per atom (lane r; half hi holds columns 16 st + 8 hi ...) the sum of its type-k neighbours' rows
            f32x4 s[4][2];
#pragma unroll
            for (int st = 0; st < 4; ++st) s[st][0] = s[st][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto add_row = [&](int w) {
                const char* row = HT + w * 256;
                const int sw = w & 15;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int c0 = 4 * st + 2 * hi;
                    s[st][0] += *reinterpret_cast<const f32x4*>(row + ((c0 ^ sw) << 4));
                    s[st][1] += *reinterpret_cast<const f32x4*>(row + (((c0 + 1) ^ sw) << 4));
                }
            };
            {
                const int16_t* sl = SL + base * 32 + r;
                int w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ww = sl[32 * (q < cnt ? q : 0)];
                    w[q] = q < cnt ? ww : MW_TV;
                }
                add_row(w[0]);
                add_row(w[1]);
                if (cnt > 2) {
                    add_row(w[2]);
                    add_row(w[3]);
                    for (int q = 4; q < cnt; ++q) add_row(sl[32 * q]);
                }
            }
            // ---- range guard: one power-of-two scale per atom, lowered when a later type's sums outgrow it
            float mx = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fabsf(s[st][0][u]), fabsf(s[st][1][u])));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const bool unset = row_sc == 0.f;
            const bool grow = unset ? mx > 0.f : mx * row_sc >= 32768.0f;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                int e = (__float_as_int(mx) >> 23) & 0xff;
                e = e < 40 ? 40 : (e > 240 ? 240 : e);
                const float ns = grow ? __int_as_float((267 - e) << 23) : row_sc;    // mx * ns in [2^13, 2^14)
                const float ni = grow ? __int_as_float((e - 13) << 23) : row_inv;
                if (__builtin_amdgcn_ballot_w64(grow && !unset) != 0) {
                    const float ratio = unset ? 1.0f : ns * row_inv;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int dr = 8 * (i >> 2) + (i & 3);
                        const float f_lo = readlane_f(ratio, dr), f_hi = readlane_f(ratio, 4 + dr);
                        const float f = hi ? f_hi : f_lo;
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[c][i] *= f;
                    }
                }
                row_sc = ns;
                row_inv = ni;
            }
            // ---- split and contract: 4 K = 16 steps x 2 column tiles x 3 MFMAs
            const char* ak = AR + (2 * k) * M64_AIMG;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                h16x8 ah, al;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = s[st][0][u] * row_sc, b = s[st][1][u] * row_sc;
                    ah[u] = (_Float16)a;
                    al[u] = (_Float16)(a - (float)ah[u]);
                    ah[4 + u] = (_Float16)b;
                    al[4 + u] = (_Float16)(b - (float)ah[4 + u]);
                }
                const int o = 2 * st + hi;
                const int n0 = r, n1 = 32 + r;
                const char* p0 = ak + n0 * 128 + ((o ^ ((n0 >> 1) & 7)) << 4);
                const char* p1 = ak + n1 * 128 + ((o ^ ((n1 >> 1) & 7)) << 4);
                const h16x8 b0h = *reinterpret_cast<const h16x8*>(p0), b0l = *reinterpret_cast<const h16x8*>(p0 + M64_AIMG);
                const h16x8 b1h = *reinterpret_cast<const h16x8*>(p1), b1l = *reinterpret_cast<const h16x8*>(p1 + M64_AIMG);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc[1], 0, 0, 0);
            }
        }
        // ---- out rows
        int atom[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) atom[i] = AT[32 * wv + 8 * (i >> 2) + (i & 3) + 4 * hi];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int dr = 8 * (i >> 2) + (i & 3);
            const float u_lo = readlane_f(row_inv, dr), u_hi = readlane_f(row_inv, 4 + dr);
            const float un = (hi ? u_hi : u_lo) * a_inv;
            if (atom[i] >= 0) {
                float* o = out + (int64_t)atom[i] * F + r;
#pragma unroll
                for (int c = 0; c < CT; ++c) __builtin_nontemporal_store(acc[c][i] * un, o + 32 * c);
            }
        }
    }
}

static int launch_message_res64(const float* h, const float* A, const int32_t* tile_rec, const int32_t* tile_atom,
                                const int32_t* blk_off, const int16_t* slots, float* out, void* workspace,
                                int64_t num_tiles, int K, hipStream_t s) {
    static const hipError_t attr = [] {
        LdsOptIn opt_in_;
        opt_in_((const void*)message_sum_res64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, m64_lds_bytes());
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    char* ws = (char*)workspace;
    hipLaunchKernelGGL(mw_absmax_kernel, dim3(1), dim3(1024), 0, s, A, (int64_t)K * 64 * 64, (float*)ws);
    hipLaunchKernelGGL(m64_split_kernel, dim3((unsigned)((K * 512 + 255) / 256)), dim3(256), 0, s, A, ws, K);
    int64_t blocks = 256;
    if (blocks > num_tiles) blocks = num_tiles;
    hipLaunchKernelGGL(message_sum_res64_kernel, dim3((unsigned)blocks), dim3(512), m64_lds_bytes(), s, h, ws, tile_rec,
                       tile_atom, blk_off, slots, out, (int)num_tiles, K);
    return launch_status("mpnn_message_aggregate_wide_f32(64)");
}

size_t message_wide_workspace_bytes(int K, int F) { return 64 + (size_t)K * F * F * 4; }
size_t message_wide_gated_workspace_bytes(int K, int F, int64_t num_tiles) {
    return message_wide_workspace_bytes(K, F) + (size_t)num_tiles * MW_TV * K * sizeof(float2);
}

template <int F>
static int launch_message_wide_t(const float* h, const float* A, const int32_t* tile_rec, const int32_t* tile_atom,
                                 const int32_t* blk_off, const int16_t* slots, float* out, void* workspace,
                                 int64_t num_tiles, int K, hipStream_t s, const float* z_atom = nullptr,
                                 const float* q = nullptr) {
    static const hipError_t attr = [] {
        LdsOptIn opt_in_;
        opt_in_((const void*)message_sum_wide_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, mw_lds_bytes<F>());
        if (F == 128)
            opt_in_((const void*)message_sum_wide_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mw_lds_bytes<128>());
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    char* ws = (char*)workspace;
    hipLaunchKernelGGL(mw_absmax_kernel, dim3(1), dim3(1024), 0, s, A, (int64_t)K * F * F, (float*)ws);
    const int64_t units = (int64_t)K * F * (F / 8);
    hipLaunchKernelGGL(mw_split_kernel<F>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, A, ws, K);
    int64_t blocks = 256;                                 // one block per CU
    if (blocks > num_tiles) blocks = num_tiles;
    const int dbg = 0;                                    // (bits 2 / 4 / 8 switch off the products / the copies / all work: timing ablations)
    if (z_atom) {
        if constexpr (F == 128) {
            float2* stats = reinterpret_cast<float2*>(ws + message_wide_workspace_bytes(K, F));
            hipLaunchKernelGGL(mw_gate_stats_kernel, dim3((unsigned)(num_tiles < 2048 ? num_tiles : 2048)), dim3(512), 0, s,
                               z_atom, q, tile_rec, tile_atom, stats, (int)num_tiles, K);
            hipLaunchKernelGGL((message_sum_wide_kernel<128, true>), dim3((unsigned)blocks), dim3(512), mw_lds_bytes<128>(), s, h,
                               ws, tile_rec, tile_atom, blk_off, slots, out, (int)num_tiles, K, dbg, MwGate{z_atom, q, stats});
            return launch_status("mpnn_message_aggregate_wide_gated_f32");
        }
        return 1;
    }
    hipLaunchKernelGGL(message_sum_wide_kernel<F>, dim3((unsigned)blocks), dim3(512), mw_lds_bytes<F>(), s, h, ws, tile_rec,
                       tile_atom, blk_off, slots, out, (int)num_tiles, K, dbg, MwGate{nullptr, nullptr, nullptr});
    return launch_status("mpnn_message_aggregate_wide_f32");
}

constexpr int MB_BLOCKS = 256;                           // one block per CU
static int launch_att_message_bwd(const float* h, const float* A, const float* z_atom, const float* q, const float* dout,
                                  const float2* stats, const int32_t* tile_rec, const int32_t* tile_atom,
                                  const int32_t* blk_off, const int16_t* slots, float* dz, float* dq_part, float2* stats_atom,
                                  void* workspace, int64_t num_tiles, int K, hipStream_t s) {
    static const hipError_t attr = [] {
        LdsOptIn opt_in_;
        opt_in_((const void*)att_message_bwd_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mb_lds_bytes());
        return opt_in_.err;
    }();
    if (attr != hipSuccess) return lds_opt_in_failed(attr);
    char* ws = (char*)workspace;
    hipLaunchKernelGGL(mw_absmax_kernel, dim3(1), dim3(1024), 0, s, A, (int64_t)K * 128 * 128, (float*)ws);
    hipLaunchKernelGGL(mw_split_t_kernel, dim3((unsigned)((K * 2048 + 255) / 256)), dim3(256), 0, s, A, ws, K);
    int64_t blocks = MB_BLOCKS;
    if (blocks > num_tiles) blocks = num_tiles;
    hipLaunchKernelGGL(att_message_bwd_tile_kernel, dim3((unsigned)blocks), dim3(512), mb_lds_bytes(), s, h, dout, ws, tile_rec,
                       tile_atom, blk_off, slots, MwGate{z_atom, q, stats}, dz, dq_part, stats_atom, (int)num_tiles, K);
    return launch_status("mpnn_message_aggregate_wide_gated_bwd_f32");
}

}  // namespace mpnn

using namespace mpnn;

#ifdef MB_STAMP
extern "C" int mpnn_debug_mb_stamps(unsigned long long* host16, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mb_stamps), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_mb_stamps), 16 * sizeof(unsigned long long));
}
#endif
#ifdef MW_STAMP
extern "C" int mpnn_debug_mw_stamps(unsigned long long* host32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mw_stamps), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_mw_stamps), 32 * sizeof(unsigned long long));
}
#endif

extern "C" int mpnn_message_aggregate_wide_tile_atoms(void) { return MW_TV; }
extern "C" int mpnn_message_aggregate_wide_max_types(void) { return MW_KMAX; }
extern "C" int mpnn_message_aggregate_wide_max_rows(void) { return MW_ROWS; }
extern "C" size_t mpnn_message_aggregate_wide_workspace_bytes(int K, int nf) {
    return message_wide_workspace_bytes(K, nf);
}

extern "C" int mpnn_message_aggregate_wide_f32(const float* h, const float* A, const int32_t* tile_rec,
                                               const int32_t* tile_atom, const int32_t* blk_off, const int16_t* slots,
                                               float* out, void* workspace, size_t workspace_bytes, int64_t V,
                                               int64_t num_tiles, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(nf == mf && (nf == 64 || nf == 128 || nf == 256),
                 "mpnn_message_aggregate_wide_f32: nf = mf in {64, 128, 256} only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(nf != 64 || K <= M64_K, "mpnn_message_aggregate_wide_f32: at most %d bond types at width 64 (got %d)", M64_K, K);
    MPNN_REQUIRE(K >= 1 && K <= MW_KMAX, "mpnn_message_aggregate_wide_f32: 1 <= K <= %d bond types (got %d)", MW_KMAX, K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 24), "mpnn_message_aggregate_wide_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && tile_rec && tile_atom && blk_off && slots && out && workspace,
                 "mpnn_message_aggregate_wide_f32: NULL buffer");
    MPNN_REQUIRE(workspace_bytes >= message_wide_workspace_bytes(K, nf),
                 "mpnn_message_aggregate_wide_f32: workspace of %zu bytes, need %zu", workspace_bytes,
                 message_wide_workspace_bytes(K, nf));
    hipStream_t s = (hipStream_t)stream;
    if (nf == 64) return launch_message_res64(h, A, tile_rec, tile_atom, blk_off, slots, out, workspace, num_tiles, K, s);
    if (nf == 128) return launch_message_wide_t<128>(h, A, tile_rec, tile_atom, blk_off, slots, out, workspace, num_tiles, K, s);
    return launch_message_wide_t<256>(h, A, tile_rec, tile_atom, blk_off, slots, out, workspace, num_tiles, K, s);
}

extern "C" size_t mpnn_message_aggregate_wide_gated_workspace_bytes(int K, int nf, int64_t num_tiles) {
    return message_wide_gated_workspace_bytes(K, nf, num_tiles);
}

extern "C" int mpnn_message_aggregate_wide_gated_f32(const float* h, const float* A, const float* z_atom, const float* q,
                                                     const int32_t* tile_rec, const int32_t* tile_atom,
                                                     const int32_t* blk_off, const int16_t* slots, float* out,
                                                     void* workspace, size_t workspace_bytes, int64_t V, int64_t num_tiles,
                                                     int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(nf == mf && nf == 128, "mpnn_message_aggregate_wide_gated_f32: nf = mf = 128 only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(K >= 1 && K <= 4, "mpnn_message_aggregate_wide_gated_f32: 1 <= K <= 4 bond types (got %d)", K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 24), "mpnn_message_aggregate_wide_gated_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && z_atom && q && tile_rec && tile_atom && blk_off && slots && out && workspace,
                 "mpnn_message_aggregate_wide_gated_f32: NULL buffer");
    MPNN_REQUIRE(workspace_bytes >= message_wide_gated_workspace_bytes(K, nf, num_tiles),
                 "mpnn_message_aggregate_wide_gated_f32: workspace of %zu bytes, need %zu", workspace_bytes,
                 message_wide_gated_workspace_bytes(K, nf, num_tiles));
    return launch_message_wide_t<128>(h, A, tile_rec, tile_atom, blk_off, slots, out, workspace, num_tiles, K,
                                      (hipStream_t)stream, z_atom, q);
}

extern "C" size_t mpnn_message_aggregate_wide_gated_bwd_workspace_bytes(int K, int nf) {
    return message_wide_workspace_bytes(K, nf);
}
extern "C" int mpnn_message_aggregate_wide_gated_bwd_parts(void) { return MB_BLOCKS * MW_NB; }

extern "C" int mpnn_message_aggregate_wide_gated_bwd_f32(const float* h, const float* A, const float* z_atom, const float* q,
                                                         const float* dagg, const void* fwd_workspace,
                                                         size_t fwd_workspace_bytes, const int32_t* tile_rec,
                                                         const int32_t* tile_atom, const int32_t* blk_off,
                                                         const int16_t* slots, float* dz_atom, float* dq_part,
                                                         float* stats_by_atom, void* workspace, size_t workspace_bytes,
                                                         int64_t V, int64_t num_tiles, int K, int nf, int mf, void* stream) {
    MPNN_REQUIRE(nf == mf && nf == 128, "mpnn_message_aggregate_wide_gated_bwd_f32: nf = mf = 128 only (got %d, %d)", nf, mf);
    MPNN_REQUIRE(K >= 1 && K <= 4, "mpnn_message_aggregate_wide_gated_bwd_f32: 1 <= K <= 4 bond types (got %d)", K);
    MPNN_REQUIRE(V >= 0 && num_tiles >= 0 && num_tiles < (1ll << 24), "mpnn_message_aggregate_wide_gated_bwd_f32: bad sizes");
    if (V == 0 || num_tiles == 0) return MPNN_OK;
    MPNN_REQUIRE(h && A && z_atom && q && dagg && fwd_workspace && tile_rec && tile_atom && blk_off && slots && dz_atom &&
                     dq_part && stats_by_atom && workspace,
                 "mpnn_message_aggregate_wide_gated_bwd_f32: NULL buffer");
    MPNN_REQUIRE(fwd_workspace_bytes >= message_wide_gated_workspace_bytes(K, nf, num_tiles),
                 "mpnn_message_aggregate_wide_gated_bwd_f32: forward workspace of %zu bytes, need %zu", fwd_workspace_bytes,
                 message_wide_gated_workspace_bytes(K, nf, num_tiles));
    MPNN_REQUIRE(workspace_bytes >= message_wide_workspace_bytes(K, nf),
                 "mpnn_message_aggregate_wide_gated_bwd_f32: workspace of %zu bytes, need %zu", workspace_bytes,
                 message_wide_workspace_bytes(K, nf));
    const float2* stats = reinterpret_cast<const float2*>((const char*)fwd_workspace + message_wide_workspace_bytes(K, nf));
    return launch_att_message_bwd(h, A, z_atom, q, dagg, stats, tile_rec, tile_atom, blk_off, slots, dz_atom, dq_part,
                                  reinterpret_cast<float2*>(stats_by_atom), workspace, num_tiles, K, (hipStream_t)stream);
}
