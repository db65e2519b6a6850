// GRU backward at H = 128 WITHOUT a gate-gradient workspace: the two consumers -- dm | dh and dW -- each form the gate
// gradients they contract in registers, from the slices of (dout, h, r, z, n, gh_n) they read anyway.  Reference:
// mpnn_functions/update/gru_update.py:26-35 (autograd of it).
//
// Why.  The gate gradients (dar daz dan dnh) are COLUMN-LOCAL: entry j of an atom depends on entry j of dout, h and the
// four saved gate arrays only.  gru_bwd128_f16.hip computes them once in a kernel of their own and hands them to the two
// contractions through HBM as fp16 pieces (16 H bytes per atom written, then read H / 128 times by dm | dh and again by
// dW): 96 H bytes per atom for a backward whose operands are 36 H.  Here nothing travels: dm | dh reads 24 H and writes
// 8 H, dW reads 28 H, 60 H in all, and one launch (and the 16 H B / atom workspace) is gone.
//
// dm | dh (gru_rc_dx_kernel) is written TRANSPOSED: out^T[feature][atom] = W[feature][gate column] . G^T[gate column][atom].
//   * The weights are the A operand (pre-split fp16 images copied global -> LDS, as before); the gate gradients are the B
//     operand, whose lane layout -- lane (atom = lane & 31, half = lane >> 5) holds 8 k-values of ITS atom -- is what a
//     lane gets by loading its own atom's row: no exchange between lanes, and the per-atom power-of-two scale of the fp16
//     pieces is a per-LANE scalar.
//   * The k order inside a 16-wide step is chosen as k-slot (half, j) <-> column 16 s + 8 (j >> 2) + 4 half + (j & 3): these
//     are exactly the FEATURES the lane owns in the accumulator of column block cc (row = 8 (i >> 2) + 4 half + (i & 3)), so
//     the direct term dout * mask * z of dh is added to the accumulator in registers, where round 3 wrote it to HBM in one
//     kernel and read it back in the next.
//   * A row's scale cannot wait for the whole row (it would take a pass of its own): it is set from the first 16 columns
//     with three bits of headroom and lowered -- accumulators multiplied by the ratio, a power of two -- when a later step
//     outgrows it.  Headroom costs nothing: a value 2^k below its row's largest still has 22 - max(0, k - 10) bits.
//   * Outputs leave as 16-byte stores (4 consecutive features of one atom); round 3 stored dwords.
// dW (gru_rc_dw_kernel) keeps round 3's structure (32-atom tiles, row-major fp16 images in LDS, columns by transposed
// reads) with the gate pieces made in place of copied: thread (row, 8 columns) computes them while the tile before is
// being contracted, takes the row's exact maximum over its 16 neighbours, and parks the pieces; the bias gradients
// (column sums) ride in the same registers.
// hipcc-flags: -fno-slp-vectorize      (mpnn_amd/build.py: per-file flags.  Adjacent scalar multiplies must stay scalar here:
//                                      packed fp32 instructions are an anti-lever beside MFMAs, MI355X_MICROARCH.md)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace mpnn {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#ifdef RC_STAMP   // diagnostic build only (-DRC_STAMP): cycle sums per part of a step, block 3, consumer 0 and producer 4 (dm | dh), waves 0 and 7 (dW)
__device__ unsigned long long g_rc_stamps[64];
#define RC_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define RC_ADD_STAMP(slot, val) atomicAdd(g_rc_stamps + (slot), (unsigned long long)(val))
#else
#define RC_T(var)
#endif

namespace {
constexpr int R_IMG = 32 * 256;                        // bytes of one [32 rows][128 x fp16] image (dW kernel)
constexpr int R_SUB = 2 * 2 * 128 * 16;                // dm | dh: one (matrix, gate) weight sub-image of a 16-wide step: [piece][half][128 rows][8 x fp16]
constexpr int R_STEP = 6 * R_SUB;                      // 48 KB: (r: ih hh) (z: ih hh) (n: ih) (gh_n: hh)

// power of two s with maxabs * s in [2^(14 - HEAD), 2^(15 - HEAD)), as a biased exponent pair: s = 2^(141 - HEAD - e)
__device__ __forceinline__ int r_exp(float maxabs) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    return e < 51 ? 51 : (e > 187 ? 187 : e);
}
__device__ __forceinline__ float r_pow2(int field) { return __int_as_float(field << 23); }

// maximum over the 16 lanes of a DPP row (every lane gets it): two quad permutes, then rotations by 4 and 8 within the row
__device__ __forceinline__ float r_row16_max(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true);         // row_ror:4
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true);         // row_ror:8
    return fmaxf(v, __int_as_float(x));
}

// maximum over the wave (every lane gets it): the four rows' maxima by readlane
__device__ __forceinline__ float r_wave_max(float v) {
    v = r_row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

__device__ __forceinline__ void r_split8(const f32x4& x0, const f32x4& x1, float sc, h16x8& ph, h16x8& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * sc, b = x1[j] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
        ph[4 + j] = (_Float16)b;
        pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
    }
}

typedef _Float16 r_h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void r_split4(const f32x4& x, float sc, r_h16x4& ph, r_h16x4& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = x[j] * sc;
        ph[j] = (_Float16)v;
        pl[j] = (_Float16)(v - (float)ph[j]);
    }
}

__device__ __forceinline__ void r_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one wave copies 1 KB global -> LDS without registers (gru_bwd128_f16.hip has the reasons for the inline assembly)
__device__ __forceinline__ void r_copy_to_lds(const char* src, const char* lds_dst) {
    typedef __attribute__((address_space(3))) const char lds_char;
    const unsigned dst = (unsigned)(uintptr_t)(lds_char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}

// the same with a wave-uniform base in scalar registers and a 32-bit lane offset: no 64-bit address per copy and lane to keep
__device__ __forceinline__ void r_copy_to_lds_s(const char* sbase, unsigned voff, const char* lds_dst) {
    typedef __attribute__((address_space(3))) const char lds_char;
    const unsigned dst = (unsigned)(uintptr_t)(lds_char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(dst)
                 : "memory");
}

__device__ __forceinline__ h16x8 r_tr8(const char* a0, const char* a1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(h16x8, v);
}

// the gate gradients of four columns of one atom (gru_update.py:29-34 differentiated).  NORM: dout is the gradient of
// norm(y), y = this update's output; dy = dout k1 + y k2 + k4 on rows with mask 1 (include/mpnn_amd.h).
template <bool NORM>
__device__ __forceinline__ void r_gate_grads4(const f32x4& dout, const f32x4& hv, const f32x4& r, const f32x4& z,
                                              const f32x4& n, const f32x4& nh, float mk, const f32x4& k1, const f32x4& k2,
                                              const f32x4& k4, f32x4& dar, f32x4& daz, f32x4& dan, f32x4& dnh, f32x4& gz) {
    // (element by element on purpose: written on 4-vectors these become v_pk_* instructions, which cost the vector pipe
    // three times a plain one while the SIMD's other wave issues MFMAs -- 12 cycles per instruction measured in the producers)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float g = dout[t] * mk;
        if (NORM) {
            const float y = ((1.0f - z[t]) * n[t] + z[t] * hv[t]) * mk;
            g = (dout[t] * k1[t] + y * k2[t] + k4[t]) * mk;
        }
        const float omz = 1.0f - z[t];
        const float gm = g * mk;
        const float a_n = gm * omz * (1.0f - n[t] * n[t]);   // n = tanh(.) * mask
        dan[t] = a_n;
        dar[t] = a_n * nh[t] * mk * r[t] * (1.0f - r[t]);
        daz[t] = gm * (hv[t] - n[t]) * z[t] * omz;
        dnh[t] = a_n * r[t];
        gz[t] = g * z[t];
    }
}

// maximum over the 4 lanes of a quad, by DPP (no trip through the LDS crossbar, which the consumers keep busy)
__device__ __forceinline__ float r_quad_max(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    return fmaxf(v, __int_as_float(x));
}

__device__ __forceinline__ float r_max4(float m, const f32x4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
}  // namespace

// ------------------------------------------------------------------------------------------ dm | dh: weight images
// grid = NS slices x (H / 16) steps.  Image of (slice, step): six sub-images (r: W_ih W_hh, z: W_ih W_hh, n: W_ih,
// gh_n: W_hh), each [piece hi | lo][half][128 output features of the slice][8 x fp16]: an A fragment of 32 features is 32
// consecutive 16-byte entries (conflict-free ds_read_b128).  Entry (half, feature) holds the k-slots j = 0..7 of that half:
// gate column 16 step + 8 (j >> 2) + 4 half + (j & 3).  One power-of-two scale per slice (float `slice` of the region).
template <int H>
__global__ void __launch_bounds__(512) gru_rc_presplit_kernel(const float* __restrict__ W_ih, const float* __restrict__ W_hh,
                                                              char* __restrict__ ws) {
    constexpr int NSTEP = H / 16;
    __shared__ float redw[8];
    const int slice = blockIdx.x / NSTEP, step = blockIdx.x % NSTEP;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = 0.f;
    for (int idx = tid; idx < 2 * 128 * (3 * H / 4); idx += 512) {
        const int mat = idx / (128 * (3 * H / 4)), rem = idx % (128 * (3 * H / 4));
        const f32x4 w4 = *reinterpret_cast<const f32x4*>((mat ? W_hh : W_ih) + (int64_t)(128 * slice) * 3 * H + 4 * rem);
        mx = r_max4(mx, w4);
    }
    mx = r_wave_max(mx);
    if (lane == 0) redw[wv] = mx;
    __syncthreads();
    mx = redw[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) mx = fmaxf(mx, redw[u]);
    int e = (__float_as_int(mx) >> 23) & 0xff;
    e = e < 111 ? 111 : (e > 187 ? 187 : e);               // scale in [2^-46, 2^30] (as g_guard_scale<30>)
    const float sw = r_pow2(268 - e), inv_sw = r_pow2(e - 14);
    if (step == 0 && tid == 0) reinterpret_cast<float*>(ws)[slice] = inv_sw;
    char* img = ws + 64 + (int64_t)(slice * NSTEP + step) * R_STEP;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int idx = it * 512 + tid;                    // (sub-image, feature, half)
        const int sub = idx >> 8, n = (idx & 255) >> 1, hf = idx & 1;
        const int gate = sub >> 1;                         // sub 0,1: r; 2,3: z; 4: n (W_ih); 5: gh_n (W_hh)
        const float* src = ((sub & 1) ? W_hh : W_ih) + (int64_t)(128 * slice + n) * 3 * H + gate * H + 16 * step + 4 * hf;
        h16x8 ph, pl;
        r_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 8), sw, ph, pl);
        char* dst = img + sub * R_SUB + hf * 2048 + n * 16;
        *reinterpret_cast<h16x8*>(dst) = ph;
        *reinterpret_cast<h16x8*>(dst + 4096) = pl;
    }
}

// ------------------------------------------------------------------------------------------ dm | dh
// Block = 8 waves in two ROLES: waves 4-7 are PRODUCERS (vector pipe: read the row slices, form the gate gradients, scale,
// split), waves 0-3 are CONSUMERS (matrix pipe: hold the accumulators, 72 MFMAs per step); producer p + 4 and consumer p
// share a SIMD and a tile of 32 atoms (lane = atom | half) and meet in LDS.  One role per wave because one wave cannot hold
// both: 128 accumulator registers + the next step's 48 row registers in flight + pieces and fragments is more than the 256
// registers of a wave at two per SIMD (it spilled), and a wave alone on its SIMD runs its vector and matrix phases one after
// the other (profiles/r04_gru_bwd_ablation.md has both measurements).  Split, each role fits, and the two pipes of a SIMD
// work at the same time.
//   step g, phase 1:  consumers: start the copy of weight image g + 1 -> rescale / direct term of dh (from the producer's
//                     note) -> 72 MFMAs on (weight image g, pieces g) -> the copy has landed;
//                     producers: gate gradients of step g + 1 from the row slices requested two steps ago -> pieces in
//                     registers -> request the slices of step g + 3 into the registers just consumed;
//            barrier
//            phase 2: producers park pieces g + 1 (+ direct term, scale note) in LDS;          barrier
// The consumer's accumulators are transposed (rows = features, columns = atoms): see the head of this file.
// NORM_OUT: the backward of the norm behind this update is applied to dout (column constants kn = k1 | k2 | k4, in LDS).
// NORM_IN: `h` = norm(y_in): the column sums of dh and dh * y_in over all atoms go to `sums` (2 H doubles, accumulated).
// a value every lane holds alike, moved to scalar registers (what the callee of a non-inlined call cannot know by itself)
template <typename T>
__device__ __forceinline__ T* r_uniform(T* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t r_uniform(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int r_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float r_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

struct RcDxCtx {
    const float *dout, *h, *mask, *saved, *y_in;
    const char* wws;
    float *dm, *dh;
    int64_t V, tiles, nrounds, total;
    int slice, pblock, pblocks, pair;
    float inv_sw;
};
// LDS of the dm | dh kernel: two weight images | per pair: 8 KB of pieces [segment][hi | lo][position 16 B], 2 KB direct term
// [u][position 16 B], 512 B note [ratio 64 floats | exponent 64 ints] | per consumer 4.5 KB for the epilogue's transposition
// | (NORM_OUT) kn | (NORM_IN) the sums
constexpr int R_PIECES = 2 * R_STEP, R_GZ = R_PIECES + 4 * 8192, R_NOTE = R_GZ + 4 * 2048, R_EPI = R_NOTE + 4 * 512;
constexpr int R_KN = R_EPI + 4 * 4608;

#define RC_DX_PROLOGUE()                                                                                                  \
    constexpr int NSTEP = H / 16;                                                                                         \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                           \
    float* kn_s = reinterpret_cast<float*>(smem + R_KN);                                                                  \
    double* stat_s = reinterpret_cast<double*>(smem + R_KN + (NORM_OUT ? 3 * H * 4 : 0));                                 \
    (void)kn_s; (void)stat_s;                                                                                             \
    const float *dout = r_uniform(c.dout), *h = r_uniform(c.h), *mask = r_uniform(c.mask), *saved = r_uniform(c.saved);   \
    const float* y_in = r_uniform(c.y_in);                                                                                \
    const char* wws = r_uniform(c.wws);                                                                                   \
    float *dm = r_uniform(c.dm), *dh = r_uniform(c.dh);                                                                   \
    (void)dout; (void)h; (void)mask; (void)saved; (void)y_in; (void)wws; (void)dm; (void)dh;                              \
    const int64_t V = r_uniform(c.V), tiles = r_uniform(c.tiles), nrounds = r_uniform(c.nrounds), total = r_uniform(c.total); \
    (void)tiles;                                                                                                          \
    const int slice = r_uniform(c.slice), pblock = r_uniform(c.pblock), pblocks = r_uniform(c.pblocks), pair = r_uniform(c.pair); \
    (void)slice;                                                                                                          \
    const float inv_sw = r_uniform(c.inv_sw);                                                                             \
    const int lane = threadIdx.x & 63;                                                                                    \
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                                      \
    (void)wv;                                                                                                             \
    const int a = lane & 31, hi = lane >> 5;                                                                              \
    (void)a; (void)hi;                                                                                                    \
    /* the tile of this pair in round rd (clamped to the last round: such a pair computes and stores nothing new) */      \
    auto tile_of = [&](int64_t rd) {                                                                                      \
        if (rd >= nrounds) rd = nrounds - 1;                                                                              \
        return ((int64_t)pblock + rd * pblocks) * 4 + pair;                                                               \
    };                                                                                                                    \
    auto row_of = [&](int64_t t) {                                                                                        \
        int64_t rw = t * 32 + a;                                                                                          \
        return rw < V ? rw : V - 1;                                                                                       \
    };                                                                                                                    \
    (void)row_of

// The two roles are separate functions on purpose: compiled as two branches of one function, the invariants of one role
// (store and copy addresses, LDS positions) stayed in registers through the other's loop and the kernel spilled.
template <int H, bool HAS_MASK, bool NORM_OUT, bool NORM_IN>
__device__ __attribute__((noinline)) void gru_rc_dx_producer(const RcDxCtx& c) {
    RC_DX_PROLOGUE();
        // ================================================================================================ producers
        // Lane = (chunk c = lane & 3: 4 columns of the step's 16, rows lane >> 2 and 16 + (lane >> 2) of the tile): a load
        // instruction covers 16 rows x 64 contiguous bytes (lane = atom | half, the layout the matrix pipe wants, touches 32
        // rows x 32 bytes and kept the producers waiting at the address unit: profiles/r04_gru_bwd_ablation.md).  The pieces
        // are parked where the consumer's lane (atom, half) reads them: chunk c of row a is k-slots 4 (c >> 1) .. of lane
        // (a, c & 1).  A row's maximum is a maximum over the 4 lanes of a quad.
        const float sw = 1.0f / inv_sw;
        const int c4 = lane & 3, q16 = lane >> 2;
        struct Rows { f32x4 d[2], hv[2], r[2], z[2], n[2], nh[2]; float mk[2]; };   // [i]: row q16 + 16 i
        auto load_rows = [&](int64_t n) {                  // the slices of global step n (clamped to the last step)
            Rows q;
            if (n >= total) n = total - 1;
            const int st = (int)(n % NSTEP);
            const int64_t t = tile_of(n / NSTEP);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int64_t rw = t * 32 + q16 + 16 * i;
                if (rw >= V) rw = V - 1;
#ifdef MPNN_ABL_RC_HOTROWS      // timing experiment only (wrong results): every tile reads the first 256 atoms' rows (L2-resident)
                rw &= 255;
#endif
                const int c0 = 16 * st + 4 * c4;
                const float* ps = saved + rw * 4 * H + c0;
                q.d[i] = *reinterpret_cast<const f32x4*>(dout + rw * H + c0);
                q.z[i] = *reinterpret_cast<const f32x4*>(ps + H);
                q.n[i] = *reinterpret_cast<const f32x4*>(ps + 2 * H);
                q.hv[i] = *reinterpret_cast<const f32x4*>(h + rw * H + c0);
                q.r[i] = *reinterpret_cast<const f32x4*>(ps);
                q.nh[i] = *reinterpret_cast<const f32x4*>(ps + 3 * H);
                q.mk[i] = HAS_MASK ? mask[rw] : 1.0f;      // (times "row < V" where it is used)
            }
            return q;
        };
        int e_cur[2] = {51, 51};
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        h16x4 ph[4][2], pl[4][2];                          // pieces of (dar, daz, dan, dnh), rows i, this lane's 4 columns
        f32x4 gzv[2];
        float ratio[2] = {1.0f, 1.0f};
        // gate gradients of global step n from `q` -> pieces, direct term, scale note (all in registers)
        auto produce = [&](int64_t n, const Rows& q) {
            const int st = (int)(n % NSTEP);
            const int64_t t = tile_of(n / NSTEP);
            f32x4 k1 = {0.f, 0.f, 0.f, 0.f}, k2 = k1, k4 = k1;
            if (NORM_OUT) {
                const int c = 16 * st + 4 * c4;
                k1 = *reinterpret_cast<const f32x4*>(kn_s + c);
                k2 = *reinterpret_cast<const f32x4*>(kn_s + H + c);
                k4 = *reinterpret_cast<const f32x4*>(kn_s + 2 * H + c);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float mk = (t * 32 + q16 + 16 * i < V) ? q.mk[i] : 0.0f;
                f32x4 dar, daz, dan, dnh, gz;
                r_gate_grads4<NORM_OUT>(q.d[i], q.hv[i], q.r[i], q.z[i], q.n[i], q.nh[i], mk, k1, k2, k4, dar, daz, dan, dnh, gz);
                float mx = r_max4(r_max4(r_max4(r_max4(0.f, dar), daz), dan), dnh);
                mx = r_quad_max(mx);                       // the row's other three chunks
                const int e_mx = r_exp(mx);
                ratio[i] = 1.0f;
                if (st == 0) {
                    e_cur[i] = e_mx;
                } else if (e_mx > e_cur[i] + 3) {          // three bits of headroom are used up
                    ratio[i] = r_pow2(127 + e_cur[i] - e_mx);
                    e_cur[i] = e_mx;
                }
                const float sg = r_pow2(265 - e_cur[i]);   // row maximum * sg in [2^11, 2^12) when set, < 2^16 always
                const float gsc = sg * sw;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) gzv[i][t4] = gz[t4] * gsc;
                r_split4(dar, sg, ph[0][i], pl[0][i]);
                r_split4(daz, sg, ph[1][i], pl[1][i]);
                r_split4(dan, sg, ph[2][i], pl[2][i]);
                r_split4(dnh, sg, ph[3][i], pl[3][i]);
            }
        };
        // consumer lane (a, c & 1): pieces at 16 lane' + 8 (c >> 1); direct term u = c >> 1
        char* dst_p = smem + R_PIECES + pair * 8192 + (2 * q16 + (c4 & 1)) * 16 + (c4 >> 1) * 8;
        char* dst_g = smem + R_GZ + pair * 2048 + (c4 >> 1) * 1024 + (2 * q16 + (c4 & 1)) * 16;
        float* dst_r = reinterpret_cast<float*>(smem + R_NOTE + pair * 512) + q16 + 32 * (c4 & 1);
        auto park = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    *reinterpret_cast<h16x4*>(dst_p + sgm * 2048 + i * 512) = ph[sgm][i];
                    *reinterpret_cast<h16x4*>(dst_p + sgm * 2048 + 1024 + i * 512) = pl[sgm][i];
                }
                *reinterpret_cast<f32x4*>(dst_g + i * 512) = gzv[i];
                if (c4 < 2) {
                    dst_r[16 * i] = ratio[i];
                    reinterpret_cast<int*>(dst_r + 64)[16 * i] = e_cur[i];
                }
            }
        };
        // the consumers are the older waves and would win every tie at the SIMD's issue port; they wait on the matrix pipe
        // most of the time, so the vector work goes first
        __builtin_amdgcn_s_setprio(2);
        Rows ra = load_rows(0), rb = load_rows(1);
        produce(0, ra);
        ra = load_rows(2);
        park();
        r_barrier_lds();                                   // (P0) pieces of step 0 are parked, weight image 0 has landed
        // one step: pieces of step g + 1 from `q` (requested two steps ago), then `q` takes the slices of step g + 3
#ifdef RC_STAMP
#define RC_PSTAMP()                                                                                                      \
    if (blockIdx.x == 3 && lane == 0 && wv == 4) {                                                                        \
        RC_ADD_STAMP(16, p1 - p0); RC_ADD_STAMP(17, p2 - p1); RC_ADD_STAMP(18, p3 - p2); RC_ADD_STAMP(19, p4 - p3);       \
        RC_ADD_STAMP(20, p5 - p4); RC_ADD_STAMP(21, 1);                                                                   \
    }
#else
#define RC_PSTAMP()
#endif
#define RC_PRODUCER_STEP(G, Q)                                                                       \
    {                                                                                                \
        RC_T(p0);                                                                                    \
        const bool more = (G) + 1 < total;                                                           \
        if (more) produce((G) + 1, Q);                                                               \
        /* (the pieces are FINISHED here: without this the conversions sink below the barrier, into the phase the    \
           consumers spend waiting) */                                                                 \
        _Pragma("unroll") for (int sgm_ = 0; sgm_ < 4; ++sgm_)                                        \
            asm volatile("" ::"v"(ph[sgm_][0]), "v"(pl[sgm_][0]), "v"(ph[sgm_][1]), "v"(pl[sgm_][1]));   \
        asm volatile("" ::"v"(gzv[0]), "v"(gzv[1]));                                                  \
        RC_T(p1);                                                                                    \
        Q = load_rows((G) + 3);                                                                      \
        RC_T(p2);                                                                                    \
        r_barrier_lds(); /* (B1) the consumers are done with pieces g */                             \
        RC_T(p3);                                                                                    \
        if (more) park();                                                                            \
        RC_T(p4);                                                                                    \
        r_barrier_lds(); /* (B2) */                                                                  \
        RC_T(p5);                                                                                    \
        RC_PSTAMP()                                                                                  \
    }
#pragma unroll 1
        for (int64_t g = 0; g < total; g += 2) {
            RC_PRODUCER_STEP(g, rb)
            RC_PRODUCER_STEP(g + 1, ra)
        }
#undef RC_PRODUCER_STEP
#undef RC_PSTAMP
}

template <int H, bool HAS_MASK, bool NORM_OUT, bool NORM_IN>
__device__ __attribute__((noinline)) void gru_rc_dx_consumer(const RcDxCtx& c) {
    RC_DX_PROLOGUE();
    // consumer lane (a, half) reads its 16 bytes at position 2 a + half: the producers' 8-byte writes (4 lanes per atom: chunks
    // 0..3 -> positions 2 a, 2 a + 1, each half-filled twice) then fall on 32 different banks
    char* my_pieces = smem + R_PIECES + pair * 8192 + (2 * a + hi) * 16;
    char* my_gz = smem + R_GZ + pair * 2048 + (2 * a + hi) * 16;
    float* my_ratio = reinterpret_cast<float*>(smem + R_NOTE + pair * 512) + lane;
    int* my_exp = reinterpret_cast<int*>(smem + R_NOTE + pair * 512 + 256) + lane;
    // this wave's share of weight image `st` -> buffer buf: twelve 1 KB copies
    auto w_issue = [&](int st, int buf) {
        const char* src = wws + 64 + (int64_t)(slice * NSTEP + st) * R_STEP + (12 * pair) * 1024;   // (wave-uniform)
        const char* dst = smem + buf * R_STEP + (12 * pair) * 1024;
        const unsigned voff = lane * 16;
#pragma unroll
        for (int i = 0; i < 12; ++i) r_copy_to_lds_s(src + i * 1024, voff, dst + i * 1024);
    };
    f32x16 d_m[4], d_h[4];
    auto afrag = [&](const char* wb, int sub, int piece, int nb) {
        return *reinterpret_cast<const h16x8*>(wb + sub * R_SUB + piece * 4096 + hi * 2048 + (32 * nb + a) * 16);
    };
    // one gate segment's pieces against the sub-images `sub0` -> d0 and (sub1 >= 0) `sub1` -> d1
    auto product = [&](int sgm, const char* wb, f32x16 (&d0)[4], int sub0, f32x16 (&d1)[4], int sub1) {
        const h16x8 gh = *reinterpret_cast<const h16x8*>(my_pieces + sgm * 2048);
        const h16x8 gl = *reinterpret_cast<const h16x8*>(my_pieces + sgm * 2048 + 1024);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const h16x8 w0h = afrag(wb, sub0, 0, nb), w0l = afrag(wb, sub0, 1, nb);
            d0[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, gh, d0[nb], 0, 0, 0);
            if (sub1 >= 0) {
                const h16x8 w1h = afrag(wb, sub1, 0, nb), w1l = afrag(wb, sub1, 1, nb);
                d1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, gh, d1[nb], 0, 0, 0);
                d0[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, gl, d0[nb], 0, 0, 0);
                d1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, gl, d1[nb], 0, 0, 0);
                d0[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, gh, d0[nb], 0, 0, 0);
                d1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, gh, d1[nb], 0, 0, 0);
            } else {
                d0[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, gl, d0[nb], 0, 0, 0);
                d0[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, gh, d0[nb], 0, 0, 0);
            }
        }
    };
    // the direct term of dh, into the accumulator entries this lane owns: block nb, entries 8 s + 4 u + t
    auto add_gz = [&](int nb, int s, const f32x4& g0, const f32x4& g1) {
#define RC_ADD(NB, S)                                                                                                  \
    case 2 * NB + S:                                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) { d_h[NB][8 * S + t] += g0[t]; d_h[NB][8 * S + 4 + t] += g1[t]; } \
        break;
        switch (2 * nb + s) {
            RC_ADD(0, 0) RC_ADD(0, 1) RC_ADD(1, 0) RC_ADD(1, 1) RC_ADD(2, 0) RC_ADD(2, 1) RC_ADD(3, 0) RC_ADD(3, 1)
            default: break;
        }
#undef RC_ADD
    };

    __builtin_amdgcn_s_setprio(3);                         // short vector bursts of the consumer go first; its MFMA phase yields
    w_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    r_barrier_lds();                                       // (P0)
    int st = 0;
    int64_t rd = 0;
#pragma unroll 1
    for (int64_t g = 0; g < total; ++g) {
        const int buf = (int)(g & 1);
        RC_T(c0);
        // the producer's note for this step, requested before the copies are issued (their issue covers the LDS latency)
        const float ratio = *my_ratio;                     // < 1 where this step's gradients outgrew the atom's scale
        const f32x4 gz0 = *reinterpret_cast<const f32x4*>(my_gz), gz1 = *reinterpret_cast<const f32x4*>(my_gz + 1024);
        const int e_note = *my_exp;
        if (g + 1 < total) w_issue(st + 1 == NSTEP ? 0 : st + 1, buf ^ 1);   // (its issue covers the latency of the reads above)
        RC_T(c1);
        if (st == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { d_m[nb][i] = 0.f; d_h[nb][i] = 0.f; }
        } else {
            if (__builtin_amdgcn_ballot_w64(ratio != 1.0f) != 0) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { d_m[nb][i] *= ratio; d_h[nb][i] *= ratio; }
            }
        }
        {
            const int nb = (st >> 1) - 4 * slice;
            if (nb >= 0 && nb < 4) add_gz(nb, st & 1, gz0, gz1);
        }
        const char* wb = smem + buf * R_STEP;
        RC_T(c2);
        __builtin_amdgcn_s_setprio(0);                     // (while this wave waits on the matrix pipe the producer's vector work goes first)
        product(0, wb, d_m, 0, d_h, 1);
        product(1, wb, d_m, 2, d_h, 3);
        product(2, wb, d_m, 4, d_h, -1);
        product(3, wb, d_h, 5, d_m, -1);
        __builtin_amdgcn_s_setprio(3);
        RC_T(c3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of weight image g + 1 (issued a step ago) has landed
        RC_T(c4);
        if (st + 1 == NSTEP) {
            // ---- epilogue: this lane's atom, 16 x 4 consecutive features of dm and of dh
            const int64_t t = tile_of(rd);
            const float un = r_pow2(e_note - 11) * inv_sw;                     // 1 / (sg * sw)
            {
                // A lane holds 4 x 4 consecutive features of ITS atom per block: stored from here an instruction would touch 32
                // rows x 32 bytes (measured: 10.7 k cycles per tile).  One 32 x 32 block at a time goes through this wave's
                // 4.5 KB of LDS (rows of 36 floats) and leaves as whole 128-byte row pieces: lane -> row (lane >> 3) + 8 k,
                // 16-byte chunk lane & 7.
                char* epi = smem + R_EPI + pair * 4608;
                unsigned ln = (unsigned)lane;              // opaque: keeps the lane's offsets from becoming loop invariants
                asm volatile("" : "+v"(ln));               // (as invariants they sat in registers across the MFMA phase and spilled)
                const int rr = ln >> 3, ch = ln & 7;
                const int a = ln & 31, hi = ln >> 5;
#pragma unroll
                for (int mat = 0; mat < 2; ++mat)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const f32x16& d = mat ? d_h[nb] : d_m[nb];
                        // NORM_IN: the column sums over the tile's atoms of dh and dh * y_in ride on the row layout the block
                        // leaves in: a lane adds its four rows in registers, the eight lanes of a column group meet in three
                        // steps (as a butterfly over the 32 lanes of every accumulator entry this was 640 ds_bpermute per
                        // tile and +1.7 ms per launch at c3's size); y_in is requested here, whole lines, ahead of the round trip
                        f32x4 yv[4];
                        if (NORM_IN && mat == 1) {
                            const float* py = y_in + 128 * slice + 32 * nb + 4 * ch;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int64_t gr = t * 32 + rr + 8 * k;    // (a block's last round can hold tiles past V: clamped,
                                yv[k] = *reinterpret_cast<const f32x4*>(py + (gr < V ? gr : V - 1) * H);   // and not summed below)
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = {d[4 * q] * un, d[4 * q + 1] * un, d[4 * q + 2] * un, d[4 * q + 3] * un};
                            *reinterpret_cast<f32x4*>(epi + a * 144 + (8 * q + 4 * hi) * 4) = v;
                        }
                        float* po = (mat ? dh : dm) + (t * 32) * H + 128 * slice + 32 * nb + 4 * ch;
                        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = rr + 8 * k;
                            const f32x4 v = *reinterpret_cast<const f32x4*>(epi + r * 144 + ch * 16);
                            if (t * 32 + r < V) {
                                *reinterpret_cast<f32x4*>(po + (int64_t)r * H) = v;
                                if (NORM_IN && mat == 1) {
                                    s1 += v;
                                    s2 += v * yv[k];
                                }
                            }
                        }
                        if (NORM_IN && mat == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float x1 = s1[j], x2 = s2[j];
                                x1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x1), 0x128, 0xf, 0xf, true));   // row_ror:8 = lane ^ 8
                                x2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x2), 0x128, 0xf, 0xf, true));
                                x1 += __shfl_xor(x1, 16);
                                x2 += __shfl_xor(x2, 16);
                                x1 += __shfl_xor(x1, 32);
                                x2 += __shfl_xor(x2, 32);
                                if (rr == 0) {
                                    const int f = 32 * nb + 4 * ch + j;
                                    atomicAdd(&stat_s[f], (double)x1);
                                    atomicAdd(&stat_s[128 + f], (double)x2);
                                }
                            }
                        }
                    }
            }
            ++rd;
            st = 0;
        } else {
            ++st;
        }
        RC_T(c5);
        r_barrier_lds();                                   // (B1) pieces g are consumed, and so is weight image g
        RC_T(c6);
        r_barrier_lds();                                   // (B2) pieces g + 1 are parked
        RC_T(c7);
#ifdef RC_STAMP
        if (blockIdx.x == 3 && lane == 0 && wv == 0) {
            RC_ADD_STAMP(0, c1 - c0); RC_ADD_STAMP(1, c2 - c1); RC_ADD_STAMP(2, c3 - c2); RC_ADD_STAMP(3, c4 - c3);
            RC_ADD_STAMP(4, c5 - c4); RC_ADD_STAMP(5, c6 - c5); RC_ADD_STAMP(6, c7 - c6); RC_ADD_STAMP(7, 1);
        }
#endif
    }
}
#undef RC_DX_PROLOGUE

template <int H, bool HAS_MASK, bool NORM_OUT, bool NORM_IN>
__global__ void __launch_bounds__(512) gru_rc_dx_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                                        const float* __restrict__ mask, const float* __restrict__ saved,
                                                        const float* __restrict__ kn, const char* __restrict__ wws,
                                                        float* __restrict__ dm, float* __restrict__ dh, int64_t V,
                                                        const float* __restrict__ y_in, double* sums) {
    constexpr int NS = H / 128, NSTEP = H / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* kn_s = reinterpret_cast<float*>(smem + R_KN);
    double* stat_s = reinterpret_cast<double*>(smem + R_KN + (NORM_OUT ? 3 * H * 4 : 0));   // [2][128]
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RcDxCtx c;
    c.dout = dout; c.h = h; c.mask = mask; c.saved = saved; c.y_in = y_in; c.wws = wws; c.dm = dm; c.dh = dh; c.V = V;
    c.slice = jb % NS;
    c.pblock = (jb / NS) * 8 + xcd;
    c.pblocks = gridDim.x / NS;
    c.pair = wv & 3;
    c.tiles = (V + 31) / 32;
    const int64_t rounds_total = (c.tiles + 3) / 4;            // a round = 4 tiles, one per pair
    if (c.pblock >= rounds_total) return;
    c.nrounds = (rounds_total - c.pblock + c.pblocks - 1) / c.pblocks;
    c.total = c.nrounds * NSTEP;
    if (NORM_OUT)
        for (int i = tid; i < 3 * H; i += 512) kn_s[i] = kn[i];
    if (NORM_IN)
        for (int i = tid; i < 2 * 128; i += 512) stat_s[i] = 0.0;
    if (NORM_OUT || NORM_IN) __syncthreads();
    c.inv_sw = reinterpret_cast<const float*>(wws)[c.slice];
    if (wv >= 4) gru_rc_dx_producer<H, HAS_MASK, NORM_OUT, NORM_IN>(c);
    else gru_rc_dx_consumer<H, HAS_MASK, NORM_OUT, NORM_IN>(c);
    if (NORM_IN) {
        __syncthreads();                                   // both roles arrive here after the same number of barriers
        if (tid < 256) atomicAdd(sums + (tid >> 7) * H + 128 * c.slice + (tid & 127), stat_s[tid]);
    }
}

// ----------------------------------------------------------------------------------------------------------- dW
// H = 128.  dW_ih = m^T [dar daz dan], dW_hh = h^T [dar daz dnh] (128 x 384 each).  Block type = COLUMN HALF: a block owns
// columns 64 half .. 64 half + 63 of every gate segment, for BOTH matrices (2 x 4 x 6 tiles of 32 x 32; wave = matrix x 2
// a-tiles x 3 b-tiles).  A block per matrix (the first form of this kernel) had every gate gradient formed twice, once per
// matrix, and the kernel is bound by its vector instructions (profiles/r04_gru_bwd_ablation.md); here each is formed once,
// and what is read twice is m | h (128 columns of each per block: the A operand of both products).
// LDS per buffer: four images x two pieces: slot 0 = dar | daz, slot 1 = dan | dnh (64 columns each), slot 2 = m, slot 3 = h;
// an image = [kstep 8][32 rows][16 columns] of a 32-atom tile; two buffers.  Thread (row = tid >> 4, c16 = tid & 15) owns
// gate columns 64 half + 4 c16 .. + 3 and m | h columns 8 c16 .. + 7: it requests them two tiles ahead of their contraction,
// forms the four gate segments while the tile before is contracted, takes the row's maximum over the 16 lanes that share
// the row (the pieces' power-of-two scale sg_row: the largest magnitude lands in [2^14, 2^15)), parks the pieces, and keeps
// the column sums (bias gradients).  m | h are split behind sx_row = C / sg_row, C = running minimum over the block's tiles of
// (smallest sg_row of the tile) x (best scale of the tile's m | h): every product carries C (gru_bwd_f16.hip has the argument).
template <bool HAS_MASK, bool NORM_OUT>
__global__ void __launch_bounds__(512) gru_rc_dw_kernel(const float* __restrict__ dout, const float* __restrict__ m,
                                                        const float* __restrict__ h, const float* __restrict__ mask,
                                                        const float* __restrict__ saved, const float* __restrict__ kn,
                                                        float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, int64_t V) {
    constexpr int H = 128, NB = 3, NACC = 6;
    constexpr int BUF = 8 * R_IMG;             // image (piece, slot) at (4 * piece + slot) * R_IMG
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 2 * BUF);              // [2 parities][max |x| 8 | max 1/sg 8]
    float* bsum = reinterpret_cast<float*>(smem);                        // (after the loop) column sums across the waves

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int half = jb & 1;                                   // the two halves' blocks of a tile stream sit on one XCD
    const int64_t tiles = (V + 31) / 32;
    const int64_t t0 = (jb >> 1) * 8 + xcd, tstep = gridDim.x / 2;
    if (t0 >= tiles) return;

    const int srow = tid >> 4, c16 = tid & 15;
    const int g_dst = (c16 >> 2) * 1024 + srow * 32 + (c16 & 3) * 8;    // this thread's 8 bytes (4 gate columns) inside 64 columns of an image
    const int x_dst = (c16 >> 1) * 1024 + srow * 32 + (c16 & 1) * 16;   // this thread's 16 bytes (8 columns of m or h) inside an image
    struct Rows { f32x4 d, hv, r, z, n, nh, xm[2], xh[2]; float mk, live; };   // nothing here is USED before park_*: no wait at the requests
    auto load_rows = [&](int64_t t) {
        Rows q;
        int64_t row = t * 32 + srow;
        const bool ok = row < V;
        if (!ok) row = V - 1;
#ifdef MPNN_ABL_RC_HOTROWS      // timing experiment only (wrong results): every tile reads the first 256 atoms' rows (L2-resident)
        row &= 255;
#endif
        q.mk = HAS_MASK ? mask[row] : 1.0f;               // (times `live` where it is used)
        q.live = ok ? 1.0f : 0.0f;                         // rows past V count as zeros
        const int c4 = 64 * half + 4 * c16;
        const float* ps = saved + row * 4 * H + c4;
        q.d = *reinterpret_cast<const f32x4*>(dout + row * H + c4);
        q.hv = *reinterpret_cast<const f32x4*>(h + row * H + c4);
        q.r = *reinterpret_cast<const f32x4*>(ps);
        q.z = *reinterpret_cast<const f32x4*>(ps + H);
        q.n = *reinterpret_cast<const f32x4*>(ps + 2 * H);
        q.nh = *reinterpret_cast<const f32x4*>(ps + 3 * H);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            q.xm[u] = *reinterpret_cast<const f32x4*>(m + row * H + 8 * c16 + 4 * u);
            q.xh[u] = *reinterpret_cast<const f32x4*>(h + row * H + 8 * c16 + 4 * u);
        }
        return q;
    };
    float cs[4][4];                                            // column sums of the four segments, this thread's 4 columns
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) cs[s][u] = 0.f;
    unsigned kofs = 0;
    if (NORM_OUT) asm volatile("" : "+v"(kofs));               // (opaque: the column constants are re-read per tile, not kept)

    // gate pieces of one tile -> buffer T; returns this row's 1 / sg
    auto park_gates = [&](const Rows& q, char* T) {
        f32x4 k1 = {0.f, 0.f, 0.f, 0.f}, k2 = k1, k4 = k1;
        if (NORM_OUT) {
            const float* kp = kn + kofs + 64 * half + 4 * c16;
            k1 = *reinterpret_cast<const f32x4*>(kp);
            k2 = *reinterpret_cast<const f32x4*>(kp + H);
            k4 = *reinterpret_cast<const f32x4*>(kp + 2 * H);
        }
        f32x4 seg[4], gz;
        r_gate_grads4<NORM_OUT>(q.d, q.hv, q.r, q.z, q.n, q.nh, q.mk * q.live, k1, k2, k4, seg[0], seg[1], seg[2], seg[3], gz);
        float mx = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mx = r_max4(mx, seg[s]);
#pragma unroll
            for (int t = 0; t < 4; ++t) cs[s][t] += seg[s][t];
        }
        mx = r_row16_max(mx);                              // the 16 lanes that share the row are one DPP row
        const int e = r_exp(mx);
        const float sg = r_pow2(268 - e);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                      // image slot s >> 1, columns 64 (s & 1) ..
            r_h16x4 ph, pl;
            r_split4(seg[s], sg, ph, pl);
            char* dst = T + (s >> 1) * R_IMG + (s & 1) * 4096 + g_dst;
            *reinterpret_cast<r_h16x4*>(dst) = ph;
            *reinterpret_cast<r_h16x4*>(dst + 4 * R_IMG) = pl;
        }
        return r_pow2(e - 14);
    };
    auto publish = [&](const Rows& q, float inv_sg, int par) {
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) mx = r_max4(r_max4(mx, q.xm[u]), q.xh[u]);
        mx = r_wave_max(mx * q.live);
        const float iv = r_wave_max(inv_sg);               // largest inverse = the scale of the tile's largest row
        if (lane == 0) { red[16 * par + wv] = mx; red[16 * par + 8 + wv] = iv; }
    };
    float C_run = 3.0e38f;
    auto park_x = [&](const Rows& q, float inv_sg, char* T, int par) {   // after the barrier that follows publish()
        float xm = red[16 * par], ivm = red[16 * par + 8];
#pragma unroll
        for (int u = 1; u < 8; ++u) { xm = fmaxf(xm, red[16 * par + u]); ivm = fmaxf(ivm, red[16 * par + 8 + u]); }
        int ex = (__float_as_int(xm) >> 23) & 0xff;
        ex = ex < 111 ? 111 : (ex > 187 ? 187 : ex);
        const float sxo = r_pow2(268 - ex);
        const float sgm = r_pow2(254 - ((__float_as_int(ivm) >> 23) & 0xff));
        C_run = fminf(C_run, sgm * sxo);
        const float sx = C_run * inv_sg * q.live;
        h16x8 ph, pl;
        r_split8(q.xm[0], q.xm[1], sx, ph, pl);
        *reinterpret_cast<h16x8*>(T + 2 * R_IMG + x_dst) = ph;
        *reinterpret_cast<h16x8*>(T + 6 * R_IMG + x_dst) = pl;
        r_split8(q.xh[0], q.xh[1], sx, ph, pl);
        *reinterpret_cast<h16x8*>(T + 3 * R_IMG + x_dst) = ph;
        *reinterpret_cast<h16x8*>(T + 7 * R_IMG + x_dst) = pl;
    };

    // transposed reads (as gru_bwd128_f16.hip): a 16-lane group takes rows 8 (g2 >> 1) + 4 j + (0..3), columns 16 (g2 & 1) +
    // (0..15) of 32-column block cb of an image; lane 4 q + p supplies row q, columns 4 p .. 4 p + 3
    const int mat = wv & 1, ag = (wv >> 1) & 1, bg = wv >> 2;
    const int g2 = lane >> 4, u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3;
    auto tr_addr = [&](int slot, int cb, int j) {
        return slot * R_IMG + (2 * cb + (g2 & 1)) * 1024 + (8 * (g2 >> 1) + 4 * j + q4) * 32 + p4 * 8;
    };
    // b-tile bt = 3 bg + b of this wave's matrix: segment bt >> 1 (the third one is dan for W_ih, dnh for W_hh), 32-column
    // half bt & 1 of the block's 64 columns
    auto seg_of = [&](int bt) { return (bt >> 1) < 2 ? (bt >> 1) : 2 + mat; };
    int LA[NB][2], LX[2][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int bt = NB * bg + b, sg_ = seg_of(bt);
            LA[b][j] = tr_addr(sg_ >> 1, 2 * (sg_ & 1) + (bt & 1), j);
        }
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int j = 0; j < 2; ++j) LX[a2][j] = tr_addr(2 + mat, 2 * ag + a2, j);

    f32x16 R[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) R[j][q] = 0.f;

    Rows nxt;                                              // tile t + 1's slices while tile t is contracted
    {
        const Rows first = load_rows(t0);
        nxt = load_rows(t0 + tstep < tiles ? t0 + tstep : t0);
        const float iv = park_gates(first, smem);
        publish(first, iv, 0);
        __syncthreads();
        park_x(first, iv, smem, 0);
    }
    float C_acc = C_run, C_cur = C_run;
    int cur = 0;
#pragma unroll 1
    for (int64_t t = t0; t < tiles; t += tstep) {
        RC_T(w0);
        r_barrier_lds();                                   // buffer `cur` is complete; the other one is free
        RC_T(w1);
        const char* T = smem + cur * BUF;
        char* Tn = smem + (cur ^ 1) * BUF;
        const bool has1 = t + tstep < tiles;
        // tile t + 1: gate pieces into the free buffer, then its registers take tile t + 2
        float iv = 0.f;
        if (has1) iv = park_gates(nxt, Tn);
        RC_T(w2);
        Rows keep;                                         // (only m | h of tile t + 1 are still needed)
        keep.xm[0] = nxt.xm[0]; keep.xm[1] = nxt.xm[1]; keep.xh[0] = nxt.xh[0]; keep.xh[1] = nxt.xh[1]; keep.live = nxt.live;
        const int64_t t2 = t + 2 * tstep;
        nxt = load_rows(t2 < tiles ? t2 : t);              // unconditional, clamped
        if (__builtin_amdgcn_readfirstlane(__float_as_int(C_cur)) != __builtin_amdgcn_readfirstlane(__float_as_int(C_acc))) {
            const float ratio = C_cur / C_acc;             // < 1, a power of two
#pragma unroll
            for (int j = 0; j < NACC; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) R[j][q] *= ratio;
            C_acc = C_cur;
        }
        RC_T(w3);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const char* Tb = T + 512 * st;                 // rows +16
            const h16x8 a0h = r_tr8(Tb + LX[0][0], Tb + LX[0][1]);
            const h16x8 a0l = r_tr8(Tb + 4 * R_IMG + LX[0][0], Tb + 4 * R_IMG + LX[0][1]);
            const h16x8 a1h = r_tr8(Tb + LX[1][0], Tb + LX[1][1]);
            const h16x8 a1l = r_tr8(Tb + 4 * R_IMG + LX[1][0], Tb + 4 * R_IMG + LX[1][1]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const h16x8 bh = r_tr8(Tb + LA[b][0], Tb + LA[b][1]);
                const h16x8 bl = r_tr8(Tb + 4 * R_IMG + LA[b][0], Tb + 4 * R_IMG + LA[b][1]);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, R[NB + b], 0, 0, 0);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, R[NB + b], 0, 0, 0);
                R[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, R[b], 0, 0, 0);
                R[NB + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, R[NB + b], 0, 0, 0);
            }
        }
        RC_T(w4);
        if (has1) {
            publish(keep, iv, cur ^ 1);
            RC_T(w5);
            r_barrier_lds();                               // the maxima of tile t + 1 are in LDS
            RC_T(w6);
            park_x(keep, iv, Tn, cur ^ 1);
            C_cur = C_run;
            RC_T(w7);
#ifdef RC_STAMP
            if (blockIdx.x == 16 && lane == 0 && (wv == 0 || wv == 7)) {
                const int b = 32 + 12 * (wv == 7);
                RC_ADD_STAMP(b + 0, w1 - w0); RC_ADD_STAMP(b + 1, w2 - w1); RC_ADD_STAMP(b + 2, w3 - w2); RC_ADD_STAMP(b + 3, w4 - w3);
                RC_ADD_STAMP(b + 4, w5 - w4); RC_ADD_STAMP(b + 5, w6 - w5); RC_ADD_STAMP(b + 6, w7 - w6); RC_ADD_STAMP(b + 7, 1);
            }
#endif
        }
        cur ^= 1;
    }
    const float inv_C = 1.0f / C_acc;
    float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int bt = NB * bg + b;
            const int col = (bt >> 1) * H + 64 * half + 32 * (bt & 1) + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (2 * ag + a2) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, R[NB * a2 + b][q] * inv_C);
            }
        }
    // bias gradients: rows of one column group sit 16 lanes apart in a wave; then across the eight waves through LDS
    __syncthreads();                                        // (every wave is done with the tile buffers)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = cs[s][u];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) bsum[(wv * 16 + lane) * 17 + 4 * s + u] = v;
        }
    __syncthreads();
    for (int idx = tid; idx < 16 * 16; idx += 512) {
        const int cg = idx >> 4, k = idx & 15, s = k >> 2;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += bsum[(w * 16 + cg) * 17 + k];
        const int col = 64 * half + 4 * cg + (k & 3);
        if (s < 2) {                                       // dar, daz: the r, z blocks of both bias gradients
            atomicAdd(db_ih + s * H + col, v);
            atomicAdd(db_hh + s * H + col, v);
        } else if (s == 2) {                               // dan
            atomicAdd(db_ih + 2 * H + col, v);
        } else {                                           // dnh
            atomicAdd(db_hh + 2 * H + col, v);
        }
    }
}

// one float per slice, then the step images of the dm | dh kernel
size_t gru_bwd_rc_workspace_bytes(int H) { return 64 + (size_t)(H / 128) * (H / 16) * R_STEP; }
bool gru_bwd_rc_covers(int H) { return H == 128; }

template <bool HAS_MASK, bool NORM_OUT, bool NORM_IN>
static int launch_rc_128(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                         const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh,
                         float* db_ih, float* db_hh, void* workspace, int64_t V, const float* out_norm_k, double* in_norm_sums,
                         const float* in_norm_raw, hipStream_t s) {
    constexpr int H = 128;
    const int64_t tiles = (V + 31) / 32;
    char* dxw = (char*)workspace;
    const size_t lds_dx = (size_t)2 * R_STEP + 4 * (8192 + 2048 + 512 + 4608) + (NORM_OUT ? 3 * H * 4 : 0) + (NORM_IN ? 2 * 128 * 8 : 0);
    const size_t lds_dw = (size_t)2 * 8 * R_IMG + 128;
    static const hipError_t attr_done = [&] {   // once per process and instantiation, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_rc_dx_kernel<H, HAS_MASK, NORM_OUT, NORM_IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dx);
        opt_in_((const void*)gru_rc_dw_kernel<HAS_MASK, NORM_OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);

    hipLaunchKernelGGL(gru_rc_presplit_kernel<H>, dim3((unsigned)(H / 16)), dim3(512), 0, s, W_ih, W_hh, dxw);
    {
        const int64_t rounds = (tiles + 3) / 4;
        int64_t pblocks = 256;                               // one block per CU
        if (pblocks > rounds) pblocks = rounds;
        pblocks = (pblocks + 7) / 8 * 8;
        hipLaunchKernelGGL((gru_rc_dx_kernel<H, HAS_MASK, NORM_OUT, NORM_IN>), dim3((unsigned)pblocks), dim3(512), lds_dx, s,
                           dout, h, mask, saved, out_norm_k, (const char*)dxw, dm, dh, V, in_norm_raw, in_norm_sums);
    }
    int rc = launch_status("mpnn_gru_update_bwd_f32(dm | dh, gate gradients in registers)");
    if (rc) return rc;
    int64_t gx = 128;                                        // tile streams per matrix: one block per CU, whole XCD groups
    while (gx > 8 && gx - 8 >= tiles) gx -= 8;
    hipLaunchKernelGGL((gru_rc_dw_kernel<HAS_MASK, NORM_OUT>), dim3((unsigned)(gx * 2)), dim3(512), lds_dw, s, dout, m, h, mask,
                       saved, out_norm_k, dW_ih, dW_hh, db_ih, db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dW, gate gradients in registers)");
}

#ifdef RC_STAMP
extern "C" int mpnn_debug_rc_stamps(unsigned long long* host64, int reset) {
    if (reset) {
        unsigned long long z[64] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rc_stamps), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_rc_stamps), 64 * sizeof(unsigned long long));
}
#endif

int launch_gru_bwd_rc(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                      const float* W_hh, const float* saved, float* dm, float* dh, float* dW_ih, float* dW_hh, float* db_ih,
                      float* db_hh, void* workspace, int64_t V, int H, const float* out_norm_k, double* in_norm_sums,
                      const float* in_norm_raw, hipStream_t s) {
#define RC_GO(M, NO, NI)                                                                                                  \
    return launch_rc_128<M, NO, NI>(dout, m, h, mask, W_ih, W_hh, saved, dm, dh, dW_ih, dW_hh, db_ih, db_hh, workspace, V,   \
                                    out_norm_k, in_norm_sums, in_norm_raw, s)
    const bool no = out_norm_k != nullptr, ni = in_norm_sums != nullptr;
    if (mask) {
        if (no && ni) RC_GO(true, true, true);
        if (no) RC_GO(true, true, false);
        if (ni) RC_GO(true, false, true);
        RC_GO(true, false, false);
    }
    if (no && ni) RC_GO(false, true, true);
    if (no) RC_GO(false, true, false);
    if (ni) RC_GO(false, false, true);
    RC_GO(false, false, false);
#undef RC_GO
}

}  // namespace mpnn
