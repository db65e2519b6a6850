// GRU forward on the bf16 matrix pipe with 3-way operand splitting ("bf16x6").
//
// Every fp32 operand x is written as h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// three 8-bit pieces = the full 24-bit fp32 mantissa.  A product a*b is then the six partial
// products  ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm  (the dropped am*bl, al*bm, al*bl are
// <= 3 * 2^-24 relative), each EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Accuracy
// is that of an fp32 GEMM with a different summation order (tests: <= 1e-5 abs vs the CPU oracle,
// same bar as the fp32-MFMA kernels), while six bf16 MFMAs of K=16 take 6*32 = 192 matrix-pipe
// cycles against 8*64 = 512 for v_mfma_f32_32x32x2_f32: 2.7x less time on the unit that bounds
// this kernel (12*H^2 flops per atom against 12*H bytes).
//
// Structure = gru_update_resident_kernel: persistent waves, weight slices resident in LDS, the
// A fragments (a lane's own contiguous half-row of m / h) gathered straight into registers, no
// barrier in the loop.  Differences:
//   * weights are split ONCE per block into three bf16 images  W{h,m,l}[matrix][col][k]  (col-major,
//     k contiguous: a B fragment of one piece = one ds_read_b128), 16-byte chunks XOR-swizzled by the
//     column so the 16-lane read groups are conflict-free;
//   * activations are split in registers right before use (8 floats -> 3 x bf16x8 per K=16 step).
#include <stdlib.h>

#include <type_traits>
#include "split_math.h"

namespace mpnn {

__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}

template <int H>
__device__ __forceinline__ int col_swizzle(int col) {
    // rows are H bf16 = H/2 dwords: 32 dwords (H=64) alternate between the two halves of the 64 banks,
    // 64 dwords (H=128) all start on bank 0
    return H == 64 ? ((col >> 1) & 7) : (col & 15);
}

// ---- "fp16x3" (F16 = true): two fp16 pieces per operand and three MFMAs per product instead of three bf16 pieces and six.
// fp16 has 11 significant bits but only 5 exponent bits, so the operands are range-guarded by exact power-of-two scales:
// one for the weight images of a block (its largest |w| lands in [2^14, 2^15)), one per 32-atom tile for its m and h
// rows together (they share the r / z accumulators); x*s = hi + lo, hi = fp16(x*s), lo = fp16(x*s - hi).  Entries more
// than 2^18 below their tile's largest lose relative (not absolute) accuracy: the error of a gate pre-activation is
// ~2^-22 of (tile max) x (weight max) per term, as for a GEMM.  The scales are undone in the epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pow2_scale_of(float maxabs, float& scale, float& inv) {
    int e = (__float_as_int(maxabs) >> 23) & 0xff;
    e = e < 20 ? 20 : e;
    scale = __int_as_float((268 - e) << 23);
    inv = __int_as_float((e - 14) << 23);
}

__device__ __forceinline__ void split8_f16(const f32x4& x0, const f32x4& x1, float sc, f16x8& ph, f16x8& pl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x0[j] * sc, b = x1[j] * sc;
        ph[j] = (_Float16)a;
        pl[j] = (_Float16)(a - (float)ph[j]);
        ph[4 + j] = (_Float16)b;
        pl[4 + j] = (_Float16)(b - (float)ph[4 + j]);
    }
}

// three partial products of one K=16 step, two independent accumulators that share the A pieces, small terms first
__device__ __forceinline__ void mma3x2_a(f32x16& c0, f32x16& c1, const f16x8& ah, const f16x8& al, const f16x8& b0h,
                                         const f16x8& b0l, const f16x8& b1h, const f16x8& b1l) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, c1, 0, 0, 0);
}

template <int H, int NCS, int NW, bool HAS_MASK, bool SAVE, bool F16 = false>
__global__ void __launch_bounds__(64 * NW) gru_update_split_kernel(
    const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask,
    const float* __restrict__ W_ih, const float* __restrict__ W_hh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ saved, int64_t V, int slices) {
    constexpr int CS = 32 * NCS;
    constexpr int NCOL = 3 * CS;               // weight columns held by this block (r | z | n slices)
    constexpr int ROWB = 2 * H;                // bytes per column image row (H bf16)
    constexpr int NCH = H / 8;                 // 16-byte chunks per row
    constexpr int NF4 = H / 8;                 // float4 fragments per lane per operand
    constexpr int STEPS = H / 16;              // K=16 steps per operand (each lane half covers H/2)
    constexpr int IMG = NCOL * ROWB;           // bytes of one (matrix, piece) image
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 matrices][3 pieces][NCOL][H] bf16

    const int slice = blockIdx.x % slices;
    const int pblock = blockIdx.x / slices, pblocks = gridDim.x / slices;
    const int c0 = slice * CS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // ---- one-time: split this block's weight slices into the three bf16 images (F16: two fp16 images, one scale) ----
    float w_sc = 1.0f, w_inv = 1.0f;
    if (F16) {
        float mx = 0.f;
        for (int idx = tid; idx < 2 * H * (NCOL / 4); idx += 64 * NW) {
            const int mat = idx / (H * (NCOL / 4));
            const int rem = idx % (H * (NCOL / 4));
            const int k = rem / (NCOL / 4), q = rem % (NCOL / 4);
            const int g = (4 * q) / CS, cc = (4 * q) % CS;
            const float* W = mat == 0 ? W_ih : W_hh;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)k * 3 * H + g * H + c0 + cc);
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(w4[j]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float* redw = reinterpret_cast<float*>(smem);                 // scratch: the images are written after the barrier
        if (lane == 0) redw[wv] = mx;
        __syncthreads();
        mx = 0.f;
        for (int u = 0; u < NW; ++u) mx = fmaxf(mx, redw[u]);
        __syncthreads();
        pow2_scale_of(mx, w_sc, w_inv);
    }
    for (int idx = tid; idx < 2 * H * (NCOL / 4); idx += 64 * NW) {
        const int mat = idx / (H * (NCOL / 4));
        const int rem = idx % (H * (NCOL / 4));
        const int k = rem / (NCOL / 4), q = rem % (NCOL / 4);
        const int g = (4 * q) / CS, cc = (4 * q) % CS;            // 4 consecutive columns of one gate
        const float* W = mat == 0 ? W_ih : W_hh;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)k * 3 * H + g * H + c0 + cc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = g * CS + cc + j;
            const int off = col * ROWB + (((k >> 3) ^ col_swizzle<H>(col)) << 4) + ((k & 7) << 1);
            if (F16) {
                const float a = w4[j] * w_sc;
                const _Float16 ph = (_Float16)a;
                const _Float16 pl = (_Float16)(a - (float)ph);
                *reinterpret_cast<_Float16*>(smem + (mat * 3 + 0) * IMG + off) = ph;
                *reinterpret_cast<_Float16*>(smem + (mat * 3 + 1) * IMG + off) = pl;
            } else {
                __bf16 ph, pm, pl;
                split3(w4[j], ph, pm, pl);
                *reinterpret_cast<__bf16*>(smem + (mat * 3 + 0) * IMG + off) = ph;
                *reinterpret_cast<__bf16*>(smem + (mat * 3 + 1) * IMG + off) = pm;
                *reinterpret_cast<__bf16*>(smem + (mat * 3 + 2) * IMG + off) = pl;
            }
        }
    }
    __syncthreads();

    const int r = lane & 31, hi = lane >> 5;
    // gate biases of this block's columns live in LDS ([r | z | n_i | n_h][CS]) and are read at the start of every
    // epilogue: as loop-invariant registers they are the first thing the allocator spills
    float* bias_lds = reinterpret_cast<float*>(smem + 6 * IMG);
    for (int idx = tid; idx < CS; idx += 64 * NW) {
        const int col = c0 + idx;
        bias_lds[idx] = b_ih[col] + b_hh[col];
        bias_lds[CS + idx] = b_ih[H + col] + b_hh[H + col];
        bias_lds[2 * CS + idx] = b_ih[2 * H + col];
        bias_lds[3 * CS + idx] = b_hh[2 * H + col];
    }
    __syncthreads();

    const int64_t tiles = (V + 31) / 32;
    const int64_t stride = (int64_t)pblocks * NW;
    // the wave index is uniform: as a scalar it keeps the tile counter and every row base pointer in SGPRs
    int64_t t = (int64_t)pblock * NW + __builtin_amdgcn_readfirstlane(wv);
    if (t >= tiles) return;

    f32x4 fa[NF4], fb[NF4];
    auto load_part = [&](const float* __restrict__ X, int64_t tile, f32x4 (&f)[NF4], int q0, int q1) {
        const int64_t left = V - tile * 32;                      // rows this tile really has (scalar)
        const int rr = left >= 32 ? r : (r < (int)left ? r : (int)left - 1);
        const float* p = X + tile * 32 * H;                      // scalar base + 32-bit lane offset
        const unsigned off = (unsigned)(rr * H + hi * (H / 2));
#pragma unroll
        for (int q = q0; q < q1; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + off + 4 * q);
    };
    auto load_frags = [&](const float* __restrict__ X, int64_t tile, f32x4 (&f)[NF4]) { load_part(X, tile, f, 0, NF4); };
    // B fragment of (matrix, piece) for weight column `col`, K step `s`
    auto bfrag = [&](int mat, int piece, int col, int s) {
        const int chunk = hi * (NCH / 2) + s;
        return *reinterpret_cast<const bf16x8*>(smem + (mat * 3 + piece) * IMG + col * ROWB +
                                                ((chunk ^ col_swizzle<H>(col)) << 4));
    };

    auto bfrag16 = [&](int mat, int piece, int col, int s) {
        const int chunk = hi * (NCH / 2) + s;
        return *reinterpret_cast<const f16x8*>(smem + (mat * 3 + piece) * IMG + col * ROWB +
                                               ((chunk ^ col_swizzle<H>(col)) << 4));
    };
    load_frags(m, t, fa);
    for (; t < tiles; t += stride) {
        f32x16 acc_r[NCS], acc_z[NCS], acc_ni[NCS], acc_nh[NCS];
#pragma unroll
        for (int s = 0; s < NCS; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_r[s][i] = 0.f; acc_z[s][i] = 0.f; acc_ni[s][i] = 0.f; acc_nh[s][i] = 0.f; }

        // F16: every atom's m row and h row get their own power-of-two scale (a row scale factors out of the product; the
        // whole row is in the registers of its two lanes).  r and z sum an m product and an h product: their accumulators
        // are rescaled row by row from the m scale to the h scale between the two phases (exact: powers of two).
        float m_inv = 1.0f, h_inv = 1.0f;
        auto row_scale = [&](const f32x4 (&f)[NF4], float& sc, float& inv) {
            float mx = 0.f;
#pragma unroll
            for (int q = 0; q < NF4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(f[q][u]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));          // the other half of the row
            int e = (__float_as_int(mx) >> 23) & 0xff;
            e = e < 87 ? 87 : e;
            sc = __int_as_float((268 - e) << 23);        // row maximum in [2^14, 2^15)
            inv = __int_as_float((e - 14) << 23);
        };
        load_frags(h, t, fb);                            // in flight while the m-products run
        __builtin_amdgcn_sched_barrier(0);
        if (F16) {
            static_assert(!F16 || NCS == 2, "the fp16 variant is written for two column slices per gate");
            float m_sc;
            row_scale(fa, m_sc, m_inv);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                f16x8 ah, al;
                split8_f16(fa[2 * st], fa[2 * st + 1], m_sc, ah, al);
                auto pair = [&](f32x16 (&acc)[NCS], int g) {
                    const int c0_ = g * CS + r, c1_ = g * CS + 32 + r;
                    mma3x2_a(acc[0], acc[NCS - 1], ah, al, bfrag16(0, 0, c0_, st), bfrag16(0, 1, c0_, st),
                             bfrag16(0, 0, c1_, st), bfrag16(0, 1, c1_, st));
                };
                pair(acc_r, 0); pair(acc_z, 1); pair(acc_ni, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            float h_sc;
            row_scale(fb, h_sc, h_inv);
            {
                const float ratio = h_sc * m_inv;        // lane j (< 32): factor for the tile's row j
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dr = 8 * (i >> 2) + (i & 3);
                    const float f_lo = readlane_f(ratio, dr), f_hi = readlane_f(ratio, 4 + dr);
                    const float f = hi ? f_hi : f_lo;
#pragma unroll
                    for (int s2 = 0; s2 < NCS; ++s2) { acc_r[s2][i] *= f; acc_z[s2][i] *= f; }
                }
            }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                f16x8 ah, al;
                split8_f16(fb[2 * st], fb[2 * st + 1], h_sc, ah, al);
                auto pair = [&](f32x16 (&acc)[NCS], int g) {
                    const int c0_ = g * CS + r, c1_ = g * CS + 32 + r;
                    mma3x2_a(acc[0], acc[NCS - 1], ah, al, bfrag16(1, 0, c0_, st), bfrag16(1, 1, c0_, st),
                             bfrag16(1, 0, c1_, st), bfrag16(1, 1, c1_, st));
                };
                pair(acc_r, 0); pair(acc_z, 1); pair(acc_nh, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int st = 0; st < (F16 ? 0 : STEPS); ++st) {
            bf16x8 ah, am, al;
            split8(fa[2 * st], fa[2 * st + 1], ah, am, al);
            if (NCS == 2) {
                // the two slices of a gate are independent accumulators: issued alternately (split_math.h)
                auto pair = [&](f32x16 (&acc)[NCS], int g) {
                    const int c0_ = g * CS + r, c1_ = g * CS + 32 + r;
                    mma6x2_a(acc[0], acc[NCS - 1], ah, am, al, bfrag(0, 0, c0_, st), bfrag(0, 1, c0_, st), bfrag(0, 2, c0_, st),
                             bfrag(0, 0, c1_, st), bfrag(0, 1, c1_, st), bfrag(0, 2, c1_, st));
                };
                pair(acc_r, 0); pair(acc_z, 1); pair(acc_ni, 2);
            } else {
#pragma unroll
            for (int s = 0; s < NCS; ++s) {
                const int cr = 32 * s + r;
                mma6(acc_r[s], ah, am, al, bfrag(0, 0, cr, st), bfrag(0, 1, cr, st), bfrag(0, 2, cr, st));
                mma6(acc_z[s], ah, am, al, bfrag(0, 0, CS + cr, st), bfrag(0, 1, CS + cr, st), bfrag(0, 2, CS + cr, st));
                mma6(acc_ni[s], ah, am, al, bfrag(0, 0, 2 * CS + cr, st), bfrag(0, 1, 2 * CS + cr, st),
                     bfrag(0, 2, 2 * CS + cr, st));
            }
            }
            __builtin_amdgcn_sched_barrier(0);           // bounds how far ahead weight fragments are read (registers)
        }
#pragma unroll
        for (int st = 0; st < (F16 ? 0 : STEPS); ++st) {
            bf16x8 ah, am, al;
            split8(fb[2 * st], fb[2 * st + 1], ah, am, al);
            if (NCS == 2) {
                auto pair = [&](f32x16 (&acc)[NCS], int g) {
                    const int c0_ = g * CS + r, c1_ = g * CS + 32 + r;
                    mma6x2_a(acc[0], acc[NCS - 1], ah, am, al, bfrag(1, 0, c0_, st), bfrag(1, 1, c0_, st), bfrag(1, 2, c0_, st),
                             bfrag(1, 0, c1_, st), bfrag(1, 1, c1_, st), bfrag(1, 2, c1_, st));
                };
                pair(acc_r, 0); pair(acc_z, 1); pair(acc_nh, 2);
            } else {
#pragma unroll
            for (int s = 0; s < NCS; ++s) {
                const int cr = 32 * s + r;
                mma6(acc_r[s], ah, am, al, bfrag(1, 0, cr, st), bfrag(1, 1, cr, st), bfrag(1, 2, cr, st));
                mma6(acc_z[s], ah, am, al, bfrag(1, 0, CS + cr, st), bfrag(1, 1, CS + cr, st), bfrag(1, 2, CS + cr, st));
                mma6(acc_nh[s], ah, am, al, bfrag(1, 0, 2 * CS + cr, st), bfrag(1, 1, 2 * CS + cr, st),
                     bfrag(1, 2, 2 * CS + cr, st));
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue.  Every load of the tile (its h values in accumulator layout, one mask value per lane) is requested
        // before the first store: vmcnt retires in issue order and counts a store until L2 acknowledges it, so a load
        // issued behind a group of stores would wait for every one of them.  The registers of the h fragments are free
        // by now.  A full tile addresses everything as one base + compile-time offsets; the last, ragged tile clamps
        // its loads and predicates its stores.
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            __builtin_amdgcn_sched_barrier(0);                       // keep these loads below the last MFMA phase
            int zero = 0;
            asm volatile("" : "+v"(zero));                           // opaque: keeps the bias reads inside the loop
            float br[NCS], bz[NCS], bni[NCS], bnh[NCS];
#pragma unroll
            for (int s = 0; s < NCS; ++s) {
                br[s] = bias_lds[zero + 32 * s + r];
                bz[s] = bias_lds[zero + CS + 32 * s + r];
                bni[s] = bias_lds[zero + 2 * CS + 32 * s + r];
                bnh[s] = bias_lds[zero + 3 * CS + 32 * s + r];
            }
            const int64_t row0 = t * 32 + 4 * hi;                    // + 8*(i>>2) + (i&3)
            const unsigned eo = (unsigned)(4 * hi * H + c0 + r);
            const float* hb = h + t * 32 * H + eo;                   // scalar tile base + lane offset; the rest is immediates
            float hv[16][NCS];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int dr = 8 * (i >> 2) + (i & 3);
#pragma unroll
                for (int s = 0; s < NCS; ++s)
                    if (FULL) hv[i][s] = hb[dr * H + 32 * s];        // (the ragged tile reads its h values row by row)
            }
            float mkl = 1.0f;
            if (HAS_MASK) {
                const int left = FULL ? 32 : (int)(V - t * 32);
                mkl = (mask + t * 32)[(unsigned)(r < left ? r : left - 1)];   // lane j (< 32): mask of the tile's row j
            }
            // next tile's m rows: behind the epilogue loads, ahead of the stores (both operand fragments are dead here);
            // the ragged tile is the last one
            // (unconditional, tile index clamped: under a condition the old fragments would stay live as the other arm)
            if (FULL) {
                load_frags(m, t + stride < tiles ? t + stride : t, fa);
            } else {
#pragma unroll
                for (int q = 0; q < NF4; ++q) fa[q] = f32x4{0.f, 0.f, 0.f, 0.f};   // last tile: defined, never used
            }
            __builtin_amdgcn_sched_barrier(0);
            float* ob = out + t * 32 * H + eo;
            float* sb = saved + t * 32 * 4 * H + (unsigned)(4 * hi * 4 * H + c0 + r);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int dr = 8 * (i >> 2) + (i & 3);
                // mask of row 4*hi + dr: two scalar lane reads and a select (a shuffle would keep 16 lane indices live)
                float mk = 1.0f;
                if (HAS_MASK) {
                    const float mk_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mkl), dr));
                    const float mk_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mkl), 4 + dr));
                    mk = hi ? mk_hi : mk_lo;
                }
                float un_m = 1.0f, un_h = 1.0f;                    // F16: undo the row's and the weights' scales
                if (F16) {
                    const float a_lo = readlane_f(m_inv, dr), a_hi = readlane_f(m_inv, 4 + dr);
                    const float b_lo = readlane_f(h_inv, dr), b_hi = readlane_f(h_inv, 4 + dr);
                    un_m = (hi ? a_hi : a_lo) * w_inv;
                    un_h = (hi ? b_hi : b_lo) * w_inv;
                }
#pragma unroll
                for (int s = 0; s < NCS; ++s) {
                    const float rg = sigmoid_fast((F16 ? acc_r[s][i] * un_h : acc_r[s][i]) + br[s]) * mk;
                    const float zg = sigmoid_fast((F16 ? acc_z[s][i] * un_h : acc_z[s][i]) + bz[s]) * mk;
                    const float nh = (F16 ? acc_nh[s][i] * un_h : acc_nh[s][i]) + bnh[s];
                    const float ng = tanh_fast((F16 ? acc_ni[s][i] * un_m : acc_ni[s][i]) + bni[s] + rg * nh) * mk;
                    const float hval = FULL ? hv[i][s] : (row0 + dr < V ? hb[dr * H + 32 * s] : 0.f);
                    const float o = ((1.0f - zg) * ng + zg * hval) * mk;
                    if (FULL || row0 + dr < V) {
                        __builtin_nontemporal_store(o, ob + dr * H + 32 * s);
                        if (SAVE) {                                  // 16*H bytes per atom, read back once by the backward
                            float* sv = sb + dr * 4 * H + 32 * s;
                            __builtin_nontemporal_store(rg, sv);
                            __builtin_nontemporal_store(zg, sv + H);
                            __builtin_nontemporal_store(ng, sv + 2 * H);
                            __builtin_nontemporal_store(nh, sv + 3 * H);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                   // row by row: nothing is gained by hoisting gate math
            }
        };
        if (t * 32 + 32 <= V) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
}

template <int H, int NCS, int NW>
static int launch_split(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                        const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, hipStream_t s) {
    constexpr int CS = 32 * NCS;
    constexpr int slices = H / CS;
    const size_t lds = (size_t)2 * 3 * (3 * CS) * (2 * H) + 16 * CS;    // weight images + gate biases
    // two row-guarded fp16 pieces per operand, three MFMAs per product (the F16 = false arm of the kernel template, three
    // bf16 pieces and six MFMAs, was the default of round 1 and is no longer instantiated)
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        const int n = (int)lds;
        opt_in_((const void*)gru_update_split_kernel<H, NCS, NW, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
        opt_in_((const void*)gru_update_split_kernel<H, NCS, NW, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
        opt_in_((const void*)gru_update_split_kernel<H, NCS, NW, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
        opt_in_((const void*)gru_update_split_kernel<H, NCS, NW, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t tiles = (V + 31) / 32;
    int64_t pblocks = (256 + slices - 1) / slices;        // one block per CU (144 KB of LDS)
    if (pblocks * NW > tiles) pblocks = (tiles + NW - 1) / NW;
    if (pblocks < 1) pblocks = 1;
    const dim3 grid((unsigned)(pblocks * slices)), block(64 * NW);
#define MPNN_LAUNCH_F16(MASKED, SAVED)                                                                                     \
    hipLaunchKernelGGL((gru_update_split_kernel<H, NCS, NW, MASKED, SAVED, true>), grid, block, lds, s, m, h, mask, W_ih, W_hh, \
                       b_ih, b_hh, out, saved, V, slices)
    if (mask && saved) MPNN_LAUNCH_F16(true, true);
    else if (mask) MPNN_LAUNCH_F16(true, false);
    else if (saved) MPNN_LAUNCH_F16(false, true);
    else MPNN_LAUNCH_F16(false, false);
#undef MPNN_LAUNCH_F16
    return launch_status("mpnn_gru_update_f32(fp16x3, row guards)");
}

// ---------------------------------------------------------------------------------------------------------------
// H = 128 / 256, weights STREAMED.  Neither matrix's split images fit in LDS next to anything else (590 KB as three bf16
// images at H = 128), so a block owns a 64-feature output slice (192 gate columns of both matrices) and the contraction
// is cut into 32-wide K chunks whose weight images pass through a double-buffered LDS buffer, one barrier per chunk.
// Accumulators r, z, gi_n, gh_n stay in registers across the whole contraction; operand rows arrive as 16-float pieces
// one half-chunk ahead.  Row operands are read H / 64 times (2 at H = 128, 4 at H = 256); the weights come out of L2.
// (Rounds 1-2 also had a resident-slice kernel at H = 128 and a 32 x 32 wave tile at H = 256: 1.5 % / 15 % behind this
// kernel, removed in round 3.)
// ---- pre-split weights for the streamed wide kernel (fp16 pieces), one workspace per launch ----
// [0, 64): inverse weight scale of slice s at float s; then the LDS image of (slice, chunk c) at 64 + (slice * H / 32 + c) * 48 KB:
// [matrix][piece][192 columns = (gate, column of the slice)][32 k] exactly as gru_update_stream_wide_kernel reads it, so a
// chunk is copied global -> LDS verbatim (global_load_lds_dwordx4), with no vector work and no registers in the kernel.
template <int H>
__global__ void __launch_bounds__(512) gru_fwd_presplit_kernel(const float* __restrict__ W_ih, const float* __restrict__ W_hh,
                                                               char* __restrict__ ws) {
    constexpr int NCHUNK = H / 32, COLS = 192, IMGC = COLS * 64;
    __shared__ float redw[8];
    const int slice = blockIdx.x / NCHUNK, c = blockIdx.x % NCHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = 0.f;
    for (int idx = tid; idx < 2 * H * 48; idx += 512) {
        const int mat = idx / (H * 48), rem = idx % (H * 48);
        const int kk = rem / 48, q = rem % 48;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>((mat ? W_hh : W_ih) + (int64_t)kk * 3 * H + (q / 16) * H +
                                                         64 * slice + 4 * (q % 16));
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(w4[u]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) redw[wv] = mx;
    __syncthreads();
    mx = redw[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) mx = fmaxf(mx, redw[u]);
    float w_sc, w_inv;
    pow2_scale_of(mx, w_sc, w_inv);
    if (c == 0 && tid == 0) reinterpret_cast<float*>(ws)[slice] = w_inv;
    char* img = ws + 64 + (int64_t)(slice * NCHUNK + c) * (4 * IMGC);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = tid + 512 * j;
        const int mat = u / 768, rem = u % 768;
        const int o = rem / COLS, cl = rem % COLS;
        const float* W = (mat ? W_hh : W_ih) + (int64_t)(32 * c + 8 * o) * 3 * H + (cl / 64) * H + 64 * slice + (cl % 64);
        f32x4 x0, x1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] = W[(int64_t)i * 3 * H]; x1[i] = W[(int64_t)(4 + i) * 3 * H]; }
        f16x8 ph, pl;
        split8_f16(x0, x1, w_sc, ph, pl);
        char* base = img + mat * 2 * IMGC + cl * 64 + ((o ^ ((cl >> 2) & 3)) << 4);
        *reinterpret_cast<f16x8*>(base) = ph;
        *reinterpret_cast<f16x8*>(base + IMGC) = pl;
    }
}

// one wave copies 1 KB global -> LDS (lane L's 16 bytes land at lds_dst + 16 L); inline assembly so that the compiler
// neither drains the copy before the next LDS read nor counts it (gru_bwd128_f16.hip has the details)
__device__ __forceinline__ void w_copy_to_lds(const char* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

// NORM (the masked batch norm of models/mask_batch_norm.py:5-38 fused into the update): `h` is the RAW
// output of the previous update; the state the reference would have normalised in a pass of its own is
// hn = (h * hs[col] + ht[col]) * mask, formed here where the epilogue reads h anyway (the contraction sees it through
// weights and biases the caller folded: W_hh' = diag(hs) W_hh, b_hh' = b_hh + ht W_hh).  The column sums of `out` and of
// its squares over the rows written -- the moments the NEXT norm needs -- leave through `stats` (2 H doubles, the caller
// zeroes them), and with SAVE `hn` is written for the backward pass.
struct GruNormArgs {
    const float* hs;     // (H) scale of h
    const float* ht;     // (H) shift of h
    float* hnorm;        // (V, H) normalised state, written with SAVE
    double* stats;       // [0, H) sum of out, [H, 2H) sum of out^2
};

template <int H, bool HAS_MASK, bool SAVE, bool F16 = false, bool WS = false, bool NORM = false>
__global__ void __launch_bounds__(512) gru_update_stream_wide_kernel(
    const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask,
    const float* __restrict__ W_ih, const float* __restrict__ W_hh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ saved, int64_t V,
    const char* __restrict__ wws, GruNormArgs na) {
    static_assert(!WS || F16, "pre-split weights are fp16 pieces");
    static_assert(!NORM || WS, "the fused norm rides on the pre-split kernel");
    constexpr int NS = H / 64, NCHUNK = H / 32, COLS = 192;
    constexpr int IMGC = COLS * 64;            // bytes of one (matrix, piece) chunk image: 192 columns x 32 k of 16 bits
    constexpr int NP = F16 ? 2 : 3;            // pieces per operand
    constexpr int BUF = 2 * NP * IMGC;         // 72 KB (bf16x6) / 48 KB (fp16x3)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][2 matrices][NP pieces][192][32]
    __shared__ float redw[8];
    __shared__ float bias_s[4][64];            // the slice's gate biases (b_r, b_z, b_in, b_hn): read per tile from LDS --
                                               // as per-lane global pointers they were loop invariants that got spilled
    __shared__ float norm_s[NORM ? 2 : 1][64];            // NORM: the slice's hs | ht
    __shared__ double stat_s[NORM ? 8 : 1][2][64];        // NORM: per wave, column sums of out | out^2 over its tiles

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % NS;                                 // the NS slice blocks of a row group share an XCD
    const int pblock = (jb / NS) * 8 + xcd, pblocks = gridDim.x / NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;

    const int64_t rounds_total = (V + 255) / 256;              // a round = 256 rows: every wave its own 32-row tile
    if (pblock >= rounds_total) return;                        // block-uniform
    const int64_t nrounds = (rounds_total - pblock + pblocks - 1) / pblocks;
    if (tid < 64) {
        const int fc = 64 * slice + tid;
        bias_s[0][tid] = b_ih[fc] + b_hh[fc];
        bias_s[1][tid] = b_ih[H + fc] + b_hh[H + fc];
        bias_s[2][tid] = b_ih[2 * H + fc];
        bias_s[3][tid] = b_hh[2 * H + fc];
        if (NORM) {
            norm_s[0][tid] = na.hs[fc];
            norm_s[1][tid] = na.ht[fc];
        }
    }                                                          // (the first chunk's barrier publishes them)
    if (NORM)
        for (int i = tid; i < 8 * 2 * 64; i += 512) (&stat_s[0][0][0])[i] = 0.0;

    // F16: one power-of-two scale for the block's weights (its 64 features x 3 gates of both matrices land below 2^15)
    float w_sc = 1.0f, w_inv = 1.0f;
    if (WS) {
        w_inv = reinterpret_cast<const float*>(wws)[slice];
    } else if (F16) {
        float mx = 0.f;
        for (int idx = tid; idx < 2 * H * 48; idx += 512) {    // (matrix, k, 48 float4 of the slice's three gate blocks)
            const int mat = idx / (H * 48), rem = idx % (H * 48);
            const int kk = rem / 48, q = rem % 48;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>((mat ? W_hh : W_ih) + (int64_t)kk * 3 * H + (q / 16) * H +
                                                             64 * slice + 4 * (q % 16));
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(w4[u]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) redw[wv] = mx;
        __syncthreads();
        mx = redw[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) mx = fmaxf(mx, redw[u]);
        pow2_scale_of(mx, w_sc, w_inv);
    }
    // ---- weight staging: unit = (matrix, k-octet of the chunk, column); 1536 units, three per thread ----
    // per-thread constants of its three units: source pointer at chunk 0 and LDS byte offset
    const float* wsrc[3];
    int ldst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = tid + 512 * j;
        const int mat = u / 768, rem = u % 768;
        const int o = rem / COLS, cl = rem % COLS;
        wsrc[j] = (mat ? W_hh : W_ih) + (int64_t)(8 * o) * 3 * H + (cl / 64) * H + 64 * slice + (cl % 64);
        ldst[j] = mat * NP * IMGC + cl * 64 + ((o ^ ((cl >> 2) & 3)) << 4);
    }
    // one staging unit at a time (8 registers live): loaded before a third of the chunk's MFMAs, written after it
    float raw[8];
    auto stage_load1 = [&](int c, int j) {
        const float* W = wsrc[j] + (int64_t)(32 * c) * 3 * H;
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = W[(int64_t)i * 3 * H];
    };
    auto stage_write1 = [&](int buf, int j) {
        const f32x4 x0 = {raw[0], raw[1], raw[2], raw[3]};
        const f32x4 x1 = {raw[4], raw[5], raw[6], raw[7]};
        char* base = smem + buf * BUF + ldst[j];
        if (F16) {
            f16x8 ph, pl;
            split8_f16(x0, x1, w_sc, ph, pl);
            *reinterpret_cast<f16x8*>(base) = ph;
            *reinterpret_cast<f16x8*>(base + IMGC) = pl;
        } else {
            bf16x8 ph, pm, pl;
            split8(x0, x1, ph, pm, pl);
            *reinterpret_cast<bf16x8*>(base) = ph;
            *reinterpret_cast<bf16x8*>(base + IMGC) = pm;
            *reinterpret_cast<bf16x8*>(base + 2 * IMGC) = pl;
        }
    };
    // WS: the chunk's image is copied verbatim from the pre-split workspace, 48 x 1 KB, six per wave
    auto stage_copy = [&](int c, int buf) {
        typedef __attribute__((address_space(3))) const char lds_char;
        const char* src = wws + 64 + (int64_t)(slice * NCHUNK + c) * BUF + lane * 16;
        const unsigned dst = (unsigned)(uintptr_t)(lds_char*)(smem + buf * BUF);
#pragma unroll
        for (int it = 0; it < BUF / 8192; ++it) {
            const int blk = (BUF / 8192) * wv + it;
            w_copy_to_lds(src + blk * 1024, dst + blk * 1024);
        }
    };
    // B fragment: column cl = gate*64 + 32*nb + r, k-octet 2*hi + st of the chunk
    auto bfrag = [&](int buf, int mat, int piece, int gate, int nb, int st) {
        const int cl = gate * 64 + 32 * nb + r;
        const int o = 2 * hi + st;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BUF + (mat * NP + piece) * IMGC + cl * 64 +
                                                ((o ^ ((cl >> 2) & 3)) << 4));
    };
    auto bfrag16 = [&](int buf, int mat, int piece, int gate, int nb, int st) {
        return __builtin_bit_cast(f16x8, bfrag(buf, mat, piece, gate, nb, st));
    };
    // this lane's 16 floats of chunk c of operand X for row tile `tile`
    auto load_rows = [&](const float* __restrict__ X, int64_t tile, int c, f32x4 (&f)[4]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;
#ifdef MPNN_ABL_HOT_ROWS        // timing experiment only: every wave re-reads one L2-resident tile (wrong results)
        row = wv * 32 + r;
#endif
        const float* p = X + row * H + 32 * c + 16 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };

    f32x16 acc_r[2], acc_z[2], acc_ni[2], acc_nh[2];       // 32 rows x 64 features per wave
    f32x4 a0[4], a1[4];                                     // operand half-chunks: one multiplying, one in flight
    // F16: the rows of a tile are range-guarded one by one (a row scale factors out of the product): row_sc = power of two
    // that put the row's largest operand entry seen so far into [2^11, 2^12) when it was chosen -- eight-fold headroom,
    // so later chunks rarely force a change; when one does, the row's accumulator entries are multiplied by the ratio.
    float row_sc = 1.0f, row_inv = 1.0f;
    auto rescale_rows = [&](float ratio) {                  // ratio of lane j (< 32) = factor for the tile's row j
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int dr = 8 * (i >> 2) + (i & 3);
            const float f_lo = readlane_f(ratio, dr), f_hi = readlane_f(ratio, 4 + dr);
            const float f = hi ? f_hi : f_lo;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc_r[nb][i] *= f; acc_z[nb][i] *= f; acc_ni[nb][i] *= f; acc_nh[nb][i] *= f;
            }
        }
    };
    auto guard_rows = [&](int hc, const f32x4 (&x)[4]) {
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fabsf(x[q][u]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));                 // both halves of the row
        const bool grow = hc == 0 || mx * row_sc >= 32768.0f;
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
            int e = (__float_as_int(mx) >> 23) & 0xff;
            e = e < 87 ? 87 : e;
            const float ns = grow ? __int_as_float((265 - e) << 23) : row_sc;    // mx * ns in [2^11, 2^12)
            const float ni = grow ? __int_as_float((e - 11) << 23) : row_inv;
            if (hc != 0) rescale_rows(ns * row_inv);
            row_sc = ns;
            row_inv = ni;
        }
    };
    int cur = 0;
    int64_t tile = ((int64_t)pblock) * 8 + wv;

    // Half-chunk hc of a round: operand hc & 1 (m, h) of K chunk hc >> 1.  While it multiplies (six groups x two
    // column blocks = 72 MFMAs), the next half-chunk's rows are in flight and one or two units of the NEXT chunk's
    // weights are split and parked; the barrier sits at the start of every chunk.
    auto half = [&](int hc, int64_t tile_next, f32x4 (&x)[4], f32x4 (&nx)[4]) {
        const int c = hc >> 1, mat = hc & 1;
        if (mat == 0) __syncthreads();                     // buffer `cur` is complete, `cur ^ 1` is free
#ifndef MPNN_ABL_NO_WCOPY       // timing experiment only: the weight chunks are never refreshed (wrong results)
        if (WS && mat == 0) stage_copy((c + 1) % NCHUNK, cur ^ 1);   // lands while this chunk multiplies
#endif
        const int hn = (hc + 1) % (2 * NCHUNK);
        load_rows((hn & 1) ? h : m, hn == 0 ? tile_next : tile, hn >> 1, nx);
        const int cn = (c + 1) % NCHUNK;
        if (F16) guard_rows(hc, x);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a_h, a_m, a_l;
        f16x8 f_h, f_l;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int st = i / 3, gate = i % 3;
            if (i % 3 == 0) {
                // staging units: 0 and 1 ride on the m half, 2 on the h half
                if (!WS) {
                    if (mat == 0) stage_load1(cn, i / 3);
                    else if (i == 0) stage_load1(cn, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (F16) split8_f16(x[2 * st], x[2 * st + 1], row_sc, f_h, f_l);
                else split8(x[2 * st], x[2 * st + 1], a_h, a_m, a_l);
            }
            {   // the two column blocks are independent accumulators: issued alternately (split_math.h)
                f32x16(&acc)[2] = gate == 0 ? acc_r : gate == 1 ? acc_z : (mat == 0 ? acc_ni : acc_nh);
                if (F16)
                    mma3x2_a(acc[0], acc[1], f_h, f_l, bfrag16(cur, mat, 0, gate, 0, st), bfrag16(cur, mat, 1, gate, 0, st),
                             bfrag16(cur, mat, 0, gate, 1, st), bfrag16(cur, mat, 1, gate, 1, st));
                else
                    mma6x2_a(acc[0], acc[1], a_h, a_m, a_l, bfrag(cur, mat, 0, gate, 0, st), bfrag(cur, mat, 1, gate, 0, st),
                             bfrag(cur, mat, 2, gate, 0, st), bfrag(cur, mat, 0, gate, 1, st), bfrag(cur, mat, 1, gate, 1, st),
                             bfrag(cur, mat, 2, gate, 1, st));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!WS && i % 3 == 2) {
                if (mat == 0) stage_write1(cur ^ 1, i / 3);
                else if (i == 2) stage_write1(cur ^ 1, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (mat == 1) {
            if (WS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of the next chunk's image has landed
            cur ^= 1;
        }
    };

    if (WS) {
        stage_copy(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            stage_load1(0, j);
            stage_write1(0, j);
        }
    }
    load_rows(m, tile, 0, a0);
    for (int64_t rd = 0; rd < nrounds; ++rd) {
        const int64_t tile_next = rd + 1 < nrounds ? tile + (int64_t)pblocks * 8 : tile;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { acc_r[nb][i] = 0.f; acc_z[nb][i] = 0.f; acc_ni[nb][i] = 0.f; acc_nh[nb][i] = 0.f; }
#pragma unroll 1
        for (int hc = 0; hc < 2 * NCHUNK; hc += 2) {
            half(hc, tile_next, a0, a1);
            half(hc + 1, tile_next, a1, a0);
        }
        // epilogue: every load of the tile (h values in accumulator layout, one mask value per lane) goes out before
        // the first store -- vmcnt retires in order and counts a store until L2 acknowledges it, so a load issued
        // behind stores waits for all of them.  A full tile is one scalar base + lane offset + immediates; the last,
        // ragged tile reads row by row and predicates its stores.
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            __builtin_amdgcn_sched_barrier(0);
            // the lane offsets are recomputed per tile behind an opaque move: as loop invariants the compiler folded them
            // into per-lane 64-bit pointers (h + offset, out + offset, mask + lane, ...) that it kept across the K loop and
            // spilled -- and a reload in here is a scratch access whose s_waitcnt vmcnt(0) drains every load just issued
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const unsigned lr = ln & 31u, h4 = (ln >> 5) << 2;
            const unsigned eo = h4 * H + 64 * slice + lr, so = h4 * 4 * H + 64 * slice + lr;
            const int64_t row0 = tile * 32 + h4;                       // + 8*(i>>2) + (i&3)
            const float* hb = h + tile * 32 * H + eo;
            float hv[2][16];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (FULL) hv[nb][i] = hb[(8 * (i >> 2) + (i & 3)) * H + 32 * nb];
            float mkl = 1.0f;
            if (HAS_MASK) {
                const int left = FULL ? 32 : (int)(V - tile * 32);
                mkl = (mask + tile * 32)[lr < (unsigned)left ? lr : (unsigned)(left - 1)];   // lane j (< 32): mask of the tile's row j
            }
            __builtin_amdgcn_sched_barrier(0);
            float* ob = out + tile * 32 * H + eo;
            float* sb = saved + tile * 32 * 4 * H + so;
            float* nb_out = NORM && SAVE ? na.hnorm + tile * 32 * H + eo : nullptr;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float br = bias_s[0][32 * nb + r], bz = bias_s[1][32 * nb + r];      // this lane's output feature
                const float bni = bias_s[2][32 * nb + r], bnh = bias_s[3][32 * nb + r];
                float hsc = 1.0f, hsh = 0.0f, sum1 = 0.0f, sum2 = 0.0f;
                if (NORM) { hsc = norm_s[0][32 * nb + r]; hsh = norm_s[1][32 * nb + r]; }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dr = 8 * (i >> 2) + (i & 3);
                    float mk = 1.0f;
                    if (HAS_MASK) {
                        const float mk_lo = readlane_f(mkl, dr), mk_hi = readlane_f(mkl, 4 + dr);
                        mk = hi ? mk_hi : mk_lo;
                    }
                    float un = 1.0f;                       // F16: undo the row's and the weights' scale
                    if (F16) {
                        const float u_lo = readlane_f(row_inv, dr), u_hi = readlane_f(row_inv, 4 + dr);
                        un = (hi ? u_hi : u_lo) * w_inv;
                    }
#ifdef MPNN_ABL_NO_EPI_MATH     // timing experiment only: the gate nonlinearities left out (wrong results)
                    const float rg = (acc_r[nb][i] * un + br) * mk;
                    const float zg = (acc_z[nb][i] * un + bz) * mk;
                    const float nh = acc_nh[nb][i] * un + bnh;
                    const float ng = (acc_ni[nb][i] * un + bni + rg * nh) * mk;
#else
                    const float rg = sigmoid_fast(acc_r[nb][i] * un + br) * mk;
                    const float zg = sigmoid_fast(acc_z[nb][i] * un + bz) * mk;
                    const float nh = acc_nh[nb][i] * un + bnh;
                    const float ng = tanh_fast(acc_ni[nb][i] * un + bni + rg * nh) * mk;
#endif
                    float hval = FULL ? hv[nb][i] : (row0 + dr < V ? hb[dr * H + 32 * nb] : 0.f);
                    if (NORM) hval = fmaf(hval, hsc, hsh) * mk;
                    const float o = ((1.0f - zg) * ng + zg * hval) * mk;
                    if (FULL || row0 + dr < V) {
                        __builtin_nontemporal_store(o, ob + dr * H + 32 * nb);
                        if (NORM) {
                            sum1 += o;
                            sum2 = fmaf(o, o, sum2);
                            if (SAVE) __builtin_nontemporal_store(hval, nb_out + dr * H + 32 * nb);
                        }
                        if (SAVE) {
                            float* sv = sb + dr * 4 * H + 32 * nb;
                            __builtin_nontemporal_store(rg, sv);
                            __builtin_nontemporal_store(zg, sv + H);
                            __builtin_nontemporal_store(ng, sv + 2 * H);
                            __builtin_nontemporal_store(nh, sv + 3 * H);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (NORM) {                                // the tile's 32 rows of this column, then the wave's running sum
                    sum1 += __shfl_xor(sum1, 32);
                    sum2 += __shfl_xor(sum2, 32);
                    if (hi == 0) {
                        stat_s[wv][0][32 * nb + r] += (double)sum1;
                        stat_s[wv][1][32 * nb + r] += (double)sum2;
                    }
                }
            }
        };
        // (in the last round a wave's whole tile can lie past V: nothing to do then)
        if (tile * 32 + 32 <= V) epilogue(std::true_type{});
        else if (tile * 32 < V) epilogue(std::false_type{});
        tile = tile_next;
    }
    if (NORM) {
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, cl = tid & 63;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += stat_s[w][k][cl];
            atomicAdd(na.stats + k * H + 64 * slice + cl, t);
        }
    }
}

size_t gru_fwd_workspace_bytes(int H) {
    return (H == 128 || H == 256) ? 64 + (size_t)(H / 64) * (H / 32) * (2 * 4 * 192 * 64 / 2) : 0;
}

template <int H>
static int launch_stream_wide(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                         const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, void* workspace,
                         const GruNormArgs* norm, hipStream_t s) {
    constexpr int NS = H / 64;
    const bool presplit = workspace != nullptr;              // weights split once per launch, copied global -> LDS
    const size_t lds = (size_t)2 * 4 * 192 * 64;             // two fp16 pieces per operand, row-wise range guards
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        const int n16 = (int)2 * 4 * 192 * 64;
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, true, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        opt_in_((const void*)gru_update_stream_wide_kernel<H, false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, n16);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t rounds = (V + 255) / 256;
    int64_t pblocks = 256 / NS;                             // x NS slices = one block per CU (144 KB of LDS)
    if (pblocks > rounds) pblocks = rounds;
    pblocks = (pblocks + 7) / 8 * 8;                        // XCD-aware numbering wants groups of 8 row blocks
    const dim3 grid((unsigned)(pblocks * NS)), block(512);
    const char* wws = (const char*)workspace;
    if (norm && !presplit) return 1;                         // the fused norm exists on the pre-split kernel only
    const GruNormArgs na = norm ? *norm : GruNormArgs{nullptr, nullptr, nullptr, nullptr};
    if (presplit)
        hipLaunchKernelGGL(gru_fwd_presplit_kernel<H>, dim3(NS * (H / 32)), dim3(512), 0, s, W_ih, W_hh, (char*)workspace);
#define MPNN_LAUNCH_WIDE(MASKED, SAVED)                                                                                  \
    do {                                                                                                                 \
        if (norm)                                                                                                        \
            hipLaunchKernelGGL((gru_update_stream_wide_kernel<H, MASKED, SAVED, true, true, true>), grid, block, lds, s, \
                               m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, wws, na);                              \
        else if (presplit)                                                                                               \
            hipLaunchKernelGGL((gru_update_stream_wide_kernel<H, MASKED, SAVED, true, true>), grid, block, lds, s, m, h, \
                               mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, wws, na);                                    \
        else                                                                                                             \
            hipLaunchKernelGGL((gru_update_stream_wide_kernel<H, MASKED, SAVED, true>), grid, block, lds, s, m, h, mask, \
                               W_ih, W_hh, b_ih, b_hh, out, saved, V, wws, na);                                          \
    } while (0)
    if (mask && saved) MPNN_LAUNCH_WIDE(true, true);
    else if (mask) MPNN_LAUNCH_WIDE(true, false);
    else if (saved) MPNN_LAUNCH_WIDE(false, true);
    else MPNN_LAUNCH_WIDE(false, false);
#undef MPNN_LAUNCH_WIDE
    return launch_status("mpnn_gru_update_f32(streamed weights, wide tile)");
}

// returns 1 when the width has no split-precision path
int launch_gru_split(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                     const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, int H, void* workspace,
                     hipStream_t s) {
    if (H == 64) return launch_split<64, 2, 8>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, s);
    if (H == 128) return launch_stream_wide<128>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, workspace, nullptr, s);
    if (H == 256) return launch_stream_wide<256>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, workspace, nullptr, s);
    return 1;
}

// the update with the masked batch norm of its `h` input folded in and the moments of its output taken (NORM above);
// returns 1 when the width has no such kernel
int launch_gru_split_norm(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                          const float* b_ih, const float* b_hh, const float* hs, const float* ht, float* out,
                          float* saved, float* hnorm, double* stats, int64_t V, int H, void* workspace, hipStream_t s) {
    const GruNormArgs na{hs, ht, hnorm, stats};
    if (H == 128) return launch_stream_wide<128>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, workspace, &na, s);
    if (H == 256) return launch_stream_wide<256>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, workspace, &na, s);
    return 1;
}

}  // namespace mpnn
