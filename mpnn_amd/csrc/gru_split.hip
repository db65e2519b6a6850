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

template <int H, int NCS, int NW, bool HAS_MASK>
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

    // ---- one-time: split this block's weight slices into the three bf16 images ----
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
            __bf16 ph, pm, pl;
            split3(w4[j], ph, pm, pl);
            const int off = col * ROWB + (((k >> 3) ^ col_swizzle<H>(col)) << 4) + ((k & 7) << 1);
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 0) * IMG + off) = ph;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 1) * IMG + off) = pm;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 2) * IMG + off) = pl;
        }
    }
    __syncthreads();

    const int r = lane & 31, hi = lane >> 5;
    float br[NCS], bz[NCS], bni[NCS], bnh[NCS];
#pragma unroll
    for (int s = 0; s < NCS; ++s) {
        const int col = c0 + 32 * s + r;
        br[s] = b_ih[col] + b_hh[col];
        bz[s] = b_ih[H + col] + b_hh[H + col];
        bni[s] = b_ih[2 * H + col];
        bnh[s] = b_hh[2 * H + col];
    }

    const int64_t tiles = (V + 31) / 32;
    const int64_t stride = (int64_t)pblocks * NW;
    int64_t t = (int64_t)pblock * NW + wv;
    if (t >= tiles) return;

    f32x4 fa[NF4], fb[NF4];
    auto load_frags = [&](const float* __restrict__ X, int64_t tile, f32x4 (&f)[NF4]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;
        const float* p = X + row * H + hi * (H / 2);
#pragma unroll
        for (int q = 0; q < NF4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    // B fragment of (matrix, piece) for weight column `col`, K step `s`
    auto bfrag = [&](int mat, int piece, int col, int s) {
        const int chunk = hi * (NCH / 2) + s;
        return *reinterpret_cast<const bf16x8*>(smem + (mat * 3 + piece) * IMG + col * ROWB +
                                                ((chunk ^ col_swizzle<H>(col)) << 4));
    };

    load_frags(m, t, fa);
    for (; t < tiles; t += stride) {
        f32x16 acc_r[NCS], acc_z[NCS], acc_ni[NCS], acc_nh[NCS];
#pragma unroll
        for (int s = 0; s < NCS; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_r[s][i] = 0.f; acc_z[s][i] = 0.f; acc_ni[s][i] = 0.f; acc_nh[s][i] = 0.f; }

        load_frags(h, t, fb);                            // in flight while the m-products run
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            bf16x8 ah, am, al;
            split8(fa[2 * st], fa[2 * st + 1], ah, am, al);
#pragma unroll
            for (int s = 0; s < NCS; ++s) {
                const int cr = 32 * s + r;
                mma6(acc_r[s], ah, am, al, bfrag(0, 0, cr, st), bfrag(0, 1, cr, st), bfrag(0, 2, cr, st));
                mma6(acc_z[s], ah, am, al, bfrag(0, 0, CS + cr, st), bfrag(0, 1, CS + cr, st), bfrag(0, 2, CS + cr, st));
                mma6(acc_ni[s], ah, am, al, bfrag(0, 0, 2 * CS + cr, st), bfrag(0, 1, 2 * CS + cr, st),
                     bfrag(0, 2, 2 * CS + cr, st));
            }
        }
        if (t + stride < tiles) load_frags(m, t + stride, fa);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            bf16x8 ah, am, al;
            split8(fb[2 * st], fb[2 * st + 1], ah, am, al);
#pragma unroll
            for (int s = 0; s < NCS; ++s) {
                const int cr = 32 * s + r;
                mma6(acc_r[s], ah, am, al, bfrag(1, 0, cr, st), bfrag(1, 1, cr, st), bfrag(1, 2, cr, st));
                mma6(acc_z[s], ah, am, al, bfrag(1, 0, CS + cr, st), bfrag(1, 1, CS + cr, st), bfrag(1, 2, CS + cr, st));
                mma6(acc_nh[s], ah, am, al, bfrag(1, 0, 2 * CS + cr, st), bfrag(1, 1, 2 * CS + cr, st),
                     bfrag(1, 2, 2 * CS + cr, st));
            }
        }
        // epilogue: 4 groups of 4 consecutive atoms, loads unconditional (row clamped), stores predicated
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float mk4[4], hv4[4][NCS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t row = t * 32 + 8 * g + 4 * hi + u;
                if (row >= V) row = V - 1;
                mk4[u] = HAS_MASK ? mask[row] : 1.0f;
#pragma unroll
                for (int s = 0; s < NCS; ++s) hv4[u][s] = h[row * H + c0 + 32 * s + r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * g + u;
                const int64_t row = t * 32 + 8 * g + 4 * hi + u;
                const float mk = mk4[u];
#pragma unroll
                for (int s = 0; s < NCS; ++s) {
                    const int col = c0 + 32 * s + r;
                    const float rg = sigmoid_fast(acc_r[s][i] + br[s]) * mk;
                    const float zg = sigmoid_fast(acc_z[s][i] + bz[s]) * mk;
                    const float nh = acc_nh[s][i] + bnh[s];
                    const float ng = tanh_fast(acc_ni[s][i] + bni[s] + rg * nh) * mk;
                    const float o = ((1.0f - zg) * ng + zg * hv4[u][s]) * mk;
                    if (row < V) {
                        __builtin_nontemporal_store(o, out + row * H + col);
                        if (saved) {                     // 16*H bytes per atom, read back once by the backward
                            float* sv = saved + row * 4 * H + col;
                            __builtin_nontemporal_store(rg, sv);
                            __builtin_nontemporal_store(zg, sv + H);
                            __builtin_nontemporal_store(ng, sv + 2 * H);
                            __builtin_nontemporal_store(nh, sv + 3 * H);
                        }
                    }
                }
            }
        }
    }
}

template <int H, int NCS, int NW>
static int launch_split(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                        const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, hipStream_t s) {
    constexpr int CS = 32 * NCS;
    constexpr int slices = H / CS;
    const size_t lds = (size_t)2 * 3 * (3 * CS) * (2 * H);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gru_update_split_kernel<H, NCS, NW, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gru_update_split_kernel<H, NCS, NW, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int64_t tiles = (V + 31) / 32;
    int64_t pblocks = (256 + slices - 1) / slices;        // one block per CU (144 KB of LDS)
    if (pblocks * NW > tiles) pblocks = (tiles + NW - 1) / NW;
    if (pblocks < 1) pblocks = 1;
    const dim3 grid((unsigned)(pblocks * slices)), block(64 * NW);
    if (mask)
        hipLaunchKernelGGL((gru_update_split_kernel<H, NCS, NW, true>), grid, block, lds, s, m, h, mask, W_ih, W_hh, b_ih,
                           b_hh, out, saved, V, slices);
    else
        hipLaunchKernelGGL((gru_update_split_kernel<H, NCS, NW, false>), grid, block, lds, s, m, h, mask, W_ih, W_hh,
                           b_ih, b_hh, out, saved, V, slices);
    return launch_status("mpnn_gru_update_f32(bf16x6)");
}


// ---------------------------------------------------------------------------------------------------------------
// H = 128.  The three bf16 images of both matrices are 590 KB, so a block keeps one 32-feature column slice
// (r|z|n columns of both matrices = 144 KB) and four blocks cover a row tile.  The four blocks of one row tile
// are numbered to land on the same XCD (blockIdx % 8), so the tile's m/h rows come out of one L2.
// A lane's half-row is 64 floats per operand: holding both operands plus a prefetched pair is 256 registers on
// its own, so the rows move through a RING of four 16-float chunks (two K=16 steps each), fetched three chunks
// ahead (~3.4k matrix-pipe cycles of cover); scheduling barriers pin that order, otherwise the compiler renames
// the ring away and hoists every LDS read (558 spilled registers).
template <bool HAS_MASK>
__global__ void __launch_bounds__(512) gru_update_split128_kernel(
    const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask,
    const float* __restrict__ W_ih, const float* __restrict__ W_hh, const float* __restrict__ b_ih,
    const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ saved, int64_t V) {
    constexpr int H = 128, CS = 32, NCOL = 96, ROWB = 2 * H, NCH = H / 8, IMG = NCOL * ROWB, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 matrices][3 pieces][96][128] bf16

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = j & 3;
    const int pblock = (j >> 2) * 8 + xcd, pblocks = gridDim.x >> 2;
    const int c0 = slice * CS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    for (int idx = tid; idx < 2 * H * (NCOL / 4); idx += 64 * NW) {
        const int mat = idx / (H * (NCOL / 4));
        const int rem = idx % (H * (NCOL / 4));
        const int k = rem / (NCOL / 4), q = rem % (NCOL / 4);
        const int g = (4 * q) / CS, cc = (4 * q) % CS;
        const float* W = mat == 0 ? W_ih : W_hh;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)k * 3 * H + g * H + c0 + cc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = g * CS + cc + u;
            __bf16 ph, pm, pl;
            split3(w4[u], ph, pm, pl);
            const int off = col * ROWB + (((k >> 3) ^ col_swizzle<H>(col)) << 4) + ((k & 7) << 1);
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 0) * IMG + off) = ph;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 1) * IMG + off) = pm;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 2) * IMG + off) = pl;
        }
    }
    __syncthreads();

    const int r = lane & 31, hi = lane >> 5;
    const int col = c0 + r;
    const float br = b_ih[col] + b_hh[col], bz = b_ih[H + col] + b_hh[H + col];
    const float bni = b_ih[2 * H + col], bnh = b_hh[2 * H + col];

    const int64_t tiles = (V + 31) / 32;
    const int64_t stride = (int64_t)pblocks * NW;
    int64_t t = (int64_t)pblock * NW + wv;
    if (t >= tiles) return;

    f32x4 ring[4][4];
    // chunk c of a tile: operand c>>2 (m, h), floats [16*(c&3), +16) of the lane's half-row
    auto load_chunk = [&](int64_t tile, int c, f32x4 (&f)[4]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;
        const float* p = ((c >> 2) ? h : m) + row * H + hi * (H / 2) + 16 * (c & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    auto bfrag = [&](int mat, int piece, int wcol, int st) {
        const int chunk = hi * (NCH / 2) + st;
        return *reinterpret_cast<const bf16x8*>(smem + (mat * 3 + piece) * IMG + wcol * ROWB +
                                                ((chunk ^ col_swizzle<H>(wcol)) << 4));
    };

    load_chunk(t, 0, ring[0]);
    load_chunk(t, 1, ring[1]);
    load_chunk(t, 2, ring[2]);
    for (; t < tiles; t += stride) {
        const int64_t tn = t + stride < tiles ? t + stride : t;     // last tile: harmless re-read
        f32x16 acc_r, acc_z, acc_ni, acc_nh;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_r[i] = 0.f; acc_z[i] = 0.f; acc_ni[i] = 0.f; acc_nh[i] = 0.f; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c + 3 < 8) load_chunk(t, c + 3, ring[(c + 3) & 3]);
            else load_chunk(tn, c + 3 - 8, ring[(c + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            const int mat = c >> 2;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int st = 2 * (c & 3) + s2;
                bf16x8 ah, am, al;
                split8(ring[c & 3][2 * s2], ring[c & 3][2 * s2 + 1], ah, am, al);
                mma6(acc_r, ah, am, al, bfrag(mat, 0, r, st), bfrag(mat, 1, r, st), bfrag(mat, 2, r, st));
                mma6(acc_z, ah, am, al, bfrag(mat, 0, CS + r, st), bfrag(mat, 1, CS + r, st), bfrag(mat, 2, CS + r, st));
                if (mat == 0)
                    mma6(acc_ni, ah, am, al, bfrag(0, 0, 2 * CS + r, st), bfrag(0, 1, 2 * CS + r, st),
                         bfrag(0, 2, 2 * CS + r, st));
                else
                    mma6(acc_nh, ah, am, al, bfrag(1, 0, 2 * CS + r, st), bfrag(1, 1, 2 * CS + r, st),
                         bfrag(1, 2, 2 * CS + r, st));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float mk4[4], hv4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t row = t * 32 + 8 * g + 4 * hi + u;
                if (row >= V) row = V - 1;
                mk4[u] = HAS_MASK ? mask[row] : 1.0f;
                hv4[u] = h[row * H + col];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * g + u;
                const int64_t row = t * 32 + 8 * g + 4 * hi + u;
                const float mk = mk4[u];
                const float rg = sigmoid_fast(acc_r[i] + br) * mk;
                const float zg = sigmoid_fast(acc_z[i] + bz) * mk;
                const float nh = acc_nh[i] + bnh;
                const float ng = tanh_fast(acc_ni[i] + bni + rg * nh) * mk;
                const float o = ((1.0f - zg) * ng + zg * hv4[u]) * mk;
                if (row < V) {
                    __builtin_nontemporal_store(o, out + row * H + col);
                    if (saved) {
                        float* sv = saved + row * 4 * H + col;
                        __builtin_nontemporal_store(rg, sv);
                        __builtin_nontemporal_store(zg, sv + H);
                        __builtin_nontemporal_store(ng, sv + 2 * H);
                        __builtin_nontemporal_store(nh, sv + 3 * H);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

static int launch_split128(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                           const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * 96 * 256;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gru_update_split128_kernel<true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gru_update_split128_kernel<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int64_t tiles = (V + 31) / 32;
    int64_t pblocks = 64;                                   // x 4 slices = one block per CU
    if (pblocks * 8 > tiles) pblocks = (tiles + 7) / 8;
    pblocks = (pblocks + 7) / 8 * 8;                        // the XCD-aware numbering wants groups of 8 row blocks
    const dim3 grid((unsigned)(pblocks * 4)), block(512);
    if (mask)
        hipLaunchKernelGGL((gru_update_split128_kernel<true>), grid, block, lds, s, m, h, mask, W_ih, W_hh, b_ih, b_hh,
                           out, saved, V);
    else
        hipLaunchKernelGGL((gru_update_split128_kernel<false>), grid, block, lds, s, m, h, mask, W_ih, W_hh, b_ih, b_hh,
                           out, saved, V);
    return launch_status("mpnn_gru_update_f32(bf16x6, H=128)");
}

// returns 1 when the width has no split-precision path
int launch_gru_split(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                     const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, int H, hipStream_t s) {
    if (H == 64) return launch_split<64, 2, 8>(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, s);
    if (H == 128) return launch_split128(m, h, mask, W_ih, W_hh, b_ih, b_hh, out, saved, V, s);
    return 1;
}

}  // namespace mpnn
