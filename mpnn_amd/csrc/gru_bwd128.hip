// Backward of the masked GRU update at hidden width 128: the two GEMM families on the bf16 matrix pipe with
// 3-way operand splitting (split_math.h), reading the pre-activation gradients the gate-gradient kernel wrote
// (backward.hip, compact layout: ws[row] = [dar daz dan dnh], 4H floats).
//
//   gru_bwd_dx128_kernel   dm = dgi W_ih^T,  dh = dgh W_hh^T + (dout*mask*z already in dh)
//
// The fused single-kernel arrangement of gru_bwd.hip does not carry over: at H = 128 the split images of one
// weight matrix are 295 KB, so the weights are sliced by OUTPUT column (32 columns of dm and of dh per block =
// 32 rows of each matrix, 144 KB of LDS) exactly like the forward kernel (gru_split.hip, H = 128), four blocks
// per row tile placed on one XCD, and the contraction rows (the gate gradients) stream through a register ring.
#include "split_math.h"

namespace mpnn {

__global__ void __launch_bounds__(512) gru_bwd_dx128_kernel(const float* __restrict__ ws, const float* __restrict__ W_ih,
                                                            const float* __restrict__ W_hh, float* __restrict__ dm,
                                                            float* __restrict__ dh, int64_t V) {
    constexpr int H = 128, LDW = 4 * H, KC = 3 * H;
    constexpr int ROWB = 2 * KC;               // one image row = one weight row (3H gate columns) in bf16
    constexpr int IMG = 32 * ROWB;             // one (matrix, piece) image: 32 output columns
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 matrices][3 pieces][32][384] bf16

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb & 3;
    const int pblock = (jb >> 2) * 8 + xcd, pblocks = gridDim.x >> 2;
    const int c0 = slice * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // image[mat][piece][n][k] = W_mat[c0 + n][k]: the weight rows as they lie in memory, 16-byte chunks
    // XOR-swizzled by the row (rows are 192 dwords apart = the same bank)
    for (int idx = tid; idx < 2 * 32 * (KC / 4); idx += 64 * NW) {
        const int mat = idx / (32 * (KC / 4));
        const int rem = idx % (32 * (KC / 4));
        const int n = rem / (KC / 4), q = rem % (KC / 4);
        const float* W = mat == 0 ? W_ih : W_hh;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)(c0 + n) * KC + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * q + u;
            __bf16 ph, pm, pl;
            split3(w4[u], ph, pm, pl);
            const int off = n * ROWB + (((k >> 3) ^ (n & 15)) << 4) + ((k & 7) << 1);
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 0) * IMG + off) = ph;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 1) * IMG + off) = pm;
            *reinterpret_cast<__bf16*>(smem + (mat * 3 + 2) * IMG + off) = pl;
        }
    }
    __syncthreads();

    const int r = lane & 31, hi = lane >> 5;
    const int64_t tiles = (V + 31) / 32;
    const int64_t stride = (int64_t)pblocks * NW;
    int64_t t = (int64_t)pblock * NW + wv;
    if (t >= tiles) return;

    // chunk c (0..15) of a tile: gradient segment g = c >> 2 (dar, daz, dan, dnh), 16 floats at 16*(c&3) of the
    // lane half's 64 floats of that segment.  K index inside the weight row: gate block (g, or 2 for dnh).
    f32x4 ring[4][4];
    auto load_chunk = [&](int64_t tile, int c, f32x4 (&f)[4]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;
        const int g = c >> 2;
        const float* p = ws + row * LDW + g * H + hi * (H / 2) + 16 * (c & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };
    auto bfrag = [&](int mat, int piece, int chunk) {
        return *reinterpret_cast<const bf16x8*>(smem + (mat * 3 + piece) * IMG + r * ROWB + ((chunk ^ (r & 15)) << 4));
    };

    load_chunk(t, 0, ring[0]);
    load_chunk(t, 1, ring[1]);
    load_chunk(t, 2, ring[2]);
    for (; t < tiles; t += stride) {
        const int64_t tn = t + stride < tiles ? t + stride : t;
        f32x16 d_m, d_h;
#pragma unroll
        for (int i = 0; i < 16; ++i) { d_m[i] = 0.f; d_h[i] = 0.f; }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c + 3 < 16) load_chunk(t, c + 3, ring[(c + 3) & 3]);
            else load_chunk(tn, c + 3 - 16, ring[(c + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            const int g = c >> 2;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ah, am, al;
                split8(ring[c & 3][2 * s2], ring[c & 3][2 * s2 + 1], ah, am, al);
                const int chunk = 16 * (g == 3 ? 2 : g) + 8 * hi + 2 * (c & 3) + s2;
                if (g != 3) mma6(d_m, ah, am, al, bfrag(0, 0, chunk), bfrag(0, 1, chunk), bfrag(0, 2, chunk));
                if (g != 2) mma6(d_h, ah, am, al, bfrag(1, 0, chunk), bfrag(1, 1, chunk), bfrag(1, 2, chunk));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int col = c0 + r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float prev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t row = t * 32 + 8 * g4 + 4 * hi + u;
                if (row >= V) row = V - 1;
                prev[u] = dh[row * H + col];                      // dout*mask*z from the gate-gradient kernel
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * g4 + u;
                const int64_t row = t * 32 + 8 * g4 + 4 * hi + u;
                if (row < V) {
                    dm[row * H + col] = d_m[i];
                    dh[row * H + col] = d_h[i] + prev[u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

int launch_gru_bwd_dx128(const float* ws, const float* W_ih, const float* W_hh, float* dm, float* dh, int64_t V,
                         hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * 32 * (2 * 3 * 128);
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_dx128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t tiles = (V + 31) / 32;
    int64_t pblocks = 64;                                   // x 4 slices = one block per CU
    if (pblocks * 8 > tiles) pblocks = (tiles + 7) / 8;
    pblocks = (pblocks + 7) / 8 * 8;
    hipLaunchKernelGGL(gru_bwd_dx128_kernel, dim3((unsigned)(pblocks * 4)), dim3(512), lds, s, ws, W_ih, W_hh, dm, dh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dx, H=128)");
}

// ---------------------------------------------------------------------------------------------------------------
//   gru_bwd_dw128_kernel   dW_ih = m^T dgi, db_ih = colsum(dgi)   (blockIdx.y = 0)
//                          dW_hh = h^T dgh, db_hh = colsum(dgh)   (blockIdx.y = 1)
// The contraction runs over ATOMS, so an MFMA fragment is 8 consecutive atoms of one column.  A thread owns one
// of the 512 columns [X (128) | G (384)] of the step's 16 atoms: its 2 x 8 coalesced dword loads ARE two
// fragments; it splits them once and parks the three bf16 pieces in LDS ([piece][octet][column] 16-byte slots,
// conflict-free both ways), double-buffered, one barrier per 16-atom step.  The 8 waves then share the step:
// wave = (a-pair, b-triple) owns 2 x 3 of the 4 x 12 output tiles, 36 MFMAs per step, accumulators live in
// registers for the whole kernel and are flushed with one atomic pass.
__global__ void __launch_bounds__(512) gru_bwd_dw128_kernel(const float* __restrict__ m, const float* __restrict__ h,
                                                            const float* __restrict__ ws, float* dW_ih, float* dW_hh,
                                                            float* db_ih, float* db_hh, int64_t V) {
    constexpr int H = 128, LDW = 4 * H, NC = 4 * H;        // 512 staged columns
    constexpr int SLOT = NC * 16;                          // bytes of one (piece, octet) plane
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][3 pieces][2 octets][512][8] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;
    const int mat = blockIdx.y;
    const float* X = mat == 0 ? m : h;
    // this thread's column of the staged tile and where it lives in global memory
    // G columns: dgi = blocks 0,1,2 of ws; dgh = blocks 0,1,3
    const float* colp = tid < H ? X + tid : ws + (tid - H) + ((mat == 1 && tid >= 3 * H) ? H : 0);
    const int64_t ldc = tid < H ? H : LDW;
    const int ag = wv & 1, bg = wv >> 1;

    f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    float colsum = 0.f;

    const int64_t steps = (V + 15) / 16;
    float raw[16];
    // fetch is branch- and select-free (rows clamped); masking of the ragged last step happens in park(), so
    // nothing waits on the loads until the MFMAs of the current step are issued
    auto load_raw = [&](int64_t st) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int64_t row = st * 16 + u;
            if (row >= V) row = V - 1;
            raw[u] = colp[row * ldc];
        }
    };
    auto park = [&](int buf, int64_t st) {
        if (st * 16 + 16 > V) {                            // ragged last step only (block-uniform)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (st * 16 + u >= V) raw[u] = 0.f;
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const f32x4 x0 = {raw[8 * o], raw[8 * o + 1], raw[8 * o + 2], raw[8 * o + 3]};
            const f32x4 x1 = {raw[8 * o + 4], raw[8 * o + 5], raw[8 * o + 6], raw[8 * o + 7]};
            bf16x8 ph, pm, pl;
            split8(x0, x1, ph, pm, pl);
            char* base = smem + (size_t)buf * 6 * SLOT + o * SLOT + tid * 16;
            *reinterpret_cast<bf16x8*>(base) = ph;
            *reinterpret_cast<bf16x8*>(base + 2 * SLOT) = pm;
            *reinterpret_cast<bf16x8*>(base + 4 * SLOT) = pl;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) colsum += raw[u];
    };
    auto frag = [&](int buf, int piece, int col) {
        return *reinterpret_cast<const bf16x8*>(smem + (size_t)buf * 6 * SLOT + (piece * 2 + hi) * SLOT + col * 16);
    };

    int64_t st = blockIdx.x;
    int cur = 0;
    if (st < steps) {
        load_raw(st);
        park(0, st);
    }
    for (; st < steps; st += gridDim.x) {
        __syncthreads();                                   // buffer `cur` is complete, `cur^1` is free
        const bool more = st + gridDim.x < steps;
        if (more) load_raw(st + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bh[3], bm[3], bl[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int col = H + 32 * (3 * bg + b) + i;
            bh[b] = frag(cur, 0, col);
            bm[b] = frag(cur, 1, col);
            bl[b] = frag(cur, 2, col);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int col = 32 * (2 * ag + a) + i;
            const bf16x8 ah = frag(cur, 0, col), am = frag(cur, 1, col), al = frag(cur, 2, col);
#pragma unroll
            for (int b = 0; b < 3; ++b) mma6(acc[a][b], ah, am, al, bh[b], bm[b], bl[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) park(cur ^ 1, st + gridDim.x);
        cur ^= 1;
    }

    float* dW = mat == 0 ? dW_ih : dW_hh;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int col = 32 * (3 * bg + b) + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (2 * ag + a) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, acc[a][b][q]);
            }
        }
    if (tid >= H && blockIdx.x < steps) atomicAdd((mat == 0 ? db_ih : db_hh) + (tid - H), colsum);
}

int launch_gru_bwd_dw128(const float* m, const float* h, const float* ws, float* dW_ih, float* dW_hh, float* db_ih,
                         float* db_hh, int64_t V, hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * 2 * 512 * 16;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_dw128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t steps = (V + 15) / 16;
    int64_t gx = 128;                                       // x 2 matrices = one block per CU (96 KB of LDS)
    if (gx > steps) gx = steps;
    hipLaunchKernelGGL(gru_bwd_dw128_kernel, dim3((unsigned)gx, 2), dim3(512), lds, s, m, h, ws, dW_ih, dW_hh, db_ih,
                       db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dW, H=128)");
}

// ---------------------------------------------------------------------------------------------------------------
// The same parked-fragment weight-gradient kernel at H = 256.  The step still stages 512 columns, now
// [X (256) | one gate block of G (256)], so a block owns one (matrix, gate) pair: blockIdx.y = 3*matrix + gate,
// output = the 256 x 256 block dW_matrix[:, gate*H : (gate+1)*H] as 8 x 8 tiles, wave = (a-pair, b-quad).
// Reads the compact workspace (ws[row] = [dar daz dan dnh]): gate block 2 of W_hh pairs with block 3 (dnh).
__global__ void __launch_bounds__(512) gru_bwd_dw256_kernel(const float* __restrict__ m, const float* __restrict__ h,
                                                            const float* __restrict__ ws, float* dW_ih, float* dW_hh,
                                                            float* db_ih, float* db_hh, int64_t V) {
    constexpr int H = 256, LDW = 4 * H, NC = 2 * H;
    constexpr int SLOT = NC * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][3 pieces][2 octets][512][8] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;
    const int mat = blockIdx.y / 3, gate = blockIdx.y % 3;
    const float* X = mat == 0 ? m : h;
    const float* colp = tid < H ? X + tid : ws + ((mat == 1 && gate == 2) ? 3 : gate) * H + (tid - H);
    const int64_t ldc = tid < H ? H : LDW;
    const int ag = wv & 3, bg = wv >> 2;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    float colsum = 0.f;

    const int64_t steps = (V + 15) / 16;
    float raw[16];
    auto load_raw = [&](int64_t st) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int64_t row = st * 16 + u;
            if (row >= V) row = V - 1;
            raw[u] = colp[row * ldc];
        }
    };
    auto park = [&](int buf, int64_t st) {
        if (st * 16 + 16 > V) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (st * 16 + u >= V) raw[u] = 0.f;
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const f32x4 x0 = {raw[8 * o], raw[8 * o + 1], raw[8 * o + 2], raw[8 * o + 3]};
            const f32x4 x1 = {raw[8 * o + 4], raw[8 * o + 5], raw[8 * o + 6], raw[8 * o + 7]};
            bf16x8 ph, pm, pl;
            split8(x0, x1, ph, pm, pl);
            char* base = smem + (size_t)buf * 6 * SLOT + o * SLOT + tid * 16;
            *reinterpret_cast<bf16x8*>(base) = ph;
            *reinterpret_cast<bf16x8*>(base + 2 * SLOT) = pm;
            *reinterpret_cast<bf16x8*>(base + 4 * SLOT) = pl;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) colsum += raw[u];
    };
    auto frag = [&](int buf, int piece, int col) {
        return *reinterpret_cast<const bf16x8*>(smem + (size_t)buf * 6 * SLOT + (piece * 2 + hi) * SLOT + col * 16);
    };

    int64_t st = blockIdx.x;
    int cur = 0;
    if (st < steps) {
        load_raw(st);
        park(0, st);
    }
    for (; st < steps; st += gridDim.x) {
        __syncthreads();
        const bool more = st + gridDim.x < steps;
        if (more) load_raw(st + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = H + 32 * (4 * bg + b) + i;
            bh[b] = frag(cur, 0, col);
            bm[b] = frag(cur, 1, col);
            bl[b] = frag(cur, 2, col);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int col = 32 * (2 * ag + a) + i;
            const bf16x8 ah = frag(cur, 0, col), am = frag(cur, 1, col), al = frag(cur, 2, col);
#pragma unroll
            for (int b = 0; b < 4; ++b) mma6(acc[a][b], ah, am, al, bh[b], bm[b], bl[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) park(cur ^ 1, st + gridDim.x);
        cur ^= 1;
    }

    float* dW = (mat == 0 ? dW_ih : dW_hh) + gate * H;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = 32 * (4 * bg + b) + i;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * (2 * ag + a) + acc_row(q, lane);
                atomicAdd(dW + (int64_t)row * 3 * H + col, acc[a][b][q]);
            }
        }
    if (tid >= H && blockIdx.x < steps) atomicAdd((mat == 0 ? db_ih : db_hh) + gate * H + (tid - H), colsum);
}

int launch_gru_bwd_dw256(const float* m, const float* h, const float* ws, float* dW_ih, float* dW_hh, float* db_ih,
                         float* db_hh, int64_t V, hipStream_t s) {
    const size_t lds = (size_t)2 * 3 * 2 * 512 * 16;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_dw256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t steps = (V + 15) / 16;
    int64_t gx = 42;                                        // x 6 (matrix, gate) jobs = 252 blocks: one per CU, ONE wave of blocks (43 x 6 = 258 left two blocks for a second pass and doubled the kernel time)
    if (gx > steps) gx = steps;
    hipLaunchKernelGGL(gru_bwd_dw256_kernel, dim3((unsigned)gx, 6), dim3(512), lds, s, m, h, ws, dW_ih, dW_hh, db_ih,
                       db_hh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dW, H=256)");
}

// ---------------------------------------------------------------------------------------------------------------
//   gru_bwd_dx_stream_kernel<H>   dm, dh with the weights STREAMED (same idea as gru_update_stream_kernel): a block
// owns 64 output features of dm and of dh, the contraction over the 3H gate columns is cut into 64-wide chunks of one
// gate block, all threads split the next chunk of W_ih / W_hh rows into a double-buffered LDS image while the waves
// multiply the current one.  For the r and z blocks both products share the operand (dar, daz); the third block
// pairs dan with W_ih and dnh with W_hh.  Reads the compact workspace ws[row] = [dar daz dan dnh].
template <int H>
__global__ void __launch_bounds__(512) gru_bwd_dx_stream_kernel(const float* __restrict__ ws,
                                                                const float* __restrict__ W_ih,
                                                                const float* __restrict__ W_hh, float* __restrict__ dm,
                                                                float* __restrict__ dh, int64_t V) {
    constexpr int NS = H / 64, CPS = H / 64, NCT = 3 * CPS, LDW = 4 * H;
    constexpr int IMGC = 64 * 128;             // one (matrix, piece) chunk image: 64 output rows x 64 k bf16
    constexpr int BUF = 6 * IMGC;              // 48 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % NS;
    const int pblock = (jb / NS) * 8 + xcd, pblocks = gridDim.x / NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;

    const int64_t rounds_total = (V + 255) / 256;              // a round = 256 rows: every wave its own 32-row tile
    if (pblock >= rounds_total) return;
    const int64_t nrounds = (rounds_total - pblock + pblocks - 1) / pblocks;

    // staging unit = (matrix, output row n, k-octet): 1024 units, two per thread, 32 contiguous bytes each
    const float* wsrc[2];
    int ldst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = tid + 512 * j;
        const int mat = u / 512, rem = u % 512;
        const int n = rem / 8, o = rem % 8;
        wsrc[j] = (mat ? W_hh : W_ih) + (int64_t)(64 * slice + n) * 3 * H + 8 * o;
        ldst[j] = mat * 3 * IMGC + n * 128 + ((o ^ ((n >> 1) & 7)) << 4);
    }
    f32x4 raw[2][2];
    auto stage_load = [&](int ct) {                        // chunk ct = gate block ct / CPS, 64 columns at 64*(ct % CPS)
        const int off = (ct / CPS) * H + 64 * (ct % CPS);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            raw[j][0] = *reinterpret_cast<const f32x4*>(wsrc[j] + off);
            raw[j][1] = *reinterpret_cast<const f32x4*>(wsrc[j] + off + 4);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 ph, pm, pl;
            split8(raw[j][0], raw[j][1], ph, pm, pl);
            char* base = smem + buf * BUF + ldst[j];
            *reinterpret_cast<bf16x8*>(base) = ph;
            *reinterpret_cast<bf16x8*>(base + IMGC) = pm;
            *reinterpret_cast<bf16x8*>(base + 2 * IMGC) = pl;
        }
    };
    auto bfrag = [&](int buf, int mat, int piece, int nb, int st) {
        const int n = 32 * nb + r;
        const int o = 4 * hi + st;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BUF + (mat * 3 + piece) * IMGC + n * 128 +
                                                ((o ^ ((n >> 1) & 7)) << 4));
    };
    // this lane's 32 floats of chunk ct of the gradient block `blk` for row tile `tile`
    auto load_rows = [&](int64_t tile, int blk, int cc, f32x4 (&f)[8]) {
        int64_t row = tile * 32 + r;
        if (row >= V) row = V - 1;
        const float* p = ws + row * LDW + blk * H + 64 * cc + 32 * hi;
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
    };

    f32x16 d_m[2], d_h[2];                                 // 32 rows x 64 features of dm and of dh per wave
    f32x4 a0[8], a1[8], b0[8], b1[8];                      // operand of dm (a*) and of dh (b*, only in the n block)
    int cur = 0;
    int64_t tile = (int64_t)pblock * 8 + wv;

    auto chunk = [&](int ct, int64_t tile_next, f32x4 (&xa)[8], f32x4 (&xb)[8], f32x4 (&na)[8], f32x4 (&nb)[8]) {
        __syncthreads();
        const int cn = (ct + 1) % NCT;
        const int gn = cn / CPS;
        stage_load(cn);
        load_rows(cn == 0 ? tile_next : tile, gn, cn % CPS, na);
        if (gn == 2) load_rows(tile, 3, cn % CPS, nb);     // dnh (never the first chunk of a round)
        __builtin_amdgcn_sched_barrier(0);
        const bool shared = ct / CPS < 2;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 a_h, a_m, a_l;
            split8(xa[2 * st], xa[2 * st + 1], a_h, a_m, a_l);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                mma6(d_m[nb], a_h, a_m, a_l, bfrag(cur, 0, 0, nb, st), bfrag(cur, 0, 1, nb, st), bfrag(cur, 0, 2, nb, st));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!shared) split8(xb[2 * st], xb[2 * st + 1], a_h, a_m, a_l);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                mma6(d_h[nb], a_h, a_m, a_l, bfrag(cur, 1, 0, nb, st), bfrag(cur, 1, 1, nb, st), bfrag(cur, 1, 2, nb, st));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stage_write(cur ^ 1);
        cur ^= 1;
    };

    stage_load(0);
    stage_write(0);
    load_rows(tile, 0, 0, a0);
    for (int64_t rd = 0; rd < nrounds; ++rd) {
        const int64_t tile_next = rd + 1 < nrounds ? tile + (int64_t)pblocks * 8 : tile;
#pragma unroll
        for (int i = 0; i < 16; ++i) { d_m[0][i] = 0.f; d_m[1][i] = 0.f; d_h[0][i] = 0.f; d_h[1][i] = 0.f; }
#pragma unroll 1
        for (int ct = 0; ct < NCT; ct += 2) {
            chunk(ct, tile_next, a0, b0, a1, b1);
            chunk(ct + 1, tile_next, a1, b1, a0, b0);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int fcol = 64 * slice + 32 * nb + r;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float prev[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int64_t row = tile * 32 + 8 * g4 + 4 * hi + u;
                    if (row >= V) row = V - 1;
                    prev[u] = dh[row * H + fcol];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = 4 * g4 + u;
                    const int64_t row = tile * 32 + 8 * g4 + 4 * hi + u;
                    if (row < V) {
                        dm[row * H + fcol] = d_m[nb][i];
                        dh[row * H + fcol] = d_h[nb][i] + prev[u];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tile = tile_next;
    }
}

template <int H>
static int launch_dx_stream(const float* ws, const float* W_ih, const float* W_hh, float* dm, float* dh, int64_t V,
                            hipStream_t s) {
    constexpr int NS = H / 64;
    const size_t lds = (size_t)2 * 6 * 64 * 128;
    static const hipError_t attr_done = [&] {   // once per process, thread-safe (C++11 static initialisation)
        LdsOptIn opt_in_;
        opt_in_((const void*)gru_bwd_dx_stream_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return opt_in_.err;
    }();
    if (attr_done != hipSuccess) return lds_opt_in_failed(attr_done);
    const int64_t rounds = (V + 255) / 256;
    int64_t pblocks = 256 / NS;
    if (pblocks > rounds) pblocks = rounds;
    pblocks = (pblocks + 7) / 8 * 8;
    hipLaunchKernelGGL(gru_bwd_dx_stream_kernel<H>, dim3((unsigned)(pblocks * NS)), dim3(512), lds, s, ws, W_ih, W_hh, dm,
                       dh, V);
    return launch_status("mpnn_gru_update_bwd_f32(dx, streamed weights)");
}

int launch_gru_bwd_dx_stream256(const float* ws, const float* W_ih, const float* W_hh, float* dm, float* dh, int64_t V,
                                hipStream_t s) {
    return launch_dx_stream<256>(ws, W_ih, W_hh, dm, dh, V, s);
}
int launch_gru_bwd_dx_stream128(const float* ws, const float* W_ih, const float* W_hh, float* dm, float* dh, int64_t V,
                                hipStream_t s) {
    return launch_dx_stream<128>(ws, W_ih, W_hh, dm, dh, V, s);
}

}  // namespace mpnn
