// Masked GRU node update at SMALL widths (H <= 40: the Lipophilicity model's 22-38 features), forward and the contractions
// of the backward, on the vector pipe.
// replaces: mpnn_functions/update/gru_update.py:13-35 (GRUCell.forward + the mask) and its autograd, as gru.hip /
//           backward.hip do at every width.
//
// Why a second form: the generic kernels pad the contraction to the shapes of v_mfma_f32_32x32x2_f32 (K in chunks of 64,
// 32 output columns, 128-row tiles; 64 cycles per instruction) and cost 17-20 us per forward launch and ~70 us per backward
// (gate gradients + two row GEMMs + two accumulating contractions) WHATEVER the batch -- at the reference driver's batches
// of 16 molecules (~430 atoms, test_lipo.py:150) that is half of a recorded training step.  At these widths the whole
// problem is a few hundred thousand multiply-adds: both weight matrices (2 x H x 3H floats, 12-38 KB) sit in LDS, a thread
// owns one (atom, feature) and runs its 6 H multiply-adds from LDS operands.
//   forward   gru_update_small_kernel<NORM>      same contract as gru_update_kernel<NORM> (gru.hip), incl. the fused norm
//   backward  gru_bwd_small_kernel<NQ, NORM>     gate gradients, dm, dh, dW_ih, dW_hh, db_ih, db_hh -- one launch for five, no
//                                                (V, 6H) workspace
// fp32 throughout (fused multiply-adds in a fixed order): no operand splitting, nothing to range-guard.
#include "common.h"

namespace mpnn {

constexpr int kSmallMaxH = 40;      // 6 H <= 256: one thread per (matrix, gate column) in the backward's second phase

namespace {
__device__ __forceinline__ float s_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float s_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}
}  // namespace

// LDS: Wi [H][3H] | Wh [H][3H] | Xm [RB][H] | Xh [RB][H] | (NORM) red [RB][2][HP] doubles
template <bool NORM>
__global__ void __launch_bounds__(256) gru_update_small_kernel(
    const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask, const float* __restrict__ W_ih,
    const float* __restrict__ W_hh, const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ out,
    float* __restrict__ saved, int64_t V, int H, const float* __restrict__ hs, const float* __restrict__ ht,
    float* __restrict__ hnorm, double* stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HP = H <= 32 ? 32 : 64, RB = 256 / HP;       // lanes per atom row, atoms per pass
    float* Wi = reinterpret_cast<float*>(smem);
    float* Wh = Wi + H * 3 * H;
    float* Xm = Wh + H * 3 * H;
    float* Xh = Xm + RB * H;
    const int tid = threadIdx.x;
    const int rl = tid / HP, j = tid % HP;
    for (int i = tid; i < H * 3 * H; i += 256) {
        Wi[i] = W_ih[i];
        Wh[i] = W_hh[i];
    }
    const bool col = j < H;
    float br = 0.f, bz = 0.f, bni = 0.f, bnh = 0.f, hsc = 1.0f, hsh = 0.0f;
    if (col) {
        br = b_ih[j] + b_hh[j];
        bz = b_ih[H + j] + b_hh[H + j];
        bni = b_ih[2 * H + j];
        bnh = b_hh[2 * H + j];
        if (NORM) { hsc = hs[j]; hsh = ht[j]; }
    }
    double sum1 = 0.0, sum2 = 0.0;
    const int64_t groups = (V + RB - 1) / RB;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t row = g * RB + rl;
        const bool live = col && row < V;
        float mk = 1.0f, hv = 0.f;
        __syncthreads();                                    // the previous pass is done with Xm / Xh (first pass: W is in)
        if (live) {
            mk = mask ? mask[row] : 1.0f;
            hv = h[row * H + j];
            Xm[rl * H + j] = m[row * H + j];
            Xh[rl * H + j] = hv;                            // NORM: the contraction takes the RAW state (the caller folded the map
            if (NORM) hv = fmaf(hv, hsc, hsh) * mk;         // into W_hh / b_hh); the normalised one enters the z * h term
        }
        __syncthreads();
        if (!live) continue;
        float ar = br, az = bz, ani = bni, anh = bnh;
        const float* xm = Xm + rl * H;
        const float* xh = Xh + rl * H;
#pragma unroll 4
        for (int k = 0; k < H; ++k) {                       // (unrolled: the LDS reads of several steps go out together)
            const float a = xm[k], b = xh[k];
            const float* wi = Wi + k * 3 * H + j;
            const float* wh = Wh + k * 3 * H + j;
            ar = fmaf(a, wi[0], ar);
            az = fmaf(a, wi[H], az);
            ani = fmaf(a, wi[2 * H], ani);
            ar = fmaf(b, wh[0], ar);
            az = fmaf(b, wh[H], az);
            anh = fmaf(b, wh[2 * H], anh);
        }
        const float r = s_sigmoid(ar) * mk;
        const float z = s_sigmoid(az) * mk;
        const float n = s_tanh(ani + r * anh) * mk;
        const float o = ((1.0f - z) * n + z * hv) * mk;
        out[row * H + j] = o;
        if (NORM) {
            sum1 += (double)o;
            sum2 += (double)o * (double)o;
        }
        if (saved) {
            float* sv = saved + row * 4 * H + j;
            sv[0] = r;
            sv[H] = z;
            sv[2 * H] = n;
            sv[3 * H] = anh;
            if (NORM) hnorm[row * H + j] = hv;
        }
    }
    if (NORM) {                                            // column sums over the block's rows, then one atomic per column
        double* red = reinterpret_cast<double*>(smem);     // (the weights are no longer needed)
        __syncthreads();
        red[(rl * 2 + 0) * HP + j] = sum1;
        red[(rl * 2 + 1) * HP + j] = sum2;
        __syncthreads();
        if (rl == 0 && col) {
            for (int q = 1; q < RB; ++q) {
                sum1 += red[(q * 2 + 0) * HP + j];
                sum2 += red[(q * 2 + 1) * HP + j];
            }
            atomicAdd(stats + j, sum1);
            atomicAdd(stats + H + j, sum2);
        }
    }
}

static size_t gru_small_lds(int H) {
    const int HP = H <= 32 ? 32 : 64, RB = 256 / HP;
    size_t b = (size_t)(2 * H * 3 * H + 2 * RB * H) * sizeof(float);
    const size_t red = (size_t)RB * 2 * HP * sizeof(double);
    return b > red ? b : red;
}

// small widths AND small batches: past a few thousand atoms the per-block weight staging, the serial passes of a block and
// its atomics stop paying (27 k atoms: 19 / 75 us against the matrix-pipe kernels' 20 / 70), so those keep the old path
bool gru_small_covers(int H, int64_t V) { return H >= 1 && H <= kSmallMaxH && V <= 8192; }

int launch_gru_small(const float* m, const float* h, const float* mask, const float* W_ih, const float* W_hh,
                     const float* b_ih, const float* b_hh, float* out, float* saved, int64_t V, int H, const float* hs,
                     const float* ht, float* hnorm, double* stats, hipStream_t s) {
    const int RB = H <= 32 ? 8 : 4;
    int64_t blocks = (V + RB - 1) / RB;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = gru_small_lds(H);
    if (stats)
        hipLaunchKernelGGL(gru_update_small_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, m, h, mask, W_ih, W_hh, b_ih,
                           b_hh, out, saved, V, H, hs, ht, hnorm, stats);
    else
        hipLaunchKernelGGL(gru_update_small_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, m, h, mask, W_ih, W_hh, b_ih,
                           b_hh, out, saved, V, H, hs, ht, hnorm, stats);
    return launch_status("mpnn_gru_update_f32(small width)");
}

// ---------------------------------------------------------------------------------------------------------- backward
// The whole backward of one update in one launch (the generic path: a gate-gradient pass into a (V, 6H) workspace, two row
// GEMMs, two accumulating contractions).  Per pass of RB atoms:
//   staging, thread (atom, feature j): the gate gradients of (atom, j) from dout, h and the saved gates (gru_update.py:29-34
//            differentiated; NORM: dout is the gradient of norm(out), see gru_gate_grad_kernel) -> G [RB][6H] in LDS as
//            [dgi_r dgi_z dgi_n | dgh_r dgh_z dgh_n]; the direct term g * z of dh stays in a register
//   phase A, thread (atom, feature j):   dm[j] = sum_g dgi[g] W_ih[j][g],   dh[j] = g z + sum_g dgh[g] W_hh[j][g]
//   phase B, thread (matrix, gate column g): dW[j][g] += x[j] * d[g] for every feature j (H register accumulators),
//                                            db[g] += d[g]; the block's sums leave by atomicAdd at its end
// LDS: Wi [H][3H + 1] | Wh [H][3H + 1] (odd row stride: a lane per row reads conflict-free) | G [RB][6H] | Xm [RB][H] | Xh [RB][H]
template <int NQ, bool NORM>   // H <= NQ: accumulators of phase B (a multiple of 8: the loop over features carries no branch)
__global__ void __launch_bounds__(256) gru_bwd_small_kernel(
    const float* __restrict__ dout, const float* __restrict__ m, const float* __restrict__ h, const float* __restrict__ mask,
    const float* __restrict__ W_ih, const float* __restrict__ W_hh, const float* __restrict__ saved,
    const float* __restrict__ kn, float* __restrict__ dm, float* __restrict__ dh, float* dW_ih, float* dW_hh, float* db_ih,
    float* db_hh, int64_t V, int H, int64_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HP = H <= 32 ? 32 : 64, RB = 256 / HP, LDW = 3 * H + 1;
    float* Wi = reinterpret_cast<float*>(smem);
    float* Wh = Wi + H * LDW;
    float* G = Wh + H * LDW;
    float* Xm = G + RB * 6 * H;
    float* Xh = Xm + RB * H;
    const int tid = threadIdx.x;
    const int rl = tid / HP, j = tid % HP;
    for (int i = tid; i < H * 3 * H; i += 256) {
        const int jj = i / (3 * H), g = i - jj * 3 * H;
        Wi[jj * LDW + g] = W_ih[i];
        Wh[jj * LDW + g] = W_hh[i];
    }
    const bool col = j < H;
    float k1 = 0.f, k2 = 0.f, k4 = 0.f;
    if (NORM && col) { k1 = kn[j]; k2 = kn[H + j]; k4 = kn[2 * H + j]; }
    // phase B role: tid -> (matrix, gate column)
    const bool bwork = tid < 6 * H;
    const int bmat = tid >= 3 * H ? 1 : 0;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    float bsum = 0.f;

    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    int64_t r_end = r_begin + rows_per_block;
    if (r_end > V) r_end = V;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += RB) {
        __syncthreads();                                    // the previous pass is done with G / Xm / Xh (first pass: W is in)
        const int nrows = (int)(r_end - r0 < RB ? r_end - r0 : RB);
        const int64_t row = r0 + rl;
        const bool live = col && rl < nrows;
        float direct = 0.f;
        if (live) {
            const float mk = mask ? mask[row] : 1.0f;
            const float* sv = saved + row * 4 * H + j;
            const float r = sv[0], z = sv[H], n = sv[2 * H], nh = sv[3 * H];
            const float hv = h[row * H + j], d = dout[row * H + j];
            float g = d * mk;                               // through the final "* mask"
            if (NORM) {
                const float y = ((1.0f - z) * n + z * hv) * mk;
                g = (d * k1 + y * k2 + k4) * mk;
            }
            const float dan = g * (1.0f - z) * mk * (1.0f - n * n);     // n = tanh(.) * mask
            const float dar = dan * nh * mk * r * (1.0f - r);
            const float daz = g * (hv - n) * mk * z * (1.0f - z);
            float* gr = G + rl * 6 * H + j;
            gr[0] = dar;
            gr[H] = daz;
            gr[2 * H] = dan;
            gr[3 * H] = dar;
            gr[4 * H] = daz;
            gr[5 * H] = dan * r;
            direct = g * z;
            Xm[rl * H + j] = m[row * H + j];
            Xh[rl * H + j] = hv;
        }
        __syncthreads();
        if (live) {
            const float* gi = G + rl * 6 * H;
            const float* gh = gi + 3 * H;
            const float* wi = Wi + j * LDW;
            const float* wh = Wh + j * LDW;
            float am = 0.f, ah = direct;
#pragma unroll 6
            for (int g = 0; g < 3 * H; ++g) {
                am = fmaf(gi[g], wi[g], am);
                ah = fmaf(gh[g], wh[g], ah);
            }
            dm[row * H + j] = am;
            dh[row * H + j] = ah;
        }
        if (bwork) {
            const float* X = bmat ? Xh : Xm;
            for (int rr = 0; rr < nrows; ++rr) {
                const float d = G[rr * 6 * H + tid];        // column tid of the row: [dgi | dgh] is (matrix, gate column) order
                bsum += d;
                const float* x = X + rr * H;
                // every accumulator, no test on q: with one the reads could not be issued ahead of the multiply-adds and a row
                // cost 22 LDS round trips (43 us per launch at 430 atoms); x[q] beyond H reads the next row or the padding
                // behind Xh, and accumulators beyond H are never stored
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fmaf(x[q], d, acc[q]);
            }
        }
    }
    if (bwork && r_begin < r_end) {
        const int g = tid - bmat * 3 * H;
        float* dW = bmat ? dW_hh : dW_ih;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q < H) atomicAdd(dW + q * 3 * H + g, acc[q]);
        atomicAdd((bmat ? db_hh : db_ih) + g, bsum);
    }
}

int launch_gru_bwd_small(const float* dout, const float* m, const float* h, const float* mask, const float* W_ih,
                         const float* W_hh, const float* saved, const float* out_norm_k, float* dm, float* dh, float* dW_ih,
                         float* dW_hh, float* db_ih, float* db_hh, int64_t V, int H, hipStream_t s) {
    const int RB = H <= 32 ? 8 : 4;
    int64_t rows_per_block = RB;                            // one pass per block (a pass is a chain of memory round trips);
    if ((V + rows_per_block - 1) / rows_per_block > 256)    // at most 256 blocks: every block ends in 6 H^2 atomics
        rows_per_block = ((V + 255) / 256 + RB - 1) / RB * RB;
    const int64_t blocks = (V + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)(2 * H * (3 * H + 1) + RB * 6 * H + 2 * RB * H + kSmallMaxH) * sizeof(float);   // + padding read by phase B
#define MPNN_SMALL_BWD(NQ)                                                                                                    \
    do {                                                                                                                      \
        if (out_norm_k)                                                                                                       \
            hipLaunchKernelGGL((gru_bwd_small_kernel<NQ, true>), dim3((unsigned)blocks), dim3(256), lds, s, dout, m, h, mask,  \
                               W_ih, W_hh, saved, out_norm_k, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, H, rows_per_block);      \
        else                                                                                                                  \
            hipLaunchKernelGGL((gru_bwd_small_kernel<NQ, false>), dim3((unsigned)blocks), dim3(256), lds, s, dout, m, h, mask, \
                               W_ih, W_hh, saved, out_norm_k, dm, dh, dW_ih, dW_hh, db_ih, db_hh, V, H, rows_per_block);      \
    } while (0)
    if (H <= 24) MPNN_SMALL_BWD(24);
    else if (H <= 32) MPNN_SMALL_BWD(32);
    else MPNN_SMALL_BWD(40);
#undef MPNN_SMALL_BWD
    return launch_status("mpnn_gru_update_bwd_f32(small width)");
}

}  // namespace mpnn
