// Masked batch normalisation over all atoms of a batch (the layers between message and GRU update in
// the lipophilicity and attention models).
//
// Reference: models/mask_batch_norm.py.  Two variants, selected by flags:
//   MaskBatchNorm1d (:18-38)  mean = sum(x*mask)/n ; var = sum(((x-mean)*mask)^2)/n (biased) ;
//                             y = ((x-mean)/(sqrt(var)+eps) * weight + bias) * mask     -- eps OUTSIDE the root;
//                             eval mode uses the running statistics the same way
//   MaskBatchNorm   (:5-15)   mean = sum(x)/n (numerator NOT masked) ; c = (x-mean)*mask ; var = sum(c^2)/n ;
//                             y = c / sqrt(var+eps)                                        -- eps inside
// n = sum(mask).  The variance is taken around the mean in a second pass, as the reference does (not
// E[x^2]-E[x]^2), so results agree to fp32 rounding.
//
// HBM-bound column reductions: a block walks a strip of rows, a thread owns one 16-byte column group of
// every RL-th row, partials are combined across the block in LDS and leave through one float atomic per
// column per block.  Forward = 2 reduction passes + 1 normalise pass (x read 3 times, y written once);
// backward = 1 reduction pass (three column sums) + 1 elementwise pass.
#include "common.h"

namespace mpnn {

constexpr int kBnMaskedMean = 1;   // mean numerator is masked (MaskBatchNorm1d)
constexpr int kBnEpsInside = 2;    // sqrt(var + eps) instead of sqrt(var) + eps (MaskBatchNorm)
constexpr int kBnUseStats = 4;     // eval mode: mean / var are given (running statistics)

// workspace (floats): [0,F) sum1  [F,2F) sum2  [2F,3F) sum3  [3F] count
template <int MODE>   // 0: sum1 = sum x*(mask|1), count ; 1: sum2 = sum ((x-mean)*mask)^2 ; 2: backward sums
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                        const float* __restrict__ dout, const float* __restrict__ mean_in,
                                                        float* __restrict__ ws, int64_t V, int F, int flags) {
    __shared__ float red[3][256];
    const int tid = threadIdx.x;
    const int cg = (F + 3) / 4;                       // column groups of 4 floats
    const int tpr = cg < 256 ? cg : 256;              // threads per row
    const int rl = 256 / tpr;                         // row lanes per block
    const int c4 = tid % tpr, lane_row = tid / tpr;
    const bool vec = (F & 3) == 0;
    const float cnt = (MODE == 0) ? 0.f : ws[3 * F];
    float count_part = 0.f;
    for (int cbase = c4; cbase < cg; cbase += tpr) {  // usually one pass (F <= 1024)
        const int c = 4 * cbase;
        f32x4 mu = {0.f, 0.f, 0.f, 0.f};
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < F) mu[j] = mean_in ? mean_in[c + j] : ws[c + j] / cnt;
        }
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
        if (lane_row < rl) {
            for (int64_t row = (int64_t)blockIdx.x * rl + lane_row; row < V; row += (int64_t)gridDim.x * rl) {
                const float mk = mask ? mask[row] : 1.0f;
                f32x4 xv = {0.f, 0.f, 0.f, 0.f}, dv = xv;
                const float* px = x + row * F + c;
                if (vec) {
                    xv = *reinterpret_cast<const f32x4*>(px);
                    if (MODE == 2) dv = *reinterpret_cast<const f32x4*>(dout + row * F + c);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < F) {
                            xv[j] = px[j];
                            if (MODE == 2) dv[j] = dout[row * F + c + j];
                        }
                }
                if (MODE == 0) {
                    s1 += xv * ((flags & kBnMaskedMean) ? mk : 1.0f);
                    if (cbase == c4 && c4 == 0) count_part += mk;
                } else if (MODE == 1) {
                    const f32x4 cen = (xv - mu) * mk;
                    s2 += cen * cen;
                } else {
                    const f32x4 u = xv - mu;
                    const f32x4 g = dv * mk;
                    s1 += g;                           // S_b = sum dout*mask
                    s2 += g * u;                       // S_g = sum dout*mask*(x-mean)
                    s3 += u * (mk * mk);               // S_c = sum (x-mean)*mask^2
                }
            }
        }
        // block reduction over the row lanes of each column, one component at a time
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __syncthreads();
            red[0][tid] = s1[j];
            red[1][tid] = s2[j];
            red[2][tid] = s3[j];
            __syncthreads();
            if (lane_row == 0 && c + j < F) {
                float a = 0.f, b = 0.f, d = 0.f;
                for (int q = 0; q < rl; ++q) {
                    a += red[0][q * tpr + c4];
                    b += red[1][q * tpr + c4];
                    d += red[2][q * tpr + c4];
                }
                if (MODE == 0) {
                    atomicAdd(ws + c + j, a);
                } else if (MODE == 1) {
                    atomicAdd(ws + F + c + j, b);
                } else {
                    atomicAdd(ws + c + j, a);
                    atomicAdd(ws + F + c + j, b);
                    atomicAdd(ws + 2 * F + c + j, d);
                }
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        red[0][tid] = count_part;
        __syncthreads();
        if (tid == 0) {
            float a = 0.f;
            for (int q = 0; q < 256; ++q) a += red[0][q];
            atomicAdd(ws + 3 * F, a);
        }
    }
}

__device__ __forceinline__ float bn_scale(float var, float eps, int flags) {
    return (flags & kBnEpsInside) ? sqrtf(var + eps) : sqrtf(var) + eps;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                       const float* __restrict__ weight, const float* __restrict__ bias,
                                                       const float* __restrict__ ws, float* __restrict__ y,
                                                       float* __restrict__ mean_io, float* __restrict__ var_io,
                                                       float* __restrict__ count_out, int64_t V, int F, float eps,
                                                       int flags) {
    const int64_t total = V * F;
    const bool given = flags & kBnUseStats;
    const float cnt = given ? 1.f : ws[3 * F];
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / F;
        const int c = (int)(idx - row * F);
        const float mean = given ? mean_io[c] : ws[c] / cnt;
        const float var = given ? var_io[c] : ws[F + c] / cnt;
        const float mk = mask ? mask[row] : 1.0f;
        float v = (x[idx] - mean) / bn_scale(var, eps, flags);
        if (weight) v = weight[c] * v + bias[c];
        y[idx] = v * mk;
    }
    if (!given && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < F; c += 256) {
            mean_io[c] = ws[c] / cnt;
            var_io[c] = ws[F + c] / cnt;
        }
        if (threadIdx.x == 0 && count_out) *count_out = cnt;
    }
}

// dx and (by block 0) dweight / dbias from the three column sums
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ weight, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ ws,
                                                           float* __restrict__ dx, float* __restrict__ dweight,
                                                           float* __restrict__ dbias, int64_t V, int F, float eps,
                                                           int flags, const float* __restrict__ count_dev) {
    const int64_t total = V * F;
    const float count = *count_dev;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / F;
        const int c = (int)(idx - row * F);
        const float mk = mask ? mask[row] : 1.0f;
        const float gamma = weight ? weight[c] : 1.0f;
        const float s = bn_scale(var[c], eps, flags), rs = 1.0f / s;
        const float Sb = ws[c], Sg = ws[F + c], Sc = ws[2 * F + c];
        const float u = x[idx] - mean[c];
        // d/ds = -gamma*Sg/s^2 ; ds/dvar = 1/(2*sqrt(var)) (eps outside) or 1/(2*s) (eps inside)
        const float root = (flags & kBnEpsInside) ? s : sqrtf(var[c]);
        const float dvar = (root > 0.f) ? (-gamma * Sg * rs * rs) / (2.0f * root) : 0.f;
        const float du = dout[idx] * mk * gamma * rs + dvar * 2.0f * u * mk * mk / count;
        const float dmu = -(gamma * Sb * rs + dvar * 2.0f * Sc / count);
        dx[idx] = du + dmu * ((flags & kBnMaskedMean) ? mk : 1.0f) / count;
    }
    if (blockIdx.x == 0 && dweight)
        for (int c = threadIdx.x; c < F; c += 256) {
            dweight[c] = ws[F + c] / bn_scale(var[c], eps, flags);     // sum dout*mask*xhat
            dbias[c] = ws[c];
        }
}


// ---- vectorised twins of the two apply kernels for F = 4 * 2^k (<= 1024): a thread keeps ONE group of four
// columns for the whole kernel, so the per-column constants (1/scale, shift; the five backward coefficients) are
// computed once instead of per element (the scalar kernels spend their time on a 64-bit divide, a square root and a
// divide per element: 1.6 ms for 3 GB of traffic), and rows move as nontemporal float4.
__global__ void __launch_bounds__(256) bn_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                           const float* __restrict__ weight, const float* __restrict__ bias,
                                                           const float* __restrict__ ws, float* __restrict__ y,
                                                           float* __restrict__ mean_io, float* __restrict__ var_io,
                                                           float* __restrict__ count_out, int64_t V, int F, float eps,
                                                           int flags) {
    const int tpr = F / 4, rpb = 256 / tpr;                // threads per row, rows per block pass
    const int c = 4 * (threadIdx.x % tpr), rsub = threadIdx.x / tpr;
    const bool given = flags & kBnUseStats;
    const float cnt = given ? 1.f : ws[3 * F];
    f32x4 sc, sh;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float mean = given ? mean_io[c + u] : ws[c + u] / cnt;
        const float var = given ? var_io[c + u] : ws[F + c + u] / cnt;
        const float inv = 1.0f / bn_scale(var, eps, flags);
        const float g = weight ? weight[c + u] : 1.0f, b = weight ? bias[c + u] : 0.0f;
        sc[u] = g * inv;                                   // y = (x - mean) * g / s + b
        sh[u] = b - mean * g * inv;
    }
    for (int64_t row = (int64_t)blockIdx.x * rpb + rsub; row < V; row += (int64_t)gridDim.x * rpb) {
        const float mk = mask ? mask[row] : 1.0f;
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * F + c));
        __builtin_nontemporal_store((v * sc + sh) * mk, reinterpret_cast<f32x4*>(y + row * F + c));
    }
    if (!given && blockIdx.x == 0) {
        for (int cc = threadIdx.x; cc < F; cc += 256) {
            mean_io[cc] = ws[cc] / cnt;
            var_io[cc] = ws[F + cc] / cnt;
        }
        if (threadIdx.x == 0 && count_out) *count_out = cnt;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_vec_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ mean, const float* __restrict__ var,
                                                               const float* __restrict__ ws, float* __restrict__ dx,
                                                               float* __restrict__ dweight, float* __restrict__ dbias,
                                                               int64_t V, int F, float eps, int flags,
                                                               const float* __restrict__ count_dev) {
    const int tpr = F / 4, rpb = 256 / tpr;
    const int c = 4 * (threadIdx.x % tpr), rsub = threadIdx.x / tpr;
    const float count = *count_dev;
    // dx = dout*mk*k1 + (x - mean)*mk*mk*k2 + k3*(masked mean ? mk : 1)
    f32x4 k1, k2, k3, mu;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float gamma = weight ? weight[c + u] : 1.0f;
        const float s = bn_scale(var[c + u], eps, flags), rs = 1.0f / s;
        const float Sb = ws[c + u], Sg = ws[F + c + u], Sc = ws[2 * F + c + u];
        const float root = (flags & kBnEpsInside) ? s : sqrtf(var[c + u]);
        const float dvar = (root > 0.f) ? (-gamma * Sg * rs * rs) / (2.0f * root) : 0.f;
        k1[u] = gamma * rs;
        k2[u] = dvar * 2.0f / count;
        k3[u] = -(gamma * Sb * rs + dvar * 2.0f * Sc / count) / count;
        mu[u] = mean[c + u];
    }
    const bool masked_mean = flags & kBnMaskedMean;
    for (int64_t row = (int64_t)blockIdx.x * rpb + rsub; row < V; row += (int64_t)gridDim.x * rpb) {
        const float mk = mask ? mask[row] : 1.0f;
        const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * F + c));
        const f32x4 dv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dout + row * F + c));
        const f32x4 r = dv * mk * k1 + (xv - mu) * (mk * mk) * k2 + k3 * (masked_mean ? mk : 1.0f);
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(dx + row * F + c));
    }
    if (blockIdx.x == 0 && dweight)
        for (int cc = threadIdx.x; cc < F; cc += 256) {
            dweight[cc] = ws[F + cc] / bn_scale(var[cc], eps, flags);
            dbias[cc] = ws[cc];
        }
}

static bool bn_vectorisable(int F) {
    const int g = F / 4;
    return (F & 3) == 0 && g >= 1 && g <= 256 && (g & (g - 1)) == 0;
}

static int bn_grid(int64_t V, int F) {
    const int cg = (F + 3) / 4, tpr = cg < 256 ? cg : 256, rl = 256 / tpr;
    int64_t g = ceil_div(V, (int64_t)rl * 8);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- the norm fused into the GRU update (gru_split.hip NORM, gru_bwd128_f16.hip NORM): per-column constants ----
// moments (2F doubles: sum y | sum y^2 over the masked atoms) -> mean, var (biased, as the reference), the affine map
// hn = y * hs + ht of the norm, and the hidden weights / biases with that map folded in:
//   Wf[k, :] = hs[k] * W_hh[k, :],  bf = b_hh + ht W_hh.          One thread per gate column j < 3F.
__global__ void __launch_bounds__(256) bn_fold_kernel(const double* __restrict__ moments, const float* __restrict__ count,
                                                      const float* __restrict__ weight, const float* __restrict__ bias,
                                                      const float* __restrict__ W_hh, const float* __restrict__ b_hh,
                                                      float* __restrict__ mean, float* __restrict__ var,
                                                      float* __restrict__ hs, float* __restrict__ ht,
                                                      float* __restrict__ Wf, float* __restrict__ bf, int F, float eps,
                                                      int flags) {
    __shared__ float s_hs[256], s_ht[256];
    const double n = (double)*count;
    const bool given = flags & kBnUseStats;                // eval mode: `mean` / `var` are inputs (running statistics)
    for (int c = threadIdx.x; c < F; c += 256) {
        const double mu = moments[c] / n;
        double v = moments[F + c] / n - mu * mu;
        v = v > 0.0 ? v : 0.0;
        const float vf = given ? var[c] : (float)v, mf = given ? mean[c] : (float)mu;
        const float g = weight ? weight[c] : 1.0f, b = weight ? bias[c] : 0.0f;
        const float sc = g / bn_scale(vf, eps, flags);
        s_hs[c] = sc;
        s_ht[c] = b - mf * sc;
        if (blockIdx.x == 0) {
            if (!given) {
                mean[c] = mf;
                var[c] = vf;
            }
            hs[c] = sc;
            ht[c] = b - mf * sc;
        }
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 3 * F) return;
    float acc = b_hh[j];
    for (int k = 0; k < F; ++k) {
        const float w = W_hh[(int64_t)k * 3 * F + j];
        Wf[(int64_t)k * 3 * F + j] = w * s_hs[k];
        acc = fmaf(s_ht[k], w, acc);
    }
    bf[j] = acc;
}

// S_b = sum d * mask and S_y = sum d * mask * y over all rows (2F doubles, accumulated), y = the RAW input of the norm: the
// sums of the norm backward for a norm whose output left the chain (the last one), where no dm | dh epilogue produces them.
// One read of d and y.  (Round 3 summed against the norm's OUTPUT g xhat + b and divided by g afterwards: a weight entry
// of exactly 0 then lost its gradient for good, a tiny one amplified rounding -- ADVICE r3.)
__global__ void __launch_bounds__(256) bn_bwd_sums_kernel(const float* __restrict__ d, const float* __restrict__ hn,
                                                          const float* __restrict__ mask, double* __restrict__ sums,
                                                          int64_t V, int F) {
    __shared__ float red[2][256][4];
    const int tpr = F / 4, rpb = 256 / tpr;                // F = 4 * 2^k <= 1024 (checked by the caller)
    const int c = 4 * (threadIdx.x % tpr), rsub = threadIdx.x / tpr;
    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    double db[4] = {0.0, 0.0, 0.0, 0.0}, dh[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
    for (int64_t row = (int64_t)blockIdx.x * rpb + rsub; row < V; row += (int64_t)gridDim.x * rpb) {
        const float mk = mask ? mask[row] : 1.0f;
        const f32x4 dv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(d + row * F + c)) * mk;
        const f32x4 hv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(hn + row * F + c));
        sb += dv;
        sh += dv * hv;
        if (++n == 64) {                                   // fp32 over 64 rows, float64 beyond
#pragma unroll
            for (int u = 0; u < 4; ++u) { db[u] += (double)sb[u]; dh[u] += (double)sh[u]; sb[u] = 0.f; sh[u] = 0.f; }
            n = 0;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        red[0][threadIdx.x][u] = (float)(db[u] + (double)sb[u]);
        red[1][threadIdx.x][u] = (float)(dh[u] + (double)sh[u]);
    }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < rpb; ++q) {
                a += (double)red[0][q * tpr + threadIdx.x][u];
                b += (double)red[1][q * tpr + threadIdx.x][u];
            }
            atomicAdd(sums + c + u, a);
            atomicAdd(sums + F + c + u, b);
        }
    }
}

// any width: a block takes a strip of rows, thread = (row lane, column), float64 partials through LDS
__global__ void __launch_bounds__(256) bn_bwd_sums_any_kernel(const float* __restrict__ d, const float* __restrict__ hn,
                                                              const float* __restrict__ mask, double* __restrict__ sums,
                                                              int64_t V, int F) {
    __shared__ double red[2][256];
    const int tpr = F < 256 ? F : 256, rl = 256 / tpr;     // threads per row, row lanes (F <= 256: one pass over the columns)
    const int c = threadIdx.x % tpr, lr = threadIdx.x / tpr;
    double a = 0.0, b = 0.0;
    if (lr < rl)
        for (int64_t row = (int64_t)blockIdx.x * rl + lr; row < V; row += (int64_t)gridDim.x * rl) {
            const float mk = mask ? mask[row] : 1.0f;
            const float dv = d[row * F + c] * mk;
            a += (double)dv;
            b += (double)dv * (double)hn[row * F + c];
        }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (lr == 0 && threadIdx.x < tpr) {
        for (int q = 1; q < rl; ++q) { a += red[0][q * tpr + c]; b += red[1][q * tpr + c]; }
        atomicAdd(sums + c, a);
        atomicAdd(sums + F + c, b);
    }
}

// column sums of d hn and d hn * y (2F doubles, from the dm | dh kernel's epilogue; y = the norm's raw input) + the norm's
// statistics -> the three constants of  dy = d hn * k1 + y * k2 + k4  (include/mpnn_amd.h has the derivation), and the
// norm's own parameter gradients, accumulated
__global__ void __launch_bounds__(256) bn_bwd_consts_kernel(const double* __restrict__ sums, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ count,
                                                            const float* __restrict__ weight,
                                                            float* __restrict__ kn, float* __restrict__ dweight,
                                                            float* __restrict__ dbias, int F, float eps, int flags) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    const double n = (double)*count;
    const double g = weight ? (double)weight[c] : 1.0;
    const double s = (double)bn_scale(var[c], eps, flags), rs = 1.0 / s;
    const double Sb = sums[c], Sy = sums[F + c];
    const double Sg = Sy - (double)mean[c] * Sb;                       // sum d hn * mask * (y - mean): no division by g
    const double root = (flags & kBnEpsInside) ? s : sqrt((double)var[c]);
    const bool given = flags & kBnUseStats;               // eval mode: mean / var are constants, the norm is a plain affine map
    const double dvar = (root > 0.0 && !given) ? (-g * Sg * rs * rs) / (2.0 * root) : 0.0;
    const double k2 = 2.0 * dvar / n;
    kn[c] = (float)(g * rs);
    kn[F + c] = (float)k2;
    kn[2 * F + c] = given ? 0.0f : (float)(-g * Sb * rs / n - (double)mean[c] * k2);
    if (dweight) {
        dweight[c] += (float)(Sg * rs);
        dbias[c] += (float)Sb;
    }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" size_t mpnn_masked_bn_workspace_bytes(int F) { return F > 0 ? (size_t)(3 * F + 4) * sizeof(float) : 0; }

extern "C" int mpnn_masked_bn_fwd_f32(const float* x, const float* mask, const float* weight, const float* bias,
                                      float* y, float* mean, float* var, float* count_out, int64_t V, int F,
                                      float eps, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    MPNN_REQUIRE(V >= 0 && F > 0 && F <= 1024, "mpnn_masked_bn_fwd_f32: V=%lld F=%d out of range", (long long)V, F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(x && y && mean && var, "mpnn_masked_bn_fwd_f32: NULL buffer");
    MPNN_REQUIRE((weight == nullptr) == (bias == nullptr), "mpnn_masked_bn_fwd_f32: weight and bias go together");
    if (!workspace || workspace_bytes < mpnn_masked_bn_workspace_bytes(F)) {
        set_error("mpnn_masked_bn_fwd_f32: workspace %zu < %zu", workspace_bytes, mpnn_masked_bn_workspace_bytes(F));
        return MPNN_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    int64_t ag = ceil_div(V * F, 256);
    if (ag > 256 * 32) ag = 256 * 32;
    if (!(flags & kBnUseStats)) {
        (void)hipMemsetAsync(ws, 0, mpnn_masked_bn_workspace_bytes(F), s);
        const int g = bn_grid(V, F);
        hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(g), dim3(256), 0, s, x, mask, (const float*)nullptr,
                           (const float*)nullptr, ws, V, F, flags);
        hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(g), dim3(256), 0, s, x, mask, (const float*)nullptr,
                           (const float*)nullptr, ws, V, F, flags);
    }
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y);
    if (bn_vectorisable(F) && al % 16 == 0) {
        int64_t vg = ceil_div(V, (int64_t)(256 / (F / 4)));
        if (vg > 256 * 16) vg = 256 * 16;
        hipLaunchKernelGGL(bn_apply_vec_kernel, dim3((unsigned)vg), dim3(256), 0, s, x, mask, weight, bias, ws, y, mean, var,
                           count_out, V, F, eps, flags);
    } else {
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ag), dim3(256), 0, s, x, mask, weight, bias, ws, y, mean, var,
                           count_out, V, F, eps, flags);
    }
    return launch_status("mpnn_masked_bn_fwd_f32");
}

extern "C" int mpnn_masked_bn_bwd_f32(const float* dout, const float* x, const float* mask, const float* weight,
                                      const float* mean, const float* var, float* dx, float* dweight, float* dbias,
                                      int64_t V, int F, float eps, int flags, const float* count, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    MPNN_REQUIRE(V >= 0 && F > 0 && F <= 1024, "mpnn_masked_bn_bwd_f32: V=%lld F=%d out of range", (long long)V, F);
    MPNN_REQUIRE(!(flags & kBnUseStats), "mpnn_masked_bn_bwd_f32: eval-mode backward is a plain scale (not provided)");
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(dout && x && mean && var && dx && count, "mpnn_masked_bn_bwd_f32: NULL buffer");
    if (!workspace || workspace_bytes < mpnn_masked_bn_workspace_bytes(F)) {
        set_error("mpnn_masked_bn_bwd_f32: workspace %zu < %zu", workspace_bytes, mpnn_masked_bn_workspace_bytes(F));
        return MPNN_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    (void)hipMemsetAsync(ws, 0, mpnn_masked_bn_workspace_bytes(F), s);
    // the reduction kernel reads `count` from ws[3F] only for the mean; here the mean is given
    hipLaunchKernelGGL((bn_reduce_kernel<2>), dim3(bn_grid(V, F)), dim3(256), 0, s, x, mask, dout, mean, ws, V, F, flags);
    int64_t ag = ceil_div(V * F, 256);
    if (ag > 256 * 32) ag = 256 * 32;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dx);
    if (bn_vectorisable(F) && al % 16 == 0) {
        int64_t vg = ceil_div(V, (int64_t)(256 / (F / 4)));
        if (vg > 256 * 16) vg = 256 * 16;
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3((unsigned)vg), dim3(256), 0, s, dout, x, mask, weight, mean, var, ws,
                           dx, dweight, dbias, V, F, eps, flags, count);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ag), dim3(256), 0, s, dout, x, mask, weight, mean, var, ws, dx,
                           dweight, dbias, V, F, eps, flags, count);
    }
    return launch_status("mpnn_masked_bn_bwd_f32");
}

extern "C" int mpnn_norm_fold_f32(const double* moments, const float* count, const float* weight, const float* bias,
                                  const float* W_hh, const float* b_hh, float* mean, float* var, float* h_scale,
                                  float* h_shift, float* W_hh_folded, float* b_hh_folded, int F, float eps, int flags,
                                  void* stream) {
    MPNN_REQUIRE(F > 0 && F <= 256, "mpnn_norm_fold_f32: F=%d out of range (<= 256)", F);
    MPNN_REQUIRE(moments && count && W_hh && b_hh && mean && var && h_scale && h_shift && W_hh_folded && b_hh_folded,
                 "mpnn_norm_fold_f32: NULL buffer");
    MPNN_REQUIRE((weight == nullptr) == (bias == nullptr), "mpnn_norm_fold_f32: weight and bias go together");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((3 * F + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, count,
                       weight, bias, W_hh, b_hh, mean, var, h_scale, h_shift, W_hh_folded, b_hh_folded, F, eps, flags);
    return launch_status("mpnn_norm_fold_f32");
}

extern "C" int mpnn_norm_bwd_consts_f32(const double* sums, const float* mean, const float* var, const float* count,
                                        const float* weight, float* out_norm_k, float* dweight, float* dbias, int F,
                                        float eps, int flags, void* stream) {
    MPNN_REQUIRE(F > 0 && F <= 1024, "mpnn_norm_bwd_consts_f32: F=%d out of range", F);
    MPNN_REQUIRE(sums && mean && var && count && out_norm_k, "mpnn_norm_bwd_consts_f32: NULL buffer");
    MPNN_REQUIRE((weight == nullptr) == (dweight == nullptr) && (dweight == nullptr) == (dbias == nullptr),
                 "mpnn_norm_bwd_consts_f32: the weight and the parameter gradients go together");
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sums, mean,
                       var, count, weight, out_norm_k, dweight, dbias, F, eps, flags);
    return launch_status("mpnn_norm_bwd_consts_f32");
}

extern "C" int mpnn_norm_bwd_sums_f32(const float* dout, const float* y_raw, const float* mask, double* sums, int64_t V,
                                      int F, void* stream) {
    MPNN_REQUIRE(V >= 0 && F > 0 && (bn_vectorisable(F) || F <= 256),
                 "mpnn_norm_bwd_sums_f32: V=%lld F=%d (F <= 256, or 4 * 2^k <= 1024)", (long long)V, F);
    if (V == 0) return MPNN_OK;
    MPNN_REQUIRE(dout && y_raw && sums, "mpnn_norm_bwd_sums_f32: NULL buffer");
    if (!bn_vectorisable(F) || (reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(y_raw)) % 16 != 0) {
        MPNN_REQUIRE(F <= 256, "mpnn_norm_bwd_sums_f32: unaligned buffers need F <= 256");
        const int rl = 256 / (F < 256 ? F : 256);
        int64_t g = ceil_div(V, (int64_t)rl * 16);
        if (g > 1024) g = 1024;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(bn_bwd_sums_any_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dout, y_raw, mask,
                           sums, V, F);
        return launch_status("mpnn_norm_bwd_sums_f32");
    }
    int64_t vg = ceil_div(V, (int64_t)(256 / (F / 4)) * 16);
    if (vg > 2048) vg = 2048;
    if (vg < 1) vg = 1;
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((unsigned)vg), dim3(256), 0, (hipStream_t)stream, dout, y_raw, mask, sums, V, F);
    return launch_status("mpnn_norm_bwd_sums_f32");
}
