// elementwise.hip - HBM-bound kernels of the BigGAN step: spectral-norm power iteration, batch
// statistics, (conditional) batch-norm + PReLU forward/backward, pooling, softmax, small
// elementwise ops, bias gradient, fused TF-Adam + EMA.  16-byte vector accesses, wave64 shuffle
// reductions, grid-stride loops capped at 2048 blocks (256 CUs x 8).
#include <stdlib.h>
#include "common.h"

namespace bg {

#define EW_BLOCK 256
#define EW_MAX_BLOCKS 2048

static inline int ew_grid(int64_t work_items) {
    int64_t b = (work_items + EW_BLOCK - 1) / EW_BLOCK;
    if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// ==========================================================================================
// column-reduction skeleton: rows x C matrix (optionally segmented by blockIdx.y), each thread owns
// VEC consecutive columns and a strided subset of the block's rows; NQ quantities are reduced
// over rows, combined across the block's row-lanes through LDS and added to out[q*qstride + seg*C + c]
// with one float atomic per column per block.
// ==========================================================================================
template <int NQ, int VEC, class Fn, class OutT>
__global__ __launch_bounds__(EW_BLOCK) void colreduce_kernel(Fn fn, OutT* out, int64_t qstride,
                                                             int64_t rows_per_seg, int C, int rows_per_block) {
    __shared__ float red[NQ * EW_BLOCK * VEC];   // VEC*NQ floats per thread
    const int CV = C / VEC;                      // vector columns
    const int tx = CV < EW_BLOCK ? CV : EW_BLOCK;
    const int lanes = EW_BLOCK / tx;             // row-lanes per block
    const int cq = threadIdx.x % tx, rl = threadIdx.x / tx;
    const int seg = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows_per_seg) r1 = rows_per_seg;
    for (int cbase = 0; cbase < CV; cbase += tx) {
        const int cv = cbase + cq;
        float acc[NQ][VEC];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[q][j] = 0.f;
        if (cv < CV && rl < lanes) {
            // 4 independent partial sums keep 4 rows' loads in flight per thread (HBM latency hiding)
            float acc4[4][NQ][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc4[u][q][j] = 0.f;
            int64_t r = r0 + rl;
            for (; r + 3 * (int64_t)lanes < r1; r += 4 * (int64_t)lanes) {
#pragma unroll
                for (int u = 0; u < 4; ++u) fn(seg, r + u * (int64_t)lanes, cv * VEC, acc4[u]);
            }
            for (; r < r1; r += lanes) fn(seg, r, cv * VEC, acc4[0]);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    acc[q][j] = (acc4[0][q][j] + acc4[1][q][j]) + (acc4[2][q][j] + acc4[3][q][j]);
        }
        // combine row-lanes
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[(q * VEC + j) * EW_BLOCK + threadIdx.x] = acc[q][j];
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float s = 0.f;
                    for (int l = 0; l < lanes; ++l) s += red[(q * VEC + j) * EW_BLOCK + l * tx + cq];
                    atomicAdd(&out[q * qstride + (int64_t)seg * C + cv * VEC + j], (OutT)s);
                }
        }
        __syncthreads();
    }
}

// WIDE: 8 columns per thread when C % 8 == 0 - the 16-byte loads of a bf16 tensor (4 columns would be 8-byte loads and
// half the bytes in flight the block counts below were tuned for)
template <int NQ, bool WIDE = false, class Fn, class OutT>
static void launch_colreduce(Fn fn, OutT* out, int64_t qstride, int64_t rows_per_seg, int nseg, int C,
                             hipStream_t s) {
    // Every block ends with one atomic per column on the SAME addresses, and those serialise (~0.1 us per
    // block, measured: 1024 blocks = 105 us whatever the tensor size).  192 blocks x 4 loads in flight per
    // thread already cover the HBM bandwidth-delay product (268 MB in 55 us = 4.9 TB/s); 128 for < 100 MB.
    static const int env_blocks = getenv("BG_COLRED_BLOCKS") ? atoi(getenv("BG_COLRED_BLOCKS")) : 0;
    const int64_t bytes = rows_per_seg * (int64_t)nseg * C * 4;
    const int target_blocks = env_blocks > 0 ? env_blocks : (bytes > (100ll << 20) ? 192 : 128);
    int64_t blocks_per_seg = (target_blocks + nseg - 1) / nseg;
    int64_t rpb = (rows_per_seg + blocks_per_seg - 1) / blocks_per_seg;
    if (rpb < 32) rpb = 32;
    blocks_per_seg = (rows_per_seg + rpb - 1) / rpb;
    dim3 grid((unsigned)blocks_per_seg, (unsigned)nseg);
    if constexpr (WIDE) {
        // (small tensors: the 4-column form keeps twice the threads busy - measured 31 vs 38 us per launch at 32 images)
        if (C % 8 == 0 && rows_per_seg * (int64_t)nseg * C >= (int64_t(16) << 20)) {
            hipLaunchKernelGGL((colreduce_kernel<NQ, 8, Fn, OutT>), grid, dim3(EW_BLOCK), 0, s, fn, out, qstride,
                               rows_per_seg, C, (int)rpb);
            return;
        }
    }
    if (C % 4 == 0)
        hipLaunchKernelGGL((colreduce_kernel<NQ, 4, Fn, OutT>), grid, dim3(EW_BLOCK), 0, s, fn, out, qstride, rows_per_seg,
                           C, (int)rpb);
    else
        hipLaunchKernelGGL((colreduce_kernel<NQ, 1, Fn, OutT>), grid, dim3(EW_BLOCK), 0, s, fn, out, qstride, rows_per_seg,
                           C, (int)rpb);
}

template <int VEC>
__device__ __forceinline__ void loadv(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        float4 t = ldg4(p);
        v[0] = t.x;
        v[1] = t.y;
        v[2] = t.z;
        v[VEC - 1] = t.w;
    } else if constexpr (VEC == 8) {
        const float4 a = ldg4(p), b = ldg4(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[VEC - 1] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = p[j];
    }
}

// ------------------------------------------------------------------------------------------
// batch statistics (tf.nn.moments, ops.py:630)
// ------------------------------------------------------------------------------------------
struct BnStatsFn {
    const float* x;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[2][VEC]) const {
        float v[VEC];
        loadv<VEC>(x + r * C + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[0][j] += v[j];
            acc[1][j] += v[j] * v[j];
        }
    }
};

// inference: normalise with the population statistics (ops.py:643; tf.layers.batch_normalization(training=False))
__global__ void bn_population_kernel(const float* __restrict__ pop_mean, const float* __restrict__ pop_var, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = pop_mean[c];
    rstd[c] = rsqrtf(pop_var[c] + eps);
}

__global__ void bn_finalize_kernel(const double* sums, double count, float eps, float momentum, int unbiased,
                                   float* mean, float* rstd, float* mm, float* mv, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    rstd[c] = rsqrtf((float)var + eps);
    if (mm) mm[c] = mm[c] * momentum + (float)m * (1.f - momentum);
    if (mv) {
        double vv = var;
        if (unbiased && count > 1) vv = var * (count / (count - 1.0));
        mv[c] = mv[c] * momentum + (float)vv * (1.f - momentum);
    }
}

// ------------------------------------------------------------------------------------------
// Forward-mode tangent of training-mode batch norm and its backward (the gradient penalty, BigGAN.py:717-742, through a
// discriminator with --bn_in_d, ops.py:546-561).  With xh = (x - mu) r, r = rstd:
//   tangent   yd = g r (xd - m1 - xh m2),            m1 = mean(xd), m2 = mean(xd xh)
//   backward  (s = dL/dyd):  d_xd = g r (s - mean(s) - xh mean(s xh))          (the operator is symmetric)
//             d_g  = r (sum(s xd) - m1 sum(s) - m2 sum(s xh))
//             d_x  = xh (3 m2 Swx - Swu) / n - m2 w - (Swx / n)(xd - m1) + m2 Sw / n,   w = g r^2 s, S.. = sums of w (.)
//   (derived in DESIGN section 7; checked against finite differences on the CPU: tests/test_oracle.py).
// Everything elementwise is a per-channel linear combination of (s | xd), x and xd, so the pass is: one 3-quantity column
// reduction (chan_dots), one single-block coefficient kernel, one lincomb kernel - forward and backward alike.
// ------------------------------------------------------------------------------------------
struct ChanDotsFn {            // sum p, sum p q, sum p r  (r may be null: third sum 0)
    const float* p;
    const float* q;
    const float* r;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t row, int c, float (&acc)[3][VEC]) const {
        float pv[VEC], qv[VEC], rv[VEC];
        loadv<VEC>(p + row * C + c, pv);
        loadv<VEC>(q + row * C + c, qv);
        if (r) loadv<VEC>(r + row * C + c, rv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[0][j] += pv[j];
            acc[1][j] += pv[j] * qv[j];
            if (r) acc[2][j] += pv[j] * rv[j];
        }
    }
};

// forward coefficients: yd = cf[0] xd + cf[1] x + cf[2];  m12 = (m1 | m2) kept for the backward
__global__ void bn_tangent_fwd_coefs_kernel(const double* __restrict__ sums, double count, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                            float* __restrict__ cf, float* __restrict__ m12, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = mean[c], r = rstd[c], g = gamma[c];
    const double m1 = sums[c] / count;
    const double m2 = r * (sums[C + c] - mu * sums[c]) / count;
    cf[c] = (float)(g * r);
    cf[C + c] = (float)(-g * r * r * m2);
    cf[2 * C + c] = (float)(-g * r * m1 + g * r * r * m2 * mu);
    m12[c] = (float)m1;
    m12[C + c] = (float)m2;
}

// backward coefficients from S = (sum s | sum s x | sum s xd):
//   d_xd = cd[0] s + cd[1] x + cd[2];   d_x = cx[0] s + cx[1] x + cx[2] xd + cx[3];   dgamma
__global__ void bn_tangent_bwd_coefs_kernel(const double* __restrict__ S, double count, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                            const float* __restrict__ m12, float* __restrict__ cd, float* __restrict__ cx,
                                            float* __restrict__ dgamma, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = mean[c], r = rstd[c], g = gamma[c], n = count;
    const double m1 = m12[c], m2 = m12[C + c];
    const double S1 = S[c], S2 = S[C + c], S3 = S[2 * C + c];
    const double q = r * (S2 - mu * S1) / n;                 // mean(s xh)
    cd[c] = (float)(g * r);
    cd[C + c] = (float)(-g * r * r * q);
    cd[2 * C + c] = (float)(-g * r * S1 / n + g * r * r * q * mu);
    dgamma[c] = (float)(r * (S3 - m1 * S1 - m2 * n * q));
    const double Sw = g * r * r * S1, Swx = g * r * r * n * q, Swu = g * r * r * (S3 - m1 * S1);
    const double K = (3.0 * m2 * Swx - Swu) / n;
    cx[c] = (float)(-m2 * g * r * r);
    cx[C + c] = (float)(r * K);
    cx[2 * C + c] = (float)(-Swx / n);
    cx[3 * C + c] = (float)(-r * mu * K + (Swx / n) * m1 + m2 * Sw / n);
}

// out = cp[c] p + cq[c] q (+ cr[c] r) + c0[c]
__global__ __launch_bounds__(EW_BLOCK) void chan_lincomb3_kernel(const float* __restrict__ p, const float* __restrict__ cp,
                                                                  const float* __restrict__ q, const float* __restrict__ cq,
                                                                  const float* __restrict__ r, const float* __restrict__ cr,
                                                                  const float* __restrict__ c0, float* __restrict__ out,
                                                                  int64_t total, int C) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % C);
        float v = cp[c] * p[i] + cq[c] * q[i] + c0[c];
        if (r) v += cr[c] * r[i];
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// batch renormalisation: clipped corrections r, d against running statistics (ops.py:600-609, 645-715)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void renorm_coeffs_kernel(const double* __restrict__ sums, double count,
                                                            float* ref_mean, float* ref_scale, int scale_is_var,
                                                            float* weight, float eps, float rmin, float rmax, float dmax,
                                                            float decay, float fade, int update, float* __restrict__ r,
                                                            float* __restrict__ d, int C) {
    const float w = weight ? *weight : 1.f;              // every thread reads the fade-in weight before it moves
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double m64 = sums[c] / count;
        double v64 = sums[C + c] / count - m64 * m64;
        if (v64 < 0) v64 = 0;
        const float m = (float)m64, var = (float)v64;
        const float sigma = sqrtf(var + eps);
        const float rm = ref_mean[c], rs = ref_scale[c];
        const float sigma_ref = scale_is_var ? sqrtf(rs + eps) : fmaxf(rs, sqrtf(eps));
        const float sigma_w = w * sigma_ref + (1.f - w) * sigma;
        const float mean_w = w * rm + (1.f - w) * m;
        r[c] = fminf(fmaxf(sigma / sigma_w, rmin), rmax);
        d[c] = fminf(fmaxf((m - mean_w) / sigma_w, -dmax), dmax);
        if (update) {
            ref_mean[c] = rm * decay + m * (1.f - decay);
            ref_scale[c] = rs * decay + (scale_is_var ? var : sigma) * (1.f - decay);
        }
    }
    if (update && weight && threadIdx.x == 0) *weight = w * fade + 1.0f * (1.f - fade);
}

// dir 0: a = r*gamma, b = beta + d*gamma ; dir 1: a = r*dgamma_eff + d*dbeta_eff (gradient of gamma)
__global__ __launch_bounds__(EW_BLOCK) void renorm_affine_kernel(const float* __restrict__ g, const float* __restrict__ b,
                                                                 const float* __restrict__ r, const float* __restrict__ d,
                                                                 float* __restrict__ oa, float* __restrict__ ob,
                                                                 int64_t total, int C, int dir) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        if (dir == 0) {
            oa[i] = r[c] * g[i];
            ob[i] = b[i] + d[c] * g[i];
        } else {
            oa[i] = r[c] * g[i] + d[c] * b[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// (conditional) batch-norm apply + PReLU
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float prelu_f(float v, float a) { return v > 0.f ? v : a * v; }
// TF gradient of relu(x) + a*(x-|x|)/2: 1 for x>0, a for x<0, a/2 at x==0
__device__ __forceinline__ float prelu_d(float v, float a) { return v > 0.f ? 1.f : (v < 0.f ? a : 0.5f * a); }

template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_fwd_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, int per_sample,
                                                                     const float* __restrict__ alpha,
                                                                     float* __restrict__ y, int N, int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        const int n = (int)(i / ((int64_t)HW * CV));
        const float* gp = gamma + (per_sample ? (int64_t)n * C : 0) + c;
        const float* bp = beta + (per_sample ? (int64_t)n * C : 0) + c;
        float xv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gp, ga);
        loadv<VEC>(bp, be);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float inv = rs[j] * ga[j];                  // tf.nn.batch_normalization: inv = rsqrt(var+eps)*scale
            float v = xv[j] * inv + (be[j] - mu[j] * inv);    //   x*inv + (offset - mean*inv)
            out[j] = alpha ? prelu_f(v, al[j]) : v;
        }
        if constexpr (VEC == 4)
            stg4(y + i * 4, make_float4(out[0], out[1], out[2], out[VEC - 1]));
        else
            y[i] = out[0];
    }
}

struct BnBwdReduceFn {
    const float *x, *dy, *mean, *rstd, *gamma, *beta, *alpha;
    int per_sample, HW, C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int n, int64_t r, int c, float (&acc)[3][VEC]) const {
        const int64_t off = ((int64_t)n * HW + r) * C + c;
        float xv[VEC], dv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC];
        loadv<VEC>(x + off, xv);
        loadv<VEC>(dy + off, dv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gamma + (per_sample ? (int64_t)n * C : 0) + c, ga);
        loadv<VEC>(beta + (per_sample ? (int64_t)n * C : 0) + c, be);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float pre = xh * ga[j] + be[j];
            const float g = alpha ? dv[j] * prelu_d(pre, al[j]) : dv[j];
            acc[0][j] += g;
            acc[1][j] += g * xh;
            acc[2][j] += alpha ? dv[j] * fminf(pre, 0.f) : 0.f;
        }
    }
};

template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_bwd_dx_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, int per_sample,
    const float* __restrict__ alpha, const float* __restrict__ cm, float* __restrict__ dx, int N, int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        const int n = (int)(i / ((int64_t)HW * CV));
        float xv[VEC], dv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC], m1[VEC], m2[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(dy + i * VEC, dv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gamma + (per_sample ? (int64_t)n * C : 0) + c, ga);
        loadv<VEC>(beta + (per_sample ? (int64_t)n * C : 0) + c, be);
        loadv<VEC>(cm + c, m1);
        loadv<VEC>(cm + C + c, m2);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float pre = xh * ga[j] + be[j];
            const float g = alpha ? dv[j] * prelu_d(pre, al[j]) : dv[j];
            out[j] = rs[j] * (g * ga[j] - m1[j] - xh * m2[j]);
        }
        if constexpr (VEC == 4)
            stg4(dx + i * 4, make_float4(out[0], out[1], out[2], out[VEC - 1]));
        else
            dx[i] = out[0];
    }
}

// 64 channels x 4 sample lanes per block: a thread walks every 4th sample with 8 loads in flight (one thread per channel
// walking all N samples was N / 8 dependent memory round trips: 36 us per launch at N = 256, 11 launches per iteration)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                              int per_sample, double count, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                              float* __restrict__ cm, int N, int C) {
    __shared__ double red[5][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const bool live = c < C;
    const float* p0 = part;
    const float* p1 = part + (int64_t)N * C;
    const float* p2 = part + 2 * (int64_t)N * C;
    double s0 = 0, s1 = 0, s2 = 0, g0 = 0, g1 = 0;
    if (live) {
        for (int n0 = ry; n0 < N; n0 += 32) {
            float a[8], b[8], d[8], ga[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + 4 * u < N ? n0 + 4 * u : N - 1;
                a[u] = p0[(int64_t)n * C + c];
                b[u] = p1[(int64_t)n * C + c];
                d[u] = p2[(int64_t)n * C + c];
                ga[u] = gamma[(per_sample ? (int64_t)n * C : 0) + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + 4 * u;
                if (n >= N) break;
                s0 += a[u];
                s1 += b[u];
                s2 += d[u];
                g0 += (double)ga[u] * a[u];
                g1 += (double)ga[u] * b[u];
                if (per_sample) {
                    dbeta[(int64_t)n * C + c] = a[u];
                    dgamma[(int64_t)n * C + c] = b[u];
                }
            }
        }
    }
    red[0][ry][cx] = s0; red[1][ry][cx] = s1; red[2][ry][cx] = s2; red[3][ry][cx] = g0; red[4][ry][cx] = g1;
    __syncthreads();
    if (ry != 0 || !live) return;
    s0 = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
    s1 = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
    s2 = (red[2][0][cx] + red[2][1][cx]) + (red[2][2][cx] + red[2][3][cx]);
    g0 = (red[3][0][cx] + red[3][1][cx]) + (red[3][2][cx] + red[3][3][cx]);
    g1 = (red[4][0][cx] + red[4][1][cx]) + (red[4][2][cx] + red[4][3][cx]);
    if (!per_sample) {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
    }
    if (dalpha) dalpha[c] = (float)s2;
    // local (per-rank) numerators; caller all-reduces cm for cross-replica BN before the dx pass
    cm[c] = (float)(g0 / count);
    cm[C + c] = (float)(g1 / count);
}

// ------------------------------------------------------------------------------------------
// stand-alone PReLU
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void prelu_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ alpha, float* __restrict__ y,
                                                              int64_t total_v, int C) {
    const int CV = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total_v; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        float xv[VEC], al[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(alpha + c, al);
        if constexpr (VEC == 4)
            stg4(y + i * 4, make_float4(prelu_f(xv[0], al[0]), prelu_f(xv[1], al[1]), prelu_f(xv[2], al[2]),
                                        prelu_f(xv[VEC - 1], al[VEC - 1])));
        else
            y[i] = prelu_f(xv[0], al[0]);
    }
}

template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void prelu_bwd_dx_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 const float* __restrict__ alpha,
                                                                 float* __restrict__ dx, int64_t total_v, int C) {
    const int CV = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total_v; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        float xv[VEC], dv[VEC], al[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(dy + i * VEC, dv);
        loadv<VEC>(alpha + c, al);
        if constexpr (VEC == 4)
            stg4(dx + i * 4, make_float4(dv[0] * prelu_d(xv[0], al[0]), dv[1] * prelu_d(xv[1], al[1]),
                                         dv[2] * prelu_d(xv[2], al[2]), dv[VEC - 1] * prelu_d(xv[VEC - 1], al[VEC - 1])));
        else
            dx[i] = dv[0] * prelu_d(xv[0], al[0]);
    }
}

struct PreluDalphaFn {
    const float *x, *dy;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float xv[VEC], dv[VEC];
        loadv<VEC>(x + r * C + c, xv);
        loadv<VEC>(dy + r * C + c, dv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += dv[j] * fminf(xv[j], 0.f);
    }
};

struct BiasGradFn {
    const float* dy;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float dv[VEC];
        loadv<VEC>(dy + r * C + c, dv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += dv[j];
    }
};

// ------------------------------------------------------------------------------------------
// max pool 2x2 / global sum pool
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int N, int H, int W, int C) {
    const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const float* p = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + cv * VEC;
        float a[VEC], b[VEC], c_[VEC], d[VEC], o[VEC];
        loadv<VEC>(p, a);
        loadv<VEC>(p + C, b);
        loadv<VEC>(p + (int64_t)W * C, c_);
        loadv<VEC>(p + (int64_t)W * C + C, d);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c_[j], d[j]));
        if constexpr (VEC == 4)
            stg4(y + i * 4, make_float4(o[0], o[1], o[2], o[VEC - 1]));
        else
            y[i] = o[0];
    }
}

// 2x2 box reduce: y[n,ho,wo,c] = scale * (x[2ho,2wo] + x[2ho,2wo+1] + x[2ho+1,2wo] + x[2ho+1,2wo+1])
//   avg_pooling forward (scale 1/4, ops.py:512-514) and the backward of up_sample (scale 1)
template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void box2_down_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int N, int H, int W, int C, float scale) {
    const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const float* p = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + cv * VEC;
        float a[VEC], b[VEC], c_[VEC], d[VEC];
        loadv<VEC>(p, a);
        loadv<VEC>(p + C, b);
        loadv<VEC>(p + (int64_t)W * C, c_);
        loadv<VEC>(p + (int64_t)W * C + C, d);
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = scale * ((a[j] + b[j]) + (c_[j] + d[j]));
        if constexpr (VEC == 4)
            stg4(y + i * 4, make_float4(o[0], o[1], o[2], o[VEC - 1]));
        else
            y[i] = o[0];
    }
}

// 2x2 replicate: y[n,2hi+a,2wi+b,c] = scale * x[n,hi,wi,c]
//   up_sample forward (tf.image.resize_nearest_neighbor x2, scale 1, ops.py:516-519) and avg_pooling backward (1/4)
template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void box2_up_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                            int H, int W, int C, float scale) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wi = (int)(t % W);
        t /= W;
        const int hi = (int)(t % H);
        const int n = (int)(t / H);
        float a[VEC];
        loadv<VEC>(x + i * VEC, a);
        float* q = y + (((int64_t)n * 2 * H + 2 * hi) * 2 * W + 2 * wi) * C + cv * VEC;
        const int64_t row = (int64_t)2 * W * C;
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] *= scale;
        if constexpr (VEC == 4) {
            const float4 v = make_float4(a[0], a[1], a[2], a[VEC - 1]);
            stg4(q, v);
            stg4(q + C, v);
            stg4(q + row, v);
            stg4(q + row + C, v);
        } else {
            q[0] = a[0];
            q[C] = a[0];
            q[row] = a[0];
            q[row + C] = a[0];
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_bwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy, float* __restrict__ dx,
                                                                 int N, int H, int W, int C) {
    const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t base = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + cv * VEC;
        const int64_t o1 = C, o2 = (int64_t)W * C, o3 = (int64_t)W * C + C;
        float a[VEC], b[VEC], c_[VEC], d[VEC], g[VEC];
        loadv<VEC>(x + base, a);
        loadv<VEC>(x + base + o1, b);
        loadv<VEC>(x + base + o2, c_);
        loadv<VEC>(x + base + o3, d);
        loadv<VEC>(dy + i * VEC, g);
        float ra[VEC], rb[VEC], rc[VEC], rd[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float m = fmaxf(fmaxf(a[j], b[j]), fmaxf(c_[j], d[j]));
            // first maximum in window order (0,0),(0,1),(1,0),(1,1)
            const bool sa = a[j] == m;
            const bool sb = !sa && b[j] == m;
            const bool sc = !sa && !sb && c_[j] == m;
            const bool sd = !sa && !sb && !sc;
            ra[j] = sa ? g[j] : 0.f;
            rb[j] = sb ? g[j] : 0.f;
            rc[j] = sc ? g[j] : 0.f;
            rd[j] = sd ? g[j] : 0.f;
        }
        if constexpr (VEC == 4) {
            stg4(dx + base, make_float4(ra[0], ra[1], ra[2], ra[VEC - 1]));
            stg4(dx + base + o1, make_float4(rb[0], rb[1], rb[2], rb[VEC - 1]));
            stg4(dx + base + o2, make_float4(rc[0], rc[1], rc[2], rc[VEC - 1]));
            stg4(dx + base + o3, make_float4(rd[0], rd[1], rd[2], rd[VEC - 1]));
        } else {
            dx[base] = ra[0];
            dx[base + o1] = rb[0];
            dx[base + o2] = rc[0];
            dx[base + o3] = rd[0];
        }
    }
}

__global__ __launch_bounds__(EW_BLOCK) void sum_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int N, int HW, int C) {
    const int64_t total = (int64_t)N * C;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % C);
        const int n = (int)(i / C);
        float s = 0.f;
        for (int r = 0; r < HW; ++r) s += x[((int64_t)n * HW + r) * C + c];
        y[i] = s;
    }
}

__global__ __launch_bounds__(EW_BLOCK) void sum_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 int N, int HW, int C) {
    const int64_t total = (int64_t)N * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((int64_t)HW * C));
        dx[i] = dy[(int64_t)n * C + c];
    }
}

// ------------------------------------------------------------------------------------------
// softmax over rows of <= 1024 columns: one wave per row, row held in registers
// ------------------------------------------------------------------------------------------
#define SM_MAXPER 16
__global__ __launch_bounds__(EW_BLOCK) void softmax_fwd_kernel(const float* __restrict__ s, float* __restrict__ p,
                                                                int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* in = s + r * cols;
        float v[SM_MAXPER];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < cols ? in[c] : -INFINITY;
            mx = fmaxf(mx, v[j]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            v[j] = (lane + 64 * j) < cols ? __expf(v[j] - mx) : 0.f;
            sum += v[j];
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float* o = p + r * cols;
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) o[c] = v[j] * inv;
        }
    }
}

__global__ __launch_bounds__(EW_BLOCK) void softmax_bwd_kernel(const float* __restrict__ p,
                                                                const float* __restrict__ dp, float* __restrict__ ds,
                                                                int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nwaves) {
        float pv[SM_MAXPER], dv[SM_MAXPER];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            pv[j] = c < cols ? p[r * cols + c] : 0.f;
            dv[j] = c < cols ? dp[r * cols + c] : 0.f;
            dot += pv[j] * dv[j];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) ds[r * cols + c] = pv[j] * (dv[j] - dot);
        }
    }
}

// Second-order support (gradient penalty, BigGAN.py:717-742): derivative of the softmax tangent map
//   pdot = p * (sdot - sum_j p_j sdot_j)   (the same arithmetic as softmax_bwd_kernel)
// w.r.t. its two inputs, given g = dL/dpdot:
//   dsdot = p * (g - u) ;  dp = g * (sdot - t) - u * sdot ;  t = sum p*sdot, u = sum g*p
__global__ __launch_bounds__(EW_BLOCK) void softmax_tangent_bwd_kernel(const float* __restrict__ p,
                                                                        const float* __restrict__ sdot,
                                                                        const float* __restrict__ g,
                                                                        float* __restrict__ dp,
                                                                        float* __restrict__ dsdot, int64_t rows,
                                                                        int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nwaves) {
        float pv[SM_MAXPER], sv[SM_MAXPER], gv[SM_MAXPER];
        float t = 0.f, u = 0.f;
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            pv[j] = c < cols ? p[r * cols + c] : 0.f;
            sv[j] = c < cols ? sdot[r * cols + c] : 0.f;
            gv[j] = c < cols ? g[r * cols + c] : 0.f;
            t += pv[j] * sv[j];
            u += gv[j] * pv[j];
        }
        t = wave_sum(t);
        u = wave_sum(u);
#pragma unroll
        for (int j = 0; j < SM_MAXPER; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) {
                dsdot[r * cols + c] = pv[j] * (gv[j] - u);
                dp[r * cols + c] = gv[j] * (sv[j] - t) - u * sv[j];
            }
        }
    }
}

// tangent of the 2x2 max pool: y = t at the first maximum of x's window (the position maxpool2_bwd routes to)
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_gather_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                                    float* __restrict__ y, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % C);
        int64_t q = i / C;
        const int wo = (int)(q % Wo);
        q /= Wo;
        const int ho = (int)(q % Ho);
        const int n = (int)(q / Ho);
        const int64_t base = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c;
        const int64_t o1 = C, o2 = (int64_t)W * C, o3 = (int64_t)W * C + C;
        const float a = x[base], b = x[base + o1], c_ = x[base + o2], d = x[base + o3];
        const float m = fmaxf(fmaxf(a, b), fmaxf(c_, d));
        const int64_t sel = a == m ? 0 : (b == m ? o1 : (c_ == m ? o2 : o3));
        y[i] = t[base + sel];
    }
}

// d(alpha) of the PReLU tangent map ydot = xdot * prelu'(x): sum over rows of dy * xdot * [x < 0] (1/2 at x == 0)
struct PreluTangentDalphaFn {
    const float *x, *xdot, *dy;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float xv[VEC], tv[VEC], dv[VEC];
        loadv<VEC>(x + r * C + c, xv);
        loadv<VEC>(xdot + r * C + c, tv);
        loadv<VEC>(dy + r * C + c, dv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += dv[j] * tv[j] * (xv[j] < 0.f ? 1.f : (xv[j] == 0.f ? 0.5f : 0.f));
    }
};

// ------------------------------------------------------------------------------------------
// small elementwise family
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_BLOCK) void axpby_kernel(const float* __restrict__ x, float a, float* __restrict__ y,
                                                          float b, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float4 xv = ldg4(x + i * 4), yv;
        if (b != 0.f) {
            yv = ldg4(y + i * 4);
            yv = make_float4(a * xv.x + b * yv.x, a * xv.y + b * yv.y, a * xv.z + b * yv.z, a * xv.w + b * yv.w);
        } else {
            yv = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
        }
        stg4(y + i * 4, yv);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = b != 0.f ? a * x[i] + b * y[i] : a * x[i];
}

__global__ __launch_bounds__(EW_BLOCK) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ y, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float4 av = ldg4(a + i * 4), bv = ldg4(b + i * 4);
        stg4(y + i * 4, make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(EW_BLOCK) void scale_add_kernel(const float* __restrict__ o,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ x, float* __restrict__ y,
                                                              int64_t n) {
    const float g = *gamma;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float4 ov = ldg4(o + i * 4), xv = ldg4(x + i * 4);
        stg4(y + i * 4, make_float4(g * ov.x + xv.x, g * ov.y + xv.y, g * ov.z + xv.z, g * ov.w + xv.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = g * o[i] + x[i];
}

__global__ __launch_bounds__(EW_BLOCK) void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ sp,
                                                              float* __restrict__ y, int64_t n) {
    const float s = *sp;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = s * x[i];
}

__global__ __launch_bounds__(EW_BLOCK) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* out, int64_t n) {
    __shared__ float sh[4];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float4 av = ldg4(a + i * 4), bv = ldg4(b + i * 4);
        s += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        s += a[i] * b[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(EW_BLOCK) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = tanhf(x[i]);
}

__global__ __launch_bounds__(EW_BLOCK) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                             float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        dx[i] = dy[i] * (1.f - y[i] * y[i]);
}


// ==========================================================================================
// Typed variants for the bf16-resident data path (BASELINE configs 3-5): the same arithmetic in fp32 registers,
// activation tensors fp32 or bf16 in HBM (8-byte bf16x4 / 16-byte float4 accesses per thread).
// ==========================================================================================
template <int VEC>
__device__ __forceinline__ void loadv(const __bf16* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        v[0] = __builtin_bit_cast(float, r.x << 16);
        v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16);
        v[VEC - 1] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    } else if constexpr (VEC == 8) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[VEC - 1] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = (float)p[j];
    }
}
__device__ __forceinline__ uint32_t bf16_pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
template <int VEC>
__device__ __forceinline__ void storev(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4)
        stg4(p, make_float4(v[0], v[1], v[2], v[VEC - 1]));
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) p[j] = v[j];
    }
}
template <int VEC>
__device__ __forceinline__ void storev(__bf16* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        uint2 r;
        r.x = bf16_pack2(v[0], v[1]);
        r.y = bf16_pack2(v[2], v[VEC - 1]);
        *reinterpret_cast<uint2*>(p) = r;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) p[j] = (__bf16)v[j];
    }
}

// Widen or narrow the middle dimension of an [outer][C][inner] view (Cs -> Cd entries), converting the element type.
//   mode 0  zero fill / truncate:  dst[c] = c < Cs ? src[c] : 0
//   mode 1  split (Cd >= 2 Cs):    dst[c] = hi = bf16(src[c]),  dst[Cs + c] = src[c] - hi,  rest 0
//   mode 2  duplicate (Cd >= 2 Cs): dst[c] = dst[Cs + c] = src[c],  rest 0
//   mode 3  fold (Cs >= 2 Cd):     dst[c] = src[c] + src[Cd + c]
// Modes 1-3 serve the 3-channel image layers of the bf16-resident path: the five padding channels a 3 -> 8 widening
// leaves idle carry the bf16 rounding residual of the thin operand (image, image gradient, thin-side weights), so the
// bf16 MFMA sees it to ~16 mantissa bits at no extra cost; the partner operand is duplicated, or the two partial
// results are folded.
template <class TS, class TD>
__global__ __launch_bounds__(EW_BLOCK) void pad_channels_kernel(const TS* __restrict__ src, TD* __restrict__ dst,
                                                                 int64_t total, int Cs, int Cd, int64_t inner, int mode) {
    for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
        const int64_t i = idx % inner, t = idx / inner;
        const int c = (int)(t % Cd);
        const int64_t o = t / Cd;
        const TS* sp = src + o * Cs * inner + i;
        float v = 0.f;
        if (mode == 3) {
            v = (float)sp[(int64_t)c * inner] + (float)sp[(int64_t)(c + Cd) * inner];
        } else if (c < Cs) {
            v = (float)sp[(int64_t)c * inner];
            if (mode == 1) v = (float)(__bf16)v;
        } else if (mode != 0 && c < 2 * Cs) {
            v = (float)sp[(int64_t)(c - Cs) * inner];
            if (mode == 1) v -= (float)(__bf16)v;
        }
        dst[idx] = (TD)v;
    }
}

// The image layers' two hot cases of pad_channels_kernel, one PIXEL per thread (the generic kernel spends two 64-bit
// divisions and a 2-byte store per element: 1 TB/s on a 230 MB job):
//   fp32 [pixels, Cs <= 4] -> bf16 [pixels, 8] (modes 0 / 1 / 2: zero fill, hi | lo split, duplicate), one 16-byte store;
//   fp32 [pixels, 8] -> TD [pixels, Cd <= 4] (mode 3: fold the two halves), two 16-byte loads.
__global__ __launch_bounds__(EW_BLOCK) void pad_pixels_to8_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                  int64_t pixels, int Cs, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * EW_BLOCK) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.f;
        const float* sp = src + i * Cs;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < Cs) {
                const float s = sp[c];
                if (mode == 1) {
                    const float hi = (float)(__bf16)s;
                    v[c] = hi;
                    v[c + Cs] = s - hi;
                } else {
                    v[c] = s;
                    if (mode == 2) v[c + Cs] = s;
                }
            }
        typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
        bf16x8v o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (__bf16)v[c];
        *reinterpret_cast<bf16x8v*>(dst + i * 8) = o;
    }
}
template <class TD>
__global__ __launch_bounds__(EW_BLOCK) void fold_pixels_from8_kernel(const float* __restrict__ src, TD* __restrict__ dst,
                                                                     int64_t pixels, int Cd) {
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float4 a = *reinterpret_cast<const float4*>(src + i * 8);
        const float4 b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < Cd) dst[i * Cd + c] = (TD)(v[c] + v[c + Cd]);
    }
}

template <class TX, class TY>
__global__ __launch_bounds__(EW_BLOCK) void cast_kernel(const TX* __restrict__ x, TY* __restrict__ y, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float v[4];
        loadv<4>(x + i * 4, v);
        storev<4>(y + i * 4, v);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK)
        y[i] = (TY)(float)x[i];
}

template <class TX>
struct BnStatsFnT {
    const TX* x;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[2][VEC]) const {
        float v[VEC];
        loadv<VEC>(x + r * C + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[0][j] += v[j];
            acc[1][j] += v[j] * v[j];
        }
    }
};

template <int VEC, class TX, class TY>
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_fwd_t_kernel(const TX* __restrict__ x,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, int per_sample,
                                                                       const float* __restrict__ alpha,
                                                                       TY* __restrict__ y, int N, int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        const int n = (int)(i / ((int64_t)HW * CV));
        const float* gp = gamma + (per_sample ? (int64_t)n * C : 0) + c;
        const float* bp = beta + (per_sample ? (int64_t)n * C : 0) + c;
        float xv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gp, ga);
        loadv<VEC>(bp, be);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float inv = rs[j] * ga[j];
            float v = xv[j] * inv + (be[j] - mu[j] * inv);
            out[j] = alpha ? prelu_f(v, al[j]) : v;
        }
        storev<VEC>(y + i * VEC, out);
    }
}

// ---- bf16 fast forms of batch-norm apply (+PReLU) and its input gradient --------------------------------------
// The generic kernels above spend two 64-bit divisions and five to nine 16-byte coefficient loads per 8 bytes of
// activation (measured r02: 2.9 - 3.3 TB/s of tensor traffic, ~10 % of a config-3 iteration).  Here a thread keeps ONE
// group of 8 channels for its whole walk - the grid stride is a multiple of C / 8 - so the per-channel coefficients
// sit in registers (reloaded only when a conditional batch norm moves to the next sample) and an item is one 16-byte
// load, 8 FMAs + PReLU, one 16-byte store.
__device__ __forceinline__ void bf16x8_load(const __bf16* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
    v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
}
__device__ __forceinline__ void bf16x8_store(__bf16* p, const float (&v)[8]) {
    uint4 r;
    r.x = bf16_pack2(v[0], v[1]); r.y = bf16_pack2(v[2], v[3]); r.z = bf16_pack2(v[4], v[5]); r.w = bf16_pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void f32x8_load(const float* p, float (&v)[8]) {
    const float4 a = ldg4(p), b = ldg4(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// rows = N * HW pixels of C = 8 CV channels; (gridDim.x * EW_BLOCK) % CV == 0
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_fwd_bf16x8_kernel(
    const __bf16* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int per_sample, const float* __restrict__ alpha,
    __bf16* __restrict__ y, int rows, int HW, int CV) {
    const int gtid = blockIdx.x * EW_BLOCK + threadIdx.x;
    const int cv = gtid % CV, c = cv * 8, C = CV * 8;
    const int rstep = (gridDim.x * EW_BLOCK) / CV;
    float mu[8], rs[8], al[8], inv[8], sh[8];
    f32x8_load(mean + c, mu);
    f32x8_load(rstd + c, rs);
    if (alpha) f32x8_load(alpha + c, al);
    int r_end = 0;                       // rows [.., r_end) share the coefficients in registers
    for (int r = gtid / CV; r < rows; r += rstep) {
        if (r >= r_end) {
            const int n = per_sample ? r / HW : 0;
            r_end = per_sample ? (n + 1) * HW : rows;
            float ga[8], be[8];
            f32x8_load(gamma + (int64_t)n * C * per_sample + c, ga);
            f32x8_load(beta + (int64_t)n * C * per_sample + c, be);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                inv[j] = rs[j] * ga[j];
                sh[j] = be[j] - mu[j] * inv[j];
            }
        }
        float xv[8], out[8];
        bf16x8_load(x + (int64_t)r * C + c, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = xv[j] * inv[j] + sh[j];
            out[j] = alpha ? prelu_f(v, al[j]) : v;
        }
        bf16x8_store(y + (int64_t)r * C + c, out);
    }
}

__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_bwd_dx_bf16x8_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, int per_sample,
    const float* __restrict__ alpha, const float* __restrict__ cm, __bf16* dx, const __bf16* add, int rows, int HW,
    int CV) {
    // add (may alias dx): dx = add + gradient - the sum with the other branch of a forked tensor, fused (ops.py:253/263)
    const int gtid = blockIdx.x * EW_BLOCK + threadIdx.x;
    const int cv = gtid % CV, c = cv * 8, C = CV * 8;
    const int rstep = (gridDim.x * EW_BLOCK) / CV;
    float mu[8], rs[8], al[8], m1[8], m2[8], ga[8], be[8];
    f32x8_load(mean + c, mu);
    f32x8_load(rstd + c, rs);
    f32x8_load(cm + c, m1);
    f32x8_load(cm + C + c, m2);
    if (alpha) f32x8_load(alpha + c, al);
    int r_end = 0;
    for (int r = gtid / CV; r < rows; r += rstep) {
        if (r >= r_end) {
            const int n = per_sample ? r / HW : 0;
            r_end = per_sample ? (n + 1) * HW : rows;
            f32x8_load(gamma + (int64_t)n * C * per_sample + c, ga);
            f32x8_load(beta + (int64_t)n * C * per_sample + c, be);
        }
        float xv[8], dv[8], out[8];
        bf16x8_load(x + (int64_t)r * C + c, xv);
        bf16x8_load(dy + (int64_t)r * C + c, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float pre = xh * ga[j] + be[j];
            const float g = alpha ? dv[j] * prelu_d(pre, al[j]) : dv[j];
            out[j] = rs[j] * (g * ga[j] - m1[j] - xh * m2[j]);
        }
        if (add) {
            float av[8];
            bf16x8_load(add + (int64_t)r * C + c, av);
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] += av[j];
        }
        bf16x8_store(dx + (int64_t)r * C + c, out);
    }
}

// stand-alone PReLU (ops.py:532) and its input gradient, same scheme
template <bool BWD>
__global__ __launch_bounds__(EW_BLOCK) void prelu_bf16x8_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                                 const float* __restrict__ alpha, __bf16* out,
                                                                 const __bf16* add, int rows, int CV) {
    const int gtid = blockIdx.x * EW_BLOCK + threadIdx.x;
    const int c = (gtid % CV) * 8, C = CV * 8;
    const int rstep = (gridDim.x * EW_BLOCK) / CV;
    float al[8];
    f32x8_load(alpha + c, al);
    for (int r = gtid / CV; r < rows; r += rstep) {
        float xv[8], dv[8], o[8];
        bf16x8_load(x + (int64_t)r * C + c, xv);
        if (BWD) bf16x8_load(dy + (int64_t)r * C + c, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = BWD ? dv[j] * prelu_d(xv[j], al[j]) : prelu_f(xv[j], al[j]);
        if (BWD && add) {
            float av[8];
            bf16x8_load(add + (int64_t)r * C + c, av);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += av[j];
        }
        bf16x8_store(out + (int64_t)r * C + c, o);
    }
}

// y = sa a + sb b on bf16 tensors, 16 bytes per operand and item
__global__ __launch_bounds__(EW_BLOCK) void lincomb_bf16x8_kernel(const __bf16* __restrict__ a, const float* sa_dev, float sa,
                                                                   const __bf16* __restrict__ b, float sb,
                                                                   __bf16* __restrict__ y, int64_t n8) {
    const float s = sa_dev ? *sa_dev : sa;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EW_BLOCK) {
        float av[8], bv[8], o[8];
        bf16x8_load(a + i * 8, av);
        if (b) {
            bf16x8_load(b + i * 8, bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s * av[j] + sb * bv[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s * av[j];
        }
        bf16x8_store(y + i * 8, o);
    }
}

// grid for the x8 kernels: enough blocks to fill the chip, a multiple of CV / gcd(CV, EW_BLOCK)
static inline int bn_x8_grid(int64_t rows, int CV) {
    int a = CV, b = EW_BLOCK;
    while (b) { const int t = a % b; a = b; b = t; }
    const int unit = CV / a;
    int64_t want = (rows * CV + EW_BLOCK - 1) / EW_BLOCK;
    if (want > 4096) want = 4096;
    if (want < 1) want = 1;
    return (int)((want + unit - 1) / unit * unit);
}

template <class TX, class TY>
struct BnBwdReduceFnT {
    const TX* x;
    const TY* dy;
    const float *mean, *rstd, *gamma, *beta, *alpha;
    int per_sample, HW, C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int n, int64_t r, int c, float (&acc)[3][VEC]) const {
        const int64_t off = ((int64_t)n * HW + r) * C + c;
        float xv[VEC], dv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC];
        loadv<VEC>(x + off, xv);
        loadv<VEC>(dy + off, dv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gamma + (per_sample ? (int64_t)n * C : 0) + c, ga);
        loadv<VEC>(beta + (per_sample ? (int64_t)n * C : 0) + c, be);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float pre = xh * ga[j] + be[j];
            const float g = alpha ? dv[j] * prelu_d(pre, al[j]) : dv[j];
            acc[0][j] += g;
            acc[1][j] += g * xh;
            acc[2][j] += alpha ? dv[j] * fminf(pre, 0.f) : 0.f;
        }
    }
};

template <int VEC, class TX, class TY>
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_act_bwd_dx_t_kernel(
    const TX* __restrict__ x, const TY* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, int per_sample,
    const float* __restrict__ alpha, const float* __restrict__ cm, TX* dx, const TX* add, int N, int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        const int n = (int)(i / ((int64_t)HW * CV));
        float xv[VEC], dv[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC], al[VEC], m1[VEC], m2[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(dy + i * VEC, dv);
        loadv<VEC>(mean + c, mu);
        loadv<VEC>(rstd + c, rs);
        loadv<VEC>(gamma + (per_sample ? (int64_t)n * C : 0) + c, ga);
        loadv<VEC>(beta + (per_sample ? (int64_t)n * C : 0) + c, be);
        loadv<VEC>(cm + c, m1);
        loadv<VEC>(cm + C + c, m2);
        if (alpha) loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mu[j]) * rs[j];
            const float pre = xh * ga[j] + be[j];
            const float g = alpha ? dv[j] * prelu_d(pre, al[j]) : dv[j];
            out[j] = rs[j] * (g * ga[j] - m1[j] - xh * m2[j]);
        }
        if (add) {
            float av[VEC];
            loadv<VEC>(add + i * VEC, av);
#pragma unroll
            for (int j = 0; j < VEC; ++j) out[j] += av[j];
        }
        storev<VEC>(dx + i * VEC, out);
    }
}

template <int VEC, class TX, class TY>
__global__ __launch_bounds__(EW_BLOCK) void prelu_fwd_t_kernel(const TX* __restrict__ x, const float* __restrict__ alpha,
                                                                TY* __restrict__ y, int64_t total_v, int C) {
    const int CV = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total_v; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        float xv[VEC], al[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[j] = prelu_f(xv[j], al[j]);
        storev<VEC>(y + i * VEC, out);
    }
}

template <int VEC, class TX, class TY>
__global__ __launch_bounds__(EW_BLOCK) void prelu_bwd_dx_t_kernel(const TX* __restrict__ x, const TY* __restrict__ dy,
                                                                   const float* __restrict__ alpha, TX* dx, const TX* add,
                                                                   int64_t total_v, int C) {
    const int CV = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total_v; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int c = (int)(i % CV) * VEC;
        float xv[VEC], dv[VEC], al[VEC], out[VEC];
        loadv<VEC>(x + i * VEC, xv);
        loadv<VEC>(dy + i * VEC, dv);
        loadv<VEC>(alpha + c, al);
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[j] = dv[j] * prelu_d(xv[j], al[j]);
        if (add) {
            float av[VEC];
            loadv<VEC>(add + i * VEC, av);
#pragma unroll
            for (int j = 0; j < VEC; ++j) out[j] += av[j];
        }
        storev<VEC>(dx + i * VEC, out);
    }
}

template <class TX, class TY>
struct PreluDalphaFnT {
    const TX* x;
    const TY* dy;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float xv[VEC], dv[VEC];
        loadv<VEC>(x + r * C + c, xv);
        loadv<VEC>(dy + r * C + c, dv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += dv[j] * fminf(xv[j], 0.f);
    }
};

template <class T>
struct BiasGradFnT {
    const T* dy;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float dv[VEC];
        loadv<VEC>(dy + r * C + c, dv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += dv[j];
    }
};

template <int VEC, class T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_fwd_t_kernel(const T* __restrict__ x, T* __restrict__ y, int N,
                                                                   int H, int W, int C) {
    const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const T* p = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + cv * VEC;
        float a[VEC], b[VEC], c_[VEC], d[VEC], o[VEC];
        loadv<VEC>(p, a);
        loadv<VEC>(p + C, b);
        loadv<VEC>(p + (int64_t)W * C, c_);
        loadv<VEC>(p + (int64_t)W * C + C, d);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c_[j], d[j]));
        storev<VEC>(y + i * VEC, o);
    }
}

template <int VEC, class T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_bwd_t_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                   T* __restrict__ dx, int N, int H, int W, int C) {
    const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t base = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + cv * VEC;
        const int64_t o1 = C, o2 = (int64_t)W * C, o3 = (int64_t)W * C + C;
        float a[VEC], b[VEC], c_[VEC], d[VEC], g[VEC];
        loadv<VEC>(x + base, a);
        loadv<VEC>(x + base + o1, b);
        loadv<VEC>(x + base + o2, c_);
        loadv<VEC>(x + base + o3, d);
        loadv<VEC>(dy + i * VEC, g);
        float ra[VEC], rb[VEC], rc[VEC], rd[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float m = fmaxf(fmaxf(a[j], b[j]), fmaxf(c_[j], d[j]));
            const bool sa = a[j] == m;
            const bool sb = !sa && b[j] == m;
            const bool sc = !sa && !sb && c_[j] == m;
            const bool sd = !sa && !sb && !sc;
            ra[j] = sa ? g[j] : 0.f;
            rb[j] = sb ? g[j] : 0.f;
            rc[j] = sc ? g[j] : 0.f;
            rd[j] = sd ? g[j] : 0.f;
        }
        storev<VEC>(dx + base, ra);
        storev<VEC>(dx + base + o1, rb);
        storev<VEC>(dx + base + o2, rc);
        storev<VEC>(dx + base + o3, rd);
    }
}

// y[n][c] = sum_hw x[n][hw][c]: one thread per (n, 4 channels), the HW rows streamed with 4 loads in flight
template <int VEC, class TX>
__global__ __launch_bounds__(EW_BLOCK) void sum_pool_fwd_t_kernel(const TX* __restrict__ x, float* __restrict__ y, int N,
                                                                   int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        const int n = (int)(i / CV);
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        for (int r = 0; r < HW; ++r) {
            float v[VEC];
            loadv<VEC>(x + ((int64_t)n * HW + r) * C + cv * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += v[j];
        }
        storev<VEC>(y + (int64_t)n * C + cv * VEC, s);
    }
}

template <int VEC, class TX>
__global__ __launch_bounds__(EW_BLOCK) void sum_pool_bwd_t_kernel(const float* __restrict__ dy, TX* __restrict__ dx,
                                                                   int N, int HW, int C) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int cv = (int)(i % CV);
        const int n = (int)(i / ((int64_t)HW * CV));
        float v[VEC];
        loadv<VEC>(dy + (int64_t)n * C + cv * VEC, v);
        storev<VEC>(dx + i * VEC, v);
    }
}

// y = sa * a + sb * b (sa, sb: host scalars or device scalars when the pointers are non-null), n % 4 == 0
template <class T>
__global__ __launch_bounds__(EW_BLOCK) void lincomb_t_kernel(const T* __restrict__ a, const float* sa_dev, float sa,
                                                              const T* __restrict__ b, float sb, T* __restrict__ y,
                                                              int64_t n4) {
    const float s = sa_dev ? *sa_dev : sa;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float av[4], bv[4], o[4];
        loadv<4>(a + i * 4, av);
        if (b) {
            loadv<4>(b + i * 4, bv);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = s * av[j] + sb * bv[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = s * av[j];
        }
        storev<4>(y + i * 4, o);
    }
}

template <class T>
__global__ __launch_bounds__(EW_BLOCK) void dot_t_kernel(const T* __restrict__ a, const T* __restrict__ b, float* out,
                                                          int64_t n4) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        float av[4], bv[4];
        loadv<4>(a + i * 4, av);
        loadv<4>(b + i * 4, bv);
        s += av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2] + av[3] * bv[3];
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// w / sigma for one [taps][R][Cc] weight plus its two bf16 packed copies (BgSnItem::pack_p / pack_t): 64 x 64 tiles
// of one tap through LDS so that both copies are written in 128-byte row segments
__device__ __forceinline__ void sn_normalize_pack_body(const float* __restrict__ w, float sigma,
                                                       float* __restrict__ wn, __bf16* __restrict__ pp,
                                                       __bf16* __restrict__ pt, int taps, int R, int Cc, int pld,
                                                       int bid, int nblocks, float (*tile)[65]) {
    const int tr = (R + 63) / 64, tc = (Cc + 63) / 64;
    const int ntiles = taps * tr * tc;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 element pairs x 8 rows per pass
    for (int tl = bid; tl < ntiles; tl += nblocks) {
        const int tap = tl / (tr * tc);
        const int rem = tl - tap * (tr * tc);
        const int r0 = (rem / tc) * 64, c0 = (rem % tc) * 64;
        const int64_t base = (int64_t)tap * R * Cc;
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int r = r0 + ty + 8 * ps, c = c0 + 2 * tx;
            float v0 = 0.f, v1 = 0.f;
            if (r < R && c < Cc) {            // Cc is even on this path
                const float2 wv = *reinterpret_cast<const float2*>(w + base + (int64_t)r * Cc + c);
                v0 = wv.x / sigma;
                v1 = wv.y / sigma;
                if (wn) *reinterpret_cast<float2*>(wn + base + (int64_t)r * Cc + c) = make_float2(v0, v1);
                *reinterpret_cast<uint32_t*>(pp + (int64_t)tap * R * pld + (int64_t)r * pld + c) = bf16_pack2(v0, v1);
            }
            tile[ty + 8 * ps][2 * tx] = v0;
            tile[ty + 8 * ps][2 * tx + 1] = v1;
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int c = c0 + ty + 8 * ps, r = r0 + 2 * tx;
            if (c < Cc && r < R)              // R is even on this path
                *reinterpret_cast<uint32_t*>(pt + base + (int64_t)c * R + r) =
                    bf16_pack2(tile[2 * tx][ty + 8 * ps], tile[2 * tx + 1][ty + 8 * ps]);
        }
    }
}

__global__ __launch_bounds__(EW_BLOCK) void weight_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ pp,
                                                                __bf16* __restrict__ pt, int taps, int R, int Cc) {
    __shared__ float tile[64][65];
    sn_normalize_pack_body(w, 1.0f, nullptr, pp, pt, taps, R, Cc, Cc, blockIdx.x, gridDim.x, tile);
}

// ------------------------------------------------------------------------------------------
// spectral norm (ops.py:718-747)
// scratch layout (floats): [0]=sum v_^2, [1]=sigma, [2]=rs_v, [3]=<G,Wn>; [4 .. 4+rows) = v_ ; then cols of u_raw
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sn_rowdot_body(const float* __restrict__ w, const float* __restrict__ u,
                                               float* __restrict__ vraw, double* ssv, int rows, int cols, int bid,
                                               int nblocks) {
    const int lane = threadIdx.x & 63;
    const int wave = bid * 4 + (threadIdx.x >> 6);
    const int nwaves = nblocks * 4;
    double ss = 0.0;
    // four rows per wave at a time: four independent 16-byte loads in flight per lane whatever the row length (a wave
    // walking ONE row of 768 floats has a single 1 KB load in flight per round trip: measured 1.9 TB/s over a network)
    for (int r0 = wave; r0 < rows; r0 += 4 * nwaves) {
        const float* wr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rk = r0 + k * nwaves;
            wr[k] = w + (int64_t)(rk < rows ? rk : r0) * cols;
        }
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if ((cols & 3) == 0) {
            int c = lane * 4;
            for (; c + 256 < cols; c += 512) {           // eight 16-byte loads of W in flight per lane
                const float4 b0 = ldg4(u + c), b1 = ldg4(u + c + 256);
                float4 a0[4], a1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0[k] = ldg4(wr[k] + c);
                    a1[k] = ldg4(wr[k] + c + 256);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    s[k] += (a0[k].x * b0.x + a0[k].y * b0.y + a0[k].z * b0.z + a0[k].w * b0.w) +
                            (a1[k].x * b1.x + a1[k].y * b1.y + a1[k].z * b1.z + a1[k].w * b1.w);
            }
            for (; c < cols; c += 256) {
                const float4 b = ldg4(u + c);
                float4 a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = ldg4(wr[k] + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += a[k].x * b.x + a[k].y * b.y + a[k].z * b.z + a[k].w * b.w;
            }
        } else {
            for (int c = lane; c < cols; c += 64) {
                const float b = u[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += wr[k][c] * b;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = wave_sum(s[k]);
            const int rk = r0 + k * nwaves;
            if (lane == 0 && rk < rows) {
                vraw[rk] = t;
                ss += (double)t * (double)t;
            }
        }
    }
    // fp64 accumulation: the sum is the same to ~1e-16 whatever the arrival order of the waves, so
    // the forward pass is reproducible from run to run
    if (ssv && lane == 0 && ss != 0.0) atomicAdd(ssv, ss);      // (multi-tensor form: ssv == nullptr, see its finalize)
}

__global__ __launch_bounds__(EW_BLOCK) void sn_rowdot_kernel(const float* __restrict__ w,
                                                              const float* __restrict__ u, float* __restrict__ vraw,
                                                              double* ssv, int rows, int cols) {
    sn_rowdot_body(w, u, vraw, ssv, rows, cols, blockIdx.x, gridDim.x);
}

// out[r] = sum_c W[r][c] * v[c]   (one wave per row; v == nullptr: plain row sums)
__global__ __launch_bounds__(EW_BLOCK) void gemv_rows_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                                              float* __restrict__ out, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int r = wave; r < rows; r += nwaves) {
        const float* wr = w + (int64_t)r * cols;
        float s = 0.f;
        if ((cols & 3) == 0) {
            for (int c = lane * 4; c < cols; c += 256) {
                const float4 a = ldg4(wr + c);
                if (v) {
                    const float4 b = ldg4(v + c);
                    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
                } else {
                    s += a.x + a.y + a.z + a.w;
                }
            }
        } else {
            for (int c = lane; c < cols; c += 64) s += v ? wr[c] * v[c] : wr[c];
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s;
    }
}

// uraw[c] += sum_{r in chunk} vraw[r] * W[r][c]
__device__ __forceinline__ void sn_colsum_body(const float* __restrict__ w, const float* __restrict__ vraw,
                                               double* uraw, int rows, int cols, int rows_per_block, int cb, int rc) {
    const int r0 = rc * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const int c = cb * EW_BLOCK + threadIdx.x;
    if (c >= cols) return;
    // 8 rows' loads in flight per thread (one dependent round trip per row otherwise)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* wp = w + c;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = wp[(int64_t)(r + u) * cols];
        s0 += vraw[r] * a[0] + vraw[r + 4] * a[4];
        s1 += vraw[r + 1] * a[1] + vraw[r + 5] * a[5];
        s2 += vraw[r + 2] * a[2] + vraw[r + 6] * a[6];
        s3 += vraw[r + 3] * a[3] + vraw[r + 7] * a[7];
    }
    for (; r < r1; ++r) s0 += vraw[r] * wp[(int64_t)r * cols];
    atomicAdd(&uraw[c], (double)((s0 + s1) + (s2 + s3)));
}

// split of the [rows, cols] column reduction into (column block, row chunk) work units
__host__ __device__ inline void sn_colsum_plan(int rows, int cols, int* cblocks, int* rpb, int* rchunks) {
    const int cb = (cols + EW_BLOCK - 1) / EW_BLOCK;
    int rch = (512 + cb - 1) / cb;
    int r = (rows + rch - 1) / rch;
    if (r < 8) r = 8;
    *cblocks = cb;
    *rpb = r;
    *rchunks = (rows + r - 1) / r;
}
// multi-tensor form: work units of 64 rows x 256 columns (64 KB), so that an item's share of the grid follows its size
#define SN_COLSUM_RPB 64
__device__ __forceinline__ int sn_colsum_units(int rows, int cols) {
    return ((cols + EW_BLOCK - 1) / EW_BLOCK) * ((rows + SN_COLSUM_RPB - 1) / SN_COLSUM_RPB);
}

__global__ __launch_bounds__(EW_BLOCK) void sn_colsum_kernel(const float* __restrict__ w,
                                                              const float* __restrict__ vraw, double* uraw, int rows,
                                                              int cols, int rows_per_block) {
    sn_colsum_body(w, vraw, uraw, rows, cols, rows_per_block, blockIdx.x, blockIdx.y);
}

// single block: norms, u_out, sigma.  scr (doubles): [0] = sum v_^2, [1] = sigma, [2] = rs_v
__device__ __forceinline__ void sn_finalize_body(double* scr, const double* uraw, float* u_out, float* sigma_out,
                                                 int cols, float* sh) {
    const float rs_v = rsqrtf(fmaxf((float)scr[0], 1e-12f));       // l2_normalize(v_)
    float ss = 0.f;
    for (int c = threadIdx.x; c < cols; c += EW_BLOCK) {
        const float t = (float)uraw[c] * rs_v;                     // u_ = v_hat W
        ss += t * t;
    }
    ss = block_sum_256(ss, sh);
    const float rs_u = rsqrtf(fmaxf(ss, 1e-12f));                  // l2_normalize(u_)
    for (int c = threadIdx.x; c < cols; c += EW_BLOCK) u_out[c] = (float)uraw[c] * rs_v * rs_u;
    if (threadIdx.x == 0) {
        const float sigma = ss * rs_u;                             // v_hat W u_hat^T = u_ . u_hat
        scr[1] = sigma;
        scr[2] = rs_v;
        *sigma_out = sigma;
    }
}

__global__ __launch_bounds__(EW_BLOCK) void sn_finalize_kernel(double* scr, const double* uraw, float* u_out,
                                                                float* sigma_out, int cols) {
    __shared__ float sh[4];
    sn_finalize_body(scr, uraw, u_out, sigma_out, cols, sh);
}

__device__ __forceinline__ void sn_normalize_body(const float* __restrict__ w, const double* scr, const float* vraw,
                                                  float* __restrict__ wn, float* v_out, int64_t n, int rows,
                                                  int bid, int nblocks) {
    const float sigma = (float)scr[1];
    const float rs_v = (float)scr[2];
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)nblocks * EW_BLOCK) {
        float4 a = ldg4(w + i * 4);
        stg4(wn + i * 4, make_float4(a.x / sigma, a.y / sigma, a.z / sigma, a.w / sigma));
    }
    for (int64_t i = n4 * 4 + (int64_t)bid * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)nblocks * EW_BLOCK)
        wn[i] = w[i] / sigma;
    for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < rows; i += (int64_t)nblocks * EW_BLOCK)
        v_out[i] = vraw[i] * rs_v;
}

__global__ __launch_bounds__(EW_BLOCK) void sn_normalize_kernel(const float* __restrict__ w, const double* scr,
                                                                 const float* vraw, float* __restrict__ wn,
                                                                 float* v_out, int64_t n, int rows) {
    sn_normalize_body(w, scr, vraw, wn, v_out, n, rows, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(EW_BLOCK) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                           const float* __restrict__ v, const float* sigma,
                                                           const float* dotp, float* __restrict__ dw, int rows,
                                                           int cols) {
    const float inv_sigma = 1.f / *sigma;
    const float d = *dotp;
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dw[i] = (g[i] - d * v[r] * u[c]) * inv_sigma;
    }
}

// ---- multi-tensor form: blockIdx.y = item of a device-resident BgSnItem table ------------------
struct SnMask { uint64_t w[4]; };
__device__ __forceinline__ bool sn_bit(const SnMask& m, int i) { return (m.w[i >> 6] >> (i & 63)) & 1ull; }

__device__ __forceinline__ double* sn_scr(const BgSnItem& it, char* ws) {
    return reinterpret_cast<double*>(ws + it.ws_offset);
}

// Work schedule of the multi-tensor kernels.  One launch covers every spectrally normalised weight of a network: 3 KB
// biases-of-biases up to 150 MB transposed-conv kernels.  With blockIdx.y = item and a fixed number of blocks per item
// the few big weights - all of the bytes - ran on 128 blocks each with one load in flight per wave (measured r02:
// 0.8 - 1.5 TB/s).  Here the grid is flat: every block derives the same table of work units per item (units follow the
// item's size, `kind` says how) and strides over the units.
#define SN_GRID 2048
#define SN_UNIT_ELEMS 16384          // 64 KB of fp32 per streaming unit
#define SN_UNIT_ELEMS_REDUCE 262144  // 1 MB per unit where every unit ends in an atomic on ONE address of its item
                                     // (same-address atomics retire at ~10 M/s: 2304 of them cost more than the pass)
__device__ __forceinline__ int sn_units_of(const BgSnItem& it, int kind) {
    const int64_t n = (int64_t)it.rows * it.cols;
    if (kind == 1) return sn_colsum_units(it.rows, it.cols);
    const int64_t per = kind == 2 ? SN_UNIT_ELEMS / 2 : kind == 3 ? SN_UNIT_ELEMS_REDUCE : kind == 4 ? 4 * SN_UNIT_ELEMS
                                                                                                  : SN_UNIT_ELEMS;
    const int64_t u = (n + per - 1) / per;
    return u < 1 ? 1 : (int)u;
}
// start[i] .. start[i + 1] = units of item i; returns the total
__device__ __forceinline__ int sn_sched_build(int* start, const BgSnItem* __restrict__ items, int n_items, int kind,
                                              const SnMask* enable) {
    for (int i = threadIdx.x; i < n_items; i += EW_BLOCK)
        start[i + 1] = (enable && !sn_bit(*enable, i)) ? 0 : sn_units_of(items[i], kind);
    if (threadIdx.x == 0) start[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i <= n_items; ++i) start[i] += start[i - 1];
    __syncthreads();
    return start[n_items];
}
__device__ __forceinline__ int sn_sched_find(const int* start, int n_items, int unit) {
    int lo = 0, hi = n_items;            // largest i with start[i] <= unit
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= unit) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(EW_BLOCK) void sn_batch_rowdot_kernel(const BgSnItem* __restrict__ items, int n_items,
                                                                    char* ws) {
    __shared__ int start[257];
    // 64 KB units and NO atomic: sum v_^2 is taken from v_ itself by the item's finalize block (a same-address atomic
    // per wave capped this pass at 1.9 TB/s with 1 MB units, and at 1.2 TB/s with small ones)
    const int total = sn_sched_build(start, items, n_items, 4, nullptr);      // 256 KB units: ~10 rows per wave
    for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
        const int j = sn_sched_find(start, n_items, unit);
        const BgSnItem it = items[j];
        double* scr = sn_scr(it, ws);
        float* vraw = reinterpret_cast<float*>(scr + 4 + it.cols);
        sn_rowdot_body(it.w, it.u, vraw, nullptr, it.rows, it.cols, unit - start[j], start[j + 1] - start[j]);
    }
}

__global__ __launch_bounds__(EW_BLOCK) void sn_batch_colsum_kernel(const BgSnItem* __restrict__ items, int n_items,
                                                                    char* ws) {
    __shared__ int start[257];
    const int total = sn_sched_build(start, items, n_items, 1, nullptr);
    for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
        const int j = sn_sched_find(start, n_items, unit);
        const BgSnItem it = items[j];
        double* scr = sn_scr(it, ws);
        float* vraw = reinterpret_cast<float*>(scr + 4 + it.cols);
        const int cblocks = (it.cols + EW_BLOCK - 1) / EW_BLOCK;
        const int u = unit - start[j];
        sn_colsum_body(it.w, vraw, scr + 4, it.rows, it.cols, SN_COLSUM_RPB, u % cblocks, u / cblocks);
    }
}

// write_v: also store v_hat (power-iteration-only calls: the normalisation that would store it runs elsewhere)
__global__ __launch_bounds__(EW_BLOCK) void sn_batch_finalize_kernel(const BgSnItem* __restrict__ items, char* ws,
                                                                      int write_v) {
    __shared__ float sh[4];
    const BgSnItem it = items[blockIdx.x];
    double* scr = sn_scr(it, ws);
    const float* vraw = reinterpret_cast<const float*>(scr + 4 + it.cols);
    double ssd = 0.0;
    for (int r = threadIdx.x; r < it.rows; r += EW_BLOCK) ssd += (double)vraw[r] * (double)vraw[r];
    const float ss = block_sum_256((float)ssd, sh);
    if (threadIdx.x == 0) scr[0] = (double)ss;
    __syncthreads();
    sn_finalize_body(scr, scr + 4, it.u, it.sigma, it.cols, sh);
    if (write_v) {
        __syncthreads();
        const float rs_v = (float)scr[2];
        for (int r = threadIdx.x; r < it.rows; r += EW_BLOCK) it.v[r] = vraw[r] * rs_v;
    }
}

// from_state: sigma is read from the item's sigma slot and v_hat is left alone - the power iteration ran elsewhere (on
// the rank that owns the weight under data parallelism) and its results arrived by an all-gather of sigma | u | v_hat
__global__ __launch_bounds__(EW_BLOCK) void sn_batch_normalize_kernel(const BgSnItem* __restrict__ items, int n_items,
                                                                       char* ws, int from_state) {
    __shared__ float tile[64][65];
    __shared__ int start[257];
    const int total = sn_sched_build(start, items, n_items, 2, nullptr);
    for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
        const int j = sn_sched_find(start, n_items, unit);
        const BgSnItem it = items[j];
        const int bid = unit - start[j], nb = start[j + 1] - start[j];
        double* scr = sn_scr(it, ws);
        const float* vraw = reinterpret_cast<const float*>(scr + 4 + it.cols);
        const float sigma = from_state ? *it.sigma : (float)scr[1];
        if (it.pack_p) {
            // conv / transposed-conv kernel of the bf16-resident path: w / sigma in fp32 plus the two bf16 packed copies
            sn_normalize_pack_body(it.w, sigma, it.w_norm, reinterpret_cast<__bf16*>(it.pack_p),
                                   reinterpret_cast<__bf16*>(it.pack_t), it.taps, it.rows / it.taps, it.cols,
                                   it.pack_p_ld > 0 ? it.pack_p_ld : it.cols, bid, nb, tile);
            if (!from_state) {
                const float rs_v = (float)scr[2];
                for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < it.rows; i += (int64_t)nb * EW_BLOCK)
                    it.v[i] = vraw[i] * rs_v;
            }
            continue;
        }
        if (from_state) {
            const int64_t n = (int64_t)it.rows * it.cols;
            for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)nb * EW_BLOCK)
                it.w_norm[i] = it.w[i] / sigma;
            continue;
        }
        sn_normalize_body(it.w, scr, vraw, it.w_norm, it.v, (int64_t)it.rows * it.cols, it.rows, bid, nb);
    }
}

// dots[item] = <g_wnorm, w_norm>
__global__ __launch_bounds__(EW_BLOCK) void sn_batch_dot_kernel(const BgSnItem* __restrict__ items, int n_items,
                                                                 double* dots, SnMask enable) {
    __shared__ float sh[4];
    __shared__ int start[257];
    const int total = sn_sched_build(start, items, n_items, 3, &enable);
    for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
        const int j = sn_sched_find(start, n_items, unit);
        const BgSnItem it = items[j];
        const int bid = unit - start[j], nb = start[j + 1] - start[j];
        const int64_t n = (int64_t)it.rows * it.cols;
        const int64_t n4 = n / 4;
        float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int64_t step = (int64_t)nb * EW_BLOCK;
        int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x;
        for (; i + 3 * step < n4; i += 4 * step) {          // eight 16-byte loads in flight per thread
            const float4 a0 = ldg4(it.g_wnorm + i * 4), b0 = ldg4(it.w_norm + i * 4);
            const float4 a1 = ldg4(it.g_wnorm + (i + step) * 4), b1 = ldg4(it.w_norm + (i + step) * 4);
            const float4 a2 = ldg4(it.g_wnorm + (i + 2 * step) * 4), b2 = ldg4(it.w_norm + (i + 2 * step) * 4);
            const float4 a3 = ldg4(it.g_wnorm + (i + 3 * step) * 4), b3 = ldg4(it.w_norm + (i + 3 * step) * 4);
            s += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
            s1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
            s2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
            s3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
        }
        for (; i < n4; i += step) {
            float4 av = ldg4(it.g_wnorm + i * 4), bv = ldg4(it.w_norm + i * 4);
            s += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
        }
        s = (s + s1) + (s2 + s3);
        for (int64_t t = n4 * 4 + (int64_t)bid * EW_BLOCK + threadIdx.x; t < n; t += step)
            s += it.g_wnorm[t] * it.w_norm[t];
        __syncthreads();                 // sh is reused across units
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0) atomicAdd(&dots[j], (double)s);
    }
}

// dw (+)= (g - dot * v^T u) / sigma
__global__ __launch_bounds__(EW_BLOCK) void sn_batch_bwd_kernel(const BgSnItem* __restrict__ items, int n_items,
                                                                 const double* dots, SnMask enable, SnMask accumulate) {
    __shared__ int start[257];
    const int total = sn_sched_build(start, items, n_items, 0, &enable);
    for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
    const int j = sn_sched_find(start, n_items, unit);
    const BgSnItem it = items[j];
    const int bid = unit - start[j], nb = start[j + 1] - start[j];
    const bool acc = sn_bit(accumulate, j);
    const float inv_sigma = 1.f / *it.sigma;
    const float d = (float)dots[j];
    const int cols = it.cols;
    const int64_t n = (int64_t)it.rows * cols;
    if ((cols & 3) == 0) {
        const int64_t n4 = n / 4;
        for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)nb * EW_BLOCK) {
            const int64_t e = i * 4;
            const int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
            const float4 g = ldg4(it.g_wnorm + e), u = ldg4(it.u + c);
            const float dv = d * it.v[r];
            float4 o = make_float4((g.x - dv * u.x) * inv_sigma, (g.y - dv * u.y) * inv_sigma,
                                   (g.z - dv * u.z) * inv_sigma, (g.w - dv * u.w) * inv_sigma);
            if (acc) {
                const float4 p = ldg4(it.dw + e);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            stg4(it.dw + e, o);
        }
    } else {
        for (int64_t i = (int64_t)bid * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)nb * EW_BLOCK) {
            const int r = (int)(i / cols), c = (int)(i % cols);
            const float o = (it.g_wnorm[i] - d * it.v[r] * it.u[c]) * inv_sigma;
            it.dw[i] = acc ? it.dw[i] + o : o;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------
// TF Adam + EMA over a flat arena
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_BLOCK) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ ema, float lr_arg, const float* __restrict__ lr_dev,
                                                         float b1, float b2,
                                                         float eps, float decay, float gscale, int64_t n) {
    const float lr_t = lr_dev ? *lr_dev : lr_arg;      // device scalar: the launch can be replayed from a HIP graph
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
        p[i] = pi;
        if (ema) ema[i] = decay * ema[i] + (1.f - decay) * pi;
    }
}

// Four elements per thread (16-byte loads and stores on up to five streams) and, with beta1 = 0 (the BASELINE configurations),
// no read of the first moment: m <- g exactly (0 * m + 1 * g for any finite m).  The scalar kernel above moved the 5 GB of a
// config-3 update at 3.1 TB/s.
template <bool B1ZERO>
__global__ __launch_bounds__(EW_BLOCK) void adam_x4_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ ema, float lr_arg,
                                                           const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                           float decay, float gscale, int64_t n4) {
    const float lr_t = lr_dev ? *lr_dev : lr_arg;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_BLOCK) {
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        const float4 v4 = reinterpret_cast<const float4*>(v)[i];
        const float4 p4 = reinterpret_cast<const float4*>(p)[i];
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f), e4 = m4;
        if (!B1ZERO) m4 = reinterpret_cast<const float4*>(m)[i];
        if (ema) e4 = reinterpret_cast<const float4*>(ema)[i];
        const float gi[4] = {g4.x * gscale, g4.y * gscale, g4.z * gscale, g4.w * gscale};
        const float mo[4] = {m4.x, m4.y, m4.z, m4.w}, vo[4] = {v4.x, v4.y, v4.z, v4.w}, po[4] = {p4.x, p4.y, p4.z, p4.w};
        const float eo[4] = {e4.x, e4.y, e4.z, e4.w};
        float mn[4], vn[4], pn[4], en[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mn[j] = B1ZERO ? gi[j] : b1 * mo[j] + (1.f - b1) * gi[j];
            vn[j] = b2 * vo[j] + (1.f - b2) * gi[j] * gi[j];
            pn[j] = po[j] - lr_t * mn[j] / (sqrtf(vn[j]) + eps);
            en[j] = decay * eo[j] + (1.f - decay) * pn[j];
        }
        reinterpret_cast<float4*>(m)[i] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vn[0], vn[1], vn[2], vn[3]);
        reinterpret_cast<float4*>(p)[i] = make_float4(pn[0], pn[1], pn[2], pn[3]);
        if (ema) reinterpret_cast<float4*>(ema)[i] = make_float4(en[0], en[1], en[2], en[3]);
    }
}

static void launch_adam(float* p, const float* g, float* m, float* v, float* ema, float lr_t, const float* lr_dev, float b1,
                        float b2, float eps, float decay, float gscale, int64_t n, hipStream_t s) {
    const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema;
    if (n % 4 == 0 && (al & 15) == 0) {
        if (b1 == 0.f)
            hipLaunchKernelGGL((adam_x4_kernel<true>), dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, s, p, g, m, v, ema, lr_t,
                               lr_dev, b1, b2, eps, decay, gscale, n / 4);
        else
            hipLaunchKernelGGL((adam_x4_kernel<false>), dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, s, p, g, m, v, ema, lr_t,
                               lr_dev, b1, b2, eps, decay, gscale, n / 4);
        return;
    }
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, p, g, m, v, ema, lr_t, lr_dev, b1, b2, eps, decay,
                       gscale, n);
}

struct PartialRowsFn {
    const float* part;
    int C;
    template <int VEC>
    __device__ __forceinline__ void operator()(int, int64_t r, int c, float (&acc)[1][VEC]) const {
        float v[VEC];
        loadv<VEC>(part + r * C + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] += v[j];
    }
};

int launch_partial_colsum(const float* part, double* sums, int64_t rows, int cols, hipStream_t s) {
    if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)cols, s) != hipSuccess) {
        set_error("partial column sums: memset failed");
        return BG_ERR_LAUNCH;
    }
    PartialRowsFn fn{part, cols};
    launch_colreduce<1>(fn, sums, (int64_t)0, rows, 1, cols, s);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" {

int bg_bn_stats(const float* x, double* sums, int64_t rows, int C, void* stream) {
    BG_REQUIRE(x && sums && rows > 0 && C > 0, "bg_bn_stats: bad argument");
    BnStatsFn fn{x, C};
    launch_colreduce<2>(fn, sums, (int64_t)C, rows, 1, C, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_finalize(const double* sums, double count, float eps, float momentum, int unbiased_moving_var, float* mean,
                   float* rstd, float* moving_mean, float* moving_var, int C, void* stream) {
    BG_REQUIRE(sums && mean && rstd && C > 0 && count > 0, "bg_bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, count, eps,
                       momentum, unbiased_moving_var, mean, rstd, moving_mean, moving_var, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_chan_dots3(const float* p, const float* q, const float* r, double* sums, int64_t rows, int C, void* stream) {
    BG_REQUIRE(p && q && sums && rows > 0 && C > 0, "bg_chan_dots3: bad argument");
    if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)C, as_stream(stream)) != hipSuccess) {
        set_error("bg_chan_dots3: memset failed");
        return BG_ERR_LAUNCH;
    }
    ChanDotsFn fn{p, q, r, C};
    launch_colreduce<3>(fn, sums, (int64_t)C, rows, 1, C, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_tangent_fwd_coefs(const double* sums, double count, const float* mean, const float* rstd, const float* gamma,
                            float* coefs, float* m12, int C, void* stream) {
    BG_REQUIRE(sums && mean && rstd && gamma && coefs && m12 && C > 0 && count > 0, "bg_bn_tangent_fwd_coefs: bad argument");
    hipLaunchKernelGGL(bn_tangent_fwd_coefs_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, count,
                       mean, rstd, gamma, coefs, m12, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_tangent_bwd_coefs(const double* sums, double count, const float* mean, const float* rstd, const float* gamma,
                            const float* m12, float* coefs_dxd, float* coefs_dx, float* dgamma, int C, void* stream) {
    BG_REQUIRE(sums && mean && rstd && gamma && m12 && coefs_dxd && coefs_dx && dgamma && C > 0 && count > 0,
               "bg_bn_tangent_bwd_coefs: bad argument");
    hipLaunchKernelGGL(bn_tangent_bwd_coefs_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, count,
                       mean, rstd, gamma, m12, coefs_dxd, coefs_dx, dgamma, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_chan_lincomb3(const float* p, const float* cp, const float* q, const float* cq, const float* r, const float* cr,
                     const float* c0, float* out, int64_t rows, int C, void* stream) {
    BG_REQUIRE(p && cp && q && cq && c0 && out && rows > 0 && C > 0 && (!r || cr), "bg_chan_lincomb3: bad argument");
    const int64_t total = rows * C;
    hipLaunchKernelGGL(chan_lincomb3_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), p, cp, q, cq, r, cr,
                       c0, out, total, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_population(const float* pop_mean, const float* pop_var, float eps, float* mean, float* rstd, int C,
                     void* stream) {
    BG_REQUIRE(pop_mean && pop_var && mean && rstd && C > 0, "bg_bn_population: bad argument");
    hipLaunchKernelGGL(bn_population_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), pop_mean, pop_var,
                       eps, mean, rstd, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_renorm_coeffs(const double* sums, double count, float* ref_mean, float* ref_scale, int scale_is_var, float* weight,
                     float eps, float rmin, float rmax, float dmax, float decay, float fadein_decay, int update, float* r,
                     float* d, int C, void* stream) {
    BG_REQUIRE(sums && ref_mean && ref_scale && r && d && C > 0 && count > 0, "bg_renorm_coeffs: bad argument");
    BG_REQUIRE(rmin > 0 && rmax >= rmin && dmax >= 0, "bg_renorm_coeffs: bad clipping range");
    hipLaunchKernelGGL(renorm_coeffs_kernel, dim3(1), dim3(256), 0, as_stream(stream), sums, count, ref_mean, ref_scale,
                       scale_is_var, weight, eps, rmin, rmax, dmax, decay, fadein_decay, update, r, d, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_renorm_affine_fwd(const float* gamma, const float* beta, const float* r, const float* d, float* gamma_eff,
                         float* beta_eff, int64_t rows, int C, void* stream) {
    BG_REQUIRE(gamma && beta && r && d && gamma_eff && beta_eff && rows > 0 && C > 0, "bg_renorm_affine_fwd: bad argument");
    const int64_t total = rows * C;
    hipLaunchKernelGGL(renorm_affine_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), gamma, beta, r, d,
                       gamma_eff, beta_eff, total, C, 0);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_renorm_affine_bwd(const float* dgamma_eff, const float* dbeta_eff, const float* r, const float* d, float* dgamma,
                         int64_t rows, int C, void* stream) {
    BG_REQUIRE(dgamma_eff && dbeta_eff && r && d && dgamma && rows > 0 && C > 0, "bg_renorm_affine_bwd: bad argument");
    const int64_t total = rows * C;
    hipLaunchKernelGGL(renorm_affine_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), dgamma_eff,
                       dbeta_eff, r, d, dgamma, (float*)nullptr, total, C, 1);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        int per_sample, const float* alpha, float* y, int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && mean && rstd && gamma && beta && y && N > 0 && HW > 0 && C > 0, "bg_bn_apply_act_fwd: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((bn_apply_act_fwd_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream),
                           x, mean, rstd, gamma, beta, per_sample, alpha, y, N, HW, C);
    else
        hipLaunchKernelGGL((bn_apply_act_fwd_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x,
                           mean, rstd, gamma, beta, per_sample, alpha, y, N, HW, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int per_sample, const float* alpha, float* part,
                               int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && dy && mean && rstd && gamma && beta && part && N > 0 && HW > 0 && C > 0,
               "bg_bn_apply_act_bwd_reduce: bad argument");
    if (hipMemsetAsync(part, 0, sizeof(float) * 3 * (size_t)N * C, as_stream(stream)) != hipSuccess) {
        set_error("bg_bn_apply_act_bwd_reduce: memset failed");
        return BG_ERR_LAUNCH;
    }
    BnBwdReduceFn fn{x, dy, mean, rstd, gamma, beta, alpha, per_sample, HW, C};
    launch_colreduce<3>(fn, part, (int64_t)N * C, HW, N, C, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_bwd_finalize(const float* part, const float* gamma, int per_sample, double count, float* dgamma, float* dbeta,
                       float* dalpha, float* cm, int N, int C, void* stream) {
    BG_REQUIRE(part && gamma && dgamma && dbeta && cm && N > 0 && C > 0 && count > 0, "bg_bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, as_stream(stream), part, gamma,
                       per_sample, count, dgamma, dbeta, dalpha, cm, N, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_bwd_dx(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int per_sample, const float* alpha, const float* cm, float* dx, int N,
                           int HW, int C, void* stream) {
    BG_REQUIRE(x && dy && mean && rstd && gamma && beta && cm && dx && N > 0 && HW > 0 && C > 0,
               "bg_bn_apply_act_bwd_dx: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((bn_apply_act_bwd_dx_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0,
                           as_stream(stream), x, dy, mean, rstd, gamma, beta, per_sample, alpha, cm, dx, N, HW, C);
    else
        hipLaunchKernelGGL((bn_apply_act_bwd_dx_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                           x, dy, mean, rstd, gamma, beta, per_sample, alpha, cm, dx, N, HW, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_prelu_fwd(const float* x, const float* alpha, float* y, int64_t rows, int C, void* stream) {
    BG_REQUIRE(x && alpha && y && rows > 0 && C > 0, "bg_prelu_fwd: bad argument");
    const int64_t total = rows * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((prelu_fwd_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream), x,
                           alpha, y, total / 4, C);
    else
        hipLaunchKernelGGL((prelu_fwd_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, alpha,
                           y, total, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// d alpha of a PReLU on a tensor with C <= 4 channels (the discriminator's activation of the IMAGE, ops.py:299): the column
// skeleton would put 3 threads of a row-lane on a row (4-byte loads, 0.6 TB/s); here a thread owns whole pixels, 4 in flight.
__global__ __launch_bounds__(EW_BLOCK) void prelu_dalpha_thin_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dalpha, int64_t rows, int C) {
    __shared__ float red[4][EW_BLOCK];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    int64_t r = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    for (; r + 3 * stride < rows; r += 4 * stride) {
        float xv[4][4], gv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) {
                    xv[u][c] = x[(r + u * stride) * C + c];
                    gv[u][c] = dy[(r + u * stride) * C + c];
                }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) acc[c] += xv[u][c] < 0.f ? gv[u][c] * xv[u][c] : 0.f;
    }
    for (; r < rows; r += stride)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) {
                const float xs = x[r * C + c];
                acc[c] += xs < 0.f ? dy[r * C + c] * xs : 0.f;
            }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int s = EW_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
        __syncthreads();
    }
    if ((int)threadIdx.x < C) atomicAdd(&dalpha[threadIdx.x], red[threadIdx.x][0]);
}

int bg_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha, int64_t rows, int C,
                 void* stream) {
    BG_REQUIRE(x && dy && alpha && rows > 0 && C > 0, "bg_prelu_bwd: bad argument");
    const int64_t total = rows * C;
    if (dx) {
        if (C % 4 == 0)
            hipLaunchKernelGGL((prelu_bwd_dx_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream),
                               x, dy, alpha, dx, total / 4, C);
        else
            hipLaunchKernelGGL((prelu_bwd_dx_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x,
                               dy, alpha, dx, total, C);
        BG_LAUNCH_CHECK();
    }
    if (dalpha) {
        if (C <= 4 && rows >= (int64_t(1) << 16)) {
            hipLaunchKernelGGL(prelu_dalpha_thin_kernel, dim3(256), dim3(EW_BLOCK), 0, as_stream(stream), x, dy, dalpha, rows,
                               C);
            BG_LAUNCH_CHECK();
            return BG_OK;
        }
        PreluDalphaFn fn{x, dy, C};
        launch_colreduce<1>(fn, dalpha, (int64_t)C, rows, 1, C, as_stream(stream));
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

int bg_bias_grad(const float* dy, float* db, int64_t rows, int C, void* stream) {
    BG_REQUIRE(dy && db && rows > 0 && C > 0, "bg_bias_grad: bad argument");
    if (hipMemsetAsync(db, 0, sizeof(float) * (size_t)C, as_stream(stream)) != hipSuccess) {
        set_error("bg_bias_grad: memset failed");
        return BG_ERR_LAUNCH;
    }
    BiasGradFn fn{dy, C};
    launch_colreduce<1>(fn, db, (int64_t)C, rows, 1, C, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    BG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "bg_maxpool2_fwd: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((maxpool2_fwd_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream), x,
                           y, N, H, W, C);
    else
        hipLaunchKernelGGL((maxpool2_fwd_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, y,
                           N, H, W, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    BG_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0,
               "bg_maxpool2_bwd: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((maxpool2_bwd_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream), x,
                           dy, dx, N, H, W, C);
    else
        hipLaunchKernelGGL((maxpool2_bwd_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, dy,
                           dx, N, H, W, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_box2_down(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream) {
    BG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "bg_box2_down: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((box2_down_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream), x, y,
                           N, H, W, C, scale);
    else
        hipLaunchKernelGGL((box2_down_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, y, N,
                           H, W, C, scale);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_box2_up(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream) {
    BG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "bg_box2_up: bad argument");
    const int64_t total = (int64_t)N * H * W * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL((box2_up_kernel<4>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0, as_stream(stream), x, y, N,
                           H, W, C, scale);
    else
        hipLaunchKernelGGL((box2_up_kernel<1>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, y, N, H,
                           W, C, scale);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_softmax_fwd(const float* s, float* p, int64_t rows, int cols, void* stream) {
    BG_REQUIRE(s && p && rows > 0 && cols > 0, "bg_softmax_fwd: bad argument");
    if (cols > 64 * SM_MAXPER) {
        set_error("bg_softmax_fwd: cols=%d > %d unsupported", cols, 64 * SM_MAXPER);
        return BG_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(ew_grid(rows * 64)), dim3(EW_BLOCK), 0, as_stream(stream), s, p, rows,
                       cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int cols, void* stream) {
    BG_REQUIRE(p && dp && ds && rows > 0 && cols > 0, "bg_softmax_bwd: bad argument");
    if (cols > 64 * SM_MAXPER) {
        set_error("bg_softmax_bwd: cols=%d > %d unsupported", cols, 64 * SM_MAXPER);
        return BG_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ew_grid(rows * 64)), dim3(EW_BLOCK), 0, as_stream(stream), p, dp, ds,
                       rows, cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_softmax_tangent_bwd(const float* p, const float* sdot, const float* g, float* dp, float* dsdot, int64_t rows,
                           int cols, void* stream) {
    BG_REQUIRE(p && sdot && g && dp && dsdot && rows > 0 && cols > 0, "bg_softmax_tangent_bwd: bad argument");
    if (cols > 64 * SM_MAXPER) {
        set_error("bg_softmax_tangent_bwd: cols=%d > %d unsupported", cols, 64 * SM_MAXPER);
        return BG_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(softmax_tangent_bwd_kernel, dim3(ew_grid(rows * 64)), dim3(EW_BLOCK), 0, as_stream(stream), p, sdot,
                       g, dp, dsdot, rows, cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_maxpool2_gather(const float* x, const float* t, float* y, int N, int H, int W, int C, void* stream) {
    BG_REQUIRE(x && t && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0,
               "bg_maxpool2_gather: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(maxpool2_gather_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), x, t, y, N, H,
                       W, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_prelu_tangent_dalpha(const float* x, const float* xdot, const float* dy, float* dalpha, int64_t rows, int C,
                            void* stream) {
    BG_REQUIRE(x && xdot && dy && dalpha && rows > 0 && C > 0, "bg_prelu_tangent_dalpha: bad argument");
    if (hipMemsetAsync(dalpha, 0, sizeof(float) * (size_t)C, as_stream(stream)) != hipSuccess) {
        set_error("bg_prelu_tangent_dalpha: memset failed");
        return BG_ERR_LAUNCH;
    }
    PreluTangentDalphaFn fn{x, xdot, dy, C};
    launch_colreduce<1>(fn, dalpha, (int64_t)C, rows, 1, C, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_sum_pool_fwd(const float* x, float* y, int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "bg_sum_pool_fwd: bad argument");
    hipLaunchKernelGGL(sum_pool_fwd_kernel, dim3(ew_grid((int64_t)N * C)), dim3(EW_BLOCK), 0, as_stream(stream), x, y,
                       N, HW, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_sum_pool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream) {
    BG_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0, "bg_sum_pool_bwd: bad argument");
    hipLaunchKernelGGL(sum_pool_bwd_kernel, dim3(ew_grid((int64_t)N * HW * C)), dim3(EW_BLOCK), 0, as_stream(stream),
                       dy, dx, N, HW, C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_axpby(const float* x, float a, float* y, float b, int64_t n, void* stream) {
    BG_REQUIRE(x && y && n > 0, "bg_axpby: bad argument");
    BG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "bg_axpby: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, as_stream(stream), x, a, y, b, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
    BG_REQUIRE(a && b && y && n > 0, "bg_add: bad argument");
    BG_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)y & 15) == 0,
               "bg_add: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, as_stream(stream), a, b, y, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_scale_add(const float* o, const float* gamma_dev, const float* x, float* y, int64_t n, void* stream) {
    BG_REQUIRE(o && gamma_dev && x && y && n > 0, "bg_scale_add: bad argument");
    BG_REQUIRE(((uintptr_t)o & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
               "bg_scale_add: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(scale_add_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, as_stream(stream), o, gamma_dev,
                       x, y, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_dot(const float* a, const float* b, float* out_accum, int64_t n, void* stream) {
    BG_REQUIRE(a && b && out_accum && n > 0, "bg_dot: bad argument");
    BG_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "bg_dot: pointers must be 16-byte aligned");
    int grid = ew_grid(n / 4 + 1);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(EW_BLOCK), 0, as_stream(stream), a, b, out_accum, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_scale_dev(const float* x, const float* s_dev, float* y, int64_t n, void* stream) {
    BG_REQUIRE(x && s_dev && y && n > 0, "bg_scale_dev: bad argument");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, as_stream(stream), x, s_dev, y, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    BG_REQUIRE(x && y && n > 0, "bg_tanh_fwd: bad argument");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, as_stream(stream), x, y, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
    BG_REQUIRE(y && dy && dx && n > 0, "bg_tanh_bwd: bad argument");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, as_stream(stream), y, dy, dx, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gemv_rows(const float* w, const float* v, float* out, int rows, int cols, void* stream) {
    BG_REQUIRE(w && out && rows > 0 && cols > 0, "bg_gemv_rows: bad argument");
    BG_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)v & 15) == 0, "bg_gemv_rows: pointers must be 16-byte aligned");
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(grid), dim3(EW_BLOCK), 0, as_stream(stream), w, v, out, rows, cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

size_t bg_spectral_norm_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    // doubles: [4 + cols] (sum v^2, sigma, rs_v, pad, u accumulators) ; floats: [rows] (v_)
    return sizeof(double) * (size_t)(4 + cols) + sizeof(float) * (size_t)rows;
}

int bg_spectral_norm_fwd(const float* w, const float* u_in, float* u_out, float* v_out, float* sigma_out,
                         float* w_norm, int rows, int cols, void* ws, size_t ws_bytes, void* stream) {
    BG_REQUIRE(w && u_in && u_out && v_out && sigma_out && w_norm && rows > 0 && cols > 0,
               "bg_spectral_norm_fwd: bad argument");
    BG_REQUIRE(ws && ws_bytes >= bg_spectral_norm_workspace_bytes(rows, cols), "bg_spectral_norm_fwd: workspace too small");
    BG_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)w_norm & 15) == 0 && ((uintptr_t)u_in & 15) == 0 &&
                   ((uintptr_t)ws & 15) == 0,
               "bg_spectral_norm_fwd: pointers must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    double* scr = reinterpret_cast<double*>(ws);
    double* uraw = scr + 4;
    float* vraw = reinterpret_cast<float*>(uraw + cols);
    if (hipMemsetAsync(scr, 0, bg_spectral_norm_workspace_bytes(rows, cols), s) != hipSuccess) {
        set_error("bg_spectral_norm_fwd: memset failed");
        return BG_ERR_LAUNCH;
    }
    int g1 = (rows + 3) / 4;
    if (g1 > 1024) g1 = 1024;
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3(g1), dim3(EW_BLOCK), 0, s, w, u_in, vraw, scr, rows, cols);
    BG_LAUNCH_CHECK();
    const int cblocks = (cols + EW_BLOCK - 1) / EW_BLOCK;
    int rchunks = (512 + cblocks - 1) / cblocks;
    int rpb = (rows + rchunks - 1) / rchunks;
    if (rpb < 8) rpb = 8;
    rchunks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(sn_colsum_kernel, dim3(cblocks, rchunks), dim3(EW_BLOCK), 0, s, w, vraw, uraw, rows, cols, rpb);
    BG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(EW_BLOCK), 0, s, scr, uraw, u_out, sigma_out, cols);
    BG_LAUNCH_CHECK();
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, s, w, scr, vraw, w_norm, v_out,
                       n, rows);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_spectral_norm_bwd(const float* g_wnorm, const float* w_norm, const float* u_hat, const float* v_hat,
                         const float* sigma, float* dw, int rows, int cols, void* ws, size_t ws_bytes, void* stream) {
    BG_REQUIRE(g_wnorm && w_norm && u_hat && v_hat && sigma && dw && rows > 0 && cols > 0,
               "bg_spectral_norm_bwd: bad argument");
    BG_REQUIRE(ws && ws_bytes >= 16, "bg_spectral_norm_bwd: workspace too small");
    BG_REQUIRE(((uintptr_t)g_wnorm & 15) == 0 && ((uintptr_t)w_norm & 15) == 0,
               "bg_spectral_norm_bwd: pointers must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    float* scr = reinterpret_cast<float*>(ws);
    if (hipMemsetAsync(scr, 0, 16, s) != hipSuccess) {
        set_error("bg_spectral_norm_bwd: memset failed");
        return BG_ERR_LAUNCH;
    }
    const int64_t n = (int64_t)rows * cols;
    int grid = ew_grid(n / 4 + 1);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(EW_BLOCK), 0, s, g_wnorm, w_norm, scr + 3, n);
    BG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, g_wnorm, u_hat, v_hat, sigma, scr + 3, dw,
                       rows, cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

static bool sn_masks(const uint64_t* m, int n, SnMask* out, bool dflt) {
    for (int i = 0; i < 4; ++i) out->w[i] = dflt ? ~0ull : 0ull;
    if (m)
        for (int i = 0; i < (n + 63) / 64; ++i) out->w[i] = m[i];
    return true;
}

static int sn_batch_phases(const BgSnItem* items_dev, int n_items, void* ws, size_t ws_bytes, int phase, void* stream,
                           const char* who) {
    BG_REQUIRE(items_dev && n_items > 0 && ws && ws_bytes > 0, "%s: bad argument", who);
    BG_REQUIRE(((uintptr_t)ws & 15) == 0, "%s: workspace must be 16-byte aligned", who);
    BG_REQUIRE(n_items <= 256, "%s: 1..256 items per call", who);
    hipStream_t s = as_stream(stream);
    char* w8 = reinterpret_cast<char*>(ws);
    if (phase != BG_SN_NORMALIZE) {
        if (hipMemsetAsync(ws, 0, ws_bytes, s) != hipSuccess) {
            set_error("%s: memset failed", who);
            return BG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(sn_batch_rowdot_kernel, dim3(SN_GRID), dim3(EW_BLOCK), 0, s, items_dev, n_items, w8);
        BG_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_batch_colsum_kernel, dim3(SN_GRID), dim3(EW_BLOCK), 0, s, items_dev, n_items, w8);
        BG_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_batch_finalize_kernel, dim3(n_items), dim3(EW_BLOCK), 0, s, items_dev, w8,
                           phase == BG_SN_POWER ? 1 : 0);
        BG_LAUNCH_CHECK();
    }
    if (phase != BG_SN_POWER) {
        hipLaunchKernelGGL(sn_batch_normalize_kernel, dim3(SN_GRID), dim3(EW_BLOCK), 0, s, items_dev, n_items, w8,
                           phase == BG_SN_NORMALIZE ? 1 : 0);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

int bg_spectral_norm_batch_fwd(const BgSnItem* items_dev, int n_items, void* ws, size_t ws_bytes, void* stream) {
    return sn_batch_phases(items_dev, n_items, ws, ws_bytes, BG_SN_ALL, stream, "bg_spectral_norm_batch_fwd");
}

int bg_spectral_norm_batch_phase(const BgSnItem* items_dev, int n_items, void* ws, size_t ws_bytes, int phase,
                                 void* stream) {
    BG_REQUIRE(phase == BG_SN_ALL || phase == BG_SN_POWER || phase == BG_SN_NORMALIZE,
               "bg_spectral_norm_batch_phase: phase %d", phase);
    return sn_batch_phases(items_dev, n_items, ws, ws_bytes, phase, stream, "bg_spectral_norm_batch_phase");
}

int bg_spectral_norm_batch_bwd(const BgSnItem* items_dev, int n_items, const uint64_t* enable_mask,
                               const uint64_t* accumulate_mask, void* ws, size_t ws_bytes, void* stream) {
    BG_REQUIRE(items_dev && n_items > 0 && n_items <= 256, "bg_spectral_norm_batch_bwd: 1..256 items per call");
    BG_REQUIRE(ws && ws_bytes >= sizeof(double) * (size_t)n_items && ((uintptr_t)ws & 7) == 0,
               "bg_spectral_norm_batch_bwd: workspace too small");
    hipStream_t s = as_stream(stream);
    SnMask en, acc;
    sn_masks(enable_mask, n_items, &en, true);
    sn_masks(accumulate_mask, n_items, &acc, false);
    if (hipMemsetAsync(ws, 0, sizeof(double) * (size_t)n_items, s) != hipSuccess) {
        set_error("bg_spectral_norm_batch_bwd: memset failed");
        return BG_ERR_LAUNCH;
    }
    double* dots = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(sn_batch_dot_kernel, dim3(SN_GRID), dim3(EW_BLOCK), 0, s, items_dev, n_items, dots, en);
    BG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_batch_bwd_kernel, dim3(SN_GRID), dim3(EW_BLOCK), 0, s, items_dev, n_items, dots, en, acc);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_adam_tf_ema_step(float* p, const float* g, float* m, float* v, float* ema, float lr_t, float b1, float b2,
                        float eps, float ema_decay, float grad_scale, int64_t n, void* stream) {
    BG_REQUIRE(p && g && m && v && n > 0, "bg_adam_tf_ema_step: bad argument");
    launch_adam(p, g, m, v, ema, lr_t, nullptr, b1, b2, eps, ema_decay, grad_scale, n, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_adam_tf_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, const float* lr_t_dev, float b1,
                            float b2, float eps, float ema_decay, float grad_scale, int64_t n, void* stream) {
    BG_REQUIRE(p && g && m && v && lr_t_dev && n > 0, "bg_adam_tf_ema_step_dev: bad argument");
    launch_adam(p, g, m, v, ema, 0.f, lr_t_dev, b1, b2, eps, ema_decay, grad_scale, n, as_stream(stream));
    BG_LAUNCH_CHECK();
    return BG_OK;
}


// ------------------------------------------------------------------------------------------
// "_t" entry points: activation tensors fp32 or bf16 (BG_F32 / BG_BF16), parameters and statistics fp32
// ------------------------------------------------------------------------------------------
#define BG_DT_OK(d) ((d) == BG_F32 || (d) == BG_BF16)
// run EXPR with TX / TY bound to the element types named by (xd, yd)
#define BG_DISPATCH_XY(xd, yd, ...)                                          \
    do {                                                                     \
        if ((xd) == BG_F32 && (yd) == BG_F32) {                              \
            using TX = float; using TY = float; __VA_ARGS__;                 \
        } else if ((xd) == BG_F32) {                                         \
            using TX = float; using TY = __bf16; __VA_ARGS__;                \
        } else if ((yd) == BG_F32) {                                         \
            using TX = __bf16; using TY = float; __VA_ARGS__;                \
        } else {                                                             \
            using TX = __bf16; using TY = __bf16; __VA_ARGS__;               \
        }                                                                    \
    } while (0)
#define BG_DISPATCH_T(d, ...)                                                \
    do {                                                                     \
        if ((d) == BG_F32) { using T = float; __VA_ARGS__; }                 \
        else { using T = __bf16; __VA_ARGS__; }                              \
    } while (0)

int bg_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream) {
    BG_REQUIRE(x && y && n > 0 && BG_DT_OK(x_dtype) && BG_DT_OK(y_dtype), "bg_cast: bad argument");
    BG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "bg_cast: tensors must be 16-byte aligned");
    BG_DISPATCH_XY(x_dtype, y_dtype,
                   hipLaunchKernelGGL((cast_kernel<TX, TY>), dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, as_stream(stream),
                                      (const TX*)x, (TY*)y, n));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_pad_channels(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t outer, int Cs, int Cd, int64_t inner,
                    int mode, void* stream) {
    BG_REQUIRE(src && dst && outer > 0 && Cs > 0 && Cd > 0 && inner > 0 && BG_DT_OK(src_dtype) && BG_DT_OK(dst_dtype),
               "bg_pad_channels: bad argument");
    BG_REQUIRE(mode == 0 || ((mode == 1 || mode == 2) && Cd >= 2 * Cs) || (mode == 3 && Cs >= 2 * Cd),
               "bg_pad_channels: mode %d does not fit %d -> %d channels", mode, Cs, Cd);
    const int64_t total = outer * Cd * inner;
    if (inner == 1 && src_dtype == BG_F32 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        if (dst_dtype == BG_BF16 && Cd == 8 && Cs <= 4 && mode != 3) {            // image / image-gradient -> 8 channels
            hipLaunchKernelGGL(pad_pixels_to8_kernel, dim3(ew_grid(outer)), dim3(EW_BLOCK), 0, as_stream(stream),
                               (const float*)src, (__bf16*)dst, outer, Cs, mode);
            BG_LAUNCH_CHECK();
            return BG_OK;
        }
        if (mode == 3 && Cs == 8 && Cd <= 4) {                                      // fold of the fp32 8-channel result
            if (dst_dtype == BG_F32)
                hipLaunchKernelGGL((fold_pixels_from8_kernel<float>), dim3(ew_grid(outer)), dim3(EW_BLOCK), 0,
                                   as_stream(stream), (const float*)src, (float*)dst, outer, Cd);
            else
                hipLaunchKernelGGL((fold_pixels_from8_kernel<__bf16>), dim3(ew_grid(outer)), dim3(EW_BLOCK), 0,
                                   as_stream(stream), (const float*)src, (__bf16*)dst, outer, Cd);
            BG_LAUNCH_CHECK();
            return BG_OK;
        }
    }
    BG_DISPATCH_XY(src_dtype, dst_dtype,
                   hipLaunchKernelGGL((pad_channels_kernel<TX, TY>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                      as_stream(stream), (const TX*)src, (TY*)dst, total, Cs, Cd, inner, mode));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_weight_pack(const float* w, int taps, int rows_per_tap, int cols, void* pack_p, void* pack_t, void* stream) {
    BG_REQUIRE(w && pack_p && pack_t && taps > 0 && rows_per_tap > 0 && cols > 0, "bg_weight_pack: bad argument");
    BG_REQUIRE(rows_per_tap % 2 == 0 && cols % 2 == 0, "bg_weight_pack: inner dimensions must be even");
    const int tiles = taps * ((rows_per_tap + 63) / 64) * ((cols + 63) / 64);
    hipLaunchKernelGGL(weight_pack_kernel, dim3(tiles < 2048 ? tiles : 2048), dim3(EW_BLOCK), 0, as_stream(stream), w,
                       (__bf16*)pack_p, (__bf16*)pack_t, taps, rows_per_tap, cols);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_stats_t(const void* x, int x_dtype, double* sums, int64_t rows, int C, void* stream) {
    BG_REQUIRE(x && sums && rows > 0 && C > 0 && BG_DT_OK(x_dtype), "bg_bn_stats_t: bad argument");
    BG_DISPATCH_T(x_dtype, BnStatsFnT<T> fn{(const T*)x, C};
                  launch_colreduce<2, sizeof(T) == 2>(fn, sums, (int64_t)C, rows, 1, C, as_stream(stream)));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_fwd_t(const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int per_sample, const float* alpha, void* y, int y_dtype, int N, int HW,
                          int C, void* stream) {
    BG_REQUIRE(x && mean && rstd && gamma && beta && y && N > 0 && HW > 0 && C > 0 && BG_DT_OK(x_dtype) && BG_DT_OK(y_dtype),
               "bg_bn_apply_act_fwd_t: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    if (x_dtype == BG_BF16 && y_dtype == BG_BF16 && C % 8 == 0 && (int64_t)N * HW < (int64_t(1) << 31) &&
        ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL(bn_apply_act_fwd_bf16x8_kernel, dim3(bn_x8_grid((int64_t)N * HW, C / 8)), dim3(EW_BLOCK), 0,
                           as_stream(stream), (const __bf16*)x, mean, rstd, gamma, beta, per_sample, alpha, (__bf16*)y,
                           N * HW, HW, C / 8);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (C % 4 == 0)
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((bn_apply_act_fwd_t_kernel<4, TX, TY>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK),
                                          0, as_stream(stream), (const TX*)x, mean, rstd, gamma, beta, per_sample, alpha,
                                          (TY*)y, N, HW, C));
    else
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((bn_apply_act_fwd_t_kernel<1, TX, TY>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                          as_stream(stream), (const TX*)x, mean, rstd, gamma, beta, per_sample, alpha,
                                          (TY*)y, N, HW, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_bwd_reduce_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int per_sample,
                                 const float* alpha, float* part, int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && dy && mean && rstd && gamma && beta && part && N > 0 && HW > 0 && C > 0 && BG_DT_OK(x_dtype) &&
                   BG_DT_OK(y_dtype), "bg_bn_apply_act_bwd_reduce_t: bad argument");
    if (hipMemsetAsync(part, 0, sizeof(float) * 3 * (size_t)N * C, as_stream(stream)) != hipSuccess) {
        set_error("bg_bn_apply_act_bwd_reduce_t: memset failed");
        return BG_ERR_LAUNCH;
    }
    BG_DISPATCH_XY(x_dtype, y_dtype,
                   BnBwdReduceFnT<TX, TY> fn{(const TX*)x, (const TY*)dy, mean, rstd, gamma, beta, alpha, per_sample, HW, C};
                   launch_colreduce<3, sizeof(TX) == 2 && sizeof(TY) == 2>(fn, part, (int64_t)N * C, HW, N, C,
                                                                           as_stream(stream)));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_bn_apply_act_bwd_dx_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int per_sample, const float* alpha, const float* cm,
                             void* dx, const void* dx_add, int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && dy && mean && rstd && gamma && beta && cm && dx && N > 0 && HW > 0 && C > 0 && BG_DT_OK(x_dtype) &&
                   BG_DT_OK(y_dtype), "bg_bn_apply_act_bwd_dx_t: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    if (x_dtype == BG_BF16 && y_dtype == BG_BF16 && C % 8 == 0 && (int64_t)N * HW < (int64_t(1) << 31) &&
        ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dx_add & 15) == 0) {
        hipLaunchKernelGGL(bn_apply_act_bwd_dx_bf16x8_kernel, dim3(bn_x8_grid((int64_t)N * HW, C / 8)), dim3(EW_BLOCK), 0,
                           as_stream(stream), (const __bf16*)x, (const __bf16*)dy, mean, rstd, gamma, beta, per_sample,
                           alpha, cm, (__bf16*)dx, (const __bf16*)dx_add, N * HW, HW, C / 8);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (C % 4 == 0)
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((bn_apply_act_bwd_dx_t_kernel<4, TX, TY>), dim3(ew_grid(total / 4)),
                                          dim3(EW_BLOCK), 0, as_stream(stream), (const TX*)x, (const TY*)dy, mean, rstd,
                                          gamma, beta, per_sample, alpha, cm, (TX*)dx, (const TX*)dx_add, N, HW, C));
    else
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((bn_apply_act_bwd_dx_t_kernel<1, TX, TY>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                          as_stream(stream), (const TX*)x, (const TY*)dy, mean, rstd, gamma, beta,
                                          per_sample, alpha, cm, (TX*)dx, (const TX*)dx_add, N, HW, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_prelu_fwd_t(const void* x, int x_dtype, const float* alpha, void* y, int y_dtype, int64_t rows, int C,
                   void* stream) {
    BG_REQUIRE(x && alpha && y && rows > 0 && C > 0 && BG_DT_OK(x_dtype) && BG_DT_OK(y_dtype), "bg_prelu_fwd_t: bad argument");
    const int64_t total = rows * C;
    if (x_dtype == BG_BF16 && y_dtype == BG_BF16 && C % 8 == 0 && rows < (int64_t(1) << 31) && ((uintptr_t)x & 15) == 0 &&
        ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL((prelu_bf16x8_kernel<false>), dim3(bn_x8_grid(rows, C / 8)), dim3(EW_BLOCK), 0, as_stream(stream),
                           (const __bf16*)x, (const __bf16*)nullptr, alpha, (__bf16*)y, (const __bf16*)nullptr, (int)rows,
                           C / 8);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (C % 4 == 0)
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((prelu_fwd_t_kernel<4, TX, TY>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0,
                                          as_stream(stream), (const TX*)x, alpha, (TY*)y, total / 4, C));
    else
        BG_DISPATCH_XY(x_dtype, y_dtype,
                       hipLaunchKernelGGL((prelu_fwd_t_kernel<1, TX, TY>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                          as_stream(stream), (const TX*)x, alpha, (TY*)y, total, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_prelu_bwd_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* alpha, void* dx, float* dalpha,
                   const void* dx_add, int64_t rows, int C, void* stream) {
    BG_REQUIRE(x && dy && alpha && rows > 0 && C > 0 && BG_DT_OK(x_dtype) && BG_DT_OK(y_dtype), "bg_prelu_bwd_t: bad argument");
    const int64_t total = rows * C;
    const bool x8 = x_dtype == BG_BF16 && y_dtype == BG_BF16 && C % 8 == 0 && rows < (int64_t(1) << 31) &&
                    ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 &&
                    ((uintptr_t)dx_add & 15) == 0;
    if (dx && x8) {
        hipLaunchKernelGGL((prelu_bf16x8_kernel<true>), dim3(bn_x8_grid(rows, C / 8)), dim3(EW_BLOCK), 0, as_stream(stream),
                           (const __bf16*)x, (const __bf16*)dy, alpha, (__bf16*)dx, (const __bf16*)dx_add, (int)rows, C / 8);
        BG_LAUNCH_CHECK();
    } else if (dx) {
        if (C % 4 == 0)
            BG_DISPATCH_XY(x_dtype, y_dtype,
                           hipLaunchKernelGGL((prelu_bwd_dx_t_kernel<4, TX, TY>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK),
                                              0, as_stream(stream), (const TX*)x, (const TY*)dy, alpha, (TX*)dx,
                                              (const TX*)dx_add, total / 4, C));
        else
            BG_DISPATCH_XY(x_dtype, y_dtype,
                           hipLaunchKernelGGL((prelu_bwd_dx_t_kernel<1, TX, TY>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                              as_stream(stream), (const TX*)x, (const TY*)dy, alpha, (TX*)dx,
                                              (const TX*)dx_add, total, C));
        BG_LAUNCH_CHECK();
    }
    if (dalpha) {
        BG_DISPATCH_XY(x_dtype, y_dtype, PreluDalphaFnT<TX, TY> fn{(const TX*)x, (const TY*)dy, C};
                       launch_colreduce<1, sizeof(TX) == 2 && sizeof(TY) == 2>(fn, dalpha, 0, rows, 1, C,
                                                                               as_stream(stream)));
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

int bg_bias_grad_t(const void* dy, int dtype, float* db, int64_t rows, int C, void* stream) {
    BG_REQUIRE(dy && db && rows > 0 && C > 0 && BG_DT_OK(dtype), "bg_bias_grad_t: bad argument");
    if (hipMemsetAsync(db, 0, sizeof(float) * (size_t)C, as_stream(stream)) != hipSuccess) {
        set_error("bg_bias_grad_t: memset failed");
        return BG_ERR_LAUNCH;
    }
    BG_DISPATCH_T(dtype, BiasGradFnT<T> fn{(const T*)dy, C};
                  launch_colreduce<1, sizeof(T) == 2>(fn, db, 0, rows, 1, C, as_stream(stream)));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_maxpool2_fwd_t(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream) {
    BG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && BG_DT_OK(dtype),
               "bg_maxpool2_fwd_t: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (C % 4 == 0)
        BG_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_fwd_t_kernel<4, T>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0,
                                                as_stream(stream), (const T*)x, (T*)y, N, H, W, C));
    else
        BG_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_fwd_t_kernel<1, T>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                                as_stream(stream), (const T*)x, (T*)y, N, H, W, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_maxpool2_bwd_t(const void* x, const void* dy, void* dx, int dtype, int N, int H, int W, int C, void* stream) {
    BG_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 &&
                   BG_DT_OK(dtype), "bg_maxpool2_bwd_t: bad argument");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (C % 4 == 0)
        BG_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_bwd_t_kernel<4, T>), dim3(ew_grid(total / 4)), dim3(EW_BLOCK), 0,
                                                as_stream(stream), (const T*)x, (const T*)dy, (T*)dx, N, H, W, C));
    else
        BG_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_bwd_t_kernel<1, T>), dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                                as_stream(stream), (const T*)x, (const T*)dy, (T*)dx, N, H, W, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_sum_pool_fwd_t(const void* x, int x_dtype, float* y, int N, int HW, int C, void* stream) {
    BG_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && BG_DT_OK(x_dtype), "bg_sum_pool_fwd_t: bad argument");
    if (C % 4 == 0)
        BG_DISPATCH_T(x_dtype, hipLaunchKernelGGL((sum_pool_fwd_t_kernel<4, T>), dim3(ew_grid((int64_t)N * C / 4)),
                                                  dim3(EW_BLOCK), 0, as_stream(stream), (const T*)x, y, N, HW, C));
    else
        BG_DISPATCH_T(x_dtype, hipLaunchKernelGGL((sum_pool_fwd_t_kernel<1, T>), dim3(ew_grid((int64_t)N * C)),
                                                  dim3(EW_BLOCK), 0, as_stream(stream), (const T*)x, y, N, HW, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_sum_pool_bwd_t(const float* dy, void* dx, int x_dtype, int N, int HW, int C, void* stream) {
    BG_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0 && BG_DT_OK(x_dtype), "bg_sum_pool_bwd_t: bad argument");
    if (C % 4 == 0)
        BG_DISPATCH_T(x_dtype, hipLaunchKernelGGL((sum_pool_bwd_t_kernel<4, T>), dim3(ew_grid((int64_t)N * HW * C / 4)),
                                                  dim3(EW_BLOCK), 0, as_stream(stream), dy, (T*)dx, N, HW, C));
    else
        BG_DISPATCH_T(x_dtype, hipLaunchKernelGGL((sum_pool_bwd_t_kernel<1, T>), dim3(ew_grid((int64_t)N * HW * C)),
                                                  dim3(EW_BLOCK), 0, as_stream(stream), dy, (T*)dx, N, HW, C));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_lincomb_t(const void* a, const float* sa_dev, float sa, const void* b, float sb, void* y, int dtype, int64_t n,
                 void* stream) {
    BG_REQUIRE(a && y && n > 0 && n % 4 == 0 && BG_DT_OK(dtype), "bg_lincomb_t: bad argument (n %% 4 == 0)");
    if (dtype == BG_BF16 && n % 8 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
        (!b || ((uintptr_t)b & 15) == 0)) {
        hipLaunchKernelGGL(lincomb_bf16x8_kernel, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, as_stream(stream),
                           (const __bf16*)a, sa_dev, sa, (const __bf16*)b, sb, (__bf16*)y, n / 8);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    BG_DISPATCH_T(dtype, hipLaunchKernelGGL((lincomb_t_kernel<T>), dim3(ew_grid(n / 4)), dim3(EW_BLOCK), 0, as_stream(stream),
                                            (const T*)a, sa_dev, sa, (const T*)b, sb, (T*)y, n / 4));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_dot_t(const void* a, const void* b, int dtype, float* out_accum, int64_t n, void* stream) {
    BG_REQUIRE(a && b && out_accum && n > 0 && n % 4 == 0 && BG_DT_OK(dtype), "bg_dot_t: bad argument (n %% 4 == 0)");
    int blocks = ew_grid(n / 4);
    if (blocks > 512) blocks = 512;
    BG_DISPATCH_T(dtype, hipLaunchKernelGGL((dot_t_kernel<T>), dim3(blocks), dim3(EW_BLOCK), 0, as_stream(stream),
                                            (const T*)a, (const T*)b, out_accum, n / 4));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
