// augment_loss.hip - DiffAugment fused gather (DiffAugment_tf.py:8-73), hinge + flood losses
// (ops.py:788-797, 832-840, 847-848) and the orthogonal-cosine regulariser rows (utils.py:180-235).
#include "common.h"

namespace bg {

#define AL_BLOCK 256
#define DA_MAXC 4

// integer constants of rand_translation / rand_cutout for image size S (DiffAugment_tf.py:43,56)
__host__ __device__ inline int da_cutout_size(int S) { return (int)((float)S * 0.5f + 0.5f); }

struct DaArgs {
    const float *u_b, *u_s, *u_c;
    const int32_t *t_x, *t_y, *o_x, *o_y;
    int N, S, C, policy;
};

// color ops on one pixel (brightness :20-23, saturation :26-30); contrast needs the sample mean
__device__ __forceinline__ void da_color_pre(float (&v)[DA_MAXC], int C, float ub, float us) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < DA_MAXC; ++c)
        if (c < C) {
            v[c] += ub - 0.5f;
            m += v[c];
        }
    m /= (float)C;
    const float mag = us * 2.f;
#pragma unroll
    for (int c = 0; c < DA_MAXC; ++c)
        if (c < C) v[c] = (v[c] - m) * mag + m;
}

// per-sample sum of the brightness/saturation-adjusted image (reduce_mean of rand_contrast, :35)
__global__ __launch_bounds__(AL_BLOCK) void da_mean_kernel(const float* __restrict__ x, DaArgs a, double* sums) {
    __shared__ float sh[4];
    const int n = blockIdx.y;
    const int64_t px = (int64_t)a.S * a.S;
    const float ub = a.u_b[n], us = a.u_s[n];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; i < px; i += (int64_t)gridDim.x * AL_BLOCK) {
        const float* p = x + ((int64_t)n * px + i) * a.C;
        float v[DA_MAXC];
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c) v[c] = c < a.C ? p[c] : 0.f;
        da_color_pre(v, a.C, ub, us);
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < a.C) s += v[c];
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(&sums[n], (double)s);     // fp64: order-independent to ~1e-16
}

__device__ __forceinline__ bool da_cut(const DaArgs& a, int n, int i, int j) {
    // DiffAugment_tf.py:53-66: zero rows [max(0,o-cs/2), min(S-1,o-cs/2+cs-1)] x cols likewise
    const int cs = da_cutout_size(a.S);
    const int r0 = max(0, a.o_x[n] - cs / 2), r1 = min(a.S - 1, a.o_x[n] - cs / 2 + cs - 1);
    const int c0 = max(0, a.o_y[n] - cs / 2), c1 = min(a.S - 1, a.o_y[n] - cs / 2 + cs - 1);
    return i >= r0 && i <= r1 && j >= c0 && j <= c1;
}

__global__ __launch_bounds__(AL_BLOCK) void da_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, DaArgs a,
                                                           const double* sums) {
    const int64_t px = (int64_t)a.S * a.S;
    const int64_t total = (int64_t)a.N * px;
    const bool color = a.policy & 1, trans = a.policy & 2, cut = a.policy & 4;
    for (int64_t idx = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * AL_BLOCK) {
        const int n = (int)(idx / px);
        const int r = (int)(idx % px);
        const int i = r / a.S, j = r % a.S;
        float v[DA_MAXC];
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c) v[c] = 0.f;
        bool live = !(cut && da_cut(a, n, i, j));
        int si = i, sj = j;
        if (trans) {                                  // DiffAugment_tf.py:40-50: out[i,j] = x[i+t_x, j+t_y] or 0
            si = i + a.t_x[n];
            sj = j + a.t_y[n];
            live = live && si >= 0 && si < a.S && sj >= 0 && sj < a.S;
        }
        if (live) {
            const float* p = x + (((int64_t)n * a.S + si) * a.S + sj) * a.C;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c) v[c] = c < a.C ? p[c] : 0.f;
            if (color) {
                da_color_pre(v, a.C, a.u_b[n], a.u_s[n]);
                const float m = (float)(sums[n] / (double)(px * a.C));   // rand_contrast :33-37
                const float mag = a.u_c[n] + 0.5f;
#pragma unroll
                for (int c = 0; c < DA_MAXC; ++c)
                    if (c < a.C) v[c] = (v[c] - m) * mag + m;
            }
        }
        float* q = y + idx * a.C;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < a.C) q[c] = v[c];
    }
}

// gradient w.r.t. the colour-adjusted image x3 at source pixel (p,q): dy[p - t_x, q - t_y] * mask, or 0
__device__ __forceinline__ bool da_bwd_src(const DaArgs& a, int n, int p, int q, int& i, int& j) {
    i = p;
    j = q;
    if (a.policy & 2) {
        i = p - a.t_x[n];
        j = q - a.t_y[n];
        if (i < 0 || i >= a.S || j < 0 || j >= a.S) return false;
    }
    if ((a.policy & 4) && da_cut(a, n, i, j)) return false;
    return true;
}

__global__ __launch_bounds__(AL_BLOCK) void da_bwd_sum_kernel(const float* __restrict__ dy, DaArgs a, double* sums) {
    __shared__ float sh[4];
    const int n = blockIdx.y;
    const int64_t px = (int64_t)a.S * a.S;
    float s = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; r < px; r += (int64_t)gridDim.x * AL_BLOCK) {
        int i, j;
        if (da_bwd_src(a, n, (int)(r / a.S), (int)(r % a.S), i, j)) {
            const float* p = dy + (((int64_t)n * a.S + i) * a.S + j) * a.C;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c)
                if (c < a.C) s += p[c];
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) atomicAdd(&sums[n], (double)s);
}

__global__ __launch_bounds__(AL_BLOCK) void da_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, DaArgs a,
                                                           const double* sums) {
    const int64_t px = (int64_t)a.S * a.S;
    const int64_t total = (int64_t)a.N * px;
    const bool color = a.policy & 1;
    for (int64_t idx = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * AL_BLOCK) {
        const int n = (int)(idx / px);
        const int r = (int)(idx % px);
        float g[DA_MAXC];
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c) g[c] = 0.f;
        int i, j;
        if (da_bwd_src(a, n, r / a.S, r % a.S, i, j)) {
            const float* p = dy + (((int64_t)n * a.S + i) * a.S + j) * a.C;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c) g[c] = c < a.C ? p[c] : 0.f;
        }
        if (color) {
            // contrast: x3 = (x2 - m) c + m, m = mean_hwc(x2)  ->  dx2 = c dx3 + (1-c) mean_hwc(dx3)
            const float cm = a.u_c[n] + 0.5f;
            const float mg = (float)(sums[n] / (double)(px * a.C));
            float mc = 0.f;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c)
                if (c < a.C) {
                    g[c] = cm * g[c] + (1.f - cm) * mg;
                    mc += g[c];
                }
            // saturation: x2 = (x1 - mean_c) s + mean_c  ->  dx1 = s dx2 + (1-s) mean_c(dx2)
            mc /= (float)a.C;
            const float sm = a.u_s[n] * 2.f;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c)
                if (c < a.C) g[c] = sm * g[c] + (1.f - sm) * mc;
        }
        float* q = dx + idx * a.C;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < a.C) q[c] = g[c];
    }
}

// ------------------------------------------------------------------------------------------
// hinge + flood
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_BLOCK) void hinge_d_sums_kernel(const float* real, const float* fake, float* sums, int n) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += AL_BLOCK) {
        a += fmaxf(1.f - real[i], 0.f);
        b += fmaxf(1.f + fake[i], 0.f);
    }
    a = block_sum_256(a, sh);
    b = block_sum_256(b, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], a);
        atomicAdd(&sums[1], b);
    }
}

__global__ __launch_bounds__(AL_BLOCK) void hinge_d_grad_kernel(const float* real, const float* fake, const float* sums,
                                                                 double n_global, float flood, float* d_real,
                                                                 float* d_fake, float* loss_out, int n) {
    const float inv = (float)(1.0 / n_global);
    float L = sums[0] * inv + sums[1] * inv;                  // ops.py:788-792
    float sgn = 1.f;
    if (flood != 0.f) {                                       // ops.py:794-795, 847-848
        const float d = L - flood;
        sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        L = fabsf(d) + flood;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) *loss_out = L;
    for (int i = blockIdx.x * AL_BLOCK + threadIdx.x; i < n; i += gridDim.x * AL_BLOCK) {
        d_real[i] = (1.f - real[i]) > 0.f ? -sgn * inv : 0.f;
        d_fake[i] = (1.f + fake[i]) > 0.f ? sgn * inv : 0.f;
    }
}

__global__ __launch_bounds__(AL_BLOCK) void hinge_g_sums_kernel(const float* fake, float* sums, int n) {
    __shared__ float sh[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += AL_BLOCK) a += fake[i];
    a = block_sum_256(a, sh);
    if (threadIdx.x == 0) atomicAdd(&sums[0], a);
}

__global__ __launch_bounds__(AL_BLOCK) void hinge_g_grad_kernel(const float* sums, double n_global, float flood,
                                                                 float* d_fake, float* loss_out, int n) {
    const float inv = (float)(1.0 / n_global);
    float L = -sums[0] * inv;                                 // ops.py:832-833
    float sgn = 1.f;
    if (flood != 0.f) {                                       // ops.py:837-838
        const float d = L - flood;
        sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        L = fabsf(d) + flood;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) *loss_out = L;
    for (int i = blockIdx.x * AL_BLOCK + threadIdx.x; i < n; i += gridDim.x * AL_BLOCK) d_fake[i] = -sgn * inv;
}

// ------------------------------------------------------------------------------------------
// class-label loss (utils.py:366-369): scale * sum(sigmoid_cross_entropy_with_logits(t, x) * w[j])
//   ce = max(x,0) - x t + log1p(exp(-|x|)) ; d/dx = sigmoid(x) - t.   Single block: the sum is deterministic.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_BLOCK) void sigmoid_ce_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                               const float* __restrict__ w, float scale,
                                                               float* loss_out, float* __restrict__ dx, int B, int n) {
    __shared__ float sh[4];
    float acc = 0.f;
    const int total = B * n;
    for (int i = threadIdx.x; i < total; i += AL_BLOCK) {
        const float xi = x[i], ti = t[i], wi = w ? w[i % n] : 1.f;
        const float e = __expf(-fabsf(xi));
        acc += (fmaxf(xi, 0.f) - xi * ti + log1pf(e)) * wi;
        const float sig = xi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        dx[i] = (sig - ti) * wi * scale;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) *loss_out = acc * scale;
}

// ------------------------------------------------------------------------------------------
// ortho-cosine regulariser on the Gram matrix A [c,c]: one block per row
//   Ahat = l2n_rows(A); R[i,j] = (sum_k Ahat[i,k] - Ahat[i,j]) / sqrt(c-1); loss += scale/2 sum R^2
//   dR = scale R ; dAhat[i,j] = (sum_j' dR[i,j'] - dR[i,j]) / sqrt(c-1) ;
//   dA[i,j] = rn_i (dAhat[i,j] - Ahat[i,j] sum_k dAhat[i,k] Ahat[i,k])   (when sum A^2 >= 1e-12)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_BLOCK) void ortho_cosine_kernel(const float* __restrict__ A, float scale,
                                                                 float* loss_accum, float* __restrict__ dA, int c) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const float* a = A + (int64_t)i * c;
    float ss = 0.f, sa = 0.f;
    for (int k = threadIdx.x; k < c; k += AL_BLOCK) {
        const float t = a[k];
        ss += t * t;
        sa += t;
    }
    ss = block_sum_256(ss, sh);
    sa = block_sum_256(sa, sh);
    const bool clamped = ss < 1e-12f;
    const float rn = rsqrtf(fmaxf(ss, 1e-12f));
    const float r = sa * rn;                                   // sum_k Ahat[i,k]
    const float isq = c > 1 ? rsqrtf((float)(c - 1)) : 0.f;   // l2n of an all-zero row of (1 - I) is 0
    // sum_j R^2, sum_j R, sum_j R*Ahat
    float s2 = 0.f, s1 = 0.f, sra = 0.f;
    for (int k = threadIdx.x; k < c; k += AL_BLOCK) {
        const float ah = a[k] * rn;
        const float R = (r - ah) * isq;
        s2 += R * R;
        s1 += R;
        sra += R * ah;
    }
    s2 = block_sum_256(s2, sh);
    s1 = block_sum_256(s1, sh);
    sra = block_sum_256(sra, sh);
    if (threadIdx.x == 0) atomicAdd(loss_accum, 0.5f * scale * s2);
    if (!dA) return;
    // dAhat[j] = scale*isq*(s1 - R_j); T = sum_k dAhat[k]*Ahat[k] = scale*isq*(s1*r - sra)
    const float T = scale * isq * (s1 * r - sra);
    float* o = dA + (int64_t)i * c;
    for (int k = threadIdx.x; k < c; k += AL_BLOCK) {
        const float ah = a[k] * rn;
        const float R = (r - ah) * isq;
        const float dah = scale * isq * (s1 - R);
        o[k] = clamped ? rn * dah : rn * (dah - ah * T);
    }
}

// ------------------------------------------------------------------------------------------
// ortho-cosine regulariser, low-rank form for wide kernels (rows < c, e.g. first/dense2 [184, 16384]):
// with G = W W^T, s = W 1, P = G W (all O(rows^2 c)) the c x c Gram matrix is never formed:
//   n_i^2 = q_i = w_i^T G w_i = sum_r W[r,i] P[r,i],  sum_k A[i,k] = t_i = w_i^T s,
//   loss = kappa sum_i t_i^2/q_i + scale c / (2(c-1)),  kappa = scale (c-2) / (2(c-1))
//   dW = s alpha^T + (W alpha) 1^T + 2 P diag(beta) + 2 (W diag(beta) W^T) W,
//   alpha_i = 2 kappa t_i / q_i, beta_i = -kappa t_i^2 / q_i^2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_BLOCK) void ortho_lowrank_cols_kernel(const float* __restrict__ W,
                                                                       const float* __restrict__ P,
                                                                       const float* __restrict__ s, float scale,
                                                                       float* __restrict__ alpha,
                                                                       float* __restrict__ Wb, float* loss_accum,
                                                                       int rows, int c) {
    __shared__ float sh[4];
    const int i = blockIdx.x * AL_BLOCK + threadIdx.x;
    float contrib = 0.f;
    if (i < c) {
        float q = 0.f, t = 0.f;
        int r = 0;
        for (; r + 8 <= rows; r += 8) {           // 16 loads in flight (a plain loop is one memory round trip per row: 20 us per launch)
            float wv[8], pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                wv[u] = W[(int64_t)(r + u) * c + i];
                pv[u] = P[(int64_t)(r + u) * c + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                q += wv[u] * pv[u];
                t += wv[u] * s[r + u];
            }
        }
        for (; r < rows; ++r) {
            const float w = W[(int64_t)r * c + i];
            q += w * P[(int64_t)r * c + i];
            t += w * s[r];
        }
        const float kappa = c > 1 ? scale * (float)(c - 2) / (2.f * (float)(c - 1)) : 0.f;
        const bool clamped = q < 1e-12f;
        const float qc = fmaxf(q, 1e-12f);
        const float a = 2.f * kappa * t / qc;
        const float b = clamped ? 0.f : -kappa * t * t / (qc * qc);
        alpha[i] = a;
        contrib = kappa * t * t / qc + (c > 1 ? scale / (2.f * (float)(c - 1)) * (clamped ? 0.f : 1.f) : 0.f);
        r = 0;
        for (; r + 8 <= rows; r += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = W[(int64_t)(r + u) * c + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) Wb[(int64_t)(r + u) * c + i] = wv[u] * b;
        }
        for (; r < rows; ++r) Wb[(int64_t)r * c + i] = W[(int64_t)r * c + i] * b;
        // stash beta in the first row's slot of alpha's companion: beta = Wb / W is not safe (W may be 0),
        // so it is kept in alpha[c + i]
        alpha[c + i] = b;
    }
    contrib = block_sum_256(contrib, sh);
    if (threadIdx.x == 0) atomicAdd(loss_accum, contrib);
}

// dW = 2*T + s alpha^T + (W alpha) 1^T + 2 P diag(beta), T = (W diag(beta) W^T) W given in dW
__global__ __launch_bounds__(AL_BLOCK) void ortho_lowrank_finish_kernel(float* __restrict__ dW,
                                                                         const float* __restrict__ P,
                                                                         const float* __restrict__ s,
                                                                         const float* __restrict__ Walpha,
                                                                         const float* __restrict__ ab, int rows,
                                                                         int c) {
    const int64_t n = (int64_t)rows * c;
    for (int64_t idx = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * AL_BLOCK) {
        const int r = (int)(idx / c), i = (int)(idx % c);
        dW[idx] = 2.f * dW[idx] + s[r] * ab[i] + Walpha[r] + 2.f * P[idx] * ab[c + i];
    }
}

// ------------------------------------------------------------------------------------------
// General (non-penalty) GAN losses (ops.py:753-840): lsgan, gan, ra-lsgan, ra-gan, ra-hinge (+ hinge).
//   u_i = real_i - m_f, v_j = fake_j - m_r with m = batch means for the relativistic ("ra-") kinds, else u = real,
//   v = fake;  L = mean_i phi_r(u_i) + mean_j phi_f(v_j), flooded.  phi per (kind, D / G):
//     hinge    D: relu(1-u), relu(1+v)            G: -, -v
//     lsgan    D: (u-1)^2,  v^2                   G: -, (v-1)^2
//     gan      D: softplus(-u), softplus(v)       G: -, softplus(-v)
//     ra-lsgan D: (u-1)^2, (v+1)^2                G: (u+1)^2, (v-1)^2
//     ra-gan   D: softplus(-u), softplus(v)       G: softplus(u), softplus(-v)
//     ra-hinge D: relu(1-u), relu(1+v)            G: relu(1+u), relu(1-v)
//   Gradients include the coupling through the means: dL/dreal_i = (phi_r'(u_i) - mean_j phi_f'(v_j)) / N_r.
// Single-block kernels (logits are [B, 1]); sums are exchanged across data-parallel ranks by the caller.
// ------------------------------------------------------------------------------------------
enum { GL_HINGE = 0, GL_LSGAN = 1, GL_GAN = 2, GL_RA_LSGAN = 3, GL_RA_GAN = 4, GL_RA_HINGE = 5, GL_WGAN = 6 };
__device__ __forceinline__ bool gl_relativistic(int kind) { return kind >= GL_RA_LSGAN && kind <= GL_RA_HINGE; }

__device__ __forceinline__ float gl_softplus(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float gl_sigmoid(float x) {
    const float e = __expf(-fabsf(x));
    return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

// value and derivative of the real-side / fake-side term
__device__ __forceinline__ void gl_phi(int kind, int gen, bool real_side, float x, float* val, float* der) {
    const bool ra = gl_relativistic(kind);
    if (real_side && gen && !ra) {          // non-relativistic generator losses do not look at the real logits
        *val = 0.f;
        *der = 0.f;
        return;
    }
    // sign convention: s = +1 -> "push x up" form, s = -1 -> "push x down" form
    //   D real, G fake : wants x large  -> relu(1 - x), (x - 1)^2, softplus(-x)
    //   D fake, G real : wants x small  -> relu(1 + x), (x + 1)^2 (ra) / x^2 (lsgan), softplus(x)
    const bool up = (real_side != (gen != 0));
    switch (kind) {
        case GL_WGAN:                        // ops.py:757-759, 804-805: -mean(real) + mean(fake) ; G: -mean(fake)
            *val = up ? -x : x;
            *der = up ? -1.f : 1.f;
            break;
        case GL_HINGE:
        case GL_RA_HINGE:
            if (kind == GL_HINGE && gen) {   // ops.py:832-833: -mean(fake)
                *val = -x;
                *der = -1.f;
            } else if (up) {
                *val = fmaxf(1.f - x, 0.f);
                *der = (1.f - x) > 0.f ? -1.f : 0.f;
            } else {
                *val = fmaxf(1.f + x, 0.f);
                *der = (1.f + x) > 0.f ? 1.f : 0.f;
            }
            break;
        case GL_LSGAN:
        case GL_RA_LSGAN: {
            const float t = up ? 1.f : (kind == GL_RA_LSGAN ? -1.f : 0.f);
            *val = (x - t) * (x - t);
            *der = 2.f * (x - t);
            break;
        }
        default:                            // GL_GAN, GL_RA_GAN: sigmoid cross-entropy with label 1 (up) / 0 (down)
            if (up) {
                *val = gl_softplus(-x);
                *der = -gl_sigmoid(-x);
            } else {
                *val = gl_softplus(x);
                *der = gl_sigmoid(x);
            }
    }
}

// sums[0] = sum real, sums[1] = sum fake
__global__ __launch_bounds__(AL_BLOCK) void gan_means_kernel(const float* real, const float* fake, float* sums, int nr,
                                                              int nf) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nr; i += AL_BLOCK) a += real[i];
    for (int i = threadIdx.x; i < nf; i += AL_BLOCK) b += fake[i];
    a = block_sum_256(a, sh);
    b = block_sum_256(b, sh);
    if (threadIdx.x == 0) {
        sums[0] = a;
        sums[1] = b;
    }
}

// tsums = {sum phi_r, sum phi_f, sum phi_r', sum phi_f'} given the GLOBAL means
__global__ __launch_bounds__(AL_BLOCK) void gan_terms_kernel(int kind, int gen, const float* real, const float* fake,
                                                              const float* sums, double nr_g, double nf_g, float* tsums,
                                                              int nr, int nf) {
    __shared__ float sh[4];
    const bool ra = gl_relativistic(kind);
    const float mr = ra && nr_g > 0 ? (float)(sums[0] / nr_g) : 0.f;
    const float mf = ra && nf_g > 0 ? (float)(sums[1] / nf_g) : 0.f;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nr; i += AL_BLOCK) {
        float v, d;
        gl_phi(kind, gen, true, real[i] - mf, &v, &d);
        t[0] += v;
        t[2] += d;
    }
    for (int i = threadIdx.x; i < nf; i += AL_BLOCK) {
        float v, d;
        gl_phi(kind, gen, false, fake[i] - mr, &v, &d);
        t[1] += v;
        t[3] += d;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float s = block_sum_256(t[q], sh);
        if (threadIdx.x == 0) tsums[q] = s;
    }
}

__global__ __launch_bounds__(AL_BLOCK) void gan_grad_kernel(int kind, int gen, const float* real, const float* fake,
                                                             const float* sums, const float* tsums, double nr_g,
                                                             double nf_g, float flood, float* d_real, float* d_fake,
                                                             float* loss_out, int nr, int nf) {
    const bool ra = gl_relativistic(kind);
    const bool use_real = !(gen && !ra) && nr_g > 0;
    const float inv_r = use_real ? (float)(1.0 / nr_g) : 0.f, inv_f = (float)(1.0 / nf_g);
    const float mr = ra && nr_g > 0 ? (float)(sums[0] / nr_g) : 0.f;
    const float mf = ra ? (float)(sums[1] / nf_g) : 0.f;
    float L = tsums[0] * inv_r + tsums[1] * inv_f;
    float sgn = 1.f;
    if (flood != 0.f) {                                         // ops.py:794-795, 837-838
        const float d = L - flood;
        sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        L = fabsf(d) + flood;
    }
    if (threadIdx.x == 0 && loss_out) *loss_out = L;
    const float cross_r = ra ? tsums[3] * inv_f : 0.f;          // mean_j phi_f'(v_j): reaches real through m_r
    const float cross_f = ra ? tsums[2] * inv_r : 0.f;
    if (d_real)
        for (int i = threadIdx.x; i < nr; i += AL_BLOCK) {
            float v, d;
            gl_phi(kind, gen, true, real[i] - mf, &v, &d);
            d_real[i] = sgn * (d - cross_r) * inv_r;
        }
    for (int i = threadIdx.x; i < nf; i += AL_BLOCK) {
        float v, d;
        gl_phi(kind, gen, false, fake[i] - mr, &v, &d);
        d_fake[i] = sgn * (d - cross_f) * inv_f;
    }
}

// ------------------------------------------------------------------------------------------
// gradient penalty (BigGAN.py:717-742)
// ------------------------------------------------------------------------------------------
// interpolated = real + alpha_n * (other - real)                       (wgan-gp / wgan-lp: other = fake)
//              = real + alpha_n * 0.5 * std(real) * other              (dragan: other = eps ~ U[0,1), sums != NULL)
__global__ __launch_bounds__(AL_BLOCK) void gp_interpolate_kernel(const float* __restrict__ real,
                                                                   const float* __restrict__ other,
                                                                   const float* __restrict__ alpha,
                                                                   const double* __restrict__ sums, double count,
                                                                   float* __restrict__ out, int64_t per, int64_t total) {
    float half_std = 0.f;
    if (sums) {
        const double m = sums[0] / count;
        double var = sums[1] / count - m * m;
        if (var < 0) var = 0;
        half_std = 0.5f * sqrtf((float)var);
    }
    for (int64_t i = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * AL_BLOCK) {
        const float a = alpha[i / per], r = real[i];
        out[i] = sums ? r + a * (half_std * other[i]) : r + a * (other[i] - r);
    }
}

__global__ __launch_bounds__(AL_BLOCK) void gp_sumsq_kernel(const float* __restrict__ g, double* __restrict__ nsq,
                                                             int64_t per) {
    __shared__ float sh[4];
    const float* gs = g + (int64_t)blockIdx.y * per;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; i < per; i += (int64_t)gridDim.x * AL_BLOCK)
        acc += gs[i] * gs[i];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(nsq + blockIdx.y, (double)acc);
}

// per sample: n = ||g_n|| ; penalty (n-1)^2 (gp) or max(0,n-1)^2 (lp) ; coeff_n = ld * d(pen)/dn / count / n
__global__ __launch_bounds__(AL_BLOCK) void gp_finish_kernel(double* ws, int N, double count, float ld, int lp,
                                                              float* loss) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int n = threadIdx.x; n < N; n += AL_BLOCK) {
        const float nrm = sqrtf((float)ws[n]);
        float e = nrm - 1.f;
        if (lp) e = fmaxf(e, 0.f);
        acc += e * e;
        ws[N + n] = nrm > 0.f ? (double)(ld * 2.f * e / nrm) / count : 0.0;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) *loss = (float)((double)(ld * acc) / count);
}

__global__ __launch_bounds__(AL_BLOCK) void gp_scale_kernel(const float* __restrict__ g, const double* __restrict__ coeff,
                                                             float* __restrict__ v, int64_t per, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * AL_BLOCK)
        v[i] = (float)coeff[i / per] * g[i];
}

// S = A + A^T for a c x c matrix (32 x 32 tiles through LDS): the regulariser's backward W dA + W dA^T becomes
// one GEMM W S instead of two
__global__ __launch_bounds__(AL_BLOCK) void symmetrize_kernel(const float* __restrict__ A, float* __restrict__ S, int c) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int row = bx + r, col = by + tx;                       // transposed block
        tile[r][tx] = (row < c && col < c) ? A[(int64_t)row * c + col] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int row = by + r, col = bx + tx;
        if (row < c && col < c) S[(int64_t)row * c + col] = A[(int64_t)row * c + col] + tile[tx][r];
    }
}

// plain 'ortho' regulariser (utils.py:199-200): reg = A - I, loss = scale * l2_loss(reg), dA = scale * reg
__global__ __launch_bounds__(AL_BLOCK) void ortho_identity_kernel(const float* __restrict__ A, float scale,
                                                                   float* loss_accum, float* __restrict__ dA, int c) {
    __shared__ float sh[4];
    const int64_t n = (int64_t)c * c;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * AL_BLOCK) {
        const float r = A[i] - ((i / c) == (i % c) ? 1.f : 0.f);
        acc += r * r;
        if (dA) dA[i] = scale * r;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss_accum, 0.5f * scale * acc);
}

}  // namespace bg

using namespace bg;

static int da_check(const void* a, const void* b, int N, int S, int C, int policy) {
    BG_REQUIRE(a && b && N > 0 && S > 0 && C > 0 && C <= DA_MAXC, "bg_diffaugment: bad argument (C must be <= %d)", DA_MAXC);
    BG_REQUIRE((policy & ~7) == 0, "bg_diffaugment: unknown policy bits");
    return BG_OK;
}

extern "C" {

int bg_diffaugment_fwd(const float* x, float* y, const float* u_b, const float* u_s, const float* u_c,
                       const int32_t* t_x, const int32_t* t_y, const int32_t* o_x, const int32_t* o_y, int N, int S,
                       int C, int policy, double* mean_ws, void* stream) {
    int rc = da_check(x, y, N, S, C, policy);
    if (rc) return rc;
    BG_REQUIRE(!(policy & 1) || (u_b && u_s && u_c && mean_ws), "bg_diffaugment_fwd: color needs u_b,u_s,u_c,mean_ws");
    BG_REQUIRE(!(policy & 2) || (t_x && t_y), "bg_diffaugment_fwd: translation needs t_x,t_y");
    BG_REQUIRE(!(policy & 4) || (o_x && o_y), "bg_diffaugment_fwd: cutout needs o_x,o_y");
    hipStream_t s = as_stream(stream);
    DaArgs a{u_b, u_s, u_c, t_x, t_y, o_x, o_y, N, S, C, policy};
    const int64_t px = (int64_t)S * S;
    if (policy & 1) {
        if (hipMemsetAsync(mean_ws, 0, sizeof(double) * N, s) != hipSuccess) {
            set_error("bg_diffaugment_fwd: memset failed");
            return BG_ERR_LAUNCH;
        }
        int bx = (int)((px + AL_BLOCK * 8 - 1) / (AL_BLOCK * 8));
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(da_mean_kernel, dim3(bx, N), dim3(AL_BLOCK), 0, s, x, a, mean_ws);
        BG_LAUNCH_CHECK();
    }
    int64_t blocks = ((int64_t)N * px + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(da_fwd_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, s, x, y, a, mean_ws);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_diffaugment_bwd(const float* dy, float* dx, const float* u_s, const float* u_c, const int32_t* t_x,
                       const int32_t* t_y, const int32_t* o_x, const int32_t* o_y, int N, int S, int C, int policy,
                       double* mean_ws, void* stream) {
    int rc = da_check(dy, dx, N, S, C, policy);
    if (rc) return rc;
    BG_REQUIRE(!(policy & 1) || (u_s && u_c && mean_ws), "bg_diffaugment_bwd: color needs u_s,u_c,mean_ws");
    BG_REQUIRE(!(policy & 2) || (t_x && t_y), "bg_diffaugment_bwd: translation needs t_x,t_y");
    BG_REQUIRE(!(policy & 4) || (o_x && o_y), "bg_diffaugment_bwd: cutout needs o_x,o_y");
    hipStream_t s = as_stream(stream);
    DaArgs a{nullptr, u_s, u_c, t_x, t_y, o_x, o_y, N, S, C, policy};
    const int64_t px = (int64_t)S * S;
    if (policy & 1) {
        if (hipMemsetAsync(mean_ws, 0, sizeof(double) * N, s) != hipSuccess) {
            set_error("bg_diffaugment_bwd: memset failed");
            return BG_ERR_LAUNCH;
        }
        int bx = (int)((px + AL_BLOCK * 8 - 1) / (AL_BLOCK * 8));
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(da_bwd_sum_kernel, dim3(bx, N), dim3(AL_BLOCK), 0, s, dy, a, mean_ws);
        BG_LAUNCH_CHECK();
    }
    int64_t blocks = ((int64_t)N * px + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(da_bwd_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, s, dy, dx, a, mean_ws);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_hinge_d_sums(const float* real, const float* fake, float* sums, int n, void* stream) {
    BG_REQUIRE(real && fake && sums && n > 0, "bg_hinge_d_sums: bad argument");
    hipLaunchKernelGGL(hinge_d_sums_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), real, fake, sums, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_hinge_d_grad(const float* real, const float* fake, const float* sums, double n_global, float flood,
                    float* d_real, float* d_fake, float* loss_out, int n, void* stream) {
    BG_REQUIRE(real && fake && sums && d_real && d_fake && n > 0 && n_global > 0, "bg_hinge_d_grad: bad argument");
    hipLaunchKernelGGL(hinge_d_grad_kernel, dim3((n + AL_BLOCK - 1) / AL_BLOCK), dim3(AL_BLOCK), 0, as_stream(stream),
                       real, fake, sums, n_global, flood, d_real, d_fake, loss_out, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_hinge_g_sums(const float* fake, float* sums, int n, void* stream) {
    BG_REQUIRE(fake && sums && n > 0, "bg_hinge_g_sums: bad argument");
    hipLaunchKernelGGL(hinge_g_sums_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), fake, sums, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_hinge_g_grad(const float* sums, double n_global, float flood, float* d_fake, float* loss_out, int n,
                    void* stream) {
    BG_REQUIRE(sums && d_fake && n > 0 && n_global > 0, "bg_hinge_g_grad: bad argument");
    hipLaunchKernelGGL(hinge_g_grad_kernel, dim3((n + AL_BLOCK - 1) / AL_BLOCK), dim3(AL_BLOCK), 0, as_stream(stream),
                       sums, n_global, flood, d_fake, loss_out, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_sigmoid_ce(const float* logits, const float* truth, const float* weights, float scale, float* loss_out,
                  float* dlogits, int B, int n, void* stream) {
    BG_REQUIRE(logits && truth && loss_out && dlogits && B > 0 && n > 0, "bg_sigmoid_ce: bad argument");
    hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), logits, truth, weights, scale,
                       loss_out, dlogits, B, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_ortho_lowrank_cols(const float* W, const float* P, const float* s, float scale, float* alpha_beta, float* Wb,
                          float* loss_accum, int rows, int c, void* stream) {
    BG_REQUIRE(W && P && s && alpha_beta && Wb && loss_accum && rows > 0 && c > 0, "bg_ortho_lowrank_cols: bad argument");
    hipLaunchKernelGGL(ortho_lowrank_cols_kernel, dim3((c + AL_BLOCK - 1) / AL_BLOCK), dim3(AL_BLOCK), 0,
                       as_stream(stream), W, P, s, scale, alpha_beta, Wb, loss_accum, rows, c);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_ortho_lowrank_finish(float* dW, const float* P, const float* s, const float* Walpha, const float* alpha_beta,
                            int rows, int c, void* stream) {
    BG_REQUIRE(dW && P && s && Walpha && alpha_beta && rows > 0 && c > 0, "bg_ortho_lowrank_finish: bad argument");
    int64_t blocks = ((int64_t)rows * c + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ortho_lowrank_finish_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, as_stream(stream), dW,
                       P, s, Walpha, alpha_beta, rows, c);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_ortho_cosine_fwd_bwd(const float* A, float scale, float* loss_accum, float* dA, int c, void* stream) {
    BG_REQUIRE(A && loss_accum && c > 0, "bg_ortho_cosine_fwd_bwd: bad argument");
    hipLaunchKernelGGL(ortho_cosine_kernel, dim3(c), dim3(AL_BLOCK), 0, as_stream(stream), A, scale, loss_accum, dA, c);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gan_loss_means(const float* real, const float* fake, float* sums, int nr, int nf, void* stream) {
    BG_REQUIRE(fake && sums && nf > 0 && nr >= 0 && (nr == 0 || real), "bg_gan_loss_means: bad argument");
    hipLaunchKernelGGL(gan_means_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), real, fake, sums, nr, nf);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gan_loss_terms(int kind, int generator, const float* real, const float* fake, const float* sums,
                      double n_real_global, double n_fake_global, float* tsums, int nr, int nf, void* stream) {
    BG_REQUIRE(kind >= 0 && kind <= 6 && fake && sums && tsums && nf > 0 && nr >= 0 && (nr == 0 || real) &&
                   n_fake_global > 0, "bg_gan_loss_terms: bad argument");
    hipLaunchKernelGGL(gan_terms_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), kind, generator, real, fake,
                       sums, n_real_global, n_fake_global, tsums, nr, nf);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gan_loss_grad(int kind, int generator, const float* real, const float* fake, const float* sums,
                     const float* tsums, double n_real_global, double n_fake_global, float flood, float* d_real,
                     float* d_fake, float* loss_out, int nr, int nf, void* stream) {
    BG_REQUIRE(kind >= 0 && kind <= 6 && fake && sums && tsums && d_fake && nf > 0 && nr >= 0 &&
                   (nr == 0 || (real && d_real)) && n_fake_global > 0, "bg_gan_loss_grad: bad argument");
    hipLaunchKernelGGL(gan_grad_kernel, dim3(1), dim3(AL_BLOCK), 0, as_stream(stream), kind, generator, real, fake,
                       sums, tsums, n_real_global, n_fake_global, flood, nr ? d_real : nullptr, d_fake, loss_out, nr,
                       nf);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gp_interpolate(const float* real, const float* other, const float* alpha, const double* sums, double count,
                      float* out, int N, int64_t per, void* stream) {
    BG_REQUIRE(real && other && alpha && out && N > 0 && per > 0 && (!sums || count > 0), "bg_gp_interpolate: bad argument");
    const int64_t total = (int64_t)N * per;
    int64_t blocks = (total + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gp_interpolate_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, as_stream(stream), real, other,
                       alpha, sums, count, out, per, total);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gp_penalty(const float* g, int N, int64_t per, double count_global, float ld, int lp, double* ws, float* loss,
                  float* v, void* stream) {
    BG_REQUIRE(g && ws && loss && v && N > 0 && per > 0 && count_global > 0, "bg_gp_penalty: bad argument");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * (size_t)N, st) != hipSuccess) {
        set_error("bg_gp_penalty: memset failed");
        return BG_ERR_LAUNCH;
    }
    int64_t gx = (per + AL_BLOCK * 8 - 1) / (AL_BLOCK * 8);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gp_sumsq_kernel, dim3((unsigned)gx, (unsigned)N), dim3(AL_BLOCK), 0, st, g, ws, per);
    BG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(AL_BLOCK), 0, st, ws, N, count_global, ld, lp, loss);
    BG_LAUNCH_CHECK();
    const int64_t total = (int64_t)N * per;
    int64_t blocks = (total + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gp_scale_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, st, g, ws + N, v, per, total);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_symmetrize(const float* A, float* S, int c, void* stream) {
    BG_REQUIRE(A && S && A != S && c > 0, "bg_symmetrize: bad argument");
    const unsigned nb = (unsigned)((c + 31) / 32);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(AL_BLOCK), 0, as_stream(stream), A, S, c);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_ortho_identity_fwd_bwd(const float* A, float scale, float* loss_accum, float* dA, int c, void* stream) {
    BG_REQUIRE(A && loss_accum && c > 0, "bg_ortho_identity_fwd_bwd: bad argument");
    int64_t blocks = ((int64_t)c * c + AL_BLOCK - 1) / AL_BLOCK;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(ortho_identity_kernel, dim3((unsigned)blocks), dim3(AL_BLOCK), 0, as_stream(stream), A, scale,
                       loss_accum, dA, c);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
