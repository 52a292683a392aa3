// attention.hip - fused (flash-style) attention of self_attention_2 (ops.py:481-485) for gfx950:
//     o = softmax(q k^T) v       q [B,N,d]  k [B,Nk,d]  v [B,Nk,dv]  o [B,N,dv]   (no 1/sqrt(d) scale)
// The [N,Nk] logits / probabilities never reach HBM: forward keeps a running max / sum per query and
// saves lse = max + log(sum); backward recomputes the probabilities from q, k and lse.
//
// All contractions are v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  One wave owns 32 queries (or,
// in the dK/dV kernel, 32 keys).  Scores are produced TRANSPOSED (S^T = K Q^T: rows = keys, columns =
// queries), so in the 32x32 accumulator layout (column = lane & 31, row = (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5)) a lane holds 16 keys of ONE query: the softmax reductions are 15 in-lane ops + one
// cross-half shuffle, and accumulator register j is directly the B operand (k = lane >> 5,
// n = lane & 31) of the next contraction over keys, whose A operand is read from LDS in the same
// permuted key order  key(j, half) = (j & 3) + 8 (j >> 2) + 4 half.  Probabilities never touch LDS.
//
// Deterministic: no atomics (dQ and dK/dV are separate kernels, each recomputing S).
#include "common.h"

namespace bg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Cooperative [32 x cols] fp32 tile loader (rows contiguous in global memory, cols % 4 == 0): 16-byte
// global loads into registers, scalar LDS stores into rows of STRIDE floats (odd strides allowed).
template <int MAXCOLS, int STRIDE>
struct Tile32 {
    static constexpr int NPF = (32 * MAXCOLS / 4 + 255) / 256;
    float4 reg[NPF];
    int off[NPF];      // LDS offset of the float4, -1 = nothing to do
    int goff[NPF];     // global offset inside the tile
    __device__ __forceinline__ void init(int cols) {
        const int n4 = 32 * cols / 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (idx < n4) {
                const int e = idx * 4;
                const int row = e / cols;
                off[i] = row * STRIDE + (e - row * cols);
                goff[i] = e;
            } else {
                off[i] = -1;
                goff[i] = 0;
            }
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ src) {
#pragma unroll
        for (int i = 0; i < NPF; ++i)
            if (off[i] >= 0) reg[i] = *reinterpret_cast<const float4*>(src + goff[i]);
    }
    __device__ __forceinline__ void store(float* dst) const {
#pragma unroll
        for (int i = 0; i < NPF; ++i)
            if (off[i] >= 0) {
                float* p = dst + off[i];
                p[0] = reg[i].x;
                p[1] = reg[i].y;
                p[2] = reg[i].z;
                p[3] = reg[i].w;
            }
    }
};

__device__ __forceinline__ void lds_zero(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0.f;
}

// ------------------------------------------------------------------------------------------
// forward: grid (N/128, B), 4 waves x 32 queries, key tiles of 32
// ------------------------------------------------------------------------------------------
template <int DQ, int DVT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o,
                                                       float* __restrict__ lse, int N, int Nk, int d, int dv) {
    constexpr int DV = DVT * 32;
    constexpr int KS = DQ + 1;     // odd: rows = lanes reads are conflict free
    constexpr int VS = DV + 4;
    __shared__ float Ks[2][32 * KS];
    __shared__ float Vs[2][32 * VS];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;

    lds_zero(&Ks[0][0], 2 * 32 * KS);
    lds_zero(&Vs[0][0], 2 * 32 * VS);

    float qreg[DQ / 2];
    {
        const float* qb = q + ((int64_t)b * N + q0 + col) * d;
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) {
            const int c = 2 * s + half;
            qreg[s] = c < d ? qb[c] : 0.f;
        }
    }
    const float* kb = k + (int64_t)b * Nk * d;
    const float* vb = v + (int64_t)b * Nk * dv;
    Tile32<DQ, KS> tk;
    Tile32<DV, VS> tv;
    tk.init(d);
    tv.init(dv);
    tk.load(kb);
    tv.load(vb);
    __syncthreads();                 // zero fill done
    tk.store(&Ks[0][0]);
    tv.store(&Vs[0][0]);
    __syncthreads();

    f32x16 oacc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = Nk / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntiles) {
            tk.load(kb + (int64_t)(it + 1) * 32 * d);
            tv.load(vb + (int64_t)(it + 1) * 32 * dv);
        }
        const float* ks = &Ks[cur][0];
        const float* vs = &Vs[cur][0];
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) st = MFMA(ks[col * KS + 2 * s + half], qreg[s], st);
        // online softmax for query `col` (this lane holds 16 of the tile's 32 keys, the other half the rest)
        float mx = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_new);
            rs += st[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = acc_row(j, half);
#pragma unroll
            for (int t = 0; t < DVT; ++t) oacc[t] = MFMA(vs[key * VS + t * 32 + col], st[j], oacc[t]);
        }
        if (it + 1 < ntiles) {
            tk.store(&Ks[cur ^ 1][0]);
            tv.store(&Vs[cur ^ 1][0]);
        }
        __syncthreads();
    }
    const float inv = 1.f / l_run;
    float* ob = o + ((int64_t)b * N + q0 + col) * dv;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < dv)
                *reinterpret_cast<float4*>(ob + c0) = make_float4(oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv,
                                                                  oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
        }
    if (half == 0) lse[(int64_t)b * N + q0 + col] = m_run + __logf(l_run);
}

// delta[row] = sum_c dO[row,c] * O[row,c]   (one wave per row)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t rows, int dv) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < dv; c += 64) s += o[row * dv + c] * dout[row * dv + c];
    s = wave_sum(s);
    if (lane == 0) delta[row] = s;
}

// ------------------------------------------------------------------------------------------
// backward, dQ: grid (N/128, B); a wave owns 32 queries and walks the keys
//   S^T = K Q^T ; P^T = exp(S^T - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - delta) ; dQ^T += K^T dS^T
// ------------------------------------------------------------------------------------------
template <int DQ, int DVT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v,
                                                          const float* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta, float* __restrict__ dq,
                                                          int N, int Nk, int d, int dv) {
    constexpr int DV = DVT * 32;
    constexpr int KS = DQ + 1;
    constexpr int VS = DV + 1;
    constexpr int MT = (DQ + 31) / 32;
    __shared__ float Ks[2][32 * KS];
    __shared__ float Vs[2][32 * VS];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;

    lds_zero(&Ks[0][0], 2 * 32 * KS);
    lds_zero(&Vs[0][0], 2 * 32 * VS);

    float qreg[DQ / 2], doreg[DV / 2];
    {
        const float* qb = q + ((int64_t)b * N + q0 + col) * d;
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) {
            const int c = 2 * s + half;
            qreg[s] = c < d ? qb[c] : 0.f;
        }
        const float* gb = dout + ((int64_t)b * N + q0 + col) * dv;
#pragma unroll
        for (int s = 0; s < DV / 2; ++s) {
            const int c = 2 * s + half;
            doreg[s] = c < dv ? gb[c] : 0.f;
        }
    }
    const float lse_q = lse[(int64_t)b * N + q0 + col];
    const float delta_q = delta[(int64_t)b * N + q0 + col];
    const float* kb = k + (int64_t)b * Nk * d;
    const float* vb = v + (int64_t)b * Nk * dv;
    Tile32<DQ, KS> tk;
    Tile32<DV, VS> tv;
    tk.init(d);
    tv.init(dv);
    tk.load(kb);
    tv.load(vb);
    __syncthreads();
    tk.store(&Ks[0][0]);
    tv.store(&Vs[0][0]);
    __syncthreads();

    f32x16 dqacc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[t][r] = 0.f;

    const int ntiles = Nk / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntiles) {
            tk.load(kb + (int64_t)(it + 1) * 32 * d);
            tv.load(vb + (int64_t)(it + 1) * 32 * dv);
        }
        const float* ks = &Ks[cur][0];
        const float* vs = &Vs[cur][0];
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = 0.f;
            dp[r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) st = MFMA(ks[col * KS + 2 * s + half], qreg[s], st);
#pragma unroll
        for (int s = 0; s < DV / 2; ++s) dp = MFMA(vs[col * VS + 2 * s + half], doreg[s], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - lse_q) * (dp[r] - delta_q);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = acc_row(j, half);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int c = t * 32 + col;
                const float a = ks[key * KS + (c < DQ ? c : 0)];
                dqacc[t] = MFMA(c < DQ ? a : 0.f, st[j], dqacc[t]);
            }
        }
        if (it + 1 < ntiles) {
            tk.store(&Ks[cur ^ 1][0]);
            tv.store(&Vs[cur ^ 1][0]);
        }
        __syncthreads();
    }
    float* qo = dq + ((int64_t)b * N + q0 + col) * d;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < d)
                *reinterpret_cast<float4*>(qo + c0) =
                    make_float4(dqacc[t][4 * g], dqacc[t][4 * g + 1], dqacc[t][4 * g + 2], dqacc[t][4 * g + 3]);
        }
}

// ------------------------------------------------------------------------------------------
// backward, dK and dV: grid (Nk/128, B); a wave owns 32 keys and walks the queries
//   S = Q K^T ; P = exp(S - lse_row) ; dV^T += dO^T P ; dP = dO V^T ; dS = P (dP - delta_row) ; dK^T += Q^T dS
// ------------------------------------------------------------------------------------------
template <int DQ, int DVT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta, float* __restrict__ dk,
                                                           float* __restrict__ dvo, int N, int Nk, int d, int dv) {
    constexpr int DV = DVT * 32;
    constexpr int QS = DQ + 1;
    constexpr int OS = DV + 1;
    constexpr int MT = (DQ + 31) / 32;
    __shared__ float Qs[2][32 * QS];
    __shared__ float Os[2][32 * OS];
    __shared__ float Ls[2][32];
    __shared__ float Ds[2][32];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int k0 = blockIdx.x * 128 + wave * 32;

    lds_zero(&Qs[0][0], 2 * 32 * QS);
    lds_zero(&Os[0][0], 2 * 32 * OS);

    float kreg[DQ / 2], vreg[DV / 2];
    {
        const float* kp = k + ((int64_t)b * Nk + k0 + col) * d;
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) {
            const int c = 2 * s + half;
            kreg[s] = c < d ? kp[c] : 0.f;
        }
        const float* vp = v + ((int64_t)b * Nk + k0 + col) * dv;
#pragma unroll
        for (int s = 0; s < DV / 2; ++s) {
            const int c = 2 * s + half;
            vreg[s] = c < dv ? vp[c] : 0.f;
        }
    }
    const float* qb = q + (int64_t)b * N * d;
    const float* gb = dout + (int64_t)b * N * dv;
    const float* lb = lse + (int64_t)b * N;
    const float* db = delta + (int64_t)b * N;
    Tile32<DQ, QS> tq;
    Tile32<DV, OS> tg;
    tq.init(d);
    tg.init(dv);
    tq.load(qb);
    tg.load(gb);
    float l_pf = 0.f, d_pf = 0.f;
    if (threadIdx.x < 32) {
        l_pf = lb[threadIdx.x];
        d_pf = db[threadIdx.x];
    }
    __syncthreads();
    tq.store(&Qs[0][0]);
    tg.store(&Os[0][0]);
    if (threadIdx.x < 32) {
        Ls[0][threadIdx.x] = l_pf;
        Ds[0][threadIdx.x] = d_pf;
    }
    __syncthreads();

    f32x16 dvacc[DVT], dkacc[MT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dvacc[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[t][r] = 0.f;

    const int ntiles = N / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntiles) {
            tq.load(qb + (int64_t)(it + 1) * 32 * d);
            tg.load(gb + (int64_t)(it + 1) * 32 * dv);
            if (threadIdx.x < 32) {
                l_pf = lb[(it + 1) * 32 + threadIdx.x];
                d_pf = db[(it + 1) * 32 + threadIdx.x];
            }
        }
        const float* qs = &Qs[cur][0];
        const float* os = &Os[cur][0];
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = 0.f;
            dp[r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < DQ / 2; ++s) st = MFMA(qs[col * QS + 2 * s + half], kreg[s], st);   // rows = queries
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - Ls[cur][acc_row(r, half)]);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qr = acc_row(j, half);
#pragma unroll
            for (int t = 0; t < DVT; ++t) dvacc[t] = MFMA(os[qr * OS + t * 32 + col], st[j], dvacc[t]);
        }
#pragma unroll
        for (int s = 0; s < DV / 2; ++s) dp = MFMA(os[col * OS + 2 * s + half], vreg[s], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] *= dp[r] - Ds[cur][acc_row(r, half)];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qr = acc_row(j, half);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int c = t * 32 + col;
                const float a = qs[qr * QS + (c < DQ ? c : 0)];
                dkacc[t] = MFMA(c < DQ ? a : 0.f, st[j], dkacc[t]);
            }
        }
        if (it + 1 < ntiles) {
            tq.store(&Qs[cur ^ 1][0]);
            tg.store(&Os[cur ^ 1][0]);
            if (threadIdx.x < 32) {
                Ls[cur ^ 1][threadIdx.x] = l_pf;
                Ds[cur ^ 1][threadIdx.x] = d_pf;
            }
        }
        __syncthreads();
    }
    float* ko = dk + ((int64_t)b * Nk + k0 + col) * d;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < d)
                *reinterpret_cast<float4*>(ko + c0) =
                    make_float4(dkacc[t][4 * g], dkacc[t][4 * g + 1], dkacc[t][4 * g + 2], dkacc[t][4 * g + 3]);
        }
    float* vo = dvo + ((int64_t)b * Nk + k0 + col) * dv;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < dv)
                *reinterpret_cast<float4*>(vo + c0) =
                    make_float4(dvacc[t][4 * g], dvacc[t][4 * g + 1], dvacc[t][4 * g + 2], dvacc[t][4 * g + 3]);
        }
}

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
static bool attn_shape_ok(int N, int Nk, int d, int dv) {
    return N > 0 && Nk > 0 && N % 128 == 0 && Nk % 128 == 0 && d > 0 && dv > 0 && d % 4 == 0 && dv % 4 == 0 &&
           d <= 32 && dv <= 128;
}

template <int DQ, int DVT>
static void launch_fwd(hipStream_t s, const float* q, const float* k, const float* v, float* o, float* lse, int B,
                       int N, int Nk, int d, int dv) {
    prof_kernel("attn_fwd_kernel<%d, %d>", DQ, DVT);
    hipLaunchKernelGGL((attn_fwd_kernel<DQ, DVT>), dim3(N / 128, B), dim3(256), 0, s, q, k, v, o, lse, N, Nk, d, dv);
}
template <int DQ, int DVT>
static void launch_bwd(hipStream_t s, const float* q, const float* k, const float* v, const float* dout,
                       const float* lse, const float* delta, float* dq, float* dk, float* dvo, int B, int N, int Nk,
                       int d, int dv) {
    prof_kernel("attn_bwd_dq_kernel + attn_bwd_dkv_kernel<%d, %d>", DQ, DVT);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DQ, DVT>), dim3(N / 128, B), dim3(256), 0, s, q, k, v, dout, lse, delta,
                       dq, N, Nk, d, dv);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DQ, DVT>), dim3(Nk / 128, B), dim3(256), 0, s, q, k, v, dout, lse, delta,
                       dk, dvo, N, Nk, d, dv);
}

#define ATTN_DISPATCH(FN, ...)                                      \
    do {                                                            \
        const int dvt = (dv + 31) / 32;                             \
        if (d <= 16) {                                              \
            if (dvt == 1) FN<16, 1>(__VA_ARGS__);                   \
            else if (dvt == 2) FN<16, 2>(__VA_ARGS__);              \
            else if (dvt == 3) FN<16, 3>(__VA_ARGS__);              \
            else FN<16, 4>(__VA_ARGS__);                            \
        } else {                                                    \
            if (dvt == 1) FN<32, 1>(__VA_ARGS__);                   \
            else if (dvt == 2) FN<32, 2>(__VA_ARGS__);              \
            else if (dvt == 3) FN<32, 3>(__VA_ARGS__);              \
            else FN<32, 4>(__VA_ARGS__);                            \
        }                                                           \
    } while (0)

}  // namespace bg

using namespace bg;

extern "C" {

int bg_attention2_supported(int N, int Nk, int d, int dv) { return attn_shape_ok(N, Nk, d, dv) ? 1 : 0; }

int bg_attention2_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int N, int Nk,
                      int d, int dv, void* stream) {
    BG_REQUIRE(q && k && v && o && lse && B > 0, "bg_attention2_fwd: bad argument");
    BG_REQUIRE(attn_shape_ok(N, Nk, d, dv),
               "bg_attention2_fwd: unsupported shape N=%d Nk=%d d=%d dv=%d (need N,Nk %% 128 == 0, d,dv %% 4 == 0, "
               "d <= 32, dv <= 128)", N, Nk, d, dv);
    BG_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0,
               "bg_attention2_fwd: pointers must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    ProfScope prof(s, 2.0 * B * (double)N * Nk * (d + dv), "attention2_fwd");
    ATTN_DISPATCH(launch_fwd, s, q, k, v, o, lse, B, N, Nk, d, dv);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_attention2_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                      const float* lse, float* dq, float* dk, float* dv_out, float* delta_ws, int B, int N, int Nk,
                      int d, int dv, void* stream) {
    BG_REQUIRE(q && k && v && o && dout && lse && dq && dk && dv_out && delta_ws && B > 0,
               "bg_attention2_bwd: bad argument");
    BG_REQUIRE(attn_shape_ok(N, Nk, d, dv), "bg_attention2_bwd: unsupported shape N=%d Nk=%d d=%d dv=%d", N, Nk, d, dv);
    BG_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk |
                 (uintptr_t)dv_out) & 15) == 0,
               "bg_attention2_bwd: pointers must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const int64_t rows = (int64_t)B * N;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, o, dout, delta_ws, rows,
                       dv);
    BG_LAUNCH_CHECK();
    // algorithmic backward work of the materialised form: dV, dP, dQ, dK = 2 (d + dv) MACs per (query, key)
    ProfScope prof(s, 2.0 * B * (double)N * Nk * (2.0 * d + 2.0 * dv), "attention2_bwd");
    ATTN_DISPATCH(launch_bwd, s, q, k, v, dout, lse, delta_ws, dq, dk, dv_out, B, N, Nk, d, dv);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
