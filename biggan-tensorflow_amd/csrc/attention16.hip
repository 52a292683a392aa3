// attention16.hip - fused (flash-style) attention of self_attention_2 (ops.py:481-485) on the bf16 MFMA, for the
// bf16-resident data path (BASELINE configs 3-5):   o = softmax(q k^T) v   (no 1/sqrt(d) scale)
//     q [B,N,d]  k [B,Nk,d]  v [B,Nk,dv]  o [B,N,dv]  bf16, each with its own row stride (they are column slices of the
//     fused f|g|h projection and of its max-pooled copy), d <= 64, dv <= 256: every BASELINE topology (the 256^2 / 512^2
//     generators use d = 48 / 64, dv = 192 / 256, BigGAN.py:292-293).  lse [B,N] fp32 is kept for backward.
// The [N,Nk] logits / probabilities never reach HBM; backward recomputes them from q, k and lse (two deterministic
// kernels: no atomics).
//
// Same register choreography as the fp32 kernels of attention.hip, on v_mfma_f32_32x32x16_bf16: scores are produced
// TRANSPOSED (S^T = K Q^T: rows = keys, columns = queries), so a lane holds 16 keys of ONE query (softmax reductions are
// in-lane plus one cross-half shuffle) and the accumulator, converted pairwise to bf16, is directly the B operand of the
// next product over keys (MI355X guide: "an accumulator tile as the next MFMA's operand"): element j of lane half h in
// k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3).  The other operand of that product (V^T, K^T, dO^T, Q^T) is read from
// a row-major LDS tile with ds_read_b64_tr_b16 in exactly that key order (two transposed reads of 4 consecutive rows).
// LDS tiles: 32 rows x 128 bf16 images with 256-byte rows, 16-byte chunks XOR-swizzled by ((row & 3) << 2) | ((row >> 2) & 3):
// conflict-free for the row reads (ds_read_b128) AND the transposed reads of the 32x32x16 operands.
#include "common.h"

namespace bg {

typedef float a16_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 a16_bf16x8 __attribute__((ext_vector_type(8)));
typedef short a16_s16x4 __attribute__((ext_vector_type(4)));
typedef short a16_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) a16_s16x4 a16_lds_s16x4;

#define A16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define A16_IMG 8192            // bytes of one 32 x 128 bf16 image

__device__ __forceinline__ int a16_off(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ int a16_acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ uint32_t a16_pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

// accumulator registers 8 s .. 8 s + 7 -> the bf16 B-operand fragment of k-step s
__device__ __forceinline__ a16_bf16x8 a16_frag_of_acc(const a16_f32x16& x, int s) {
    a16_bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)x[8 * s + j];
    return f;
}

// Cooperative loader of a [32 rows x COLS] bf16 tile (row stride ld elements, 8-byte granularity so that 8-byte aligned
// column slices work; columns >= valid are zero) into swizzled images; 256 threads.
template <int COLS>
struct A16Tile {
    static constexpr int UPR = COLS / 4;                 // 8-byte units per row
    static constexpr int U = 32 * UPR;
    static constexpr int NPT = (U + 255) / 256;
    uint2 reg[NPT];
    __device__ __forceinline__ void load(const __bf16* __restrict__ src, int64_t ld, int valid) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int u = threadIdx.x + 256 * i;
            uint2 v = make_uint2(0u, 0u);
            if (u < U) {
                const int row = u / UPR, c4 = u - row * UPR;
                if (4 * c4 < valid) v = *reinterpret_cast<const uint2*>(src + (int64_t)row * ld + 4 * c4);
            }
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void store(unsigned char* img) const {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int u = threadIdx.x + 256 * i;
            if (u < U) {
                const int row = u / UPR, c4 = u - row * UPR;
                const int chunk = c4 >> 1;
                *reinterpret_cast<uint2*>(img + (chunk >> 4) * A16_IMG + a16_off(row, chunk & 15) + 8 * (c4 & 1)) = reg[i];
            }
        }
    }
};

// A-operand fragment by ROWS of an image: lane -> row (lane & 31), 8 consecutive columns of 16-byte chunk `chunk`
__device__ __forceinline__ a16_bf16x8 a16_row_frag(const unsigned char* img, int lane, int chunk) {
    return *reinterpret_cast<const a16_bf16x8*>(img + (chunk >> 4) * A16_IMG + a16_off(lane & 31, chunk & 15));
}

// per-lane LDS byte offsets (k-step 0) of the two transposed reads that deliver column (32 t + (lane & 31)) of the rows
// 4 h + q .. (first read) and 8 + 4 h + q .. (second read); k-step s adds 4096 s
__device__ __forceinline__ void a16_tr_offsets(int lane, int t, int (&off)[2]) {
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p = lane & 3;
    const int chunk = 4 * t + 2 * g16 + (p >> 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
        off[jj] = (chunk >> 4) * A16_IMG + a16_off(8 * jj + 4 * h + q4, chunk & 15) + 8 * (p & 1);
}
__device__ __forceinline__ a16_bf16x8 a16_tr_frag(uint32_t img_lds, const int (&off)[2], int s) {
    const a16_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<a16_lds_s16x4*>(img_lds + off[0] + 4096 * s));
    const a16_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<a16_lds_s16x4*>(img_lds + off[1] + 4096 * s));
    const a16_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(a16_bf16x8, v);
}
__device__ __forceinline__ uint32_t a16_lds_addr(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const unsigned char*)p));
}

// B-operand fragments held in registers for a whole kernel: row (lane & 31) of a [rows, cols] bf16 matrix,
// columns 16 s + 8 h .. + 7 (8-byte loads; zero beyond `valid`)
template <int KS>
__device__ __forceinline__ void a16_load_row_regs(const __bf16* __restrict__ rowp, int valid, int half,
                                                  a16_bf16x8 (&f)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 16 * s + 8 * half;
        uint2 lo = make_uint2(0u, 0u), hi = lo;
        if (c < valid) lo = *reinterpret_cast<const uint2*>(rowp + c);
        if (c + 4 < valid) hi = *reinterpret_cast<const uint2*>(rowp + c + 4);
        const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        f[s] = __builtin_bit_cast(a16_bf16x8, v);
    }
}

// store 4 consecutive channels (accumulator registers 4 g .. 4 g + 3) as bf16
__device__ __forceinline__ void a16_store4(__bf16* p, float a, float b, float c, float d) {
    uint2 v;
    v.x = a16_pack2(a, b);
    v.y = a16_pack2(c, d);
    *reinterpret_cast<uint2*>(p) = v;
}

// Barrier of the tile loops: this wave's LDS reads / writes are complete (lgkmcnt), then the block barrier - and NOTHING
// about global memory.  __syncthreads() carries a workgroup fence, i.e. s_waitcnt vmcnt(0): it drained the register
// prefetch of the next key / value tiles at every tile, so each iteration (8 - 20 MFMAs, ~0.2 us) waited out a full global
// load (~1.5 us): 1.7 us per tile in all three kernels of round 2, whatever their arithmetic.
__device__ __forceinline__ void a16_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

constexpr float A16_LOG2E = 1.4426950408889634f;

struct A16Geom {
    int N, Nk, d, dv;
    int64_t ldq, ldk, ldv, ldo;          // row strides (elements)
    int64_t sq, sk, sv, so;              // batch strides (elements)
};

// ------------------------------------------------------------------------------------------
// forward: grid (N / 128, B); a wave owns 32 queries, key tiles of 32
// ------------------------------------------------------------------------------------------
template <int DQT, int DVT>          // d <= 16 DQT, dv <= 32 DVT
__global__ __launch_bounds__(256) void attn16_fwd_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                         const __bf16* __restrict__ v, __bf16* __restrict__ o,
                                                         float* __restrict__ lse, const A16Geom gm) {
    constexpr int VIMG = (DVT + 3) / 4;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[2][A16_IMG];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[2][VIMG * A16_IMG];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;

    a16_bf16x8 qf[DQT];
    a16_load_row_regs<DQT>(q + b * gm.sq + (int64_t)(q0 + col) * gm.ldq, gm.d, half, qf);
    const __bf16* kb = k + b * gm.sk;
    const __bf16* vb = v + b * gm.sv;
    // Key / value tiles of 32 keys travel global -> registers -> LDS (double-buffered).  TWO tiles are kept in flight in
    // registers (8 + 24 bytes per thread each): a tile's load is issued two iterations before its LDS store.  With one in
    // flight (r02) an iteration - 8 MFMAs, ~0.2 us - could not cover the ~1.5 us of a global load, and the kernel ran
    // at the load latency: 860 us for the generator's shape at batch 256 against ~110 us of MFMA time.
    A16Tile<16 * DQT> tk0, tk1;
    A16Tile<32 * DVT> tv0, tv1;
    const int ntiles = gm.Nk / 32;
    tk0.load(kb, gm.ldk, gm.d);
    tv0.load(vb, gm.ldv, gm.dv);
    tk0.store(Ks[0]);
    tv0.store(Vs[0]);
    if (ntiles > 1) {
        tk1.load(kb + (int64_t)32 * gm.ldk, gm.ldk, gm.d);
        tv1.load(vb + (int64_t)32 * gm.ldv, gm.ldv, gm.dv);
    }
    if (ntiles > 2) {
        tk0.load(kb + (int64_t)64 * gm.ldk, gm.ldk, gm.d);
        tv0.load(vb + (int64_t)64 * gm.ldv, gm.ldv, gm.dv);
    }
    __syncthreads();

    int vtr[DVT][2];
#pragma unroll
    for (int t = 0; t < DVT; ++t) a16_tr_offsets(lane, t, vtr[t]);
    const uint32_t vs_lds = a16_lds_addr(&Vs[0][0]);

    a16_f32x16 oacc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // one key tile: scores, online softmax, P V on LDS buffer `cur`; then tile it + 1 (registers nk / nv, loaded two
    // iterations ago) goes into the other buffer and those registers take tile it + 3
    auto tile = [&](int it, auto& nk, auto& nv) {
        const int cur = it & 1;
        a16_f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < DQT; ++s) st = A16_MFMA(a16_row_frag(Ks[cur], lane, 2 * s + half), qf[s], st);
        // online softmax for query `col` (this lane holds 16 of the tile's 32 keys, the other half the rest), in base 2:
        // 2^(log2e (s - m)) - one multiply per score folded into the exponent's argument (v_exp_f32 is 2^x)
        float mx = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float mneg = -m_new * A16_LOG2E;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], A16_LOG2E, mneg));
            rs += st[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        // the running maximum settles after a few tiles: the 16 DVT rescaling multiplies are skipped (wave-uniform branch)
        // while no query of this wave raised its maximum
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * A16_LOG2E);
            l_run *= alpha;
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
        l_run += rs;
        m_run = m_new;
        const uint32_t vcur = vs_lds + cur * (VIMG * A16_IMG);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const a16_bf16x8 pb = a16_frag_of_acc(st, s);
#pragma unroll
            for (int t = 0; t < DVT; ++t) oacc[t] = A16_MFMA(a16_tr_frag(vcur, vtr[t], s), pb, oacc[t]);
        }
        if (it + 1 < ntiles) {
            nk.store(Ks[cur ^ 1]);
            nv.store(Vs[cur ^ 1]);
        }
        if (it + 3 < ntiles) {
            nk.load(kb + (int64_t)(it + 3) * 32 * gm.ldk, gm.ldk, gm.d);
            nv.load(vb + (int64_t)(it + 3) * 32 * gm.ldv, gm.ldv, gm.dv);
        }
        a16_lds_barrier();
    };
    for (int it = 0; it < ntiles; it += 2) {
        tile(it, tk1, tv1);                       // (tile it + 1 sits in the odd register set)
        if (it + 1 < ntiles) tile(it + 1, tk0, tv0);
    }
    const float inv = 1.f / l_run;
    __bf16* ob = o + b * gm.so + (int64_t)(q0 + col) * gm.ldo;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < gm.dv)
                a16_store4(ob + c0, oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv, oacc[t][4 * g + 2] * inv,
                           oacc[t][4 * g + 3] * inv);
        }
    if (half == 0) lse[(int64_t)b * gm.N + q0 + col] = m_run + __logf(l_run);
}

// delta[row] = sum_c dO[row,c] * O[row,c]   (one wave per row)
__global__ __launch_bounds__(256) void attn16_delta_kernel(const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                           float* __restrict__ delta, int B, int N, int dv, int64_t ldo,
                                                           int64_t so, int64_t ldg, int64_t sg) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * N) return;
    const int b = (int)(row / N), n = (int)(row - (int64_t)b * N);
    const __bf16* op = o + b * so + n * ldo;
    const __bf16* gp = dout + b * sg + n * ldg;
    float s = 0.f;
    for (int c = lane; c < dv; c += 64) s += (float)op[c] * (float)gp[c];
    s = wave_sum(s);
    if (lane == 0) delta[row] = s;
}

// ------------------------------------------------------------------------------------------
// backward, dQ: grid (N / 128, B); a wave owns 32 queries and walks the keys
//   S^T = K Q^T ; P^T = exp(S^T - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - delta) ; dQ^T += K^T dS^T
// ------------------------------------------------------------------------------------------
template <int DQT, int DVT>
__global__ __launch_bounds__(256) void attn16_bwd_dq_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                            const __bf16* __restrict__ v,
                                                            const __bf16* __restrict__ dout,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta, __bf16* __restrict__ dq,
                                                            const A16Geom gm, int64_t ldg, int64_t sg, int64_t lddq,
                                                            int64_t sdq) {
    constexpr int VIMG = (DVT + 3) / 4;
    constexpr int MT = (DQT + 1) / 2;           // 32-row tiles of dQ^T
    __shared__ __attribute__((aligned(16))) unsigned char Ks[2][A16_IMG];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[2][VIMG * A16_IMG];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;

    a16_bf16x8 qf[DQT], gf[2 * DVT];
    a16_load_row_regs<DQT>(q + b * gm.sq + (int64_t)(q0 + col) * gm.ldq, gm.d, half, qf);
    a16_load_row_regs<2 * DVT>(dout + b * sg + (int64_t)(q0 + col) * ldg, gm.dv, half, gf);
    const float lse_q = lse[(int64_t)b * gm.N + q0 + col];
    const float delta_q = delta[(int64_t)b * gm.N + q0 + col];
    const __bf16* kb = k + b * gm.sk;
    const __bf16* vb = v + b * gm.sv;
    A16Tile<16 * DQT> tk;
    A16Tile<32 * DVT> tv;
    tk.load(kb, gm.ldk, gm.d);
    tv.load(vb, gm.ldv, gm.dv);
    tk.store(Ks[0]);
    tv.store(Vs[0]);
    __syncthreads();

    int ktr[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) a16_tr_offsets(lane, t, ktr[t]);
    const uint32_t ks_lds = a16_lds_addr(&Ks[0][0]);

    a16_f32x16 dqacc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[t][r] = 0.f;

    const int ntiles = gm.Nk / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntiles) {
            tk.load(kb + (int64_t)(it + 1) * 32 * gm.ldk, gm.ldk, gm.d);
            tv.load(vb + (int64_t)(it + 1) * 32 * gm.ldv, gm.ldv, gm.dv);
        }
        a16_f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = 0.f;
            dp[r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < DQT; ++s) st = A16_MFMA(a16_row_frag(Ks[cur], lane, 2 * s + half), qf[s], st);
#pragma unroll
        for (int s = 0; s < 2 * DVT; ++s) dp = A16_MFMA(a16_row_frag(Vs[cur], lane, 2 * s + half), gf[s], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - lse_q) * (dp[r] - delta_q);
        const uint32_t kcur = ks_lds + cur * A16_IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const a16_bf16x8 dsb = a16_frag_of_acc(st, s);
#pragma unroll
            for (int t = 0; t < MT; ++t) dqacc[t] = A16_MFMA(a16_tr_frag(kcur, ktr[t], s), dsb, dqacc[t]);
        }
        if (it + 1 < ntiles) {
            tk.store(Ks[cur ^ 1]);
            tv.store(Vs[cur ^ 1]);
        }
        __syncthreads();
    }
    __bf16* qo = dq + b * sdq + (int64_t)(q0 + col) * lddq;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < gm.d) a16_store4(qo + c0, dqacc[t][4 * g], dqacc[t][4 * g + 1], dqacc[t][4 * g + 2], dqacc[t][4 * g + 3]);
        }
}

// ------------------------------------------------------------------------------------------
// backward, dK and dV: grid (Nk / 128, B); a wave owns 32 keys and walks the queries
//   S = Q K^T ; P = exp(S - lse_row) ; dV^T += dO^T P ; dP = dO V^T ; dS = P (dP - delta_row) ; dK^T += Q^T dS
// ------------------------------------------------------------------------------------------
template <int DQT, int DVT>
__global__ __launch_bounds__(256) void attn16_bwd_dkv_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                             const __bf16* __restrict__ v,
                                                             const __bf16* __restrict__ dout,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta, __bf16* __restrict__ dk,
                                                             __bf16* __restrict__ dvo, const A16Geom gm, int64_t ldg,
                                                             int64_t sg, int64_t lddk, int64_t sdk, int64_t lddv,
                                                             int64_t sdv) {
    constexpr int GIMG = (DVT + 3) / 4;
    constexpr int MT = (DQT + 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char Qs[2][A16_IMG];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[2][GIMG * A16_IMG];
    __shared__ float Ls[2][32];
    __shared__ float Ds[2][32];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int k0 = blockIdx.x * 128 + wave * 32;

    a16_bf16x8 kf[DQT], vf[2 * DVT];
    a16_load_row_regs<DQT>(k + b * gm.sk + (int64_t)(k0 + col) * gm.ldk, gm.d, half, kf);
    a16_load_row_regs<2 * DVT>(v + b * gm.sv + (int64_t)(k0 + col) * gm.ldv, gm.dv, half, vf);
    const __bf16* qb = q + b * gm.sq;
    const __bf16* gb = dout + b * sg;
    const float* lb = lse + (int64_t)b * gm.N;
    const float* db = delta + (int64_t)b * gm.N;
    A16Tile<16 * DQT> tq;
    A16Tile<32 * DVT> tg;
    tq.load(qb, gm.ldq, gm.d);
    tg.load(gb, ldg, gm.dv);
    float l_pf = 0.f, d_pf = 0.f;
    if (threadIdx.x < 32) {
        l_pf = lb[threadIdx.x];
        d_pf = db[threadIdx.x];
    }
    tq.store(Qs[0]);
    tg.store(Gs[0]);
    if (threadIdx.x < 32) {
        Ls[0][threadIdx.x] = l_pf;
        Ds[0][threadIdx.x] = d_pf;
    }
    __syncthreads();

    int gtr[DVT][2], qtr[MT][2];
#pragma unroll
    for (int t = 0; t < DVT; ++t) a16_tr_offsets(lane, t, gtr[t]);
#pragma unroll
    for (int t = 0; t < MT; ++t) a16_tr_offsets(lane, t, qtr[t]);
    const uint32_t gs_lds = a16_lds_addr(&Gs[0][0]);
    const uint32_t qs_lds = a16_lds_addr(&Qs[0][0]);

    a16_f32x16 dvacc[DVT], dkacc[MT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dvacc[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[t][r] = 0.f;

    const int ntiles = gm.N / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        if (it + 1 < ntiles) {
            tq.load(qb + (int64_t)(it + 1) * 32 * gm.ldq, gm.ldq, gm.d);
            tg.load(gb + (int64_t)(it + 1) * 32 * ldg, ldg, gm.dv);
            if (threadIdx.x < 32) {
                l_pf = lb[(it + 1) * 32 + threadIdx.x];
                d_pf = db[(it + 1) * 32 + threadIdx.x];
            }
        }
        a16_f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = 0.f;
            dp[r] = 0.f;
        }
        // rows = queries of the tile, columns = this wave's keys
#pragma unroll
        for (int s = 0; s < DQT; ++s) st = A16_MFMA(a16_row_frag(Qs[cur], lane, 2 * s + half), kf[s], st);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] - Ls[cur][a16_acc_row(r, half)]);
        const uint32_t gcur = gs_lds + cur * (GIMG * A16_IMG);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const a16_bf16x8 pb = a16_frag_of_acc(st, s);
#pragma unroll
            for (int t = 0; t < DVT; ++t) dvacc[t] = A16_MFMA(a16_tr_frag(gcur, gtr[t], s), pb, dvacc[t]);
        }
#pragma unroll
        for (int s = 0; s < 2 * DVT; ++s) dp = A16_MFMA(a16_row_frag(Gs[cur], lane, 2 * s + half), vf[s], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] *= dp[r] - Ds[cur][a16_acc_row(r, half)];
        const uint32_t qcur = qs_lds + cur * A16_IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const a16_bf16x8 dsb = a16_frag_of_acc(st, s);
#pragma unroll
            for (int t = 0; t < MT; ++t) dkacc[t] = A16_MFMA(a16_tr_frag(qcur, qtr[t], s), dsb, dkacc[t]);
        }
        if (it + 1 < ntiles) {
            tq.store(Qs[cur ^ 1]);
            tg.store(Gs[cur ^ 1]);
            if (threadIdx.x < 32) {
                Ls[cur ^ 1][threadIdx.x] = l_pf;
                Ds[cur ^ 1][threadIdx.x] = d_pf;
            }
        }
        __syncthreads();
    }
    __bf16* ko = dk + b * sdk + (int64_t)(k0 + col) * lddk;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < gm.d) a16_store4(ko + c0, dkacc[t][4 * g], dkacc[t][4 * g + 1], dkacc[t][4 * g + 2], dkacc[t][4 * g + 3]);
        }
    __bf16* vo = dvo + b * sdv + (int64_t)(k0 + col) * lddv;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = t * 32 + 8 * g + 4 * half;
            if (c0 < gm.dv) a16_store4(vo + c0, dvacc[t][4 * g], dvacc[t][4 * g + 1], dvacc[t][4 * g + 2], dvacc[t][4 * g + 3]);
        }
}

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
static bool a16_shape_ok(int N, int Nk, int d, int dv) {
    return N > 0 && Nk > 0 && N % 128 == 0 && Nk % 128 == 0 && d > 0 && dv > 0 && d % 4 == 0 && dv % 4 == 0 && d <= 64 &&
           dv <= 256;
}

template <int DQT, int DVT>
static void a16_launch_fwd(hipStream_t s, const __bf16* q, const __bf16* k, const __bf16* v, __bf16* o, float* lse, int B,
                           const A16Geom& gm) {
    prof_kernel("attn16_fwd_kernel<%d, %d>", DQT, DVT);
    hipLaunchKernelGGL((attn16_fwd_kernel<DQT, DVT>), dim3(gm.N / 128, B), dim3(256), 0, s, q, k, v, o, lse, gm);
}
template <int DQT, int DVT>
static void a16_launch_bwd(hipStream_t s, const __bf16* q, const __bf16* k, const __bf16* v, const __bf16* dout,
                           const float* lse, const float* delta, __bf16* dq, __bf16* dk, __bf16* dvo, int B,
                           const A16Geom& gm, int64_t ldg, int64_t sg, int64_t lddq, int64_t sdq, int64_t lddk, int64_t sdk,
                           int64_t lddv, int64_t sdv) {
    prof_kernel(dq ? "attn16_bwd_dq_kernel<%d, %d>" : "attn16_bwd_dkv_kernel<%d, %d>", DQT, DVT);
    if (dq)
        hipLaunchKernelGGL((attn16_bwd_dq_kernel<DQT, DVT>), dim3(gm.N / 128, B), dim3(256), 0, s, q, k, v, dout, lse, delta,
                           dq, gm, ldg, sg, lddq, sdq);
    if (dk)
        hipLaunchKernelGGL((attn16_bwd_dkv_kernel<DQT, DVT>), dim3(gm.Nk / 128, B), dim3(256), 0, s, q, k, v, dout, lse,
                           delta, dk, dvo, gm, ldg, sg, lddk, sdk, lddv, sdv);
}

// (d, dv) -> (DQT, DVT) instances: every BASELINE topology's pair plus a generic ladder
#define A16_DISPATCH(FN, ...)                                                       \
    do {                                                                            \
        const int dqt = (gm.d + 15) / 16, dvt = (gm.dv + 31) / 32;                  \
        if (dqt <= 1) {                                                             \
            if (dvt <= 1) FN<1, 1>(__VA_ARGS__);                                    \
            else if (dvt <= 2) FN<1, 2>(__VA_ARGS__);                               \
            else FN<1, 4>(__VA_ARGS__);                                             \
        } else if (dqt <= 2) {                                                      \
            if (dvt <= 2) FN<2, 2>(__VA_ARGS__);                                    \
            else if (dvt <= 3) FN<2, 3>(__VA_ARGS__);                               \
            else if (dvt <= 4) FN<2, 4>(__VA_ARGS__);                               \
            else FN<2, 8>(__VA_ARGS__);                                             \
        } else {                                                                    \
            if (dvt <= 4) FN<4, 4>(__VA_ARGS__);                                    \
            else if (dvt <= 6) FN<4, 6>(__VA_ARGS__);                               \
            else FN<4, 8>(__VA_ARGS__);                                             \
        }                                                                           \
    } while (0)

}  // namespace bg

using namespace bg;

extern "C" {

int bg_attention16_supported(int N, int Nk, int d, int dv) { return a16_shape_ok(N, Nk, d, dv) ? 1 : 0; }

static int a16_check(const char* who, const BgAttn16Desc* g) {
    BG_REQUIRE(g != nullptr && g->B > 0, "%s: null / empty descriptor", who);
    BG_REQUIRE(a16_shape_ok(g->N, g->Nk, g->d, g->dv),
               "%s: unsupported shape N=%d Nk=%d d=%d dv=%d (need N, Nk %% 128 == 0, d, dv %% 4 == 0, d <= 64, dv <= 256)", who,
               g->N, g->Nk, g->d, g->dv);
    BG_REQUIRE(g->ldq % 4 == 0 && g->ldk % 4 == 0 && g->ldv % 4 == 0 && g->ldo % 4 == 0 && g->ldq >= g->d &&
                   g->ldk >= g->d && g->ldv >= g->dv && g->ldo >= g->dv,
               "%s: row strides must be multiples of 4 elements and cover the row", who);
    return BG_OK;
}

static A16Geom a16_geom(const BgAttn16Desc* g) {
    A16Geom gm;
    gm.N = g->N; gm.Nk = g->Nk; gm.d = g->d; gm.dv = g->dv;
    gm.ldq = g->ldq; gm.ldk = g->ldk; gm.ldv = g->ldv; gm.ldo = g->ldo;
    gm.sq = g->sq; gm.sk = g->sk; gm.sv = g->sv; gm.so = g->so;
    return gm;
}

int bg_attention16_fwd(const BgAttn16Desc* g, const void* q, const void* k, const void* v, void* o, float* lse,
                       void* stream) {
    int rc = a16_check("bg_attention16_fwd", g);
    if (rc) return rc;
    BG_REQUIRE(q && k && v && o && lse, "bg_attention16_fwd: null tensor pointer");
    BG_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 7) == 0,
               "bg_attention16_fwd: pointers must be 8-byte aligned");
    hipStream_t s = as_stream(stream);
    const A16Geom gm = a16_geom(g);
    const int B = g->B;
    const double abytes = 2.0 * B * ((double)gm.N * (gm.d + gm.dv) + (double)gm.Nk * (gm.d + gm.dv)) + 4.0 * B * gm.N;
    ProfScope prof(s, 2.0 * B * (double)gm.N * gm.Nk * (gm.d + gm.dv), "attention16_fwd", abytes);
    A16_DISPATCH(a16_launch_fwd, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (__bf16*)o, lse, B, gm);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_attention16_bwd(const BgAttn16Desc* g, const void* q, const void* k, const void* v, const void* o, const void* dout,
                       const float* lse, void* dq, void* dk, void* dv_out, float* delta_ws, void* stream) {
    int rc = a16_check("bg_attention16_bwd", g);
    if (rc) return rc;
    BG_REQUIRE(q && k && v && o && dout && lse && delta_ws, "bg_attention16_bwd: null tensor pointer");
    BG_REQUIRE((dk != nullptr) == (dv_out != nullptr) && (dq || dk), "bg_attention16_bwd: dk and dv come together; nothing to do");
    BG_REQUIRE(g->ldg % 4 == 0 && g->lddq % 4 == 0 && g->lddk % 4 == 0 && g->lddv % 4 == 0 && g->ldg >= g->dv &&
                   g->lddq >= g->d && g->lddk >= g->d && g->lddv >= g->dv,
               "bg_attention16_bwd: gradient row strides must be multiples of 4 elements and cover the row");
    BG_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk |
                 (uintptr_t)dv_out) & 7) == 0, "bg_attention16_bwd: pointers must be 8-byte aligned");
    hipStream_t s = as_stream(stream);
    const A16Geom gm = a16_geom(g);
    const int B = g->B;
    const int64_t rows = (int64_t)B * gm.N;
    if (dk) {       // delta_ws is (re)computed by the call that produces dk / dv and reused by a later dq-only call
        hipLaunchKernelGGL(attn16_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const __bf16*)o,
                           (const __bf16*)dout, delta_ws, B, gm.N, gm.dv, gm.ldo, gm.so, g->ldg, g->sg);
        BG_LAUNCH_CHECK();
    }
    // algorithmic backward work = dV, dP, dQ, dK = 2 x forward over BOTH calls of one backward (the dk / dv call and the
    // dq call each recompute S; the recomputation is not counted): dk / dv call d + 2 dv, dq call d
    const double aflops = 2.0 * B * (double)gm.N * gm.Nk * ((dk ? gm.d + 2.0 * gm.dv : 0.0) + (dq ? (double)gm.d : 0.0));
    const double abytes = 2.0 * B * ((double)gm.N * (gm.d + 2.0 * gm.dv) + (double)gm.Nk * (gm.d + gm.dv)) +
                          2.0 * B * ((dq ? (double)gm.N * gm.d : 0.0) + (dk ? (double)gm.Nk * (gm.d + gm.dv) : 0.0));
    ProfScope prof(s, aflops, "attention16_bwd", abytes);
    A16_DISPATCH(a16_launch_bwd, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)dout, lse, delta_ws,
                 (__bf16*)dq, (__bf16*)dk, (__bf16*)dv_out, B, gm, g->ldg, g->sg, g->lddq, g->sdq, g->lddk, g->sdk, g->lddv,
                 g->sdv);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
