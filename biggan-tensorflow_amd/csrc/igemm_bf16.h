// igemm_bf16.h - bf16-compute variants of the two implicit-GEMM kernels (v_mfma_f32_32x32x16_bf16,
// fp32 accumulate).  Activations, weights and outputs stay fp32 in HBM (the reference's dtype,
// ops.py:14); operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are staged into LDS.
// This is the first step towards BASELINE config 3 (bf16): the MFMA rate is 16x the fp32 one, so
// these kernels are bound by operand traffic (L2 -> LDS), not by the matrix pipe.
//
// LDS images are [row][k] with 32 bf16 (64 B) of payload per row padded to 80 B: the 16-byte operand
// reads of a 32x32x16 MFMA (lane = row, 8 consecutive k) then hit 16 distinct 4-bank groups per
// 16-lane read group, and the transposing 8-byte writes (lanes 0-7 consecutive k, lanes 8-15 the next
// row quad, 4 * 20 dwords = 16 mod 32) are conflict-free as well.
#pragma once
#include "igemm_dev.h"

namespace bg {

typedef float floatx16_b __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BKB 32       // K tile (elements)
#define LROW 40      // bf16 elements per LDS row (80 bytes)

__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}

__device__ __forceinline__ bf16x4 cvt4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
    return r;
}

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// 4 rows x 16 columns of 16-bit elements, delivered column-major: lane i of a 16-lane group gets column i
__device__ __forceinline__ s16x4_t lds_read_tr16(uint32_t lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(lds_byte_addr));
}

__device__ __forceinline__ int tr_swz_off(int row, int chunk) {
    return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// TM x TN MFMA tiles over one 32-deep K tile: 2 k-steps of 16
template <int TM, int TN>
__device__ __forceinline__ void mma_tile_bf16(const __bf16* __restrict__ as, const __bf16* __restrict__ bs, int a_rd,
                                              int b_rd, floatx16_b (&acc)[TM][TN]) {
#pragma unroll
    for (int s = 0; s < BKB / 16; ++s) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(as + a_rd + 32 * i * LROW + s * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bs + b_rd + 32 * j * LROW + s * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// NN kernel, bf16 compute (vector path only: C % 4 == 0, 16-byte aligned operands)
// ------------------------------------------------------------------------------------------
// BTR (only !BT, TN == 2): the [k][n] weight tile stays k-major in LDS (32 rows of 128 bf16, swizzled like the wgrad
// kernel's images) and the B operand is read through ds_read_b64_tr_b16 instead of being transposed by the staging code.
template <int TM, int TN, bool BT, int MODE, bool MIRROR, bool BTR = false>
__global__ __launch_bounds__(256) void nn_kernel_bf16(const NNParams p) {
    static_assert(!BTR || (!BT && TN == 2), "BTR: non-BT weights, 128-wide tile");
    constexpr int WN = 2;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AITEMS = BM * 4 / 256;            // (row, k8) items per thread
    constexpr int NSRC = MIRROR ? 4 : 1;
    constexpr int BITEMS_T = BN * 4 / 256;          // BT: (n, k8) items per thread
    constexpr int BPATCH = (BKB / 4) * (BN / 4);    // non-BT: 4k x 4n patches in the tile
    constexpr int BITEMS_N = (BPATCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN * LROW];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const Gather& g = p.g;
    int tile, bz = blockIdx.z;
    if (p.zfold > 0) {
        const int lin = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.zfold);
        bz = lin % p.zfold;
        tile = lin / p.zfold;
    } else {
        tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int zs = bz % p.splitk, zo = bz / p.splitk;
    const float* Abase = p.A;
    const float* Bbase = p.B;
    float* Obase = p.out;
    int ph = 0, pw = 0;
    if (MODE == GATHER_PLAIN) {
        Abase += (int64_t)zo * p.strideA;
        Bbase += (int64_t)zo * p.strideB;
        Obase += (int64_t)zo * p.strideC;
    } else if (MODE == GATHER_TCONV) {
        ph = g.pstep - 1 - zo / g.pstep;
        pw = g.pstep - 1 - zo % g.pstep;
    }
    int kh0 = 0, kw0 = 0, kstep = 1, nkh = g.k, nkw = g.k;
    if (MODE == GATHER_TCONV && g.pstep > 1) {
        kstep = g.stride;
        kh0 = (ph + g.pad) % g.stride;
        kw0 = (pw + g.pad) % g.stride;
        nkh = (g.k - kh0 + g.stride - 1) / g.stride;
        nkw = (g.k - kw0 + g.stride - 1) / g.stride;
    }
    if (MODE == GATHER_PLAIN) {
        nkh = 1;
        nkw = 1;
    }
    const int kc = (p.C + BKB - 1) / BKB;
    const int niter_all = nkh * nkw * kc;
    const int ips = (niter_all + p.splitk - 1) / p.splitk;
    const int it0 = zs * ips;
    const int it1 = min(niter_all, it0 + ips);
    const int niter = max(0, it1 - it0);

    const int a_k8 = (t & 3) * 8;
    RowPos rows[AITEMS];
#pragma unroll
    for (int i = 0; i < AITEMS; ++i) rows[i] = decompose_row<MODE>(g, m0 + (t >> 2) + 64 * i, p.M, ph, pw);
    int64_t aoff[AITEMS][NSRC];

    // K order: channel chunks outermost, taps inside a chunk, K-steps of the chunk innermost.  The taps of
    // one output tile gather overlapping input pixels, so walking all taps over a narrow channel chunk
    // keeps the re-reads inside the XCD's 4 MB L2 instead of streaming the whole input once per tap.
    const int kci = p.kchunk > 0 ? min(kc, p.kchunk) : kc;      // K-steps per chunk
    const int nchunk = (kc + kci - 1) / kci;
    const int ntap = nkh * nkw;
    int l_chunk = min(it0 / (ntap * kci), nchunk - 1);
    int l_csteps = min(kci, kc - l_chunk * kci);                // K-steps in the current chunk
    int l_ic, l_iw, l_ih;
    {
        const int rem = it0 - l_chunk * ntap * kci;
        const int tap = rem / l_csteps;
        l_ic = rem - tap * l_csteps;
        l_iw = tap % nkw;
        l_ih = tap / nkw;
    }
    bool need_off = true;
    float4 ra[AITEMS][2];
    constexpr int BREG = BT ? BITEMS_T * 2 : BITEMS_N * 4;
    float4 rb[BREG];

    auto load_tile = [&]() {
        const int kh = kh0 + l_ih * kstep, kw = kw0 + l_iw * kstep;
        if (need_off || l_ic == 0) {
#pragma unroll
            for (int i = 0; i < AITEMS; ++i) tap_sources<MODE, MIRROR>(g, rows[i], kh, kw, aoff[i]);
            need_off = false;
        }
        const int c0 = (l_chunk * kci + l_ic) * BKB;
        {
#pragma unroll
        for (int i = 0; i < AITEMS; ++i) {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                float4 v = load_chan4<true>(Abase, aoff[i][0], c0 + a_k8 + 4 * hlf, p.C);
                if (MIRROR) {
#pragma unroll
                    for (int s = 1; s < NSRC; ++s)
                        add4(v, load_chan4<true>(Abase, aoff[i][s], c0 + a_k8 + 4 * hlf, p.C));
                }
                ra[i][hlf] = v;
            }
        }
        const float* wt = Bbase + (int64_t)(kh * g.k + kw) * p.tap_stride;
        if (BT) {
#pragma unroll
            for (int i = 0; i < BITEMS_T; ++i) {
                const int n = n0 + (t >> 2) + 64 * i;
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int c = c0 + a_k8 + 4 * hlf;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n < p.N && c < p.C) v = ld4(wt + (int64_t)n * p.ldn + c);
                    rb[i * 2 + hlf] = v;
                }
            }
        } else if (BTR) {
            // items: k row (t + 256 i) >> 4, 16-byte chunk (t + 256 i) & 15 = 8 consecutive n: coalesced along n
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = t + 256 * i;
                const int c = c0 + (idx >> 4), n = n0 + (idx & 15) * 8;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (c < p.C) {
                    const float* src = wt + (int64_t)c * p.ldk + n;
                    if (n < p.N) v0 = ld4(src);
                    if (n + 4 < p.N) v1 = ld4(src + 4);
                }
                rb[i * 2] = v0;
                rb[i * 2 + 1] = v1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BITEMS_N; ++i) {
                const int idx = t + 256 * i;
                const int k4 = idx & 7, nq = idx >> 3;          // lanes 0-7: consecutive k quads
                const int n = n0 + nq * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + k4 * 4 + j;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < BPATCH && c < p.C && n < p.N) v = ld4(wt + (int64_t)c * p.ldk + n);
                    rb[i * 4 + j] = v;
                }
            }
        }
        }
        if (++l_ic == l_csteps) {
            l_ic = 0;
            if (++l_iw == nkw) {
                l_iw = 0;
                if (++l_ih == nkh) {
                    l_ih = 0;
                    ++l_chunk;
                    l_csteps = min(kci, kc - l_chunk * kci);
                }
            }
        }
    };

    auto store_tile = [&](int buf) {
        __bf16* as = As[buf];
        __bf16* bs = Bs[buf];
#pragma unroll
        for (int i = 0; i < AITEMS; ++i) {
            const int r = (t >> 2) + 64 * i;
            *reinterpret_cast<bf16x8*>(as + r * LROW + a_k8) = cvt8(ra[i][0], ra[i][1]);
        }
        if (BT) {
#pragma unroll
            for (int i = 0; i < BITEMS_T; ++i) {
                const int n = (t >> 2) + 64 * i;
                *reinterpret_cast<bf16x8*>(bs + n * LROW + a_k8) = cvt8(rb[i * 2], rb[i * 2 + 1]);
            }
        } else if (BTR) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = t + 256 * i;
                *reinterpret_cast<bf16x8*>(reinterpret_cast<unsigned char*>(bs) + tr_swz_off(idx >> 4, idx & 15)) =
                    cvt8(rb[i * 2], rb[i * 2 + 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < BITEMS_N; ++i) {
                const int idx = t + 256 * i;
                if (idx < BPATCH) {
                    const int k4 = idx & 7, nq = idx >> 3;
                    const float4 r0 = rb[i * 4], r1 = rb[i * 4 + 1], r2 = rb[i * 4 + 2], r3 = rb[i * 4 + 3];
                    __bf16* base = bs + (nq * 4) * LROW + k4 * 4;
                    *reinterpret_cast<bf16x4*>(base) = cvt4(r0.x, r1.x, r2.x, r3.x);
                    *reinterpret_cast<bf16x4*>(base + LROW) = cvt4(r0.y, r1.y, r2.y, r3.y);
                    *reinterpret_cast<bf16x4*>(base + 2 * LROW) = cvt4(r0.z, r1.z, r2.z, r3.z);
                    *reinterpret_cast<bf16x4*>(base + 3 * LROW) = cvt4(r0.w, r1.w, r2.w, r3.w);
                }
            }
        }
    };

    floatx16_b acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (niter > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();

    // operand reads: lane l -> row (l & 31), k offset 8 * (l >> 5)
    const int a_rd = (wm * 32 * TM + (lane & 31)) * LROW + (lane >> 5) * 8;
    const int b_rd = (wn * 32 * TN + (lane & 31)) * LROW + (lane >> 5) * 8;
    uint32_t btr_ad[TN][2];
    if constexpr (BTR) {
        const int kblk = lane >> 5, g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
        const uint32_t b_lds = static_cast<uint32_t>(
            reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)&Bs[0][0]));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                btr_ad[j][jj] = b_lds + tr_swz_off(8 * kblk + 4 * jj + q, 4 * (wn * TN + j) + 2 * g16 + (pq >> 1)) +
                                8 * (pq & 1);
    }

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) load_tile();
        if constexpr (BTR) {
#pragma unroll
            for (int s2 = 0; s2 < BKB / 16; ++s2) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const bf16x8*>(As[cur] + a_rd + 32 * i * LROW + s2 * 16);
                const uint32_t boff = (uint32_t)cur * (uint32_t)(BN * LROW * 2) + 4096u * s2;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const s16x4_t lo = lds_read_tr16(btr_ad[j][0] + boff), hi = lds_read_tr16(btr_ad[j][1] + boff);
                    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    b[j] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            mma_tile_bf16<TM, TN>(As[cur], Bs[cur], a_rd, b_rd, acc);
        }
        if (it + 1 < niter) store_tile(cur ^ 1);
        __syncthreads();
    }

    const bool partial = p.splitk > 1;
    const float alpha = (!partial && p.alpha) ? *p.alpha : 1.0f;
    if (partial) Obase = p.slabs + (int64_t)zs * p.slab_stride + (Obase - p.out);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.M) continue;
            int64_t ooff;
            if (MODE != GATHER_TCONV) {
                ooff = (int64_t)row * p.out_ld;
            } else {
                RowPos rp = decompose_row<MODE>(g, row, p.M, ph, pw);
                ooff = (((int64_t)rp.b * g.Ho + rp.ho) * g.Wo + rp.wo) * p.out_ld;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 32 * TN + 32 * j + (lane & 31);
                if (col < p.N) {
                    float v = acc[i][j][r] * alpha;
                    float* o = Obase + ooff + col;
                    if (!partial) {
                        if (p.bias) v += p.bias[col];
                        if (p.accumulate) v += *o;
                    }
                    *o = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel, bf16 compute: out[(tap, ca)][cb] = sum_pixels A(pixel, tap)[ca] * Bv(pixel)[cb]
// both operands are transposed on the way into LDS ([channel][pixel] images)
// ------------------------------------------------------------------------------------------
template <int TM, int TN, int MODE>
__global__ __launch_bounds__(256) void tn_kernel_bf16(const TNParams p) {
    constexpr int WN = 2;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int APATCH = (BKB / 4) * (BM / 4), BPATCH = (BKB / 4) * (BN / 4);
    constexpr int AIT = (APATCH + 255) / 256, BIT = (BPATCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN * LROW];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const Gather& g = p.g;

    int tile, bz = blockIdx.z;
    if (p.zfold > 0) {
        // split-major logical ids: every XCD owns whole pixel ranges (splits), so the slice of x / dy a
        // split reduces over is fetched into ONE L2 and shared by all its (M, N) tiles
        const int ntile = p.tiles_m * p.tiles_n;
        const int lin = xcd_remap(blockIdx.x, ntile * p.zfold);
        bz = lin / ntile;
        tile = lin - bz * ntile;
    } else {
        tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int mf0 = tile_m * BM, cb0 = tile_n * BN;
    const int zb = bz / p.splitk, zs = bz % p.splitk;
    const float* Abase = p.A + (int64_t)zb * p.strideA;
    const float* Bbase = p.Bv + (int64_t)zb * p.strideB;

    const int row_begin = zs * p.rows_per_split;
    const int row_end = min(p.M, row_begin + p.rows_per_split);
    const int niter = max(0, (row_end - row_begin + BKB - 1) / BKB);

    // patch (k4, cq): pixels l_row + 4*k4 .. +3, channels 4*cq .. +3 ; lanes 0-7 take consecutive k4
    int a_kh[AIT], a_kw[AIT], a_c[AIT];
    RowPos apos[AIT][4];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int idx = t + 256 * i;
        const int k4 = idx & 7, cq = idx >> 3;
        const int mf = mf0 + cq * 4;
        int tap = 0, c = mf;
        if (MODE != GATHER_PLAIN) {
            tap = mf / p.Ca;
            c = mf - tap * p.Ca;
        }
        a_kh[i] = tap / g.k;
        a_kw[i] = tap % g.k;
        a_c[i] = (idx < APATCH && mf < p.Mf) ? c : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) apos[i][j] = decompose_row<MODE>(g, row_begin + k4 * 4 + j, p.M, 0, 0);
    }

    float4 ra[AIT][4], rb[BIT][4];
    int l_row = row_begin;

    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int k4 = (t + 256 * i) & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = l_row + k4 * 4 + j;
                RowPos rp = apos[i][j];
                rp.valid = m < p.M;
                if (MODE == GATHER_PLAIN) {
                    apos[i][j].b += BKB;
                } else {
                    apos[i][j].wo += BKB;
                    while (apos[i][j].wo >= g.Wq) {
                        apos[i][j].wo -= g.Wq;
                        if (++apos[i][j].ho == g.Hq) {
                            apos[i][j].ho = 0;
                            ++apos[i][j].b;
                        }
                    }
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_c[i] >= 0 && m < row_end) {
                    int64_t off[1];
                    tap_sources<MODE, false>(g, rp, a_kh[i], a_kw[i], off);
                    v = load_chan4<true>(Abase, off[0], a_c[i], p.Ca);
                }
                ra[i][j] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int idx = t + 256 * i;
            const int k4 = idx & 7, cq = idx >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = l_row + k4 * 4 + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < BPATCH && m < row_end) v = load_chan4<true>(Bbase, (int64_t)m * p.b_ld, cb0 + cq * 4, p.Cb);
                rb[i][j] = v;
            }
        }
        l_row += BKB;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int idx = t + 256 * i;
            if (idx < APATCH) {
                const int k4 = idx & 7, cq = idx >> 3;
                __bf16* base = As[buf] + (cq * 4) * LROW + k4 * 4;
                *reinterpret_cast<bf16x4*>(base) = cvt4(ra[i][0].x, ra[i][1].x, ra[i][2].x, ra[i][3].x);
                *reinterpret_cast<bf16x4*>(base + LROW) = cvt4(ra[i][0].y, ra[i][1].y, ra[i][2].y, ra[i][3].y);
                *reinterpret_cast<bf16x4*>(base + 2 * LROW) = cvt4(ra[i][0].z, ra[i][1].z, ra[i][2].z, ra[i][3].z);
                *reinterpret_cast<bf16x4*>(base + 3 * LROW) = cvt4(ra[i][0].w, ra[i][1].w, ra[i][2].w, ra[i][3].w);
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int idx = t + 256 * i;
            if (idx < BPATCH) {
                const int k4 = idx & 7, cq = idx >> 3;
                __bf16* base = Bs[buf] + (cq * 4) * LROW + k4 * 4;
                *reinterpret_cast<bf16x4*>(base) = cvt4(rb[i][0].x, rb[i][1].x, rb[i][2].x, rb[i][3].x);
                *reinterpret_cast<bf16x4*>(base + LROW) = cvt4(rb[i][0].y, rb[i][1].y, rb[i][2].y, rb[i][3].y);
                *reinterpret_cast<bf16x4*>(base + 2 * LROW) = cvt4(rb[i][0].z, rb[i][1].z, rb[i][2].z, rb[i][3].z);
                *reinterpret_cast<bf16x4*>(base + 3 * LROW) = cvt4(rb[i][0].w, rb[i][1].w, rb[i][2].w, rb[i][3].w);
            }
        }
    };

    floatx16_b acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (niter > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();

    const int a_rd = (wm * 32 * TM + (lane & 31)) * LROW + (lane >> 5) * 8;
    const int b_rd = (wn * 32 * TN + (lane & 31)) * LROW + (lane >> 5) * 8;

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) load_tile();
        mma_tile_bf16<TM, TN>(As[cur], Bs[cur], a_rd, b_rd, acc);
        if (it + 1 < niter) store_tile(cur ^ 1);
        __syncthreads();
    }

    const float alpha = p.alpha ? *p.alpha : 1.0f;
    float* obase = p.out + (int64_t)zb * p.strideC + (int64_t)zs * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mf0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.Mf) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = cb0 + wn * 32 * TN + 32 * j + (lane & 31);
                if (col < p.Cb) obase[(int64_t)row * p.out_ld + col] = acc[i][j][r] * alpha;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel, bf16 compute, hardware-transposed operand reads (gfx950 ds_read_b64_tr_b16).
//   The tiles stay PIXEL-major in LDS ([32 pixels][128 channels] bf16, 256-byte rows, 16-byte chunks XOR-swizzled
//   by (row & 3) << 2 | (row >> 2) & 3): the staging pass is a conversion plus one conflict-free 16-byte write per
//   (pixel, 8 channels) item - no VALU transposition and half the gather index math of tn_kernel_bf16 - and the
//   MFMA operands (8 consecutive pixels of one channel per lane) come out of the LDS read path already transposed:
//   lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block and lane i receives column i.
//   128 x 128 tiles, Ca % 8 == 0, Cb % 8 == 0; everything else as tn_kernel_bf16.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void tn_kernel_bf16_tr(const TNParams p) {
    constexpr int TM = 2, TN = 2, WN = 2, BM = 128, BN = 128;
    constexpr int TILE_B = BKB * 256;                 // bytes of one [32][128] bf16 image
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][TILE_B];
    __shared__ __attribute__((aligned(16))) unsigned char Bp[2][TILE_B];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const Gather& g = p.g;

    int tile, bz = blockIdx.z;
    if (p.zfold > 0) {
        const int ntile = p.tiles_m * p.tiles_n;
        const int lin = xcd_remap(blockIdx.x, ntile * p.zfold);
        bz = lin / ntile;
        tile = lin - bz * ntile;
    } else {
        tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int mf0 = tile_m * BM, cb0 = tile_n * BN;
    const int zb = bz / p.splitk, zs = bz % p.splitk;
    const float* Abase = p.A + (int64_t)zb * p.strideA;
    const float* Bbase = p.Bv + (int64_t)zb * p.strideB;

    const int row_begin = zs * p.rows_per_split;
    const int row_end = min(p.M, row_begin + p.rows_per_split);
    const int niter = max(0, (row_end - row_begin + BKB - 1) / BKB);

    // staging items: idx = t + 256 i -> pixel idx >> 4 of the K tile, 16-byte chunk idx & 15 (8 channels)
    int a_kh[2], a_kw[2], a_c[2], b_c[2], st_off[2], pix[2];
    RowPos apos[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        const int chunk = idx & 15;
        pix[i] = idx >> 4;
        const int mf = mf0 + chunk * 8;
        int tap = 0, c = mf;
        if (MODE != GATHER_PLAIN) {
            tap = mf / p.Ca;
            c = mf - tap * p.Ca;
        }
        a_kh[i] = tap / g.k;
        a_kw[i] = tap % g.k;
        a_c[i] = mf < p.Mf ? c : -1;
        b_c[i] = cb0 + chunk * 8;
        st_off[i] = tr_swz_off(pix[i], chunk);
        apos[i] = decompose_row<MODE>(g, row_begin + pix[i], p.M, 0, 0);
    }

    float4 ra[2][2], rb[2][2];
    int l_row = row_begin;

    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = l_row + pix[i];
            RowPos rp = apos[i];
            rp.valid = m < p.M;
            if (MODE == GATHER_PLAIN) {
                apos[i].b += BKB;
            } else {
                apos[i].wo += BKB;
                while (apos[i].wo >= g.Wq) {
                    apos[i].wo -= g.Wq;
                    if (++apos[i].ho == g.Hq) {
                        apos[i].ho = 0;
                        ++apos[i].b;
                    }
                }
            }
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (a_c[i] >= 0 && m < row_end) {
                int64_t off[1];
                tap_sources<MODE, false>(g, rp, a_kh[i], a_kw[i], off);
                v0 = load_chan4<true>(Abase, off[0], a_c[i], p.Ca);
                v1 = load_chan4<true>(Abase, off[0], a_c[i] + 4, p.Ca);
            }
            ra[i][0] = v0;
            ra[i][1] = v1;
            float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
            if (m < row_end) {
                w0 = load_chan4<true>(Bbase, (int64_t)m * p.b_ld, b_c[i], p.Cb);
                w1 = load_chan4<true>(Bbase, (int64_t)m * p.b_ld, b_c[i] + 4, p.Cb);
            }
            rb[i][0] = w0;
            rb[i][1] = w1;
        }
        l_row += BKB;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<bf16x8*>(Ap[buf] + st_off[i]) = cvt8(ra[i][0], ra[i][1]);
            *reinterpret_cast<bf16x8*>(Bp[buf] + st_off[i]) = cvt8(rb[i][0], rb[i][1]);
        }
    };

    floatx16_b acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (niter > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();

    // transposed-read addresses (buffer 0, k-step 0): lane -> k block lane >> 5, 16-channel group (lane >> 4) & 1,
    // block row q = (lane & 15) >> 2, column quad pq = lane & 3; read jj covers pixels 8 kblk + 4 jj .. + 3
    const int kblk = lane >> 5, g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
    const uint32_t a_lds = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)&Ap[0][0]));
    const uint32_t b_lds = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)&Bp[0][0]));
    uint32_t a_ad[TM][2], b_ad[TN][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int row = 8 * kblk + 4 * jj + q;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a_ad[i][jj] = a_lds + tr_swz_off(row, 4 * (wm * TM + i) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b_ad[j][jj] = b_lds + tr_swz_off(row, 4 * (wn * TN + j) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
    }

    auto operand = [&](uint32_t lo_addr, uint32_t hi_addr) {
        const s16x4_t lo = lds_read_tr16(lo_addr), hi = lds_read_tr16(hi_addr);
        const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mma = [&](uint32_t bufoff) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = operand(a_ad[i][0] + bufoff + 4096 * s, a_ad[i][1] + bufoff + 4096 * s);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = operand(b_ad[j][0] + bufoff + 4096 * s, b_ad[j][1] + bufoff + 4096 * s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) load_tile();
        mma(cur ? (uint32_t)TILE_B : 0u);
        if (it + 1 < niter) store_tile(cur ^ 1);
        __syncthreads();
    }

    const float alpha = p.alpha ? *p.alpha : 1.0f;
    float* obase = p.out + (int64_t)zb * p.strideC + (int64_t)zs * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mf0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.Mf) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = cb0 + wn * 32 * TN + 32 * j + (lane & 31);
                if (col < p.Cb) obase[(int64_t)row * p.out_ld + col] = acc[i][j][r] * alpha;
            }
        }
    }
}


}  // namespace bg
