// igemm.hip - fp32 MFMA implicit-GEMM kernels for conv / transposed conv / dense / attention
// matmuls (gfx950, v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//
// Reference call sites replaced: tf.pad+tf.nn.conv2d (ops.py:82,94-98), tf.nn.conv2d_transpose
// (ops.py:127-132), tf.matmul (ops.py:163-165, 481, 485; utils.py:198,222) and their gradients.
//
// Tiling: 256 threads = 4 waves (64 lanes).  Each wave owns TM x TN MFMA tiles of 32x32; the block
// tile is (32*TM*WM) x (32*TN*WN); K advances in steps of 16 floats through double-buffered LDS,
// with the next tile's global loads in flight (held in registers) while the current one is on the
// matrix pipe, and the MFMA operand reads of k-step kk+1 issued before the MFMAs of k-step kk.
// A tiles are stored K-major in LDS ([k][m], row stride == 2 mod 8) so that both the transposing
// ds_write_b32 (4 per float4) and the operand ds_read_b32 (lane = row) are bank-conflict free.
// Blocks are renumbered so that the tiles sharing operand panels run on one XCD (shared L2).
#include <stdlib.h>
#include "common.h"
#include "igemm.h"
#include "igemm_dev.h"
#include "igemm_bf16.h"
#include "igemm16.h"


namespace bg {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define BKT 16  // K tile (floats)

// TM x TN MFMA tiles of one wave over one 16-deep K tile, operand reads software-pipelined
template <int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mma_tile(const float* __restrict__ as, const float* __restrict__ bs, int a_rd, int b_rd,
                                         floatx16 (&acc)[TM][TN]) {
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = as[a_rd + 32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = bs[b_rd + 32 * j];
#pragma unroll
    for (int kk = 0; kk < BKT / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BKT / 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[nxt][i] = as[a_rd + 2 * (kk + 1) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[nxt][j] = bs[b_rd + 2 * (kk + 1) * LDB + 32 * j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// NN kernel
// ------------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN, bool BT, int MODE, bool MIRROR, bool VEC>
__global__ __launch_bounds__(256, (TM * TN == 4 && !MIRROR) ? 4 : 1) void nn_kernel(const NNParams p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDA = BM + 2;                 // == 2 (mod 8): conflict-free transposing writes
    constexpr int LDB = BT ? BN + 2 : BN + 4;   // BT: transposing writes ; else 16-byte aligned rows
    constexpr int AROWS = BM / 64;              // float4 A loads per thread per K tile
    constexpr int NSRC = MIRROR ? 4 : 1;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[2][BKT * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT * LDB];

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
        // heaviest phase first: for k3 s2 the odd phases have 2 taps per axis, the even ones 1
        ph = g.pstep - 1 - zo / g.pstep;
        pw = g.pstep - 1 - zo % g.pstep;
    }

    // tap enumeration of this phase
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
    const int kc = (p.C + BKT - 1) / BKT;
    const int niter_all = nkh * nkw * kc;
    const int ips = (niter_all + p.splitk - 1) / p.splitk;
    const int it0 = zs * ips;
    const int it1 = min(niter_all, it0 + ips);
    const int niter = max(0, it1 - it0);

    // A-load rows of this thread
    const int a_kq = (t & 3) * 4;
    RowPos rows[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) rows[i] = decompose_row<MODE>(g, m0 + (t >> 2) + 64 * i, p.M, ph, pw);
    int64_t aoff[AROWS][NSRC];

    // load-stream state, positioned at iteration it0
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
    float4 ra[AROWS];
    constexpr int BLOADS = BT ? (BN + 63) / 64 : (BKT * BN / 4 + 255) / 256;
    float4 rb[BLOADS];

    auto load_tile = [&]() {
        const int kh = kh0 + l_ih * kstep, kw = kw0 + l_iw * kstep;
        if (need_off || l_ic == 0) {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) tap_sources<MODE, MIRROR>(g, rows[i], kh, kw, aoff[i]);
            need_off = false;
        }
        const int c0 = (l_chunk * kci + l_ic) * BKT;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            float4 v = load_chan4<VEC>(Abase, aoff[i][0], c0 + a_kq, p.C);
            if (MIRROR) {
#pragma unroll
                for (int s = 1; s < NSRC; ++s) add4(v, load_chan4<VEC>(Abase, aoff[i][s], c0 + a_kq, p.C));
            }
            ra[i] = v;
        }
        const float* wt = Bbase + (int64_t)(kh * g.k + kw) * p.tap_stride;
        if (BT) {
            // weight is K-contiguous: B[c][n] at wt + c*1 + n*ldn ; float4 along c
#pragma unroll
            for (int i = 0; i < BLOADS; ++i) {
                const int nrel = (t >> 2) + 64 * i;
                const int n = n0 + nrel;
                const int c = c0 + a_kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nrel < BN && n < p.N) {
                    const float* q = wt + (int64_t)n * p.ldn + c;
                    if (VEC) {
                        if (c < p.C) v = ld4(q);
                    } else {
                        if (c + 0 < p.C) v.x = q[0];
                        if (c + 1 < p.C) v.y = q[1];
                        if (c + 2 < p.C) v.z = q[2];
                        if (c + 3 < p.C) v.w = q[3];
                    }
                }
                rb[i] = v;
            }
        } else {
            // weight is N-contiguous: B[c][n] at wt + c*ldk + n ; float4 along n
#pragma unroll
            for (int i = 0; i < BLOADS; ++i) {
                const int idx = t + 256 * i;
                const int kk = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int c = c0 + kk, n = n0 + nq;
                if (idx < BKT * BN / 4 && c < p.C) {
                    const float* q = wt + (int64_t)c * p.ldk + n;
                    if (VEC) {
                        if (n < p.N) v = ld4(q);
                    } else {
                        if (n + 0 < p.N) v.x = q[0];
                        if (n + 1 < p.N) v.y = q[1];
                        if (n + 2 < p.N) v.z = q[2];
                        if (n + 3 < p.N) v.w = q[3];
                    }
                }
                rb[i] = v;
            }
        }
        // advance the load stream
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
        float* as = As[buf];
        float* bs = Bs[buf];
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int r = (t >> 2) + 64 * i;
            as[(a_kq + 0) * LDA + r] = ra[i].x;
            as[(a_kq + 1) * LDA + r] = ra[i].y;
            as[(a_kq + 2) * LDA + r] = ra[i].z;
            as[(a_kq + 3) * LDA + r] = ra[i].w;
        }
        if (BT) {
#pragma unroll
            for (int i = 0; i < BLOADS; ++i) {
                const int n = (t >> 2) + 64 * i;
                if (n < BN) {
                    bs[(a_kq + 0) * LDB + n] = rb[i].x;
                    bs[(a_kq + 1) * LDB + n] = rb[i].y;
                    bs[(a_kq + 2) * LDB + n] = rb[i].z;
                    bs[(a_kq + 3) * LDB + n] = rb[i].w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BLOADS; ++i) {
                const int idx = t + 256 * i;
                if (idx < BKT * BN / 4) {
                    const int kk = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
                    *reinterpret_cast<float4*>(&bs[kk * LDB + nq]) = rb[i];
                }
            }
        }
    };

    floatx16 acc[TM][TN];
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

    const int a_rd = (lane >> 5) * LDA + wm * 32 * TM + (lane & 31);
    const int b_rd = (lane >> 5) * LDB + wn * 32 * TN + (lane & 31);

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) load_tile();
        mma_tile<TM, TN, LDA, LDB>(As[cur], Bs[cur], a_rd, b_rd, acc);
        if (it + 1 < niter) store_tile(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
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

// out[i] = alpha * sum_z slabs[z][i] (+ bias[i % N]) (+ out[i])
__global__ __launch_bounds__(256) void nn_slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                             const float* __restrict__ bias, const float* alpha_p,
                                                             int64_t n, int N, int splitk, int64_t slab,
                                                             int accumulate) {
    const float alpha = alpha_p ? *alpha_p : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // 4 slabs in flight per thread (scalar path: outputs with N % 4 != 0 are rare and small)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 3 < splitk; z += 4) {
            s0 += ws[(int64_t)z * slab + i];
            s1 += ws[(int64_t)(z + 1) * slab + i];
            s2 += ws[(int64_t)(z + 2) * slab + i];
            s3 += ws[(int64_t)(z + 3) * slab + i];
        }
        for (; z < splitk; ++z) s0 += ws[(int64_t)z * slab + i];
        float s = (s0 + s1) + (s2 + s3);
        s *= alpha;
        if (bias) s += bias[i % N];
        if (accumulate) s += out[i];
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel: out[(tap, ca)][cb] = sum_rows A(row, tap)[ca] * Bv(row)[cb]   (taps flattened into M)
// ------------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN, int MODE, bool VEC>
__global__ __launch_bounds__(256, (TM * TN == 4) ? 4 : 1) void tn_kernel(const TNParams p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int ALOADS = (BKT * BM / 4 + 255) / 256;
    constexpr int BLOADS = (BKT * BN / 4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[2][BKT * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT * LDB];

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
    const int niter = max(0, (row_end - row_begin + BKT - 1) / BKT);

    // per-thread A-load columns: flat row mf -> (tap, channel); fixed for the whole K loop
    int a_kh[ALOADS], a_kw[ALOADS], a_c[ALOADS];
#pragma unroll
    for (int i = 0; i < ALOADS; ++i) {
        const int idx = t + 256 * i;
        const int mf = mf0 + (idx % (BM / 4)) * 4;
        int tap = 0, c = mf;
        if (MODE != GATHER_PLAIN) {
            tap = mf / p.Ca;
            c = mf - tap * p.Ca;
        }
        a_kh[i] = tap / g.k;
        a_kw[i] = tap % g.k;
        a_c[i] = (mf < p.Mf) ? c : -1;
    }

    float4 ra[ALOADS], rb[BLOADS];
    int l_row = row_begin;

    // pixel position of each A-load row, advanced incrementally by BKT pixels per K tile (no divisions
    // in the loop: the decode would otherwise cost as many VALU cycles as the tile's MFMAs)
    RowPos apos[ALOADS];
#pragma unroll
    for (int i = 0; i < ALOADS; ++i)
        apos[i] = decompose_row<MODE>(g, row_begin + (t + 256 * i) / (BM / 4), p.M, 0, 0);

    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ALOADS; ++i) {
            const int idx = t + 256 * i;
            const int kr = idx / (BM / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int m = l_row + kr;
            RowPos rp = apos[i];
            rp.valid = m < p.M;
            if (MODE == GATHER_PLAIN) {
                apos[i].b += BKT;
            } else {
                apos[i].wo += BKT;
                while (apos[i].wo >= g.Wq) {
                    apos[i].wo -= g.Wq;
                    if (++apos[i].ho == g.Hq) {
                        apos[i].ho = 0;
                        ++apos[i].b;
                    }
                }
            }
            if (idx < BKT * BM / 4 && m < row_end) {
                if (VEC) {
                    if (a_c[i] >= 0) {
                        int64_t off[1];
                        tap_sources<MODE, false>(g, rp, a_kh[i], a_kw[i], off);
                        v = load_chan4<true>(Abase, off[0], a_c[i], p.Ca);
                    }
                } else {
                    // scalar path (Ca % 4 != 0): every element may belong to a different tap
                    const int mf = mf0 + (idx % (BM / 4)) * 4;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        e[j] = 0.f;
                        const int mj = mf + j;
                        if (mj < p.Mf) {
                            int tap = 0, c = mj;
                            if (MODE != GATHER_PLAIN) {
                                tap = mj / p.Ca;
                                c = mj - tap * p.Ca;
                            }
                            int64_t off[1];
                            tap_sources<MODE, false>(g, rp, tap / g.k, tap % g.k, off);
                            if (off[0] >= 0) e[j] = Abase[off[0] + c];
                        }
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BLOADS; ++i) {
            const int idx = t + 256 * i;
            const int kr = idx / (BN / 4), cq = (idx % (BN / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int m = l_row + kr;
            if (idx < BKT * BN / 4 && m < row_end) v = load_chan4<VEC>(Bbase, (int64_t)m * p.b_ld, cb0 + cq, p.Cb);
            rb[i] = v;
        }
        l_row += BKT;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ALOADS; ++i) {
            const int idx = t + 256 * i;
            if (idx < BKT * BM / 4) {
                const int kr = idx / (BM / 4), cq = (idx % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(&As[buf][kr * LDA + cq]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < BLOADS; ++i) {
            const int idx = t + 256 * i;
            if (idx < BKT * BN / 4) {
                const int kr = idx / (BN / 4), cq = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][kr * LDB + cq]) = rb[i];
            }
        }
    };

    floatx16 acc[TM][TN];
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

    const int a_rd = (lane >> 5) * LDA + wm * 32 * TM + (lane & 31);
    const int b_rd = (lane >> 5) * LDB + wn * 32 * TN + (lane & 31);

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) load_tile();
        mma_tile<TM, TN, LDA, LDB>(As[cur], Bs[cur], a_rd, b_rd, acc);
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

// sum split-K slabs: out[i] = sum_z ws[z*slab + i]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                          int64_t n, int splitk, int64_t slab) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t1 = s, t2 = s, t3 = s;
        int z = 0;
        for (; z + 3 < splitk; z += 4) {                 // 4 slabs in flight per thread
            const float4 a = ld4(ws + (int64_t)z * slab + i * 4), b = ld4(ws + (int64_t)(z + 1) * slab + i * 4);
            const float4 c = ld4(ws + (int64_t)(z + 2) * slab + i * 4), d = ld4(ws + (int64_t)(z + 3) * slab + i * 4);
            add4(s, a);
            add4(t1, b);
            add4(t2, c);
            add4(t3, d);
        }
        for (; z < splitk; ++z) add4(s, ld4(ws + (int64_t)z * slab + i * 4));
        add4(s, t1);
        add4(t2, t3);
        add4(s, t2);
        *reinterpret_cast<float4*>(out + i * 4) = s;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < splitk; ++z) s += ws[(int64_t)z * slab + i];
        out[i] = s;
    }
}

// Few outputs, many slabs (3-channel and narrow layers: a 27 x 64 gradient split 768 ways): one thread per
// output walking all slabs serially is latency-bound (120 us); here 8 z-lanes x 4 loads in flight share
// each float4 column and combine through LDS (fixed order: deterministic).  n % 4 == 0.
__global__ __launch_bounds__(256) void slab_reduce_small_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                int64_t n4, int splitk, int64_t slab) {
    __shared__ float4 part[8][32];
    const int tx = threadIdx.x & 31, tz = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + tx;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (i < n4) {
        const float* p = ws + i * 4;
        int z = tz;
        for (; z + 24 < splitk; z += 32) {
            const float4 a = ld4(p + (int64_t)z * slab), b = ld4(p + (int64_t)(z + 8) * slab);
            const float4 c = ld4(p + (int64_t)(z + 16) * slab), d = ld4(p + (int64_t)(z + 24) * slab);
            add4(s0, a);
            add4(s1, b);
            add4(s2, c);
            add4(s3, d);
        }
        for (; z < splitk; z += 8) add4(s0, ld4(p + (int64_t)z * slab));
        add4(s0, s1);
        add4(s2, s3);
        add4(s0, s2);
    }
    part[tz][tx] = s0;
    __syncthreads();
    if (tz == 0 && i < n4) {
        float4 s = part[0][tx];
#pragma unroll
        for (int l = 1; l < 8; ++l) add4(s, part[l][tx]);
        *reinterpret_cast<float4*>(out + i * 4) = s;
    }
}

void launch_slab_reduce(const float* ws, float* out, int64_t n, int splitk, int64_t slab, hipStream_t s) {
    if ((n & 3) == 0 && (n >> 2) <= 16384 && splitk >= 16) {
        const int64_t n4 = n >> 2;
        hipLaunchKernelGGL(slab_reduce_small_kernel, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, s, ws, out, n4,
                           splitk, slab);
        return;
    }
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, out, n, splitk, slab);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct NNPlan {
    int bm, bn, splitk;
};

// tile: the largest that still fills the 256 CUs; split K when even the smallest cannot
// Stride phases with unequal tap counts (k3 s2: 1, 2, 2, 4 taps): a block of the heaviest phase runs 4x longer
// than one of the lightest, so the grid needs more (smaller) blocks to keep 256 CUs busy until the end.
static thread_local int g_plan_uneven = 0;
struct UnevenPhases {
    explicit UnevenPhases(const BgConvDesc* d) { g_plan_uneven = (d && d->stride > 1 && d->k % d->stride != 0) ? 1 : 0; }
    ~UnevenPhases() { g_plan_uneven = 0; }
};

static NNPlan plan_nn(int64_t M, int N, int zdim, int niter_min, bool allow_split) {
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * zdim; };
    NNPlan pl;
    pl.splitk = 1;
    static const int want_uneven = getenv("BG_WANT_UNEVEN") ? atoi(getenv("BG_WANT_UNEVEN")) : 768;   // measured: 512->1024 8x8 dgrad 50 -> 79 TF/s
    static const int want_even = getenv("BG_WANT") ? atoi(getenv("BG_WANT")) : 768;      // 3 blocks per CU (sweep: profiles/README)
    const int64_t want = g_plan_uneven ? want_uneven : want_even;
    // narrow outputs with a long K and a handful of tiles (low-rank Gram of the cond-BN kernels: 32 x 32 x 1024
    // in ONE block = 64 serial K-steps = 44 us): split K
    auto narrow_split = [&](int bm, int bn) {
        const int64_t b = blocks(bm, bn);
        if (!allow_split || b >= 64 || niter_min < 16) return;
        int sk = (int)((128 + b - 1) / b);
        if (sk > niter_min / 4) sk = niter_min / 4;
        if (sk > 32) sk = 32;
        if (sk > 1) pl.splitk = sk;
    };
    if (N <= 32) {
        pl.bm = 128; pl.bn = 32;
        narrow_split(128, 32);
    } else if (N <= 64) {
        pl.bn = 64;
        pl.bm = blocks(128, 64) >= want ? 128 : 64;
        narrow_split(pl.bm, 64);
    } else if (blocks(128, 128) >= want) {
        pl.bm = 128; pl.bn = 128;
    } else if (blocks(128, 64) >= want) {
        pl.bm = 128; pl.bn = 64;
    } else if (allow_split && niter_min >= 32) {
        // few output tiles but a long K (4x4 / 8x8 feature maps, wide dense layers): keep the
        // efficient 128x128 tile and split K over blockIdx.z instead of shrinking the tile
        pl.bm = 128; pl.bn = 128;
        const int64_t b = blocks(128, 128);
        static const int sk_target = getenv("BG_NN_SPLIT") ? atoi(getenv("BG_NN_SPLIT")) : 768;
        int sk = (int)((sk_target + b - 1) / b);
        if (sk > niter_min / 8) sk = niter_min / 8;
        if (sk > 32) sk = 32;
        if (sk < 1) sk = 1;
        pl.splitk = sk;
    } else {
        pl.bm = 64; pl.bn = 64;
    }
    return pl;
}

template <int TM, int TN, int WM, int WN, bool BT, int MODE, bool MIRROR>
static void launch_nn_inst(const NNParams& p, bool vec, dim3 grid, hipStream_t s) {
    prof_kernel("nn_kernel<%d, %d, %d, %d, %s, %d, %s, %s>", TM, TN, WM, WN, BT ? "true" : "false", MODE,
                MIRROR ? "true" : "false", vec ? "true" : "false");
    if (vec)
        hipLaunchKernelGGL((nn_kernel<TM, TN, WM, WN, BT, MODE, MIRROR, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((nn_kernel<TM, TN, WM, WN, BT, MODE, MIRROR, false>), grid, dim3(256), 0, s, p);
}

template <bool BT, int MODE, bool MIRROR>
static void launch_nn_tile(NNParams& p, const NNPlan& pl, bool vec, bool bf16, int zdim, hipStream_t s) {
    p.tiles_m = (p.M + pl.bm - 1) / pl.bm;
    p.tiles_n = (p.N + pl.bn - 1) / pl.bn;
    dim3 grid(p.tiles_m * p.tiles_n, 1, zdim * p.splitk);
    // stride-phase launches: the phases of one M-tile gather the same input rows, so run them back to back
    // on one XCD (z fastest) instead of one phase after the other over the whole image
    static const int zfold_mode = getenv("BG_ZFOLD") ? atoi(getenv("BG_ZFOLD")) : 1;
    // (measured, round 1: a chunked K order costs more in per-tap index math than it saves in L2 misses:
    //  off by default, kept as a tuning knob)
    static const int kchunk_ch = getenv("BG_KCHUNK") ? atoi(getenv("BG_KCHUNK")) : 0;    // channels per chunk
    const bool use_bf16 = bf16 && vec && pl.bn >= 64;
    p.kchunk = (MODE == GATHER_PLAIN || kchunk_ch <= 0) ? 0 : (kchunk_ch + (use_bf16 ? 31 : 15)) / (use_bf16 ? 32 : 16);
    p.zfold = 0;
    // only when every phase has the same number of taps (k % stride == 0, e.g. k4 s2): unequal phases keep the
    // heavy-phase-first z-major order, which balances the tail of the grid better
    if (zfold_mode && MODE == GATHER_TCONV && zdim > 1 && p.splitk == 1 && p.g.k % p.g.stride == 0) {
        p.zfold = zdim;
        grid = dim3(p.tiles_m * p.tiles_n * zdim, 1, 1);
    }
    if constexpr (!BT) {
        static const int nn_btr = getenv("BG_NN_BTR") ? atoi(getenv("BG_NN_BTR")) : 1;   // weights read through ds_read_b64_tr_b16
        if (nn_btr && bf16 && vec && pl.bm == 128 && pl.bn == 128) {
            prof_kernel("nn_kernel_bf16<2, 2, false, %d, %s, true>", MODE, MIRROR ? "true" : "false");
            hipLaunchKernelGGL((nn_kernel_bf16<2, 2, false, MODE, MIRROR, true>), grid, dim3(256), 0, s, p);
            return;
        }
    }
    if (bf16 && vec && pl.bn >= 64) {
        prof_kernel("nn_kernel_bf16<%d, %d, %s, %d, %s, false>", pl.bm / 64, pl.bn / 64, BT ? "true" : "false", MODE,
                    MIRROR ? "true" : "false");
        if (pl.bm == 128 && pl.bn == 128)
            hipLaunchKernelGGL((nn_kernel_bf16<2, 2, BT, MODE, MIRROR>), grid, dim3(256), 0, s, p);
        else if (pl.bm == 128 && pl.bn == 64)
            hipLaunchKernelGGL((nn_kernel_bf16<2, 1, BT, MODE, MIRROR>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((nn_kernel_bf16<1, 1, BT, MODE, MIRROR>), grid, dim3(256), 0, s, p);
        return;
    }
    if (pl.bm == 128 && pl.bn == 128)
        launch_nn_inst<2, 2, 2, 2, BT, MODE, MIRROR>(p, vec, grid, s);
    else if (pl.bm == 128 && pl.bn == 64)
        launch_nn_inst<2, 1, 2, 2, BT, MODE, MIRROR>(p, vec, grid, s);
    else if (pl.bm == 64 && pl.bn == 64)
        launch_nn_inst<1, 1, 2, 2, BT, MODE, MIRROR>(p, vec, grid, s);
    else
        launch_nn_inst<1, 1, 4, 1, BT, MODE, MIRROR>(p, vec, grid, s);
}

static size_t nn_workspace_bytes(int64_t M, int N, int zdim, int niter_min, int64_t out_elems) {
    NNPlan pl = plan_nn(M, N, zdim, niter_min, true);
    return pl.splitk > 1 ? (size_t)pl.splitk * out_elems * sizeof(float) : 0;
}

// the (MODE, BT) pairs that exist: CONV/BT0, TCONV/BT1 (+MIRROR), PLAIN/BT0, PLAIN/BT1
static int launch_nn(NNParams& p, int mode, bool bt, bool mirror, bool vec, int zdim, int niter_min,
                     int64_t out_elems, bool out_dense, void* ws, size_t ws_bytes, hipStream_t s,
                     bool bf16 = false) {
    NNPlan pl = plan_nn(p.M, p.N, zdim, niter_min, out_dense && ws != nullptr);
    if (pl.splitk > 1 && ws_bytes < (size_t)pl.splitk * out_elems * sizeof(float)) {
        pl = plan_nn(p.M, p.N, zdim, niter_min, false);
    }
    p.splitk = pl.splitk;
    p.slabs = reinterpret_cast<float*>(ws);
    p.slab_stride = out_elems;
    if (mode == GATHER_CONV)
        launch_nn_tile<false, GATHER_CONV, false>(p, pl, vec, bf16, zdim, s);
    else if (mode == GATHER_TCONV && mirror)
        launch_nn_tile<true, GATHER_TCONV, true>(p, pl, vec, bf16, zdim, s);
    else if (mode == GATHER_TCONV)
        launch_nn_tile<true, GATHER_TCONV, false>(p, pl, vec, bf16, zdim, s);
    else if (bt)
        launch_nn_tile<true, GATHER_PLAIN, false>(p, pl, vec, bf16, zdim, s);
    else
        launch_nn_tile<false, GATHER_PLAIN, false>(p, pl, vec, bf16, zdim, s);
    BG_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        int blocks = (int)((out_elems + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(nn_slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, p.out, p.bias, p.alpha,
                           out_elems, p.N, pl.splitk, out_elems, p.accumulate);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

struct TNPlan {
    int bm, bn, splitk, rows_per_split;
};

static TNPlan plan_tn(int Mf, int Cb, int batch, int M) {
    TNPlan pl;
    // available tiles: (128,128) (128,64) (64,64) (128,32)
    if (Cb <= 32) { pl.bm = 128; pl.bn = 32; }
    else if (Cb <= 64) { pl.bn = 64; pl.bm = Mf <= 64 ? 64 : 128; }
    else if (Mf <= 64) { pl.bm = 64; pl.bn = 64; }
    else { pl.bm = 128; pl.bn = 128; }
    const int64_t tiles = (int64_t)((Mf + pl.bm - 1) / pl.bm) * ((Cb + pl.bn - 1) / pl.bn) * batch;
    static const int tn_want = getenv("BG_TN_WANT") ? atoi(getenv("BG_TN_WANT")) : 2048;    // sweep 512..3072: best at 2048 (8 blocks per CU)
    int64_t want = tn_want;
    int sk = (int)((want + tiles - 1) / tiles);
    const int max_sk = (M + 8 * BKT - 1) / (8 * BKT);      // at least 128 rows per split
    if (sk > max_sk) sk = max_sk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    int rps = (M + sk - 1) / sk;
    rps = (rps + BKT - 1) / BKT * BKT;
    sk = (M + rps - 1) / rps;
    pl.splitk = sk;
    pl.rows_per_split = rps;
    return pl;
}

static size_t tn_workspace_bytes(int Mf, int Cb, int batch, int M) {
    TNPlan pl = plan_tn(Mf, Cb, batch, M);
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * Mf * Cb * batch * sizeof(float);
}

template <int TM, int TN, int WM, int WN>
static void launch_tn_inst(const TNParams& p, int mode, bool vec, dim3 grid, hipStream_t s) {
    prof_kernel("tn_kernel<%d, %d, %d, %d, %d, %s>", TM, TN, WM, WN, mode == GATHER_PLAIN ? GATHER_PLAIN : GATHER_CONV,
                vec ? "true" : "false");
    if (mode == GATHER_PLAIN) {
        if (vec)
            hipLaunchKernelGGL((tn_kernel<TM, TN, WM, WN, GATHER_PLAIN, true>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((tn_kernel<TM, TN, WM, WN, GATHER_PLAIN, false>), grid, dim3(256), 0, s, p);
    } else {
        if (vec)
            hipLaunchKernelGGL((tn_kernel<TM, TN, WM, WN, GATHER_CONV, true>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((tn_kernel<TM, TN, WM, WN, GATHER_CONV, false>), grid, dim3(256), 0, s, p);
    }
}

// out must be a dense [batch][Mf][Cb] block when split-K is used
static int launch_tn(TNParams& p, int mode, bool vec, float* final_out, void* ws, size_t ws_bytes, hipStream_t s,
                     bool allow_bf16 = false) {
    const bool bf16 = allow_bf16 && vec;
    TNPlan pl = plan_tn(p.Mf, p.Cb, p.batch, p.M);
    const int64_t total = (int64_t)p.batch * p.Mf * p.Cb;
    const bool dense = (p.out_ld == p.Cb) && (p.batch == 1 || p.strideC == (int64_t)p.Mf * p.Cb);
    if (pl.splitk > 1 && (!dense || ws == nullptr || ws_bytes < (size_t)pl.splitk * total * sizeof(float))) {
        pl.splitk = 1;
        pl.rows_per_split = (p.M + BKT - 1) / BKT * BKT;
    }
    p.splitk = pl.splitk;
    p.rows_per_split = pl.rows_per_split;
    if (pl.splitk > 1) {
        p.out = reinterpret_cast<float*>(ws);
        p.slab_stride = total;
    } else {
        p.out = final_out;
        p.slab_stride = 0;
    }
    p.tiles_m = (p.Mf + pl.bm - 1) / pl.bm;
    p.tiles_n = (p.Cb + pl.bn - 1) / pl.bn;
    dim3 grid(p.tiles_m * p.tiles_n, 1, p.batch * p.splitk);
    // (measured, round 1: +20 % on two bf16 wgrad shapes, -5..-9 % on most fp32 ones: off by default)
    static const int tn_zfold = getenv("BG_TN_ZFOLD") ? atoi(getenv("BG_TN_ZFOLD")) : 0;
    p.zfold = 0;
    if (tn_zfold && p.batch == 1 && p.splitk > 1) {
        p.zfold = p.splitk;
        grid = dim3(p.tiles_m * p.tiles_n * p.splitk, 1, 1);
    }
    static const int tn_tr = getenv("BG_TN_TR") ? atoi(getenv("BG_TN_TR")) : 1;   // hardware-transposed operand reads (0: VALU transposition)
    if (tn_tr && bf16 && pl.bm == 128 && pl.bn == 128 && p.Ca % 8 == 0 && p.Cb % 8 == 0) {
        prof_kernel("tn_kernel_bf16_tr<%d>", mode == GATHER_CONV ? GATHER_CONV : GATHER_PLAIN);
        if (mode == GATHER_CONV)
            hipLaunchKernelGGL((tn_kernel_bf16_tr<GATHER_CONV>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((tn_kernel_bf16_tr<GATHER_PLAIN>), grid, dim3(256), 0, s, p);
        BG_LAUNCH_CHECK();
        goto tn_reduce;
    }
    if (bf16 && pl.bn >= 64) prof_kernel("tn_kernel_bf16<%d, %d, %d>", pl.bm / 64, pl.bn / 64, mode);
    if (bf16 && pl.bn >= 64 && mode == GATHER_CONV) {
        if (pl.bm == 128 && pl.bn == 128)
            hipLaunchKernelGGL((tn_kernel_bf16<2, 2, GATHER_CONV>), grid, dim3(256), 0, s, p);
        else if (pl.bm == 128 && pl.bn == 64)
            hipLaunchKernelGGL((tn_kernel_bf16<2, 1, GATHER_CONV>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((tn_kernel_bf16<1, 1, GATHER_CONV>), grid, dim3(256), 0, s, p);
    } else if (bf16 && pl.bn >= 64) {
        if (pl.bm == 128 && pl.bn == 128)
            hipLaunchKernelGGL((tn_kernel_bf16<2, 2, GATHER_PLAIN>), grid, dim3(256), 0, s, p);
        else if (pl.bm == 128 && pl.bn == 64)
            hipLaunchKernelGGL((tn_kernel_bf16<2, 1, GATHER_PLAIN>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((tn_kernel_bf16<1, 1, GATHER_PLAIN>), grid, dim3(256), 0, s, p);
    } else if (pl.bm == 128 && pl.bn == 128)
        launch_tn_inst<2, 2, 2, 2>(p, mode, vec, grid, s);
    else if (pl.bm == 128 && pl.bn == 64)
        launch_tn_inst<2, 1, 2, 2>(p, mode, vec, grid, s);
    else if (pl.bm == 64 && pl.bn == 64)
        launch_tn_inst<1, 1, 2, 2>(p, mode, vec, grid, s);
    else
        launch_tn_inst<1, 1, 4, 1>(p, mode, vec, grid, s);     // 128 x 32
    BG_LAUNCH_CHECK();
tn_reduce:
    if (pl.splitk > 1) {
        launch_slab_reduce(reinterpret_cast<const float*>(ws), final_out, total, pl.splitk, total, s);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

static int check_desc(const BgConvDesc* d) {
    BG_REQUIRE(d != nullptr, "null BgConvDesc");
    BG_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0,
               "BgConvDesc: non-positive dimension");
    BG_REQUIRE(d->k >= 1 && d->k <= 7 && d->stride >= 1 && d->stride <= 2, "BgConvDesc: k=%d stride=%d unsupported",
               d->k, d->stride);
    BG_REQUIRE(d->pad_mode == BG_PAD_REFLECT || d->pad_mode == BG_PAD_ZERO, "BgConvDesc: bad pad_mode");
    BG_REQUIRE(d->pad_lo >= 0 && d->pad_lo < d->k, "BgConvDesc: bad pad_lo");
    BG_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31),
               "BgConvDesc: more than 2^31 pixels");
    return BG_OK;
}

static int check_conv(const BgConvDesc* d) {
    int rc = check_desc(d);
    if (rc) return rc;
    BG_REQUIRE(d->pad_mode == BG_PAD_ZERO || d->pad_lo < d->H, "reflect padding wider than the image");
    BG_REQUIRE((d->Ho - 1) * d->stride - d->pad_lo < d->H && (d->Wo - 1) * d->stride - d->pad_lo < d->W,
               "conv output extent inconsistent with input");
    if (d->pad_mode == BG_PAD_REFLECT) {
        const int hi_h = (d->Ho - 1) * d->stride + d->k - 1 - d->pad_lo;
        const int hi_w = (d->Wo - 1) * d->stride + d->k - 1 - d->pad_lo;
        BG_REQUIRE(hi_h <= 2 * (d->H - 1) && hi_w <= 2 * (d->W - 1), "reflect padding out of range");
    }
    return BG_OK;
}

static int check_deconv(const BgConvDesc* d) {
    int rc = check_desc(d);
    if (rc) return rc;
    BG_REQUIRE(d->Ho == d->H * d->stride && d->Wo == d->W * d->stride, "deconv: Ho must equal stride*H (SAME)");
    BG_REQUIRE(d->pad_mode == BG_PAD_ZERO, "deconv: zero padding only");
    return BG_OK;
}

struct Tag {
    char s[112];
    double bytes = 0.0;     // algorithmic HBM bytes of the call: both activation tensors once in their stored type, the
                            // weights once (2 bytes packed / 4 bytes fp32) or, for a weight gradient, written once in fp32
    Tag(const char* op, const BgConvDesc* d) {
        // a 1 x 1 "convolution" over an [H, 1] column of ONE image is a plain matrix product through this entry point (the
        // regulariser's W (dA + dA^T) from the packed weights): its own class, so that the measurement does not count it
        // among the convolutions whose FLOPs SURVEY 8(d) defines
        if (d->N == 1 && d->W == 1 && d->Wo == 1 && d->k == 1)
            snprintf(s, sizeof(s), "gemm16 M%d N%d K%d (%s)", d->H, d->Cout, d->Cin, op);
        else
            snprintf(s, sizeof(s), "%s N%d H%d Cin%d Cout%d Ho%d k%d s%d", op, d->N, d->H, d->Cin, d->Cout, d->Ho, d->k,
                     d->stride);
        const bool wgrad = strstr(op, "wgrad") != nullptr;
        const double sx = d->x_dtype == BG_BF16 ? 2.0 : 4.0, sy = d->y_dtype == BG_BF16 ? 2.0 : 4.0;
        const double sw = wgrad ? 4.0 : (d->w_packed ? 2.0 : 4.0);
        bytes = (double)d->N * d->H * d->W * d->Cin * sx + (double)d->N * d->Ho * d->Wo * d->Cout * sy +
                (double)d->k * d->k * d->Cin * d->Cout * sw;
    }
    Tag(const char* op, int m, int n, int k) { snprintf(s, sizeof(s), "%s M%d N%d K%d", op, m, n, k); }
    Tag(const char* op, const BgGemmDesc* d) {
        snprintf(s, sizeof(s), "%s M%d N%d K%d tA%d tB%d b%d", op, d->M, d->N, d->K, d->transA, d->transB, d->batch);
    }
};

static double conv_flops(const BgConvDesc* d, bool deconv) {
    const double px = deconv ? (double)d->N * d->H * d->W : (double)d->N * d->Ho * d->Wo;
    return 2.0 * px * d->k * d->k * d->Cin * d->Cout;
}

static inline int kc_of(int C) { return (C + BKT - 1) / BKT; }

// ---- parameter builders shared by the entry points and their workspace queries ----
static void conv_fwd_params(const BgConvDesc* d, NNParams& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->H; g.Ws = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.Hq = d->Ho; g.Wq = d->Wo; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = d->pad_mode == BG_PAD_REFLECT; g.ld = d->Cin;
    p.C = d->Cin; p.M = d->N * d->Ho * d->Wo; p.N = d->Cout;
    p.tap_stride = (int64_t)d->Cin * d->Cout; p.ldk = d->Cout; p.ldn = 1;
    p.out_ld = d->Cout; p.batch = 1;
}

static void conv_dgrad_params(const BgConvDesc* d, NNParams& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->Ho; g.Ws = d->Wo; g.Ho = d->H; g.Wo = d->W;
    g.pstep = d->stride; g.Hq = d->H / d->stride; g.Wq = d->W / d->stride;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo;
    g.reflect = (d->pad_mode == BG_PAD_REFLECT) && d->pad_lo > 0; g.ld = d->Cout;
    p.C = d->Cout; p.M = d->N * g.Hq * g.Wq; p.N = d->Cin;
    p.tap_stride = (int64_t)d->Cin * d->Cout; p.ldk = 1; p.ldn = d->Cout;   // B[c=co][n=ci] = w[tap][ci][co]
    p.out_ld = d->Cin; p.batch = 1;
}

static void deconv_fwd_params(const BgConvDesc* d, NNParams& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->H; g.Ws = d->W; g.Ho = d->Ho; g.Wo = d->Wo;
    g.pstep = d->stride; g.Hq = d->Ho / d->stride; g.Wq = d->Wo / d->stride;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = 0; g.ld = d->Cin;
    p.C = d->Cin; p.M = d->N * g.Hq * g.Wq; p.N = d->Cout;
    p.tap_stride = (int64_t)d->Cin * d->Cout; p.ldk = 1; p.ldn = d->Cin;    // B[c=ci][n=co] = w[tap][co][ci]
    p.out_ld = d->Cout; p.batch = 1;
}

static void deconv_dgrad_params(const BgConvDesc* d, NNParams& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->Ho; g.Ws = d->Wo; g.Ho = d->H; g.Wo = d->W; g.Hq = d->H; g.Wq = d->W; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = 0; g.ld = d->Cout;
    p.C = d->Cout; p.M = d->N * d->H * d->W; p.N = d->Cin;
    p.tap_stride = (int64_t)d->Cin * d->Cout; p.ldk = d->Cin; p.ldn = 1;    // B[c=co][n=ci] = w[tap][co][ci]
    p.out_ld = d->Cin; p.batch = 1;
}

// fewest K iterations over the phases of a transposed gather (k3 s2 has 1x1 .. 2x2 taps per phase)
static int tconv_min_iters(const BgConvDesc* d, int C) {
    int per_axis = d->stride == 1 ? d->k : d->k / d->stride;   // smallest tap count of a phase
    if (per_axis < 1) per_axis = 1;
    return per_axis * per_axis * kc_of(C);
}

// ---- bf16-resident path (igemm16.hip): parameter builders -------------------------------------------
static bool resident_fwd(const BgConvDesc* d) { return d->x_dtype == BG_BF16; }
static bool resident_dgrad(const BgConvDesc* d) { return d->y_dtype == BG_BF16 && d->w_packed; }
static bool resident_wgrad(const BgConvDesc* d) { return d->x_dtype == BG_BF16 && d->y_dtype == BG_BF16; }

static void nn16_from(const NNParams& q, NN16Params& p) {
    memset(&p, 0, sizeof(p));
    p.g = q.g;
    p.C = q.C; p.M = q.M; p.N = q.N;
    p.tap_stride = (int64_t)q.N * q.C;
    p.out_ld = q.out_ld;
}

// extent of the padded grid the gradient of a reflect-padded conv is computed on (a multiple of the stride)
static int padded_extent(int out, int k, int stride) {
    const int used = (out - 1) * stride + k;
    return (used + stride - 1) / stride * stride;
}

static void conv16_dgrad_params(const BgConvDesc* d, bool padded, NN16Params& p) {
    NNParams q;
    conv_dgrad_params(d, q);
    nn16_from(q, p);
    if (padded) {
        Gather& g = p.g;
        g.Ho = padded_extent(d->Ho, d->k, d->stride);
        g.Wo = padded_extent(d->Wo, d->k, d->stride);
        g.Hq = g.Ho / d->stride;
        g.Wq = g.Wo / d->stride;
        g.pad = 0;
        g.reflect = 0;
        p.M = d->N * g.Hq * g.Wq;
    }
}

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// Input gradient of a reflect-padded 3 x 3 convolution WITHOUT the padded grid (ops.py:81-82, 94 under autodiff): the
// plain transposed gather on the H x W map (zero-padding semantics: halo-tile kernel on 16 / 32 / 64 maps, position-major
// tap kernel on 4 x 4 / 8 x 8) plus two thin launches that add the taps of the mirrored padded rows / columns into
// rows / columns 1 and H - 2 (NN16Params::ring).  BG_DGRAD_RING=0: the padded grid + reflect_fold_kernel of round 2 (A/B).
// When it pays (measured r03 on config 3, per call, padded-grid form -> this form): at N = 512 the 64 x 64 / 32 x 32 /
// 16 x 16 / 8 x 8 / 4 x 4 maps gain 0.32 / 0.24 / 0.09 / 0.07 / 0.12 ms, in proportion to N, while the mirrored-tap
// launch costs a fixed 20 - 50 us (a few tiles walking up to 7 taps x C / 64 K steps).  Below the break-even batch the
// padded grid stays (BG_DGRAD_RING=1 forces this form, =0 the padded grid).
static bool dgrad_ring_ok(const BgConvDesc* d) {
    const char* e = getenv("BG_DGRAD_RING");
    if (e && atoi(e) == 0) return false;
    const bool forced = e && atoi(e) == 1;
    const int need = d->H >= 32 ? 32 : (d->H >= 16 ? 96 : 160);
    if (!forced && d->N < need) return false;
    return d->pad_mode == BG_PAD_REFLECT && d->pad_lo == 1 && d->k == 3 && (d->stride == 1 || d->stride == 2) &&
           d->H >= 4 && d->W >= 4 && d->H % d->stride == 0 && d->W % d->stride == 0 && d->Ho == d->H / d->stride &&
           d->Wo == d->W / d->stride;
}

}  // namespace bg

using namespace bg;

extern "C" {

size_t bg_conv2d_fwd_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    if (resident_fwd(d)) {
        NNParams q;
        conv_fwd_params(d, q);
        NN16Params p;
        nn16_from(q, p);
        return nn16_workspace_bytes(p, GATHER_CONV, 1, (int64_t)p.M * d->Cout);
    }
    return nn_workspace_bytes((int64_t)d->N * d->Ho * d->Wo, d->Cout, 1, d->k * d->k * kc_of(d->Cin),
                              (int64_t)d->N * d->Ho * d->Wo * d->Cout);
}

int bg_conv2d_fwd(const BgConvDesc* d, const void* x, const void* w, const float* bias, const float* alpha_dev,
                  void* y, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_conv(d);
    if (rc) return rc;
    BG_REQUIRE(x && w && y, "bg_conv2d_fwd: null tensor pointer");
    Tag tag("conv2d_fwd", d);
    NNParams p;
    conv_fwd_params(d, p);
    if (resident_fwd(d)) {
        BG_REQUIRE(d->w_packed, "bg_conv2d_fwd: bf16 input needs the packed bf16 weights (w_packed)");
        NN16Params r;
        nn16_from(p, r);
        r.A = x; r.B = w; r.bias = bias; r.alpha = alpha_dev; r.out = y; r.accumulate = accumulate;
        r.out_f32 = d->y_dtype == BG_F32;
        ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
        return launch_nn16(r, GATHER_CONV, 1, (int64_t)r.M * d->Cout, ws, ws_bytes, as_stream(stream));
    }
    BG_REQUIRE(d->y_dtype == BG_F32 && !d->w_packed, "bg_conv2d_fwd: fp32 input needs fp32 weights and output");
    p.A = (const float*)x; p.B = (const float*)w; p.bias = bias; p.alpha = alpha_dev; p.out = (float*)y;
    p.accumulate = accumulate;
    const bool vec = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(w);
    ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
    return launch_nn(p, GATHER_CONV, false, false, vec, 1, d->k * d->k * kc_of(d->Cin),
                     (int64_t)p.M * d->Cout, true, ws, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
}

size_t bg_conv2d_dgrad_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    const int z = d->stride * d->stride;
    if (resident_dgrad(d)) {
        const bool padded = d->pad_mode == BG_PAD_REFLECT && d->pad_lo > 0 && !dgrad_ring_ok(d);
        NN16Params p;
        conv16_dgrad_params(d, padded, p);
        p.g.reflect = 0;
        const int64_t out_elems = (int64_t)d->N * p.g.Ho * p.g.Wo * d->Cin;
        size_t b = nn16_workspace_bytes(p, GATHER_TCONV, z, out_elems);
        if (padded) b = align256(b) + align256((size_t)out_elems * (d->x_dtype == BG_F32 ? 4 : 2));
        return b;
    }
    UnevenPhases uneven(d);
    return nn_workspace_bytes((int64_t)d->N * (d->H / d->stride) * (d->W / d->stride), d->Cin, z,
                              tconv_min_iters(d, d->Cout), (int64_t)d->N * d->H * d->W * d->Cin);
}

int bg_conv2d_dgrad(const BgConvDesc* d, const void* dy, const void* w, const float* alpha_dev, void* dx,
                    int accumulate, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_conv(d);
    if (rc) return rc;
    BG_REQUIRE(dy && w && dx, "bg_conv2d_dgrad: null tensor pointer");
    BG_REQUIRE(d->H % d->stride == 0 && d->W % d->stride == 0, "conv dgrad: H,W must be multiples of stride");
    Tag tag("conv2d_dgrad", d);
    if (resident_dgrad(d)) {
        const bool ring = d->pad_mode == BG_PAD_REFLECT && d->pad_lo > 0 && dgrad_ring_ok(d);
        const bool padded = d->pad_mode == BG_PAD_REFLECT && d->pad_lo > 0 && !ring;
        NN16Params r;
        conv16_dgrad_params(d, padded, r);
        r.g.reflect = 0;                        // (the mirrored taps are the ring launches' / the fold's business)
        r.A = dy; r.B = w; r.alpha = alpha_dev;
        r.out_f32 = d->x_dtype == BG_F32;
        const int64_t out_elems = (int64_t)d->N * r.g.Ho * r.g.Wo * d->Cin;
        ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
        if (!padded) {
            r.out = dx; r.accumulate = accumulate;
            const NN16Params plain = r;
            if (nn16h_d2s_ok(plain))            // 8-channel image layers, stride 2: one 2 x 2-window launch (igemm16.hip)
                rc = launch_nn16h_d2s(plain, as_stream(stream));
            else
                rc = launch_nn16(r, GATHER_TCONV, d->stride * d->stride, out_elems, ws, ws_bytes, as_stream(stream));
            if (rc || !ring) return rc;
            NN16Params q = plain;
            q.ring = 1;
            q.ring_lines = d->stride == 1 ? 2 : 1;
            q.accumulate = 1;
            return launch_nn16_ring(q, as_stream(stream));
        }
        // gradient on the reflect-padded grid, then every padded position is added to the pixel it mirrors
        const size_t pbytes = align256((size_t)out_elems * (r.out_f32 ? 4 : 2));
        BG_REQUIRE(ws && ws_bytes >= pbytes, "bg_conv2d_dgrad: workspace too small for the padded gradient");
        r.out = ws; r.accumulate = 0;
        char* slabs = reinterpret_cast<char*>(ws) + pbytes;
        rc = launch_nn16(r, GATHER_TCONV, d->stride * d->stride, out_elems, ws_bytes > pbytes ? slabs : nullptr,
                         ws_bytes - pbytes, as_stream(stream));
        if (rc) return rc;
        return launch_reflect_fold(ws, dx, r.out_f32, d->N, d->H, d->W, d->Cin, r.g.Ho, r.g.Wo, d->pad_lo, accumulate,
                                   as_stream(stream));
    }
    BG_REQUIRE(d->x_dtype == BG_F32 && d->y_dtype == BG_F32 && !d->w_packed, "bg_conv2d_dgrad: unsupported dtype mix");
    UnevenPhases uneven(d);
    NNParams p;
    conv_dgrad_params(d, p);
    p.A = (const float*)dy; p.B = (const float*)w; p.alpha = alpha_dev; p.out = (float*)dx; p.accumulate = accumulate;
    const bool vec = (d->Cout % 4 == 0) && aligned16(dy) && aligned16(w);
    ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
    return launch_nn(p, GATHER_TCONV, true, p.g.reflect != 0, vec, d->stride * d->stride, tconv_min_iters(d, d->Cout),
                     (int64_t)d->N * d->H * d->W * d->Cin, true, ws, ws_bytes, as_stream(stream),
                     d->compute == BG_COMPUTE_BF16);
}

static void conv16_wgrad_params(const BgConvDesc* d, TN16Params& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->H; g.Ws = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.Hq = d->Ho; g.Wq = d->Wo; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = d->pad_mode == BG_PAD_REFLECT; g.ld = d->Cin;
    p.Ca = d->Cin; p.Cb = d->Cout; p.Mf = d->k * d->k * d->Cin; p.b_ld = d->Cout; p.M = d->N * d->Ho * d->Wo;
    p.out_ld = d->Cout;
}

size_t bg_conv2d_wgrad_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    if (resident_wgrad(d)) {
        TN16Params p;
        conv16_wgrad_params(d, p);
        return tn16_workspace_bytes(p);
    }
    return tn_workspace_bytes(d->k * d->k * d->Cin, d->Cout, 1, d->N * d->Ho * d->Wo);
}

int bg_conv2d_wgrad(const BgConvDesc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes,
                    void* stream) {
    int rc = check_conv(d);
    if (rc) return rc;
    BG_REQUIRE(x && dy && dw, "bg_conv2d_wgrad: null tensor pointer");
    Tag tag("conv2d_wgrad", d);
    if (resident_wgrad(d)) {
        TN16Params r;
        conv16_wgrad_params(d, r);
        r.A = x; r.Bv = dy;
        ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
        return launch_tn16(r, GATHER_CONV, dw, ws, ws_bytes, as_stream(stream));
    }
    BG_REQUIRE(d->x_dtype == BG_F32 && d->y_dtype == BG_F32, "bg_conv2d_wgrad: x and dy must both be fp32 or both bf16");
    TNParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const float*)x; p.Bv = (const float*)dy;
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->H; g.Ws = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.Hq = d->Ho; g.Wq = d->Wo; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = d->pad_mode == BG_PAD_REFLECT; g.ld = d->Cin;
    p.Ca = d->Cin; p.Cb = d->Cout; p.Mf = d->k * d->k * d->Cin; p.b_ld = d->Cout; p.M = d->N * d->Ho * d->Wo;
    p.out_ld = d->Cout; p.batch = 1;
    const bool vec = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(dy);
    ProfScope prof(as_stream(stream), conv_flops(d, false), tag.s, tag.bytes);
    return launch_tn(p, GATHER_CONV, vec, dw, ws, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
}

size_t bg_deconv2d_fwd_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    const int z = d->stride * d->stride;
    if (resident_fwd(d)) {
        NNParams q;
        deconv_fwd_params(d, q);
        NN16Params p;
        nn16_from(q, p);
        return nn16_workspace_bytes(p, GATHER_TCONV, z, (int64_t)d->N * d->Ho * d->Wo * d->Cout);
    }
    return nn_workspace_bytes((int64_t)d->N * d->H * d->W, d->Cout, z, tconv_min_iters(d, d->Cin),
                              (int64_t)d->N * d->Ho * d->Wo * d->Cout);
}

int bg_deconv2d_fwd(const BgConvDesc* d, const void* x, const void* w, const float* bias, const float* alpha_dev,
                    void* y, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_deconv(d);
    if (rc) return rc;
    BG_REQUIRE(x && w && y, "bg_deconv2d_fwd: null tensor pointer");
    Tag tag("deconv2d_fwd", d);
    NNParams p;
    deconv_fwd_params(d, p);
    if (resident_fwd(d)) {
        BG_REQUIRE(d->w_packed, "bg_deconv2d_fwd: bf16 input needs the packed bf16 weights (w_packed)");
        NN16Params r;
        nn16_from(p, r);
        r.A = x; r.B = w; r.bias = bias; r.alpha = alpha_dev; r.out = y; r.accumulate = accumulate;
        r.out_f32 = d->y_dtype == BG_F32;
        ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
        return launch_nn16(r, GATHER_TCONV, d->stride * d->stride, (int64_t)d->N * d->Ho * d->Wo * d->Cout, ws, ws_bytes,
                           as_stream(stream));
    }
    BG_REQUIRE(d->y_dtype == BG_F32 && !d->w_packed, "bg_deconv2d_fwd: fp32 input needs fp32 weights and output");
    p.A = (const float*)x; p.B = (const float*)w; p.bias = bias; p.alpha = alpha_dev; p.out = (float*)y;
    p.accumulate = accumulate;
    const bool vec = (d->Cin % 4 == 0) && aligned16(x) && aligned16(w);
    ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
    return launch_nn(p, GATHER_TCONV, true, false, vec, d->stride * d->stride, tconv_min_iters(d, d->Cin),
                     (int64_t)d->N * d->Ho * d->Wo * d->Cout, true, ws, ws_bytes, as_stream(stream),
                     d->compute == BG_COMPUTE_BF16);
}

size_t bg_deconv2d_fwd_stats_workspace_bytes(const BgConvDesc* d) {
    if (!d || !resident_fwd(d) || d->y_dtype != BG_BF16) return 0;
    NNParams q;
    deconv_fwd_params(d, q);
    NN16Params p;
    nn16_from(q, p);
    return (size_t)nn16_stats_rows(p, GATHER_TCONV, d->stride * d->stride) * 2 * (size_t)d->Cout * sizeof(float);
}

int bg_deconv2d_fwd_stats(const BgConvDesc* d, const void* x, const void* w, const float* bias, const float* alpha_dev,
                          void* y, int accumulate, double* sums, void* stats_ws, size_t stats_ws_bytes, void* ws,
                          size_t ws_bytes, void* stream) {
    int rc = check_deconv(d);
    if (rc) return rc;
    BG_REQUIRE(x && w && y && sums && stats_ws, "bg_deconv2d_fwd_stats: null tensor pointer");
    const size_t need = bg_deconv2d_fwd_stats_workspace_bytes(d);
    BG_REQUIRE(need > 0 && d->w_packed, "bg_deconv2d_fwd_stats: this launch has no fused statistics (query the workspace)");
    BG_REQUIRE(stats_ws_bytes >= need && (reinterpret_cast<uintptr_t>(stats_ws) & 15) == 0,
               "bg_deconv2d_fwd_stats: statistics workspace too small");
    Tag tag("deconv2d_fwd", d);
    NNParams p;
    deconv_fwd_params(d, p);
    NN16Params r;
    nn16_from(p, r);
    r.A = x; r.B = w; r.bias = bias; r.alpha = alpha_dev; r.out = y; r.accumulate = accumulate;
    r.out_f32 = 0;
    r.stats_part = reinterpret_cast<float*>(stats_ws);
    {
        ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
        rc = launch_nn16(r, GATHER_TCONV, d->stride * d->stride, (int64_t)d->N * d->Ho * d->Wo * d->Cout, ws, ws_bytes,
                         as_stream(stream));
        if (rc) return rc;
    }
    return launch_partial_colsum(reinterpret_cast<const float*>(stats_ws), sums, (int64_t)(need / (2 * (size_t)d->Cout * sizeof(float))),
                                 2 * d->Cout, as_stream(stream));
}

size_t bg_deconv2d_dgrad_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    if (resident_dgrad(d)) {
        NNParams q;
        deconv_dgrad_params(d, q);
        NN16Params p;
        nn16_from(q, p);
        return nn16_workspace_bytes(p, GATHER_CONV, 1, (int64_t)d->N * d->H * d->W * d->Cin);
    }
    return nn_workspace_bytes((int64_t)d->N * d->H * d->W, d->Cin, 1, d->k * d->k * kc_of(d->Cout),
                              (int64_t)d->N * d->H * d->W * d->Cin);
}

int bg_deconv2d_dgrad(const BgConvDesc* d, const void* dy, const void* w, const float* alpha_dev, void* dx,
                      int accumulate, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_deconv(d);
    if (rc) return rc;
    BG_REQUIRE(dy && w && dx, "bg_deconv2d_dgrad: null tensor pointer");
    Tag tag("deconv2d_dgrad", d);
    NNParams p;
    deconv_dgrad_params(d, p);
    if (resident_dgrad(d)) {
        NN16Params r;
        nn16_from(p, r);
        r.A = dy; r.B = w; r.alpha = alpha_dev; r.out = dx; r.accumulate = accumulate;
        r.out_f32 = d->x_dtype == BG_F32;
        ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
        return launch_nn16(r, GATHER_CONV, 1, (int64_t)r.M * d->Cin, ws, ws_bytes, as_stream(stream));
    }
    BG_REQUIRE(d->x_dtype == BG_F32 && d->y_dtype == BG_F32 && !d->w_packed, "bg_deconv2d_dgrad: unsupported dtype mix");
    p.A = (const float*)dy; p.B = (const float*)w; p.alpha = alpha_dev; p.out = (float*)dx; p.accumulate = accumulate;
    const bool vec = (d->Cout % 4 == 0) && (d->Cin % 4 == 0) && aligned16(dy) && aligned16(w);
    ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
    return launch_nn(p, GATHER_CONV, false, false, vec, 1, d->k * d->k * kc_of(d->Cout),
                     (int64_t)p.M * d->Cin, true, ws, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
}

static void deconv16_wgrad_params(const BgConvDesc* d, TN16Params& p) {
    memset(&p, 0, sizeof(p));
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->Ho; g.Ws = d->Wo; g.Ho = d->H; g.Wo = d->W; g.Hq = d->H; g.Wq = d->W; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = 0; g.ld = d->Cout;
    p.Ca = d->Cout; p.Cb = d->Cin; p.Mf = d->k * d->k * d->Cout; p.b_ld = d->Cin; p.M = d->N * d->H * d->W;
    p.out_ld = d->Cin;
}

size_t bg_deconv2d_wgrad_workspace_bytes(const BgConvDesc* d) {
    if (!d) return 0;
    if (resident_wgrad(d)) {
        TN16Params p;
        deconv16_wgrad_params(d, p);
        return tn16_workspace_bytes(p);
    }
    return tn_workspace_bytes(d->k * d->k * d->Cout, d->Cin, 1, d->N * d->H * d->W);
}

int bg_deconv2d_wgrad(const BgConvDesc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes,
                      void* stream) {
    int rc = check_deconv(d);
    if (rc) return rc;
    BG_REQUIRE(x && dy && dw, "bg_deconv2d_wgrad: null tensor pointer");
    Tag tag("deconv2d_wgrad", d);
    // dw[kh,kw,co,ci] = sum_{b,hi,wi} dy[b, hi*s+kh-pad, wi*s+kw-pad, co] * x[b,hi,wi,ci]
    if (resident_wgrad(d)) {
        TN16Params r;
        deconv16_wgrad_params(d, r);
        r.A = dy; r.Bv = x;
        ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
        return launch_tn16(r, GATHER_CONV, dw, ws, ws_bytes, as_stream(stream));
    }
    BG_REQUIRE(d->x_dtype == BG_F32 && d->y_dtype == BG_F32, "bg_deconv2d_wgrad: x and dy must both be fp32 or both bf16");
    TNParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const float*)dy; p.Bv = (const float*)x;
    Gather& g = p.g;
    g.Nb = d->N; g.Hs = d->Ho; g.Ws = d->Wo; g.Ho = d->H; g.Wo = d->W; g.Hq = d->H; g.Wq = d->W; g.pstep = 1;
    g.k = d->k; g.stride = d->stride; g.pad = d->pad_lo; g.reflect = 0; g.ld = d->Cout;
    p.Ca = d->Cout; p.Cb = d->Cin; p.Mf = d->k * d->k * d->Cout; p.b_ld = d->Cin; p.M = d->N * d->H * d->W;
    p.out_ld = d->Cin; p.batch = 1;
    const bool vec = (d->Cout % 4 == 0) && (d->Cin % 4 == 0) && aligned16(dy) && aligned16(x);
    ProfScope prof(as_stream(stream), conv_flops(d, true), tag.s, tag.bytes);
    return launch_tn(p, GATHER_CONV, vec, dw, ws, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
}

size_t bg_gemm_workspace_bytes(const BgGemmDesc* d) {
    if (!d) return 0;
    const int batch = d->batch > 0 ? d->batch : 1;
    if (d->transA) return tn_workspace_bytes(d->M, d->N, batch, d->K);
    if (d->ldc != d->N || batch != 1) return 0;
    return nn_workspace_bytes(d->M, d->N, 1, kc_of(d->K), (int64_t)d->M * d->N);
}

int bg_gemm(const BgGemmDesc* d, const float* A, const float* B, const float* bias, const float* alpha_dev,
            float* C, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    BG_REQUIRE(d != nullptr, "null BgGemmDesc");
    BG_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1, "BgGemmDesc: non-positive dimension");
    BG_REQUIRE(A && B && C, "bg_gemm: null tensor pointer");
    const double flops = 2.0 * d->M * d->N * d->K * d->batch;
    Tag tag("gemm", d);
    if (!d->transA) {
        NNParams p;
        memset(&p, 0, sizeof(p));
        p.A = A; p.B = B; p.bias = bias; p.alpha = alpha_dev; p.out = C;
        Gather& g = p.g;
        g.Nb = d->M; g.Hs = 1; g.Ws = 1; g.Ho = 1; g.Wo = 1; g.Hq = 1; g.Wq = 1; g.pstep = 1;
        g.k = 1; g.stride = 1; g.pad = 0; g.reflect = 0; g.ld = d->lda;
        p.C = d->K; p.M = d->M; p.N = d->N; p.tap_stride = 0;
        if (d->transB) { p.ldk = 1; p.ldn = d->ldb; } else { p.ldk = d->ldb; p.ldn = 1; }
        p.out_ld = d->ldc; p.accumulate = accumulate; p.batch = d->batch;
        p.strideA = d->strideA; p.strideB = d->strideB; p.strideC = d->strideC;
        bool vec = (d->K % 4 == 0) && (d->lda % 4 == 0) && (d->strideA % 4 == 0) && aligned16(A) &&
                   (d->ldb % 4 == 0) && (d->strideB % 4 == 0) && aligned16(B);
        if (!d->transB) vec = vec && (d->N % 4 == 0);
        ProfScope prof(as_stream(stream), flops, tag.s);
        const bool dense = d->ldc == d->N && d->batch == 1;
        return launch_nn(p, GATHER_PLAIN, d->transB != 0, false, vec, d->batch, kc_of(d->K), (int64_t)d->M * d->N,
                         dense, dense ? ws : nullptr, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
    }
    BG_REQUIRE(!d->transB, "bg_gemm: transA && transB unsupported");
    BG_REQUIRE(bias == nullptr && !accumulate, "bg_gemm: transA path has no bias / accumulate");
    TNParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.Bv = B; p.alpha = alpha_dev;
    Gather& g = p.g;
    g.Nb = d->K; g.Hs = 1; g.Ws = 1; g.Ho = 1; g.Wo = 1; g.Hq = 1; g.Wq = 1; g.pstep = 1;
    g.k = 1; g.stride = 1; g.pad = 0; g.reflect = 0; g.ld = d->lda;
    p.Ca = d->M; p.Cb = d->N; p.Mf = d->M; p.b_ld = d->ldb; p.M = d->K;
    p.out_ld = d->ldc; p.batch = d->batch;
    p.strideA = d->strideA; p.strideB = d->strideB; p.strideC = d->strideC;
    const bool vec = (d->M % 4 == 0) && (d->lda % 4 == 0) && (d->strideA % 4 == 0) && aligned16(A) &&
                     (d->N % 4 == 0) && (d->ldb % 4 == 0) && (d->strideB % 4 == 0) && aligned16(B);
    ProfScope prof(as_stream(stream), flops, tag.s);
    return launch_tn(p, GATHER_PLAIN, vec, C, ws, ws_bytes, as_stream(stream), d->compute == BG_COMPUTE_BF16);
}

// Gram matrix of a bf16 row-major matrix: out[c1][c2] = sum_r a[r][c1] a[r][c2] (fp32), the ortho-cosine regulariser's
// W^T W (utils.py:198) taken from the packed bf16 copy of w / sigma that the spectral-norm pass already wrote: the
// pixel-reduction kernel in PLAIN mode with both operands the same matrix.
static void gram16_params(const void* a, int rows, int cols, int ld, TN16Params& p) {
    memset(&p, 0, sizeof(p));
    p.A = a; p.Bv = a;
    p.g.ld = ld; p.b_ld = ld;
    p.Ca = cols; p.Cb = cols; p.Mf = cols; p.M = rows; p.out_ld = cols;
}

size_t bg_gram16_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    TN16Params p;
    gram16_params(nullptr, rows, cols, cols, p);
    return tn16_workspace_bytes(p);
}

int bg_gram16(const void* a, int rows, int cols, int ld, float* out, void* ws, size_t ws_bytes, void* stream) {
    BG_REQUIRE(a && out && rows > 0 && cols > 0 && ld >= cols, "bg_gram16: bad argument");
    BG_REQUIRE(cols % 8 == 0 && ld % 8 == 0, "bg_gram16: cols and ld must be multiples of 8");
    TN16Params p;
    gram16_params(a, rows, cols, ld, p);
    Tag tag("gram16", cols, cols, rows);
    ProfScope prof(as_stream(stream), 2.0 * rows * (double)cols * cols, tag.s);
    return launch_tn16(p, GATHER_PLAIN, out, ws, ws_bytes, as_stream(stream));
}

}  // extern "C"
