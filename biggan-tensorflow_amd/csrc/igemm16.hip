// igemm16.hip - bf16-RESIDENT implicit-GEMM kernels for conv / transposed conv and their gradients (gfx950).
//
// Reference call sites replaced: tf.pad + tf.nn.conv2d (ops.py:82,94-98), tf.nn.conv2d_transpose (ops.py:127-132)
// and the gradients TensorFlow derives for them, in the bf16 configurations of BASELINE.json (configs 3-5).
//
// Why a second family next to igemm.hip / igemm_bf16.h: those kernels read fp32 tensors and round to bf16 while
// staging through VGPRs, which left the bf16 MFMA pipe 80-90 % idle.  Here every operand is
// ALREADY bf16 in HBM, so a tile goes global -> LDS with global_load_lds_dwordx4 (16 bytes per lane, no VGPRs, no
// VALU conversion), the gather of the implicit GEMM is just the per-lane SOURCE address of that instruction, and the
// K loop is {issue next tile's LDS-DMA ; ds_read_b128 fragments ; MFMA ; barrier}.
//
//   nn16_kernel  out[m, n] = sum_tap sum_c A(pixel(m), tap)[c] * B[tap][n][c]        (forward, input gradients)
//       block tile 128 x (32*TN) x 64, 4 waves (2 x 2), v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//       LDS image: rows of 64 bf16 (128 B); 16-byte chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7):
//       LDS-DMA writes are lane-linear, so the swizzle is applied to the SOURCE chunk each lane fetches, and the
//       ds_read_b128 operand reads (lane -> row lane & 15, chunk lane >> 4) are bank-conflict free.
//       K is flattened over (tap, channel chunk): a 64-wide K step may straddle taps (C = 96), every lane keeps its
//       own (tap, channel) cursor.  Out-of-image taps, rows >= M, columns >= N and the K tail fetch a zero page.
//       The epilogue transposes the accumulators through LDS and stores whole 16-byte row segments
//       (bias, alpha, residual accumulate, bf16 or fp32 output).
//       PM = true (NN16Params::posmajor): rows enumerate (position, image) and a tile walks only the taps that have a
//       source at its positions - the frame of a reflect-padded gradient grid, the borders of 4 x 4 / 8 x 8 maps.
//   nn16h_kernel  the same product for 3 x 3 stride-1 gathers and the phases of 4 x 4 stride-2 transposed gathers on maps
//       whose sides are multiples of 16: a block owns a 16 x 16 pixel patch, loads its (16 + NT - 1)^2 source pixels per
//       64-channel chunk ONCE (pixel rows swizzled so that every tap shift reads conflict-free) and streams only the
//       weight tiles; 8 waves, two blocks per CU.
//   tn16_kernel  out[(tap, ca)][cb] = sum_pixels A(pixel, tap)[ca] * Bv(pixel)[cb]   (weight gradients)
//       both tiles stay PIXEL-major in LDS ([64 pixels][128 channels] bf16, chunk-swizzled) and the MFMA operands
//       are read transposed with ds_read_b64_tr_b16; fp32 split-K slabs over pixel ranges.
//   tn16x_kernel  the wide form of it: 256 x 128 output tile, 8 waves, v_mfma_f32_32x32x16_bf16, 3-stage LDS ring of
//       32-pixel stages filled by hand-counted LDS-DMA (s_waitcnt vmcnt(N)), two blocks per CU.
#include <stdlib.h>

#include "common.h"
#include "igemm.h"
#include "igemm16.h"
#include <type_traits>
#include "igemm_dev.h"

namespace bg {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) s16x4v lds_s16x4v;

// the source of every out-of-range 16-byte chunk (zero-initialised device memory of the code object)
__device__ uint4 g_zero_page[512];       // 8 KB: a gather row without a source reads zeros at any channel offset (C <= 4096)

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
    // LDS destination = wave-uniform base + lane * 16 ; the global source address is per lane
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA in inline assembly (MI355X guide 5.7: M0 written in the statement that reads it).  hipcc does not
// model it: it neither counts it in vmcnt nor knows that LDS is written, so the kernel owns the counted
// s_waitcnt vmcnt(N) + barrier before the ds_reads of that tile - and the compiler does NOT drain the queue with a
// vmcnt(0) in front of every ds_read that follows an LDS-DMA in program order, which it does for the builtin
// (SIInsertWaitcnts treats an in-flight LDS-DMA as a store that any LDS read may alias).  That drain is what caps a
// ring with more than one tile in flight; the 2-stage kernels, which wait for everything at their barrier anyway,
// keep the builtin.  lds_byte_addr must be wave-uniform.
__device__ __forceinline__ void glds16_asm(const void* src, uint32_t lds_byte_addr) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// ------------------------------------------------------------------------------------------
// NN kernel
// ------------------------------------------------------------------------------------------
constexpr int NN16_BM = 128;      // block tile rows; the K step is 64 bf16 = one 128-byte LDS row
constexpr int NN16_TAPS = 4;      // taps per axis one block walks (kernel sizes 1, 3, 4)
constexpr int NN16_RING_TAPS = 8; // per-axis tap slots of a ring launch (NN16Params::ring): k real + k mirrored, k <= 4
constexpr int nn16_lds_bytes(int TN, int TAPS = NN16_TAPS) {
    const int stage = (NN16_BM + 32 * TN) * 128;
    const int epi = NN16_BM * (32 * TN + 4) * 4;
    const int tab = 2 * TAPS * NN16_BM * 4;                    // gather tables [axis][tap][row] of int32 behind the stages
    // + output pixel of every row + the tile's tap flags (in the dynamic block on purpose: a static __shared__ variable
    // moves the stages off their 1 KB alignment, which costs the LDS-DMA writes of the HBM-bound layers ~25 %)
    return (2 * stage + tab > epi ? 2 * stage + tab : epi) + NN16_BM * 4 + 16;
}
constexpr int NN16_NOSRC = -(1 << 30);                         // table entry of a tap without a source pixel

// element offset of the source pixel of row r under tap (kh, kw), or -1 (selects, no divergent branches)
template <int MODE>
__device__ __forceinline__ int64_t nn16_src_off(const Gather& g, const RowPos& r, int kh, int kw) {
    int h, w;
    if (MODE == GATHER_CONV) {
        h = conv_src(r.ho, kh, g.stride, g.pad, g.reflect, g.Hs);
        w = conv_src(r.wo, kw, g.stride, g.pad, g.reflect, g.Ws);
    } else {
        h = tconv_src_from_num(r.ho + g.pad - kh, g.stride, g.Hs);
        w = tconv_src_from_num(r.wo + g.pad - kw, g.stride, g.Ws);
    }
    const bool ok = r.valid && h >= 0 && w >= 0;
    const int pix = (r.b * g.Hs + h) * g.Ws + w;          // < 2^31 (checked on the host)
    return ok ? (int64_t)pix * g.ld : (int64_t)-1;
}

// row of the implicit GEMM -> output pixel of this stride phase (shifts when the phase grid is a power of two: the
// divisions of decompose_row cost ~90 VALU instructions per row, as much as several K steps of a C = 96 layer)
// padded position that tf.pad(REFLECT) (pad 1) fills from pixel o of an axis of n pixels, beyond o itself: -1 for o = 1,
// n for o = n - 2 (when the high border is padded at all); NN16_NOMIRROR otherwise
constexpr int NN16_NOMIRROR = -(1 << 20);
__device__ __forceinline__ int nn16_mirror_pos(int o, int n, bool hi) {
    return o == 1 ? -1 : ((hi && o == n - 2) ? n : NN16_NOMIRROR);
}

template <int MODE, bool PM, bool RING = false>
__device__ __forceinline__ RowPos nn16_row(const NN16Params& p, int m, int ph, int pw) {
    RowPos r;
    if (PM) {                                      // image fastest: every row of a tile sits at (nearly) the same position
        r.valid = m < p.M;
        const int pos = m / p.g.Nb;
        r.b = m - pos * p.g.Nb;
        if (RING) {
            // the pixels that receive mirrored taps, each once: rows {1, Ho - 2} in full, then columns {1, Wo - 2} without
            // those rows (ring_lines == 1: row 1 and column 1 only)
            const int L = p.ring_lines, first = L * p.g.Wo;
            if (pos < first) {
                const int line = pos / p.g.Wo;
                r.ho = line == 0 ? 1 : p.g.Ho - 2;
                r.wo = pos - line * p.g.Wo;
            } else {
                const int per = p.g.Ho - L, q = pos - first;
                const int line = q / per, idx = q - line * per;
                r.wo = line == 0 ? 1 : p.g.Wo - 2;
                r.ho = idx == 0 ? 0 : ((L == 2 && idx > p.g.Ho - 4) ? p.g.Ho - 1 : idx + 1);
            }
            return r;
        }
        const int hq = pos / p.g.Wq;
        r.ho = hq * p.g.pstep + ph;
        r.wo = (pos - hq * p.g.Wq) * p.g.pstep + pw;
        return r;
    }
    if (MODE == GATHER_PLAIN || !p.pow2) return decompose_row<MODE>(p.g, m, p.M, ph, pw);
    r.valid = m < p.M;
    const int wq = m & (p.g.Wq - 1);
    const int t = m >> p.wq_shift;
    const int hq = t & (p.g.Hq - 1);
    r.b = t >> p.hq_shift;
    r.ho = hq * p.g.pstep + ph;
    r.wo = wq * p.g.pstep + pw;
    return r;
}

// PM: position-major rows and a K walk over the tile's own taps (NN16Params::posmajor); the image-major form is its own
// instantiation because the tap bookkeeping costs the short-K launches (1 x 1 convolutions: 3 K steps) 25 %.
// RING (NN16Params::ring; PM launches of transposed gathers only): 8 tap slots per axis = real taps followed by the taps
// of the mirrored padded position.
template <int TN, int MODE, bool PM, bool RING = false>
__global__ __launch_bounds__(256, 2) void nn16_kernel(const NN16Params p) {
    static_assert(!RING || (PM && MODE == GATHER_TCONV), "ring launches are position-major transposed gathers");
    constexpr int TAPS = RING ? NN16_RING_TAPS : NN16_TAPS;
    constexpr int LDS_BYTES = nn16_lds_bytes(TN, TAPS);
    typedef typename std::conditional<RING, uint64_t, unsigned>::type mask_t;
    constexpr int BM = NN16_BM, BN = 32 * TN;
    constexpr int JA = BM / 32, JB = BN / 32;      // LDS-DMA instructions per wave and K step
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const Gather& g = p.g;

    // Tile order inside an XCD's contiguous share of the grid (NN16Params::mfast): N-tiles fastest keeps an M-tile's gathered
    // rows in that L2 while the weight slabs stream past (right when the activations are the big operand); M-tiles fastest
    // keeps ONE weight slab (one phase, one N-column: <= 1.5 MB) resident while the rows stream (right for the 4 x 4 ...
    // 16 x 16 layers, whose weights are 5 - 20 x their activations and were re-fetched once per M-tile: 1.9 GB per launch
    // in round 2's counters).
    int tile_m, tile_n, bz = blockIdx.z;
    if (p.zfold > 0) {
        const int lin = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.zfold);
        if (p.mfast) {
            tile_m = lin % p.tiles_m;
            const int r_ = lin / p.tiles_m;
            bz = r_ % p.zfold;
            tile_n = r_ / p.zfold;
        } else {
            bz = lin % p.zfold;
            const int tile = lin / p.zfold;
            tile_n = tile % p.tiles_n;
            tile_m = tile / p.tiles_n;
        }
    } else {
        const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
        if (p.mfast) {
            tile_m = tile % p.tiles_m;
            tile_n = tile / p.tiles_m;
        } else {
            tile_n = tile % p.tiles_n;
            tile_m = tile / p.tiles_n;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int zs = bz % p.splitk, zo = bz / p.splitk;

    int ph = 0, pw = 0;
    if (MODE == GATHER_TCONV) {
        ph = g.pstep - 1 - zo / g.pstep;           // heaviest stride phase first
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
    // ring launches: per axis g.k real taps (kk = i) followed by g.k taps of the mirrored position (kk = i - g.k); the
    // (real, real) pairs belong to the plain launch and are left out of the walk
    int nrh = 0, nrw = 0;
    if (RING) {
        nrh = nrw = g.k;
        nkh = nkw = 2 * g.k;
    }
    const int C8 = p.C >> 3;
    const int ntap = nkh * nkw;

    // ---- gather tables ----
    // The source pixel of (row, tap) separates into a row part and a column part: tabh[ih][row] = (b Hs + h) Ws and
    // tabw[iw][row] = w (NN16_NOSRC where the tap has no source).  Built once per block; a tap change in the K walk
    // then costs two LDS reads and an add per row instead of the whole gather arithmetic (which made the C = 96
    // layers, where the tap changes every 1.5 K steps, VALU-bound at ~11 VALU instructions per MFMA).
    // s_any: bit (axis * NN16_TAPS + i) = some row of this tile has a source under tap i of that axis.  Taps without
    // any are left out of the K walk: the border tiles of a position-major launch (p.posmajor) then cost what their
    // sources cost, not k x k zero-page tiles.
    unsigned& s_any = *reinterpret_cast<unsigned*>(smem + LDS_BYTES - 16);
    if (PM) {
        if (t == 0) s_any = 0;
        __syncthreads();
    }
    unsigned any_local = 0;
    int* tabh = reinterpret_cast<int*>(smem + 2 * STAGE);
    int* tabw = tabh + TAPS * BM;
    for (int e = t; e < 2 * TAPS * BM; e += 256) {
        const int axis = e / (TAPS * BM), i = (e / BM) % TAPS, row = e % BM;
        const RowPos r = nn16_row<MODE, PM, RING>(p, m0 + row, ph, pw);
        int v = NN16_NOSRC;
        if (r.valid && i < (axis ? nkw : nkh)) {
            int kk = (axis ? kw0 : kh0) + i * kstep;
            int o = axis ? r.wo : r.ho;
            const int n = axis ? g.Ws : g.Hs;
            if (RING) {
                const int nr = axis ? nrw : nrh;
                if (i >= nr) {                         // tap of the mirrored padded position
                    kk = i - nr;
                    o = nn16_mirror_pos(o, axis ? g.Wo : g.Ho, p.ring_lines == 2);
                }
            }
            const int src = MODE == GATHER_CONV ? conv_src(o, kk, g.stride, g.pad, g.reflect, n)
                                                : (o == NN16_NOMIRROR ? -1 : tconv_src_from_num(o + g.pad - kk, g.stride, n));
            if (src >= 0) {
                v = axis ? src : (r.b * g.Hs + src) * g.Ws;
                any_local |= 1u << (axis * TAPS + i);
            }
        }
        tabh[e] = v;
    }
    if (PM && any_local) atomicOr(&s_any, any_local);
    // output pixel of every row (transposed gathers: rows enumerate a stride phase, possibly of a padded grid), behind
    // everything the epilogue overlays; 0xffffffff = no output
    unsigned* tabo = reinterpret_cast<unsigned*>(smem + LDS_BYTES - 16 - BM * 4);
    constexpr bool use_tabo = MODE == GATHER_TCONV || PM;
    if (use_tabo && t < BM) {
        const RowPos r = nn16_row<MODE, PM, RING>(p, m0 + t, ph, pw);
        tabo[t] = (r.valid && r.ho < g.Ho && r.wo < g.Wo) ? (unsigned)((r.b * g.Ho + r.ho) * g.Wo + r.wo) : 0xffffffffu;
    }

    // ---- staging role of this lane: 16-byte chunk `cch` of the K step, rows 32 j + rsub of both tiles ----
    const int cch = (lane & 7) ^ ((((w & 1) << 2) | (lane >> 4)) & 7);
    const int rsub = 8 * w + (lane >> 3);
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(p.B);
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
    const unsigned ald2 = 2u * (unsigned)g.ld;
    const uint32_t lds0 = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));

    __syncthreads();                     // tables (and s_any) complete
    // PM: taps this tile walks, bit (ih * NN16_TAPS + iw), in the (ih, iw) order of the packed weights
    mask_t tapmask = 0;
    if (PM) {
        const unsigned any = __builtin_amdgcn_readfirstlane(s_any);      // block-uniform: the K loop stays scalar
        for (int i = 0; i < nkh; ++i)
            for (int j = 0; j < nkw; ++j)
                if (((any >> i) & (any >> (TAPS + j)) & 1u) != 0 && !(RING && i < nrh && j < nrw))
                    tapmask |= (mask_t)1 << (i * TAPS + j);
    }
    const int nsteps_all = ((PM ? __builtin_popcountll((uint64_t)tapmask) : ntap) * C8 + 7) >> 3;
    const int sps = (nsteps_all + p.splitk - 1) / p.splitk;
    const int it0 = zs * sps;
    const int nsteps = max(0, min(nsteps_all, it0 + sps) - it0);

    // K cursor of this lane.  Image-major: tap index tidx = (ih, iw).  PM: rem = the taps from the current one on
    // (lowest set bit = current).
    mask_t rem = tapmask;
    int tidx = 0, c8, ih = 0, iw = 0;
    {
        const int kk = it0 * 8 + cch;
        tidx = kk / C8;
        c8 = kk - tidx * C8;
        if (PM) {
            for (int i = 0; i < tidx && rem; ++i) rem &= rem - 1;
        } else {
            ih = tidx / nkw;
            iw = tidx - ih * nkw;
        }
    }
    const unsigned char* asrc[JA];       // source of this lane's chunk at channel 0 of the current tap (or the zero page)
    const unsigned char* wsrc[JB];
    auto set_tap = [&]() {
        bool kvalid;
        if (PM) {
            kvalid = rem != 0;
            const int bit = kvalid ? __builtin_ctzll((uint64_t)rem) : 0;
            ih = bit / TAPS;
            iw = bit % TAPS;
        } else {
            kvalid = tidx < ntap;
        }
        const int* th = tabh + (ih & (TAPS - 1)) * BM + rsub;
        const int* tw = tabw + (iw & (TAPS - 1)) * BM + rsub;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int pix = th[32 * j] + tw[32 * j];
            asrc[j] = (kvalid & (pix >= 0)) ? abase + (uint64_t)(unsigned)pix * ald2 : zero;
        }
        const int kh = RING ? (ih >= nrh ? ih - nrh : ih) : kh0 + ih * kstep;
        const int kw = RING ? (iw >= nrw ? iw - nrw : iw) : kw0 + iw * kstep;
        const unsigned char* wt = bbase + 2 * (int64_t)(kh * g.k + kw) * p.tap_stride;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int n = n0 + 32 * j + rsub;
            wsrc[j] = (kvalid & (n < p.N)) ? wt + 2 * (uint64_t)((unsigned)n * (unsigned)p.C) : zero;
        }
    };
    set_tap();

    // LDS-DMA through glds16_asm: the table reads of set_tap follow the DMA issue in program order, and behind the
    // builtin the compiler would drain the queue (vmcnt(0)) in front of them; the wait before the barrier is explicit
    auto stage = [&](int buf) {
        const uint32_t sa = lds0 + buf * STAGE + (8 * w) * 128;
        const uint32_t sb = sa + A_BYTES;
        const unsigned cbyte = 16u * (unsigned)c8;
#pragma unroll
        for (int j = 0; j < JA; ++j) glds16_asm(asrc[j] + cbyte, sa + j * 32 * 128);
#pragma unroll
        for (int j = 0; j < JB; ++j) glds16_asm(wsrc[j] + cbyte, sb + j * 32 * 128);
        c8 += 8;
        if (c8 >= C8) {
            do {
                c8 -= C8;
                if (PM) {
                    rem &= rem - 1;
                } else {
                    ++tidx;
                    if (++iw == nkw) {
                        iw = 0;
                        ++ih;
                    }
                }
            } while (c8 >= C8);
            set_tap();
        }
    };

    f32x4_t acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // operand reads: lane -> row (lane & 15) of a 16-row fragment, chunk 4 s + (lane >> 4), swizzled by (row >> 1) & 7
    const int fsw = (lane & 15) >> 1;
    const int a_row = (wm * 64 + (lane & 15)) * 128;
    const int b_row = A_BYTES + (wn * 16 * TN + (lane & 15)) * 128;
    int koff[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) koff[s2] = 16 * ((4 * s2 + (lane >> 4)) ^ fsw);

    if (nsteps > 0) stage(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < nsteps; ++it) {
        const int cur = it & 1;
        if (it + 1 < nsteps) stage(cur ^ 1);
        const unsigned char* sbuf = smem + cur * STAGE;
        // all operand reads of the K step are issued up front (the second half lands behind the first half's MFMAs)
        bf16x8_t a[2][4], b[2][TN];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[s2][i] = *reinterpret_cast<const bf16x8_t*>(sbuf + a_row + i * 16 * 128 + koff[s2]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[s2][j] = *reinterpret_cast<const bf16x8_t*>(sbuf + b_row + j * 16 * 128 + koff[s2]);
        }
        // weights as the MFMA's first operand: D rows = output channels (4 per lane), D columns = pixels
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[s2][j], a[s2][i], acc[i][j], 0, 0, 0);
        // the next tile has landed (this wave's part; the barrier extends it to all) and this one is read out
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: accumulators -> LDS [128][BN + 4] fp32 -> whole 16-byte row segments ----
    constexpr int ELD = BN + 4;
    float* est = reinterpret_cast<float*>(smem);
    const bool partial = p.splitk > 1;
    const float alpha = (!partial && p.alpha) ? *p.alpha : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wm * 64 + 16 * i + (lane & 15);
            const int col = wn * 16 * TN + 16 * j + 4 * (lane >> 4);
            f32x4_t v = acc[i][j];
            v[0] *= alpha; v[1] *= alpha; v[2] *= alpha; v[3] *= alpha;
            *reinterpret_cast<f32x4_t*>(est + row * ELD + col) = v;
        }
    __syncthreads();
    constexpr int CPR = BN / 8;                     // 8-column chunks per row
    float* slab = partial ? p.slabs + (int64_t)zs * p.slab_stride : nullptr;
#pragma unroll
    for (int q = 0; q < BM * CPR / 256; ++q) {
        const int idx = t + 256 * q;
        const int row = idx / CPR, cc = idx - row * CPR;
        const int m = m0 + row, col = n0 + cc * 8;
        if (m >= p.M || col >= p.N) continue;
        int64_t ooff;
        if (use_tabo) {
            const unsigned op = tabo[row];
            if (op == 0xffffffffu) continue;
            ooff = (int64_t)op * p.out_ld + col;
        } else {
            ooff = (int64_t)m * p.out_ld + col;
        }
        f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(est + row * ELD + cc * 8);
        f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(est + row * ELD + cc * 8 + 4);
        if (partial) {
            *reinterpret_cast<f32x4_t*>(slab + ooff) = v0;
            *reinterpret_cast<f32x4_t*>(slab + ooff + 4) = v1;
            continue;
        }
        if (p.bias) {
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + col);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(p.bias + col + 4);
            v0 += b0;
            v1 += b1;
        }
        if (p.out_f32) {
            float* o = reinterpret_cast<float*>(p.out) + ooff;
            if (p.accumulate) {
                v0 += *reinterpret_cast<const f32x4_t*>(o);
                v1 += *reinterpret_cast<const f32x4_t*>(o + 4);
            }
            *reinterpret_cast<f32x4_t*>(o) = v0;
            *reinterpret_cast<f32x4_t*>(o + 4) = v1;
        } else {
            __bf16* o = reinterpret_cast<__bf16*>(p.out) + ooff;
            if (p.accumulate) {
                const uint4 r = *reinterpret_cast<const uint4*>(o);
                v0[0] += bf16_lo(r.x); v0[1] += bf16_hi(r.x); v0[2] += bf16_lo(r.y); v0[3] += bf16_hi(r.y);
                v1[0] += bf16_lo(r.z); v1[1] += bf16_hi(r.z); v1[2] += bf16_lo(r.w); v1[3] += bf16_hi(r.w);
            }
            uint4 r;
            r.x = pack_bf16x2(v0[0], v0[1]);
            r.y = pack_bf16x2(v0[2], v0[3]);
            r.z = pack_bf16x2(v1[0], v1[1]);
            r.w = pack_bf16x2(v1[2], v1[3]);
            *reinterpret_cast<uint4*>(o) = r;
        }
    }
}

// out = alpha * sum_z slabs[z] (+ bias) (+ out), 8 elements per thread; n8 = elements / 8, N % 8 == 0
__global__ __launch_bounds__(256) void nn16_slab_reduce_kernel(const float* __restrict__ slabs, void* __restrict__ out,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ alpha_dev, int64_t n8, int N,
                                                               int splitk, int64_t slab_stride, int accumulate,
                                                               int out_f32) {
    const float alpha = alpha_dev ? *alpha_dev : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        for (int z = 0; z < splitk; ++z) {
            const float* sp = slabs + (int64_t)z * slab_stride + i * 8;
            s0 += *reinterpret_cast<const f32x4_t*>(sp);
            s1 += *reinterpret_cast<const f32x4_t*>(sp + 4);
        }
        s0 *= alpha;
        s1 *= alpha;
        if (bias) {
            const int col = (int)((i * 8) % N);
            s0 += *reinterpret_cast<const f32x4_t*>(bias + col);
            s1 += *reinterpret_cast<const f32x4_t*>(bias + col + 4);
        }
        if (out_f32) {
            float* o = reinterpret_cast<float*>(out) + i * 8;
            if (accumulate) {
                s0 += *reinterpret_cast<const f32x4_t*>(o);
                s1 += *reinterpret_cast<const f32x4_t*>(o + 4);
            }
            *reinterpret_cast<f32x4_t*>(o) = s0;
            *reinterpret_cast<f32x4_t*>(o + 4) = s1;
        } else {
            __bf16* o = reinterpret_cast<__bf16*>(out) + i * 8;
            if (accumulate) {
                const uint4 r = *reinterpret_cast<const uint4*>(o);
                s0[0] += bf16_lo(r.x); s0[1] += bf16_hi(r.x); s0[2] += bf16_lo(r.y); s0[3] += bf16_hi(r.y);
                s1[0] += bf16_lo(r.z); s1[1] += bf16_hi(r.z); s1[2] += bf16_lo(r.w); s1[3] += bf16_hi(r.w);
            }
            uint4 r;
            r.x = pack_bf16x2(s0[0], s0[1]);
            r.y = pack_bf16x2(s0[2], s0[3]);
            r.z = pack_bf16x2(s1[0], s1[1]);
            r.w = pack_bf16x2(s1[2], s1[3]);
            *reinterpret_cast<uint4*>(o) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// NN kernel, halo-tile form: stride-1 window gathers (3 x 3 convolutions forward, 3 x 3 transposed convolutions in
// both directions, each stride phase of a 4 x 4 stride-2 transposed convolution = a 2 x 2 window on the input grid).
//   nn16_kernel re-reads the source pixels once per tap: on the C <= 192 layers (the high-resolution end of every
//   generator / discriminator, most of the FLOPs of the 256^2 / 512^2 models) it is bound by L2 -> LDS bandwidth on
//   that 9-fold im2col re-read (measured r02: ~7 TB/s of TCC traffic at 0.24 - 0.30 MFMA busy).  Here a block owns a
//   16 x 16 patch of output pixels of one image and a slice of BN output channels; per 64-channel chunk it loads the
//   (16 + NT - 1)^2 source pixels ONCE into LDS (128-byte pixel rows, 16-byte chunks XOR-swizzled by (pixel & 6): found by
//   brute force to be conflict-free for a 16-lane fragment starting at ANY pixel, i.e. for every tap shift), and the
//   NT^2 taps read their operands from that tile at shifted pixel indices while only the weight tile of each
//   (chunk, tap) streams in through a 3-stage ring.  L2 -> LDS bytes per MAC: 0.0098 against 0.031.
//   8 waves = 4 (pixel rows) x 2 (channels), wave tile 64 pixels x 16 NF channels, v_mfma_f32_16x16x32_bf16.
// ------------------------------------------------------------------------------------------
constexpr int NH_T = 16;                                   // patch edge
template <int NT> struct NHGeom {
    static constexpr int HW = NH_T + NT - 1;               // halo edge
    static constexpr int NPIX = HW * HW;
    static constexpr int A_INSTR = (NPIX * 8 + 511) / 512; // LDS-DMA instructions per thread and chunk
    static constexpr int A_BYTES = A_INSTR * 8192;
};
constexpr int nh_w_instr(int NF) { return (32 * NF * 8 + 511) / 512; }
// Two blocks per CU (80 KB of LDS each): ONE halo buffer and a 2-tile weight ring.  A block alone on its CU (two halo
// buffers, 3- or 4-tile ring, 147 - 160 KB) was measured 10 - 20 % slower than nn16_kernel: nothing runs while it computes
// its per-lane halo addresses, waits for its first 77 KB or stores its epilogue (~7 of 21 us per block at C = 96).
constexpr int NH_WS = 2;
template <int NT, int NF> constexpr int nn16h_lds_bytes() {
    const int ring = NHGeom<NT>::A_BYTES + NH_WS * nh_w_instr(NF) * 8192;
    const int epi = 128 * (32 * NF + 4) * 4;               // the epilogue goes through LDS in two halves of 128 pixels
    return ring > epi ? ring : epi;
}

// Epilogue of the halo-tile kernels (both forms): accumulators -> LDS [128][BN + 4] fp32 (two halves of the patch) ->
// 16-byte row segments.
// p.stats_part (batch-norm statistics of the OUTPUT, fused: ops.py:630 tf.nn.moments of the tensor this launch writes):
// every stored value, as rounded to its stored type, is written back into the staging tile; the block then reduces the
// tile's columns to one row of per-channel partial sums [sum | sum of squares] - summed over blocks in a fixed order by
// bn_partial_finalize, no atomics.
struct NHTile {                     // which patch / column tile / phase a block owns
    int b, y0, x0, n0, ph, pw, phase, nph, tx, ty, tiles_x, tiles_y;
};
template <int NF, int THIN>
__device__ __forceinline__ void nn16h_epilogue(const NN16Params& p, const NHTile& tl, f32x4_t (&acc)[4][NF],
                                               unsigned char* smem) {
    constexpr int BN = 32 * NF;
    const Gather& g = p.g;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int px = lane & 15, kb = lane >> 4;
    const int b = tl.b, y0 = tl.y0, x0 = tl.x0, n0 = tl.n0, ph = tl.ph, pw = tl.pw, phase = tl.phase, nph = tl.nph;
    const int tx = tl.tx, ty = tl.ty, tiles_x = tl.tiles_x, tiles_y = tl.tiles_y;
    constexpr int ELD = BN + 4;
    float* est = reinterpret_cast<float*>(smem);
    const float alpha = p.alpha ? *p.alpha : 1.0f;
    constexpr int CPR = BN / 8;
    constexpr int RG = 512 / BN;                        // row groups of the column reduction (threads RG * BN .. 511 idle)
    for (int half = 0; half < 2; ++half) {
        if ((wm >> 1) == half) {                        // waves of pixel rows 8 half .. 8 half + 7
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int row = ((wm & 1) * 4 + i) * 16 + px;
                    const int col = wn * 16 * NF + 16 * j + 4 * kb;
                    f32x4_t v = acc[i][j];
                    v[0] *= alpha; v[1] *= alpha; v[2] *= alpha; v[3] *= alpha;
                    *reinterpret_cast<f32x4_t*>(est + row * ELD + col) = v;
                }
        }
        __syncthreads();
        for (int idx = t; idx < 128 * CPR; idx += 512) {
            const int row = idx / CPR, cc = idx - row * CPR;
            const int gy = y0 + 8 * half + (row >> 4), gx = x0 + (row & 15);
            const int col = THIN == 1 ? 0 : n0 + cc * 8;
            const int oy = THIN == 1 ? 2 * gy + (cc >> 1) : gy * g.pstep + ph;
            const int ox = THIN == 1 ? 2 * gx + (cc & 1) : gx * g.pstep + pw;
            f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(est + row * ELD + cc * 8);
            f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(est + row * ELD + cc * 8 + 4);
            const bool live = gy < g.Hq && gx < g.Wq && col < p.N && oy < g.Ho && ox < g.Wo;
            if (live) {
                const int64_t ooff = ((int64_t)(b * g.Ho + oy) * g.Wo + ox) * p.out_ld + col;
                if (p.bias) {
                    v0 += *reinterpret_cast<const f32x4_t*>(p.bias + col);
                    v1 += *reinterpret_cast<const f32x4_t*>(p.bias + col + 4);
                }
                if (p.out_f32) {
                    float* o = reinterpret_cast<float*>(p.out) + ooff;
                    if (p.accumulate) {
                        v0 += *reinterpret_cast<const f32x4_t*>(o);
                        v1 += *reinterpret_cast<const f32x4_t*>(o + 4);
                    }
                    *reinterpret_cast<f32x4_t*>(o) = v0;
                    *reinterpret_cast<f32x4_t*>(o + 4) = v1;
                } else {
                    __bf16* o = reinterpret_cast<__bf16*>(p.out) + ooff;
                    if (p.accumulate) {
                        const uint4 rr = *reinterpret_cast<const uint4*>(o);
                        v0[0] += bf16_lo(rr.x); v0[1] += bf16_hi(rr.x); v0[2] += bf16_lo(rr.y); v0[3] += bf16_hi(rr.y);
                        v1[0] += bf16_lo(rr.z); v1[1] += bf16_hi(rr.z); v1[2] += bf16_lo(rr.w); v1[3] += bf16_hi(rr.w);
                    }
                    uint4 rr;
                    rr.x = pack_bf16x2(v0[0], v0[1]);
                    rr.y = pack_bf16x2(v0[2], v0[3]);
                    rr.z = pack_bf16x2(v1[0], v1[1]);
                    rr.w = pack_bf16x2(v1[2], v1[3]);
                    *reinterpret_cast<uint4*>(o) = rr;
                    if (p.stats_part) {                 // the values as stored
                        v0[0] = bf16_lo(rr.x); v0[1] = bf16_hi(rr.x); v0[2] = bf16_lo(rr.y); v0[3] = bf16_hi(rr.y);
                        v1[0] = bf16_lo(rr.z); v1[1] = bf16_hi(rr.z); v1[2] = bf16_lo(rr.w); v1[3] = bf16_hi(rr.w);
                    }
                }
            } else if (p.stats_part) {
                v0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
                v1 = v0;
            }
            if (p.stats_part) {
                *reinterpret_cast<f32x4_t*>(est + row * ELD + cc * 8) = v0;
                *reinterpret_cast<f32x4_t*>(est + row * ELD + cc * 8 + 4) = v1;
            }
        }
        __syncthreads();
        if (p.stats_part) {                             // (block-uniform)
            // column sums of this half's 128 stored rows -> partial row 2 * block + half
            float st_s = 0.f, st_q = 0.f;
            if (t < RG * BN) {
                const int colr = t % BN;
                for (int row = t / BN; row < 128; row += RG) {
                    const float v = est[row * ELD + colr];
                    st_s += v;
                    st_q = fmaf(v, v, st_q);
                }
            }
            __syncthreads();                            // every read of the tile is done: reuse its first rows
            if (t < RG * BN) {
                est[t] = st_s;
                est[RG * BN + t] = st_q;
            }
            __syncthreads();
            if (t < BN && n0 + t < p.N) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < RG; ++r2) {
                    s += est[r2 * BN + t];
                    q += est[RG * BN + r2 * BN + t];
                }
                const int64_t prow = 2 * ((((int64_t)b * tiles_y + ty) * tiles_x + tx) * nph + phase) + half;
                p.stats_part[prow * (2 * (int64_t)p.N) + n0 + t] = s;
                p.stats_part[prow * (2 * (int64_t)p.N) + p.N + n0 + t] = q;
            }
            __syncthreads();                            // (the next half overwrites the tile)
        }
    }
}

// THIN = 1 (r03): the input gradient of a 3 x 3 stride-2 (pad 1) convolution whose INPUT has 8 channels (the image layers of
// the discriminator) as ONE stride-1 2 x 2-window launch: the 2 x 2 output pixels (2 y + py, 2 x + px) of source pixel
// window (y .. y + 1, x .. x + 1) are the 4 x 8 = 32 output columns of a tile (tap k = p + 1 - 2 w per axis, a zero weight
// tile where that falls outside 0 .. 2), stored depth-to-space.  The four stride phases of the generic path each walked
// their own taps over 32 padded columns: 0.63 ms per call at 46 TF/s for 170 MB of tensors.
template <int NT, int NF, int MODE, int THIN = 0>
__global__ __launch_bounds__(512, 4) void nn16h_kernel(const NN16Params p) {       // 4 waves per SIMD: <= 128 VGPRs
    using G = NHGeom<NT>;
    constexpr int WW = G::HW, NPIX = G::NPIX, A_INSTR = G::A_INSTR, A_BYTES = G::A_BYTES;
    constexpr int BN = 32 * NF, W_INSTR = nh_w_instr(NF), W_BYTES = W_INSTR * 8192, NTAPS = NT * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // halo tile | W ring (NH_WS tiles)

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const Gather& g = p.g;

    // ---- which patch ----
    const int nph = g.pstep * g.pstep;
    const int tiles_x = (g.Wq + NH_T - 1) / NH_T, tiles_y = (g.Hq + NH_T - 1) / NH_T;
    const int nt = (p.N + BN - 1) / BN;
    int r = xcd_remap(blockIdx.x, gridDim.x);
    int phase, tile_n, tx, ty, b;
    if (p.mfast) {              // patches fastest: one (phase, N-column) weight slab stays in the XCD's L2 (see nn16_kernel)
        tx = r % tiles_x; r /= tiles_x;
        ty = r % tiles_y; r /= tiles_y;
        b = r % g.Nb; r /= g.Nb;
        phase = r % nph;
        tile_n = r / nph;
    } else {
        phase = r % nph; r /= nph;
        tile_n = r % nt; r /= nt;
        tx = r % tiles_x; r /= tiles_x;
        ty = r % tiles_y;
        b = r / tiles_y;
    }
    const int y0 = ty * NH_T, x0 = tx * NH_T, n0 = tile_n * BN;
    int ph = 0, pw = 0;
    if (MODE == GATHER_TCONV && g.pstep > 1) {
        ph = g.pstep - 1 - phase / g.pstep;
        pw = g.pstep - 1 - phase % g.pstep;
    }
    // window geometry per axis: halo index hi in [0, NT) <-> weight tap, source = q + base + hi
    //   CONV  (stride 1): source = q + kh - pad                    -> base = -pad,            tap(hi) = hi
    //   TCONV (stride 1): source = q + pad - kh                    -> base = pad - (NT - 1),  tap(hi) = NT - 1 - hi
    //   TCONV phase     : kh = kh0 + 2 i, source = q + d - i       -> base = d - (NT - 1),    tap(hi) = kh0 + 2 (NT - 1 - hi)
    int base_h, base_w, kh0 = 0, kw0 = 0, kstep = 1;
    if (MODE == GATHER_CONV) {
        base_h = base_w = -g.pad;
    } else if (g.pstep > 1) {
        kstep = g.stride;
        kh0 = (ph + g.pad) % g.stride;
        kw0 = (pw + g.pad) % g.stride;
        base_h = (ph + g.pad - kh0) / g.stride - (NT - 1);
        base_w = (pw + g.pad - kw0) / g.stride - (NT - 1);
    } else {
        base_h = base_w = g.pad - (NT - 1);
    }
    // taps this block walks: all NT x NT, except in the stride phases of a kernel with k < NT * stride (3 x 3 stride 2, r03:
    // 1 / 2 / 2 / 4 taps) - there only the window rows / columns hy >= NT - nkh (hx >= NT - nkw) carry a tap
    int nkh = NT, nkw = NT;
    if (MODE == GATHER_TCONV && NT == 2 && g.pstep > 1) {
        nkh = (g.k - kh0 + kstep - 1) / kstep;
        nkw = (g.k - kw0 + kstep - 1) / kstep;
    }
    const int ntaps = (MODE == GATHER_TCONV && NT == 2) ? nkh * nkw : NTAPS;

    const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(p.B);
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
    const uint32_t lds0 = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));
    const uint32_t ldsw = lds0 + A_BYTES;

    // ---- per-lane sources of the halo tile (fixed for the whole block) ----
    const unsigned char* asrc[A_INSTR];
    int acb[A_INSTR];                                   // byte offset of this lane's 8 channels inside a 64-channel chunk
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int pp = (i * 8 + w) * 64 + lane;
        const int q = pp >> 3, c = (pp & 7) ^ (q & 6);
        const int hy = q / WW, hx = q - hy * WW;
        int sy = y0 + hy + base_h, sx = x0 + hx + base_w;
        bool ok = q < NPIX;
        if (MODE == GATHER_CONV && g.reflect) {
            sy = sy < 0 ? -sy : sy;
            sy = sy >= g.Hs ? 2 * (g.Hs - 1) - sy : sy;
            sx = sx < 0 ? -sx : sx;
            sx = sx >= g.Ws ? 2 * (g.Ws - 1) - sx : sx;
        }
        ok = ok && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
        acb[i] = 16 * c;
        asrc[i] = ok ? abase + 2 * ((int64_t)((b * g.Hs + sy) * g.Ws + sx) * g.ld) : zero;
    }
    const unsigned char* wsrc[W_INSTR];
    int wcb[W_INSTR];
    bool wok[W_INSTR];
    int wrow[W_INSTR];
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int pp = (i * 8 + w) * 64 + lane;
        const int n = pp >> 3, c = (pp & 7) ^ ((n >> 1) & 7);
        wok[i] = n < BN && n0 + n < p.N;
        wcb[i] = 16 * c;
        wrow[i] = n;
        wsrc[i] = bbase + 2 * ((int64_t)(THIN == 1 ? (n & 7) : n0 + n) * p.C);      // THIN 1: column = (py, px, channel)
    }
    const uint32_t dma_a = __builtin_amdgcn_readfirstlane(lds0 + w * 1024);
    const uint32_t dma_w = __builtin_amdgcn_readfirstlane(ldsw + w * 1024);

    auto issue_a = [&](int chunk) {                     // the (NT + 15)^2 source pixels of 64 channels
        const uint32_t dst = dma_a;
        const int cb = chunk * 128;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const bool cv = (cb + acb[i]) < 2 * p.C;
            glds16_asm(cv ? asrc[i] + (cb + acb[i]) : zero, dst + i * 8192);
        }
    };
    auto issue_w = [&](int step) {                      // weight tile [BN][64 channels] of (chunk, tap)
        const int chunk = step / ntaps, tap = step - chunk * ntaps;
        const int ty_ = tap / nkw;
        const int hy = NT - nkh + ty_, hx = NT - nkw + (tap - ty_ * nkw);
        const int kh = MODE == GATHER_CONV ? hy : kh0 + kstep * (NT - 1 - hy);
        const int kw = MODE == GATHER_CONV ? hx : kw0 + kstep * (NT - 1 - hx);
        const int64_t toff = 2 * ((int64_t)(kh * g.k + kw) * p.tap_stride);
        const uint32_t dst = dma_w + (step % NH_WS) * W_BYTES;
        const int cb = chunk * 128;
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i) {
            bool cv = wok[i] && (cb + wcb[i]) < 2 * p.C;
            int64_t to = toff;
            if (THIN == 1) {                            // window position (hy, hx), output parity (py, px) -> tap
                const int kty = ((wrow[i] >> 4) & 1) + 1 - 2 * hy, ktx = ((wrow[i] >> 3) & 1) + 1 - 2 * hx;
                cv = cv && kty >= 0 && ktx >= 0;
                to = 2 * ((int64_t)(kty * 3 + ktx) * p.tap_stride);
            }
            glds16_asm(cv ? wsrc[i] + to + (cb + wcb[i]) : zero, dst + i * 8192);
        }
    };

    // ---- operand addresses ----
    const int px = lane & 15, kb = lane >> 4;
    int qb[4];                                          // halo pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) qb[i] = (wm * 4 + i) * WW + px;
    uint32_t boff[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = wn * 16 * NF + 16 * j + px;
        boff[j] = A_BYTES + n * 128 + 16 * (kb ^ ((n >> 1) & 7));          // (byte offset into smem)
    }

    f32x4_t acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int nchunks = (p.C + 63) >> 6;
    const int T = nchunks * ntaps;
    issue_a(0);
    issue_w(0);

    for (int step = 0; step < T; ++step) {
        const int chunk = step / ntaps, tap = step - chunk * ntaps;
        // W(step) (and at a chunk's first tap its halo tile, issued in front of it) has landed: nothing else is in flight
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (step + 1 < T) {
            if (tap == ntaps - 1) {
                // the next step starts a chunk: its halo tile goes into the ONE buffer this step still reads - so this
                // step computes first (below) and issues after a barrier of its own
            } else {
                issue_w(step + 1);
            }
        }

        const int ty_ = tap / nkw;
        const int hy = NT - nkh + ty_, hx = NT - nkw + (tap - ty_ * nkw);
        const uint32_t abuf = 0;
        const uint32_t wslot = (uint32_t)(step % NH_WS) * W_BYTES;
        const int tq = hy * WW + hx;
        const int ksteps = (p.C - chunk * 64) >= 64 ? 2 : 1;           // 32-channel MFMA steps in this chunk
        uint32_t aaddr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = qb[i] + tq;
            aaddr[i] = abuf + q * 128 + 16 * (kb ^ (q & 6));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 < ksteps) {                          // (one 32-channel half at a time: 128 VGPRs at 4 waves per SIMD)
                bf16x8_t a[4], bw[NF];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const bf16x8_t*>(smem + (aaddr[i] ^ (s2 ? 64u : 0u)));
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    bw[j] = *reinterpret_cast<const bf16x8_t*>(smem + ((boff[j] + wslot) ^ (s2 ? 64u : 0u)));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        if (tap == ntaps - 1 && step + 1 < T) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every wave has read the halo tile of this chunk
            __builtin_amdgcn_s_barrier();
            issue_a(chunk + 1);
            issue_w(step + 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    NHTile tl;
    tl.b = b; tl.y0 = y0; tl.x0 = x0; tl.n0 = n0; tl.ph = ph; tl.pw = pw; tl.phase = phase; tl.nph = nph;
    tl.tx = tx; tl.ty = ty; tl.tiles_x = tiles_x; tl.tiles_y = tiles_y;
    nn16h_epilogue<NF, THIN>(p, tl, acc, smem);
}

// ------------------------------------------------------------------------------------------
// Measured and rejected (r03): a second form of the halo kernel - 32-channel chunks (64-byte pixel rows), TWO halo buffers so
// that the next chunk streams in during the current one, two K items per step with their weight tiles issued two steps
// ahead through a 3-slot ring, counted waits (s_waitcnt vmcnt(issued - needed) per wave, nothing drained), 74 - 78 KB of LDS.
// Bit-identical results, config 3 at batch 256: 101.6 -> 106.9 ms per iteration; transposed conv 192 -> 96 at 128^2:
// 3.1 -> 4.4 ms.  A block computes a 32-channel chunk of a 2 x 2 window in ~0.6 us and a halo tile takes ~2 us to arrive: one
// chunk of look-ahead hides less than the single 64-channel buffer of the first form loses, and the 64-byte rows halve the
// coalescing of the DMA's global reads.  Hiding the latency needs ~3 tiles in flight per block, which two blocks per CU
// cannot hold in 160 KB.  (The 64-byte-row swizzle that is conflict-free for ds_read_b128's lane groups, should it be
// needed again: chunk c of row r at c ^ ((r >> 1) & 2).)
// ------------------------------------------------------------------------------------------
// TN kernel (weight gradients): K = pixels, both operands pixel-major in LDS, transposed operand reads
// ------------------------------------------------------------------------------------------
constexpr int TN16_BK = 64;
constexpr int TN16_TILE = TN16_BK * 256;            // bytes of one [64 pixels][128 channels] bf16 image

__device__ __forceinline__ s16x4v lds_tr16(uint32_t lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4v*>(lds_byte_addr));
}
__device__ __forceinline__ int tr16_swz(int row, int chunk) {
    return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void tn16_kernel(const TN16Params p) {
    constexpr int BM = 128, BN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][A image | B image]

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const Gather& g = p.g;

    // split-major logical order inside each XCD: the (M, N) tiles of one pixel range run together on ONE XCD, so the
    // x / dy rows of that range are fetched into that L2 once and shared (the kernel is otherwise bound by L2 misses:
    // every tile streams the whole pixel range)
    const int ntile = p.tiles_m * p.tiles_n;
    const int lin = xcd_remap(blockIdx.x, ntile * p.splitk);
    const int zs = lin / ntile;
    const int tile = lin - zs * ntile;
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int mf0 = tile_m * BM, cb0 = tile_n * BN;
    const int row_begin = zs * p.rows_per_split;
    const int row_end = min(p.M, row_begin + p.rows_per_split);
    const int nsteps = max(0, (row_end - row_begin + TN16_BK - 1) / TN16_BK);

    // staging role: pixel rows 16 j + 4 w + prow of the K tile, channel chunk `ch` (fixed: the swizzle of those rows
    // depends on prow and w only)
    const int prow = lane >> 4;
    const int ch = (lane & 15) ^ ((prow << 2) | w);
    const int mf = mf0 + 8 * ch;
    int a_kh = 0, a_kw = 0, a_c = mf;
    if (MODE != GATHER_PLAIN) {
        const int tap = mf / p.Ca;
        a_c = mf - tap * p.Ca;
        a_kh = tap / g.k;
        a_kw = tap - a_kh * g.k;
    }
    const bool a_ok = mf < p.Mf;
    const int cb = cb0 + 8 * ch;
    const bool b_ok = cb < p.Cb;
    const __bf16* Ab = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bb = reinterpret_cast<const __bf16*>(p.Bv);
    const void* zero = reinterpret_cast<const void*>(g_zero_page);
    int m_next = row_begin + 4 * w + prow;          // pixel of instruction j = 0 of the next K tile

    auto stage = [&](int buf) {
        unsigned char* sa = smem + buf * (2 * TN16_TILE) + (4 * w) * 256;
        unsigned char* sb = sa + TN16_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m_next + 16 * j;
            const void* srca = zero;
            const void* srcb = zero;
            if (m < row_end) {
                if (a_ok) {
                    int64_t pix;
                    if (MODE == GATHER_PLAIN) {
                        pix = m;
                    } else {
                        int b, ho, wo;
                        if (p.pow2) {
                            wo = m & (g.Wq - 1);
                            const int r = m >> p.wq_shift;
                            ho = r & (g.Hq - 1);
                            b = r >> p.hq_shift;
                        } else {
                            wo = m % g.Wq;
                            const int r = m / g.Wq;
                            ho = r % g.Hq;
                            b = r / g.Hq;
                        }
                        const int hs = conv_src(ho, a_kh, g.stride, g.pad, g.reflect, g.Hs);
                        const int ws = conv_src(wo, a_kw, g.stride, g.pad, g.reflect, g.Ws);
                        pix = (hs >= 0 && ws >= 0) ? ((int64_t)b * g.Hs + hs) * g.Ws + ws : -1;
                    }
                    if (pix >= 0) srca = Ab + pix * g.ld + a_c;
                }
                if (b_ok) srcb = Bb + (int64_t)m * p.b_ld + cb;
            }
            glds16(srca, sa + j * 16 * 256);
            glds16(srcb, sb + j * 16 * 256);
        }
        m_next += TN16_BK;
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read addresses (buffer 0, k-step 0): lane -> k block lane >> 5, 16-channel group (lane >> 4) & 1,
    // block row q = (lane & 15) >> 2, column quad pq = lane & 3; read jj covers pixels 8 kblk + 4 jj .. + 3
    const int kblk = lane >> 5, g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
    const uint32_t lds0 = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));
    uint32_t a_ad[2][2], b_ad[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int row = 8 * kblk + 4 * jj + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_ad[i][jj] = lds0 + tr16_swz(row, 4 * (wm * 2 + i) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
            b_ad[i][jj] = lds0 + TN16_TILE + tr16_swz(row, 4 * (wn * 2 + i) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
        }
    }
    auto operand = [&](uint32_t lo_addr, uint32_t hi_addr) {
        const s16x4v lo = lds_tr16(lo_addr), hi = lds_tr16(hi_addr);
        const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, v);
    };

    if (nsteps > 0) stage(0);
    __syncthreads();
    for (int it = 0; it < nsteps; ++it) {
        const int cur = it & 1;
        if (it + 1 < nsteps) stage(cur ^ 1);
        const uint32_t bufoff = (uint32_t)cur * (2u * TN16_TILE);
#pragma unroll
        for (int s = 0; s < TN16_BK / 16; ++s) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = operand(a_ad[i][0] + bufoff + 4096 * s, a_ad[i][1] + bufoff + 4096 * s);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = operand(b_ad[j][0] + bufoff + 4096 * s, b_ad[j][1] + bufoff + 4096 * s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    float* obase = p.out + (int64_t)zs * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mf0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.Mf) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = cb0 + wn * 64 + 32 * j + (lane & 31);
                if (col < p.Cb) obase[(int64_t)row * p.out_ld + col] = acc[i][j][r];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// TN kernel, wide form: 256 x 128 tile, 8 waves, 3-stage LDS ring with counted vmcnt.
//   The 128 x 128 kernel above keeps one 32 KB tile in flight per block (64 KB per CU) and moves 64 FLOP per loaded
//   byte; measured (PMC, r02): 47 % of its wave cycles parked in s_waitcnt / barrier, MFMA busy 22 %, L2 -> LDS traffic
//   6.9 TB/s - throughput = bytes in flight x FLOP per byte / latency.  Here a tile is 48 KB for 1.5x the FLOPs per byte
//   (87), two tiles (96 KB) are in flight behind the one being computed, and the barrier no longer drains the queue:
//   tile t + 2 is issued right after the barrier of iteration t, the wait before that barrier leaves tile t + 1 in
//   flight (s_waitcnt vmcnt(6): 6 LDS-DMA instructions per wave and tile).  One block per CU (144 KB of LDS).
// ------------------------------------------------------------------------------------------
// BK = 32 pixels per stage: a stage is 24 KB, the ring 72 KB, and TWO blocks share a CU (the halo-tile NN kernel's lesson: an
// 8-wave block alone on its CU leaves the matrix pipe idle at every barrier and in its epilogue); BK = 64 is the first form
// (one block per CU, 144 KB), kept for A/B (BG_TN16X_BK=64).
// NBI = 2 (r03): TWO B images, a 256 x 256 output tile, wave tile 64 x 128.  The 256 x 128 form moves 87 FLOP per byte
// from L2 into LDS and reads 64 B of LDS per lane for 4 MFMAs - on the C >= 768 layers (tiny maps, everything comes from
// L2) it sat at the L2 -> LDS limit (~8 TB/s) with the LDS pipe as busy as the matrix pipe; the wider tile is 131 FLOP per
// byte and 96 B per 8 MFMAs.  96 KB of ring: one block per CU, 2 waves per SIMD, 256 VGPRs each.
template <int MODE, int BK, int NBI = 1>
__global__ __launch_bounds__(512, (BK == 32 && NBI == 1) ? 4 : 2) void tn16x_kernel(const TN16Params p) {
    constexpr int TILE = BK * 256;                      // bytes of one [BK pixels][128 channels] bf16 image
    constexpr int TNX_STAGE = (2 + NBI) * TILE;         // A image 0 | A image 1 | B image(s)
    constexpr int NJ = BK / 16;                         // LDS-DMA instructions per wave and A image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 stages

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const Gather& g = p.g;

    const int ntile = p.tiles_m * p.tiles_n;
    const int lin = xcd_remap(blockIdx.x, ntile * p.splitk);
    const int zs = lin / ntile;
    const int tile = lin - zs * ntile;
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int mf0 = tile_m * 256, cb0 = tile_n * 128 * NBI;
    const int row_begin = zs * p.rows_per_split;
    const int row_end = min(p.M, row_begin + p.rows_per_split);
    const int nsteps = max(0, (row_end - row_begin + BK - 1) / BK);

    // staging role: A image (w >> 2), pixel rows 16 j + 4 (w & 3) + prow (j < BK / 16), channel chunk `ch` of that image;
    // B: the same rows for the j's of half (w >> 2)
    const int prow = lane >> 4;
    const int wq = w & 3, ia = w >> 2;
    const int ch = (lane & 15) ^ ((prow << 2) | wq);
    const int mf = mf0 + 128 * ia + 8 * ch;
    int a_kh = 0, a_kw = 0, a_c = mf;
    if (MODE != GATHER_PLAIN) {
        const int tap = mf / p.Ca;
        a_c = mf - tap * p.Ca;
        a_kh = tap / g.k;
        a_kw = tap - a_kh * g.k;
    }
    const int cb = cb0 + (NBI == 2 ? 128 * ia : 0) + 8 * ch;       // NBI = 2: this wave stages rows of B image `ia`
    const void* zero = reinterpret_cast<const void*>(g_zero_page);
    int m_next = row_begin + 4 * wq + prow;
    const uint32_t lds0 = static_cast<uint32_t>(
        reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));

    // The pixel row changes with every K step here (in the NN kernels it is fixed per block), so the row -> source
    // address arithmetic is in the steady-state loop: straight-line integer code without divisions or branches
    // (power-of-two pixel grid: plan_tn16x), about 25 VALU instructions per LDS-DMA.  The first form of this kernel
    // reused the generic gather helpers and spent ~700 instructions per K step for 16 MFMAs - issue-bound at 23 %
    // MFMA busy with the data already in LDS.
    const int dh = a_kh - g.pad, dw = a_kw - g.pad;                   // tap displacement of this lane's channel chunk
    const unsigned hmax = g.reflect ? 0xffffffffu : (unsigned)g.Hs - 1u;   // zero padding: out-of-range sources are zeros
    const unsigned wmax = g.reflect ? 0xffffffffu : (unsigned)g.Ws - 1u;
    const int h2 = 2 * (g.Hs - 1), w2 = 2 * (g.Ws - 1);
    const int wmask = g.Wq - 1, hmask = g.Hq - 1, bsh = p.wq_shift + p.hq_shift;
    const int a_lim = mf < p.Mf ? row_end : 0, b_lim = cb < p.Cb ? row_end : 0;
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(reinterpret_cast<const __bf16*>(p.A) + a_c);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(reinterpret_cast<const __bf16*>(p.Bv) + cb);
    const unsigned ald2 = 2u * (unsigned)g.ld, bld2 = 2u * (unsigned)p.b_ld;

    // wave-uniform LDS destinations as scalars (SALU adds per LDS-DMA instead of VALU + readfirstlane)
    const uint32_t sa0 = __builtin_amdgcn_readfirstlane(lds0 + ia * TILE + (4 * wq) * 256);
    const uint32_t sb0 = __builtin_amdgcn_readfirstlane(lds0 + (2 + (NBI == 2 ? ia : 0)) * TILE + (4 * wq) * 256);
    auto stage = [&](int slot) {
        const uint32_t sa = sa0 + slot * TNX_STAGE;
        const uint32_t sb = sb0 + slot * TNX_STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = m_next + 16 * j;
            unsigned pix = (unsigned)m;
            bool ok = m < a_lim;
            if (MODE != GATHER_PLAIN) {
                const int wo = m & wmask, ho = (m >> p.wq_shift) & hmask, b = m >> bsh;
                const int h = __mul24(ho, g.stride) + dh, wsrc = __mul24(wo, g.stride) + dw;
                const int ha = h < 0 ? -h : h, wa = wsrc < 0 ? -wsrc : wsrc;          // tf.pad(REFLECT): -1 -> 1, n -> n - 2
                const int hr = min(ha, h2 - ha), wr = min(wa, w2 - wa);
                ok = ok & ((unsigned)h <= hmax) & ((unsigned)wsrc <= wmax);
                pix = __umul24(__umul24((unsigned)b, (unsigned)g.Hs) + (unsigned)hr, (unsigned)g.Ws) + (unsigned)wr;
            }
            const void* srca = ok ? static_cast<const void*>(abase + (uint64_t)pix * ald2) : zero;
            glds16_asm(srca, sa + j * 16 * 256);
            if (NBI == 2 || j / (NJ / 2) == ia) {      // (wave-uniform) this wave's share of the B image's rows
                const void* srcb = m < b_lim ? static_cast<const void*>(bbase + (uint64_t)(unsigned)m * bld2) : zero;
                glds16_asm(srcb, sb + j * 16 * 256);
            }
        }
        m_next += BK;
    };

    constexpr int NBJ = 2 * NBI;                       // 32-column MFMA tiles per wave
    f32x16_t acc[2][NBJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kblk = lane >> 5, g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
    uint32_t a_ad[2][2], b_ad[NBJ][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int row = 8 * kblk + 4 * jj + q;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            a_ad[i][jj] = lds0 + (wm >> 1) * TILE +
                          tr16_swz(row, 4 * ((wm & 1) * 2 + i) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
#pragma unroll
        for (int j = 0; j < NBJ; ++j)      // NBI = 1: 32-column blocks 2 wn, 2 wn + 1 of the one image; NBI = 2: all four of image wn
            b_ad[j][jj] = lds0 + (2 + (NBI == 2 ? wn : 0)) * TILE +
                          tr16_swz(row, 4 * (NBI == 2 ? j : wn * 2 + j) + 2 * g16 + (pq >> 1)) + 8 * (pq & 1);
    }
    auto operand = [&](uint32_t lo_addr, uint32_t hi_addr) {
        const s16x4v lo = lds_tr16(lo_addr), hi = lds_tr16(hi_addr);
        const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, v);
    };

    if (nsteps > 0) stage(0);
    if (nsteps > 1) stage(1);
    // one K step on ring slot SLOT (a compile-time constant: the operand addresses are base + immediate)
    auto kstep = [&](int it, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        // tile `it` has landed (this wave's part; the barrier extends that to every wave); tile it + 1 stays in flight
        if (it + 1 < nsteps) {                              // DMA per wave and tile: NJ (A) + NJ / 2 (B), or NJ + NJ with two B images
            if (BK == 64 && NBI == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (BK == 64) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (NBI == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 2 < nsteps) stage((SLOT + 2) % 3);      // the slot every wave finished reading before this barrier
        constexpr uint32_t bufoff = (uint32_t)SLOT * (uint32_t)TNX_STAGE;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8_t a[2], b[NBJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = operand(a_ad[i][0] + (bufoff + 4096 * s), a_ad[i][1] + (bufoff + 4096 * s));
#pragma unroll
            for (int j = 0; j < NBJ; ++j) b[j] = operand(b_ad[j][0] + (bufoff + 4096 * s), b_ad[j][1] + (bufoff + 4096 * s));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NBJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int it = 0; it < nsteps; it += 3) {
        kstep(it, std::integral_constant<int, 0>());
        if (it + 1 < nsteps) kstep(it + 1, std::integral_constant<int, 1>());
        if (it + 2 < nsteps) kstep(it + 2, std::integral_constant<int, 2>());
    }

    float* obase = p.out + (int64_t)zs * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mf0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.Mf) continue;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                const int col = cb0 + wn * 32 * NBJ + 32 * j + (lane & 31);
                if (col < p.Cb) obase[(int64_t)row * p.out_ld + col] = acc[i][j][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// transpose of tf.pad(REFLECT) (ops.py:82): fold the gradient on the padded grid back onto the image
// ------------------------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const void* __restrict__ dxp_, void* __restrict__ dx_, int N,
                                                           int H, int W, int C8, int Hp, int Wp, int pad,
                                                           int accumulate) {
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        int64_t t = i / C8;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        // padded positions that mirror onto h: h + pad, pad - h (h >= 1), 2 (H - 1) - h + pad (h <= H - 2)
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + pad;
        if (h >= 1 && pad - h >= 0) hs[nh++] = pad - h;
        if (h <= H - 2 && 2 * (H - 1) - h + pad < Hp && H > 1) hs[nh++] = 2 * (H - 1) - h + pad;
        ws[nw++] = w + pad;
        if (w >= 1 && pad - w >= 0) ws[nw++] = pad - w;
        if (w <= W - 2 && 2 * (W - 1) - w + pad < Wp && W > 1) ws[nw++] = 2 * (W - 1) - w + pad;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                if (hs[a] >= Hp || ws[b] >= Wp) continue;
                const int64_t off = ((((int64_t)n * Hp + hs[a]) * Wp + ws[b]) * C8 + c8) * 8;
                if (F32) {
                    const float* sp = reinterpret_cast<const float*>(dxp_) + off;
                    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(sp), v1 = *reinterpret_cast<const f32x4_t*>(sp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] += v0[j];
                        acc[4 + j] += v1[j];
                    }
                } else {
                    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(dxp_) + off);
                    acc[0] += bf16_lo(r.x); acc[1] += bf16_hi(r.x); acc[2] += bf16_lo(r.y); acc[3] += bf16_hi(r.y);
                    acc[4] += bf16_lo(r.z); acc[5] += bf16_hi(r.z); acc[6] += bf16_lo(r.w); acc[7] += bf16_hi(r.w);
                }
            }
        if (F32) {
            float* o = reinterpret_cast<float*>(dx_) + i * 8;
            f32x4_t v0 = {acc[0], acc[1], acc[2], acc[3]}, v1 = {acc[4], acc[5], acc[6], acc[7]};
            if (accumulate) {
                v0 += *reinterpret_cast<const f32x4_t*>(o);
                v1 += *reinterpret_cast<const f32x4_t*>(o + 4);
            }
            *reinterpret_cast<f32x4_t*>(o) = v0;
            *reinterpret_cast<f32x4_t*>(o + 4) = v1;
        } else {
            __bf16* o = reinterpret_cast<__bf16*>(dx_) + i * 8;
            if (accumulate) {
                const uint4 r = *reinterpret_cast<const uint4*>(o);
                acc[0] += bf16_lo(r.x); acc[1] += bf16_hi(r.x); acc[2] += bf16_lo(r.y); acc[3] += bf16_hi(r.y);
                acc[4] += bf16_lo(r.z); acc[5] += bf16_hi(r.z); acc[6] += bf16_lo(r.w); acc[7] += bf16_hi(r.w);
            }
            uint4 r;
            r.x = pack_bf16x2(acc[0], acc[1]);
            r.y = pack_bf16x2(acc[2], acc[3]);
            r.z = pack_bf16x2(acc[4], acc[5]);
            r.w = pack_bf16x2(acc[6], acc[7]);
            *reinterpret_cast<uint4*>(o) = r;
        }
    }
}

int launch_reflect_fold(const void* dxp, void* dx, int f32, int N, int H, int W, int C, int Hp, int Wp, int pad_lo,
                        int accumulate, hipStream_t s) {
    BG_REQUIRE(C % 8 == 0, "reflect fold: C %% 8 != 0");
    const int64_t total = (int64_t)N * H * W * (C / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (f32)
        hipLaunchKernelGGL((reflect_fold_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, dxp, dx, N, H, W, C / 8, Hp,
                           Wp, pad_lo, accumulate);
    else
        hipLaunchKernelGGL((reflect_fold_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, dxp, dx, N, H, W, C / 8,
                           Hp, Wp, pad_lo, accumulate);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct NN16Plan {
    int tn, splitk;
};

static int nn16_steps_min(const NN16Params& p, int mode) {
    int per_axis = p.g.k;
    if (mode == GATHER_TCONV && p.g.pstep > 1) per_axis = max(1, p.g.k / p.g.stride);
    return (per_axis * per_axis * (p.C >> 3) + 7) >> 3;
}

// Fraction of the (row, tap) pairs of a launch that have a source pixel: 1 for reflect-padded forward gathers, 16 / 36
// for the input gradient of a 3 x 3 convolution on the padded 6 x 6 grid of a 4 x 4 map.  Separable per axis.
static double nn16_axis_fraction(const Gather& g, int mode, int nq, int n_src) {
    int64_t valid = 0, slots = 0;
    for (int ph = 0; ph < (mode == GATHER_TCONV ? g.pstep : 1); ++ph) {
        int k0 = 0, kstep = 1, nk = g.k;
        if (mode == GATHER_TCONV && g.pstep > 1) {
            kstep = g.stride;
            k0 = (ph + g.pad) % g.stride;
            nk = (g.k - k0 + g.stride - 1) / g.stride;
        }
        for (int q = 0; q < nq; ++q) {
            const int o = q * g.pstep + ph;
            for (int i = 0; i < nk; ++i) {
                const int kk = k0 + i * kstep;
                bool ok;
                if (mode == GATHER_CONV) {
                    const int src = o * g.stride + kk - g.pad;
                    ok = g.reflect || (src >= 0 && src < n_src);
                } else {
                    const int hn = o + g.pad - kk;
                    ok = hn >= 0 && hn % g.stride == 0 && hn / g.stride < n_src;
                }
                valid += ok;
            }
            slots += nk;
        }
    }
    return slots ? (double)valid / (double)slots : 1.0;
}

// Position-major rows (NN16Params::posmajor) when at least a tenth of the tap walk would run on the zero page and a
// 128-row tile covers few positions (BG_NN16_POSMAJOR=0 switches it off; read per call so that a test can compare).
static double nn16_posmajor_fraction(const NN16Params& p, int mode) {
    const char* e = getenv("BG_NN16_POSMAJOR");
    if (e && atoi(e) == 0) return 1.0;
    const Gather& g = p.g;
    if (g.k <= 1 || g.Nb < 16 || g.Hq <= 0 || g.Wq <= 0 || (int64_t)g.Hq * g.Wq > 40 * 40) return 1.0;
    const double f = nn16_axis_fraction(g, mode, g.Hq, g.Hs) * nn16_axis_fraction(g, mode, g.Wq, g.Ws);
    return f < 0.9 ? f : 1.0;
}

// Tile width and split-K by a round model: the grid runs in rounds of `slots` co-resident blocks (2 per CU); a block's
// time is its K steps x (A-side work + one unit per 32 output columns).  Fewer, wider tiles waste padded columns and
// can leave the last round (or the only one) mostly empty; narrow ones re-read the A tile more often.
static NN16Plan plan_nn16(const NN16Params& p, int mode, int zdim, bool allow_split) {
    static const int slots = getenv("BG_NN16_SLOTS") ? atoi(getenv("BG_NN16_SLOTS")) : 512;
    const int64_t tm = (p.M + NN16_BM - 1) / NN16_BM;
    int steps = nn16_steps_min(p, mode);
    const double pm = nn16_posmajor_fraction(p, mode);
    if (pm < 1.0) steps = max(1, (int)(steps * pm + 0.5));       // taps without sources are not walked
    NN16Plan best{4, 1};
    double best_cost = 1e30;
    for (int tn = 4; tn >= 1; --tn) {
        const int bn = 32 * tn;
        const int64_t tiles = tm * ((p.N + bn - 1) / bn) * zdim;
        const double unit = 1.5 + tn;
        const int max_sk = allow_split ? (steps / 4 < 16 ? (steps / 4 < 1 ? 1 : steps / 4) : 16) : 1;
        for (int sk = 1; sk <= max_sk; ++sk) {
            const int64_t rounds = (tiles * sk + slots - 1) / slots;
            double cost = (double)rounds * ((steps + sk - 1) / sk + 2.0) * unit;     // + prologue / epilogue per block
            if (sk > 1) cost += 6.0 * unit + 0.02 * sk * (double)tiles / slots * bn; // reduce launch + slab traffic
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best = NN16Plan{tn, sk};
            }
        }
    }
    return best;
}

size_t nn16_workspace_bytes(const NN16Params& p, int mode, int zdim, int64_t out_elems) {
    const NN16Plan pl = plan_nn16(p, mode, zdim, true);
    return pl.splitk > 1 ? (size_t)pl.splitk * out_elems * sizeof(float) : 0;
}

static int nn16_mfast(const NN16Params& p, int tiles_m, int tiles_n);

// ring launches (NN16Params::ring): mirrored taps of the gradient of tf.pad(REFLECT), accumulated into dx
template <int TN>
static int launch_nn16_ring_inst(const NN16Params& p, dim3 grid, hipStream_t s) {
    constexpr int lds = nn16_lds_bytes(TN, NN16_RING_TAPS);
    if (!lds_opt_in(reinterpret_cast<const void*>(&nn16_kernel<TN, GATHER_TCONV, true, true>), lds)) return BG_ERR_LAUNCH;
    hipLaunchKernelGGL((nn16_kernel<TN, GATHER_TCONV, true, true>), grid, dim3(256), lds, s, p);
    return BG_OK;
}

int launch_nn16_ring(NN16Params& p, hipStream_t s) {
    BG_REQUIRE(p.ring == 1 && (p.ring_lines == 1 || p.ring_lines == 2), "nn16 ring launch: ring = 1, ring_lines 1 or 2");
    BG_REQUIRE(p.g.k >= 1 && p.g.k <= NN16_RING_TAPS / 2 && p.g.Ho >= 4 && p.g.Wo >= 4 && p.accumulate == 1,
               "nn16 ring launch: kernel size %d / map %d x %d not supported", p.g.k, p.g.Ho, p.g.Wo);
    BG_REQUIRE(p.C % 8 == 0 && p.N % 8 == 0 && p.g.ld % 8 == 0 && p.out_ld % 8 == 0, "nn16 ring launch: channels %% 8");
    const int npos = p.ring_lines * p.g.Wo + p.ring_lines * (p.g.Ho - p.ring_lines);
    p.g.pstep = 1;
    p.g.Hq = 1;
    p.g.Wq = npos;
    p.M = p.g.Nb * npos;
    p.posmajor = 1;
    p.splitk = 1;                       // (slabs live in output coordinates: a ring launch touches a few lines of them)
    p.slabs = nullptr;
    p.zfold = 0;
    p.pow2 = 0;
    int tn = 4, best = 1 << 30;
    for (int c = 4; c >= 1; --c) {      // least padded output channels; ties: the wider tile
        const int padded = (p.N + 32 * c - 1) / (32 * c) * (32 * c);
        if (padded < best) { best = padded; tn = c; }
    }
    p.tiles_m = (p.M + NN16_BM - 1) / NN16_BM;
    p.tiles_n = (p.N + 32 * tn - 1) / (32 * tn);
    p.mfast = nn16_mfast(p, p.tiles_m, p.tiles_n);
    dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
    int rc;
    switch (tn) {
        case 1: rc = launch_nn16_ring_inst<1>(p, grid, s); break;
        case 2: rc = launch_nn16_ring_inst<2>(p, grid, s); break;
        case 3: rc = launch_nn16_ring_inst<3>(p, grid, s); break;
        default: rc = launch_nn16_ring_inst<4>(p, grid, s); break;
    }
    if (rc) return rc;
    BG_LAUNCH_CHECK();
    return BG_OK;
}

template <int TN, int MODE, bool PM>
static int launch_nn16_inst(const NN16Params& p, dim3 grid, hipStream_t s) {
    constexpr int lds = nn16_lds_bytes(TN);
    // > 48 KB of dynamic LDS needs the opt-in once per kernel and device
    if (!lds_opt_in(reinterpret_cast<const void*>(&nn16_kernel<TN, MODE, PM>), lds)) return BG_ERR_LAUNCH;
    prof_kernel("nn16_kernel<%d, %d, %s>", TN, MODE, PM ? "true" : "false");
    hipLaunchKernelGGL((nn16_kernel<TN, MODE, PM>), grid, dim3(256), lds, s, p);
    return BG_OK;
}

template <int MODE, bool PM>
static int launch_nn16_pm(const NN16Params& p, int tn, dim3 grid, hipStream_t s) {
    switch (tn) {
        case 1: return launch_nn16_inst<1, MODE, PM>(p, grid, s);
        case 2: return launch_nn16_inst<2, MODE, PM>(p, grid, s);
        case 3: return launch_nn16_inst<3, MODE, PM>(p, grid, s);
        default: return launch_nn16_inst<4, MODE, PM>(p, grid, s);
    }
}

template <int MODE>
static int launch_nn16_mode(const NN16Params& p, int tn, dim3 grid, hipStream_t s) {
    return p.posmajor ? launch_nn16_pm<MODE, true>(p, tn, grid, s) : launch_nn16_pm<MODE, false>(p, tn, grid, s);
}

// ---- halo-tile form: which launches take it ----
static int nn16h_taps(const NN16Params& p, int mode, int zdim) {
    // On by default (BG_NN16_HALO=0 switches it off; read per call so that a test can compare both forms).  Measured r02,
    // config 3 at batch 256: forward sum over the 22 layer shapes 10.62 -> 9.18 ms, input gradients 12.31 -> 11.65 ms;
    // transposed conv 96 -> 96 at 128^2: 600 -> 825 TF/s, 768 -> 768 at 16^2: 1083 -> 1198 (forward), 1276 (input gradient).
    const char* e = getenv("BG_NN16_HALO");
    const int use = e ? atoi(e) : 1;
    static const int cmax = getenv("BG_NN16_HALO_CMAX") ? atoi(getenv("BG_NN16_HALO_CMAX")) : 4096;
    const Gather& g = p.g;
    if (!use || p.C > cmax || p.C % 32 || p.N % 8 || g.Hq < NH_T || g.Wq < NH_T || g.Hq % NH_T || g.Wq % NH_T) return 0;
    if (mode == GATHER_CONV) return (g.stride == 1 && g.k == 3 && g.pstep == 1 && zdim == 1) ? 3 : 0;
    if (g.reflect) return 0;
    if (g.stride == 1 && g.k == 3 && g.pstep == 1 && zdim == 1) return 3;
    if (g.stride == 2 && g.k == 4 && g.pstep == 2 && zdim == 4) return 2;
    // r03: the stride phases of a 3 x 3 stride-2 transposed gather (input gradients of the discriminator's down-sampling
    // convolutions): windows of 1 x 1, 1 x 2, 2 x 1 and 2 x 2 taps inside the 2 x 2 halo.  OFF by default (BG_NN16_HALO_K3S2=1
    // turns it on): measured on config 3 at batch 256, tap kernel -> this form: 64^2 96 <- 192: 0.435 -> 0.402 ms per call,
    // 32^2 192 <- 384: 0.29 -> 0.37 - a block of the 1-tap phase loads a 17 x 17 halo and stores a full patch for a
    // quarter of the MFMA work, so the halo's saving in L2 -> LDS bytes does not pay.
    if (g.stride == 2 && g.k == 3 && g.pad == 1 && g.pstep == 2 && zdim == 4) {
        const char* e3 = getenv("BG_NN16_HALO_K3S2");
        return (e3 && atoi(e3) == 1) ? 2 : 0;
    }
    return 0;
}

template <int NT, int NF, int MODE>
static int launch_nn16h_inst(const NN16Params& p, int blocks, hipStream_t s) {
    constexpr int lds = nn16h_lds_bytes<NT, NF>();
    if (!lds_opt_in(reinterpret_cast<const void*>(&nn16h_kernel<NT, NF, MODE>), lds)) return BG_ERR_LAUNCH;
    prof_kernel("nn16h_kernel<%d, %d, %d>", NT, NF, MODE);
    hipLaunchKernelGGL((nn16h_kernel<NT, NF, MODE>), dim3(blocks), dim3(512), lds, s, p);
    return BG_OK;
}

template <int NT, int MODE>
static int launch_nn16h_nf(const NN16Params& p, int nf, int blocks, hipStream_t s) {
    switch (nf) {
        case 1: return launch_nn16h_inst<NT, 1, MODE>(p, blocks, s);
        case 2: return launch_nn16h_inst<NT, 2, MODE>(p, blocks, s);
        case 3: return launch_nn16h_inst<NT, 3, MODE>(p, blocks, s);
        default: return launch_nn16h_inst<NT, 4, MODE>(p, blocks, s);
    }
}

// The THIN = 1 form of nn16h_kernel (see there): p describes the plain transposed gather (A = dy [Nb, Hs, Ws, C], B = the
// packed [9][8][C] kernel, out = dx [Nb, 2 Hs, 2 Ws, 8]).
bool nn16h_d2s_ok(const NN16Params& p) {
    const char* e = getenv("BG_THIN_D2S");              // (read per call: a test compares both forms)
    const int use = e ? atoi(e) : 1;
    const Gather& g = p.g;
    return use && p.N == 8 && p.out_ld == 8 && g.k == 3 && g.stride == 2 && g.pad == 1 && !g.reflect && p.C % 32 == 0 &&
           g.Hs % NH_T == 0 && g.Ws % NH_T == 0 && g.Ho == 2 * g.Hs && g.Wo == 2 * g.Ws && !p.bias && !p.stats_part;
}
int launch_nn16h_d2s(const NN16Params& plain, hipStream_t s) {
    BG_REQUIRE(nn16h_d2s_ok(plain), "nn16h depth-to-space form: unsupported geometry");
    NN16Params p = plain;
    Gather& g = p.g;
    g.Hq = g.Hs; g.Wq = g.Ws; g.pstep = 1; g.pad = 0; g.stride = 1;
    p.N = 32;
    p.splitk = 1; p.mfast = 0; p.ring = 0;
    const int64_t blocks = (int64_t)g.Nb * (g.Hq / NH_T) * (g.Wq / NH_T);
    BG_REQUIRE(blocks > 0 && blocks < (int64_t(1) << 31), "nn16h: grid out of range");
    constexpr int lds = nn16h_lds_bytes<2, 1>();
    if (!lds_opt_in(reinterpret_cast<const void*>(&nn16h_kernel<2, 1, GATHER_CONV, 1>), lds)) return BG_ERR_LAUNCH;
    prof_kernel("nn16h_kernel<2, 1, 0, d2s>");
    hipLaunchKernelGGL((nn16h_kernel<2, 1, GATHER_CONV, 1>), dim3((int)blocks), dim3(512), lds, s, p);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// Which operand should stay in an XCD's L2 (4 MB) while the other streams: estimated fabric bytes of the two tile orders.
// OFF by default: measured r03 (config 3, batch 256, same box): N-tiles fastest 100.9 ms / iteration, this rule 102.0, M-tiles
// fastest everywhere 103.7 - the weight slabs the N-fastest order re-fetches once per M-tile (round 2's FETCH_SIZE counters:
// 1.9 GB per launch on the 4 x 4 ... 16 x 16 layers) come out of the 256 MB Infinity Cache, not out of HBM, and cost no time,
// while M-fastest makes every co-resident block miss on its own gathered rows.  BG_NN16_MFAST=auto applies the rule, =1
// forces M-fastest (for counter runs).
static int nn16_mfast(const NN16Params& p, int tiles_m, int tiles_n) {
    const char* e = getenv("BG_NN16_MFAST");
    if (!e) return 0;
    if (strcmp(e, "auto") != 0) return atoi(e) != 0;
    const double l2 = 3.0e6;
    const double a_bytes = 2.0 * p.g.Nb * (double)p.g.Hs * p.g.Ws * p.g.ld;
    const double w_bytes = 2.0 * p.g.k * p.g.k * (double)p.N * p.C;
    const double nfast = a_bytes + (w_bytes > l2 ? w_bytes * tiles_m : w_bytes);
    const double mfast = w_bytes + (a_bytes > l2 ? a_bytes * tiles_n : a_bytes);
    return mfast < nfast;
}

static int launch_nn16h(NN16Params& p, int mode, int ntaps, hipStream_t s) {
    int nf = 4, best = 1 << 30;
    // (c = 1, a 32-column tile, r03: the generator's 96 -> 3 (8) image layer - a quarter of the padded MFMA work of the
    //  64-column tile and the input read once instead of once per tap by the tap kernel)
    for (int c = 4; c >= 1; --c) {                      // least padded output channels; ties: the wider tile
        const int padded = (p.N + 32 * c - 1) / (32 * c) * (32 * c);
        if (padded < best) { best = padded; nf = c; }
    }
    const Gather& g = p.g;
    const int64_t blocks = (int64_t)g.Nb * (g.Hq / NH_T) * (g.Wq / NH_T) * ((p.N + 32 * nf - 1) / (32 * nf)) *
                           (g.pstep * g.pstep);
    BG_REQUIRE(blocks > 0 && blocks < (int64_t(1) << 31), "nn16h: grid out of range");
    p.splitk = 1;
    p.mfast = nn16_mfast(p, g.Nb * (g.Hq / NH_T) * (g.Wq / NH_T), (p.N + 32 * nf - 1) / (32 * nf));
    int rc;
    if (mode == GATHER_CONV) rc = launch_nn16h_nf<3, GATHER_CONV>(p, nf, (int)blocks, s);
    else if (ntaps == 3) rc = launch_nn16h_nf<3, GATHER_TCONV>(p, nf, (int)blocks, s);
    else rc = launch_nn16h_nf<2, GATHER_TCONV>(p, nf, (int)blocks, s);
    if (rc) return rc;
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int64_t nn16_stats_rows(const NN16Params& p, int mode, int zdim) {
    if (p.C % 8 || p.N % 8 || !nn16h_taps(p, mode, zdim)) return 0;
    return 2 * (int64_t)p.g.Nb * (p.g.Hq / NH_T) * (p.g.Wq / NH_T) * (p.g.pstep * p.g.pstep);     // one per half patch
}

int launch_nn16(NN16Params& p, int mode, int zdim, int64_t out_elems, void* ws, size_t ws_bytes, hipStream_t s) {
    BG_REQUIRE(p.C % 8 == 0 && p.N % 8 == 0 && p.g.ld % 8 == 0 && p.out_ld % 8 == 0,
               "bf16-resident conv: channel counts must be multiples of 8 (C=%d N=%d)", p.C, p.N);
    BG_REQUIRE((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(p.out) & 15) == 0,
               "bf16-resident conv: tensors must be 16-byte aligned");
    BG_REQUIRE(p.g.k >= 1 && p.g.k <= NN16_TAPS && p.C <= 4096 && p.g.stride >= 1 && p.g.stride <= 2,
               "bf16-resident conv: kernel size %d / stride %d / %d channels not supported", p.g.k, p.g.stride, p.C);
    BG_REQUIRE((int64_t)p.g.Nb * p.g.Hs * p.g.Ws < (int64_t(1) << 30) && (int64_t)p.g.Nb * p.g.Ho * p.g.Wo < (int64_t(1) << 31),
               "bf16-resident conv: more than 2^30 source / 2^31 output pixels");
    if (const int ntaps = nn16h_taps(p, mode, zdim)) return launch_nn16h(p, mode, ntaps, s);
    BG_REQUIRE(p.stats_part == nullptr, "bf16-resident conv: fused statistics need the halo-tile form (nn16_stats_rows)");
    NN16Plan pl = plan_nn16(p, mode, zdim, ws != nullptr);
    if (pl.splitk > 1 && ws_bytes < (size_t)pl.splitk * out_elems * sizeof(float)) pl.splitk = 1;
    p.splitk = pl.splitk;
    p.slabs = reinterpret_cast<float*>(ws);
    p.slab_stride = out_elems;
    const int bn = 32 * pl.tn;
    p.tiles_m = (p.M + NN16_BM - 1) / NN16_BM;
    p.tiles_n = (p.N + bn - 1) / bn;
    p.pow2 = 0;
    if (p.g.Wq > 0 && p.g.Hq > 0 && (p.g.Wq & (p.g.Wq - 1)) == 0 && (p.g.Hq & (p.g.Hq - 1)) == 0) {
        p.pow2 = 1;
        p.wq_shift = __builtin_ctz(p.g.Wq);
        p.hq_shift = __builtin_ctz(p.g.Hq);
    }
    p.posmajor = nn16_posmajor_fraction(p, mode) < 1.0;
    p.mfast = nn16_mfast(p, p.tiles_m, p.tiles_n);
    dim3 grid(p.tiles_m * p.tiles_n, 1, zdim * p.splitk);
    p.zfold = 0;
    if (mode == GATHER_TCONV && zdim > 1 && p.splitk == 1 && p.g.k % p.g.stride == 0) {
        p.zfold = zdim;                     // the phases of one M tile gather the same input rows: same XCD
        grid = dim3(p.tiles_m * p.tiles_n * zdim, 1, 1);
    }
    int rc;
    if (mode == GATHER_CONV)
        rc = launch_nn16_mode<GATHER_CONV>(p, pl.tn, grid, s);
    else
        rc = launch_nn16_mode<GATHER_TCONV>(p, pl.tn, grid, s);
    if (rc) return rc;
    BG_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        const int64_t n8 = out_elems / 8;
        int blocks = (int)((n8 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(nn16_slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, p.out, p.bias, p.alpha, n8,
                           p.N, pl.splitk, out_elems, p.accumulate, p.out_f32);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

static void plan_tn16(TN16Params& p) {
    const int64_t tiles = (int64_t)((p.Mf + 127) / 128) * ((p.Cb + 127) / 128);
    static const int want = getenv("BG_TN16_WANT") ? atoi(getenv("BG_TN16_WANT")) : 1024;
    int sk = (int)(want / tiles);          // (floor: tiles * sk <= want = a whole number of rounds of 512 co-resident blocks)
    const int max_sk = (p.M + 4 * TN16_BK - 1) / (4 * TN16_BK);      // at least 256 pixels per split
    if (sk > max_sk) sk = max_sk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    int rps = (p.M + sk - 1) / sk;
    rps = (rps + TN16_BK - 1) / TN16_BK * TN16_BK;
    p.splitk = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
}

// wide form (tn16x_kernel): used when there are at least two 128-row tiles of output and the reduction fills the ring;
// split so that the grid is a whole number of rounds of 256 co-resident blocks (one per CU)
// 256-wide output tiles (tn16x_kernel<.., 2>) where they divide the output channels: C = 512, 768, 1536.  OFF by default:
// measured r03 (config 3, same box): 102.7 ms / iteration with it, 102.9 without at batch 256, 25.6 / 25.6 at batch 32 -
// the third fewer L2 -> LDS bytes and LDS operand reads buy nothing once only one block (2 waves per SIMD) fits a CU.
// BG_TN16X_NBI=2 switches it on (tests/test_gpu_bf16.py keeps it exact).
static int tn16x_nbi(const TN16Params& p) {
    const char* e = getenv("BG_TN16X_NBI");
    if (!e || atoi(e) != 2) return 1;
    return (p.Cb >= 512 && p.Cb % 256 == 0 && p.Mf >= 256) ? 2 : 1;
}

static bool plan_tn16x(const TN16Params& p, int* sk_out, int* rps_out) {
    static const int use_wide = getenv("BG_TN16_WIDE") ? atoi(getenv("BG_TN16_WIDE")) : 1;
    if (!use_wide || p.Mf <= 128 || p.M < 8 * TN16_BK) return false;
    // the kernel walks a power-of-two pixel grid by shifts (every BigGAN resolution is one); others take tn16_kernel
    if (p.g.k > 0 && (p.g.Wq <= 0 || p.g.Hq <= 0 || (p.g.Wq & (p.g.Wq - 1)) || (p.g.Hq & (p.g.Hq - 1)))) return false;
    // (g.k == 0: plain rows, no pixel walk)
    const int nbi = tn16x_nbi(p);
    const int tm = (p.Mf + 255) / 256, tn = (p.Cb + 128 * nbi - 1) / (128 * nbi);
    static const int wantx0 = getenv("BG_TN16X_WANT") ? atoi(getenv("BG_TN16X_WANT")) : 512;
    const int wantx = nbi == 2 ? wantx0 / 2 : wantx0;           // (one block per CU with the 96 KB ring)
    int sk = wantx / (tm * tn);
    const int max_sk = (p.M + 4 * TN16_BK - 1) / (4 * TN16_BK);
    if (sk > max_sk) sk = max_sk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    int rps = (p.M + sk - 1) / sk;
    rps = (rps + TN16_BK - 1) / TN16_BK * TN16_BK;
    *sk_out = (p.M + rps - 1) / rps;
    *rps_out = rps;
    return true;
}

size_t tn16_workspace_bytes(const TN16Params& p0) {
    TN16Params p = p0;
    plan_tn16(p);
    int sk = p.splitk, skx = 1, rps = 0;
    if (plan_tn16x(p0, &skx, &rps) && skx > sk) sk = skx;
    return sk > 1 ? (size_t)sk * p.Mf * p.Cb * sizeof(float) : 0;
}

int launch_tn16(TN16Params& p, int mode, float* final_out, void* ws, size_t ws_bytes, hipStream_t s) {
    BG_REQUIRE(p.Ca % 8 == 0 && p.Cb % 8 == 0 && p.g.ld % 8 == 0 && p.b_ld % 8 == 0,
               "bf16-resident wgrad: channel counts must be multiples of 8 (Ca=%d Cb=%d)", p.Ca, p.Cb);
    BG_REQUIRE((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.Bv) & 15) == 0,
               "bf16-resident wgrad: tensors must be 16-byte aligned");
    plan_tn16(p);
    const int64_t total = (int64_t)p.Mf * p.Cb;
    if (p.splitk > 1 && (ws == nullptr || ws_bytes < (size_t)p.splitk * total * sizeof(float) || p.out_ld != p.Cb)) {
        p.splitk = 1;
        p.rows_per_split = (p.M + TN16_BK - 1) / TN16_BK * TN16_BK;
    }
    if (p.splitk > 1) {
        p.out = reinterpret_cast<float*>(ws);
        p.slab_stride = total;
    } else {
        p.out = final_out;
        p.slab_stride = 0;
    }
    p.tiles_m = (p.Mf + 127) / 128;
    p.tiles_n = (p.Cb + 127) / 128;
    p.pow2 = 0;
    if (mode != GATHER_PLAIN && (p.g.Wq & (p.g.Wq - 1)) == 0 && (p.g.Hq & (p.g.Hq - 1)) == 0) {
        p.pow2 = 1;
        p.wq_shift = __builtin_ctz(p.g.Wq);
        p.hq_shift = __builtin_ctz(p.g.Hq);
    }
    int sk = 1, rps = 0;
    if (plan_tn16x(p, &sk, &rps)) {
        const int nbi = tn16x_nbi(p);
        const int tm = (p.Mf + 255) / 256, tn = (p.Cb + 128 * nbi - 1) / (128 * nbi);
        const bool can_split = ws != nullptr && ws_bytes >= (size_t)sk * total * sizeof(float) && p.out_ld == p.Cb;
        if (sk == 1 || can_split) {
            p.splitk = sk;
            p.rows_per_split = rps;
            p.out = sk > 1 ? reinterpret_cast<float*>(ws) : final_out;
            p.slab_stride = sk > 1 ? total : 0;
            p.tiles_m = tm;
            p.tiles_n = tn;
            static const int bk = getenv("BG_TN16X_BK") ? atoi(getenv("BG_TN16X_BK")) : 32;
            if (nbi == 2) {
                constexpr int ldsw = 3 * 4 * 32 * 256;
                const void* fw = mode == GATHER_CONV ? reinterpret_cast<const void*>(&tn16x_kernel<GATHER_CONV, 32, 2>)
                                                     : reinterpret_cast<const void*>(&tn16x_kernel<GATHER_PLAIN, 32, 2>);
                if (!lds_opt_in(fw, ldsw)) return BG_ERR_LAUNCH;
                prof_kernel("tn16x_kernel<%d, 32, 2>", mode == GATHER_CONV ? GATHER_CONV : GATHER_PLAIN);
                dim3 gridw(tm * tn * sk, 1, 1);
                if (mode == GATHER_CONV) hipLaunchKernelGGL((tn16x_kernel<GATHER_CONV, 32, 2>), gridw, dim3(512), ldsw, s, p);
                else hipLaunchKernelGGL((tn16x_kernel<GATHER_PLAIN, 32, 2>), gridw, dim3(512), ldsw, s, p);
                BG_LAUNCH_CHECK();
                if (sk > 1) {
                    launch_slab_reduce(reinterpret_cast<const float*>(ws), final_out, total, sk, total, s);
                    BG_LAUNCH_CHECK();
                }
                return BG_OK;
            }
            const int ldsx = 3 * 3 * bk * 256;
            const void* fx = mode == GATHER_CONV
                                 ? (bk == 64 ? reinterpret_cast<const void*>(&tn16x_kernel<GATHER_CONV, 64>)
                                             : reinterpret_cast<const void*>(&tn16x_kernel<GATHER_CONV, 32>))
                                 : (bk == 64 ? reinterpret_cast<const void*>(&tn16x_kernel<GATHER_PLAIN, 64>)
                                             : reinterpret_cast<const void*>(&tn16x_kernel<GATHER_PLAIN, 32>));
            if (!lds_opt_in(fx, ldsx)) return BG_ERR_LAUNCH;
            prof_kernel("tn16x_kernel<%d, %d>", mode == GATHER_CONV ? GATHER_CONV : GATHER_PLAIN, bk == 64 ? 64 : 32);
            dim3 gridx(tm * tn * sk, 1, 1);
            if (mode == GATHER_CONV) {
                if (bk == 64) hipLaunchKernelGGL((tn16x_kernel<GATHER_CONV, 64>), gridx, dim3(512), ldsx, s, p);
                else hipLaunchKernelGGL((tn16x_kernel<GATHER_CONV, 32>), gridx, dim3(512), ldsx, s, p);
            } else {
                if (bk == 64) hipLaunchKernelGGL((tn16x_kernel<GATHER_PLAIN, 64>), gridx, dim3(512), ldsx, s, p);
                else hipLaunchKernelGGL((tn16x_kernel<GATHER_PLAIN, 32>), gridx, dim3(512), ldsx, s, p);
            }
            BG_LAUNCH_CHECK();
            if (sk > 1) {
                launch_slab_reduce(reinterpret_cast<const float*>(ws), final_out, total, sk, total, s);
                BG_LAUNCH_CHECK();
            }
            return BG_OK;
        }
    }
    dim3 grid(p.tiles_m * p.tiles_n * p.splitk, 1, 1);
    constexpr int lds = 4 * TN16_TILE;
    if (!lds_opt_in(mode == GATHER_CONV ? reinterpret_cast<const void*>(&tn16_kernel<GATHER_CONV>)
                                        : reinterpret_cast<const void*>(&tn16_kernel<GATHER_PLAIN>), lds))
        return BG_ERR_LAUNCH;
    prof_kernel("tn16_kernel<%d>", mode == GATHER_CONV ? GATHER_CONV : GATHER_PLAIN);
    if (mode == GATHER_CONV)
        hipLaunchKernelGGL((tn16_kernel<GATHER_CONV>), grid, dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((tn16_kernel<GATHER_PLAIN>), grid, dim3(256), lds, s, p);
    BG_LAUNCH_CHECK();
    if (p.splitk > 1) {
        launch_slab_reduce(reinterpret_cast<const float*>(ws), final_out, total, p.splitk, total, s);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

}  // namespace bg
