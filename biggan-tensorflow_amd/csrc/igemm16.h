// igemm16.h - host interface of the bf16-RESIDENT implicit-GEMM kernels (igemm16.hip).
//
// Tensors of this path are bf16 in HBM (activations NHWC, weights as K-contiguous packed copies written by the
// spectral-norm kernels, BgConvDesc::w_packed); tiles travel global -> LDS with global_load_lds_dwordx4 (no VGPR
// staging, no conversion in the loop), accumulate in fp32 on v_mfma_f32_16x16x32_bf16 and leave as bf16 or fp32.
#pragma once
#include <stdint.h>

#include "igemm.h"

namespace bg {

struct NN16Params {
    const void* A;          // bf16 source tensor of the gather
    const void* B;          // bf16 weights [tap][N][C] (C contiguous)
    const float* bias;      // [N] fp32 or null
    const float* alpha;     // device scalar or null
    void* out;              // bf16 or fp32, rows of out_ld elements
    float* slabs;           // split-K partial sums in output coordinates (splitk > 1)
    float* stats_part;      // halo-tile launches only, nullable: per-block partial batch-norm sums of the stored output,
                            // rows of [sum (N) | sum of squares (N)], one row per (image, patch, phase)
    Gather g;
    int32_t C;              // channels per tap, multiple of 8 (K = taps * C)
    int32_t M, N;
    int64_t tap_stride;     // weight elements between taps (= N * C)
    int32_t out_ld;
    int32_t out_f32;        // 1: out is fp32
    int32_t accumulate;     // out += result (same dtype as out)
    int32_t splitk;
    int64_t slab_stride;
    int32_t tiles_m, tiles_n;
    int32_t zfold;
    int32_t mfast;          // tile order inside an XCD's share of the grid: 1 = M-tiles fastest (weight slab resident in L2)
    int32_t pow2;           // g.Wq and g.Hq are powers of two: row -> (b, hq, wq) by shifts instead of divisions
    int32_t wq_shift, hq_shift;
    int32_t posmajor;       // rows enumerate (position, image) instead of (image, position): a 128-row tile then covers one
                            // or two positions of a small map and walks only the taps that have a source there
    int32_t ring;           // 1: the mirrored-tap launch of the gradient of tf.pad(REFLECT) (pad 1) without the padded grid
                            // (position-major): rows enumerate the pixels that receive mirrored taps - rows {1, Ho - 2} in
                            // full, then columns {1, Wo - 2} without those rows - and each axis carries the g.k real taps
                            // followed by the g.k taps of the MIRRORED padded position (-1 resp. Ho); every (row tap, column
                            // tap) pair except (real, real) is walked.  Together with the plain zero-padding launch
                            // (real, real) that is every padded position folded onto the pixel it mirrors (ops.py:82)
    int32_t ring_lines;     // 2: both borders mirror (stride 1); 1: only the low border does (stride 2, even maps)
};

struct TN16Params {
    const void* A;          // bf16, gathered (CONV mode) pixel rows of Ca channels
    const void* Bv;         // bf16, pixel rows of Cb channels
    float* out;             // fp32 [Mf][Cb] or slabs [split][Mf][Cb]
    Gather g;
    int32_t Ca, Cb;
    int32_t Mf;             // taps * Ca
    int32_t b_ld;
    int32_t M;              // reduction length (pixels)
    int32_t splitk, rows_per_split;
    int64_t slab_stride;
    int32_t out_ld;
    int32_t tiles_m, tiles_n;
    int32_t pow2;           // Wq and Hq are powers of two: shifts instead of divisions in the pixel walk
    int32_t wq_shift, hq_shift;
};

// conv-family entry points of the bf16-resident path (called from igemm.hip's extern "C" functions)
size_t nn16_workspace_bytes(const NN16Params& p, int mode, int zdim, int64_t out_elems);
int launch_nn16(NN16Params& p, int mode, int zdim, int64_t out_elems, void* ws, size_t ws_bytes, hipStream_t s);
// rows of NN16Params::stats_part a launch would write (0: the launch does not take the halo-tile form - no fused statistics)
int64_t nn16_stats_rows(const NN16Params& p, int mode, int zdim);
// the mirrored-tap launch of a reflect-padded convolution's input gradient (p.ring = 1, p.ring_lines, p.g of the plain
// launch, accumulate = 1): see NN16Params::ring
int launch_nn16_ring(NN16Params& p, hipStream_t s);
bool nn16h_d2s_ok(const NN16Params& p);                    // 8-channel stride-2 input gradients: depth-to-space halo form
int launch_nn16h_d2s(const NN16Params& plain, hipStream_t s);
size_t tn16_workspace_bytes(const TN16Params& p);
int launch_tn16(TN16Params& p, int mode, float* final_out, void* ws, size_t ws_bytes, hipStream_t s);
// dx[b,h,w,:] (+)= sum of the padded-grid gradient dxp[b,i,j,:] over the padded positions (i,j) that tf.pad(REFLECT)
// filled from pixel (h,w) (the transpose of ops.py:82's padding); f32: both tensors fp32, else both bf16
int launch_reflect_fold(const void* dxp, void* dx, int f32, int N, int H, int W, int C, int Hp, int Wp, int pad_lo,
                        int accumulate, hipStream_t s);

}  // namespace bg
