// igemm.h - parameter blocks of the two MFMA implicit-GEMM kernel families (igemm.hip).
//
//   NN ("tap GEMM"):  out[m, n] = alpha * sum_{tap} sum_{c<C} Agather(m, tap)[c] * B[tap][c][n]
//       rows m enumerate output pixels (b, hq, wq) of one stride-phase; Agather reads the channel
//       vector of a source pixel chosen by the tap (im2col on the fly; reflect / zero padding and
//       transposed-conv divisibility folded into the index map).
//   TN ("pixel-reduction GEMM"):  out[tap][ca][cb] = sum_m Agather(m, tap)[ca] * Bv(m)[cb]
//       the weight-gradient / Gram / A^T B shape: both operands are channel-contiguous row tensors,
//       the reduction runs over pixels (split-K over row ranges).
#pragma once
#include <stdint.h>

namespace bg {

enum { GATHER_CONV = 0, GATHER_TCONV = 1 };

struct Gather {
    int32_t Nb, Hs, Ws;     // source tensor [Nb, Hs, Ws, *]
    int32_t Ho, Wo;         // full output spatial extent (rows enumerate a phase of it)
    int32_t Hq, Wq;         // rows of this launch: (b, hq, wq), ho = hq*pstep + ph
    int32_t pstep;          // 1, or stride for phase-decomposed transposed gathers
    int32_t k, stride, pad;
    int32_t mode;           // GATHER_CONV: src = out*stride + tap - pad ; GATHER_TCONV: src = (out + pad - tap)/stride
    int32_t reflect;        // CONV: reflect the index ; TCONV: add the mirrored sources (gradient of reflect pad)
    int32_t plain;          // 1: rows are simply row indices (no spatial decomposition): src offset = m * ld
    int32_t ld;             // elements between consecutive source pixels
};

struct NNParams {
    const float* A;
    const float* B;
    const float* bias;      // [N] or null
    const float* alpha;     // device scalar or null
    float* out;
    Gather g;
    int32_t C;              // channels per tap (K = taps * C)
    int32_t M, N;
    int64_t tap_stride;     // weight elements between taps
    int32_t ldk, ldn;       // weight strides along c and n
    int32_t out_ld;
    int32_t accumulate;
    int32_t a_vec, b_vec;   // 16-byte vector loads legal
    int32_t batch;          // >1: blockIdx.z = batch item (plain GEMM); else blockIdx.z = phase
    int64_t strideA, strideB, strideC;
    int32_t tiles_n;
};

struct TNParams {
    const float* A;
    const float* Bv;
    float* out;             // [taps][Ca][Cb] (split == 1) or slabs [split][taps][Ca][Cb]
    Gather g;               // CONV-mode gather for A rows
    int32_t Ca, Cb;
    int32_t b_ld;           // elements between consecutive Bv rows
    int32_t M;              // reduction length (rows)
    int32_t splitk, rows_per_split;
    int64_t slab_stride;    // elements per split slab
    int32_t out_ld;         // row stride of the [Ca][Cb] tile (= Cb for weights)
    int64_t out_tap_stride;
    int32_t a_vec, b_vec;
    int32_t batch;
    int64_t strideA, strideB, strideC;
    int32_t tiles_n;
    const float* alpha;
};

}  // namespace bg
