// igemm.h - parameter blocks of the two MFMA implicit-GEMM kernel families (igemm.hip).
//
//   NN ("tap GEMM"):  out[m, n] = alpha * sum_{tap} sum_{c<C} Agather(m, tap)[c] * B[tap][c][n]
//       rows m enumerate output pixels (b, hq, wq) of one stride-phase; Agather reads the channel
//       vector of a source pixel chosen by the tap (im2col on the fly; reflect / zero padding and
//       transposed-conv divisibility folded into the index map).
//   TN ("pixel-reduction GEMM"):  out[(tap, ca)][cb] = sum_m Agather(m, tap)[ca] * Bv(m)[cb]
//       the weight-gradient / Gram / A^T B shape: both operands are channel-contiguous row tensors,
//       the reduction runs over pixels (split-K over row ranges); taps are flattened into the M
//       dimension so small-channel layers still fill a 128-row tile and share the Bv tile.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace bg {

enum { GATHER_CONV = 0, GATHER_TCONV = 1, GATHER_PLAIN = 2 };

struct Gather {
    int32_t Nb, Hs, Ws;     // source tensor [Nb, Hs, Ws, *]
    int32_t Ho, Wo;         // full output spatial extent (rows enumerate a phase of it)
    int32_t Hq, Wq;         // rows of this launch: (b, hq, wq), ho = hq*pstep + ph
    int32_t pstep;          // 1, or stride for phase-decomposed transposed gathers
    int32_t k, stride, pad;
    int32_t reflect;        // CONV: reflect the index ; TCONV: add the mirrored sources (gradient of reflect pad)
    int32_t ld;             // elements between consecutive source pixels
};

struct NNParams {
    const float* A;
    const float* B;
    const float* bias;      // [N] or null
    const float* alpha;     // device scalar or null
    float* out;
    float* slabs;           // split-K partial sums [splitk][rows_total][N] (splitk > 1)
    Gather g;
    int32_t C;              // channels per tap (K = taps * C)
    int32_t M, N;
    int64_t tap_stride;     // weight elements between taps
    int32_t ldk, ldn;       // weight strides along c and n
    int32_t out_ld;
    int32_t accumulate;
    int32_t batch;          // >1: plain batched GEMM
    int32_t splitk;
    int64_t slab_stride;    // elements per slab
    int64_t strideA, strideB, strideC;
    int32_t tiles_m, tiles_n;
    int32_t kchunk;         // K-steps per channel chunk of the (chunk, tap, step) K order; 0 = all channels
    int32_t zfold;          // >0: gridDim.z folded into blockIdx.x, z fastest (phases of one M-tile share an L2)
};

struct TNParams {
    const float* A;
    const float* Bv;
    float* out;             // [Mf][Cb] (split == 1) or slabs [split][batch][Mf][Cb]
    Gather g;               // CONV-mode gather for A rows
    int32_t Ca, Cb;
    int32_t Mf;             // taps * Ca: rows of the output matrix
    int32_t b_ld;           // elements between consecutive Bv rows
    int32_t M;              // reduction length (rows)
    int32_t splitk, rows_per_split;
    int64_t slab_stride;    // elements per split slab
    int32_t out_ld;
    int32_t batch;
    int64_t strideA, strideB, strideC;
    int32_t tiles_m, tiles_n;
    int32_t zfold;          // >0: gridDim.z (batch * splitk) folded into blockIdx.x, split-major per XCD
    const float* alpha;
};

// sum split-K slabs (igemm.hip): out[i] = sum_z ws[z * slab + i]
void launch_slab_reduce(const float* ws, float* out, int64_t n, int splitk, int64_t slab, hipStream_t s);

}  // namespace bg
