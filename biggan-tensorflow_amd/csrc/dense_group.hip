// dense_group.hip - grouped launch for small dense projections that share their batch rows.
//
// Reference call site: condition_batch_norm's beta / gamma projections (ops.py:623-624: two fully_connected(z -> C)
// per conditional batch norm, ops.py:163-165 each).  A generator block has two conditional batch norms on the SAME z
// chunk, i.e. four [B, 32 (+ n_labels)] x [K, C] products of ~10 MFLOP: as separate GEMM launches they cost a launch
// each (forward) and three each backward (weight gradient, bias gradient, split-K reduce) - ~190 launches of ~6 us
// per training iteration (round 2 kernel statistics), pure launch and host overhead under strong scaling.
// Here the projections of one block are ONE launch per direction.  fp32 FMA on the vector ALU: the work is far too
// small for a matrix tile to matter; no atomics (each output element is produced by one thread, batch order fixed),
// so the result is reproducible.
#include "common.h"

namespace bg {

constexpr int DG_COLS = 256;      // threads per block = output columns per block
constexpr int DG_ROWS = 8;        // batch rows (forward) / K rows (weight gradient) per block

struct DenseGroupArgs {
    BgDenseItem item[BG_DENSE_GROUP_MAX];
    int32_t tile_start[BG_DENSE_GROUP_MAX + 1];     // first column tile of every item in blockIdx.x
    int32_t n, B;
};

__device__ __forceinline__ int dg_find(const DenseGroupArgs& a, int tile) {
    int j = 0;
    while (j + 1 < a.n && a.tile_start[j + 1] <= tile) ++j;
    return j;
}

// y[b, c] = bias[c] + sum_k x[b, k] w[k, c];  blockIdx.x = (item, column tile), blockIdx.y = strip of DG_ROWS rows
__global__ __launch_bounds__(DG_COLS) void dense_group_fwd_kernel(const DenseGroupArgs a) {
    extern __shared__ float xs[];                      // [DG_ROWS][K]
    const int j = dg_find(a, blockIdx.x);
    const BgDenseItem it = a.item[j];
    const int c = (blockIdx.x - a.tile_start[j]) * DG_COLS + threadIdx.x;
    const int b0 = blockIdx.y * DG_ROWS;
    const int K = it.K;
    for (int e = threadIdx.x; e < DG_ROWS * K; e += DG_COLS) {
        const int r = e / K, k = e - r * K;
        xs[e] = (b0 + r < a.B) ? it.x[(int64_t)(b0 + r) * it.ldx + k] : 0.f;
    }
    __syncthreads();
    if (c >= it.N) return;
    float acc[DG_ROWS];
    const float bv = it.bias ? it.bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < DG_ROWS; ++r) acc[r] = bv;
    const float* wp = it.w + c;
    for (int k = 0; k < K; ++k) {
        const float wv = wp[(int64_t)k * it.N];
#pragma unroll
        for (int r = 0; r < DG_ROWS; ++r) acc[r] = fmaf(xs[r * K + k], wv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < DG_ROWS; ++r)
        if (b0 + r < a.B) it.y[(int64_t)(b0 + r) * it.N + c] = acc[r];
}

// dw[k, c] (+)= sum_b x[b, k] dy[b, c]  (blockIdx.y = strip of DG_ROWS values of k);  the strip that holds k = 0 also
// produces db[c] (+)= sum_b dy[b, c].  The batch is walked in chunks of DG_BCH rows staged in LDS.
constexpr int DG_BCH = 64;
__global__ __launch_bounds__(DG_COLS) void dense_group_wgrad_kernel(const DenseGroupArgs a) {
    __shared__ float xs[DG_BCH][DG_ROWS];
    const int j = dg_find(a, blockIdx.x);
    const BgDenseItem it = a.item[j];
    const int c = (blockIdx.x - a.tile_start[j]) * DG_COLS + threadIdx.x;
    const int k0 = blockIdx.y * DG_ROWS;
    if (k0 >= it.K) return;                            // (block-uniform: items of a group may differ in K)
    float acc[DG_ROWS], sb = 0.f;
#pragma unroll
    for (int r = 0; r < DG_ROWS; ++r) acc[r] = 0.f;
    const bool live = c < it.N;
    for (int bb = 0; bb < a.B; bb += DG_BCH) {
        __syncthreads();
        for (int e = threadIdx.x; e < DG_BCH * DG_ROWS; e += DG_COLS) {
            const int b = e / DG_ROWS, r = e - b * DG_ROWS;
            xs[b][r] = (bb + b < a.B && k0 + r < it.K) ? it.x[(int64_t)(bb + b) * it.ldx + k0 + r] : 0.f;
        }
        __syncthreads();
        if (live) {
            const int nb = min(DG_BCH, a.B - bb);
            for (int b = 0; b < nb; ++b) {
                const float g = it.y[(int64_t)(bb + b) * it.N + c];        // (y holds dy in this direction)
                sb += g;
#pragma unroll
                for (int r = 0; r < DG_ROWS; ++r) acc[r] = fmaf(xs[b][r], g, acc[r]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < DG_ROWS; ++r)
        if (k0 + r < it.K) {
            float* o = it.dw + (int64_t)(k0 + r) * it.N + c;
            *o = it.acc_w ? *o + acc[r] : acc[r];
        }
    if (k0 == 0 && it.db) it.db[c] = it.acc_b ? it.db[c] + sb : sb;
}

static int dg_prepare(const BgDenseItem* items, int n, int B, DenseGroupArgs& a, int* kmax, const char* who, bool bwd) {
    BG_REQUIRE(items && n >= 1 && n <= BG_DENSE_GROUP_MAX && B >= 1, "%s: 1..%d items and a positive batch", who,
               BG_DENSE_GROUP_MAX);
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.B = B;
    int tiles = 0, km = 0;
    for (int i = 0; i < n; ++i) {
        const BgDenseItem& it = items[i];
        BG_REQUIRE(it.x && it.y && it.K >= 1 && it.N >= 1 && it.ldx >= it.K, "%s: item %d: bad operand", who, i);
        BG_REQUIRE(bwd ? it.dw != nullptr : it.w != nullptr, "%s: item %d: null weight operand", who, i);
        a.item[i] = it;
        a.tile_start[i] = tiles;
        tiles += (it.N + DG_COLS - 1) / DG_COLS;
        km = it.K > km ? it.K : km;
    }
    a.tile_start[n] = tiles;
    *kmax = km;
    return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" {

int bg_dense_group_fwd(const BgDenseItem* items, int n_items, int B, void* stream) {
    DenseGroupArgs a;
    int kmax = 0;
    int rc = dg_prepare(items, n_items, B, a, &kmax, "bg_dense_group_fwd", false);
    if (rc) return rc;
    const size_t lds = (size_t)DG_ROWS * kmax * sizeof(float);
    BG_REQUIRE(lds <= 48 * 1024, "bg_dense_group_fwd: K = %d is too long for the row strip in LDS", kmax);
    dim3 grid(a.tile_start[a.n], (B + DG_ROWS - 1) / DG_ROWS);
    hipLaunchKernelGGL(dense_group_fwd_kernel, grid, dim3(DG_COLS), lds, as_stream(stream), a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_dense_group_wgrad(const BgDenseItem* items, int n_items, int B, void* stream) {
    DenseGroupArgs a;
    int kmax = 0;
    int rc = dg_prepare(items, n_items, B, a, &kmax, "bg_dense_group_wgrad", true);
    if (rc) return rc;
    dim3 grid(a.tile_start[a.n], (kmax + DG_ROWS - 1) / DG_ROWS);
    hipLaunchKernelGGL(dense_group_wgrad_kernel, grid, dim3(DG_COLS), 0, as_stream(stream), a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
