// common.h - shared helpers for libbiggan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/biggan_hip.h"

namespace bg {

void set_error(const char* fmt, ...);

#define BG_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            bg::set_error(__VA_ARGS__);               \
            return BG_ERR_ARG;                        \
        }                                             \
    } while (0)

#define BG_LAUNCH_CHECK()                                                        \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            bg::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                          hipGetErrorString(e__));                               \
            return BG_ERR_LAUNCH;                                                \
        }                                                                        \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// wave64 reductions by cross-lane shuffles (CDNA wavefront = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// profiling hooks (core.hip): HIP events on the launch stream around one C-ABI call of a GEMM-family op.
//   flops / bytes = the ALGORITHMIC work of the call (each activation once in its stored type,
//   weights once, weight gradients written once in fp32); prof_kernel() names the kernel instantiation the launcher
//   chose (what rocprofv3 reports), so that bench.py can group by kernel symbol.
struct ProfScope {
    ProfScope(hipStream_t s, double flops, const char* tag = nullptr, double bytes = 0.0);
    ~ProfScope();
    hipStream_t stream;
    int slot;
};
bool prof_on();
void prof_kernel(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: opt in once per (kernel, device).
// Returns false (with the error string set) when the runtime refuses.
bool lds_opt_in(const void* fn, int bytes);

// sums[c] = sum over rows of part[row][c] (fp32 partial rows -> fp64 sums; elementwise.hip).  sums is zeroed first.
int launch_partial_colsum(const float* part, double* sums, int64_t rows, int cols, hipStream_t s);

}  // namespace bg
