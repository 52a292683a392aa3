// thinconv.hip - direct kernels for 3 x 3 convolutions whose INPUT has <= 4 channels: the discriminator's
// first layer (resblock_down_1 res1 / skip: image [B,S,S,3] -> 64 channels, stride 2, reflect padding;
// ops.py:49-113 via ops.py:293-313).  As an implicit GEMM this layer has K = 27: a 128 x 32 tile spends
// 90 % of its MFMA work on padding (4.6 TF/s measured), while the layer is a pure stream over the wide
// output (forward / wgrad) or the wide dy (dgrad).  Here a thread owns 4 consecutive output channels and
// keeps its 9 x Cin x 4 weights in registers; Cout / 4 threads cover one pixel (coalesced float4 rows).
#include "common.h"

namespace bg {

#define THIN_MAXCIN 4

struct ThinGeom {
    int N, H, W, Cin, Ho, Wo, Cout, stride, pad, reflect, tpp, ppb;   // tpp = threads per pixel, ppb = pixels per block
};

__device__ __forceinline__ int thin_src(int o, int kk, int stride, int pad, int reflect, int n) {
    int s = o * stride + kk - pad;
    if (reflect) {
        s = s < 0 ? -s : s;
        s = s >= n ? 2 * (n - 1) - s : s;
        return s;
    }
    return (s >= 0 && s < n) ? s : -1;
}

// weights of this thread's 4 output channels: wr[tap][ci] (float4 over co)
__device__ __forceinline__ void thin_load_w(const float* __restrict__ w, const ThinGeom& g, int co4,
                                            float4 (&wr)[9][THIN_MAXCIN]) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < THIN_MAXCIN; ++ci)
            wr[t][ci] = ci < g.Cin ? *reinterpret_cast<const float4*>(w + ((int64_t)(t * g.Cin + ci)) * g.Cout + co4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
}

// y[pix][co] = sum_tap sum_ci x[src(pix, tap)][ci] * w[tap][ci][co] (+ bias)
__global__ __launch_bounds__(256) void thin_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            ThinGeom g) {
    const int grp = threadIdx.x / g.tpp, co4 = (threadIdx.x % g.tpp) * 4;
    if (grp >= g.ppb) return;
    float4 wr[9][THIN_MAXCIN];
    thin_load_w(w, g, co4, wr);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + co4);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    for (int64_t pix = (int64_t)blockIdx.x * g.ppb + grp; pix < npix; pix += (int64_t)gridDim.x * g.ppb) {
        const int wo = (int)(pix % g.Wo);
        const int64_t t = pix / g.Wo;
        const int ho = (int)(t % g.Ho);
        const int64_t b = t / g.Ho;
        float4 acc = b4;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hs = thin_src(ho, kh, g.stride, g.pad, g.reflect, g.H);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ws = thin_src(wo, kw, g.stride, g.pad, g.reflect, g.W);
                if (hs < 0 || ws < 0) continue;
                const float* xp = x + ((b * g.H + hs) * g.W + ws) * g.Cin;
#pragma unroll
                for (int ci = 0; ci < THIN_MAXCIN; ++ci) {
                    if (ci >= g.Cin) break;
                    const float xv = xp[ci];
                    const float4 wv = wr[kh * 3 + kw][ci];
                    acc.x += xv * wv.x;
                    acc.y += xv * wv.y;
                    acc.z += xv * wv.z;
                    acc.w += xv * wv.w;
                }
            }
        }
        *reinterpret_cast<float4*>(y + pix * g.Cout + co4) = acc;
    }
}

// Output positions whose tap kk reads input index i: o * stride + kk - pad == i, directly or (reflect) through
// the mirrored padded index.  Writes up to 2 candidates, returns the count.
__device__ __forceinline__ int thin_dst(int i, int kk, int stride, int pad, int reflect, int n, int nout, int (&o)[2]) {
    int cnt = 0;
    int num = i + pad - kk;                      // direct: padded index i + pad
    if (num >= 0 && num % stride == 0 && num / stride < nout) o[cnt++] = num / stride;
    if (reflect) {
        // padded index q < pad mirrors input pad - q; q >= n + pad mirrors input 2 (n - 1) - (q - pad)
        int q = -1;
        if (i >= 1 && i <= pad) q = pad - i;
        else if (i >= n - 1 - pad && i <= n - 2) q = 2 * (n - 1) - i + pad;
        if (q >= 0) {
            num = q - kk;
            if (num >= 0 && num % stride == 0 && num / stride < nout) o[cnt++] = num / stride;
        }
    }
    return cnt;
}

// dx[pix][ci] = sum over (tap, output pixel reading pix through tap) sum_co dy[opix][co] * w[tap][ci][co]
__global__ __launch_bounds__(256) void thin_conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, ThinGeom g) {
    __shared__ float red[256][THIN_MAXCIN];
    const int grp = threadIdx.x / g.tpp, tl = threadIdx.x % g.tpp, co4 = tl * 4;
    const bool active = grp < g.ppb;
    float4 wr[9][THIN_MAXCIN];
    if (active) thin_load_w(w, g, co4, wr);
    const int64_t npix = (int64_t)g.N * g.H * g.W;
    const int64_t iters = (npix + (int64_t)gridDim.x * g.ppb - 1) / ((int64_t)gridDim.x * g.ppb);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t pix = (it * gridDim.x + blockIdx.x) * g.ppb + grp;
        float acc[THIN_MAXCIN] = {0.f, 0.f, 0.f, 0.f};
        if (active && pix < npix) {
            const int wi = (int)(pix % g.W);
            const int64_t t = pix / g.W;
            const int hi = (int)(t % g.H);
            const int64_t b = t / g.H;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                int hos[2];
                const int nh = thin_dst(hi, kh, g.stride, g.pad, g.reflect, g.H, g.Ho, hos);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    int wos[2];
                    const int nw = thin_dst(wi, kw, g.stride, g.pad, g.reflect, g.W, g.Wo, wos);
                    for (int a = 0; a < nh; ++a)
                        for (int e = 0; e < nw; ++e) {
                            const float4 d = *reinterpret_cast<const float4*>(
                                dy + ((b * g.Ho + hos[a]) * g.Wo + wos[e]) * g.Cout + co4);
#pragma unroll
                            for (int ci = 0; ci < THIN_MAXCIN; ++ci) {
                                const float4 wv = wr[kh * 3 + kw][ci];
                                acc[ci] += d.x * wv.x + d.y * wv.y + d.z * wv.z + d.w * wv.w;
                            }
                        }
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < THIN_MAXCIN; ++ci) red[threadIdx.x][ci] = acc[ci];
        __syncthreads();
        if (active && pix < npix && tl < g.Cin) {
            float s = 0.f;
            for (int j = 0; j < g.tpp; ++j) s += red[grp * g.tpp + j][tl];
            dx[pix * g.Cin + tl] = s;
        }
        __syncthreads();
    }
}

// dw[tap][ci][co] = sum_pix x[src(pix, tap)][ci] * dy[pix][co]; block partials -> part[block][9 * Cin * Cout]
__global__ __launch_bounds__(256) void thin_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ part, ThinGeom g) {
    __shared__ float4 red[256];
    const int grp = threadIdx.x / g.tpp, tl = threadIdx.x % g.tpp, co4 = tl * 4;
    const bool active = grp < g.ppb;
    float4 acc[9][THIN_MAXCIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < THIN_MAXCIN; ++ci) acc[t][ci] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    if (active) {
        for (int64_t pix = (int64_t)blockIdx.x * g.ppb + grp; pix < npix; pix += (int64_t)gridDim.x * g.ppb) {
            const int wo = (int)(pix % g.Wo);
            const int64_t t = pix / g.Wo;
            const int ho = (int)(t % g.Ho);
            const int64_t b = t / g.Ho;
            const float4 d = *reinterpret_cast<const float4*>(dy + pix * g.Cout + co4);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hs = thin_src(ho, kh, g.stride, g.pad, g.reflect, g.H);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ws = thin_src(wo, kw, g.stride, g.pad, g.reflect, g.W);
                    if (hs < 0 || ws < 0) continue;
                    const float* xp = x + ((b * g.H + hs) * g.W + ws) * g.Cin;
#pragma unroll
                    for (int ci = 0; ci < THIN_MAXCIN; ++ci) {
                        if (ci >= g.Cin) break;
                        const float xv = xp[ci];
                        float4& a = acc[kh * 3 + kw][ci];
                        a.x += xv * d.x;
                        a.y += xv * d.y;
                        a.z += xv * d.z;
                        a.w += xv * d.w;
                    }
                }
            }
        }
    }
    // combine the block's pixel groups (fixed order: deterministic), one (tap, ci) at a time
    float* out = part + (int64_t)blockIdx.x * 9 * g.Cin * g.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < THIN_MAXCIN; ++ci) {
            if (ci >= g.Cin) break;
            red[threadIdx.x] = active ? acc[t][ci] : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            if (threadIdx.x < g.tpp) {
                float4 s = red[threadIdx.x];
                for (int j = 1; j < g.ppb; ++j) {
                    const float4 v = red[j * g.tpp + threadIdx.x];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                *reinterpret_cast<float4*>(out + ((int64_t)(t * g.Cin + ci)) * g.Cout + threadIdx.x * 4) = s;
            }
            __syncthreads();
        }
}

// dw[i] = sum_b part[b][i]: 8 z-lanes x 4 loads in flight per output
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int n, int nblocks) {
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, tz = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int b = tz;
        for (; b + 24 < nblocks; b += 32) {
            s0 += part[(int64_t)b * n + i];
            s1 += part[(int64_t)(b + 8) * n + i];
            s2 += part[(int64_t)(b + 16) * n + i];
            s3 += part[(int64_t)(b + 24) * n + i];
        }
        for (; b < nblocks; b += 8) s0 += part[(int64_t)b * n + i];
    }
    red[tz][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tz == 0 && i < n) {
        float s = red[0][tx];
#pragma unroll
        for (int l = 1; l < 8; ++l) s += red[l][tx];
        dw[i] = s;
    }
}

}  // namespace bg

using namespace bg;

static bool thin_supported(const BgConvDesc* d) {
    return d && d->k == 3 && d->Cin >= 1 && d->Cin <= THIN_MAXCIN && d->Cout % 4 == 0 && d->Cout >= 16 &&
           d->Cout <= 1024 && (d->stride == 1 || d->stride == 2) && d->pad_lo >= 0 && d->pad_lo <= 1 && d->H >= 2 &&
           d->W >= 2;
}

static ThinGeom thin_geom(const BgConvDesc* d) {
    ThinGeom g;
    g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.Cout = d->Cout;
    g.stride = d->stride; g.pad = d->pad_lo; g.reflect = d->pad_mode == BG_PAD_REFLECT;
    g.tpp = d->Cout / 4;
    g.ppb = 256 / g.tpp;
    return g;
}

#define THIN_WGRAD_BLOCKS 512

extern "C" {

int bg_thinconv_supported(const BgConvDesc* d) { return thin_supported(d) ? 1 : 0; }

int bg_thinconv_fwd(const BgConvDesc* d, const float* x, const float* w, const float* bias, float* y, void* stream) {
    BG_REQUIRE(thin_supported(d) && x && w && y, "bg_thinconv_fwd: unsupported geometry");
    BG_REQUIRE((((uintptr_t)w | (uintptr_t)y) & 15) == 0, "bg_thinconv_fwd: pointers must be 16-byte aligned");
    ThinGeom g = thin_geom(d);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    int64_t blocks = (npix + g.ppb - 1) / g.ppb;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(thin_conv_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, w, bias, y, g);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_thinconv_dgrad(const BgConvDesc* d, const float* dy, const float* w, float* dx, void* stream) {
    BG_REQUIRE(thin_supported(d) && dy && w && dx, "bg_thinconv_dgrad: unsupported geometry");
    BG_REQUIRE((((uintptr_t)w | (uintptr_t)dy) & 15) == 0, "bg_thinconv_dgrad: pointers must be 16-byte aligned");
    ThinGeom g = thin_geom(d);
    const int64_t npix = (int64_t)g.N * g.H * g.W;
    int64_t blocks = (npix + g.ppb - 1) / g.ppb;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(thin_conv_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dy, w, dx, g);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

size_t bg_thinconv_wgrad_workspace_bytes(const BgConvDesc* d) {
    if (!thin_supported(d)) return 0;
    return (size_t)THIN_WGRAD_BLOCKS * 9 * d->Cin * d->Cout * sizeof(float);
}

int bg_thinconv_wgrad(const BgConvDesc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                      void* stream) {
    BG_REQUIRE(thin_supported(d) && x && dy && dw, "bg_thinconv_wgrad: unsupported geometry");
    BG_REQUIRE(ws && ws_bytes >= bg_thinconv_wgrad_workspace_bytes(d) && ((uintptr_t)ws & 15) == 0,
               "bg_thinconv_wgrad: workspace too small");
    BG_REQUIRE(((uintptr_t)dy & 15) == 0, "bg_thinconv_wgrad: pointers must be 16-byte aligned");
    ThinGeom g = thin_geom(d);
    float* part = reinterpret_cast<float*>(ws);
    const int n = 9 * g.Cin * g.Cout;
    hipLaunchKernelGGL(thin_conv_wgrad_kernel, dim3(THIN_WGRAD_BLOCKS), dim3(256), 0, as_stream(stream), x, dy, part, g);
    BG_LAUNCH_CHECK();
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, as_stream(stream), part, dw, n,
                       THIN_WGRAD_BLOCKS);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
