// rgbconv.hip - direct kernels for k x k stride-1 convolutions with <= 4 output channels (the
// generator's G_logit layer, ops.py:49-113 via BigGAN.py:570: 128x128x64 -> 3).  A GEMM tile would
// spend 10x its work on padding N = 3 up to 32; these layers are HBM-bound (one pass over the wide
// activation), so they run on the vector ALUs with coalesced 256-byte channel rows:
// 16 lanes x float4 cover 64 channels of one pixel, a wave covers 4 pixels per step, partial sums
// are combined with wave64 shuffles.
#include "common.h"

namespace bg {

#define RGB_MAXCO 4
#define RGB_MAXTAPS 25

struct RgbGeom {
    int N, H, W, Cin, Cout, k, pad, reflect;
};

__device__ __forceinline__ int rgb_src(int o, int kk, int pad, int reflect, int n) {
    int s = o + kk - pad;
    if (reflect) {
        s = s < 0 ? -s : s;
        s = s >= n ? 2 * (n - 1) - s : s;
        return s;
    }
    return (s >= 0 && s < n) ? s : -1;
}

// sum over the 16 lanes that share a pixel (lane bits 0..3)
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// y[pix][co] = sum_tap sum_c x[src(pix,tap)][c] * w[tap][c][co]   (+ bias) (+ y)
__global__ __launch_bounds__(256) void rgb_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            RgbGeom g, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [taps][Cin][4] (co padded to 4)
    const int taps = g.k * g.k;
    // LDS image [tap][co (padded to 4)][Cin]: the 16 lanes of a pixel read 16 consecutive float4 (4 channels each)
    // of one output channel - conflict-free ds_read_b128 (a [tap][Cin][co] image puts lanes 64 bytes apart: 4-way)
    for (int i = threadIdx.x; i < taps * g.Cin * 4; i += 256) {
        const int c = i % g.Cin, tco = i / g.Cin, co = tco & 3, tp = tco >> 2;
        wl[i] = co < g.Cout ? w[((int64_t)tp * g.Cin + c) * g.Cout + co] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane >> 4, cq = (lane & 15) * 4;
    const int64_t npix = (int64_t)g.N * g.H * g.W;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t p0 = wave * 4; p0 < npix; p0 += nwaves * 4) {
        const int64_t pix = p0 + sub;
        const bool live = pix < npix;
        float acc[RGB_MAXCO] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const int wo = (int)(pix % g.W);
            const int64_t t = pix / g.W;
            const int ho = (int)(t % g.H);
            const int64_t b = t / g.H;
            for (int kh = 0; kh < g.k; ++kh) {
                const int hs = rgb_src(ho, kh, g.pad, g.reflect, g.H);
                if (hs < 0) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int ws = rgb_src(wo, kw, g.pad, g.reflect, g.W);
                    if (ws < 0) continue;
                    const float* xp = x + ((b * g.H + hs) * g.W + ws) * g.Cin;
                    const float* wt = wl + (kh * g.k + kw) * g.Cin * 4;
                    for (int c = cq; c < g.Cin; c += 64) {
                        const float4 xv = *reinterpret_cast<const float4*>(xp + c);
                        const float4 w0 = *reinterpret_cast<const float4*>(wt + c);
                        const float4 w1 = *reinterpret_cast<const float4*>(wt + g.Cin + c);
                        const float4 w2 = *reinterpret_cast<const float4*>(wt + 2 * g.Cin + c);
                        acc[0] += xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
                        acc[1] += xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
                        acc[2] += xv.x * w2.x + xv.y * w2.y + xv.z * w2.z + xv.w * w2.w;
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < RGB_MAXCO; ++co) acc[co] = sum16(acc[co]);
        if (live && (lane & 15) < g.Cout) {
            const int co = lane & 15;
            float v = co == 0 ? acc[0] : co == 1 ? acc[1] : co == 2 ? acc[2] : acc[3];
            if (bias) v += bias[co];
            float* o = y + pix * g.Cout + co;
            if (accumulate) v += *o;
            *o = v;
        }
    }
}

// dx[pix][c] = sum_tap sum_co dy[dst(pix,tap)][co] * w[tap][c][co], with the mirrored contributions
// of reflect padding (stride 1): output positions whose window read pixel `pix` through tap (kh,kw)
__device__ __forceinline__ int rgb_dst(int i, int kk, int pad, int n) { return i + pad - kk; }   // ho with src == i

__global__ __launch_bounds__(256) void rgb_conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, RgbGeom g, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [taps][Cin][4]
    const int taps = g.k * g.k;
    // LDS image [tap][co (padded to 4)][Cin]: the 16 lanes of a pixel read 16 consecutive float4 (4 channels each)
    // of one output channel - conflict-free ds_read_b128 (a [tap][Cin][co] image puts lanes 64 bytes apart: 4-way)
    for (int i = threadIdx.x; i < taps * g.Cin * 4; i += 256) {
        const int c = i % g.Cin, tco = i / g.Cin, co = tco & 3, tp = tco >> 2;
        wl[i] = co < g.Cout ? w[((int64_t)tp * g.Cin + c) * g.Cout + co] : 0.f;
    }
    __syncthreads();
    const int CQ = g.Cin / 4;
    const int64_t total = (int64_t)g.N * g.H * g.W * CQ;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % CQ) * 4;
        const int64_t pix = idx / CQ;
        const int wi = (int)(pix % g.W);
        const int64_t t = pix / g.W;
        const int hi = (int)(t % g.H);
        const int64_t b = t / g.H;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kh = 0; kh < g.k; ++kh) {
            // output rows ho whose tap kh reads input row hi: direct and (reflect) mirrored
            int hos[2];
            hos[0] = hi + g.pad - kh;
            hos[1] = -1;
            if (g.reflect) {
                if (hi >= 1 && hi <= g.pad) hos[1] = g.pad - hi - kh;
                else if (hi >= g.H - 1 - g.pad && hi <= g.H - 2) hos[1] = 2 * (g.H - 1) + g.pad - hi - kh;
            }
            for (int a = 0; a < 2; ++a) {
                const int ho = hos[a];
                if (ho < 0 || ho >= g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    int wos[2];
                    wos[0] = wi + g.pad - kw;
                    wos[1] = -1;
                    if (g.reflect) {
                        if (wi >= 1 && wi <= g.pad) wos[1] = g.pad - wi - kw;
                        else if (wi >= g.W - 1 - g.pad && wi <= g.W - 2) wos[1] = 2 * (g.W - 1) + g.pad - wi - kw;
                    }
                    const float* wt = wl + (kh * g.k + kw) * g.Cin * 4 + c;       // [tap][co][Cin]
                    const float4 w0 = *reinterpret_cast<const float4*>(wt);            // co 0, channels c..c+3
                    const float4 w1 = *reinterpret_cast<const float4*>(wt + g.Cin);
                    const float4 w2 = *reinterpret_cast<const float4*>(wt + 2 * g.Cin);
                    for (int e = 0; e < 2; ++e) {
                        const int wo = wos[e];
                        if (wo < 0 || wo >= g.W) continue;
                        const float* d = dy + ((b * g.H + ho) * g.W + wo) * g.Cout;
                        float dv[RGB_MAXCO] = {0.f, 0.f, 0.f, 0.f};
                        for (int co = 0; co < g.Cout; ++co) dv[co] = d[co];
                        acc.x += dv[0] * w0.x + dv[1] * w1.x + dv[2] * w2.x;
                        acc.y += dv[0] * w0.y + dv[1] * w1.y + dv[2] * w2.y;
                        acc.z += dv[0] * w0.z + dv[1] * w1.z + dv[2] * w2.z;
                        acc.w += dv[0] * w0.w + dv[1] * w1.w + dv[2] * w2.w;
                    }
                }
            }
        }
        float4* o = reinterpret_cast<float4*>(dx + pix * g.Cin + c);
        if (accumulate) {
            const float4 old = *o;
            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
        }
        *o = acc;
    }
}

// dw[tap][c][co] = sum_pix x[src(pix,tap)][c] * dy[pix][co]; each block writes its partial
// [taps][Cin][Cout] slab, rgb_wgrad_reduce sums the slabs (deterministic).
template <int TAPS>
__global__ __launch_bounds__(256) void rgb_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ part, RgbGeom g, int c0) {
    __shared__ float red[4][TAPS * RGB_MAXCO * 64];    // per wave: [tap][co][64 channels]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane >> 4, cq = c0 + (lane & 15) * 4;
    const int64_t npix = (int64_t)g.N * g.H * g.W;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    float acc[TAPS][3][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][co][j] = 0.f;
    const bool chan_ok = cq < g.Cin;
    for (int64_t p0 = wave * 4; p0 < npix; p0 += nwaves * 4) {
        const int64_t pix = p0 + sub;
        if (pix >= npix || !chan_ok) continue;
        const int wo = (int)(pix % g.W);
        const int64_t t = pix / g.W;
        const int ho = (int)(t % g.H);
        const int64_t b = t / g.H;
        const float* d = dy + pix * g.Cout;
        float dv[3] = {0.f, 0.f, 0.f};
        for (int co = 0; co < g.Cout && co < 3; ++co) dv[co] = d[co];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const int kh = tp / g.k, kw = tp % g.k;
            const int hs = rgb_src(ho, kh, g.pad, g.reflect, g.H);
            const int ws = rgb_src(wo, kw, g.pad, g.reflect, g.W);
            if (hs < 0 || ws < 0) continue;
            const float4 xv = *reinterpret_cast<const float4*>(x + ((b * g.H + hs) * g.W + ws) * g.Cin + cq);
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                acc[tp][co][0] += xv.x * dv[co];
                acc[tp][co][1] += xv.y * dv[co];
                acc[tp][co][2] += xv.z * dv[co];
                acc[tp][co][3] += xv.w * dv[co];
            }
        }
    }
    // combine the 4 pixel groups of the wave (lane bits 4,5), then the 4 waves through LDS
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[tp][co][j];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (sub == 0) red[wv][(tp * RGB_MAXCO + co) * 64 + (lane & 15) * 4 + j] = v;
            }
    __syncthreads();
    float* out = part + (int64_t)blockIdx.x * TAPS * g.Cin * g.Cout;
    for (int i = threadIdx.x; i < TAPS * 3 * 64; i += 256) {
        const int ch = i & 63, co = (i >> 6) % 3, tp = i / 192;
        const int c = c0 + ch;
        if (c < g.Cin && co < g.Cout) {
            const int r = (tp * RGB_MAXCO + co) * 64 + ch;
            out[((int64_t)tp * g.Cin + c) * g.Cout + co] = red[0][r] + red[1][r] + red[2][r] + red[3][r];
        }
    }
}

// sum of the per-block partial slabs: 8 z-lanes x 4 loads in flight per output (a single thread walking 512
// slabs is one dependent memory round trip per slab: 119 us for 5184 outputs)
__global__ __launch_bounds__(256) void rgb_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
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

static bool rgb_supported(const BgConvDesc* d) {
    return d && d->Cout <= 3 && d->stride == 1 && d->k * d->k <= RGB_MAXTAPS && d->Cin % 4 == 0 &&
           d->Ho == d->H && d->Wo == d->W && (size_t)d->k * d->k * d->Cin * 4 * sizeof(float) <= 60 * 1024;
}

static RgbGeom rgb_geom(const BgConvDesc* d) {
    RgbGeom g;
    g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.k = d->k; g.pad = d->pad_lo;
    g.reflect = d->pad_mode == BG_PAD_REFLECT;
    return g;
}

#define RGB_WGRAD_BLOCKS 512

extern "C" {

int bg_rgbconv_supported(const BgConvDesc* d) { return rgb_supported(d) ? 1 : 0; }

int bg_rgbconv_fwd(const BgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int accumulate,
                   void* stream) {
    BG_REQUIRE(rgb_supported(d) && x && w && y, "bg_rgbconv_fwd: unsupported geometry");
    RgbGeom g = rgb_geom(d);
    const int64_t npix = (int64_t)g.N * g.H * g.W;
    int blocks = (int)((npix / 4 + 3) / 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)g.k * g.k * g.Cin * 4 * sizeof(float);
    hipLaunchKernelGGL(rgb_conv_fwd_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), x, w, bias, y, g,
                       accumulate);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_rgbconv_dgrad(const BgConvDesc* d, const float* dy, const float* w, float* dx, int accumulate, void* stream) {
    BG_REQUIRE(rgb_supported(d) && dy && w && dx, "bg_rgbconv_dgrad: unsupported geometry");
    RgbGeom g = rgb_geom(d);
    const int64_t total = (int64_t)g.N * g.H * g.W * (g.Cin / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    const size_t lds = (size_t)g.k * g.k * g.Cin * 4 * sizeof(float);
    hipLaunchKernelGGL(rgb_conv_dgrad_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), dy, w, dx, g,
                       accumulate);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

size_t bg_rgbconv_wgrad_workspace_bytes(const BgConvDesc* d) {
    if (!rgb_supported(d)) return 0;
    return (size_t)RGB_WGRAD_BLOCKS * d->k * d->k * d->Cin * d->Cout * sizeof(float);
}

int bg_rgbconv_wgrad(const BgConvDesc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                     void* stream) {
    BG_REQUIRE(rgb_supported(d) && x && dy && dw, "bg_rgbconv_wgrad: unsupported geometry");
    BG_REQUIRE(d->k == 3 || d->k == 1, "bg_rgbconv_wgrad: k must be 1 or 3");
    BG_REQUIRE(ws && ws_bytes >= bg_rgbconv_wgrad_workspace_bytes(d), "bg_rgbconv_wgrad: workspace too small");
    RgbGeom g = rgb_geom(d);
    float* part = reinterpret_cast<float*>(ws);
    const int n = g.k * g.k * g.Cin * g.Cout;
    for (int c0 = 0; c0 < g.Cin; c0 += 64) {
        if (g.k == 3)
            hipLaunchKernelGGL((rgb_conv_wgrad_kernel<9>), dim3(RGB_WGRAD_BLOCKS), dim3(256), 0, as_stream(stream), x,
                               dy, part, g, c0);
        else
            hipLaunchKernelGGL((rgb_conv_wgrad_kernel<1>), dim3(RGB_WGRAD_BLOCKS), dim3(256), 0, as_stream(stream), x,
                               dy, part, g, c0);
        BG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(rgb_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, as_stream(stream), part, dw, n,
                       RGB_WGRAD_BLOCKS);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
