// igemm_dev.h - device-side index math shared by the fp32 and bf16 implicit-GEMM kernels:
// row -> pixel decomposition, tap -> source-pixel maps (conv / transposed conv, reflect and zero
// padding, mirrored sources for the gradient of reflect padding), XCD-aware tile renumbering.
#pragma once
#include "common.h"
#include "igemm.h"

namespace bg {

// ------------------------------------------------------------------------------------------
// gather index math (shared by NN and TN kernels)
// ------------------------------------------------------------------------------------------
struct RowPos {
    int b, ho, wo;
    bool valid;
};

template <int MODE>
__device__ __forceinline__ RowPos decompose_row(const Gather& g, int m, int M, int ph, int pw) {
    RowPos r;
    r.valid = m < M;
    if (MODE == GATHER_PLAIN) {
        r.b = m;
        r.ho = 0;
        r.wo = 0;
        return r;
    }
    int wq = m % g.Wq;
    int t = m / g.Wq;
    int hq = t % g.Hq;
    r.b = t / g.Hq;
    r.ho = hq * g.pstep + ph;
    r.wo = wq * g.pstep + pw;
    return r;
}

// one axis of a CONV-mode gather: source index or -1
__device__ __forceinline__ int conv_src(int o, int kk, int stride, int pad, int reflect, int n) {
    int s = o * stride + kk - pad;
    if (reflect) {
        s = s < 0 ? -s : s;
        s = s >= n ? 2 * (n - 1) - s : s;
        return s;
    }
    return (s >= 0 && s < n) ? s : -1;
}

// one axis of a TCONV-mode gather: numerator hn = o + pad - kk must be a non-negative multiple of stride
__device__ __forceinline__ int tconv_src_from_num(int hn, int stride, int n) {
    if (hn < 0) return -1;
    int s = stride == 1 ? hn : (hn >> 1);
    if (stride != 1 && (hn & 1)) return -1;
    return s < n ? s : -1;
}

// mirrored source of a TCONV gather with reflect padding (gradient of tf.pad(REFLECT) folded in):
// the padded positions that alias output pixel o are  pad - o  (1 <= o <= pad)  and
// 2*(n_out-1) + pad - o  (n_out-1-pad <= o <= n_out-2).
__device__ __forceinline__ int tconv_mirror_src(int o, int kk, int stride, int pad, int n_out, int n_src) {
    int hn;
    if (o >= 1 && o <= pad)
        hn = pad - o - kk;
    else if (o >= n_out - 1 - pad && o <= n_out - 2)
        hn = 2 * (n_out - 1) + pad - o - kk;
    else
        return -1;
    return tconv_src_from_num(hn, stride, n_src);
}

// up to 4 source offsets (element offsets into the source tensor, -1 = none) for row r and tap (kh,kw)
template <int MODE, bool MIRROR>
__device__ __forceinline__ void tap_sources(const Gather& g, const RowPos& r, int kh, int kw,
                                            int64_t (&off)[MIRROR ? 4 : 1]) {
#pragma unroll
    for (int i = 0; i < (MIRROR ? 4 : 1); ++i) off[i] = -1;
    if (!r.valid) return;
    if (MODE == GATHER_PLAIN) {
        off[0] = (int64_t)r.b * g.ld;
        return;
    }
    int h0, w0, h1 = -1, w1 = -1;
    if (MODE == GATHER_CONV) {
        h0 = conv_src(r.ho, kh, g.stride, g.pad, g.reflect, g.Hs);
        w0 = conv_src(r.wo, kw, g.stride, g.pad, g.reflect, g.Ws);
    } else {
        h0 = tconv_src_from_num(r.ho + g.pad - kh, g.stride, g.Hs);
        w0 = tconv_src_from_num(r.wo + g.pad - kw, g.stride, g.Ws);
        if (MIRROR) {
            h1 = tconv_mirror_src(r.ho, kh, g.stride, g.pad, g.Ho, g.Hs);
            w1 = tconv_mirror_src(r.wo, kw, g.stride, g.pad, g.Wo, g.Ws);
        }
    }
    const int64_t base = (int64_t)r.b * g.Hs;
    if (h0 >= 0 && w0 >= 0) off[0] = ((base + h0) * g.Ws + w0) * g.ld;
    if (MIRROR) {
        if (h0 >= 0 && w1 >= 0) off[1] = ((base + h0) * g.Ws + w1) * g.ld;
        if (h1 >= 0 && w0 >= 0) off[2] = ((base + h1) * g.Ws + w0) * g.ld;
        if (h1 >= 0 && w1 >= 0) off[3] = ((base + h1) * g.Ws + w1) * g.ld;
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <bool VEC>
__device__ __forceinline__ float4 load_chan4(const float* base, int64_t off, int c, int C) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (off < 0) return v;
    const float* p = base + off + c;
    if (VEC) {
        if (c < C) v = ld4(p);
    } else {
        if (c + 0 < C) v.x = p[0];
        if (c + 1 < C) v.y = p[1];
        if (c + 2 < C) v.z = p[2];
        if (c + 3 < C) v.w = p[3];
    }
    return v;
}

__device__ __forceinline__ void add4(float4& a, const float4& b) {
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    a.w += b.w;
}

// XCD-aware renumbering: hardware deals consecutive block ids round-robin over the 8 XCDs; give each
// XCD a contiguous range of logical tiles so neighbouring tiles share one L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

}  // namespace bg
