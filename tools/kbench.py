"""Per-layer kernel micro-benchmark through the C ABI.

    python tools/kbench.py [filter] [fp32 | bf16-staged | bf16] [c2 | c3] [batch multiplier]

fp32 / bf16-staged: fp32 tensors (BgConvDesc.compute); bf16: the bf16-resident kernels (bf16 tensors + packed weights).
Prints ms and TFLOP/s of the forward, input-gradient and weight-gradient launch of every layer shape of the chosen
BASELINE config (per-GPU batch).  Random (gaussian) data; timings with events on the launch stream.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import hip  # noqa: E402
from biggan_tensorflow_amd import functional as Fn  # noqa: E402
from biggan_tensorflow_amd.hip import act, f32, stream, lib, check  # noqa: E402

L = lib()
dev = "cuda"


def T(fn, flops, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9


def rnd(*s):
    return torch.randn(*s, device=dev)


def cases(cfg):
    if cfg == "c2":       # BigGAN-128 ch 64, batch 64 (D sees 2B = 128)
        return [("deconv", 64, 32, 256, 128, 4, 2), ("deconv", 64, 64, 128, 64, 4, 2), ("deconv", 64, 64, 128, 128, 3, 1),
                ("deconv", 64, 128, 64, 64, 3, 1), ("deconv", 64, 8, 1024, 512, 4, 2), ("deconv", 64, 4, 1024, 1024, 4, 2),
                ("conv", 128, 64, 64, 64, 3, 1), ("conv", 128, 64, 64, 128, 3, 2), ("conv", 128, 16, 256, 256, 3, 1),
                ("conv", 128, 4, 1024, 1024, 3, 1), ("conv", 128, 8, 512, 1024, 3, 2), ("conv", 64, 4, 1024, 1024, 3, 1),
                ("conv", 64, 128, 64, 3, 3, 1), ("conv", 128, 128, 3, 64, 3, 2)]
    # c3: BigGAN-128 ch 96, batch 32 per GPU (D sees 64)
    return [("deconv", 32, 4, 1536, 1536, 4, 2), ("deconv", 32, 8, 1536, 1536, 3, 1), ("deconv", 32, 8, 1536, 768, 4, 2),
            ("deconv", 32, 16, 768, 768, 3, 1), ("deconv", 32, 16, 768, 384, 4, 2), ("deconv", 32, 32, 384, 384, 3, 1),
            ("deconv", 32, 32, 384, 192, 4, 2), ("deconv", 32, 64, 192, 192, 3, 1), ("deconv", 32, 64, 192, 96, 4, 2),
            ("deconv", 32, 128, 96, 96, 3, 1),
            ("conv", 64, 64, 96, 96, 3, 1), ("conv", 64, 64, 96, 192, 3, 2), ("conv", 64, 32, 192, 192, 3, 1),
            ("conv", 64, 32, 192, 384, 3, 2), ("conv", 64, 16, 384, 384, 3, 1), ("conv", 64, 16, 384, 768, 3, 2),
            ("conv", 64, 8, 768, 768, 3, 1), ("conv", 64, 8, 768, 1536, 3, 2), ("conv", 64, 4, 1536, 1536, 3, 1),
            ("conv", 32, 64, 192, 24, 1, 1), ("conv", 32, 64, 192, 96, 1, 1), ("conv", 32, 64, 96, 192, 1, 1)]


flt = sys.argv[1] if len(sys.argv) > 1 else ""
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1       # 8: config 3 at its global batch 256 on one GPU
Fn.set_precision(mode)
res = mode == "bf16"
adt = torch.bfloat16 if res else torch.float32
X, Y = (hip.BF16, hip.BF16) if res else (hip.F32, hip.F32)
print("precision %s, config %s, batch x%d" % (mode, cfg, scale))
print("%-44s %18s %18s %18s" % ("case", "fwd ms/TF", "dgrad ms/TF", "wgrad ms/TF"))
tot = [0.0, 0.0, 0.0]
for kind, N, H, Cin, Cout, k, s in cases(cfg):
    N *= scale
    name = "%s N%d H%d %d->%d k%d s%d" % (kind, N, H, Cin, Cout, k, s)
    if flt and flt not in name:
        continue
    comp = Fn.Precision.compute
    if kind == "conv":
        Ho = H // s
        pad = 1 if k == 3 else 0
        d = hip.conv_desc(N, H, H, Cin, Ho, Ho, Cout, k, s, pad, hip.PAD_REFLECT, comp, X, Y, int(res))
        x, w, y = rnd(N, H, H, Cin).to(adt), rnd(k, k, Cin, Cout), torch.empty(N, Ho, Ho, Cout, device=dev, dtype=adt)
        if res and (Cin % 8 or Cout % 8):
            continue
        wp, wt = Fn.weight_packs(w) if res else (w, w)
        wf, wd = (wt, wp) if res else (w, w)
        fl = 2.0 * N * Ho * Ho * k * k * Cin * Cout
        wsf, nbf = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, dev)
        fw = lambda: check(L.bg_conv2d_fwd(d, act(x), act(wf), None, None, act(y), 0, f32(wsf), nbf, stream()))
        dx = torch.empty_like(x)
        wsd, nbd = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, d, dev)
        dg = lambda: check(L.bg_conv2d_dgrad(d, act(y), act(wd), None, act(dx), 0, f32(wsd), nbd, stream()))
        nb = L.bg_conv2d_wgrad_workspace_bytes(d)
        ws = hip.workspace(nb, dev)
        dw = torch.empty_like(w)
        wg = lambda: check(L.bg_conv2d_wgrad(d, act(x), act(y), f32(dw), f32(ws), nb, stream()))
        if not res and L.bg_rgbconv_supported(d):
            fw = lambda: check(L.bg_rgbconv_fwd(d, f32(x), f32(w), None, f32(y), 0, stream()))
            dg = lambda: check(L.bg_rgbconv_dgrad(d, f32(y), f32(w), f32(dx), 0, stream()))
            nb2 = L.bg_rgbconv_wgrad_workspace_bytes(d)
            ws2 = hip.workspace(nb2, dev)
            wg = lambda: check(L.bg_rgbconv_wgrad(d, f32(x), f32(y), f32(dw), f32(ws2), nb2, stream()))
    else:
        Ho = H * s
        d = hip.conv_desc(N, H, H, Cin, Ho, Ho, Cout, k, s, 1, hip.PAD_ZERO, comp, X, Y, int(res))
        x, w, y = rnd(N, H, H, Cin).to(adt), rnd(k, k, Cout, Cin), torch.empty(N, Ho, Ho, Cout, device=dev, dtype=adt)
        wp, wt = Fn.weight_packs(w) if res else (w, w)
        wf, wd = (wp, wt) if res else (w, w)
        fl = 2.0 * N * H * H * k * k * Cin * Cout
        wsf, nbf = hip.scratch(L.bg_deconv2d_fwd_workspace_bytes, d, dev)
        fw = lambda: check(L.bg_deconv2d_fwd(d, act(x), act(wf), None, None, act(y), 0, f32(wsf), nbf, stream()))
        dx = torch.empty_like(x)
        wsd, nbd = hip.scratch(L.bg_deconv2d_dgrad_workspace_bytes, d, dev)
        dg = lambda: check(L.bg_deconv2d_dgrad(d, act(y), act(wd), None, act(dx), 0, f32(wsd), nbd, stream()))
        nb = L.bg_deconv2d_wgrad_workspace_bytes(d)
        ws = hip.workspace(nb, dev)
        dw = torch.empty_like(w)
        wg = lambda: check(L.bg_deconv2d_wgrad(d, act(x), act(y), f32(dw), f32(ws), nb, stream()))
    r = [T(f, fl) for f in (fw, dg, wg)]
    for i in range(3):
        tot[i] += r[i][0]
    print("%-44s %8.3f %8.1f  %8.3f %8.1f  %8.3f %8.1f" % (name, r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1]),
          flush=True)
print("%-44s %8.3f %9s %8.3f %9s %8.3f" % ("sum of ms", tot[0], "", tot[1], "", tot[2]))
