"""Kernel micro-benchmark through the C ABI: python scratch/kbench.py [filter]"""
import sys, torch, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd
from biggan_tensorflow_amd import hip
from biggan_tensorflow_amd.hip import f32, stream, lib, check
L = lib()
dev = "cuda"
def T(fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9
def rnd(*s): return torch.randn(*s, device=dev)
CASES = [
 # kind, N, H, Cin, Cout, k, s
 ("deconv", 64, 32, 256, 128, 4, 2), ("deconv", 64, 64, 128, 64, 4, 2), ("deconv", 64, 64, 128, 128, 3, 1),
 ("deconv", 64, 128, 64, 64, 3, 1), ("deconv", 64, 8, 1024, 512, 4, 2), ("deconv", 64, 4, 1024, 1024, 4, 2),
 ("conv", 128, 64, 64, 64, 3, 1), ("conv", 128, 64, 64, 128, 3, 2), ("conv", 128, 16, 256, 256, 3, 1),
 ("conv", 128, 4, 1024, 1024, 3, 1), ("conv", 128, 8, 512, 1024, 3, 2), ("conv", 64, 4, 1024, 1024, 3, 1),
 ("conv", 64, 128, 64, 3, 3, 1), ("conv", 128, 128, 3, 64, 3, 2),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
if len(sys.argv) > 2 and sys.argv[2].startswith("bf16"): L.bg_set_gemm_compute(int(sys.argv[2][4:] or 1)); print("bf16 compute mode", L.bg_get_gemm_compute())
print("%-44s %18s %18s %18s" % ("case", "fwd ms/TF", "dgrad ms/TF", "wgrad ms/TF"))
for kind, N, H, Cin, Cout, k, s in CASES:
    name = "%s N%d H%d %d->%d k%d s%d" % (kind, N, H, Cin, Cout, k, s)
    if flt and flt not in name: continue
    if kind == "conv":
        Ho = H // s
        d = hip.conv_desc(N, H, H, Cin, Ho, Ho, Cout, k, s, 1, hip.PAD_REFLECT)
        x, w, y = rnd(N, H, H, Cin), rnd(k, k, Cin, Cout), torch.empty(N, Ho, Ho, Cout, device=dev)
        fl = 2.0 * N * Ho * Ho * k * k * Cin * Cout
        wsf, nbf = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, dev)
        fw = lambda: check(L.bg_conv2d_fwd(d, f32(x), f32(w), None, None, f32(y), 0, f32(wsf), nbf, stream()))
        dx = torch.empty_like(x)
        wsd, nbd = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, d, dev)
        dg = lambda: check(L.bg_conv2d_dgrad(d, f32(y), f32(w), None, f32(dx), 0, f32(wsd), nbd, stream()))
        nb = L.bg_conv2d_wgrad_workspace_bytes(d); ws = hip.workspace(nb, dev); dw = torch.empty_like(w)
        wg = lambda: check(L.bg_conv2d_wgrad(d, f32(x), f32(y), f32(dw), f32(ws), nb, stream()))
        if L.bg_rgbconv_supported(d):
            fw = lambda: check(L.bg_rgbconv_fwd(d, f32(x), f32(w), None, f32(y), 0, stream()))
            dg = lambda: check(L.bg_rgbconv_dgrad(d, f32(y), f32(w), f32(dx), 0, stream()))
            nb2 = L.bg_rgbconv_wgrad_workspace_bytes(d); ws2 = hip.workspace(nb2, dev)
            wg = lambda: check(L.bg_rgbconv_wgrad(d, f32(x), f32(y), f32(dw), f32(ws2), nb2, stream()))
    else:
        Ho = H * s
        d = hip.conv_desc(N, H, H, Cin, Ho, Ho, Cout, k, s, 1, hip.PAD_ZERO)
        x, w, y = rnd(N, H, H, Cin), rnd(k, k, Cout, Cin), torch.empty(N, Ho, Ho, Cout, device=dev)
        fl = 2.0 * N * H * H * k * k * Cin * Cout
        wsf, nbf = hip.scratch(L.bg_deconv2d_fwd_workspace_bytes, d, dev)
        fw = lambda: check(L.bg_deconv2d_fwd(d, f32(x), f32(w), None, None, f32(y), 0, f32(wsf), nbf, stream()))
        dx = torch.empty_like(x)
        wsd, nbd = hip.scratch(L.bg_deconv2d_dgrad_workspace_bytes, d, dev)
        dg = lambda: check(L.bg_deconv2d_dgrad(d, f32(y), f32(w), None, f32(dx), 0, f32(wsd), nbd, stream()))
        nb = L.bg_deconv2d_wgrad_workspace_bytes(d); ws = hip.workspace(nb, dev); dw = torch.empty_like(w)
        wg = lambda: check(L.bg_deconv2d_wgrad(d, f32(x), f32(y), f32(dw), f32(ws), nb, stream()))
    r = [T(f, fl) for f in (fw, dg, wg)]
    print("%-44s %8.3f %8.1f  %8.3f %8.1f  %8.3f %8.1f" % (name, r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1]), flush=True)
