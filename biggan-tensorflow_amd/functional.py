"""torch.autograd.Function wrappers around the C-ABI kernels (``hip.py``).

PyTorch provides tensor storage, the stream and the backward-graph plumbing; every arithmetic
operation is a libbiggan_hip.so launch on the current stream.  Gradients of store variables are
written straight into their slice of the flat gradient arena (``scope.Arena``) instead of
``Tensor.grad``: the first write of a step overwrites, later ones accumulate.
"""
import math
import os

import torch
from torch.autograd import Function

from . import hip
from .hip import f32, i32, act, dt, stream, check, lib, workspace

BF16 = torch.bfloat16


class Precision:
    """Arithmetic / storage policy, applied per call through the descriptors (no library-wide state).

    compute   arithmetic of conv / transposed-conv launches on fp32 tensors: hip.COMPUTE_F32 (the reference's
              precision, ops.py:14) or hip.COMPUTE_BF16 (bf16 MFMA, operands rounded while staged)
    gemm_bf16 plain GEMMs with M, N, K >= 128 (the regulariser's Gram matrices) on the bf16 MFMA as well
    resident  BASELINE configs 3-5: activations between ops are bf16 tensors in HBM and every spectrally
              normalised conv kernel carries two packed bf16 copies (csrc/igemm16.hip)"""
    compute = hip.COMPUTE_F32
    gemm_bf16 = False
    resident = False
    name = "fp32"


def set_precision(mode):
    """'fp32' (default) | 'bf16-staged' (fp32 tensors, bf16 MFMA) | 'bf16' (bf16-resident activations)."""
    if mode in (None, "", "fp32", "f32"):
        Precision.compute, Precision.gemm_bf16, Precision.resident, Precision.name = hip.COMPUTE_F32, False, False, "fp32"
    elif mode == "bf16-staged":
        Precision.compute, Precision.gemm_bf16, Precision.resident, Precision.name = hip.COMPUTE_BF16, True, False, mode
    elif mode == "bf16":
        Precision.compute, Precision.gemm_bf16, Precision.resident, Precision.name = hip.COMPUTE_BF16, True, True, mode
    else:
        raise ValueError("precision %r: expected fp32, bf16-staged or bf16" % (mode,))


class precision_scope:
    """Run a region in another precision mode (the gradient penalty of a bf16-resident model: its inputs-only
    backward and its forward-mode pass are fp32-tensor kernels, so they run as 'bf16-staged' - fp32 tensors, bf16 MFMA
    operands, the same weights).  Functions record what they need in ctx at forward time, so the backward of the
    region may run after the scope has ended."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = Precision.name
        set_precision(self.mode)

    def __exit__(self, *exc):
        set_precision(self.prev)


def cast(x, dtype):
    """Element-type conversion (RNE) by a HIP kernel; returns x itself when nothing changes."""
    if x.dtype == dtype:
        return x
    x = _c(x)
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().bg_cast(act(x), dt(x), act(y), dt(y), x.numel(), stream()))
    return y


def weight_packs(w):
    """(pack_p, pack_t): the bf16 K-contiguous copies of a conv / transposed-conv kernel [k,k,A,B]: pack_p keeps the
    variable's order [k*k][A][B], pack_t is [k*k][B][A].  Spectrally normalised kernels get them from the
    multi-tensor power iteration (SnBatch); anything else is packed per call."""
    pp = getattr(w, "bg_pack_p", None)
    if pp is not None:
        return pp, w.bg_pack_t
    w = _c(w)
    k2, A, B = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
    pp = torch.empty((k2, A, B), dtype=BF16, device=w.device)
    pt = torch.empty((k2, B, A), dtype=BF16, device=w.device)
    check(lib().bg_weight_pack(f32(w), k2, A, B, act(pp), act(pt), stream()))
    return pp, pt


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------
def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def axpby(x, a, y, b):
    """y = a*x + b*y (in place on y)."""
    check(lib().bg_axpby(f32(x), float(a), f32(y), float(b), x.numel(), stream()))
    return y


def add(a, b, out=None):
    """a + b on tensors of one element type (fp32 or bf16)."""
    if b.dtype != a.dtype:
        b = cast(b, a.dtype)
    y = torch.empty_like(a) if out is None else out
    if a.dtype == torch.float32:
        check(lib().bg_add(f32(a), f32(b), f32(y), a.numel(), stream()))
    else:
        check(lib().bg_lincomb_t(act(a), None, 1.0, act(b), 1.0, act(y), dt(a), a.numel(), stream()))
    return y


def is_variable(t):
    return getattr(t, "bg_name", None) is not None


class _Mode:
    # True while the input gradient of the gradient penalty is taken (BigGAN.py:731): backward passes then
    # produce d/d(activation) only; nothing is written into the parameters' gradient slots
    inputs_only = False


class ForkState:
    """Shared by the outputs of one ForkFn (ops._fork): the branch whose backward runs first writes its input gradient
    into a fresh tensor and leaves it here; a later branch whose kernel can add while it writes (conv / transposed-conv
    input gradients: ``accumulate``; batch-norm and PReLU backward: ``dx_add``) accumulates into that same tensor and
    hands it back, so ForkFn.backward receives one buffer twice and no separate summing pass runs."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


def _fork_target(fk, like):
    """(dx, accumulate?) for a branch's input gradient of the forked tensor ``like``.  (BG_FUSE_FORK=0: A/B switch, every
    branch writes its own tensor and ForkFn.backward sums them with a separate kernel.)"""
    prev = fk.buf if (fk is not None and os.environ.get("BG_FUSE_FORK", "1") != "0") else None
    if prev is not None and prev.dtype == like.dtype and prev.shape == like.shape and prev.is_contiguous():
        return prev, True
    return torch.empty_like(like), False


def _fork_done(fk, dx):
    if fk is not None and dx is not None:
        fk.buf = dx


class KinkProbe:
    """Test instrumentation (tests/test_gpu_step.py): when ``sites`` is a list, every activation launch appends what is
    needed to reconstruct its pre-activation (the tensor itself for a stand-alone PReLU, the batch-norm operands for
    the fused apply + PReLU kernel), in call order."""
    sites = None


class inputs_only_backward:
    def __enter__(self):
        self.prev, _Mode.inputs_only = _Mode.inputs_only, True

    def __exit__(self, *exc):
        _Mode.inputs_only = self.prev


def emit_grad(var, producer):
    """Deliver d(loss)/d(var).  ``producer(out)`` must WRITE the gradient into ``out``."""
    if _Mode.inputs_only:
        return
    slot = getattr(var, "bg_grad", None)
    if slot is None:                       # stand-alone variable (not packed into an arena)
        g = torch.empty(var.shape, dtype=torch.float32, device=var.device)
        producer(g)
        if var.grad is None:
            var.grad = g
        else:
            axpby(g, 1.0, var.grad, 1.0)
        return
    if not var.bg_touched:
        producer(slot)
        var.bg_touched = True
    else:
        tmp = torch.empty_like(slot)
        producer(tmp)
        axpby(tmp, 1.0, slot, 1.0)


def param_grad(t, needed, producer):
    """Gradient for an input that may be a store variable (side-effect delivery, returns None)
    or an ordinary autograd tensor (returned)."""
    if not needed or _Mode.inputs_only:
        return None
    if is_variable(t):
        emit_grad(t, producer)
        return None
    g = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    producer(g)
    return g


def grad_slot(var):
    """(tensor, accumulate?) where a kernel that can add while it writes delivers d(loss)/d(var) itself (the emit_grad
    protocol without the temporary): the variable's arena slot, overwritten by the first writer of a step."""
    slot = getattr(var, "bg_grad", None)
    if slot is None:
        if var.grad is None:
            var.grad = torch.empty(var.shape, dtype=torch.float32, device=var.device)
            return var.grad, False
        return var.grad, True
    acc = bool(var.bg_touched)
    var.bg_touched = True
    return slot, acc


class ZeroPool:
    """The zero-initialised accumulators of one run (batch-norm sums, loss sums, regulariser terms: ~60 small tensors) as
    slices of ONE buffer cleared by ONE fill at the start of the run, instead of a torch.zeros launch each.

    Lifetime rule: a slice handed out in run k may still be READ during run k + 1 - the losses of the D op are returned to
    the caller and read after the G op has begun; under data parallelism a deferred exchange finishes inside the next
    run.  So there are TWO buffers used alternately: ``begin_run`` clears and hands out the buffer last used TWO runs
    ago; the fill is an ordinary kernel on the compute stream, i.e. ordered behind every kernel and every waited-for
    collective that touched the buffer.  Requests beyond the capacity fall back to torch.zeros and enlarge the pool for
    the following runs (retired buffers stay referenced: a captured HIP graph may still address them)."""

    def __init__(self, device, nbytes=1 << 19):
        # (the INDEXED device, as tensors report it: torch.device("cuda") != torch.device("cuda:0"), and a pool that never
        #  matched its callers' x.device silently handed every request to torch.zeros - bench.py builds its model on "cuda")
        self.device = torch.empty(0, device=device).device
        self.cap = int(nbytes)
        self.bufs = [None, None]
        self.used = [0, 0]          # bytes handed out from each buffer in its last run (= what the next clear covers)
        self.cur = 0
        self.off = 0
        self.want = 0               # bytes the largest run asked for
        self.retired = []
        self.active = False

    def begin_run(self):
        if self.want > self.cap:
            self.retired.extend(b for b in self.bufs if b is not None)
            self.cap = 2 * self.want
            self.bufs = [None, None]
            self.used = [0, 0]
        self.cur ^= 1
        i = self.cur
        if self.bufs[i] is None:
            self.bufs[i] = torch.zeros(self.cap, dtype=torch.uint8, device=self.device)
        elif self.used[i] > 0:
            self.bufs[i].narrow(0, 0, self.used[i]).zero_()
        self.used[i] = 0
        self.off = 0
        self.active = True

    def zeros(self, n, dtype):
        item = torch.empty(0, dtype=dtype).element_size()
        nbytes = (n * item + 15) // 16 * 16
        end = self.off + nbytes
        self.want = max(self.want, end)
        if not self.active or end > self.cap:
            self.off = end if self.active else self.off
            return torch.zeros(n, dtype=dtype, device=self.device)
        t = self.bufs[self.cur].narrow(0, self.off, n * item).view(dtype)
        self.off = end
        self.used[self.cur] = end
        return t


_pool = None
_run_stamp = [object()]


def new_run_stamp():
    """A fresh identity for the run that begins (ops.begin_run)."""
    _run_stamp[0] = object()
    return _run_stamp[0]


def current_run_stamp():
    return _run_stamp[0]



def set_zero_pool(pool):
    """The pool of the model whose run begins (ops.begin_run); None: every request is a torch.zeros."""
    global _pool
    _pool = pool


def zeros(n, dtype, device):
    """A zeroed 1-D tensor of ``n`` elements for an accumulator that lives for one run."""
    if _pool is not None and _pool.device == device and os.environ.get("BG_ZERO_POOL", "1") != "0":
        return _pool.zeros(n, dtype)
    return torch.zeros(n, dtype=dtype, device=device)


def _bias_grad(dy2d, out):
    if dy2d.dtype == torch.float32:
        check(lib().bg_bias_grad(f32(dy2d), f32(out), dy2d.shape[0], dy2d.shape[1], stream()))
    else:
        check(lib().bg_bias_grad_t(act(dy2d), dt(dy2d), f32(out), dy2d.shape[0], dy2d.shape[1], stream()))


# ------------------------------------------------------------------------------------------
# conv / transposed conv
# ------------------------------------------------------------------------------------------
def _resident_ok(x, cin, cout):
    return x.dtype == BF16 and cin % 8 == 0 and cout % 8 == 0


def pad_channels(x, C, dtype, axis=-1, mode=hip.PAD_ZERO_FILL):
    """Copy of ``x`` with dimension ``axis`` widened or narrowed to C entries, converted to ``dtype``
    (``mode``: zero fill / split into bf16 value + rounding residual / duplicate / fold two halves: bg_pad_channels)."""
    x = _c(x)
    axis = axis % x.dim()
    shape = list(x.shape)
    outer = int(math.prod(shape[:axis])) if axis else 1
    inner = int(math.prod(shape[axis + 1:])) if axis + 1 < len(shape) else 1
    Cs = shape[axis]
    shape[axis] = C
    y = torch.empty(shape, dtype=dtype, device=x.device)
    check(lib().bg_pad_channels(act(x), dt(x), act(y), dt(y), outer, Cs, C, inner, mode, stream()))
    return y


def _thin_plan(x, cin, cout):
    """bf16-resident mode, the 3-channel image layers (D's first block: Cin = 3; G_logit: Cout = 3): run on the
    bf16-resident GEMM kernels with the thin side zero-padded to 8 channels.  The fp32-tensor fallbacks they replace
    cost 16.8 of 137 ms per config-3 iteration at batch 256 (r02 kernel stats) for 0.3 % of the FLOPs."""
    if not Precision.resident or x.dtype not in (BF16, torch.float32) or os.environ.get("BG_IMAGE_LAYERS", "") == "fp32":
        return None                      # (BG_IMAGE_LAYERS=fp32: A/B switch, the fp32-tensor kernels of round 1)
    # (the thin side and its rounding residual share the 8 channels: at most 4; a 5 ... 7-channel side - the attention
    #  projections of a ch = 40 / 48 / 56 model - takes the fp32-tensor kernels like any other odd channel count)
    if cin <= 4 and cout % 8 == 0:
        return "in"
    if cout <= 4 and cin % 8 == 0 and x.dtype == BF16:
        return "out"
    return None


class Conv2dFn(Function):
    """tf.pad(REFLECT)+tf.nn.conv2d(VALID)+bias_add (ops.py:82,94-98) / zero 'SAME'.

    fp32 x: the fp32-tensor kernels (fp32 or bf16 MFMA per Precision.compute), fp32 y.
    bf16 x: the bf16-resident kernels (packed weights, csrc/igemm16.hip); y is bf16 unless ``out_dtype`` says fp32."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad_lo, Ho, Wo, pad_mode, out_dtype=None, accumulate_into=None):
        ctx.fork = getattr(x, "bg_fork", None)
        x = _c(x)
        N, H, W_, Cin = x.shape
        k, _, cin2, Cout = w.shape
        assert cin2 == Cin, (x.shape, w.shape)
        L = lib()
        ctx.resident = _resident_ok(x, Cin, Cout)
        ctx.in_dtype = x.dtype
        ctx.pad8 = None if ctx.resident else _thin_plan(x, Cin, Cout)
        ctx.acc = accumulate_into is not None
        if ctx.acc:          # residual sum fused into the epilogue: accumulate_into += conv(x)   (ops.py:313)
            assert ctx.pad8 is None and tuple(accumulate_into.shape) == (N, Ho, Wo, Cout), "accumulate_into: shape"
        if ctx.pad8 == "in":
            # x8 = [x_hi | x_lo | 0 0]: the image to ~16 mantissa bits in the otherwise idle padding channels;
            # forward and wgrad pair it with the kernel duplicated over the two groups of rows
            x8 = pad_channels(x, 8, BF16, mode=hip.PAD_SPLIT)
            ydt = out_dtype or BF16
            d = hip.conv_desc(N, H, W_, 8, Ho, Wo, Cout, k, stride, pad_lo, pad_mode, hip.COMPUTE_BF16, hip.BF16,
                              hip.BF16 if ydt == BF16 else hip.F32, 1)
            y = torch.empty((N, Ho, Wo, Cout), dtype=ydt, device=x.device)
            ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, x.device)
            pt = weight_packs(pad_channels(w, 8, torch.float32, axis=2, mode=hip.PAD_DUP))[1]
            check(L.bg_conv2d_fwd(d, act(x8), act(pt), f32(bias), None, act(y), 0, f32(ws), nb, stream()))
            ctx.desc, ctx.rgb = d, False
            ctx.x, ctx.w, ctx.bias = x8, w, bias
            return y
        if ctx.pad8 == "out":
            # w8 = [w_hi | w_lo | 0 0] along the output channels: y = y8[:3] + y8[3:6] sees the kernel to ~16 bits
            b8 = None if bias is None else pad_channels(bias, 8, torch.float32)
            ydt = out_dtype or torch.float32
            d = hip.conv_desc(N, H, W_, Cin, Ho, Wo, 8, k, stride, pad_lo, pad_mode, hip.COMPUTE_BF16, hip.BF16,
                              hip.F32, 1)
            y8 = torch.empty((N, Ho, Wo, 8), dtype=torch.float32, device=x.device)
            ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, x.device)
            pt = weight_packs(pad_channels(w, 8, torch.float32, axis=3, mode=hip.PAD_SPLIT))[1]
            check(L.bg_conv2d_fwd(d, act(x), act(pt), f32(b8), None, act(y8), 0, f32(ws), nb, stream()))
            ctx.desc, ctx.rgb = d, False
            ctx.x, ctx.w, ctx.bias = x, w, bias
            return pad_channels(y8, Cout, ydt, mode=hip.PAD_FOLD)
        if x.dtype == BF16 and not ctx.resident:
            x = cast(x, torch.float32)         # (odd channel counts: fp32-tensor kernels)
        if ctx.resident:
            ydt = out_dtype or BF16
            d = hip.conv_desc(N, H, W_, Cin, Ho, Wo, Cout, k, stride, pad_lo, pad_mode, hip.COMPUTE_BF16, hip.BF16,
                              hip.BF16 if ydt == BF16 else hip.F32, 1)
            if ctx.acc:
                y = accumulate_into
                assert y.dtype == ydt and y.is_contiguous(), (y.dtype, ydt)
                ctx.mark_dirty(y)
            else:
                y = torch.empty((N, Ho, Wo, Cout), dtype=ydt, device=x.device)
            ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, x.device)
            check(L.bg_conv2d_fwd(d, act(x), act(weight_packs(w)[1]), f32(bias), None, act(y), int(ctx.acc), f32(ws), nb,
                                  stream()))
            ctx.desc, ctx.rgb = d, False
            ctx.x, ctx.w, ctx.bias = x, w, bias
            return y
        w = _c(w)
        d = hip.conv_desc(N, H, W_, Cin, Ho, Wo, Cout, k, stride, pad_lo, pad_mode, Precision.compute)
        if ctx.acc:
            y = accumulate_into
            assert y.dtype == torch.float32 and y.is_contiguous()
            ctx.mark_dirty(y)
        else:
            y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        ctx.rgb = bool(L.bg_rgbconv_supported(d))      # <= 3 output channels: direct HBM-bound kernels
        if ctx.rgb:
            check(L.bg_rgbconv_fwd(d, f32(x), f32(w), f32(bias), f32(y), int(ctx.acc), stream()))
        else:
            ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, d, x.device)
            check(L.bg_conv2d_fwd(d, f32(x), f32(w), f32(bias), None, f32(y), int(ctx.acc), f32(ws), nb, stream()))
        ctx.desc = d
        ctx.x, ctx.w, ctx.bias = x, w, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        d, x, w, bias = ctx.desc, ctx.x, ctx.w, ctx.bias
        L = lib()
        dx = None
        db = None
        if ctx.pad8:
            Cin, Cout = w.shape[2], w.shape[3]
            thin_in = ctx.pad8 == "in"
            if bias is not None:
                db = param_grad(bias, ctx.needs_input_grad[2], lambda out: _bias_grad(dy.view(-1, Cout), out))
            # thin output: dy8 = [dy_hi | dy_lo | 0 0] - the gradient of the generated image, the 3-channel bottleneck
            # every generator gradient passes through, enters the GEMMs to ~16 bits
            dy8 = cast(dy, BF16) if thin_in else pad_channels(dy, 8, BF16, mode=hip.PAD_SPLIT)
            db16 = hip.conv_desc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.k, d.stride, d.pad_lo, d.pad_mode,
                                 hip.COMPUTE_BF16, hip.BF16, hip.BF16, 1)
            if ctx.needs_input_grad[0]:
                if thin_in:
                    # kernel rows [w_hi | w_lo]: dx = dx8[:3] + dx8[3:6], written by an fp32 epilogue (image gradient)
                    pp = weight_packs(pad_channels(w, 8, torch.float32, axis=2, mode=hip.PAD_SPLIT))[0]
                    dd = hip.conv_desc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.k, d.stride, d.pad_lo, d.pad_mode,
                                       hip.COMPUTE_BF16, hip.F32, hip.BF16, 1)
                    dx8 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                else:
                    pp = weight_packs(pad_channels(w, 8, torch.float32, axis=3, mode=hip.PAD_DUP))[0]
                    dd = db16
                    dx8 = torch.empty_like(x)
                ws, nb = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, dd, x.device)
                check(L.bg_conv2d_dgrad(dd, act(dy8), act(pp), None, act(dx8), 0, f32(ws), nb, stream()))
                dx = pad_channels(dx8, Cin, ctx.in_dtype, mode=hip.PAD_FOLD) if thin_in else dx8

            def wg8(out):
                # x8 = [x_hi | x_lo] (thin input) or dy8 = [dy_hi | dy_lo] (thin output): dw = the two halves folded
                nb = L.bg_conv2d_wgrad_workspace_bytes(db16)
                ws = workspace(nb, x.device)
                dw8 = torch.empty((d.k, d.k, d.Cin, d.Cout), dtype=torch.float32, device=x.device)
                check(L.bg_conv2d_wgrad(db16, act(x), act(dy8), f32(dw8), f32(ws), nb, stream()))
                if thin_in:
                    check(L.bg_pad_channels(act(dw8), hip.F32, act(out), hip.F32, d.k * d.k, 8, Cin, d.Cout,
                                            hip.PAD_FOLD, stream()))
                else:
                    check(L.bg_pad_channels(act(dw8), hip.F32, act(out), hip.F32, d.k * d.k * d.Cin, 8, Cout, 1,
                                            hip.PAD_FOLD, stream()))
            dw = param_grad(w, ctx.needs_input_grad[1], wg8)
            ctx.x = ctx.w = ctx.bias = None
            return dx, dw, db, None, None, None, None, None, None, None
        if bias is not None:
            db = param_grad(bias, ctx.needs_input_grad[2], lambda out: _bias_grad(dy.view(-1, d.Cout), out))
        if ctx.resident:
            dyb = cast(dy, BF16)               # (fp32 when the consumer asked for an fp32 output)
            db16 = hip.conv_desc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.k, d.stride, d.pad_lo, d.pad_mode,
                                 hip.COMPUTE_BF16, hip.BF16, hip.BF16, 1)
            if ctx.needs_input_grad[0]:
                dx, add = _fork_target(ctx.fork, x)
                ws, nb = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, db16, x.device)
                check(L.bg_conv2d_dgrad(db16, act(dyb), act(weight_packs(w)[0]), None, act(dx), int(add), f32(ws), nb,
                                        stream()))
                _fork_done(ctx.fork, dx)

            def wg16(out):
                nb = L.bg_conv2d_wgrad_workspace_bytes(db16)
                ws = workspace(nb, x.device)
                check(L.bg_conv2d_wgrad(db16, act(x), act(dyb), f32(out), f32(ws), nb, stream()))
            dw = param_grad(w, ctx.needs_input_grad[1], wg16)
            ctx.x = ctx.w = ctx.bias = None
            return dx, dw, db, None, None, None, None, None, None, (dy if ctx.acc else None)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.rgb:
                check(L.bg_rgbconv_dgrad(d, f32(dy), f32(w), f32(dx), 0, stream()))
            else:
                ws, nb = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, d, x.device)
                check(L.bg_conv2d_dgrad(d, f32(dy), f32(w), None, f32(dx), 0, f32(ws), nb, stream()))

        def wg(out):
            if ctx.rgb:
                nb = L.bg_rgbconv_wgrad_workspace_bytes(d)
                ws = workspace(nb, x.device)
                check(L.bg_rgbconv_wgrad(d, f32(x), f32(dy), f32(out), f32(ws), nb, stream()))
                return
            nb = L.bg_conv2d_wgrad_workspace_bytes(d)
            ws = workspace(nb, x.device)
            check(L.bg_conv2d_wgrad(d, f32(x), f32(dy), f32(out), f32(ws), nb, stream()))
        dw = param_grad(w, ctx.needs_input_grad[1], wg)
        ctx.x = ctx.w = ctx.bias = None
        if dx is not None:
            dx = cast(dx, ctx.in_dtype)
        return dx, dw, db, None, None, None, None, None, None, (dy if ctx.acc else None)


class Deconv2dFn(Function):
    """tf.nn.conv2d_transpose(SAME)+bias_add (ops.py:127-132); w is [k,k,Cout,Cin]."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad_lo, accumulate_into, stats_box=None):
        """``stats_box`` (bf16-resident mode, optional): a one-element list; when the launch can produce the batch-norm
        sums of its output in its epilogue (bg_deconv2d_fwd_stats) the fp64 [2 Cout] sums tensor is left in it."""
        ctx.fork = getattr(x, "bg_fork", None)
        x = _c(x)
        N, H, W_, Cin = x.shape
        k, _, Cout, cin2 = w.shape
        assert cin2 == Cin, (x.shape, w.shape)
        ctx.resident = _resident_ok(x, Cin, Cout)
        ctx.in_dtype = x.dtype
        if x.dtype == BF16 and not ctx.resident:
            x = cast(x, torch.float32)
        ydt = BF16 if ctx.resident else torch.float32
        if ctx.resident:
            d = hip.conv_desc(N, H, W_, Cin, H * stride, W_ * stride, Cout, k, stride, pad_lo, hip.PAD_ZERO,
                              hip.COMPUTE_BF16, hip.BF16, hip.BF16, 1)
        else:
            w = _c(w)
            d = hip.conv_desc(N, H, W_, Cin, H * stride, W_ * stride, Cout, k, stride, pad_lo, hip.PAD_ZERO,
                              Precision.compute)
        if accumulate_into is not None:
            y = accumulate_into          # residual sum fused into the epilogue: y += deconv(x)
            assert y.dtype == ydt, (y.dtype, ydt)
            ctx.mark_dirty(y)
            acc = 1
        else:
            y = torch.empty((N, H * stride, W_ * stride, Cout), dtype=ydt, device=x.device)
            acc = 0
        L = lib()
        ws, nb = hip.scratch(L.bg_deconv2d_fwd_workspace_bytes, d, x.device)
        wk = weight_packs(w)[0] if ctx.resident else w
        snb = int(L.bg_deconv2d_fwd_stats_workspace_bytes(d)) if (stats_box is not None and ctx.resident) else 0
        if snb > 0:
            sums = zeros(2 * Cout, torch.float64, x.device)
            sws = workspace(snb, x.device)
            check(L.bg_deconv2d_fwd_stats(d, act(x), act(wk), f32(bias), None, act(y), acc, hip.ptr(sums), f32(sws), snb,
                                          f32(ws), nb, stream()))
            stats_box[0] = sums
        else:
            check(L.bg_deconv2d_fwd(d, act(x), act(wk), f32(bias), None, act(y), acc, f32(ws), nb, stream()))
        ctx.desc = d
        ctx.x, ctx.w, ctx.bias = x, w, bias
        ctx.acc = acc
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        d, x, w, bias = ctx.desc, ctx.x, ctx.w, ctx.bias
        L = lib()
        dx = None
        if ctx.resident:
            dy = cast(dy, BF16)
        if ctx.needs_input_grad[0]:
            dx, add = _fork_target(ctx.fork if x.dtype == ctx.in_dtype else None, x)
            ws, nb = hip.scratch(L.bg_deconv2d_dgrad_workspace_bytes, d, x.device)
            wk = weight_packs(w)[1] if ctx.resident else w
            check(L.bg_deconv2d_dgrad(d, act(dy), act(wk), None, act(dx), int(add), f32(ws), nb, stream()))
            if x.dtype == ctx.in_dtype:
                _fork_done(ctx.fork, dx)

        def wg(out):
            nb = L.bg_deconv2d_wgrad_workspace_bytes(d)
            ws = workspace(nb, x.device)
            check(L.bg_deconv2d_wgrad(d, act(x), act(dy), f32(out), f32(ws), nb, stream()))
        dw = param_grad(w, ctx.needs_input_grad[1], wg)
        db = None
        if bias is not None:
            db = param_grad(bias, ctx.needs_input_grad[2], lambda out: _bias_grad(dy.view(-1, d.Cout), out))
        ctx.x = ctx.w = ctx.bias = None
        dacc = dy if ctx.acc else None
        if dx is not None:
            dx = cast(dx, ctx.in_dtype)
        return dx, dw, db, None, None, dacc, None


# ------------------------------------------------------------------------------------------
# matmul family
# ------------------------------------------------------------------------------------------
def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=1, sA=0, sB=0, sC=0,
         bias=None, alpha_dev=None, accumulate=False):
    """Raw bg_gemm launch on preallocated tensors (row-major, explicit leading dimensions)."""
    bf16 = Precision.gemm_bf16 and M >= 128 and N >= 128 and K >= 128
    d = hip.BgGemmDesc(M, N, K, int(transA), int(transB), lda, ldb, ldc, batch, sA, sB, sC,
                       hip.COMPUTE_BF16 if bf16 else hip.COMPUTE_F32, 0)
    L = lib()
    nb = L.bg_gemm_workspace_bytes(d)
    ws = workspace(nb, C.device) if nb else None
    check(L.bg_gemm(d, f32(A) if A.is_contiguous() else hip.c_void_p(A.data_ptr()),
                    f32(B) if B.is_contiguous() else hip.c_void_p(B.data_ptr()),
                    f32(bias), f32(alpha_dev), f32(C), int(accumulate), f32(ws) if ws is not None else None,
                    nb, stream()))
    return C


def _row_view(t):
    """[rows, cols] view with unit column stride -> (tensor, ld).  Column slices are used in place."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.float32, (t.shape, t.stride())
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


class DenseFn(Function):
    """tf.matmul(x, w) + bias (ops.py:163-165).  x may be a column slice of a wider matrix."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x, lda = _row_view(x)
        w = _c(w)
        M, K = x.shape
        N = w.shape[1]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gemm(x, w, y, M, N, K, lda, N, N, bias=bias)
        ctx.x, ctx.w, ctx.bias, ctx.lda = x, w, bias, lda
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        x, w, bias, lda = ctx.x, ctx.w, ctx.bias, ctx.lda
        M, K = x.shape
        N = w.shape[1]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
            gemm(dy, w, dx, M, K, N, N, N, K, transB=True)            # dx = dy @ w^T

        def wg(out):
            gemm(x, dy, out, K, N, M, lda, N, N, transA=True)         # dw = x^T @ dy
        dw = param_grad(w, ctx.needs_input_grad[1], wg)
        db = None
        if bias is not None:
            db = param_grad(bias, ctx.needs_input_grad[2], lambda out: _bias_grad(dy, out))
        ctx.x = ctx.w = ctx.bias = None
        return dx, dw, db


class GroupedDenseFn(Function):
    """n dense projections y_i = x_i w_i + b_i on the same batch rows in ONE launch each way (csrc/dense_group.hip): the
    beta / gamma projections of the conditional batch norms of a generator block (ops.py:623-624).  Arguments:
    n, then (x_i, w_i, b_i) flattened; x_i are [B, K_i] row views (column slices allowed) that need no gradient."""

    @staticmethod
    def forward(ctx, n, *args):
        assert len(args) == 3 * n and 1 <= n <= hip.DENSE_GROUP_MAX
        items = (hip.BgDenseItem * n)()
        ys, keep = [], []
        B = args[0].shape[0]
        for i in range(n):
            x, w, b = args[3 * i], args[3 * i + 1], args[3 * i + 2]
            x, ldx = _row_view(x)
            w = _c(w)
            assert x.shape[0] == B and w.shape[0] == x.shape[1] and not x.requires_grad
            y = torch.empty((B, w.shape[1]), dtype=torch.float32, device=x.device)
            it = items[i]
            it.x, it.ldx, it.w, it.bias, it.y = x.data_ptr(), ldx, f32(w).value, (f32(b).value if b is not None else None), \
                y.data_ptr()
            it.K, it.N = x.shape[1], w.shape[1]
            ys.append(y)
            keep.append((x, ldx, w, b))
        check(lib().bg_dense_group_fwd(items, n, B, stream()))
        ctx.keep, ctx.B = keep, B
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        if _Mode.inputs_only:
            return (None,) * (1 + 3 * len(ctx.keep))
        n = len(ctx.keep)
        items = (hip.BgDenseItem * n)()
        m = 0
        outs = [None]
        hold = []
        for i, (x, ldx, w, b) in enumerate(ctx.keep):
            dy = dys[i]
            gw = gb = None
            if dy is not None and ctx.needs_input_grad[2 + 3 * i]:
                dy = _c(dy)
                it = items[m]
                m += 1
                it.x, it.ldx, it.y, it.K, it.N = x.data_ptr(), ldx, dy.data_ptr(), x.shape[1], w.shape[1]
                if is_variable(w):
                    slot, acc = grad_slot(w)
                else:
                    slot, acc = torch.empty(w.shape, dtype=torch.float32, device=w.device), False
                    gw = slot
                it.dw, it.acc_w = slot.data_ptr(), int(acc)
                if b is not None and ctx.needs_input_grad[3 + 3 * i]:
                    if is_variable(b):
                        bs, bacc = grad_slot(b)
                    else:
                        bs, bacc = torch.empty(b.shape, dtype=torch.float32, device=b.device), False
                        gb = bs
                    it.db, it.acc_b = bs.data_ptr(), int(bacc)
                hold.append((dy, slot))
            outs += [None, gw, gb]
        if m:
            check(lib().bg_dense_group_wgrad(items, m, ctx.B, stream()))
        ctx.keep = None
        return tuple(outs)


class AttentionFn(Function):
    """softmax(g f^T) h  (ops.py:481-485): q [B,N,dq], k [B,Nk,dq], v [B,Nk,dv] -> o [B,N,dv].
    Fused (flash-style) kernels when the shape is supported (bg_attention2_supported): the [B,N,Nk]
    probabilities are never materialised; otherwise two GEMMs + softmax with P kept for backward."""

    flash = True          # tests flip this to exercise the materialised form on supported shapes

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = _c(q), _c(k), _c(v)
        B, N, dq = q.shape
        Nk, dv = k.shape[1], v.shape[2]
        L = lib()
        o = torch.empty((B, N, dv), dtype=torch.float32, device=q.device)
        ctx.fused = bool(AttentionFn.flash and L.bg_attention2_supported(N, Nk, dq, dv))
        if ctx.fused:
            lse = torch.empty((B, N), dtype=torch.float32, device=q.device)
            check(L.bg_attention2_fwd(f32(q), f32(k), f32(v), f32(o), f32(lse), B, N, Nk, dq, dv, stream()))
            ctx.save_for_backward(q, k, v, o, lse)
            return o
        p = torch.empty((B, N, Nk), dtype=torch.float32, device=q.device)
        gemm(q, k, p, N, Nk, dq, dq, dq, Nk, transB=True, batch=B, sA=N * dq, sB=Nk * dq, sC=N * Nk)
        check(L.bg_softmax_fwd(f32(p), f32(p), B * N, Nk, stream()))
        gemm(p, v, o, N, dv, Nk, Nk, dv, dv, batch=B, sA=N * Nk, sB=Nk * dv, sC=N * dv)
        ctx.save_for_backward(q, k, v, p)
        return o

    @staticmethod
    def backward(ctx, do):
        do = _c(do)
        L = lib()
        if ctx.fused:
            q, k, v, o, lse = ctx.saved_tensors
            B, N, dq = q.shape
            Nk, dv = k.shape[1], v.shape[2]
            dqq, dkk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            delta = torch.empty((B, N), dtype=torch.float32, device=q.device)
            check(L.bg_attention2_bwd(f32(q), f32(k), f32(v), f32(o), f32(do), f32(lse), f32(dqq), f32(dkk), f32(dvv),
                                      f32(delta), B, N, Nk, dq, dv, stream()))
            return dqq, dkk, dvv
        q, k, v, p = ctx.saved_tensors
        B, N, dq = q.shape
        Nk, dv = k.shape[1], v.shape[2]
        dev = q.device
        # dV = P^T dO
        dvv = torch.empty_like(v)
        gemm(p, do, dvv, Nk, dv, N, Nk, dv, dv, transA=True, batch=B, sA=N * Nk, sB=N * dv, sC=Nk * dv)
        # dP = dO V^T ; dS = P * (dP - sum(dP*P))
        dp = torch.empty((B, N, Nk), dtype=torch.float32, device=dev)
        gemm(do, v, dp, N, Nk, dv, dv, dv, Nk, transB=True, batch=B, sA=N * dv, sB=Nk * dv, sC=N * Nk)
        check(L.bg_softmax_bwd(f32(p), f32(dp), f32(dp), B * N, Nk, stream()))
        # dQ = dS K ; dK = dS^T Q
        dqq = torch.empty_like(q)
        gemm(dp, k, dqq, N, dq, Nk, Nk, dq, dq, batch=B, sA=N * Nk, sB=Nk * dq, sC=N * dq)
        dkk = torch.empty_like(k)
        gemm(dp, q, dkk, Nk, dq, N, Nk, dq, dq, transA=True, batch=B, sA=N * Nk, sB=N * dq, sC=Nk * dq)
        return dqq, dkk, dvv


class SaFrontFn(Function):
    """bf16-resident front half of self_attention_2 (ops.py:467-485): the f | g | h 1x1 projections as ONE GEMM on the
    shared packed weights, one 2x2 max pool of the result, and the fused bf16 attention reading q = g, k = pool(f),
    v = pool(h) as column slices (csrc/attention16.hip).  x [B,H,W,C] bf16 -> o [B,H,W,c_h] bf16."""

    @staticmethod
    def supported(x, wf, wg, wh):
        grp = getattr(wf, "bg_group", None)
        if grp is None or getattr(wg, "bg_group", None) is not grp or getattr(wh, "bg_group", None) is not grp:
            return False
        if x.dtype != BF16 or x.dim() != 4 or x.shape[1] % 2 or x.shape[2] % 2:
            return False
        N, Nk = x.shape[1] * x.shape[2], x.shape[1] * x.shape[2] // 4
        return bool(lib().bg_attention16_supported(N, Nk, wf.shape[3], wh.shape[3])) and wf.shape[3] == wg.shape[3]

    @staticmethod
    def _desc(B, N, Nk, d, dv, ct):
        D = hip.BgAttn16Desc()
        D.B, D.N, D.Nk, D.d, D.dv = B, N, Nk, d, dv
        D.ldq, D.sq, D.ldk, D.sk, D.ldv, D.sv = ct, N * ct, ct, Nk * ct, ct, Nk * ct
        D.ldo, D.so, D.ldg, D.sg = dv, N * dv, dv, N * dv
        D.lddq, D.sdq, D.lddk, D.sdk, D.lddv, D.sdv = ct, N * ct, ct, Nk * ct, ct, Nk * ct
        return D

    @staticmethod
    def forward(ctx, x, wf, wg, wh, bf_, bg_, bh_):
        ctx.fork = getattr(x, "bg_fork", None)
        x = _c(x)
        B, H, W_, C = x.shape
        grp = wf.bg_group
        ct = grp.pack_p.shape[2]
        of, og, oh = wf.bg_group_off, wg.bg_group_off, wh.bg_group_off
        d, dv = wf.shape[3], wh.shape[3]
        L = lib()
        bias = None
        if bf_ is not None:
            bias = torch.empty(ct, dtype=torch.float32, device=x.device)
            with torch.no_grad():
                for o_, b_ in ((of, bf_), (og, bg_), (oh, bh_)):
                    bias[o_:o_ + b_.shape[0]].copy_(b_)                   # (device-to-device copies of 3 small vectors)
        cd = hip.conv_desc(B, H, W_, C, H, W_, ct, 1, 1, 0, hip.PAD_REFLECT, hip.COMPUTE_BF16, hip.BF16, hip.BF16, 1)
        y = torch.empty((B, H, W_, ct), dtype=BF16, device=x.device)
        ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, cd, x.device)
        check(L.bg_conv2d_fwd(cd, act(x), act(grp.pack_t), f32(bias), None, act(y), 0, f32(ws), nb, stream()))
        yp = torch.empty((B, H // 2, W_ // 2, ct), dtype=BF16, device=x.device)
        check(L.bg_maxpool2_fwd_t(act(y), act(yp), hip.BF16, B, H, W_, ct, stream()))
        N, Nk = H * W_, (H // 2) * (W_ // 2)
        o = torch.empty((B, H, W_, dv), dtype=BF16, device=x.device)
        lse = torch.empty((B, N), dtype=torch.float32, device=x.device)
        D = SaFrontFn._desc(B, N, Nk, d, dv, ct)
        P = hip.c_void_p
        check(L.bg_attention16_fwd(D, P(y.data_ptr() + 2 * og), P(yp.data_ptr() + 2 * of), P(yp.data_ptr() + 2 * oh),
                                   act(o), f32(lse), stream()))
        ctx.saved = (x, y, yp, o, lse)
        ctx.vars = (wf, wg, wh, bf_, bg_, bh_)
        ctx.cd, ctx.D, ctx.offs = cd, D, (of, og, oh)
        return o

    @staticmethod
    def backward(ctx, do):
        do = cast(_c(do), BF16)
        x, y, yp, o, lse = ctx.saved
        wf, wg, wh, bf_, bg_, bh_ = ctx.vars
        of, og, oh = ctx.offs
        cd, D = ctx.cd, ctx.D
        grp = wf.bg_group
        ct = grp.pack_p.shape[2]
        B, H, W_, C = x.shape
        L = lib()
        P = hip.c_void_p
        dev = x.device
        # gradients of the pooled keys / values (the query columns of the pooled tensor receive nothing)
        dyp = torch.zeros_like(yp)
        delta = torch.empty((B, D.N), dtype=torch.float32, device=dev)
        qp, kp, vp = P(y.data_ptr() + 2 * og), P(yp.data_ptr() + 2 * of), P(yp.data_ptr() + 2 * oh)
        check(L.bg_attention16_bwd(D, qp, kp, vp, act(o), act(do), f32(lse), None, P(dyp.data_ptr() + 2 * of),
                                   P(dyp.data_ptr() + 2 * oh), f32(delta), stream()))
        dy = torch.empty_like(y)
        check(L.bg_maxpool2_bwd_t(act(y), act(dyp), act(dy), hip.BF16, B, H, W_, ct, stream()))
        # ... then the query gradient goes straight into its (so far zero) columns of dy
        check(L.bg_attention16_bwd(D, qp, kp, vp, act(o), act(do), f32(lse), P(dy.data_ptr() + 2 * og), None, None,
                                   f32(delta), stream()))
        dx = None
        if ctx.needs_input_grad[0]:
            # (the block's input forks into this projection and the gated residual: the residual's gradient - the block's
            #  incoming gradient itself, left in the fork state by ScaleAddFn.backward - is the buffer accumulated into)
            dx, add = _fork_target(ctx.fork, x)
            ws, nb = hip.scratch(L.bg_conv2d_dgrad_workspace_bytes, cd, dev)
            check(L.bg_conv2d_dgrad(cd, act(dy), act(grp.pack_p), None, act(dx), int(add), f32(ws), nb, stream()))
            _fork_done(ctx.fork, dx)
        if not _Mode.inputs_only:
            need_w = any(w.requires_grad for w in (wf, wg, wh))
            if need_w:
                dwt = torch.empty((C, ct), dtype=torch.float32, device=dev)
                nb = L.bg_conv2d_wgrad_workspace_bytes(cd)
                ws = workspace(nb, dev)
                check(L.bg_conv2d_wgrad(cd, act(x), act(dy), f32(dwt), f32(ws), nb, stream()))
                for w_, o_ in ((wf, of), (wg, og), (wh, oh)):
                    if w_.requires_grad:
                        emit_grad(w_, lambda out, o_=o_, w_=w_: out.view(C, w_.shape[3]).copy_(dwt[:, o_:o_ + w_.shape[3]]))
            if bf_ is not None and any(b_.requires_grad for b_ in (bf_, bg_, bh_)):
                dbt = torch.empty(ct, dtype=torch.float32, device=dev)
                _bias_grad(dy.view(-1, ct), dbt)
                for b_, o_ in ((bf_, of), (bg_, og), (bh_, oh)):
                    if b_.requires_grad:
                        emit_grad(b_, lambda out, o_=o_, b_=b_: out.copy_(dbt[o_:o_ + b_.shape[0]]))
        ctx.saved = None
        return dx, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------
# spectral norm
# ------------------------------------------------------------------------------------------
class SpectralNormFn(Function):
    """One power iteration + w / sigma (ops.py:718-747).  ``u`` is updated in place (ops.py:743)."""

    @staticmethod
    def forward(ctx, w, u):
        w = _c(w)
        cols = w.shape[-1]
        rows = w.numel() // cols
        L = lib()
        dev = w.device
        wn = torch.empty_like(w)
        v = torch.empty(rows, dtype=torch.float32, device=dev)
        sigma = torch.empty(1, dtype=torch.float32, device=dev)
        nb = L.bg_spectral_norm_workspace_bytes(rows, cols)
        ws = workspace(nb, dev)
        check(L.bg_spectral_norm_fwd(f32(w), f32(u), f32(u), f32(v), f32(sigma), f32(wn), rows, cols,
                                     f32(ws), nb, stream()))
        ctx.w, ctx.u, ctx.v, ctx.sigma, ctx.wn = w, u, v, sigma, wn
        ctx.rows, ctx.cols = rows, cols
        return wn

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        L = lib()
        rows, cols = ctx.rows, ctx.cols
        w, u, v, sigma, wn = ctx.w, ctx.u, ctx.v, ctx.sigma, ctx.wn

        def prod(out):
            ws = workspace(16, g.device)
            check(L.bg_spectral_norm_bwd(f32(g), f32(wn), f32(u), f32(v), f32(sigma), f32(out), rows, cols,
                                         f32(ws), 16, stream()))
        dw = param_grad(w, ctx.needs_input_grad[0], prod)
        ctx.w = ctx.wn = None
        return dw, None


class PackGroup:
    """Shared packed bf16 copies of several 1x1 kernels on one input: pack_p [1, C, ct], pack_t [1, ct, C]."""

    def __init__(self, pack_p, pack_t):
        self.pack_p, self.pack_t, self.offsets = pack_p, pack_t, []


def sn_shard_layout(numels, rows, cols, world):
    """Who iterates which weight under data parallelism, and where its results live (SnBatch(shard=...)).  Pure host
    arithmetic: ``owner[i]`` by longest-processing-time assignment on the weight sizes; ``place[i]`` = (owner, offsets of
    sigma | u | v_hat inside the owner's segment of the flat state buffer, in floats, each padded to 4); ``fill[r]`` = floats
    rank r's segment uses (the segment length is their maximum, so that one all-gather moves every segment)."""
    pad4 = lambda k: (k + 3) // 4 * 4                                  # noqa: E731
    n = len(numels)
    load = [0] * world
    owner = [0] * n
    for i in sorted(range(n), key=lambda i: (-numels[i], i)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += numels[i]
    fill = [0] * world
    place = []
    for i in range(n):
        r = owner[i]
        o = fill[r]
        place.append((r, o, o + 4, o + 4 + pad4(cols[i])))
        fill[r] = o + 4 + pad4(cols[i]) + pad4(rows[i])
    return owner, place, fill


class SnBatch:
    """Every spectrally-normalised weight of one network in one multi-tensor call (4 launches forward,
    2 backward) instead of a kernel chain per weight.  ``forward()`` runs the power iteration of all
    weights (``u`` in place) and returns the normalised weights as pseudo-variables: the consumers'
    backward kernels write dL/d(w/sigma) into a flat buffer through ``emit_grad`` and ``backward()``
    turns all of them into dL/dw inside the parameter-gradient arena (ops.py:718-747)."""

    ALIGN = 64

    def __init__(self, pairs, groups=None, shard=None):
        """``groups`` (bf16-resident mode): lists of indices into ``pairs`` of 1x1 kernels [1,1,C,c_i] applied to the
        same tensor (the f | g | h projections of self_attention_2): their packed copies are slices of one shared
        [C, sum c_i] / [sum c_i, C] pair, so the three projections are ONE GEMM in each direction.

        ``shard`` = (rank, world, process group) under data parallelism: the batch-independent power iteration (two
        GEMV passes over every weight) is SHARDED by weight - each rank iterates the weights it owns
        (longest-processing-time assignment by size) and an all-gather of sigma | u | v_hat brings everybody's results
        to everybody; only the normalisation (which every rank needs for its packed copies) runs on all weights.  The
        u vectors then live in one flat state buffer: ``self.u`` are views into it, and the caller re-points the store's
        ``u`` variables to them."""
        import ctypes
        L = lib()
        self.w = [p[0] for p in pairs]
        self.u = [p[1] for p in pairs]
        n = len(pairs)
        if not 0 < n <= 256:
            raise ValueError("SnBatch handles 1..256 weights, got %d" % n)
        dev = self.w[0].device
        offs, off, rows_tot, ws_off, ws_offs = [], 0, 0, 0, []
        self.rows, self.cols = [], []
        for w in self.w:
            cols = w.shape[-1]
            rows = w.numel() // cols
            self.rows.append(rows)
            self.cols.append(cols)
            offs.append(off)
            off += (w.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            ws_offs.append(ws_off)
            ws_off += (int(L.bg_spectral_norm_workspace_bytes(rows, cols)) + 15) // 16 * 16
            rows_tot += (rows + 3) // 4 * 4
        self.wn_flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gwn_flat = torch.zeros(off, dtype=torch.float32, device=dev)
        # bf16-resident path: every 4-D (conv / transposed-conv) kernel whose two channel counts are multiples of 8
        # also gets its two packed bf16 copies, written by the same normalisation launch
        packed = [Precision.resident and w.dim() == 4 and w.shape[2] % 8 == 0 and w.shape[3] % 8 == 0 for w in self.w]
        self.pack_flat = torch.zeros(2 * off if any(packed) else 0, dtype=BF16, device=dev)
        group_of = {}
        for gi, members in enumerate(groups or []):
            ws_ = [self.w[i] for i in members]
            cin = ws_[0].shape[2]
            if not (Precision.resident and all(w.dim() == 4 and w.shape[0] == 1 and w.shape[1] == 1 and w.shape[2] == cin
                                                and w.shape[3] % 4 == 0 for w in ws_) and cin % 8 == 0
                    and sum(w.shape[3] for w in ws_) % 8 == 0):
                continue
            ct = sum(w.shape[3] for w in ws_)
            grp = PackGroup(torch.zeros((1, cin, ct), dtype=BF16, device=dev), torch.zeros((1, ct, cin), dtype=BF16, device=dev))
            o = 0
            for i in members:
                group_of[i] = (grp, o)
                grp.offsets.append(o)
                o += self.w[i].shape[3]
        self.v_flat = torch.zeros(rows_tot, dtype=torch.float32, device=dev)
        self.sigma = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shard = shard
        state_views = None
        if shard is not None:
            rank, world, _ = shard
            self.owner, place, fill = sn_shard_layout([w.numel() for w in self.w], self.rows, self.cols, world)
            self.seg = max(max(fill), 4)
            self.state_flat = torch.zeros(world * self.seg, dtype=torch.float32, device=dev)
            state_views = []
            for i, (r, o_s, o_u, o_v) in enumerate(place):
                base = r * self.seg
                sg = self.state_flat.narrow(0, base + o_s, 1)
                uu = self.state_flat.narrow(0, base + o_u, self.cols[i]).view(self.u[i].shape)
                vv = self.state_flat.narrow(0, base + o_v, self.rows[i])
                with torch.no_grad():
                    uu.copy_(self.u[i])
                for attr in ("bg_name", "bg_grad", "bg_touched"):
                    if hasattr(self.u[i], attr):
                        setattr(uu, attr, getattr(self.u[i], attr))
                self.u[i] = uu
                state_views.append((sg, vv))
        self.ws_bytes = ws_off
        self.ws = workspace(max(ws_off, 8 * n), dev)
        self.dots = workspace(8 * n, dev)
        self.wn, self.v, self.dw = [], [], []
        items = (hip.BgSnItem * n)()
        r0 = 0
        for i, w in enumerate(self.w):
            wn = self.wn_flat.narrow(0, offs[i], w.numel()).view(w.shape)
            wn.bg_name = (getattr(w, "bg_name", None) or "w%d" % i) + ":sn"
            wn.bg_grad = self.gwn_flat.narrow(0, offs[i], w.numel()).view(w.shape)
            wn.bg_touched = False
            wn.bg_sigma = self.sigma.narrow(0, i, 1) if state_views is None else state_views[i][0]
            w.bg_sn_wn = wn                     # (the regulariser of w finds the packed copy of w / sigma through this)
            v = self.v_flat.narrow(0, r0, self.rows[i]) if state_views is None else state_views[i][1]
            r0 += (self.rows[i] + 3) // 4 * 4
            dw = getattr(w, "bg_grad", None)
            if dw is None:
                dw = torch.zeros_like(w)
                w.bg_grad = dw
                w.bg_touched = False
            self.wn.append(wn)
            self.v.append(v)
            self.dw.append(dw)
            it = items[i]
            it.w, it.u, it.v = w.data_ptr(), self.u[i].data_ptr(), v.data_ptr()
            it.sigma = wn.bg_sigma.data_ptr()
            it.w_norm, it.g_wnorm, it.dw = wn.data_ptr(), wn.bg_grad.data_ptr(), dw.data_ptr()
            it.ws_offset, it.rows, it.cols = ws_offs[i], self.rows[i], self.cols[i]
            if i in group_of:
                grp, o = group_of[i]
                wn.bg_group, wn.bg_group_off = grp, o
                it.pack_p = grp.pack_p.data_ptr() + 2 * o                      # column slice of [C, ct]
                it.pack_t = grp.pack_t.data_ptr() + 2 * o * w.shape[2]         # row block of [ct, C]
                it.taps, it.pack_p_ld = 1, grp.pack_p.shape[2]
            elif packed[i]:
                k2, A, B = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
                wn.bg_pack_p = self.pack_flat.narrow(0, 2 * offs[i], w.numel()).view(k2, A, B)
                wn.bg_pack_t = self.pack_flat.narrow(0, 2 * offs[i] + w.numel(), w.numel()).view(k2, B, A)
                it.pack_p, it.pack_t, it.taps = wn.bg_pack_p.data_ptr(), wn.bg_pack_t.data_ptr(), k2
            for t in (w, self.u[i], dw):
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise RuntimeError("SnBatch needs contiguous CUDA fp32 tensors")
        raw = bytes(items)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n = n
        self._ctypes = ctypes
        self.n_own = 0
        if shard is not None:
            own = [i for i in range(n) if self.owner[i] == shard[0]]
            self.n_own = len(own)
            if own:
                sub = (hip.BgSnItem * len(own))(*[items[i] for i in own])
                self.table_own = torch.frombuffer(bytearray(bytes(sub)), dtype=torch.uint8).to(dev)

    def forward(self, run_stamp=None):
        """``run_stamp``: identity of the run (ops.begin_run) in which the packed copies / sigma are written; consumers
        that read them outside the conv path (the regulariser's Gram from the packed weights) check it, so that packs of an
        earlier run - older weights, older sigma - are never used silently."""
        if self.shard is None:
            check(lib().bg_spectral_norm_batch_fwd(hip.ptr(self.table), self.n, hip.ptr(self.ws), self.ws_bytes, stream()))
        else:
            self.power_owned()
            self.gather_state()
            self.normalize_all()
        for w, wn in zip(self.w, self.wn):
            wn.requires_grad_(bool(w.requires_grad))
            wn.bg_touched = False
            wn.bg_run_stamp = run_stamp
        return self.wn

    def power_owned(self):
        """Sharded mode, step 1: the power iteration (u, v_hat, sigma) of the weights this rank owns."""
        if self.n_own:
            check(lib().bg_spectral_norm_batch_phase(hip.ptr(self.table_own), self.n_own, hip.ptr(self.ws), self.ws_bytes,
                                                     hip.SN_POWER, stream()))

    def gather_state(self):
        """Sharded mode, step 2: everybody's sigma | u | v_hat segment to everybody (one all-gather, in place)."""
        rank, world, pg = self.shard
        if world > 1:
            torch.distributed.all_gather_into_tensor(self.state_flat, self.state_flat.narrow(0, rank * self.seg, self.seg),
                                                     group=pg)

    def normalize_all(self):
        """Sharded mode, step 3: w / sigma (and the packed copies) of EVERY weight from the gathered sigma."""
        check(lib().bg_spectral_norm_batch_phase(hip.ptr(self.table), self.n, hip.ptr(self.ws), self.ws_bytes,
                                                 hip.SN_NORMALIZE, stream()))

    def backward(self):
        """dL/dw (+)= SN-backward of every normalised weight that received a gradient this step."""
        words = (self.n + 63) // 64
        en = [0] * words
        acc = [0] * words
        any_ = False
        for i, (w, wn) in enumerate(zip(self.w, self.wn)):
            if wn.bg_touched and w.requires_grad:
                en[i >> 6] |= 1 << (i & 63)
                any_ = True
                if w.bg_touched:
                    acc[i >> 6] |= 1 << (i & 63)
                w.bg_touched = True
            wn.bg_touched = False
        if not any_:
            return
        c = self._ctypes
        arr = c.c_uint64 * words
        check(lib().bg_spectral_norm_batch_bwd(hip.ptr(self.table), self.n, arr(*en), arr(*acc), hip.ptr(self.dots),
                                               8 * self.n, stream()))


# ------------------------------------------------------------------------------------------
# batch norm (+ PReLU)
# ------------------------------------------------------------------------------------------
class BnActFn(Function):
    """tf.nn.moments + tf.nn.batch_normalization (ops.py:630-638) / tf.layers.batch_normalization
    (ops.py:581-585), optionally fused with the PReLU that follows (ops.py:532-537).

    gamma/beta: [N,C] (per-sample, conditional BN) or [C].  ``reduce_fn`` (optional) all-reduces a
    small fp32 tensor across data-parallel ranks in place (cross-replica statistics)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, alpha, moving_mean, moving_var, momentum, eps, unbiased_mv, is_training,
                reduce_fn, world, renorm=None, out_dtype=None):
        ctx.fork = getattr(x, "bg_fork", None)
        pre_sums = getattr(x, "bg_bn_sums", None)      # left by the producing kernel's epilogue (Deconv2dFn stats_box)
        x = _c(x)
        ydt = out_dtype or x.dtype
        typed = x.dtype != torch.float32 or ydt != torch.float32
        N, H, W_, C = x.shape
        HW = H * W_
        per_sample = int(gamma.dim() == 2)
        L = lib()
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty(C, dtype=torch.float32, device=dev)
        count = float(N * HW * world)
        if is_training:
            if pre_sums is not None and pre_sums.numel() == 2 * C and os.environ.get("BG_FUSE_BNSTATS", "1") != "0":
                sums = pre_sums
            else:
                sums = zeros(2 * C, torch.float64, dev)
                if typed:
                    check(L.bg_bn_stats_t(act(x), dt(x), hip.ptr(sums), N * HW, C, stream()))
                else:
                    check(L.bg_bn_stats(f32(x), hip.ptr(sums), N * HW, C, stream()))
            if reduce_fn is not None:
                reduce_fn(sums)
            if renorm is not None:          # corrections first: they read the running statistics before any update
                r_ = torch.empty(C, dtype=torch.float32, device=dev)
                d_ = torch.empty(C, dtype=torch.float32, device=dev)
                check(L.bg_renorm_coeffs(hip.ptr(sums), count, f32(renorm["ref_mean"]), f32(renorm["ref_scale"]),
                                         int(renorm["scale_is_var"]), f32(renorm.get("weight")), eps,
                                         renorm["rmin"], renorm["rmax"], renorm["dmax"], renorm["decay"],
                                         renorm.get("fadein_decay", 0.9999), int(renorm["update"]), f32(r_), f32(d_), C,
                                         stream()))
            check(L.bg_bn_finalize(hip.ptr(sums), count, eps, momentum, int(unbiased_mv), f32(mean), f32(rstd),
                                   f32(moving_mean), f32(moving_var), C, stream()))
        else:
            # inference: population statistics (ops.py:643)
            check(L.bg_bn_population(f32(moving_mean), f32(moving_var), eps, f32(mean), f32(rstd), C, stream()))
        y = torch.empty(x.shape, dtype=ydt, device=dev)
        gamma_c, beta_c = _c(gamma), _c(beta)
        ctx.renorm_rd = None
        if renorm is not None and is_training:
            g_eff, b_eff = torch.empty_like(gamma_c), torch.empty_like(beta_c)
            check(L.bg_renorm_affine_fwd(f32(gamma_c), f32(beta_c), f32(r_), f32(d_), f32(g_eff), f32(b_eff),
                                         gamma_c.numel() // C, C, stream()))
            gamma_c, beta_c = g_eff, b_eff
            ctx.renorm_rd = (r_, d_)
        if typed:
            check(L.bg_bn_apply_act_fwd_t(act(x), dt(x), f32(mean), f32(rstd), f32(gamma_c), f32(beta_c), per_sample,
                                          f32(alpha), act(y), dt(y), N, HW, C, stream()))
        else:
            check(L.bg_bn_apply_act_fwd(f32(x), f32(mean), f32(rstd), f32(gamma_c), f32(beta_c), per_sample,
                                        f32(alpha), f32(y), N, HW, C, stream()))
        ctx.typed = typed
        if KinkProbe.sites is not None and alpha is not None:
            KinkProbe.sites.append((getattr(alpha, "bg_name", None), "bn", x.detach().float().clone(), mean.clone(),
                                    rstd.clone(), gamma_c.detach().clone(), beta_c.detach().clone(), per_sample))
        ctx.x, ctx.mean, ctx.rstd = x, mean, rstd
        BnActFn.last_stats = (mean, rstd, count)        # (the tangent pass of the gradient penalty reads them: BnTangentFn)
        ctx.gamma, ctx.beta, ctx.alpha = gamma, beta, alpha
        ctx.gamma_c, ctx.beta_c = gamma_c, beta_c
        ctx.per_sample, ctx.count = per_sample, count
        ctx.reduce_fn = reduce_fn
        ctx.is_training = is_training
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        x, mean, rstd = ctx.x, ctx.mean, ctx.rstd
        gamma, beta, alpha = ctx.gamma, ctx.beta, ctx.alpha
        N, H, W_, C = x.shape
        HW = H * W_
        L = lib()
        dev = x.device
        ps = ctx.per_sample
        part = torch.empty((3, N, C), dtype=torch.float32, device=dev)
        if ctx.typed:
            check(L.bg_bn_apply_act_bwd_reduce_t(act(x), dt(x), act(dy), dt(dy), f32(mean), f32(rstd), f32(ctx.gamma_c),
                                                 f32(ctx.beta_c), ps, f32(alpha), f32(part), N, HW, C, stream()))
        else:
            check(L.bg_bn_apply_act_bwd_reduce(f32(x), f32(dy), f32(mean), f32(rstd), f32(ctx.gamma_c), f32(ctx.beta_c),
                                               ps, f32(alpha), f32(part), N, HW, C, stream()))
        gshape = (N, C) if ps else (C,)
        dgamma = torch.empty(gshape, dtype=torch.float32, device=dev)
        dbeta = torch.empty(gshape, dtype=torch.float32, device=dev)
        dalpha = torch.empty(C, dtype=torch.float32, device=dev) if alpha is not None else None
        cm = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(L.bg_bn_bwd_finalize(f32(part), f32(ctx.gamma_c), ps, ctx.count, f32(dgamma), f32(dbeta), f32(dalpha),
                                   f32(cm), N, C, stream()))
        if ctx.renorm_rd is not None:       # gamma enters through r*gamma and d*gamma; beta unchanged
            r_, d_ = ctx.renorm_rd
            dg_eff = dgamma
            dgamma = torch.empty_like(dg_eff)
            check(L.bg_renorm_affine_bwd(f32(dg_eff), f32(dbeta), f32(r_), f32(d_), f32(dgamma), dg_eff.numel() // C, C,
                                         stream()))
        if not ctx.is_training:
            cm.zero_()                      # population statistics are constants
        elif ctx.reduce_fn is not None:
            ctx.reduce_fn(cm)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.typed:
                dx, add = _fork_target(ctx.fork, x)
                check(L.bg_bn_apply_act_bwd_dx_t(act(x), dt(x), act(dy), dt(dy), f32(mean), f32(rstd), f32(ctx.gamma_c),
                                                 f32(ctx.beta_c), ps, f32(alpha), f32(cm), act(dx), act(dx) if add else None,
                                                 N, HW, C, stream()))
                _fork_done(ctx.fork, dx)
            else:
                dx = torch.empty_like(x)
                check(L.bg_bn_apply_act_bwd_dx(f32(x), f32(dy), f32(mean), f32(rstd), f32(ctx.gamma_c), f32(ctx.beta_c),
                                               ps, f32(alpha), f32(cm), f32(dx), N, HW, C, stream()))

        def deliver(t, needed, g):
            if not needed:
                return None
            if is_variable(t):
                emit_grad(t, lambda out: out.copy_(g))
                return None
            return g
        dg = deliver(gamma, ctx.needs_input_grad[1], dgamma)
        db = deliver(beta, ctx.needs_input_grad[2], dbeta)
        da = deliver(alpha, ctx.needs_input_grad[3], dalpha) if alpha is not None else None
        ctx.x = None
        return dx, dg, db, da, None, None, None, None, None, None, None, None, None, None


class PReluFn(Function):
    """ops.py:532-537 on [..., C]."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.fork = getattr(x, "bg_fork", None)
        x = _c(x)
        C = x.shape[-1]
        if KinkProbe.sites is not None:
            KinkProbe.sites.append((getattr(alpha, "bg_name", None), "act", x.detach().float().clone()))
        y = torch.empty_like(x)
        if x.dtype == torch.float32:
            check(lib().bg_prelu_fwd(f32(x), f32(alpha), f32(y), x.numel() // C, C, stream()))
        else:
            check(lib().bg_prelu_fwd_t(act(x), dt(x), f32(alpha), act(y), dt(y), x.numel() // C, C, stream()))
        ctx.x, ctx.alpha = x, alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        x, alpha = ctx.x, ctx.alpha
        C = x.shape[-1]
        rows = x.numel() // C
        L = lib()
        dx = None
        f32s = x.dtype == torch.float32 and dy.dtype == torch.float32
        if ctx.needs_input_grad[0]:
            if f32s:
                dx = torch.empty_like(x)
                check(L.bg_prelu_bwd(f32(x), f32(dy), f32(alpha), f32(dx), None, rows, C, stream()))
            else:
                dx, add = _fork_target(ctx.fork, x)
                check(L.bg_prelu_bwd_t(act(x), dt(x), act(dy), dt(dy), f32(alpha), act(dx), None, act(dx) if add else None,
                                       rows, C, stream()))
                _fork_done(ctx.fork, dx)

        def prod(out):
            out.zero_()
            if f32s:
                check(L.bg_prelu_bwd(f32(x), f32(dy), f32(alpha), None, f32(out), rows, C, stream()))
            else:
                check(L.bg_prelu_bwd_t(act(x), dt(x), act(dy), dt(dy), f32(alpha), None, f32(out), None, rows, C, stream()))
        da = param_grad(alpha, ctx.needs_input_grad[1], prod)
        ctx.x = None
        return dx, da


# ------------------------------------------------------------------------------------------
# pooling / small elementwise
# ------------------------------------------------------------------------------------------
class MaxPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W_, C = x.shape
        y = torch.empty((N, H // 2, W_ // 2, C), dtype=x.dtype, device=x.device)
        if x.dtype == torch.float32:
            check(lib().bg_maxpool2_fwd(f32(x), f32(y), N, H, W_, C, stream()))
        else:
            check(lib().bg_maxpool2_fwd_t(act(x), act(y), dt(x), N, H, W_, C, stream()))
        ctx.x = x
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        x = ctx.x
        N, H, W_, C = x.shape
        dx = torch.empty_like(x)
        if x.dtype == torch.float32:
            check(lib().bg_maxpool2_bwd(f32(x), f32(dy), f32(dx), N, H, W_, C, stream()))
        else:
            check(lib().bg_maxpool2_bwd_t(act(x), act(cast(dy, x.dtype)), act(dx), dt(x), N, H, W_, C, stream()))
        ctx.x = None
        return dx


class AvgPool2Fn(Function):
    """tf.layers.average_pooling2d(pool 2, strides 2) on even H, W (ops.py:512-514)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W_, C = x.shape
        y = torch.empty((N, H // 2, W_ // 2, C), dtype=torch.float32, device=x.device)
        check(lib().bg_box2_down(f32(x), f32(y), N, H, W_, C, 0.25, stream()))
        ctx.shape = (N, H, W_, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H, W_, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        check(lib().bg_box2_up(f32(dy), f32(dx), N, H // 2, W_ // 2, C, 0.25, stream()))
        return dx


class UpSample2Fn(Function):
    """tf.image.resize_nearest_neighbor to twice the size (ops.py:516-519)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W_, C = x.shape
        y = torch.empty((N, 2 * H, 2 * W_, C), dtype=torch.float32, device=x.device)
        check(lib().bg_box2_up(f32(x), f32(y), N, H, W_, C, 1.0, stream()))
        ctx.shape = (N, H, W_, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H, W_, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        check(lib().bg_box2_down(f32(dy), f32(dx), N, 2 * H, 2 * W_, C, 1.0, stream()))
        return dx


class SumPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W_, C = x.shape
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        if x.dtype == torch.float32:
            check(lib().bg_sum_pool_fwd(f32(x), f32(y), N, H * W_, C, stream()))
        else:
            check(lib().bg_sum_pool_fwd_t(act(x), dt(x), f32(y), N, H * W_, C, stream()))
        ctx.shape = x.shape
        ctx.xdt = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H, W_, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=ctx.xdt, device=dy.device)
        if ctx.xdt == torch.float32:
            check(lib().bg_sum_pool_bwd(f32(dy), f32(dx), N, H * W_, C, stream()))
        else:
            check(lib().bg_sum_pool_bwd_t(f32(dy), act(dx), dt(dx), N, H * W_, C, stream()))
        return dx


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        check(lib().bg_tanh_fwd(f32(x), f32(y), x.numel(), stream()))
        ctx.y = y
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().bg_tanh_bwd(f32(ctx.y), f32(dy), f32(dx), dy.numel(), stream()))
        ctx.y = None
        return dx


class ScaleFn(Function):
    """a * x with a host scalar a (fp32 tensors)."""

    @staticmethod
    def forward(ctx, x, a):
        ctx.a = float(a)
        y = torch.empty_like(_c(x))
        check(lib().bg_axpby(f32(_c(x)), ctx.a, f32(y), 0.0, x.numel(), stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().bg_axpby(f32(dy), ctx.a, f32(dx), 0.0, dy.numel(), stream()))
        return dx, None


class AddFn(Function):
    """a + b (residual sums, ops.py:198,266,313)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        return add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class GradMarkFn(Function):
    """Identity whose backward calls ``callback(tag)``: when the engine reaches it, every node created AFTER it in the
    forward pass has already run its backward (autograd runs ready nodes in reverse creation order), i.e. all layers
    behind this point have final weight gradients.  Data parallelism uses it to start the gradient exchange of the
    finished part of the network while the rest of backward is still running (model.BigGAN._g_bucket_done)."""

    @staticmethod
    def forward(ctx, x, callback, tag):
        ctx.cb, ctx.tag = callback, tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        ctx.cb(ctx.tag)
        return dy, None, None


class CastFn(Function):
    """Element-type conversion between the fp32 and the bf16 parts of a bf16-resident network (the image layers and
    the attention core stay fp32); the gradient is converted back."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return cast(_c(dy), ctx.src), None


class ForkFn(Function):
    """Identity with two outputs for a tensor consumed by two branches (block input -> main path and
    skip path, ops.py:253/263): the two incoming gradients are summed by a HIP kernel instead of the
    autograd engine's own accumulation."""

    @staticmethod
    def forward(ctx, x, n, state=None):
        ctx.state = state
        return tuple(x.detach().view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        if ctx.state is not None:
            ctx.state.buf = None
        uniq = []
        for g in gs:             # branches that accumulated into the shared buffer (ForkState) hand back the same storage
            if g is not None and not any(g.data_ptr() == u.data_ptr() and g.shape == u.shape for u in uniq):
                uniq.append(_c(g))
        if not uniq:
            return None, None, None
        if len(uniq) == 1:
            return uniq[0], None, None
        out = add(uniq[0], uniq[1])
        for g in uniq[2:]:
            out = add(out, g, out=out)
        return out, None, None


class ScaleAddFn(Function):
    """gamma * o + x with a learned scalar gamma (ops.py:486-490)."""

    @staticmethod
    def forward(ctx, o, gamma, x):
        ctx.fork = getattr(x, "bg_fork", None)
        o, x = _c(o), _c(x)
        ctx.o_dtype = o.dtype
        if o.dtype != x.dtype:
            o = cast(o, x.dtype)
        y = torch.empty_like(x)
        if x.dtype == torch.float32:
            check(lib().bg_scale_add(f32(o), f32(gamma), f32(x), f32(y), x.numel(), stream()))
        else:
            check(lib().bg_lincomb_t(act(o), f32(gamma), 0.0, act(x), 1.0, act(y), dt(x), x.numel(), stream()))
        ctx.o, ctx.gamma = o, gamma
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        o, gamma = ctx.o, ctx.gamma
        L = lib()
        do = None
        if ctx.needs_input_grad[0]:
            do = torch.empty_like(o)
            if dy.dtype == torch.float32:
                check(L.bg_scale_dev(f32(dy), f32(gamma), f32(do), dy.numel(), stream()))
            else:
                check(L.bg_lincomb_t(act(dy), f32(gamma), 0.0, None, 0.0, act(do), dt(dy), dy.numel(), stream()))

        def prod(out):
            out.zero_()
            if dy.dtype == torch.float32:
                check(L.bg_dot(f32(dy), f32(o), f32(out), dy.numel(), stream()))
            else:
                check(L.bg_dot_t(act(dy), act(o), dt(dy), f32(out), dy.numel(), stream()))
        dg = param_grad(gamma, ctx.needs_input_grad[1], prod)
        ctx.o = None
        if do is not None:
            do = cast(do, ctx.o_dtype)
        # the residual's gradient IS dy: when x is a forked tensor, the other branch's input-gradient kernel adds into this
        # very buffer (every read of dy above is already enqueued; the tensor was handed to this node alone) and
        # ForkFn.backward then sees one storage twice - no separate sum
        if ctx.needs_input_grad[2] and dy.dtype == ctx.o_dtype:
            _fork_done(ctx.fork, dy)
        return do, dg, dy


# ------------------------------------------------------------------------------------------
# tangent (forward-mode) maps of the discriminator's non-linear ops, each with its own backward: the gradient
# penalty's parameter gradient is the gradient of a directional derivative of D (include/biggan_hip.h, "Gradient
# penalty").  Linear ops (conv, dense, pooling sums, residual adds) are their own tangent maps.
# ------------------------------------------------------------------------------------------
class BnTangentFn(Function):
    """Forward-mode tangent of training-mode batch norm (ops.py:580-585 inside the gradient penalty, BigGAN.py:717-742):
    ydot = gamma rstd (xdot - mean(xdot) - xhat mean(xdot xhat)), and its backward w.r.t. xdot, x (through the batch
    statistics) and gamma - include/biggan_hip.h "Forward-mode tangent of TRAINING-mode batch norm".  ``mean`` / ``rstd``
    are the statistics the primal pass just computed from the same x; sums are all-reduced under data parallelism."""

    @staticmethod
    def forward(ctx, xdot, x, gamma, mean, rstd, count, reduce_fn):
        xdot, x = _c(xdot), _c(x)
        C = x.shape[-1]
        rows = x.numel() // C
        L = lib()
        dev = x.device
        sums = torch.empty(3 * C, dtype=torch.float64, device=dev)
        check(L.bg_chan_dots3(f32(xdot), f32(x), None, hip.ptr(sums), rows, C, stream()))
        if reduce_fn is not None:
            reduce_fn(sums)
        cf = torch.empty(3 * C, dtype=torch.float32, device=dev)
        m12 = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(L.bg_bn_tangent_fwd_coefs(hip.ptr(sums), float(count), f32(mean), f32(rstd), f32(gamma), f32(cf), f32(m12), C,
                                        stream()))
        y = torch.empty_like(x)
        check(L.bg_chan_lincomb3(f32(xdot), f32(cf[:C]), f32(x), f32(cf[C:2 * C]), None, None, f32(cf[2 * C:]), f32(y),
                                 rows, C, stream()))
        ctx.saved = (xdot, x, gamma, mean, rstd, m12, float(count), reduce_fn)
        return y

    @staticmethod
    def backward(ctx, s_):
        s_ = _c(s_)
        xdot, x, gamma, mean, rstd, m12, count, reduce_fn = ctx.saved
        C = x.shape[-1]
        rows = x.numel() // C
        L = lib()
        dev = x.device
        sums = torch.empty(3 * C, dtype=torch.float64, device=dev)
        check(L.bg_chan_dots3(f32(s_), f32(x), f32(xdot), hip.ptr(sums), rows, C, stream()))
        if reduce_fn is not None:
            reduce_fn(sums)
        cd = torch.empty(3 * C, dtype=torch.float32, device=dev)
        cx = torch.empty(4 * C, dtype=torch.float32, device=dev)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        check(L.bg_bn_tangent_bwd_coefs(hip.ptr(sums), count, f32(mean), f32(rstd), f32(gamma), f32(m12), f32(cd), f32(cx),
                                        f32(dg), C, stream()))
        dxdot = dx = None
        if ctx.needs_input_grad[0]:
            dxdot = torch.empty_like(x)
            check(L.bg_chan_lincomb3(f32(s_), f32(cd[:C]), f32(x), f32(cd[C:2 * C]), None, None, f32(cd[2 * C:]),
                                     f32(dxdot), rows, C, stream()))
        if ctx.needs_input_grad[1]:
            dx = torch.empty_like(x)
            check(L.bg_chan_lincomb3(f32(s_), f32(cx[:C]), f32(x), f32(cx[C:2 * C]), f32(xdot), f32(cx[2 * C:3 * C]),
                                     f32(cx[3 * C:]), f32(dx), rows, C, stream()))
        dgam = param_grad(gamma, ctx.needs_input_grad[2], lambda out: out.copy_(dg))
        ctx.saved = None
        return dxdot, dx, dgam, None, None, None, None


class PReluTangentFn(Function):
    """ydot = xdot * prelu'(x; alpha)."""

    @staticmethod
    def forward(ctx, xdot, x, alpha):
        xdot, x = _c(xdot), _c(x)
        C = x.shape[-1]
        y = torch.empty_like(x)
        check(lib().bg_prelu_bwd(f32(x), f32(xdot), f32(alpha), f32(y), None, x.numel() // C, C, stream()))
        ctx.xdot, ctx.x, ctx.alpha = xdot, x, alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        xdot, x, alpha = ctx.xdot, ctx.x, ctx.alpha
        C = x.shape[-1]
        rows = x.numel() // C
        L = lib()
        dxdot = None
        if ctx.needs_input_grad[0]:
            dxdot = torch.empty_like(x)
            check(L.bg_prelu_bwd(f32(x), f32(dy), f32(alpha), f32(dxdot), None, rows, C, stream()))

        def prod(out):
            check(L.bg_prelu_tangent_dalpha(f32(x), f32(xdot), f32(dy), f32(out), rows, C, stream()))
        da = param_grad(alpha, ctx.needs_input_grad[2], prod)
        ctx.xdot = ctx.x = None
        return dxdot, None, da          # piecewise linear: no gradient reaches x through the slope


class MaxPool2TangentFn(Function):
    """ydot = xdot at the arg-max of x's 2x2 window."""

    @staticmethod
    def forward(ctx, xdot, x):
        xdot, x = _c(xdot), _c(x)
        N, H, W_, C = x.shape
        y = torch.empty((N, H // 2, W_ // 2, C), dtype=torch.float32, device=x.device)
        check(lib().bg_maxpool2_gather(f32(x), f32(xdot), f32(y), N, H, W_, C, stream()))
        ctx.x = x
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        x = ctx.x
        N, H, W_, C = x.shape
        d = torch.empty_like(x)
        check(lib().bg_maxpool2_bwd(f32(x), f32(dy), f32(d), N, H, W_, C, stream()))
        ctx.x = None
        return d, None


class BmmFn(Function):
    """Batched a @ b or a @ b^T on [B, M, K] x [B, K, N] (resp. [B, N, K]) with gradients to both operands."""

    @staticmethod
    def forward(ctx, a, b, trans_b):
        a, b = _c(a), _c(b)
        B, M, K = a.shape
        N = b.shape[1] if trans_b else b.shape[2]
        y = torch.empty((B, M, N), dtype=torch.float32, device=a.device)
        ldb = K if trans_b else N
        gemm(a, b, y, M, N, K, K, ldb, N, transB=trans_b, batch=B, sA=M * K, sB=b.shape[1] * b.shape[2], sC=M * N)
        ctx.a, ctx.b, ctx.trans_b = a, b, trans_b
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        a, b, tb = ctx.a, ctx.b, ctx.trans_b
        B, M, K = a.shape
        N = dy.shape[2]
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)        # dy @ b (trans_b) or dy @ b^T
            gemm(dy, b, da, M, K, N, N, K if tb else N, K, transB=not tb, batch=B, sA=M * N, sB=b.shape[1] * b.shape[2],
                 sC=M * K)
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            if tb:                          # b [B,N,K]: dy^T @ a
                gemm(dy, a, db, N, K, M, N, K, K, transA=True, batch=B, sA=M * N, sB=M * K, sC=N * K)
            else:                           # b [B,K,N]: a^T @ dy
                gemm(a, dy, db, K, N, M, K, N, N, transA=True, batch=B, sA=M * K, sB=M * N, sC=K * N)
        ctx.a = ctx.b = None
        return da, db, None


class SoftmaxFn(Function):
    """Row softmax over the last axis with P kept (the materialised attention of the gradient-penalty passes)."""

    @staticmethod
    def forward(ctx, s_):
        s_ = _c(s_)
        p = torch.empty_like(s_)
        cols = s_.shape[-1]
        check(lib().bg_softmax_fwd(f32(s_), f32(p), s_.numel() // cols, cols, stream()))
        ctx.p = p
        return p

    @staticmethod
    def backward(ctx, dp):
        dp = _c(dp)
        p = ctx.p
        cols = p.shape[-1]
        ds = torch.empty_like(p)
        check(lib().bg_softmax_bwd(f32(p), f32(dp), f32(ds), p.numel() // cols, cols, stream()))
        ctx.p = None
        return ds


class SoftmaxTangentFn(Function):
    """pdot = p * (sdot - sum_j p_j sdot_j), differentiable w.r.t. both p and sdot."""

    @staticmethod
    def forward(ctx, p, sdot):
        p, sdot = _c(p), _c(sdot)
        cols = p.shape[-1]
        y = torch.empty_like(p)
        check(lib().bg_softmax_bwd(f32(p), f32(sdot), f32(y), p.numel() // cols, cols, stream()))
        ctx.p, ctx.sdot = p, sdot
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        p, sdot = ctx.p, ctx.sdot
        cols = p.shape[-1]
        dp, dsdot = torch.empty_like(p), torch.empty_like(p)
        check(lib().bg_softmax_tangent_bwd(f32(p), f32(sdot), f32(g), f32(dp), f32(dsdot), p.numel() // cols, cols,
                                           stream()))
        ctx.p = ctx.sdot = None
        return dp, dsdot


class GpSurrogateFn(Function):
    """Carries the gradient penalty's VALUE forward and, backward, hands a unit gradient to the directional
    derivatives ``fdot`` whose parameter gradient is the penalty's (see "Gradient penalty" in the header)."""

    @staticmethod
    def forward(ctx, fdot, value):
        ctx.shape = fdot.shape
        return value.detach().clone()

    @staticmethod
    def backward(ctx, dy):
        ones = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        ones.fill_(1.0)
        g = torch.empty_like(ones)
        check(lib().bg_scale_dev(f32(ones), f32(_c(dy).reshape(-1)[:1]), f32(g), ones.numel(), stream()))
        return g, None


# ------------------------------------------------------------------------------------------
# DiffAugment
# ------------------------------------------------------------------------------------------
class DiffAugmentFn(Function):
    """DiffAugment_tf.py:8-73 with explicit draws (device tensors)."""

    @staticmethod
    def forward(ctx, x, u_b, u_s, u_c, t_x, t_y, o_x, o_y, policy_bits):
        x = _c(x)
        N, S, S2, C = x.shape
        assert S == S2
        y = torch.empty_like(x)
        ws = torch.empty(N, dtype=torch.float64, device=x.device)
        check(lib().bg_diffaugment_fwd(f32(x), f32(y), f32(u_b), f32(u_s), f32(u_c), i32(t_x), i32(t_y), i32(o_x),
                                       i32(o_y), N, S, C, policy_bits, hip.ptr(ws), stream()))
        ctx.draws = (u_s, u_c, t_x, t_y, o_x, o_y)
        ctx.policy = policy_bits
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        u_s, u_c, t_x, t_y, o_x, o_y = ctx.draws
        N, S, _, C = ctx.shape
        dx = torch.empty_like(dy)
        ws = torch.empty(N, dtype=torch.float64, device=dy.device)
        check(lib().bg_diffaugment_bwd(f32(dy), f32(dx), f32(u_s), f32(u_c), i32(t_x), i32(t_y), i32(o_x), i32(o_y),
                                       N, S, C, ctx.policy, hip.ptr(ws), stream()))
        return dx, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------
class HingeDLossFn(Function):
    """discriminator_loss('hinge', real, fake, flood) (ops.py:788-797)."""

    @staticmethod
    def forward(ctx, real, fake, flood, reduce_fn, world):
        real, fake = _c(real), _c(fake)
        n = real.numel()
        L = lib()
        dev = real.device
        sums = zeros(2, torch.float32, dev)
        check(L.bg_hinge_d_sums(f32(real), f32(fake), f32(sums), n, stream()))
        if reduce_fn is not None:
            reduce_fn(sums)
        d_real = torch.empty_like(real)
        d_fake = torch.empty_like(fake)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(L.bg_hinge_d_grad(f32(real), f32(fake), f32(sums), float(n * world), float(flood or 0.0), f32(d_real),
                                f32(d_fake), f32(loss), n, stream()))
        ctx.d_real, ctx.d_fake = d_real, d_fake
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        L = lib()
        dr = torch.empty_like(ctx.d_real)
        df = torch.empty_like(ctx.d_fake)
        check(L.bg_scale_dev(f32(ctx.d_real), f32(g), f32(dr), dr.numel(), stream()))
        check(L.bg_scale_dev(f32(ctx.d_fake), f32(g), f32(df), df.numel(), stream()))
        return dr, df, None, None, None


class HingeGLossFn(Function):
    """generator_loss('hinge', fake, real, flood) (ops.py:832-840)."""

    @staticmethod
    def forward(ctx, fake, flood, reduce_fn, world):
        fake = _c(fake)
        n = fake.numel()
        L = lib()
        dev = fake.device
        sums = zeros(2, torch.float32, dev)
        check(L.bg_hinge_g_sums(f32(fake), f32(sums), n, stream()))
        if reduce_fn is not None:
            reduce_fn(sums)
        d_fake = torch.empty_like(fake)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(L.bg_hinge_g_grad(f32(sums), float(n * world), float(flood or 0.0), f32(d_fake), f32(loss), n,
                                stream()))
        ctx.d_fake = d_fake
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        df = torch.empty_like(ctx.d_fake)
        check(lib().bg_scale_dev(f32(ctx.d_fake), f32(g), f32(df), df.numel(), stream()))
        return df, None, None, None


GAN_LOSS_KINDS = {"hinge": 0, "lsgan": 1, "gan": 2, "ra-lsgan": 3, "ra-gan": 4, "ra-hinge": 5,
                  "dragan": 2, "ra-dragan": 4, "wgan-gp": 6, "wgan-lp": 6}          # ops.py:757,771,775 ('wgan' substring)


class GanLossFn(Function):
    """discriminator_loss / generator_loss for the non-penalty loss types (ops.py:753-840):
    lsgan, gan, ra-lsgan, ra-gan, ra-hinge (hinge too; the model uses the dedicated hinge kernels for it)."""

    @staticmethod
    def forward(ctx, real, fake, kind, generator, flood, reduce_fn, world):
        fake = _c(fake)
        nf = fake.numel()
        nr = 0 if real is None else real.numel()
        real = None if real is None else _c(real)
        L = lib()
        dev = fake.device
        sums = zeros(2, torch.float32, dev)
        check(L.bg_gan_loss_means(f32(real), f32(fake), f32(sums), nr, nf, stream()))
        if reduce_fn is not None:
            reduce_fn(sums)
        tsums = zeros(4, torch.float32, dev)
        check(L.bg_gan_loss_terms(kind, int(generator), f32(real), f32(fake), f32(sums), float(nr * world),
                                  float(nf * world), f32(tsums), nr, nf, stream()))
        if reduce_fn is not None:
            reduce_fn(tsums)
        d_real = None if real is None else torch.empty_like(real)
        d_fake = torch.empty_like(fake)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(L.bg_gan_loss_grad(kind, int(generator), f32(real), f32(fake), f32(sums), f32(tsums), float(nr * world),
                                 float(nf * world), float(flood or 0.0), f32(d_real), f32(d_fake), f32(loss), nr, nf,
                                 stream()))
        ctx.d_real, ctx.d_fake = d_real, d_fake
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        outs = []
        for d, need in ((ctx.d_real, ctx.needs_input_grad[0]), (ctx.d_fake, ctx.needs_input_grad[1])):
            if d is None or not need:
                outs.append(None)
                continue
            o = torch.empty_like(d)
            check(lib().bg_scale_dev(f32(d), f32(g), f32(o), o.numel(), stream()))
            outs.append(o)
        return outs[0], outs[1], None, None, None, None, None


class SigmoidCeLossFn(Function):
    """cls_loss_fn('logistic', w)(truth, answer) * loss_weight (utils.py:366-369, BigGAN.py:853,894):
    mean over the GLOBAL batch x labels of the weighted sigmoid cross-entropy."""

    @staticmethod
    def forward(ctx, truth, logits, weights, loss_weight, reduce_fn, world):
        truth, logits = _c(truth), _c(logits)
        B, n = logits.shape
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dl = torch.empty_like(logits)
        check(lib().bg_sigmoid_ce(f32(logits), f32(truth), f32(weights), float(loss_weight) / float(B * world * n),
                                  f32(loss), f32(dl), B, n, stream()))
        if reduce_fn is not None:
            reduce_fn(loss)
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        out = torch.empty_like(ctx.dl)
        check(lib().bg_scale_dev(f32(ctx.dl), f32(g), f32(out), out.numel(), stream()))
        return None, out, None, None, None, None


class OrthoCosineRegFn(Function):
    """orthogonal_regularizer(scale, 'ortho_cosine')(w) (utils.py:180-235): scale * l2_loss(R).
    The weight gradient dW = W (dA + dA^T) is produced in backward."""

    @staticmethod
    def forward(ctx, w, scale, kind="ortho_cosine"):
        w = _c(w)
        c = w.shape[-1]
        rows = w.numel() // c
        W2 = w.view(rows, c)
        dev = w.device
        ctx.lowrank = 2 * rows <= c and kind == "ortho_cosine"
        if ctx.lowrank:
            # wide kernel: O(rows^2 c) form, the c x c Gram matrix is never materialised
            L = lib()
            G = torch.empty((rows, rows), dtype=torch.float32, device=dev)
            gemm(W2, W2, G, rows, rows, c, c, c, rows, transB=True)                     # G = W W^T
            s = torch.empty(rows, dtype=torch.float32, device=dev)
            check(L.bg_gemv_rows(f32(W2), None, f32(s), rows, c, stream()))              # s = W 1
            P = torch.empty((rows, c), dtype=torch.float32, device=dev)
            gemm(G, W2, P, rows, c, rows, rows, c, c)                                    # P = G W
            ab = torch.empty(2 * c, dtype=torch.float32, device=dev)
            Wb = torch.empty((rows, c), dtype=torch.float32, device=dev)
            loss = zeros(1, torch.float32, dev)
            check(L.bg_ortho_lowrank_cols(f32(W2), f32(P), f32(s), float(scale), f32(ab), f32(Wb), f32(loss),
                                          rows, c, stream()))
            Wa = torch.empty(rows, dtype=torch.float32, device=dev)
            check(L.bg_gemv_rows(f32(W2), f32(ab), f32(Wa), rows, c, stream()))          # W alpha
            H = torch.empty((rows, rows), dtype=torch.float32, device=dev)
            gemm(Wb, W2, H, rows, rows, c, c, c, rows, transB=True)                      # H = W diag(beta) W^T
            dW = torch.empty((rows, c), dtype=torch.float32, device=dev)
            gemm(H, W2, dW, rows, c, rows, rows, c, c)                                   # H W
            check(L.bg_ortho_lowrank_finish(f32(dW), f32(P), f32(s), f32(Wa), f32(ab), rows, c, stream()))
            ctx.w, ctx.dW = w, dW
            return loss
        A = torch.empty((c, c), dtype=torch.float32, device=dev)
        wn = getattr(w, "bg_sn_wn", None)
        pk = getattr(wn, "bg_pack_p", None) if wn is not None else None
        if pk is not None and getattr(wn, "bg_run_stamp", None) is not current_run_stamp():
            pk = None              # packs / sigma of an earlier run (no spectral-norm prefetch in this one): fp32 Gram instead
        if (Precision.resident and pk is not None and kind == "ortho_cosine" and c % 8 == 0
                and os.environ.get("BG_REG_GRAM", "") != "fp32"):              # (BG_REG_GRAM=fp32: A/B switch)
            # bf16-resident mode: W^T W = sigma^2 (W/sigma)^T (W/sigma) from the packed bf16 copy of this run's spectral
            # norm - half the bytes of the fp32 weight, which the staged form reads twice
            L = lib()
            nb = int(L.bg_gram16_workspace_bytes(rows, c))
            ws = workspace(nb, dev)
            check(L.bg_gram16(act(pk), rows, c, c, f32(A), f32(ws), nb, stream()))
            check(L.bg_scale_dev(f32(A), f32(wn.bg_sigma), f32(A), A.numel(), stream()))
            check(L.bg_scale_dev(f32(A), f32(wn.bg_sigma), f32(A), A.numel(), stream()))
        else:
            gemm(W2, W2, A, c, c, rows, c, c, c, transA=True)           # A = W^T W
        loss = zeros(1, torch.float32, dev)
        dA = torch.empty((c, c), dtype=torch.float32, device=dev)
        if kind == "ortho":                                              # utils.py:199-200: reg = A - I
            check(lib().bg_ortho_identity_fwd_bwd(f32(A), float(scale), f32(loss), f32(dA), c, stream()))
        else:
            check(lib().bg_ortho_cosine_fwd_bwd(f32(A), float(scale), f32(loss), f32(dA), c, stream()))
        ctx.w, ctx.dA = w, dA
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        if ctx.lowrank:
            w, dWs = ctx.w, ctx.dW

            def prod_lr(out):
                check(lib().bg_scale_dev(f32(dWs), f32(g), f32(out), dWs.numel(), stream()))
            dw = param_grad(w, ctx.needs_input_grad[0], prod_lr)
            ctx.w = ctx.dW = None
            return dw, None, None
        w, dA = ctx.w, ctx.dA
        c = w.shape[-1]
        rows = w.numel() // c
        W2 = w.view(rows, c)

        def prod(out):
            o2 = out.view(rows, c)
            S_ = torch.empty_like(dA)
            L = lib()
            check(L.bg_symmetrize(f32(dA), f32(S_), c, stream()))
            wn = getattr(w, "bg_sn_wn", None)
            pk = getattr(wn, "bg_pack_p", None) if wn is not None else None
            if (Precision.resident and pk is not None and getattr(wn, "bg_run_stamp", None) is current_run_stamp()
                    and c % 8 == 0 and rows % 8 == 0 and os.environ.get("BG_REG_GRAM", "") != "fp32"):
                # bf16-resident mode: W (dA + dA^T) = sigma (W / sigma) S on the bf16-resident GEMM - the packed copy of
                # this run's spectral norm as a [rows, c] "image" of a 1 x 1 convolution whose kernel is S (symmetric, so its
                # packed [n][c] form is S itself).  Same operand precision as the staged kernel it replaces (which rounds the
                # fp32 W and S to bf16 while staging), about twice its rate.
                S16 = cast(S_, BF16)
                gs = torch.empty(1, dtype=torch.float32, device=dA.device)
                check(L.bg_scale_dev(f32(g), f32(wn.bg_sigma), f32(gs), 1, stream()))
                cd = hip.conv_desc(1, rows, 1, c, rows, 1, c, 1, 1, 0, hip.PAD_ZERO, hip.COMPUTE_BF16, hip.BF16, hip.F32, 1)
                ws, nb = hip.scratch(L.bg_conv2d_fwd_workspace_bytes, cd, dA.device)
                check(L.bg_conv2d_fwd(cd, act(pk), act(S16), None, f32(gs), act(o2), 0, f32(ws), nb, stream()))
                return
            gemm(W2, S_, o2, rows, c, c, c, c, c, alpha_dev=g)                                  # W dA + W dA^T = W (dA + dA^T)
        dw = param_grad(w, ctx.needs_input_grad[0], prod)
        ctx.w = ctx.dA = None
        return dw, None, None


class L2RegFn(Function):
    """tf.contrib.layers.l2_regularizer(scale)(w) = scale * sum(w^2) / 2 (BigGAN.py:268-270)."""

    @staticmethod
    def forward(ctx, w, scale):
        w = _c(w)
        loss = torch.zeros(1, dtype=torch.float32, device=w.device)
        check(lib().bg_dot(f32(w), f32(w), f32(loss), w.numel(), stream()))
        axpby(loss, 0.0, loss, 0.5 * float(scale))
        ctx.w, ctx.scale = w, float(scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        w, scale = ctx.w, ctx.scale

        def prod(out):
            check(lib().bg_scale_dev(f32(w), f32(g), f32(out), w.numel(), stream()))
            axpby(out, 0.0, out, scale)
        dw = param_grad(w, ctx.needs_input_grad[0], prod)
        ctx.w = None
        return dw, None
