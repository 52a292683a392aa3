"""Operator library with the reference's ``ops.py`` function names, signatures, ``opt`` keys and
scope naming (``/root/reference/ops.py``), executing eagerly on MI355X through libbiggan_hip.so.

Tensors are NHWC fp32 CUDA tensors; parameters are created / reused implicitly by scope path
(``scope.py``).  A tensor on the ``meta`` device only propagates shapes and registers variables
(used to build the parameter manifest without a GPU); any other non-CUDA tensor raises.
"""
import os

import torch

from . import functional as Fn
from . import hip
from . import scope as S
from .scope import variable_scope, get_variable
from .utils import round_up  # noqa: F401  (ops.py:16)

##################################################################################
# Initialization (ops.py:13-14)
##################################################################################
weight_init = S.truncated_normal_initializer(mean=0.0, stddev=0.02)
gan_dtype = torch.float32


##################################################################################
# per-run context: spectral-norm cache, regularisation losses, data-parallel hooks
##################################################################################
class _Run:
    def __init__(self):
        self.sn_cache = {}       # variable name -> w / sigma of this run (one power iteration per run)
        self.sn_prefetched = set()
        self.reg_losses = []     # tf.losses.get_regularization_losses()
        self.reg_seen = set()
        self.reduce_fn = None    # in-place SUM all-reduce of a small fp32 tensor across DP ranks
        self.world = 1
        self.rank = 0
        self.reg_owner = None    # data parallel: variable name -> rank that evaluates its regulariser
        self.dense_cache = {}    # kernel name -> (x, y) of a dense projection computed ahead by a grouped launch


_run = _Run()


def begin_run(reduce_fn=None, world=1, rank=0, reg_owner=None, zero_pool=None):
    """Start of one ``sess.run``: forget cached spectral norms and regularisation losses; ``zero_pool``
    (functional.ZeroPool) serves the run's zero-initialised accumulators from one buffer cleared by one fill."""
    _run.sn_cache = {}
    _run.dense_cache = {}
    Fn.new_run_stamp()
    Fn.set_zero_pool(zero_pool)
    if zero_pool is not None:
        zero_pool.begin_run()
    _run.sn_prefetched = set()
    _run.reg_losses = []
    _run.reg_seen = set()
    _run.reduce_fn = reduce_fn
    _run.world = world
    _run.rank = rank
    _run.reg_owner = reg_owner


def sn_prefetch(batch):
    """Run the power iteration of every spectrally-normalised weight of a network in one multi-tensor
    call (functional.SnBatch) and put the results where ``spectral_norm`` looks first."""
    if batch is None or id(batch) in _run.sn_prefetched:
        return                  # a second instantiation of the network in the same run reuses the first
    _run.sn_prefetched.add(id(batch))
    needs_base = torch.is_grad_enabled()
    for w, wn in zip(batch.w, batch.forward(Fn.current_run_stamp())):
        _run.sn_cache[w.bg_name] = (wn, needs_base and w.requires_grad)


def get_regularization_losses():
    return list(_run.reg_losses)


def _is_meta(x):
    return x.device.type == "meta"


class Dual:
    """A (primal, tangent) pair travelling through the discriminator: the forward-mode pass of the gradient
    penalty (BigGAN.py:717-742; see "Gradient penalty" in include/biggan_hip.h).  Operators that receive a Dual
    apply themselves to ``p`` and their tangent map to ``t``; linear operators are their own tangent map (without
    the bias)."""

    def __init__(self, p, t):
        assert p.shape == t.shape, (p.shape, t.shape)
        self.p, self.t = p, t

    shape = property(lambda self: self.p.shape)
    device = property(lambda self: self.p.device)
    requires_grad = property(lambda self: self.p.requires_grad or self.t.requires_grad)
    is_cuda = property(lambda self: self.p.is_cuda)

    def dim(self):
        return self.p.dim()

    def reshape(self, *shape):
        return Dual(self.p.reshape(*shape), self.t.reshape(*shape))

    def __getitem__(self, idx):
        return Dual(self.p[idx], self.t[idx])


def _is_dual(x):
    return isinstance(x, Dual)


def _meta(shape):
    return torch.empty(tuple(int(s) for s in shape), device="meta")


def _regularize(w, regularizer):
    """tf.get_variable(..., regularizer=...) adds regularizer(w) to the graph's losses once per variable."""
    if regularizer is None or not torch.is_grad_enabled() or not w.requires_grad:
        return
    if w.bg_name in _run.reg_seen:
        return
    _run.reg_seen.add(w.bg_name)
    S.default_store().reg_shapes.setdefault(w.bg_name, tuple(int(d) for d in w.shape))
    if _is_meta(w) or not w.is_cuda:
        return
    if _run.reg_owner is not None and _run.reg_owner.get(w.bg_name, 0) != _run.rank:
        return                   # another rank evaluates this (batch-independent) term; gradients meet in the all-reduce
    _run.reg_losses.append(regularizer(w))


##################################################################################
# Layer
##################################################################################
def subpixel_conv(x, channels, opt, kernel=3, scale=2, use_bias=True, scope='subpixel_conv_0'):
    raise NotImplementedError("subpixel_conv (ops.py:23) is outside the default hot path")


def decode_kernel_sizes(str):
    raise NotImplementedError("mixed-kernel convolutions (ops.py:29) are outside the default hot path")


def encode_kernel_sizes(slices, ch_mul=1.0):
    raise NotImplementedError("mixed-kernel convolutions (ops.py:44) are outside the default hot path")


def _to(x, dtype):
    """Element-type conversion as a differentiable op (no-op when the type already matches)."""
    if _is_meta(x) or x.dtype == dtype:
        return x
    return Fn.CastFn.apply(x, dtype)


def _resident_out(y):
    """bf16-resident mode: a large activation produced by the fp32-tensor kernels (the image layers) joins the bf16
    part of the network."""
    if _is_meta(y):
        return y
    if Fn.Precision.resident and y.dtype == torch.float32 and y.dim() == 4 and y.shape[-1] % 8 == 0:
        return Fn.CastFn.apply(y, torch.bfloat16)
    return y


def conv(x, channels, opt, kernel=4, stride=2, pad=0, dilation=1, use_bias=True, scope='conv_0', _out_dtype=None,
         _accumulate_into=None):
    """ops.py:49-113.  ``_out_dtype`` (extension, bf16-resident mode): element type of the result.
    ``_accumulate_into`` (extension): add the result into an existing tensor in the kernel epilogue (fused residual sum)."""
    with variable_scope(scope) as full_scope:
        if isinstance(kernel, str):
            raise NotImplementedError("mixed-kernel convolutions (ops.py:52-59) are outside the default hot path")
        if dilation != 1:
            raise NotImplementedError("dilation != 1")
        N, H, W, Cin = x.shape
        pad_mode = hip.PAD_REFLECT
        pad_lo = pad_hi = 0
        if pad > 0:
            pad_type = opt.get("conv", {}).get("padding_type", 'reflect')
            if H % stride == 0:                       # ops.py:68-71
                tot = pad * 2
            else:
                tot = max(kernel - (H % stride), 0)
            pad_lo = int(tot // 2)
            pad_hi = int(tot - pad_lo)
            if pad_type == 'zero':                    # TF 'SAME'
                pad_mode = hip.PAD_ZERO
                out = -(-H // stride)
                tot = max((out - 1) * stride + kernel - H, 0)
                pad_lo, pad_hi = tot // 2, tot - tot // 2
            elif pad_type == 'reflect':
                pad_mode = hip.PAD_REFLECT
            else:
                raise ValueError("Unsupported padding type: " + str(pad_type))
        Ho = (H + pad_lo + pad_hi - kernel) // stride + 1
        Wo = (W + pad_lo + pad_hi - kernel) // stride + 1

        sn = opt.get("conv", {}).get("sn", True)
        reg = opt.get("conv", {}).get("regularizer", None) if 'generator' in full_scope else None
        w = get_variable("kernel", shape=[kernel, kernel, Cin, channels], initializer=weight_init, regularizer=reg)
        _regularize(w, reg)
        bias = None
        if sn:
            wk = spectral_norm(w, _shape_only=_is_meta(x))
            if use_bias:
                bias = get_variable("bias", [channels], initializer=S.constant_initializer(0.0))
        else:
            wk = w
            if use_bias:
                bias = get_variable("bias", [channels], initializer=S.constant_initializer(0.0))
        if _is_meta(x):
            return _meta((N, Ho, Wo, channels))
        if _is_dual(x):
            y = Dual(Fn.Conv2dFn.apply(x.p, wk, bias, stride, pad_lo, Ho, Wo, pad_mode),
                     Fn.Conv2dFn.apply(x.t, wk, None, stride, pad_lo, Ho, Wo, pad_mode))
            return y if _accumulate_into is None else _add(y, _accumulate_into)
        acc = _accumulate_into
        if acc is not None and os.environ.get("BG_FUSE_RESIDUAL", "1") == "0":       # A/B switch: separate add kernel
            y = Fn.Conv2dFn.apply(x, wk, bias, stride, pad_lo, Ho, Wo, pad_mode, _out_dtype)
            return _add(y if _out_dtype is not None else _resident_out(y), acc)
        if acc is not None:
            # the fused form needs an accumulator of the result's own type and shape that this call may overwrite
            want = (_out_dtype or (torch.bfloat16 if Fn._resident_ok(x, Cin, channels) else torch.float32))
            if (acc.dtype != want or not acc.is_contiguous() or Fn._thin_plan(x, Cin, channels) is not None
                    or (x.dtype == torch.bfloat16 and not Fn._resident_ok(x, Cin, channels))):
                y = Fn.Conv2dFn.apply(x, wk, bias, stride, pad_lo, Ho, Wo, pad_mode, _out_dtype)
                return _add(y if _out_dtype is not None else _resident_out(y), acc)
        y = Fn.Conv2dFn.apply(x, wk, bias, stride, pad_lo, Ho, Wo, pad_mode, _out_dtype, acc)
        return y if _out_dtype is not None else _resident_out(y)


def deconv(x, channels, opt, kernel=4, stride=2, padding='SAME', use_bias=True, scope='deconv_0', _accumulate_into=None,
           _stats=False):
    """ops.py:116-139.  ``_accumulate_into`` (extension): add the result into an existing tensor
    in the kernel epilogue (fused residual sum).  ``_stats`` (extension): the result feeds a batch norm - let the kernel
    leave the per-channel sums of its output on the tensor (``bg_bn_sums``) so the batch norm need not read it again."""
    with variable_scope(scope):
        N, H, W, Cin = x.shape
        if padding != 'SAME':
            raise NotImplementedError("deconv padding != 'SAME'")
        tot = max((H - 1) * stride + kernel - stride * H, 0)        # TF SAME transposed alignment
        pad_lo = tot // 2
        reg = opt.get("conv", {}).get("regularizer", None)          # attached regardless of scope (ops.py:127)
        w = get_variable("kernel", shape=[kernel, kernel, channels, Cin], initializer=weight_init, regularizer=reg)
        _regularize(w, reg)
        wk = spectral_norm(w, _shape_only=_is_meta(x)) if opt.get("conv", {}).get("sn", True) else w
        bias = None
        if use_bias:
            bias = get_variable("bias", [channels], initializer=S.constant_initializer(0.0))
        if _is_meta(x):
            return _meta((N, H * stride, W * stride, channels))
        if _accumulate_into is not None and _accumulate_into.dtype != x.dtype and x.dtype == torch.bfloat16:
            _accumulate_into = _to(_accumulate_into, x.dtype)
        box = [None] if (_stats and opt.get("is_training", True) and x.dtype == torch.bfloat16
                         and os.environ.get("BG_FUSE_BNSTATS", "1") != "0") else None
        y = _resident_out(Fn.Deconv2dFn.apply(x, wk, bias, stride, pad_lo, _accumulate_into, box))
        if box is not None and box[0] is not None:
            y.bg_bn_sums = box[0]
        return y


def get_variable_with_custom_lr(name, shape, regularizer, lrmul):
    """ops.py:141-146."""
    if lrmul != 1.0:
        raise NotImplementedError("lrmul != 1.0")
    return get_variable(name, shape, initializer=S.truncated_normal_initializer(mean=0.0, stddev=0.02 / lrmul),
                        regularizer=regularizer)


def fully_connected(x, units, opt, use_bias=True, lrmul=1.0, scope='fully_0'):
    """ops.py:148-175."""
    with variable_scope(scope) as full_scope:
        x = flatten(x)
        channels = x.shape[-1]
        reg = opt.get("fc_regularizer") if 'generator' in full_scope else None
        w = get_variable_with_custom_lr("kernel", shape=[channels, units], regularizer=reg, lrmul=lrmul)
        _regularize(w, reg)
        bias = get_variable("bias", [units], initializer=S.constant_initializer(0.0)) if use_bias else None
        wk = spectral_norm(w, _shape_only=_is_meta(x)) if opt.get("conv", {}).get("sn", True) else w
        if _is_meta(x):
            return _meta((x.shape[0], units))
        if _is_dual(x):
            return Dual(Fn.DenseFn.apply(x.p, wk, bias), Fn.DenseFn.apply(x.t, wk, None))
        hit = _run.dense_cache.pop(w.bg_name, None)
        if hit is not None and hit[0].data_ptr() == x.data_ptr() and hit[0].shape == x.shape and hit[0].stride() == x.stride():
            return hit[1]           # computed ahead by the block's grouped launch (_cbn_prefetch)
        return Fn.DenseFn.apply(x, wk, bias)


def flatten(x):
    """ops.py:177-178.  Keeps column slices as views (no copy)."""
    if _is_dual(x):
        return Dual(flatten(x.p), flatten(x.t))
    if x.dim() == 2:
        return x
    if x.dim() == 4 and x.shape[1] == 1 and x.shape[2] == 1:
        return x[:, 0, 0, :]
    return x.reshape(x.shape[0], -1)


def hw_flatten(x):
    """ops.py:180-181."""
    return x.reshape(x.shape[0], -1, x.shape[-1])


##################################################################################
# Residual-block, Self-Attention-block
##################################################################################
def _add(a, b):
    if _is_meta(a):
        return _meta(a.shape)
    if _is_dual(a):
        return Dual(Fn.AddFn.apply(a.p, b.p), Fn.AddFn.apply(a.t, b.t))
    return Fn.AddFn.apply(a, b)


def _fork(x, n=2):
    """A tensor consumed by n branches: the branch gradients are summed by a HIP kernel."""
    if _is_dual(x):
        return tuple(Dual(a, b) for a, b in zip(_fork(x.p, n), _fork(x.t, n)))
    if _is_meta(x) or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    state = Fn.ForkState()
    outs = Fn.ForkFn.apply(x, n, state)
    sums = getattr(x, "bg_bn_sums", None)
    for t in outs:
        t.bg_fork = state        # (the branch gradients meet in one buffer where the kernels can accumulate: ForkState)
        if sums is not None:
            t.bg_bn_sums = sums  # (fused batch-norm statistics of x travel with its branches)
    return outs


def resblock(x_init, channels, opt, use_bias=True, scope='resblock'):
    """ops.py:187-198."""
    with variable_scope(scope):
        x_main, x_skip = _fork(x_init)
        with variable_scope('res1'):
            x = conv(x_main, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
            x = opt["act"](x)
        with variable_scope('res2'):
            x = conv(x, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
        return _add(x, x_skip)


def upconv(x, channels, opt, use_bias=True, _accumulate_into=None, _stats=False):
    """ops.py:200-218."""
    m = opt["upsampling_method"]
    if m == 'deconv3':
        return deconv(x, channels, kernel=3, stride=2, use_bias=use_bias, opt=opt, _accumulate_into=_accumulate_into,
                      _stats=_stats)
    elif m == 'deconv4':
        return deconv(x, channels, kernel=4, stride=2, use_bias=use_bias, opt=opt, _accumulate_into=_accumulate_into,
                      _stats=_stats)
    elif m == 'deconv6':
        return deconv(x, channels, kernel=6, stride=2, use_bias=use_bias, opt=opt, _accumulate_into=_accumulate_into,
                      _stats=_stats)
    elif m == 'resize_conv':
        x = up_sample(x, 2)
        y = conv(x, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
        return y if _accumulate_into is None else _add(y, _accumulate_into)
    elif m == 'nn':
        y = up_sample(x, 2)
        return y if _accumulate_into is None else _add(y, _accumulate_into)
    elif m in ('subpixel2', 'subpixel3'):
        raise NotImplementedError("upsampling_method %s is outside the default hot path" % m)
    else:
        raise ValueError("Invalid upsampling method specified: " + str(m))


def g_conv(x, channels, opt, use_bias=True, _accumulate_into=None, _stats=False):
    """ops.py:220-230."""
    m = opt["g_conv"]
    if m == 'deconv3':
        return deconv(x, channels, kernel=3, stride=1, use_bias=use_bias, opt=opt, _accumulate_into=_accumulate_into,
                      _stats=_stats)
    elif m == 'deconv4':
        return deconv(x, channels, kernel=4, stride=1, use_bias=use_bias, opt=opt, _accumulate_into=_accumulate_into,
                      _stats=_stats)
    elif m == 'conv3':
        y = conv(x, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
        return y if _accumulate_into is None else _add(y, _accumulate_into)
    elif m == 'conv5':
        y = conv(x, channels, kernel=5, stride=1, pad=2, use_bias=use_bias, opt=opt)
        return y if _accumulate_into is None else _add(y, _accumulate_into)
    else:
        raise ValueError("Invalid generator convolution type specified: " + str(m))


def _cbn_prefetch(z, subscopes, opt):
    """The beta / gamma projections of the conditional batch norms a block is about to evaluate on ``z`` (ops.py:623-624:
    two fully_connected per condition_batch_norm) as ONE grouped launch (functional.GroupedDenseFn) instead of one GEMM
    each - and, backward, one launch instead of a weight-gradient, a bias-gradient and a split-K reduce each.
    ``subscopes``: scopes of the block, relative to the current one, whose 'batch_norm' the block will open.  The results
    wait in the run's dense cache for the fully_connected calls that would have computed them; anything unusual (first,
    shape-only pass; non-default batch-norm types; a z that needs a gradient) simply leaves the cache empty."""
    if z is None or _is_meta(z) or _is_dual(z) or not z.is_cuda or z.requires_grad:
        return
    if os.environ.get("BG_GROUP_CBN", "1") == "0":                    # A/B switch
        return
    type, bscope = _bn_type(opt, 'batch_norm')
    if type not in ('bn', 'batch_norm'):
        return
    store = S.default_store()
    base = store.scope_name
    sn = opt.get("conv", {}).get("sn", True)
    x = flatten(z)
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32:
        return
    args, names = [], []
    for sub in subscopes:
        for which in ('beta', 'gamma'):
            name = "%s/%s/%s/%s" % (base, sub, bscope, which)
            w, b = store.vars.get(name + "/kernel"), store.vars.get(name + "/bias")
            if w is None or w.shape[0] != x.shape[1]:
                return
            wk = w
            if sn:
                cached = _run.sn_cache.get(w.bg_name)
                if cached is None or cached[1] != (torch.is_grad_enabled() and w.requires_grad):
                    return
                wk = cached[0]
            args += [x, wk, b]
            names.append(w.bg_name)
    if not names or len(names) > hip.DENSE_GROUP_MAX:
        return
    ys = Fn.GroupedDenseFn.apply(len(names), *args)
    for nm, y in zip(names, ys):
        _run.dense_cache[nm] = (x, y)


def resblock_up(x_init, channels, opt, use_bias=True, scope='resblock_up'):
    """ops.py:232-248."""
    with variable_scope(scope):
        x_main, x_skip = _fork(x_init)
        with variable_scope('res1'):
            x = _bn_act(x_main, None, opt)
            x = upconv(x, channels, use_bias=use_bias, opt=opt)
        with variable_scope('res2'):
            x = _bn_act(x, None, opt)
        with variable_scope('skip'):
            skip = upconv(x_skip, channels, use_bias=use_bias, opt=opt)
        with variable_scope('res2'):
            x = g_conv(x, channels, use_bias=use_bias, opt=opt, _accumulate_into=skip)
    return x


def resblock_up_condition(x_init, z, channels, opt, use_bias=True, scope='resblock_up'):
    """ops.py:250-266.  The skip branch is evaluated before the last main-branch deconv so the
    residual sum is fused into that kernel's epilogue (same values, one pass less)."""
    with variable_scope(scope):
        _cbn_prefetch(z, ('res1', 'res2'), opt)
        x_main, x_skip = _fork(x_init)
        with variable_scope('res1'):
            x = _bn_act(x_main, z, opt)
            x = upconv(x, channels, use_bias=use_bias, opt=opt, _stats=True)       # (feeds res2's batch norm)
        with variable_scope('res2'):
            x = _bn_act(x, z, opt)
        with variable_scope('skip'):
            skip = upconv(x_skip, channels, use_bias=use_bias, opt=opt)
        with variable_scope('res2'):
            # (the block's output feeds the next block's first batch norm, or the generator's last one)
            x = g_conv(x, channels, use_bias=use_bias, opt=opt, _accumulate_into=skip, _stats=True)
    return x


def downconv(x, channels, opt, use_bias=True, method=None):
    """ops.py:269-291."""
    if method is None:
        method = opt["downsampling_method"]
    if method == 'strided_conv3':
        return conv(x, channels, kernel=3, stride=2, pad=1, use_bias=use_bias, opt=opt)
    elif method == 'resize_conv1':
        x = conv(x, channels, kernel=1, stride=1, pad=0, use_bias=use_bias, opt=opt)
        return avg_pooling(x)
    elif method == 'resize_conv3':
        x = conv(x, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
        return avg_pooling(x)
    elif method == 'pool_only':
        return avg_pooling(x)
    elif method == 'max_pool_only':
        return max_pooling(x)
    elif method == 'resize_conv35':
        raise NotImplementedError("downsampling_method %s (mixed kernels) is outside the default hot path" % method)
    else:
        raise ValueError("Invalid downsampling method specified: " + str(method))


def resblock_down(x_init, channels, opt, use_bias=True, scope='resblock_down'):
    """ops.py:293-313."""
    with variable_scope(scope):
        x_main, x_init = _fork(x_init)
        with variable_scope('res1'):
            if opt["bn_in_d"]:
                x = bn(x_main, opt=opt)
            else:
                x = x_main
            x = opt["act"](x)
            res_method = opt["downsampling_method"]
            if res_method != 'strided_conv3':
                res_method = 'resize_conv3'
            x = downconv(x, channels, use_bias=use_bias, opt=opt, method=res_method)
        with variable_scope('res2'):
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
            x = opt["act"](x)
        # (the skip branch is evaluated before the last main-branch conv so that the residual sum is fused into that
        #  kernel's epilogue: same values, one pass over the block's output less - as in resblock_up_condition)
        with variable_scope('skip'):
            x_init = downconv(x_init, channels, use_bias=use_bias, opt=opt, method=opt["downsampling_method"])
        with variable_scope('res2'):
            x = conv(x, channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt,
                     _accumulate_into=None if _is_meta(x) else x_init)
    return x


def _channel_slice(x, lo, hi):
    """tf.split along channels: a contiguous copy of x[..., lo:hi] (data movement only)."""
    if _is_meta(x):
        return _meta(tuple(x.shape[:-1]) + (hi - lo,))
    return x[..., lo:hi]


def _channel_concat(a, b):
    if _is_meta(a):
        return _meta(tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],))
    return torch.cat([a, b], dim=-1)


def resblock_up_cond_deep(x_init, z, channels_out, opt, upscale=True, use_bias=True, scope='deep_resblock'):
    """ops.py:317-358 (--deep): bottleneck 1x1 -> (upconv) -> two g_convs at (Cin + Cout) // 6 channels -> 1x1."""
    channels_in = int(x_init.shape[-1])
    inner_channels = round_up((channels_in + channels_out) // 6, 8)
    with variable_scope(scope):
        x_main, x_skip = _fork(x_init)
        with variable_scope('bottleneck'):
            x = _bn_act(x_main, z, opt)
            x = conv(x, inner_channels, kernel=1, stride=1, use_bias=False, opt=opt)
        with variable_scope('upscale'):
            x = _bn_act(x, z, opt)
            if upscale:
                x = upconv(x, inner_channels, use_bias=False, opt=opt)
        with variable_scope('inner1'):
            x = g_conv(x, inner_channels, use_bias=False, opt=opt)
            x = _bn_act(x, z, opt)
        with variable_scope('inner2'):
            x = g_conv(x, inner_channels, use_bias=False, opt=opt)
            x = _bn_act(x, None, opt)
        with variable_scope('proj'):
            x = conv(x, channels_out, kernel=1, stride=1, use_bias=use_bias, opt=opt)
        with variable_scope('skip'):
            if upscale:
                kept = _channel_slice(x_skip, 0, channels_out) if channels_in != channels_out else x_skip
                x_skip = upconv(kept, channels_out, use_bias=use_bias, opt=opt)
    return _add(x, x_skip)


def resblock_down_deep(x_init, channels_out, opt, downscale=True, use_bias=True, scope='deep_resblock'):
    """ops.py:360-401 (--deep)."""
    channels_in = int(x_init.shape[-1])
    inner_channels = round_up((channels_in + channels_out) // 6, 8)
    with variable_scope(scope):
        x_main, x_skip = _fork(x_init)
        with variable_scope('bottleneck'):
            x = x_main
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
            x = opt["act"](x)
            x = conv(x, inner_channels, kernel=1, stride=1, pad=0, use_bias=use_bias, opt=opt)
        with variable_scope('inner1'):
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
            x = opt["act"](x)
            x = conv(x, inner_channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
        with variable_scope('inner2'):
            if opt["bn_in_d"]:
                x = bn(x, opt=opt)
            x = opt["act"](x)
            x = conv(x, inner_channels, kernel=3, stride=1, pad=1, use_bias=use_bias, opt=opt)
        with variable_scope('downscale'):
            x = opt["act"](x)
            if downscale:
                x = downconv(x, inner_channels, use_bias=use_bias, opt=opt, method='pool_only')
        with variable_scope('proj'):
            x = conv(x, channels_out, kernel=1, stride=1, pad=0, use_bias=use_bias, opt=opt)
        with variable_scope('skip'):
            if downscale:
                x_skip = downconv(x_skip, channels_in, use_bias=use_bias, opt=opt, method='pool_only')
            if channels_in != channels_out:
                conv_ch = channels_out - channels_in
                dense = conv(x_skip, conv_ch, kernel=1, stride=1, pad=0, use_bias=use_bias, opt=opt)
                x_skip = _channel_concat(x_skip, dense)
    return _add(x, x_skip)


def clown_conv(x, channels, opt, use_bias=True, scope='clown', z=None):
    raise NotImplementedError("clown_conv (ops.py:403, --g_mixed_resblocks) is outside the default hot path")


def mixed_resblock(x, inner_channels, out_channels, opt, use_bias=False, z=None, scope='res_mixed'):
    raise NotImplementedError("mixed_resblock (ops.py:433, --g_mixed_resblocks) is outside the default hot path")


def self_attention(x, channels, opt, scope='self_attention'):
    raise NotImplementedError("self_attention (ops.py:445) is never called by the reference; use self_attention_2")


def _conv1x1_vars(name, cin, cout, opt, use_bias):
    """The variables conv(x, cout, kernel=1, stride=1, scope=name) creates, in its order, without running it."""
    with variable_scope(name) as full_scope:
        reg = opt.get("conv", {}).get("regularizer", None) if 'generator' in full_scope else None
        w = get_variable("kernel", shape=[1, 1, cin, cout], initializer=weight_init, regularizer=reg)
        _regularize(w, reg)
        wk = spectral_norm(w) if opt.get("conv", {}).get("sn", True) else w
        bias = get_variable("bias", [cout], initializer=S.constant_initializer(0.0)) if use_bias else None
    return wk, bias


def self_attention_2(x, channels, opt, scope='self_attention'):
    """ops.py:467-492."""
    with variable_scope(scope):
        use_bias = opt.get("self_attention_bias", False)
        if (Fn.Precision.resident and not _is_meta(x) and not _is_dual(x) and x.dtype == torch.bfloat16
                and channels % 8 == 0 and (channels // 2) % 8 == 0):
            # bf16-resident: f | g | h as one GEMM, one pool, fused bf16 attention on column slices
            C = x.shape[-1]
            wf, b_f = _conv1x1_vars('f_conv', C, channels // 8, opt, use_bias)
            wg, b_g = _conv1x1_vars('g_conv', C, channels // 8, opt, use_bias)
            wh, b_h = _conv1x1_vars('h_conv', C, channels // 2, opt, use_bias)
            if Fn.SaFrontFn.supported(x, wf, wg, wh):
                x_qkv, x_res = _fork(x, 2)
                gamma = get_variable("gamma", [1], initializer=S.constant_initializer(0.0))
                o = Fn.SaFrontFn.apply(x_qkv, wf, wg, wh, b_f, b_g, b_h)
                o = conv(o, channels, kernel=1, stride=1, opt=opt, scope='attn_conv', use_bias=use_bias)
                return Fn.ScaleAddFn.apply(o, gamma, x_res)
            # (narrow test models: d = channels // 8 is not a multiple of 4 -> the generic path below; the variables
            #  above are the ones it uses, created once per scope)
        x_f, x_g, x_h, x = _fork(x, 4)
        # (bf16-resident mode: the attention core stays fp32 - its 1x1 producers write fp32 directly)
        f32o = None if _is_meta(x) or _is_dual(x) or x.dtype == torch.float32 else torch.float32
        f = conv(x_f, channels // 8, kernel=1, stride=1, opt=opt, scope='f_conv', use_bias=use_bias, _out_dtype=f32o)
        f = max_pooling(f)
        g = conv(x_g, channels // 8, kernel=1, stride=1, opt=opt, scope='g_conv', use_bias=use_bias, _out_dtype=f32o)
        h = conv(x_h, channels // 2, kernel=1, stride=1, opt=opt, scope='h_conv', use_bias=use_bias, _out_dtype=f32o)
        h = max_pooling(h)
        gamma = get_variable("gamma", [1], initializer=S.constant_initializer(0.0))
        if _is_meta(x):
            o = _meta((x.shape[0], x.shape[1], x.shape[2], channels // 2))
        elif _is_dual(x):
            o = _attention_dual(hw_flatten(g), hw_flatten(f), hw_flatten(h))
            o = o.reshape(x.shape[0], x.shape[1], x.shape[2], channels // 2)
        else:
            o = Fn.AttentionFn.apply(hw_flatten(_to(g, torch.float32)), hw_flatten(_to(f, torch.float32)),
                                     hw_flatten(_to(h, torch.float32)))                # softmax(g f^T) h
            o = o.reshape(x.shape[0], x.shape[1], x.shape[2], channels // 2)
            if x.dtype == torch.bfloat16 and (channels // 2) % 8 == 0 and channels % 8 == 0:
                o = _to(o, torch.bfloat16)
        o = conv(o, channels, kernel=1, stride=1, opt=opt, scope='attn_conv', use_bias=use_bias)
        if _is_meta(x):
            return _meta(x.shape)
        if _is_dual(x):
            return Dual(Fn.ScaleAddFn.apply(o.p, gamma, x.p), Fn.ScaleAddFn.apply(o.t, gamma, x.t))
        return Fn.ScaleAddFn.apply(o, gamma, x)                                          # gamma * o + x


def _attention_dual(q, k, v):
    """softmax(q k^T) v and its tangent with the probabilities materialised (BmmFn / SoftmaxFn / SoftmaxTangentFn):
    sdot = qdot k^T + q kdot^T ; pdot = p * (sdot - <p, sdot>) ; odot = pdot v + p vdot."""
    q1, q2 = _fork(q.p)
    k1, k2 = _fork(k.p)
    v1, v2 = _fork(v.p)
    p = Fn.SoftmaxFn.apply(Fn.BmmFn.apply(q1, k1, True))
    p1, p2, p3 = _fork(p, 3)
    o = Fn.BmmFn.apply(p1, v1, False)
    sdot = Fn.AddFn.apply(Fn.BmmFn.apply(q.t, k2, True), Fn.BmmFn.apply(q2, k.t, True))
    pdot = Fn.SoftmaxTangentFn.apply(p2, sdot)
    odot = Fn.AddFn.apply(Fn.BmmFn.apply(pdot, v2, False), Fn.BmmFn.apply(p3, v.t, False))
    return Dual(o, odot)


##################################################################################
# Sampling
##################################################################################
def global_avg_pooling(x):
    """ops.py:498-501: tf.reduce_mean(x, axis=[1, 2]) = the sum pool scaled by 1 / (H W)."""
    if _is_meta(x):
        return _meta((x.shape[0], x.shape[-1]))
    inv = 1.0 / float(x.shape[1] * x.shape[2])
    if _is_dual(x):
        return Dual(Fn.ScaleFn.apply(Fn.SumPoolFn.apply(x.p), inv), Fn.ScaleFn.apply(Fn.SumPoolFn.apply(x.t), inv))
    return Fn.ScaleFn.apply(Fn.SumPoolFn.apply(x), inv)


def global_sum_pooling(x):
    """ops.py:503-506."""
    if _is_meta(x):
        return _meta((x.shape[0], x.shape[-1]))
    if _is_dual(x):
        return Dual(Fn.SumPoolFn.apply(x.p), Fn.SumPoolFn.apply(x.t))
    return Fn.SumPoolFn.apply(x)


def max_pooling(x):
    """ops.py:508-510 (2x2, stride 2, 'SAME'; even H and W)."""
    if x.shape[1] % 2 or x.shape[2] % 2:
        raise NotImplementedError("max_pooling on odd spatial sizes")
    if _is_meta(x):
        return _meta((x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]))
    if _is_dual(x):
        p_pool, p_sel = _fork(x.p)
        return Dual(Fn.MaxPool2Fn.apply(p_pool), Fn.MaxPool2TangentFn.apply(x.t, p_sel))
    return Fn.MaxPool2Fn.apply(x)


def avg_pooling(x):
    """ops.py:512-514 (2x2, stride 2, 'SAME'; even H and W)."""
    if x.shape[1] % 2 or x.shape[2] % 2:
        raise NotImplementedError("avg_pooling on odd spatial sizes")
    if _is_meta(x):
        return _meta((x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]))
    if _is_dual(x):
        return Dual(Fn.AvgPool2Fn.apply(x.p), Fn.AvgPool2Fn.apply(x.t))
    return _resident_out(Fn.AvgPool2Fn.apply(_to(x, torch.float32)))


def up_sample(x, scale_factor=2):
    """ops.py:516-519: tf.image.resize_nearest_neighbor to scale_factor times the size."""
    if scale_factor != 2:
        raise NotImplementedError("up_sample scale_factor != 2")
    if _is_meta(x):
        return _meta((x.shape[0], x.shape[1] * 2, x.shape[2] * 2, x.shape[3]))
    return _resident_out(Fn.UpSample2Fn.apply(_to(x, torch.float32)))


##################################################################################
# Activation function
##################################################################################
_const_alpha = {}


def _constant_alpha(C, value, device):
    key = (C, float(value), str(device))
    if key not in _const_alpha:
        _const_alpha[key] = torch.full((C,), float(value), dtype=torch.float32, device=device)
    return _const_alpha[key]


def _prelu_apply(x, alphas):
    if _is_dual(x):
        p_act, p_mask = _fork(x.p)          # the primal feeds its own activation and the tangent's slope mask
        return Dual(Fn.PReluFn.apply(p_act, alphas), Fn.PReluTangentFn.apply(x.t, p_mask, alphas))
    return Fn.PReluFn.apply(x, alphas)


def lrelu(x, alpha=0.2):
    """ops.py:525-526."""
    if _is_meta(x):
        return _meta(x.shape)
    return _prelu_apply(x, _constant_alpha(x.shape[-1], alpha, x.device))


def relu(x):
    """ops.py:529-530."""
    if _is_meta(x):
        return _meta(x.shape)
    return _prelu_apply(x, _constant_alpha(x.shape[-1], 0.0, x.device))


def prelu(x, scope=None, init_val=0.0):
    """ops.py:532-537."""
    with variable_scope(name_or_scope=scope, default_name="prelu"):
        alphas = get_variable('alpha', x.shape[-1], initializer=S.constant_initializer(init_val))
        if _is_meta(x):
            return _meta(x.shape)
        return _prelu_apply(x, alphas)


def tanh(x):
    """ops.py:539-540."""
    if _is_meta(x):
        return _meta(x.shape)
    return Fn.TanhFn.apply(x)


##################################################################################
# Normalization function
##################################################################################
def _bn_type(opt, scope):
    type = opt.get("bn", {}).get("type", "bn")
    if type == 'batch_norm_broken_renorm':
        type = 'batch_norm'
        if scope == 'batch_norm':
            scope = 'batch_renorm'
    return type, scope


def bn(x, opt={}, scope='batch_norm'):
    """ops.py:546-561."""
    type, scope = _bn_type(opt, scope)
    if _is_dual(x):
        # forward-mode pass of the gradient penalty through a discriminator with --bn_in_d: the primal goes through the
        # ordinary kernels, WITHOUT moving the population statistics a second time (the penalty's discriminator is one
        # instantiation in the reference, and pass (1) of model.gradient_penalty already made its update); the tangent
        # through functional.BnTangentFn with the statistics the primal call just computed
        if type not in ('bn', 'batch_norm') or not opt["is_training"]:
            raise NotImplementedError("gradient penalty with --bn_in_d: only training-mode batch_norm has a tangent pass")
        frozen = dict(opt)
        frozen["bn"] = dict(opt.get("bn", {}), momentum=1.0)
        yp = batch_norm(x.p, opt=frozen, scope=scope)
        mean, rstd, count = Fn.BnActFn.last_stats
        with variable_scope(scope):
            gamma = get_variable("gamma", [x.shape[-1]], initializer=S.constant_initializer(1.0))
        return Dual(yp, Fn.BnTangentFn.apply(x.t, x.p, gamma, mean, rstd, count, _run.reduce_fn))
    if type == 'bn' or type == 'batch_norm':
        return batch_norm(x, opt=opt, scope=scope)
    elif type == 'batch_renorm':
        if scope == 'batch_norm':
            scope = 'batch_renorm'
        return batch_renorm(x, opt=opt, scope=scope)
    else:
        raise ValueError("Unknown BN type: " + str(type))


def cond_bn(x, z, opt={}, scope='batch_norm'):
    """ops.py:563-578."""
    type, scope = _bn_type(opt, scope)
    if type == 'bn' or type == 'batch_norm':
        return condition_batch_norm(x, z, opt=opt, scope=scope)
    elif type == 'batch_renorm':
        if scope == 'batch_norm':
            scope = 'batch_renorm'
        return condition_batch_renorm(x, z, opt=opt, scope=scope)
    else:
        raise ValueError("Unknown BN type: " + str(type))


def _act_alpha(act, x):
    """Fusable activation -> (fused?, alpha tensor or None).  PReLU's variable is created under the
    reference's default-name scope so the parameter name is unchanged."""
    if act is prelu:
        with variable_scope(None, default_name="prelu"):
            return True, get_variable('alpha', x.shape[-1], initializer=S.constant_initializer(0.0))
    if act is relu:
        return True, (None if _is_meta(x) else _constant_alpha(x.shape[-1], 0.0, x.device))
    return False, None


def _bn_out_dtype(x, out_fp32):
    """bf16-resident mode: batch-norm outputs feed convolutions, so they are written as bf16 (fp32 on request: the
    3-channel RGB head reads fp32)."""
    if _is_meta(x) or not Fn.Precision.resident:
        return None
    return torch.float32 if out_fp32 else torch.bfloat16


def _bn_act(x, z, opt, _out_fp32=False):
    """(cond_)bn followed by opt['act'], fused into one apply kernel when the activation is PReLU/ReLU."""
    fused_types = ('bn', 'batch_norm', 'batch_norm_broken_renorm', 'batch_renorm')
    if opt.get("bn", {}).get("type", "bn") in fused_types and opt["act"] in (prelu, relu):
        type, scope = _bn_type(opt, 'batch_norm')
        if type == 'batch_renorm':
            if z is None:
                return batch_renorm(x, opt=opt, scope='batch_renorm', _act=opt["act"])
            return condition_batch_renorm(x, z, opt=opt, scope='batch_renorm', _act=opt["act"])
        if z is None:
            return batch_norm(x, opt=opt, scope=scope, _act=opt["act"], _out_fp32=_out_fp32)
        return condition_batch_norm(x, z, opt=opt, scope=scope, _act=opt["act"])
    x = bn(x, opt=opt) if z is None else cond_bn(x, z, opt=opt)
    return opt["act"](x)


def batch_norm(x, opt={}, scope='batch_norm', _act=None, _out_fp32=False):
    """ops.py:580-585: tf.layers.batch_normalization(momentum, epsilon=1e-5, training)."""
    C = x.shape[-1]
    with variable_scope(scope):
        gamma = get_variable("gamma", [C], initializer=S.constant_initializer(1.0))
        beta = get_variable("beta", [C], initializer=S.constant_initializer(0.0))
        mm = get_variable("moving_mean", [C], initializer=S.constant_initializer(0.0), trainable=False)
        mv = get_variable("moving_variance", [C], initializer=S.constant_initializer(1.0), trainable=False)
    alpha = None
    if _act is not None:
        _, alpha = _act_alpha(_act, x)
    if _is_meta(x):
        return _meta(x.shape)
    momentum = opt.get("bn", {}).get("momentum", 0.98)
    return Fn.BnActFn.apply(x, gamma, beta, alpha, mm, mv, momentum, 1e-05, True, bool(opt["is_training"]),
                            _run.reduce_fn, _run.world, None, _bn_out_dtype(x, _out_fp32))


def normalize_renorm_clipping_params(renorm_clipping):
    """ops.py:587-597."""
    if "rmax" not in renorm_clipping:
        renorm_clipping["rmax"] = 1.5
    if "dmax" not in renorm_clipping:
        renorm_clipping["dmax"] = 0.5
    if "rmax" in renorm_clipping and "rmin" not in renorm_clipping:
        renorm_clipping["rmin"] = 1.0 / renorm_clipping["rmax"]
    return renorm_clipping


def batch_renorm(x, opt={}, scope='batch_renorm', _act=None):
    """ops.py:600-609: tf.layers.batch_normalization(renorm=True, renorm_momentum, renorm_clipping), with the
    TF 1.15 layer's variables and update rule: ``renorm_mean`` / ``renorm_stddev`` (no zero-debias weights),
    corrections against max(renorm_stddev, sqrt(eps)) read before the update, the non-fused moving-variance
    update (biased batch variance), r = 1 / d = 0 at inference."""
    clip = normalize_renorm_clipping_params(dict(opt.get("bn", {}).get("renorm_clipping", {})))
    C = x.shape[-1]
    with variable_scope(scope):
        gamma = get_variable("gamma", [C], initializer=S.constant_initializer(1.0))
        beta = get_variable("beta", [C], initializer=S.constant_initializer(0.0))
        mm = get_variable("moving_mean", [C], initializer=S.constant_initializer(0.0), trainable=False)
        mv = get_variable("moving_variance", [C], initializer=S.constant_initializer(1.0), trainable=False)
        rmean = get_variable("renorm_mean", [C], initializer=S.constant_initializer(0.0), trainable=False)
        rstd = get_variable("renorm_stddev", [C], initializer=S.constant_initializer(1.0), trainable=False)
    alpha = None
    if _act is not None:
        _, alpha = _act_alpha(_act, x)
    if _is_meta(x):
        return _meta(x.shape)
    renorm = dict(ref_mean=rmean, ref_scale=rstd, scale_is_var=0, weight=None, update=1,
                  rmin=clip["rmin"], rmax=clip["rmax"], dmax=clip["dmax"],
                  decay=opt.get("bn", {}).get("renorm_momentum", 0.9))
    return Fn.BnActFn.apply(x, gamma, beta, alpha, mm, mv, opt.get("bn", {}).get("momentum", 0.98), 1e-05, False,
                            bool(opt["is_training"]), _run.reduce_fn, _run.world, renorm, _bn_out_dtype(x, False))


def condition_batch_norm(x, z, opt={}, scope='batch_norm', _act=None):
    """ops.py:611-643."""
    with variable_scope(scope):
        c = x.shape[-1]
        decay = opt.get("bn", {}).get("momentum", 0.98)
        epsilon = 1e-05
        test_mean = get_variable("pop_mean", shape=[c], initializer=S.constant_initializer(0.0), trainable=False)
        test_var = get_variable("pop_var", shape=[c], initializer=S.constant_initializer(1.0), trainable=False)
        beta = fully_connected(z, units=c, scope='beta', opt=opt)
        gamma = fully_connected(z, units=c, scope='gamma', opt=opt)
    alpha = None
    if _act is not None:
        _, alpha = _act_alpha(_act, x)
    if _is_meta(x):
        return _meta(x.shape)
    return Fn.BnActFn.apply(x, gamma, beta, alpha, test_mean, test_var, decay, epsilon, False,
                            bool(opt["is_training"]), _run.reduce_fn, _run.world, None, _bn_out_dtype(x, False))


def condition_batch_renorm(x, z, opt={}, scope='batch_renorm', _act=None):
    """ops.py:645-715.  Not shared (default): separate renorm_mean / renorm_var / renorm_weight running statistics
    with decay ``renorm_momentum``, faded in by renorm_weight; the population statistics then ALSO use
    ``renorm_momentum`` as their decay (ops.py:658-659).  Shared: the corrections are measured against
    pop_mean / pop_var themselves (weight 1), which keep the ``momentum`` decay."""
    with variable_scope(scope):
        c = x.shape[-1]
        bn_opt = opt.get("bn", {})
        clip = normalize_renorm_clipping_params(dict(bn_opt.get("renorm_clipping", {})))
        test_decay = bn_opt.get("momentum", 0.98)
        renorm_decay = bn_opt.get("renorm_momentum", 0.9)
        shared = bn_opt.get("shared_renorm", False)
        renorm_fadein_decay = bn_opt.get("renorm_fadein_decay", 0.9999)
        if not shared:
            test_decay = renorm_decay
        epsilon = 1e-05
        test_mean = get_variable("pop_mean", shape=[c], initializer=S.constant_initializer(0.0), trainable=False)
        test_var = get_variable("pop_var", shape=[c], initializer=S.constant_initializer(1.0), trainable=False)
        if not shared:
            renorm_mean = get_variable("renorm_mean", shape=[c], initializer=S.constant_initializer(0.0), trainable=False)
            renorm_var = get_variable("renorm_var", shape=[c], initializer=S.constant_initializer(1.0), trainable=False)
            renorm_weight = get_variable("renorm_weight", shape=[], initializer=S.constant_initializer(0.0), trainable=False)
        else:
            renorm_mean, renorm_var, renorm_weight = test_mean, test_var, None
        beta = fully_connected(z, units=c, scope='beta', opt=opt)
        gamma = fully_connected(z, units=c, scope='gamma', opt=opt)
    alpha = None
    if _act is not None:
        _, alpha = _act_alpha(_act, x)
    if _is_meta(x):
        return _meta(x.shape)
    renorm = dict(ref_mean=renorm_mean, ref_scale=renorm_var, scale_is_var=1, weight=renorm_weight,
                  update=int(not shared), rmin=clip["rmin"], rmax=clip["rmax"], dmax=clip["dmax"],
                  decay=renorm_decay, fadein_decay=renorm_fadein_decay)
    return Fn.BnActFn.apply(x, gamma, beta, alpha, test_mean, test_var, test_decay, epsilon, False,
                            bool(opt["is_training"]), _run.reduce_fn, _run.world, renorm, _bn_out_dtype(x, False))


def spectral_norm(w, iteration=1, _shape_only=False):
    """ops.py:718-747.  The ``u`` variable lives next to ``w`` in the current scope; one power
    iteration per weight per run (a second instantiation in the same run reuses the first)."""
    if iteration != 1:
        raise NotImplementedError("spectral_norm iteration != 1")
    u = get_variable("u", [1, w.shape[-1]], initializer=S.random_normal_initializer(), trainable=False)
    key = getattr(w, "bg_name", None)
    if key is not None:
        S.default_store().register_sn(key, u.bg_name)
    if _shape_only:
        return w          # shape-propagation (manifest) mode: only the variable is registered
    if key is not None and key in _run.sn_cache:
        cached, needs = _run.sn_cache[key]
        if needs == (torch.is_grad_enabled() and w.requires_grad):
            return cached
    wn = Fn.SpectralNormFn.apply(w, u)
    if key is not None:
        _run.sn_cache[key] = (wn, torch.is_grad_enabled() and w.requires_grad)
    return wn


##################################################################################
# Loss function
##################################################################################
def discriminator_loss(loss_func, real, fake, flood_level=0):
    """ops.py:753-797 (the gradient penalty of the wgan / dragan types is added by the caller, BigGAN.py:880)."""
    if loss_func == 'hinge':
        return Fn.HingeDLossFn.apply(real, fake, flood_level, _run.reduce_fn, _run.world)
    if loss_func in Fn.GAN_LOSS_KINDS:
        return Fn.GanLossFn.apply(real, fake, Fn.GAN_LOSS_KINDS[loss_func], 0, flood_level, _run.reduce_fn, _run.world)
    raise ValueError("discriminator_loss: unknown loss '%s'" % loss_func)


def generator_loss(loss_func, fake, real, flood_level=0):
    """ops.py:799-840 ('hinge' branch)."""
    if loss_func == 'hinge':
        return Fn.HingeGLossFn.apply(fake, flood_level, _run.reduce_fn, _run.world)
    if loss_func in Fn.GAN_LOSS_KINDS:
        if loss_func.startswith('ra-') and real is None:
            raise ValueError("generator_loss('%s') is relativistic: it needs the real logits" % loss_func)
        r = real if loss_func.startswith('ra-') else None
        return Fn.GanLossFn.apply(r, fake, Fn.GAN_LOSS_KINDS[loss_func], 1, flood_level, _run.reduce_fn, _run.world)
    raise ValueError("generator_loss: unknown loss '%s'" % loss_func)


def glu(x, opt=None):
    raise NotImplementedError("glu (ops.py:842) is outside the default hot path")


def flood_loss(loss, flood_level):
    """ops.py:847-848 (applied inside the hinge kernels; provided for API parity on host scalars)."""
    return abs(loss - flood_level) + flood_level
