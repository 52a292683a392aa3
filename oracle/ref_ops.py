"""Oracle restatement of the reference operator library (``/root/reference/ops.py``) and
``DiffAugment_tf.py`` on torch-CPU tensors (float32 or float64), NHWC / HWIO layouts.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PARITY UNPINNED (no TensorFlow here).

Variables live in a :class:`VarStore` keyed by the TensorFlow variable names the
reference's ``tf.variable_scope`` / ``tf.get_variable`` calls would produce.  Gradients
come from torch autograd applied to this forward restatement.
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# optional bf16 rounding points (tests of the product's bf16-resident mode only; OFF = the reference's arithmetic)
# ----------------------------------------------------------------------------------
class _Rounding:
    """The reference computes in fp32 (ops.py:14); the product's ``--precision bf16`` keeps the trunk's activations,
    their gradients and packed conv weights in bf16.  A float64 oracle therefore differs from that
    product by bf16 noise (10 - 30 % on first-layer gradients), which hides real errors.  With ``on`` the oracle rounds
    to bf16 (round-to-nearest-even, torch's conversion = the product's) at the points where the product stores bf16:
    the output of every conv / transposed conv (after bias and fused residual sum), batch-norm + activation, stand-alone
    activation, attention and gated residual on NHWC tensors whose channel count is a multiple of 8 (the others stay
    fp32 in the product too), the spectrally normalised conv kernels that have a packed copy, the attention
    probabilities - and, backward, the gradient arriving at each of those activation points.  Everything between the
    rounding points stays float64, so what remains against the product is accumulation order."""
    on = False


ROUND = _Rounding()


class _RoundAct(torch.autograd.Function):
    """A tensor the product stores in bf16 whose gradient it stores in bf16 as well."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """A bf16 copy of an fp32 quantity (packed weights, attention probabilities): the gradient passes unrounded."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _resident(x):
    """Tensors the product keeps in bf16: 4-D NHWC with a channel count that is a multiple of 8."""
    return ROUND.on and x.dim() == 4 and x.shape[-1] % 8 == 0


def r_act(x):
    return _RoundAct.apply(x) if _resident(x) else x


# ----------------------------------------------------------------------------------
# variable store (tf.get_variable by full scope name)
# ----------------------------------------------------------------------------------
class VarStore:
    """name -> leaf tensor.  ``create`` mirrors ``tf.get_variable`` (create on first use,
    reuse afterwards).  Initialisers follow ops.py:13 (truncated normal sigma=0.02),
    ops.py:722 (``u`` ~ N(0,1)), constant zeros/ones elsewhere."""

    def __init__(self, dtype=torch.float64, seed=0):
        self.dtype = dtype
        self.rng = np.random.default_rng(seed)
        self.vars = OrderedDict()
        self.trainable = OrderedDict()
        self.reg_losses = []          # tf.losses.get_regularization_losses()
        self.state_updates = OrderedDict()  # name -> new value (assign ops of this run)
        self.frozen = False           # True: creating a new variable is an error
        self.uv_cache = {}            # scope -> (u_hat, v_hat) of the last run
        self.freeze_uv = False        # True: reuse uv_cache (finite-difference checks: u_hat, v_hat
                                      # are stop-gradient constants, ops.py:738-739)

    def _init(self, shape, init):
        if isinstance(init, (int, float)):
            return np.full(shape, float(init))
        if init == "trunc_normal":     # tf.truncated_normal_initializer(0, 0.02): redraw beyond 2 sigma
            a = self.rng.standard_normal(shape)
            bad = np.abs(a) > 2.0
            while bad.any():
                a[bad] = self.rng.standard_normal(int(bad.sum()))
                bad = np.abs(a) > 2.0
            return a * 0.02
        if init == "normal":
            return self.rng.standard_normal(shape)
        raise ValueError(init)

    def get(self, name, shape, init, trainable=True):
        shape = tuple(int(s) for s in shape)
        if name not in self.vars:
            if self.frozen:
                raise KeyError("variable %s does not exist" % name)
            t = torch.tensor(self._init(shape, init), dtype=self.dtype)
            t.requires_grad_(trainable)
            self.vars[name] = t
            self.trainable[name] = trainable
        v = self.vars[name]
        assert tuple(v.shape) == shape, (name, tuple(v.shape), shape)
        return v

    def assign(self, name, value):
        """tf.assign executed as a control dependency: recorded, applied by commit()."""
        self.state_updates[name] = value.detach().clone()

    def current(self, name):
        """Value a second update op of the same run starts from (tf.layers' moving averages are
        assign_sub ops: two instantiations of the layer in one run compound)."""
        return self.state_updates.get(name, self.vars[name].detach())

    def commit(self):
        for k, v in self.state_updates.items():
            with torch.no_grad():
                self.vars[k].copy_(v)
        self.state_updates.clear()

    def load(self, arrays):
        for k, a in arrays.items():
            t = torch.tensor(np.asarray(a), dtype=self.dtype)
            tr = self.trainable.get(k, not _is_state_name(k))
            t.requires_grad_(tr)
            self.vars[k] = t
            self.trainable[k] = tr

    def export(self):
        return OrderedDict((k, v.detach().numpy().copy()) for k, v in self.vars.items())


def _is_state_name(name):
    leaf = name.rsplit("/", 1)[-1]
    return leaf in ("u", "pop_mean", "pop_var", "moving_mean", "moving_variance",
                    "renorm_mean", "renorm_var", "renorm_weight", "renorm_stddev")


def round_up(val, multiple):                       # utils.py:335-336
    return (int(val) + multiple - 1) // multiple * multiple


# ----------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------
def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def l2_normalize(t, axis=None, eps=1e-12):
    """tf.nn.l2_normalize: t * rsqrt(max(sum(t^2, axis), eps))."""
    if axis is None:
        ss = (t * t).sum()
    else:
        ss = (t * t).sum(dim=axis, keepdim=True)
    return t * torch.rsqrt(torch.clamp(ss, min=eps))


# ----------------------------------------------------------------------------------
# spectral norm  (ops.py:718-747)
# ----------------------------------------------------------------------------------
def spectral_norm(vs, scope, w):
    """One power iteration from the stored ``u``; returns w / sigma.  u_hat, v_hat are
    stop-gradient (ops.py:738-739); sigma = v_hat W u_hat^T keeps its explicit dependence
    on W (ops.py:741).  ``u <- u_hat`` is recorded as a state update (ops.py:743).
    One iteration per weight per run: a second call in the same run reuses the first
    call's u_hat/v_hat (race note)."""
    w_shape = w.shape
    W = w.reshape(-1, w_shape[-1])
    u = vs.get(scope + "/u", (1, w_shape[-1]), "normal", trainable=False)
    if vs.freeze_uv and scope in vs.uv_cache:
        u_hat, v_hat = vs.uv_cache[scope]
    else:
        with torch.no_grad():
            v_hat = l2_normalize(u @ W.t())          # ops.py:732-733
            u_hat = l2_normalize(v_hat @ W)          # ops.py:735-736
        vs.uv_cache[scope] = (u_hat, v_hat)
    sigma = (v_hat @ W) @ u_hat.t()              # ops.py:741
    vs.assign(scope + "/u", u_hat)               # ops.py:743
    return (W / sigma).reshape(w_shape)          # ops.py:744-745


# ----------------------------------------------------------------------------------
# regularisers (utils.py:180-235) - applied to the raw kernel in generator scopes
# ----------------------------------------------------------------------------------
def ortho_reg_loss(w, scale, kind="ortho_cosine"):
    c = w.shape[-1]
    W = w.reshape(-1, c)
    A = W.t() @ W                                # utils.py:197-198 / 221-222
    eye = torch.eye(c, dtype=w.dtype)
    if kind == "ortho":
        reg = A - eye
    elif kind == "ortho_cosine":                 # utils.py:180-183, 202
        na = l2_normalize(A, 1)
        nb = l2_normalize(torch.ones_like(eye) - eye, 1)
        reg = na @ nb.t()
    else:
        raise ValueError("Unknown regularization method.")
    return scale * 0.5 * (reg * reg).sum()       # tf.nn.l2_loss = sum(x^2)/2


def ortho_cosine_closed_form(w, scale):
    """Exact O(c^2) form of the cosine branch: R[i,j] = (sum_k Ahat[i,k] - Ahat[i,j]) / sqrt(c-1)
    (hard part 10)."""
    c = w.shape[-1]
    W = w.reshape(-1, c)
    A = W.t() @ W
    Ah = l2_normalize(A, 1)
    if c == 1:
        R = torch.zeros_like(A)      # rows of (1 - I) are all-zero -> l2_normalize gives 0
    else:
        R = (Ah.sum(1, keepdim=True) - Ah) / math.sqrt(c - 1)
    return scale * 0.5 * (R * R).sum()


def _maybe_regularize(vs, opt, scope, w, kind):
    """conv/fully_connected attach the regulariser only when 'generator' is in the scope
    name (ops.py:87,155); deconv attaches it regardless (ops.py:127)."""
    reg = opt.get("regularizer")
    if reg is None:
        return
    if kind != "deconv" and "generator" not in scope:
        return
    if any(n == scope for n, _ in vs.reg_losses):
        return                       # reuse=True instantiation: variable (and its loss) already exist
    if reg["type"] == "l2":          # tf.contrib.layers.l2_regularizer: scale * sum(w^2) / 2 (BigGAN.py:268-270)
        vs.reg_losses.append((scope, reg["scale"] * 0.5 * (w * w).sum()))
    elif reg["type"] == "ortho_cosine" and w.shape[-1] > 256:
        # literal form is O(c^3) (8.8 TFLOP for first/dense2 at ch=64); the closed form is the same
        # function (tests/test_oracle.py::test_ortho_cosine_closed_form)
        vs.reg_losses.append((scope, ortho_cosine_closed_form(w, reg["scale"])))
    else:
        vs.reg_losses.append((scope, ortho_reg_loss(w, reg["scale"], reg["type"])))


# ----------------------------------------------------------------------------------
# conv / deconv / dense   (ops.py:49-175)
# ----------------------------------------------------------------------------------
def conv(vs, scope, x, channels, opt, kernel=4, stride=2, pad=0, use_bias=True, _round_out=True):
    """ops.py:49-113, SN branch.  ``pad>0`` -> total padding 2*pad when H % stride == 0 else
    max(kernel - H % stride, 0), split low = total//2, high = rest (ops.py:65-76); reflect ->
    explicit tf.pad(REFLECT) + VALID conv (ops.py:81-82,94-95); zero -> TF 'SAME'."""
    cin = x.shape[-1]
    pad_type = opt.get("padding_type", "reflect")
    xin = _nchw(x)
    if pad > 0:
        h = x.shape[1]
        tot = pad * 2 if h % stride == 0 else max(kernel - (h % stride), 0)
        lo = int(tot // 2)
        hi = int(tot - lo)
        if pad_type == "reflect":
            xin = F.pad(xin, (lo, hi, lo, hi), mode="reflect")
        elif pad_type == "zero":
            # TF SAME: out = ceil(H/s); total = max((out-1)*s + k - H, 0); low = total//2
            out = -(-h // stride)
            tot = max((out - 1) * stride + kernel - h, 0)
            lo = tot // 2
            hi = tot - lo
            xin = F.pad(xin, (lo, hi, lo, hi))
        else:
            raise ValueError("Unsupported padding type: " + str(pad_type))
    w = vs.get(scope + "/kernel", (kernel, kernel, cin, channels), "trunc_normal")
    _maybe_regularize(vs, opt, scope, w, "conv")
    wn = spectral_norm(vs, scope, w) if opt.get("sn", True) else w
    if ROUND.on and channels % 8 == 0 and (cin % 8 == 0 or cin < 8):
        # the packed bf16 copy of w / sigma.  Image layers: a 3-channel INPUT enters as bf16 value + bf16 residual (~16
        # bits) against a plain bf16 kernel, so only the kernel rounds; a 3-channel OUTPUT uses a hi + lo kernel (~fp32)
        wn = _RoundFwd.apply(wn)
    y = F.conv2d(xin, wn.permute(3, 2, 0, 1), stride=stride)
    y = _nhwc(y)
    if use_bias:
        y = y + vs.get(scope + "/bias", (channels,), 0.0)
    return r_act(y) if _round_out else y


def deconv(vs, scope, x, channels, opt, kernel=4, stride=2, use_bias=True, _round_out=True):
    """ops.py:116-139: tf.nn.conv2d_transpose(x, SN(w), [B, sH, sW, C], strides s, 'SAME'),
    kernel [k, k, Cout, Cin].  TF's transposed conv is the input-gradient of a SAME conv:
    out[i] += x[a] * w[p] with i = a*s + p - pad_lo, pad_lo = max((H-1)*s + k - s*H, 0)//2, which for
    (k4,s2) and (k3,s1) equals torch conv_transpose2d(padding=1) with no kernel flip."""
    cin = x.shape[-1]
    h = x.shape[1]
    tot = max((h - 1) * stride + kernel - stride * h, 0)
    lo = tot // 2
    hi = tot - lo
    w = vs.get(scope + "/kernel", (kernel, kernel, channels, cin), "trunc_normal")
    _maybe_regularize(vs, opt, scope, w, "deconv")
    wn = spectral_norm(vs, scope, w) if opt.get("sn", True) else w
    if ROUND.on and cin % 8 == 0 and channels % 8 == 0:
        wn = _RoundFwd.apply(wn)
    # torch weight [Cin, Cout, kh, kw]; asymmetric TF padding handled by cropping
    y = F.conv_transpose2d(_nchw(x), wn.permute(3, 2, 0, 1), stride=stride)
    full = y.shape[-1]
    y = y[:, :, lo:full - hi, lo:full - hi]
    assert y.shape[-1] == stride * h
    y = _nhwc(y)
    if use_bias:
        y = y + vs.get(scope + "/bias", (channels,), 0.0)
    return r_act(y) if _round_out else y


def fully_connected(vs, scope, x, units, opt, use_bias=True, sn=None):
    """ops.py:148-175 (lrmul = 1).  flatten -> x @ SN(W) + b."""
    x = x.reshape(x.shape[0], -1)
    cin = x.shape[-1]
    w = vs.get(scope + "/kernel", (cin, units), "trunc_normal")
    _maybe_regularize(vs, opt, scope, w, "fc")
    use_sn = opt.get("sn", True) if sn is None else sn
    wn = spectral_norm(vs, scope, w) if use_sn else w
    y = x @ wn
    if use_bias:
        y = y + vs.get(scope + "/bias", (units,), 0.0)
    return y


# ----------------------------------------------------------------------------------
# activations / pooling   (ops.py:498-540)
# ----------------------------------------------------------------------------------
def prelu(vs, scope, x):
    """ops.py:532-537: relu(x) + alpha * (x - |x|) * 0.5, alpha [C] init 0."""
    alpha = vs.get(scope + "/alpha", (x.shape[-1],), 0.0)
    return torch.relu(x) + alpha * (x - x.abs()) * 0.5


class _Kink:
    """Test instrumentation for the activations' derivative jump at 0 (PReLU / ReLU / leaky ReLU).  A pre-activation
    within fp32 rounding of 0 lands on either side of the kink depending on summation order, which changes one
    element's derivative from 1 to alpha.  ``record`` (a list, when not None) receives (scope, pre-activation) of every
    activation call in call order; ``flip`` (a dict scope -> list of boolean masks / None, consumed per scope in call
    order) moves the marked elements to the other side of 0 (x -> -x there: a value change of ~1e-7; the derivative
    side the product took)."""
    record = None
    flip = None


KINK = _Kink()


def activation(vs, scope, x, opt):
    """opt['act'] (BigGAN.py:71-83): 'prelu' (ops.py:532, variable <scope>/alpha), 'relu' (529) or
    'lrelu' = tf.nn.leaky_relu(x, 0.2) (525, BigGAN.py:79)."""
    if KINK.flip is not None and KINK.flip.get(scope):
        m = KINK.flip[scope].pop(0)
        if m is not None:
            x = x + (-2.0) * x.detach() * m.to(x.dtype)
    if KINK.record is not None:
        KINK.record.append((scope, x.detach().clone()))
    kind = opt.get("act", "prelu")
    if kind == "prelu":
        return r_act(prelu(vs, scope, x))
    if kind == "relu":
        return r_act(torch.relu(x))
    if kind == "lrelu":
        return r_act(F.leaky_relu(x, 0.2))
    raise ValueError("Unknown activation function: " + str(kind))


def max_pooling(x):                               # ops.py:508-510 (even H, W: 2x2 windows)
    return _nhwc(F.max_pool2d(_nchw(x), 2, 2))


def avg_pooling(x):                               # ops.py:512-514 (even H, W)
    return _nhwc(F.avg_pool2d(_nchw(x), 2, 2))


def up_sample(x):                                 # ops.py:516-519: nearest neighbour x2
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def global_sum_pooling(x):                        # ops.py:503-506
    return x.sum(dim=(1, 2))


# ----------------------------------------------------------------------------------
# normalisation   (ops.py:580-643)
# ----------------------------------------------------------------------------------
BN_EPS = 1e-5


def _renorm_scope(scope):
    """ops.py:556-558 / 573-575: the default scope 'batch_norm' becomes 'batch_renorm'."""
    assert scope.endswith("batch_norm"), scope
    return scope[:-len("batch_norm")] + "batch_renorm"


def _renorm_clipping(opt):
    """ops.py:587-597 with GANBase.py:42-45."""
    rmax = opt.get("bn_renorm_rmax", 1.5)
    return 1.0 / rmax, rmax, opt.get("bn_renorm_dmax", 0.5)


def condition_batch_renorm(vs, scope, x, z, opt, is_training=True):
    """ops.py:645-715."""
    c = x.shape[-1]
    rmin, rmax, dmax = _renorm_clipping(opt)
    test_decay = opt.get("bn_momentum", 0.98)
    renorm_decay = opt.get("bn_renorm_momentum", 0.9)
    shared = opt.get("bn_renorm_shared", False)
    fadein = 0.9999                                             # ops.py:656 (no flag sets it)
    if not shared:
        test_decay = renorm_decay                               # ops.py:658-659
    pop_mean = vs.get(scope + "/pop_mean", (c,), 0.0, trainable=False)
    pop_var = vs.get(scope + "/pop_var", (c,), 1.0, trainable=False)
    if not shared:
        renorm_mean = vs.get(scope + "/renorm_mean", (c,), 0.0, trainable=False)
        renorm_var = vs.get(scope + "/renorm_var", (c,), 1.0, trainable=False)
        renorm_weight = vs.get(scope + "/renorm_weight", (), 0.0, trainable=False)
    else:
        renorm_mean, renorm_var, renorm_weight = pop_mean, pop_var, 1.0
    beta = fully_connected(vs, scope + "/beta", z, c, opt).reshape(-1, 1, 1, c)
    gamma = fully_connected(vs, scope + "/gamma", z, c, opt).reshape(-1, 1, 1, c)
    if not is_training:
        inv = torch.rsqrt(pop_var + BN_EPS) * gamma
        return x * inv + (beta - pop_mean * inv)
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    sigma = torch.sqrt(var + BN_EPS)
    renorm_sigma = torch.sqrt(renorm_var + BN_EPS)
    w_sigma = renorm_weight * renorm_sigma + (1 - renorm_weight) * sigma
    w_mean = renorm_weight * renorm_mean + (1 - renorm_weight) * mean
    r = torch.clamp(sigma / w_sigma, rmin, rmax).detach()
    d = torch.clamp((mean - w_mean) / w_sigma, -dmax, dmax).detach()
    vs.assign(scope + "/pop_mean", pop_mean * test_decay + mean * (1 - test_decay))
    vs.assign(scope + "/pop_var", pop_var * test_decay + var * (1 - test_decay))
    if not shared:
        vs.assign(scope + "/renorm_mean", renorm_mean * renorm_decay + mean * (1 - renorm_decay))
        vs.assign(scope + "/renorm_var", renorm_var * renorm_decay + var * (1 - renorm_decay))
        vs.assign(scope + "/renorm_weight", renorm_weight * fadein + 1.0 * (1 - fadein))
    # tf.nn.batch_normalization(x, batch_mean, batch_var, beta + d*gamma, r*gamma, eps)
    inv = torch.rsqrt(var + BN_EPS) * (r * gamma)
    return x * inv + ((beta + d * gamma) - mean * inv)


def batch_renorm(vs, scope, x, opt, is_training=True):
    """ops.py:600-609: tf.layers.batch_normalization(renorm=True) as the TF 1.15 layer computes it
    (keras/layers/normalization.py, _renorm_correction_and_moments): variables renorm_mean / renorm_stddev,
    r = sigma / max(renorm_stddev, sqrt(eps)), d = (mean - renorm_mean) / max(...), clipped, stop-gradient,
    measured BEFORE the moving averages move; scale = r*gamma, offset = d*gamma + beta; non-fused layer, so
    moving_variance follows the biased batch variance.  Two instantiations in one run (--bn_in_d: real, fake)
    are taken in that order (TF leaves the order of their reads and updates open)."""
    c = x.shape[-1]
    rmin, rmax, dmax = _renorm_clipping(opt)
    decay = opt.get("bn_momentum", 0.98)
    rdecay = opt.get("bn_renorm_momentum", 0.9)
    gamma = vs.get(scope + "/gamma", (c,), 1.0)
    beta = vs.get(scope + "/beta", (c,), 0.0)
    mm = vs.get(scope + "/moving_mean", (c,), 0.0, trainable=False)
    mv = vs.get(scope + "/moving_variance", (c,), 1.0, trainable=False)
    vs.get(scope + "/renorm_mean", (c,), 0.0, trainable=False)
    vs.get(scope + "/renorm_stddev", (c,), 1.0, trainable=False)
    if not is_training:
        inv = torch.rsqrt(mv + BN_EPS) * gamma
        return x * inv + (beta - mm * inv)
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    sigma = torch.sqrt(var + BN_EPS)
    rm_c, rs_c = vs.current(scope + "/renorm_mean"), vs.current(scope + "/renorm_stddev")
    ref = torch.clamp(rs_c, min=BN_EPS ** 0.5)
    r = torch.clamp(sigma / ref, rmin, rmax).detach()
    d = torch.clamp((mean - rm_c) / ref, -dmax, dmax).detach()
    vs.assign(scope + "/renorm_mean", rm_c * rdecay + mean.detach() * (1 - rdecay))
    vs.assign(scope + "/renorm_stddev", rs_c * rdecay + sigma.detach() * (1 - rdecay))
    mm_c, mv_c = vs.current(scope + "/moving_mean"), vs.current(scope + "/moving_variance")
    vs.assign(scope + "/moving_mean", mm_c * decay + mean * (1 - decay))
    vs.assign(scope + "/moving_variance", mv_c * decay + var * (1 - decay))
    inv = torch.rsqrt(var + BN_EPS) * (r * gamma)
    return x * inv + ((beta + d * gamma) - mean * inv)


def condition_batch_norm(vs, scope, x, z, opt, is_training=True):
    """ops.py:611-643.  beta/gamma = SN dense of z (bias init 0; gamma is NOT 1 + ...);
    training: biased batch moments over (B,H,W), EMA of pop stats with decay ``momentum``."""
    if opt.get("bn_type", "batch_norm") == "batch_renorm":        # cond_bn dispatch, ops.py:563-578
        return condition_batch_renorm(vs, _renorm_scope(scope), x, z, opt, is_training)
    c = x.shape[-1]
    decay = opt.get("bn_momentum", 0.98)
    pop_mean = vs.get(scope + "/pop_mean", (c,), 0.0, trainable=False)
    pop_var = vs.get(scope + "/pop_var", (c,), 1.0, trainable=False)
    beta = fully_connected(vs, scope + "/beta", z, c, opt).reshape(-1, 1, 1, c)
    gamma = fully_connected(vs, scope + "/gamma", z, c, opt).reshape(-1, 1, 1, c)
    if is_training:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))           # tf.nn.moments: biased
        vs.assign(scope + "/pop_mean", pop_mean * decay + mean * (1 - decay))
        vs.assign(scope + "/pop_var", pop_var * decay + var * (1 - decay))
    else:
        mean, var = pop_mean, pop_var
    # tf.nn.batch_normalization(x, mean, var, offset=beta, scale=gamma, eps)
    inv = torch.rsqrt(var + BN_EPS) * gamma
    return x * inv + (beta - mean * inv)


def batch_norm(vs, scope, x, opt, is_training=True):
    """ops.py:580-585: tf.layers.batch_normalization(momentum, eps=1e-5).  Normalises with the
    biased batch variance; the fused kernel's moving-variance update uses the Bessel-corrected
    one (TF documentation; affects sampling only)."""
    if opt.get("bn_type", "batch_norm") == "batch_renorm":        # bn dispatch, ops.py:546-561
        return batch_renorm(vs, _renorm_scope(scope), x, opt, is_training)
    c = x.shape[-1]
    decay = opt.get("bn_momentum", 0.98)
    gamma = vs.get(scope + "/gamma", (c,), 1.0)
    beta = vs.get(scope + "/beta", (c,), 0.0)
    mm = vs.get(scope + "/moving_mean", (c,), 0.0, trainable=False)
    mv = vs.get(scope + "/moving_variance", (c,), 1.0, trainable=False)
    if is_training:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
        n = x.shape[0] * x.shape[1] * x.shape[2]
        # with --bn_in_d the discriminator is instantiated twice per run (real, fake): the second update
        # starts from the first one's result (assign_moving_average); order real -> fake as in d_forward
        mm_c, mv_c = vs.current(scope + "/moving_mean"), vs.current(scope + "/moving_variance")
        vs.assign(scope + "/moving_mean", mm_c * decay + mean * (1 - decay))
        vs.assign(scope + "/moving_variance", mv_c * decay + var * (n / max(n - 1, 1)) * (1 - decay))
    else:
        mean, var = mm, mv
    inv = torch.rsqrt(var + BN_EPS) * gamma
    return x * inv + (beta - mean * inv)


# ----------------------------------------------------------------------------------
# residual blocks / attention   (ops.py:187-313, 467-492)
# ----------------------------------------------------------------------------------
def resblock(vs, scope, x_init, channels, opt, use_bias=True):
    """ops.py:187-198."""
    x = conv(vs, scope + "/res1/conv_0", x_init, channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias)
    if opt.get("bn_in_d"):
        x = batch_norm(vs, scope + "/res1/batch_norm", x, opt, True)
    x = activation(vs, scope + "/res1/prelu", x, opt)
    x = conv(vs, scope + "/res2/conv_0", x, channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias)
    if opt.get("bn_in_d"):
        x = batch_norm(vs, scope + "/res2/batch_norm", x, opt, True)
    return r_act(x + x_init)


def upconv(vs, scope, x, channels, opt, use_bias=True):
    """ops.py:200-218: --upsampling_method deconv3 / deconv4 (default) / deconv6 = transposed conv k, stride 2."""
    m = opt.get("upsampling_method", "deconv4")
    if m == "resize_conv":
        return conv(vs, scope + "/conv_0", up_sample(x), channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias)
    k = {"deconv3": 3, "deconv4": 4, "deconv6": 6}.get(m)
    if k is None:
        raise ValueError("Invalid upsampling method specified: " + str(m))
    return deconv(vs, scope + "/deconv_0", x, channels, opt, kernel=k, stride=2, use_bias=use_bias)


def g_conv(vs, scope, x, channels, opt, use_bias=True, _round_out=True):
    """ops.py:220-230: --g_conv deconv3 (default) / deconv4 (stride 1) / conv3 (reflect-padded conv)."""
    m = opt.get("g_conv", "deconv3")
    if m == "deconv3":
        return deconv(vs, scope + "/deconv_0", x, channels, opt, kernel=3, stride=1, use_bias=use_bias, _round_out=_round_out)
    if m == "deconv4":
        return deconv(vs, scope + "/deconv_0", x, channels, opt, kernel=4, stride=1, use_bias=use_bias, _round_out=_round_out)
    if m == "conv3":
        return conv(vs, scope + "/conv_0", x, channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias,
                    _round_out=_round_out)
    raise ValueError("Invalid generator convolution type specified: " + str(m))


def resblock_up_condition(vs, scope, x_init, z, channels, opt, use_bias=True, is_training=True):
    """ops.py:250-266; defaults: upconv = deconv k4 s2 (ops.py:203-204), g_conv = deconv k3 s1 (221-222)."""
    x = condition_batch_norm(vs, scope + "/res1/batch_norm", x_init, z, opt, is_training)
    x = activation(vs, scope + "/res1/prelu", x, opt)
    x = upconv(vs, scope + "/res1", x, channels, opt, use_bias=use_bias)
    x = condition_batch_norm(vs, scope + "/res2/batch_norm", x, z, opt, is_training)
    x = activation(vs, scope + "/res2/prelu", x, opt)
    x = g_conv(vs, scope + "/res2", x, channels, opt, use_bias=use_bias, _round_out=False)
    skip = upconv(vs, scope + "/skip", x_init, channels, opt, use_bias=use_bias)
    return r_act(x + skip)                      # (the product adds the skip branch in the last kernel's epilogue)


def resblock_up_cond_deep(vs, scope, x_init, z, channels_out, opt, upscale=True, use_bias=True, is_training=True):
    """ops.py:317-358 (--deep)."""
    cin = x_init.shape[-1]
    inner = round_up((cin + channels_out) // 6, 8)
    x = condition_batch_norm(vs, scope + "/bottleneck/batch_norm", x_init, z, opt, is_training)
    x = activation(vs, scope + "/bottleneck/prelu", x, opt)
    x = conv(vs, scope + "/bottleneck/conv_0", x, inner, opt, kernel=1, stride=1, pad=0, use_bias=False)
    x = condition_batch_norm(vs, scope + "/upscale/batch_norm", x, z, opt, is_training)
    x = activation(vs, scope + "/upscale/prelu", x, opt)
    if upscale:
        x = upconv(vs, scope + "/upscale", x, inner, opt, use_bias=False)
    x = g_conv(vs, scope + "/inner1", x, inner, opt, use_bias=False)
    x = condition_batch_norm(vs, scope + "/inner1/batch_norm", x, z, opt, is_training)
    x = activation(vs, scope + "/inner1/prelu", x, opt)
    x = g_conv(vs, scope + "/inner2", x, inner, opt, use_bias=False)
    x = batch_norm(vs, scope + "/inner2/batch_norm", x, opt, is_training)
    x = activation(vs, scope + "/inner2/prelu", x, opt)
    x = conv(vs, scope + "/proj/conv_0", x, channels_out, opt, kernel=1, stride=1, pad=0, use_bias=use_bias)
    skip = x_init
    if upscale:
        kept = x_init[..., :channels_out] if cin != channels_out else x_init
        skip = upconv(vs, scope + "/skip", kept, channels_out, opt, use_bias=use_bias)
    return x + skip


def resblock_down_deep(vs, scope, x_init, channels_out, opt, downscale=True, use_bias=True):
    """ops.py:360-401 (--deep)."""
    cin = x_init.shape[-1]
    inner = round_up((cin + channels_out) // 6, 8)
    x = x_init
    for name, k, pad in (("bottleneck", 1, 0), ("inner1", 3, 1), ("inner2", 3, 1)):
        if opt.get("bn_in_d"):
            x = batch_norm(vs, scope + "/" + name + "/batch_norm", x, opt, True)
        x = activation(vs, scope + "/" + name + "/prelu", x, opt)
        x = conv(vs, scope + "/" + name + "/conv_0", x, inner, opt, kernel=k, stride=1, pad=pad, use_bias=use_bias)
    x = activation(vs, scope + "/downscale/prelu", x, opt)
    if downscale:
        x = avg_pooling(x)
    x = conv(vs, scope + "/proj/conv_0", x, channels_out, opt, kernel=1, stride=1, pad=0, use_bias=use_bias)
    skip = x_init
    if downscale:
        skip = avg_pooling(skip)
    if cin != channels_out:
        dense = conv(vs, scope + "/skip/conv_0", skip, channels_out - cin, opt, kernel=1, stride=1, pad=0,
                     use_bias=use_bias)
        skip = torch.cat([skip, dense], dim=-1)
    return x + skip


def downconv(vs, scope, x, channels, opt, use_bias=True, method=None):
    """ops.py:269-291: strided_conv3 (default), resize_conv1 / resize_conv3 = conv k1 / k3 stride 1 + avg pool."""
    m = method or opt.get("downsampling_method", "strided_conv3")
    if m == "strided_conv3":
        return conv(vs, scope + "/conv_0", x, channels, opt, kernel=3, stride=2, pad=1, use_bias=use_bias)
    if m == "resize_conv1":
        return avg_pooling(conv(vs, scope + "/conv_0", x, channels, opt, kernel=1, stride=1, pad=0, use_bias=use_bias))
    if m == "resize_conv3":
        return avg_pooling(conv(vs, scope + "/conv_0", x, channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias))
    raise ValueError("Invalid downsampling method specified: " + str(m))


def resblock_down(vs, scope, x_init, channels, opt, use_bias=True):
    """ops.py:293-313; the residual path uses resize_conv3 whenever the method is not strided_conv3 (299-302)."""
    x = x_init
    if opt.get("bn_in_d"):
        x = batch_norm(vs, scope + "/res1/batch_norm", x, opt, True)
    x = activation(vs, scope + "/res1/prelu", x, opt)
    res_method = opt.get("downsampling_method", "strided_conv3")
    if res_method != "strided_conv3":
        res_method = "resize_conv3"
    x = downconv(vs, scope + "/res1", x, channels, opt, use_bias=use_bias, method=res_method)
    if opt.get("bn_in_d"):
        x = batch_norm(vs, scope + "/res2/batch_norm", x, opt, True)
    x = activation(vs, scope + "/res2/prelu", x, opt)
    x = conv(vs, scope + "/res2/conv_0", x, channels, opt, kernel=3, stride=1, pad=1, use_bias=use_bias, _round_out=False)
    skip = downconv(vs, scope + "/skip", x_init, channels, opt, use_bias=use_bias)
    return r_act(x + skip)


def self_attention_2(vs, scope, x, channels, opt):
    """ops.py:467-492."""
    ub = opt.get("self_attention_bias", False)
    b, h, w_, _ = x.shape
    # product, bf16-resident mode: when the three projections fit one fused GEMM (widths multiples of 4, their sum a
    # multiple of 8) f | g | h are ONE bf16 tensor feeding the fused bf16 attention (bf16 probabilities into P V);
    # otherwise (narrow test models) the projections are written in fp32 and the attention core runs in fp32
    d_qk, d_v = channels // 8, channels // 2
    fused = (ROUND.on and x.shape[-1] % 8 == 0 and channels % 8 == 0 and d_v % 8 == 0 and d_qk % 4 == 0
             and (2 * d_qk + d_v) % 8 == 0)
    rq = (lambda t: _RoundAct.apply(t)) if fused else (lambda t: t)
    f = rq(conv(vs, scope + "/f_conv", x, d_qk, opt, kernel=1, stride=1, use_bias=ub, _round_out=False))
    f = max_pooling(f)
    g = rq(conv(vs, scope + "/g_conv", x, d_qk, opt, kernel=1, stride=1, use_bias=ub, _round_out=False))
    hh = rq(conv(vs, scope + "/h_conv", x, d_v, opt, kernel=1, stride=1, use_bias=ub, _round_out=False))
    hh = max_pooling(hh)
    s = g.reshape(b, -1, g.shape[-1]) @ f.reshape(b, -1, f.shape[-1]).transpose(1, 2)
    beta = torch.softmax(s, dim=-1)
    if fused:
        beta = _RoundFwd.apply(beta)
    o = beta @ hh.reshape(b, -1, hh.shape[-1])
    gamma = vs.get(scope + "/gamma", (1,), 0.0)
    o = r_act(o.reshape(b, h, w_, channels // 2)) if (fused or channels % 8 == 0) else o.reshape(b, h, w_, channels // 2)
    o = conv(vs, scope + "/attn_conv", o, channels, opt, kernel=1, stride=1, use_bias=ub)
    return r_act(gamma * o + x)


# ----------------------------------------------------------------------------------
# losses   (ops.py:753-848)
# ----------------------------------------------------------------------------------
def flood_loss(loss, flood_level):                # ops.py:847-848
    return (loss - flood_level).abs() + flood_level


def _sce(labels_one, logits):
    """tf.nn.sigmoid_cross_entropy_with_logits with labels all ones (True) or all zeros (False)."""
    return F.softplus(-logits) if labels_one else F.softplus(logits)


def discriminator_loss(loss_func, real, fake, flood_level=0):
    """ops.py:753-797 (the penalty of the wgan / dragan types is added by the caller, BigGAN.py:880)."""
    if "wgan" in loss_func:
        real_loss, fake_loss = -real.mean(), fake.mean()                    # ops.py:757-759
    elif loss_func == "lsgan":
        real_loss, fake_loss = ((real - 1.0) ** 2).mean(), (fake ** 2).mean()
    elif loss_func == "ra-lsgan":
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        real_loss, fake_loss = ((d_xr - 1.0) ** 2).mean(), ((d_xf + 1.0) ** 2).mean()
    elif loss_func in ("gan", "dragan"):
        real_loss, fake_loss = _sce(True, real).mean(), _sce(False, fake).mean()
    elif loss_func in ("ra-gan", "ra-dragan"):
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        real_loss, fake_loss = _sce(True, d_xr).mean(), _sce(False, d_xf).mean()
    elif loss_func == "ra-hinge":
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        real_loss, fake_loss = torch.relu(1.0 - d_xr).mean(), torch.relu(1.0 + d_xf).mean()
    elif loss_func == "hinge":
        real_loss, fake_loss = torch.relu(1.0 - real).mean(), torch.relu(1.0 + fake).mean()   # ops.py:788-790
    else:
        raise NotImplementedError(loss_func)
    loss = real_loss + fake_loss
    if flood_level:
        loss = flood_loss(loss, flood_level)                                # ops.py:794-795
    return loss


def generator_loss(loss_func, fake, real=None, flood_level=0):
    """ops.py:799-840."""
    real_loss = 0.0
    if "wgan" in loss_func:
        fake_loss = -fake.mean()                                            # ops.py:804-805
    elif loss_func == "lsgan":
        fake_loss = ((fake - 1.0) ** 2).mean()
    elif loss_func == "ra-lsgan":
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        real_loss, fake_loss = ((d_xr + 1.0) ** 2).mean(), ((d_xf - 1.0) ** 2).mean()
    elif loss_func in ("gan", "dragan"):
        fake_loss = _sce(True, fake).mean()
    elif loss_func in ("ra-gan", "ra-dragan"):
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        fake_loss, real_loss = _sce(True, d_xf).mean(), _sce(False, d_xr).mean()
    elif loss_func == "ra-hinge":
        d_xr, d_xf = real - fake.mean(), fake - real.mean()
        real_loss, fake_loss = torch.relu(1.0 - d_xf).mean(), torch.relu(1.0 + d_xr).mean()
    elif loss_func == "hinge":
        fake_loss = -fake.mean()                                            # ops.py:832-833
    else:
        raise NotImplementedError(loss_func)
    loss = fake_loss + real_loss
    if flood_level:
        loss = flood_loss(loss, flood_level)                                # ops.py:837-838
    return loss


def cls_loss_logistic(truth, answer, cls_weights):
    """utils.py:366-369: mean(sigmoid_cross_entropy_with_logits(labels, logits) * w)."""
    ce = torch.clamp(answer, min=0) - answer * truth + torch.log1p(torch.exp(-answer.abs()))
    return (ce * cls_weights).mean()


# ----------------------------------------------------------------------------------
# DiffAugment   (DiffAugment_tf.py:8-73)
# ----------------------------------------------------------------------------------
def diffaugment_params(S):
    """Integer constants of the translation / cutout ops for image size S."""
    shift = int(S * 0.125 + 0.5)                  # DiffAugment_tf.py:43
    cs = int(S * 0.5 + 0.5)                       # DiffAugment_tf.py:56
    off_max = S + (1 - cs % 2)                    # exclusive maxval, DiffAugment_tf.py:57-58
    return shift, cs, off_max


def draw_diffaugment(rng, B, S):
    """The 7 draws of one DiffAugment('color,translation,cutout') call, in graph order."""
    shift, cs, off_max = diffaugment_params(S)
    return {
        "u_b": rng.random(B).astype(np.float32),
        "u_s": rng.random(B).astype(np.float32),
        "u_c": rng.random(B).astype(np.float32),
        "t_x": rng.integers(-shift, shift + 1, B).astype(np.int32),
        "t_y": rng.integers(-shift, shift + 1, B).astype(np.int32),
        "o_x": rng.integers(0, off_max, B).astype(np.int32),
        "o_y": rng.integers(0, off_max, B).astype(np.int32),
    }


def translation_index(S, t):
    """Closed form of DiffAugment_tf.py:46-49 along one axis: source index i + t, or -1 (zero fill)."""
    i = np.arange(S)[None, :] + np.asarray(t)[:, None]
    return np.where((i >= 0) & (i < S), i, -1)


def cutout_mask(S, o_x, o_y):
    """Closed form of DiffAugment_tf.py:53-66: mask [B,S,S] with the clipped box zeroed."""
    _, cs, _ = diffaugment_params(S)
    B = len(o_x)
    m = np.ones((B, S, S), dtype=np.float32)
    for b in range(B):
        r0 = max(0, int(o_x[b]) - cs // 2)
        r1 = min(S - 1, int(o_x[b]) - cs // 2 + cs - 1)
        c0 = max(0, int(o_y[b]) - cs // 2)
        c1 = min(S - 1, int(o_y[b]) - cs // 2 + cs - 1)
        m[b, r0:r1 + 1, c0:c1 + 1] = 0.0
    return m


def translation_literal(x, t_x, t_y):
    """Literal NumPy emulation of DiffAugment_tf.py:40-50 (pad + gather_nd + transpose)."""
    B, S = x.shape[0], x.shape[1]
    gx = np.clip(np.arange(S)[None, :] + t_x[:, None] + 1, 0, S + 1)
    gy = np.clip(np.arange(S)[None, :] + t_y[:, None] + 1, 0, S + 1)
    xp = np.pad(x, [(0, 0), (1, 1), (0, 0), (0, 0)])
    x1 = np.stack([xp[b][gx[b]] for b in range(B)])
    xt = np.transpose(x1, (0, 2, 1, 3))
    xtp = np.pad(xt, [(0, 0), (1, 1), (0, 0), (0, 0)])
    x2 = np.stack([xtp[b][gy[b]] for b in range(B)])
    return np.transpose(x2, (0, 2, 1, 3))


def cutout_literal(x, o_x, o_y):
    """Literal NumPy emulation of DiffAugment_tf.py:53-66 (meshgrid, clip, scatter_nd, max(1-.,0))."""
    B, S = x.shape[0], x.shape[1]
    _, cs, _ = diffaugment_params(S)
    gb, gx, gy = np.meshgrid(np.arange(B), np.arange(cs), np.arange(cs), indexing="ij")
    ix = np.clip(gx + o_x[:, None, None] - cs // 2, 0, S - 1)
    iy = np.clip(gy + o_y[:, None, None] - cs // 2, 0, S - 1)
    acc = np.zeros((B, S, S), dtype=np.float32)
    np.add.at(acc, (gb, ix, iy), 1.0)             # scatter_nd sums duplicates
    mask = np.maximum(1 - acc, 0)
    return x * mask[..., None]


def diffaugment(x, draws, policy="color,translation,cutout"):
    """DiffAugment_tf.py:8-17 on a torch tensor [B,S,S,C] with explicit draws (differentiable)."""
    if not policy:
        return x
    B, S = x.shape[0], x.shape[1]
    dt = x.dtype
    for p in policy.split(","):
        if p == "color":
            ub = torch.tensor(draws["u_b"], dtype=dt).reshape(B, 1, 1, 1)
            us = torch.tensor(draws["u_s"], dtype=dt).reshape(B, 1, 1, 1)
            uc = torch.tensor(draws["u_c"], dtype=dt).reshape(B, 1, 1, 1)
            x = x + (ub - 0.5)                                    # :20-23
            m = x.mean(dim=3, keepdim=True)                       # :26-30
            x = (x - m) * (us * 2) + m
            m = x.mean(dim=(1, 2, 3), keepdim=True)               # :33-37
            x = (x - m) * (uc + 0.5) + m
        elif p == "translation":
            ix = torch.tensor(translation_index(S, draws["t_x"]))      # [B,S] rows
            iy = torch.tensor(translation_index(S, draws["t_y"]))      # [B,S] cols
            bidx = torch.arange(B).reshape(B, 1, 1)
            g = x[bidx, ix.clamp(min=0)[:, :, None], iy.clamp(min=0)[:, None, :]]
            valid = ((ix >= 0)[:, :, None] & (iy >= 0)[:, None, :]).to(dt)
            x = g * valid[..., None]
        elif p == "cutout":
            m = torch.tensor(cutout_mask(S, draws["o_x"], draws["o_y"]), dtype=dt)
            x = x * m[..., None]
        else:
            raise KeyError(p)
    return x
