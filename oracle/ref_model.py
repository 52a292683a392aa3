"""Oracle restatement of ``BigGAN.generator`` / ``discriminator`` / ``build_model`` train ops
(``/root/reference/BigGAN.py:246-961``) and the step glue of ``utils.py:242-320``.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PARITY UNPINNED (no TensorFlow here).
Default-flag topology only; ``--gan_type hinge``.
"""
from collections import OrderedDict
import copy

import numpy as np
import torch

from . import ref_ops as R


class Config:
    """The subset of main.py flags (main.py:9-147) that shapes the hot path, with their defaults."""

    def __init__(self, **kw):
        self.img_size = 128
        self.ch = 64
        self.d_ch = 0
        self.batch_size = 16
        self.z_dim = 256
        self.c_dim = 3
        self.first_split_ratio = 3
        self.sn = True
        self.bias_in_d = False
        self.bias_in_sa = True
        self.bn_momentum = 0.98
        self.bn_type = "batch_norm"           # main.py:43
        self.bn_renorm_rmax = 1.5             # main.py:45-48
        self.bn_renorm_dmax = 0.5
        self.bn_renorm_momentum = 0.9
        self.bn_renorm_shared = False
        self.g_regularization = "ortho_cosine"
        self.g_regularization_factor = 1e-4
        self.conv_padding = "reflect"
        self.activation = "prelu"             # main.py:63
        self.upsampling_method = "deconv4"    # main.py:52
        self.g_conv = "deconv3"               # main.py:54
        self.deep = False                     # main.py (--deep)
        self.g_first_level_dense_layer = True # main.py:90
        self.g_other_level_dense_layer = False
        self.g_no_last_resblock = False
        self.d_cls_dense_layers = False       # main.py:94
        self.downsampling_method = "strided_conv3"   # main.py:53
        self.bn_in_d = False                  # main.py:40
        self.g_grow_factor = 2.0
        self.d_grow_factor = 2.0
        self.gan_type = "hinge"
        self.ld = 10.0                        # main.py:64
        self.d_flood = 0.1
        self.g_flood = 0.05
        self.da_policy = "full"
        self.n_labels = 0
        self.d_cls_loss_weight = 5.0
        self.g_cls_loss_weight = 1.0
        self.g_lr = 5e-5
        self.d_lr = 2e-4
        self.beta1 = 0.0
        self.beta2 = 0.9
        self.moving_decay = 0.999
        self.d_compat_use_sn_in_critic_output = True
        self.d_compat_use_sn_in_classification = False
        self.extension_32 = False     # labelled extension: img_size 32 (not supported by the reference)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)
        if self.d_ch <= 0:
            self.d_ch = self.ch                                  # BigGAN.py:153-154
        if self.da_policy == "full":
            self.da_policy = "color,translation,cutout"           # GANBase.py:53-55
        self.depth = self.img_size.bit_length() - 2               # BigGAN.py:19

    def scale_channels(self, ch, factor):                         # BigGAN.py:236-240
        return R.round_up(int(ch * factor), 8)

    def g_block_info(self):                                       # BigGAN.py:290-294
        s = self.img_size
        if s == 64:
            return {"counts": [1, 1, 1, 1], "sa_index": 3}
        if s == 128:
            return {"counts": [1, 1, 1, 1, 1], "sa_index": 4}
        if s == 256:
            return {"counts": [1, 2, 1, 1, 1], "sa_index": 3}
        if s == 512:
            return {"counts": [1, 2, 1, 1, 2], "sa_index": 3}
        if s == 32 and self.extension_32:
            return {"counts": [1, 1, 1], "sa_index": 2}
        raise ValueError("Invalid image size specified: " + str(s))

    def d_block_info(self):                                       # BigGAN.py:607-611
        s = self.img_size
        if s == 64:
            return {"counts": [1, 1, 1, 1], "sa_index": 1}
        if s == 128:
            return {"counts": [1, 1, 1, 1, 1], "sa_index": 1}
        if s == 256:
            return {"counts": [1, 1, 1, 2, 1], "sa_index": 2}
        if s == 512:
            return {"counts": [1, 2, 1, 1, 2], "sa_index": 2}
        if s == 32 and self.extension_32:
            return {"counts": [1, 1, 1], "sa_index": 1}
        raise ValueError("Invalid image size specified: " + str(s))

    def z_split_sizes(self):                                      # BigGAN.py:280-288
        if self.first_split_ratio > 1:
            split = self.z_dim // (self.depth - 1 + self.first_split_ratio)
            first = self.z_dim - (self.depth - 1) * split
        else:
            split = self.z_dim // self.depth
            first = split
        return [first] + [split] * (self.depth - 1)


def _conv_opt(cfg, training, generator):
    opt = {"sn": cfg.sn, "padding_type": cfg.conv_padding, "bn_momentum": cfg.bn_momentum,
           "self_attention_bias": cfg.bias_in_sa, "regularizer": None, "act": cfg.activation,
           "bn_in_d": cfg.bn_in_d, "upsampling_method": cfg.upsampling_method, "g_conv": cfg.g_conv,
           "downsampling_method": cfg.downsampling_method, "bn_type": cfg.bn_type,
           "bn_renorm_rmax": cfg.bn_renorm_rmax, "bn_renorm_dmax": cfg.bn_renorm_dmax,
           "bn_renorm_momentum": cfg.bn_renorm_momentum, "bn_renorm_shared": cfg.bn_renorm_shared}
    if generator and training and cfg.g_regularization != "none":           # BigGAN.py:257-274
        opt["regularizer"] = {"scale": cfg.g_regularization_factor, "type": cfg.g_regularization}
    return opt


def generator(vs, cfg, z, cls_z=None, is_training=True):
    """BigGAN.py:246-582, default branches.  z [B,1,1,z_dim] -> [B,S,S,c_dim]."""
    opt = _conv_opt(cfg, is_training, True)
    G = "generator"
    info = cfg.g_block_info()
    counts = info["counts"]
    sizes = cfg.z_split_sizes()
    z = z.reshape(z.shape[0], 1, 1, -1)
    z_split = list(torch.split(z, sizes, dim=-1))                           # BigGAN.py:335
    if cfg.n_labels > 0:                                                    # BigGAN.py:346-365
        cz = cls_z.reshape(-1, 1, 1, cfg.n_labels)
        z_split = [torch.cat([zz, cz], dim=-1) for zz in z_split]
    nxt = iter(range(len(z_split)))

    n_blocks = len(counts)
    ch_mul = 2 ** (n_blocks - 1)                                            # BigGAN.py:427
    ch = cfg.scale_channels(cfg.ch, cfg.g_grow_factor ** (n_blocks - 0 - 1))  # BigGAN.py:428

    zi = next(nxt)
    f_width = R.round_up((sizes[zi] + cfg.n_labels) * 1.85, 8)               # BigGAN.py:433
    first = G if cfg.activation == "relu" else G + "/first"              # BigGAN.py:434-443: no scope with relu
    if not cfg.g_first_level_dense_layer:                                   # BigGAN.py:444
        x = R.fully_connected(vs, G + "/dense", z_split[zi], 4 * 4 * ch, opt)
    else:
        x = R.fully_connected(vs, first + "/dense1", z_split[zi], f_width, opt)
        x = R.activation(vs, first + "/prelu", x, opt)
        x = R.fully_connected(vs, first + "/dense2", x, 4 * 4 * ch, opt)
    x = R.r_act(x.reshape(-1, 4, 4, ch))                                    # BigGAN.py:446 (bf16 trunk from here when R.ROUND.on)

    b_i = 0
    for block_count in counts:                                              # BigGAN.py:449-489
        scope = "resblock_up_" + str(ch_mul)
        for sb_i in range(block_count):
            zi = next(nxt)
            if block_count > 1:
                scope = scope + "_" + str(sb_i)                             # cumulative (BigGAN.py:455)
            block_z = z_split[zi]
            if cfg.g_other_level_dense_layer:                               # BigGAN.py:457-462
                zs = G + "/z" + str(ch_mul)
                zw = R.round_up((sizes[zi] + cfg.n_labels) * 1.25, 8)
                block_z = R.activation(vs, zs + "/prelu", R.fully_connected(vs, zs + "/dense1", block_z, zw, opt), opt)
                block_z = block_z.reshape(block_z.shape[0], 1, 1, -1)
            is_last = sb_i == block_count - 1 and b_i == len(counts) - 1
            if cfg.g_no_last_resblock and is_last:                          # BigGAN.py:468-473
                sc = G + "/" + scope
                x = R.upconv(vs, sc, x, ch, opt, use_bias=False)
                x = R.condition_batch_norm(vs, sc + "/batch_norm", x, block_z, opt, is_training)
                x = R.activation(vs, sc + "/prelu", x, opt)
                x = R.g_conv(vs, sc, x, ch, opt, use_bias=False)
            elif cfg.deep:                                                  # BigGAN.py:475-477
                x = R.resblock_up_cond_deep(vs, G + "/" + scope, x, block_z, ch, opt, True, True, is_training)
                x = R.resblock_up_cond_deep(vs, G + "/" + scope + "_2", x, block_z, ch, opt, False, True, is_training)
            else:
                x = R.resblock_up_condition(vs, G + "/" + scope, x, block_z, ch, opt,
                                            use_bias=False, is_training=is_training)
        b_i += 1
        if b_i == info["sa_index"]:
            x = R.self_attention_2(vs, G + "/self_attention", x, ch, opt)
        ch = cfg.scale_channels(cfg.ch, cfg.g_grow_factor ** (n_blocks - b_i - 1))
        ch_mul //= 2

    x = R.batch_norm(vs, G + "/batch_norm", x, opt, is_training)            # BigGAN.py:491
    x = R.activation(vs, G + "/prelu", x, opt)                                        # BigGAN.py:492
    x = R.conv(vs, G + "/G_logit", x, cfg.c_dim, opt, kernel=3, stride=1, pad=1, use_bias=False)  # :570
    return torch.tanh(x)                                                    # :580


def discriminator(vs, cfg, x):
    """BigGAN.py:591-715, default branches.  Returns {"real": [B,1]} (+ "cls" when n_labels>0)."""
    opt = _conv_opt(cfg, True, False)
    D = "discriminator"
    info = cfg.d_block_info()
    ch = cfg.scale_channels(cfg.d_ch, cfg.d_grow_factor ** 0)               # BigGAN.py:605
    b_i = 0
    ch_mul = 1
    for block_count in info["counts"]:                                      # BigGAN.py:624-664
        scope = "resblock_down_" + str(ch_mul)
        for sb_i in range(block_count):
            if block_count > 1:
                scope = scope + "_" + str(sb_i)
            if cfg.deep:                                                    # BigGAN.py:629-631
                x = R.resblock_down_deep(vs, D + "/" + scope, x, ch, opt, True, cfg.bias_in_d)
                x = R.resblock_down_deep(vs, D + "/" + scope + "_2", x, ch, opt, False, cfg.bias_in_d)
            else:
                x = R.resblock_down(vs, D + "/" + scope, x, ch, opt, use_bias=cfg.bias_in_d)
        b_i += 1
        if b_i == info["sa_index"]:
            x = R.self_attention_2(vs, D + "/self_attention", x, ch, opt)
        ch = cfg.scale_channels(cfg.d_ch, cfg.d_grow_factor ** b_i)
        ch_mul *= 2
    ch = cfg.scale_channels(cfg.d_ch, cfg.d_grow_factor ** (b_i - 1))       # BigGAN.py:666
    x = R.resblock(vs, D + "/resblock", x, ch, opt, use_bias=cfg.bias_in_d)  # :668
    x = R.activation(vs, D + "/prelu", x, opt)                                        # :669
    feat = R.global_sum_pooling(x)                                          # :671
    out = {}
    out["real"] = R.fully_connected(vs, D + "/D_logit", feat, 1, opt,
                                    sn=cfg.d_compat_use_sn_in_critic_output)   # :681-682, 1482-1487
    if cfg.n_labels > 0:                                                    # :689-701
        csn = cfg.d_compat_use_sn_in_classification
        if cfg.d_cls_dense_layers:                                          # :690-698
            C = D + "/classification"
            u1 = R.round_up(ch / 16.0 + cfg.n_labels * 1.25, 8)
            y = R.activation(vs, C + "/prelu", R.fully_connected(vs, C + "/dense1", feat, u1, opt, sn=csn), opt)
            u2 = R.round_up(u1 / 4.0 + cfg.n_labels * 1.1, 4)
            y = R.activation(vs, C + "/prelu_1", R.fully_connected(vs, C + "/dense2", y, u2, opt, sn=csn), opt)
            out["cls"] = R.fully_connected(vs, C + "/DC_logit", y, cfg.n_labels, opt, sn=csn)
        else:
            out["cls"] = R.fully_connected(vs, D + "/DC_logit", feat, cfg.n_labels, opt, sn=csn)
    return out


# ----------------------------------------------------------------------------------
# optimiser state: TF AdamOptimizer + MovingAverageOptimizer  (BigGAN.py:923-930)
# ----------------------------------------------------------------------------------
class AdamTF:
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; theta -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, lr, beta1, beta2, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = {}
        self.v = {}

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for k, p in params.items():
                g = grads[k]
                if k not in self.m:
                    self.m[k] = torch.zeros_like(p)
                    self.v[k] = torch.zeros_like(p)
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))


class Trainer:
    """One training iteration = D step then G step (BigGAN.py:1061-1084, n_critic=1), each a separate
    run with its own z, real batch and DiffAugment draws (utils.py:296-297)."""

    def __init__(self, cfg, dtype=torch.float64, seed=42):
        self.cfg = cfg
        self.vs = R.VarStore(dtype, seed)
        self.d_opt = AdamTF(cfg.d_lr, cfg.beta1, cfg.beta2)
        self.g_opt = AdamTF(cfg.g_lr, cfg.beta1, cfg.beta2)
        self.ema = {}
        self.dtype = dtype
        self.built = False

    def build(self):
        """Instantiate every variable once (tf.global_variables_initializer)."""
        cfg = self.cfg
        B = 2
        z = torch.zeros(B, 1, 1, cfg.z_dim, dtype=self.dtype)
        cz = torch.zeros(B, cfg.n_labels, dtype=self.dtype) if cfg.n_labels else None
        with torch.no_grad():
            img = generator(self.vs, cfg, z, cz, True)
            discriminator(self.vs, cfg, img)
        self.vs.state_updates.clear()
        self.vs.reg_losses = []
        self.vs.frozen = True
        self.built = True
        for k, p in self.g_params().items():                  # EMA shadows start at the initial value
            self.ema[k] = p.detach().clone()
        return self

    def g_params(self):
        return OrderedDict((k, v) for k, v in self.vs.vars.items() if self.vs.trainable[k] and "generator" in k)

    def d_params(self):
        return OrderedDict((k, v) for k, v in self.vs.vars.items() if self.vs.trainable[k] and "discriminator" in k)

    def _t(self, a):
        return torch.tensor(np.asarray(a), dtype=self.dtype)

    def gradient_penalty_type(self):
        """BigGAN.py:130-135 (d_loss_func == gan_type here)."""
        gt = self.cfg.gan_type
        return gt if "wgan" in gt else ("dragan" if "dragan" in gt else None)

    def gradient_penalty(self, real, fake, gp):
        """BigGAN.py:717-742 with the random inputs given: gp = {"eps" (dragan), "alpha" [B], "aug" draws}."""
        cfg, vs = self.cfg, self.vs
        kind = self.gradient_penalty_type()
        B = real.shape[0]
        if kind == "dragan":
            x_std = torch.sqrt(((real - real.mean()) ** 2).mean())            # tf.nn.moments over all axes
            fake = real + 0.5 * x_std * self._t(gp["eps"])
        alpha = self._t(gp["alpha"]).reshape(B, 1, 1, 1)
        interpolated = (real + alpha * (fake - real)).detach().requires_grad_(True)
        logit = discriminator(vs, cfg, R.diffaugment(interpolated, gp["aug"], cfg.da_policy))["real"]
        grad, = torch.autograd.grad(logit.sum(), interpolated, create_graph=True)    # tf.gradients(logit, x)[0]
        grad_norm = torch.sqrt((grad.reshape(B, -1) ** 2).sum(dim=1))
        if kind == "wgan-lp":
            return cfg.ld * (torch.clamp(grad_norm - 1.0, min=0.0) ** 2).mean()
        return cfg.ld * ((grad_norm - 1.0) ** 2).mean()

    def d_forward(self, real, z, aug_real, aug_fake, labels=None, cls_z=None, gp=None):
        cfg, vs = self.cfg, self.vs
        vs.reg_losses = []
        vs.state_updates.clear()
        real_aug = R.diffaugment(self._t(real), aug_real, cfg.da_policy)            # BigGAN.py:806
        d_real = discriminator(vs, cfg, real_aug)                                   # :807-808
        cz = self._t(cls_z) if cfg.n_labels else None
        fake = generator(vs, cfg, self._t(z), cz, True)                             # :883
        d_fake = discriminator(vs, cfg, R.diffaugment(fake, aug_fake, cfg.da_policy))   # :857
        d_loss = R.discriminator_loss(cfg.gan_type, d_real["real"], d_fake["real"], cfg.d_flood)  # :879
        gp_val = None
        if self.gradient_penalty_type():                                            # :867-868, 880
            gp_val = self.gradient_penalty(self._t(real), fake.detach(), gp)
            d_loss = d_loss + gp_val
        d_cls = None
        if cfg.n_labels:
            w = torch.ones(cfg.n_labels, dtype=self.dtype)
            d_cls = cfg.d_cls_loss_weight * R.cls_loss_logistic(self._t(labels), d_real["cls"], w)  # :853
            d_loss = d_loss + d_cls
        return {"d_loss": d_loss, "real_logits": d_real["real"], "fake_logits": d_fake["real"],
                "fake": fake, "d_cls_loss": d_cls, "gp": gp_val}

    def g_forward(self, z, aug_fake, cls_z=None, real=None, aug_real=None):
        cfg, vs = self.cfg, self.vs
        vs.reg_losses = []
        vs.state_updates.clear()
        real_logits = None
        if cfg.gan_type.startswith("ra-"):      # relativistic: g_loss reads D(aug(real)) as well (BigGAN.py:806-808,896)
            real_logits = discriminator(vs, cfg, R.diffaugment(self._t(real), aug_real, cfg.da_policy))["real"]
        cz = self._t(cls_z) if cfg.n_labels else None
        fake = generator(vs, cfg, self._t(z), cz, True)
        d_fake = discriminator(vs, cfg, R.diffaugment(fake, aug_fake, cfg.da_policy))
        g_adv = R.generator_loss(cfg.gan_type, d_fake["real"], real_logits, cfg.g_flood)   # BigGAN.py:896
        g_loss = g_adv
        g_cls = None
        if cfg.n_labels:
            w = torch.ones(cfg.n_labels, dtype=self.dtype)
            g_cls = cfg.g_cls_loss_weight * R.cls_loss_logistic(cz, d_fake["cls"], w)   # :894
            g_loss = g_loss + g_cls
        reg = sum(l for _, l in vs.reg_losses) if vs.reg_losses else torch.zeros((), dtype=self.dtype)
        if cfg.g_regularization != "none":
            g_loss = g_loss + reg                                                   # :897-898
        return {"g_loss": g_loss, "g_adv": g_adv, "g_reg": reg, "fake_logits": d_fake["real"],
                "fake": fake, "g_cls_loss": g_cls}

    def sample(self, z, cls_z=None, commit=True):
        """BigGAN.py:963-971: generator(test_z, zero_cls_z, is_training=False) reading the trainables through
        ema_getter (ExponentialMovingAverage shadows); ``u`` still advances (ops.py:743)."""
        cfg, vs = self.cfg, self.vs
        vs.state_updates.clear()
        live = {k: vs.vars[k] for k in self.ema}
        for k, e in self.ema.items():
            vs.vars[k] = e
        try:
            if cfg.n_labels and cls_z is None:
                cls_z = np.zeros((np.asarray(z).shape[0], cfg.n_labels), np.float32)
            cz = self._t(cls_z) if cfg.n_labels else None
            with torch.no_grad():
                img = generator(vs, cfg, self._t(z), cz, False)
        finally:
            for k, v in live.items():
                vs.vars[k] = v
        if commit:
            vs.commit()
        return img

    def d_step(self, real, z, aug_real, aug_fake, labels=None, cls_z=None, apply=True, gp=None):
        out = self.d_forward(real, z, aug_real, aug_fake, labels, cls_z, gp)
        params = self.d_params()
        grads = torch.autograd.grad(out["d_loss"], list(params.values()), allow_unused=True)
        gd = OrderedDict((k, (g if g is not None else torch.zeros_like(p)))
                         for (k, p), g in zip(params.items(), grads))
        if apply:
            self.d_opt.step(params, gd)
            self.vs.commit()           # u, pop stats, moving stats of BOTH nets advance (utils.py:258)
        out["grads"] = gd
        return out

    def g_step(self, z, aug_fake, cls_z=None, apply=True, real=None, aug_real=None):
        out = self.g_forward(z, aug_fake, cls_z, real, aug_real)
        params = self.g_params()
        grads = torch.autograd.grad(out["g_loss"], list(params.values()), allow_unused=True)
        gd = OrderedDict((k, (g if g is not None else torch.zeros_like(p)))
                         for (k, p), g in zip(params.items(), grads))
        if apply:
            self.g_opt.step(params, gd)
            d = self.cfg.moving_decay
            with torch.no_grad():                                  # MovingAverageOptimizer (BigGAN.py:924-926)
                for k, p in params.items():
                    self.ema[k].mul_(d).add_(p, alpha=1 - d)
            self.vs.commit()
        out["grads"] = gd
        return out


# ----------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------
def truncated_normal(rng, shape):
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return a.astype(np.float32)


def synthetic_batch(cfg, seed, B=None):
    """Images U(-1,1), two z draws (D step, G step), three DiffAugment draw sets, optional labels."""
    B = B or cfg.batch_size
    rng = np.random.default_rng(seed)
    S = cfg.img_size
    out = {
        "real": rng.uniform(-1, 1, (B, S, S, cfg.c_dim)).astype(np.float32),
        "z_d": truncated_normal(rng, (B, 1, 1, cfg.z_dim)),
        "z_g": truncated_normal(rng, (B, 1, 1, cfg.z_dim)),
        "aug_real": R.draw_diffaugment(rng, B, S),
        "aug_fake_d": R.draw_diffaugment(rng, B, S),
        "aug_fake_g": R.draw_diffaugment(rng, B, S),
    }
    if "wgan" in cfg.gan_type or "dragan" in cfg.gan_type:      # BigGAN.py:719, 726, 729
        out["gp"] = {"alpha": rng.random(B).astype(np.float32), "aug": R.draw_diffaugment(rng, B, S)}
        if "dragan" in cfg.gan_type:
            out["gp"]["eps"] = rng.random((B, S, S, cfg.c_dim)).astype(np.float32)
    if cfg.n_labels:
        def onehot():
            lab = rng.integers(0, cfg.n_labels, B)
            m = np.zeros((B, cfg.n_labels), np.float32)
            m[np.arange(B), lab] = 1
            return m
        out["labels"] = onehot()
        out["cls_z_d"] = onehot()
        out["cls_z_g"] = onehot()
    return out


def perturb_for_parity(vs, seed=7):
    """Make every zero-initialised trainable (PReLU alpha, SA gamma, biases) non-zero so that the
    attention path and all gradients are exercised (synthetic inputs)."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for k, v in vs.vars.items():
            leaf = k.rsplit("/", 1)[-1]
            if leaf == "alpha":
                v.copy_(torch.tensor(rng.uniform(0.05, 0.3, tuple(v.shape)), dtype=v.dtype))
            elif leaf == "gamma" and k.endswith("self_attention/gamma"):
                v.copy_(torch.tensor(rng.uniform(0.3, 0.8, tuple(v.shape)), dtype=v.dtype))
            elif leaf == "bias":
                v.copy_(torch.tensor(rng.normal(0, 0.05, tuple(v.shape)), dtype=v.dtype))
            elif leaf == "kernel":
                # larger weights so logits / gradients are not vanishingly small
                v.mul_(4.0)
            elif "/batch_renorm/" in k and leaf in ("renorm_mean", "pop_mean"):
                # batch renorm: at its initial state (renorm_weight 0, mean 0, var 1) the corrections are r = 1,
                # d = 0; start from a mid-training state so clipped and unclipped corrections both occur
                v.copy_(torch.tensor(rng.normal(0, 0.3, tuple(v.shape)), dtype=v.dtype))
            elif "/batch_renorm/" in k and leaf in ("renorm_var", "pop_var"):
                v.copy_(torch.tensor(rng.uniform(0.05, 2.5, tuple(v.shape)), dtype=v.dtype))
            elif leaf == "renorm_stddev":
                v.copy_(torch.tensor(rng.uniform(0.2, 1.6, tuple(v.shape)), dtype=v.dtype))
            elif leaf == "renorm_weight":
                v.fill_(0.7)
