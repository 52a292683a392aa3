"""``BigGAN`` with the reference's ``generator`` / ``discriminator`` topology and train ops
(``/root/reference/BigGAN.py:246-715, 768-961``; option plumbing ``GANBase.py:13-55``) executing
eagerly on one MI355X per process.  One training iteration = D step then G step
(BigGAN.py:1061-1084); each step is one "run" with its own z, DiffAugment draws and one spectral-norm
power iteration per weight.

Out-of-scope reference features (alternative heads, reconstruction heads, label
embeddings, mixed-kernel blocks) are accepted as flags and rejected here with
NotImplementedError.
"""
import copy
import os
import math
import time

import torch

from . import ops
from . import functional as Fn
from . import scope as S
from . import hip
from .ops import (fully_connected, resblock_up_condition, resblock_down, resblock, self_attention_2, conv, bn,
                  resblock_up_cond_deep, resblock_down_deep, upconv, g_conv, cond_bn,
                  prelu, relu, lrelu, tanh, global_sum_pooling, discriminator_loss, generator_loss)
from .DiffAugment import DiffAugment, draw as draw_augment
from .utils import orthogonal_regularizer, orthogonal_regularizer_fc, l2_regularizer, round_up, cls_loss_fn


class GANBase(object):
    """GANBase.py:13-55 (attribute plumbing, bn/conv option dicts, da_policy expansion)."""

    def __init__(self, args):
        self.dataset_name = args.dataset
        self.checkpoint_dir = args.checkpoint_dir
        self.sample_dir = args.sample_dir
        self.result_dir = args.result_dir
        self.log_dir = args.log_dir
        self.epoch = args.epoch
        self.iterations_per_epoch = args.iteration
        self.batch_size = args.batch_size
        self.virtual_batches = args.virtual_batches
        self.print_freq = args.print_freq
        self.save_freq = args.save_freq
        self.img_size = args.img_size
        self.bn_options = {"type": args.bn_type, "momentum": args.bn_momentum}       # GANBase.py:39-47
        if self.bn_options["type"] == 'batch_renorm':
            self.bn_options["renorm_clipping"] = {"rmax": args.bn_renorm_rmax, "dmax": args.bn_renorm_dmax}
            self.bn_options["renorm_momentum"] = args.bn_renorm_momentum
            self.bn_options["shared_renorm"] = args.bn_renorm_shared
        self.conv_options = {"padding_type": args.conv_padding, "sn": args.sn}       # GANBase.py:49-51
        self.da_policy = args.da_policy
        if self.da_policy == 'full':
            self.da_policy = 'color,translation,cutout'                               # GANBase.py:53-55


class BigGAN(GANBase):
    def __init__(self, args, device="cuda", store=None, process_group=None, seed=42):
        GANBase.__init__(self, args)
        self.model_name = "BigGAN"
        self.args = args
        self.device = torch.device(device)
        self.depth = args.img_size.bit_length() - 2                                   # BigGAN.py:19

        unsupported = [
            ("cls_embedding", args.cls_embedding),
            ("shared_z", args.shared_z > 0), ("g_z_dense_concat", args.g_z_dense_concat),
            ("g_mixed_resblocks", args.g_mixed_resblocks),
            ("g_final_layer", args.g_final_layer), ("multi_head", args.multi_head),
            ("z_reconstruct", args.z_reconstruct), ("d_reconstruction", args.d_reconstruction),
            ("d_reconstruction_halfres", args.d_reconstruction_halfres),
            ("d_reconstruction_texture", args.d_reconstruction_texture), ("d_final_conv", args.d_final_conv),
            ("c_dim!=3", args.c_dim != 3),
        ]
        bad = [n for n, v in unsupported if v]
        if bad:
            raise NotImplementedError("flags outside the MI355X hot path (SURVEY.md section 8): " + ", ".join(bad))
        self.gan_type = args.gan_type
        self.d_loss_func = args.d_loss_func if args.d_loss_func else self.gan_type     # BigGAN.py:127-128
        if self.gan_type not in Fn.GAN_LOSS_KINDS or self.d_loss_func not in Fn.GAN_LOSS_KINDS:
            raise ValueError("--gan_type / --d_loss_func %s / %s: available: %s"
                             % (self.gan_type, self.d_loss_func, ", ".join(sorted(Fn.GAN_LOSS_KINDS))))
        self.relativistic = self.gan_type.startswith('ra-')
        if 'wgan' in self.d_loss_func:                                                 # BigGAN.py:130-138
            self.gradient_penalty_type = self.gan_type
        elif 'dragan' in self.d_loss_func:
            self.gradient_penalty_type = 'dragan'
        else:
            self.gradient_penalty_type = None
        self.use_gradient_penalty = bool(self.gradient_penalty_type)
        if self.use_gradient_penalty and self.gradient_penalty_type not in ('wgan-gp', 'wgan-lp', 'dragan'):
            raise ValueError("gradient penalty type %r (BigGAN.py:736-740 knows wgan-gp, wgan-lp, dragan)"
                             % self.gradient_penalty_type)
        self.ld = args.ld

        self.activation = args.activation                                              # BigGAN.py:71-83
        if self.activation == 'relu':
            self.activation_fn = relu
        elif self.activation == 'prelu':
            self.activation_fn = prelu
        elif self.activation == 'lrelu':
            def lrelu_p(alpha):
                def lrelu_a(x):
                    return lrelu(x, alpha)
                return lrelu_a
            self.activation_fn = lrelu_p(0.2)                                          # BigGAN.py:76-81
        else:
            raise ValueError("Unknown activation function: " + str(self.activation))

        self.ch = args.ch
        self.d_ch = args.d_ch if args.d_ch > 0 else args.ch                            # BigGAN.py:153-154
        self.upsampling_method = args.upsampling_method
        self.downsampling_method = args.downsampling_method
        self.g_conv = args.g_conv
        self.g_grow_factor = args.g_grow_factor
        self.d_grow_factor = args.d_grow_factor
        self.g_regularization_method = args.g_regularization
        self.g_regularization_factor = args.g_regularization_factor
        self.g_sa_size = args.g_sa_size if args.g_sa_size != 0 else args.sa_size       # BigGAN.py:110-117
        self.d_sa_size = args.d_sa_size if args.d_sa_size != 0 else args.sa_size
        self.z_dim = args.z_dim
        self.first_split_ratio = args.first_split_ratio
        if self.z_dim % self.depth != 0 and self.first_split_ratio == 1:               # BigGAN.py:122-124
            self.z_dim = self.z_dim + self.depth - self.z_dim % self.depth
        self.g_flood, self.d_flood = args.g_flood, args.d_flood
        self.n_critic = args.n_critic
        self.sn = args.sn
        self.d_use_bias = args.bias_in_d
        self.bias_in_sa = args.bias_in_sa
        self.bn_in_d = args.bn_in_d
        self.c_dim = args.c_dim
        self.g_rgb_mix_kernel = args.g_rgb_mix_kernel
        self.z_trunc_train = args.z_trunc_train
        self.g_learning_rate, self.d_learning_rate = args.g_lr, args.d_lr
        self.beta1, self.beta2 = args.beta1, args.beta2
        self.moving_decay = args.moving_decay
        self.d_compat_use_sn_in_critic_output = args.d_compat_use_sn_in_critic_output
        self.extension_32 = getattr(args, "extension_32", False)
        self.deep = args.deep                                                          # BigGAN.py:20
        self.n_labels = args.n_labels                                                  # BigGAN.py:22-23
        self.acgan = self.n_labels > 0
        self.virtual_batches = max(int(args.virtual_batches), 1)
        if self.acgan:                                                                 # BigGAN.py:91-100
            self.cls_loss_type = args.cls_loss_type
            self.d_cls_loss_weight = args.d_cls_loss_weight
            self.g_cls_loss_weight = args.g_cls_loss_weight
            self.d_compat_use_sn_in_classification = args.d_compat_use_sn_in_classification
            if args.cls_loss_weights != '':
                self.cls_loss_weights = list(map(float, open(args.cls_loss_weights).read().split()))
                if len(self.cls_loss_weights) != self.n_labels:
                    raise ValueError('Number of class loss weights does not match n_labels (' +
                                     str(len(self.cls_loss_weights)) + " vs. " + str(self.n_labels) + ")")
            else:
                self.cls_loss_weights = [1.0] * self.n_labels

        # arithmetic / storage policy (extension flag --precision): must be set before build_model() sizes the
        # packed-weight buffers of the spectral-norm batches
        self.precision = getattr(args, "precision", None) or "fp32"
        Fn.set_precision(self.precision)
        self.store = store if store is not None else S.VariableStore(self.device, seed)
        S.set_default_store(self.store)
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        self.gen = torch.Generator(device=self.device) if self.device.type == "cuda" else None
        if self.gen is not None:
            self.gen.manual_seed(1234 + self.rank)
        self.built = False

    ##################################################################################
    # Generator  (BigGAN.py:236-582)
    ##################################################################################
    def round_up(self, val, multiple):
        return (int(val) + multiple - 1) // multiple * multiple

    def scale_channels(self, ch, factor):
        return self.round_up(int(ch * factor), 8)

    def g_channels_for_block(self, b_i, b_count):
        return self.scale_channels(self.ch, self.g_grow_factor ** (b_count - b_i - 1))

    def g_block_info(self):                                                            # BigGAN.py:290-294
        s = self.img_size
        if s == 64: info = {"counts": [1, 1, 1, 1], "sa_index": 3}
        elif s == 128: info = {"counts": [1, 1, 1, 1, 1], "sa_index": 4}
        elif s == 256: info = {"counts": [1, 2, 1, 1, 1], "sa_index": 3}
        elif s == 512: info = {"counts": [1, 2, 1, 1, 2], "sa_index": 3}
        elif s == 32 and self.extension_32: info = {"counts": [1, 1, 1], "sa_index": 2}
        else: raise ValueError("Invalid image size specified: " + str(self.img_size))
        if self.g_sa_size != 0:
            self.set_sa_index(info, self.g_sa_size)
        return info

    def d_block_info(self):                                                            # BigGAN.py:607-611
        s = self.img_size
        if s == 64: info = {"counts": [1, 1, 1, 1], "sa_index": 1}
        elif s == 128: info = {"counts": [1, 1, 1, 1, 1], "sa_index": 1}
        elif s == 256: info = {"counts": [1, 1, 1, 2, 1], "sa_index": 2}
        elif s == 512: info = {"counts": [1, 2, 1, 1, 2], "sa_index": 2}
        elif s == 32 and self.extension_32: info = {"counts": [1, 1, 1], "sa_index": 1}
        else: raise ValueError("Invalid image size specified: " + str(self.img_size))
        if self.d_sa_size != 0:
            self.set_sa_index(info, self.d_sa_size, scaling_down=True)
        return info

    def set_sa_index(self, block_info, sa_size, scaling_down=False):
        """BigGAN.py:1457-1480."""
        if sa_size < 0:
            block_info["sa_index"] = -1
            return
        cur_f_size = self.img_size if scaling_down else 4
        for i, bs in enumerate(block_info["counts"]):
            for j in range(bs):
                if scaling_down:
                    cur_f_size //= 2
                else:
                    cur_f_size *= 2
            if scaling_down:
                met_goal = cur_f_size <= sa_size
            else:
                met_goal = cur_f_size >= sa_size
            if met_goal or i == len(block_info["counts"]) - 1:
                block_info["sa_index"] = i + 1
                break
        if cur_f_size != sa_size:
            print("Warning: moving self-attention to " + str(cur_f_size) + "x" + str(cur_f_size) + " feature maps")

    def z_split_sizes(self):                                                           # BigGAN.py:280-288
        if self.first_split_ratio > 1:
            split_dim = self.z_dim // (self.depth - 1 + self.first_split_ratio)
            first_split_dim = self.z_dim - (self.depth - 1) * split_dim
        else:
            split_dim = self.z_dim // self.depth
            first_split_dim = split_dim
        return [first_split_dim] + ([split_dim] * (self.depth - 1))

    def generator(self, z, cls_z=None, is_training=True, reuse=False, custom_getter=None, simple_head=False):
        opt = {"sn": self.sn, "is_training": is_training, "upsampling_method": self.upsampling_method,
               "g_conv": self.g_conv, "act": self.activation_fn, "self_attention_bias": self.bias_in_sa,
               "bn": copy.deepcopy(self.bn_options), "conv": copy.deepcopy(self.conv_options)}
        if is_training:                                                                # BigGAN.py:257-274
            m = self.g_regularization_method
            if m == 'none':
                opt["conv"]["regularizer"] = None
                opt["fc_regularizer"] = None
            elif m in ('ortho', 'ortho_cosine'):
                opt["conv"]["regularizer"] = orthogonal_regularizer(self.g_regularization_factor, type=m)
                opt["fc_regularizer"] = orthogonal_regularizer_fc(self.g_regularization_factor, type=m)
            elif m == 'l2':
                opt["conv"]["regularizer"] = l2_regularizer(self.g_regularization_factor)
                opt["fc_regularizer"] = l2_regularizer(self.g_regularization_factor)
            else:
                raise ValueError("Unknown regularization method: " + str(m))
        else:
            opt["conv"]["regularizer"] = None
            opt["fc_regularizer"] = None

        self._sn_prefetch("generator", z)
        with S.variable_scope("generator", reuse=reuse):
            block_info = self.g_block_info()
            split_sizes = self.z_split_sizes()
            z2 = z.reshape(z.shape[0], -1)
            z_split = list(torch.split(z2, split_sizes, dim=-1))                       # BigGAN.py:335 (views)
            clsz_size = 0
            if self.acgan:                                                             # BigGAN.py:346-365
                clsz_size = self.n_labels
                if z.device.type == "meta":
                    z_split = [torch.empty(zz.shape[0], zz.shape[1] + clsz_size, device="meta") for zz in z_split]
                else:
                    cz = cls_z.reshape(-1, clsz_size)
                    z_split = [torch.cat([zz, cz], dim=-1) for zz in z_split]
                split_sizes = [sz + clsz_size for sz in split_sizes]
            next_zi = [0]

            def next_z_split():
                zi = next_zi[0]
                next_zi[0] += 1
                return z_split[zi], split_sizes[zi]

            counts = block_info["counts"]
            ch_mul = 2 ** (len(counts) - 1)                                            # BigGAN.py:427
            ch = self.g_channels_for_block(0, len(counts))

            layer_z, z_dim = next_z_split()
            f_width = self.round_up(z_dim * 1.85, 8)                                   # BigGAN.py:433 (z_dim includes the labels)
            if not self.args.g_first_level_dense_layer:                                # BigGAN.py:444
                x = fully_connected(layer_z, units=4 * 4 * ch, scope='dense', opt=opt)
            elif self.activation_fn is relu:                                           # BigGAN.py:434-438
                x = fully_connected(layer_z, units=f_width, scope='dense1', opt=opt)
                x = relu(x)
                x = fully_connected(x, units=4 * 4 * ch, scope='dense2', opt=opt)
            else:
                with S.variable_scope('first'):                                        # BigGAN.py:440-443
                    x = fully_connected(layer_z, units=f_width, scope='dense1', opt=opt)
                    x = opt["act"](x)
                    x = fully_connected(x, units=4 * 4 * ch, scope='dense2', opt=opt)
            x = ops._resident_out(x.reshape(-1, 4, 4, ch))                             # BigGAN.py:446 (bf16 trunk from here)

            b_i = 0
            for block_count in counts:                                                 # BigGAN.py:449-489
                x = self._grad_mark(x, 'resblock_up_' + str(ch_mul))
                scope = 'resblock_up_' + str(ch_mul)
                for sb_i in range(block_count):
                    layer_z, z_dim = next_z_split()
                    if block_count > 1:
                        scope = scope + '_' + str(sb_i)                                # cumulative (BigGAN.py:455)
                    if self.args.g_other_level_dense_layer:                            # BigGAN.py:457-462
                        with S.variable_scope('z' + str(ch_mul)):
                            zw = self.round_up(z_dim * 1.25, 8)                        # (z_dim includes the labels)
                            layer_z = fully_connected(layer_z, units=zw, scope='dense1', opt=opt)
                            layer_z = opt["act"](layer_z)
                    is_last_block = sb_i == block_count - 1 and b_i == len(counts) - 1
                    if self.args.g_no_last_resblock and is_last_block:                 # BigGAN.py:468-473
                        with S.variable_scope(scope):
                            x = upconv(x, ch, use_bias=False, opt=opt)
                            x = cond_bn(x, layer_z, opt=opt)
                            x = opt["act"](x)
                            x = g_conv(x, ch, use_bias=False, opt=opt)
                    elif self.deep:                                                    # BigGAN.py:475-477
                        x = resblock_up_cond_deep(x, layer_z, channels_out=ch, use_bias=True, opt=opt, scope=scope)
                        x = resblock_up_cond_deep(x, layer_z, channels_out=ch, upscale=False, use_bias=True, opt=opt,
                                                  scope=scope + "_2")
                    else:
                        x = resblock_up_condition(x, layer_z, channels=ch, use_bias=False, opt=opt, scope=scope)
                b_i += 1
                if b_i == block_info["sa_index"]:
                    x = self_attention_2(x, channels=ch, opt=opt, scope='self_attention')
                ch = self.g_channels_for_block(b_i, len(counts))
                ch_mul = ch_mul // 2

            x = self._grad_mark(x, 'tail')
            x = ops._bn_act(x, None, opt, _out_fp32=os.environ.get("BG_IMAGE_LAYERS", "") == "fp32")   # BigGAN.py:491-492
            x = conv(x, channels=self.c_dim, kernel=self.g_rgb_mix_kernel, stride=1, pad=1, use_bias=False, opt=opt,
                     scope='G_logit')                                                  # BigGAN.py:570
            x = tanh(x)                                                                # BigGAN.py:580
            return x

    ##################################################################################
    # Discriminator  (BigGAN.py:588-715)
    ##################################################################################
    def d_channels_for_block(self, b_i):
        return self.scale_channels(self.d_ch, self.d_grow_factor ** b_i)

    def make_opt_with_sn(self, src_opt, sn):
        """BigGAN.py:1482-1487."""
        sn_opt = copy.copy(src_opt)
        sn_opt["conv"] = copy.copy(src_opt["conv"])
        sn_opt["sn"] = sn
        sn_opt["conv"]["sn"] = sn
        return sn_opt

    def discriminator(self, x, is_training=True, reuse=False):
        opt = {"sn": self.sn, "is_training": is_training, "bn_in_d": self.bn_in_d, "act": self.activation_fn,
               "downsampling_method": self.downsampling_method, "self_attention_bias": self.bias_in_sa,
               "bn": copy.deepcopy(self.bn_options), "conv": copy.deepcopy(self.conv_options)}
        outputs = {}
        self._sn_prefetch("discriminator", x)
        with S.variable_scope("discriminator", reuse=reuse):
            ch = self.d_channels_for_block(0)
            block_info = self.d_block_info()
            b_i = 0
            ch_mul = 1
            for block_count in block_info["counts"]:                                   # BigGAN.py:624-664
                scope = 'resblock_down_' + str(ch_mul)
                for sb_i in range(block_count):
                    if block_count > 1:
                        scope = scope + '_' + str(sb_i)
                    if self.deep:                                                      # BigGAN.py:629-631
                        x = resblock_down_deep(x, channels_out=ch, use_bias=self.d_use_bias, opt=opt, scope=scope)
                        x = resblock_down_deep(x, channels_out=ch, downscale=False, use_bias=self.d_use_bias, opt=opt,
                                               scope=scope + "_2")
                    else:
                        x = resblock_down(x, channels=ch, use_bias=self.d_use_bias, opt=opt, scope=scope)
                b_i += 1
                if b_i == block_info["sa_index"]:
                    x = self_attention_2(x, channels=ch, opt=opt, scope='self_attention')
                ch = self.d_channels_for_block(b_i)
                ch_mul = ch_mul * 2
            ch = self.d_channels_for_block(b_i - 1)                                    # BigGAN.py:666
            x = resblock(x, channels=ch, use_bias=self.d_use_bias, opt=opt, scope='resblock')
            x = opt["act"](x)
            features = global_sum_pooling(x)                                           # BigGAN.py:671
            critic_opt = self.make_opt_with_sn(opt, self.d_compat_use_sn_in_critic_output)
            x = fully_connected(features, units=1, opt=critic_opt, scope='D_logit')    # BigGAN.py:681-682
            outputs["real"] = x
            if self.acgan:                                                             # BigGAN.py:686-701
                cls_opt = self.make_opt_with_sn(opt, self.d_compat_use_sn_in_classification)
                if self.args.d_cls_dense_layers:                                       # BigGAN.py:690-698
                    with S.variable_scope("classification"):
                        u1 = self.round_up(ch / 16.0 + self.n_labels * 1.25, 8)
                        y = fully_connected(features, units=u1, opt=cls_opt, scope='dense1')
                        y = opt["act"](y)
                        u2 = self.round_up(u1 / 4.0 + self.n_labels * 1.1, 4)
                        y = fully_connected(y, units=u2, opt=cls_opt, scope='dense2')
                        y = opt["act"](y)
                        y = fully_connected(y, units=self.n_labels, opt=cls_opt, scope='DC_logit')
                    outputs["cls"] = y
                else:
                    outputs["cls"] = fully_connected(features, units=self.n_labels, opt=cls_opt, scope='DC_logit')
            return outputs

    ##################################################################################
    # Model  (BigGAN.py:768-961)
    ##################################################################################
    def build_model(self):
        """Create every variable (one shape-only pass), pack trainables into flat arenas, set up the
        optimiser state (BigGAN.py:913-930: Adam for D; MovingAverageOptimizer(Adam) for G)."""
        B = 2
        z = torch.empty(B, 1, 1, self.z_dim, device="meta")
        img = self.generator(z, None, is_training=True)
        assert tuple(img.shape) == (B, self.img_size, self.img_size, self.c_dim), img.shape
        out = self.discriminator(img)
        assert tuple(out["real"].shape) == (B, 1)
        self.store.pack()
        self.g_arena = self.store.arenas["generator"]
        self.d_arena = self.store.arenas["discriminator"]
        self.d_vars = self.store.trainable_variables('discriminator')                  # BigGAN.py:915-917
        self.g_vars = self.store.trainable_variables('generator')
        self.sn_batches = {}
        if self.store.device.type == "cuda":
            for group in ("generator", "discriminator"):
                pairs = [(self.store.vars[w], self.store.vars[u]) for w, u in self.store.sn_pairs.items()
                         if w.startswith(group + "/") and w in self.store.arenas[group].offsets]
                names = [w for w, u in self.store.sn_pairs.items()
                         if w.startswith(group + "/") and w in self.store.arenas[group].offsets]
                for i in range(0, len(pairs), 256):
                    chunk = names[i:i + 256]
                    # f | g | h projections of every self-attention block share one packed GEMM operand
                    groups = []
                    for j, nm in enumerate(chunk):
                        if nm.endswith("/f_conv/kernel"):
                            pre = nm[:-len("f_conv/kernel")]
                            try:
                                groups.append([j, chunk.index(pre + "g_conv/kernel"), chunk.index(pre + "h_conv/kernel")])
                            except ValueError:
                                pass
                    # data parallelism: the power iteration is sharded by weight (SnBatch docstring; BG_SHARD_SN=0: replicated)
                    shard = None
                    if (self.world > 1 or os.environ.get("BG_SHARD_OPT", "") == "force") \
                            and os.environ.get("BG_SHARD_SN", "1") != "0" \
                            and torch.distributed.is_available() and torch.distributed.is_initialized():
                        shard = (self.rank, self.world, self.pg)
                    sb = Fn.SnBatch(pairs[i:i + 256], groups, shard)
                    if shard is not None:          # the u vectors now live in the batch's flat state buffer
                        for j, nm in enumerate(chunk):
                            self.store.vars[self.store.sn_pairs[nm]] = sb.u[j]
                    self.sn_batches.setdefault(group, []).append(sb)
        self.reg_owner = self._shard_regularisers()
        self._setup_exchange()
        self.counter = 0
        self.built = True
        return self

    def _setup_exchange(self):
        """Data parallelism: the process group of the large gradient / parameter exchanges (its own RCCL communicator
        and stream, parallel.new_gradient_group) and the STATIC partition of each arena into exchange ranges
        (parallel.ShardedRanges) - the generator's follow its stages, so that a stage's range can leave while backward
        is still running in the earlier stages; the discriminator's are 256 MiB chunks.  With --shard_optimizer (default
        under data parallelism; BG_SHARD_OPT=0 switches back to all-reduce + replicated update) every range is
        reduce-scattered, TF-Adam + EMA run on the owned 1/N and the parameters are all-gathered: Adam moments and EMA
        shadows then exist only on their owner (``sync_sharded_state`` assembles them for checkpoints and sampling)."""
        from .parallel import ShardedRanges, new_gradient_group
        self.pg_grad = None
        self._param_gathers = {"generator": [], "discriminator": []}
        self.shards = {}
        self.shard_opt = False
        force = os.environ.get("BG_SHARD_OPT", "") == "force"      # (tests: the sharded call sites on a 1-rank RCCL group)
        if (self.world == 1 and not force) or self.store.device.type != "cuda":
            return
        self.pg_grad = new_gradient_group(self.pg)
        self.shards["generator"] = ShardedRanges(self.g_arena.size, self._g_bucket_starts().values(), self.world, self.rank)
        self.shards["discriminator"] = ShardedRanges(self.d_arena.size, (), self.world, self.rank)
        want = os.environ.get("BG_SHARD_OPT", "1") != "0"
        self.shard_opt = want and (force or all(sh.sharded for sh in self.shards.values()))

    def _shard_regularisers(self):
        """The ortho-cosine terms depend on the weights only, so under data parallelism every rank would
        repeat the same 2 x 229 GFLOP (config 2) of Gram work.  Instead each kernel's term is evaluated
        on ONE rank (longest-processing-time assignment by GEMM cost) and the SUM all-reduce of the
        gradient arena delivers it everywhere."""
        if self.world == 1:
            return None
        cost = {}
        for name, shape in self.store.reg_shapes.items():
            c = shape[-1]
            rows = 1
            for d in shape[:-1]:
                rows *= d
            cost[name] = rows * rows * c if 2 * rows <= c else rows * c * c
        load = [0] * self.world
        owner = {}
        for name in sorted(cost, key=lambda k: (-cost[k], k)):
            r = min(range(self.world), key=lambda i: (load[i], i))
            owner[name] = r
            load[r] += cost[name]
        return owner

    def _begin_run(self):
        S.set_default_store(self.store)       # several models may live in one process: ops resolve variables here
        Fn.set_precision(self.precision)
        if getattr(self, "_zero_pool", None) is None and self.device.type == "cuda":
            self._zero_pool = Fn.ZeroPool(self.device)
        ops.begin_run(self._reduce_fn(), self.world, self.rank, getattr(self, "reg_owner", None),
                      getattr(self, "_zero_pool", None))

    def _sn_prefetch(self, group, x):
        if x.is_cuda:
            self._wait_params(group)          # (a sharded update's all-gather of this network may still be in flight)
            for b in getattr(self, "sn_batches", {}).get(group, ()):
                ops.sn_prefetch(b)

    def _sn_backward(self, group):
        for b in getattr(self, "sn_batches", {}).get(group, ()):
            b.backward()

    # ---- bucketed, overlapped exchange of the generator gradients (data parallel) -----------------------
    def _g_bucket_starts(self):
        """Arena offset at which the variables of each generator stage begin (stages in creation = arena order:
        first/*, resblock_up_<m> ..., [self_attention], ..., tail = batch_norm / prelu / G_logit)."""
        if getattr(self, "_g_starts", None) is None:
            arena = self.g_arena
            starts = {}
            for name in arena.names:
                off = arena.offsets[name][0]
                rest = name[len("generator/"):]
                top = rest.split("/")[0]
                if top.startswith("resblock_up_"):
                    key = "resblock_up_" + top[len("resblock_up_"):].split("_")[0]    # sub-blocks share their stage
                elif top in ("batch_norm", "prelu", "G_logit"):
                    key = "tail"
                else:
                    continue
                starts[key] = min(starts.get(key, off), off)
            self._g_starts = starts
        return self._g_starts

    def _grad_mark(self, x, tag):
        st = getattr(self, "_g_overlap", None)
        if st is None or not torch.is_grad_enabled() or not getattr(x, "requires_grad", False) or x.device.type != "cuda":
            return x
        y = Fn.GradMarkFn.apply(x, self._g_bucket_done, tag)
        sums = getattr(x, "bg_bn_sums", None)
        if sums is not None:
            y.bg_bn_sums = sums            # (fused batch-norm statistics of the block input travel with the marker's view)
        return y

    def _g_bucket_done(self, tag):
        """Backward has passed the input of stage ``tag``: every generator variable from that stage's first slot to the
        last not-yet-exchanged slot has its final gradient.  Turn dL/d(w/sigma) into dL/dw for the newly finished
        weights, zero what nothing wrote and start the asynchronous SUM all-reduce of that arena range."""
        st = self._g_overlap
        lo = self._g_bucket_starts().get(tag)
        if st is None or lo is None or lo >= st["hi"]:
            return
        self._g_exchange_range(lo, st["hi"])
        st["hi"] = lo

    def _g_exchange_range(self, lo, hi):
        st = self._g_overlap
        self._sn_backward("generator")                      # (only the weights touched since the last call)
        self.store.zero_untouched("generator", lo, hi)
        st["handles"].extend(self._exchange_begin("generator", lo, hi, st["apply"]))

    # ---- data-parallel hooks -----------------------------------------------------------------
    def _reduce_fn(self):
        if self.world == 1:
            return None
        pg = self.pg

        def red(t):
            torch.distributed.all_reduce(t, group=pg)
        return red

    def _exchange_begin(self, group, lo, hi, apply, async_op=True):
        """Start the exchange of the gradient ranges inside [lo, hi) of an arena on the gradient process group:
        reduce-scatter when the update that follows is sharded (each rank then holds the summed gradient of the part it
        owns), all-reduce otherwise (``apply`` False: callers that want the whole summed gradient; or sharding is off).
        Returns handles for ``_exchange_end``."""
        from .parallel import reduce_scatter_range
        arena = self.store.arenas[group]
        sh = self.shards[group]
        pg = self.pg_grad if self.pg_grad is not None else self.pg
        out = []
        for a, b in sh.containing(lo, hi):
            if self.shard_opt and apply:
                w = reduce_scatter_range(arena.grads, a, b, sh, pg, async_op=async_op)
                out.append((a, b, w, True))
            else:
                w = torch.distributed.all_reduce(arena.grads.narrow(0, a, b - a), group=pg, async_op=async_op)
                out.append((a, b, w, False))
        return out

    def _exchange_end(self, group, handles, lr, with_ema, apply, grad_scale=1.0):
        """Wait for each exchanged range in issue order and update it: TF-Adam (+ EMA) on the owned part followed by the
        asynchronous all-gather of the new parameters (sharded), or on the whole range (replicated).  The all-gathers
        are waited for by the first reader of that network's parameters (``_wait_params``)."""
        from .parallel import all_gather_range
        arena = self.store.arenas[group]
        sh = self.shards[group]
        pg = self.pg_grad if self.pg_grad is not None else self.pg
        for a, b, work, scattered in handles:
            if work is not None:
                work.wait()
            if not apply:
                continue
            if scattered:
                oa, ob = sh.owned(a, b)
                self._adam(arena, lr, with_ema=with_ema, grad_scale=grad_scale, lo=oa, hi=ob, prepare=False)
                self._param_gathers[group].append(all_gather_range(arena.params, a, b, sh, pg, async_op=True))
            else:
                self._adam(arena, lr, with_ema=with_ema, grad_scale=grad_scale, lo=a, hi=b, prepare=False)

    def _wait_params(self, group=None):
        """Block the compute stream until the parameters of a network are whole again after a sharded update."""
        for g in ((group,) if group else tuple(getattr(self, "_param_gathers", {}))):
            works = self._param_gathers.get(g, ())
            for w in works:
                w.wait()
            if works:
                del works[:]

    def sync_sharded_state(self):
        """Assemble the optimiser state that a sharded update keeps only on its owner - Adam m / v and the generator's EMA
        shadows - on every rank (all-gather per exchange range).  Collective: every rank calls it (checkpoints, sampling
        with the EMA weights, state comparisons in tests)."""
        self._wait_params()
        if not getattr(self, "shard_opt", False):
            return
        from .parallel import all_gather_range
        for group, sh in self.shards.items():
            arena = self.store.arenas[group]
            for buf in (arena.m, arena.v, arena.ema):
                if buf is None:
                    continue
                for a, b in sh.ranges:
                    all_gather_range(buf, a, b, sh, self.pg_grad)

    def _adam_prepare(self, arena, lr):
        """Host part of an optimiser step: advance the step count and put the bias-corrected step size
        lr * sqrt(1 - beta2^t) / (1 - beta1^t) (tf.train.AdamOptimizer) into the arena's device scalar."""
        arena.step += 1
        t = arena.step
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        if getattr(arena, "lr_dev", None) is None:
            arena.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        arena.lr_dev.fill_(lr_t)

    def _adam(self, arena, lr, with_ema, grad_scale=1.0, lo=0, hi=None, prepare=True):
        """TF-Adam (+ EMA) on the arena, or on its slice [lo, hi) (the optimiser is elementwise)."""
        if prepare and not getattr(self, "_capturing", False):
            self._adam_prepare(arena, lr)          # (a captured graph is replayed after _adam_prepare on the host)
        hi = arena.size if hi is None else hi
        sl = (lambda t: t.narrow(0, lo, hi - lo)) if (lo, hi) != (0, arena.size) else (lambda t: t)
        hip.check(hip.lib().bg_adam_tf_ema_step_dev(
            hip.f32(sl(arena.params)), hip.f32(sl(arena.grads)), hip.f32(sl(arena.m)), hip.f32(sl(arena.v)),
            hip.f32(sl(arena.ema)) if with_ema else None, hip.f32(arena.lr_dev), self.beta1, self.beta2, 1e-8,
            self.moving_decay, float(grad_scale), hi - lo, hip.stream()))

    def sample_z(self, B):
        z = torch.empty(B, 1, 1, self.z_dim, dtype=torch.float32, device=self.device)
        if self.z_trunc_train:                                                         # BigGAN.py:790-793
            torch.nn.init.trunc_normal_(z, 0.0, 1.0, -2.0, 2.0, generator=self.gen)
        else:
            z.normal_(generator=self.gen)
        return z

    def _set_requires_grad(self, variables, flag):
        for v in variables.values():
            v.requires_grad_(flag)

    # ---- the two train ops -------------------------------------------------------------------------
    def _cls_loss(self):
        if getattr(self, "_cls_loss_fn", None) is None:
            w = torch.tensor(self.cls_loss_weights, dtype=torch.float32, device=self.device)
            self._cls_loss_fn = cls_loss_fn(self.cls_loss_type, w)                     # BigGAN.py:851-852
        return self._cls_loss_fn

    def gp_draws(self, B):
        """The random inputs of one gradient_penalty() call (BigGAN.py:719, 726, 729): eps ~ U[0,1) with the
        batch's shape (dragan only), alpha ~ U[0,1) per sample, and one set of DiffAugment draws."""
        from .DiffAugment import draw
        S_ = self.img_size
        out = {"alpha": torch.rand(B, dtype=torch.float32, device=self.device, generator=self.gen),
               "aug": draw(B, S_, self.device, self.gen) if self.da_policy else None}
        if self.gradient_penalty_type == 'dragan':
            out["eps"] = torch.rand(B, S_, S_, self.c_dim, dtype=torch.float32, device=self.device, generator=self.gen)
        return out

    def gradient_penalty(self, real, fake, draws=None):
        """BigGAN.py:717-742.  GP = ld * mean phi(||d D(aug(x^))/d x^||) on x^ between the real batch and the fake
        batch (wgan-gp / wgan-lp) or a perturbed real batch (dragan).  Three discriminator passes on x^:
        (1) forward + an inputs-only backward give g = dD/dx^, its norms, the penalty's value and the constant
        direction v = ld * phi'(||g||)/count * g/||g||;  (2) a forward-mode pass carries (x^, v) through DiffAugment
        and D as a ``Dual`` and yields the directional derivatives fdot = <g(theta), v>;  (3) the ordinary backward
        of d_loss differentiates fdot w.r.t. the parameters, which IS d GP/d theta (v held constant)."""
        from .ops import Dual
        L = Fn.lib()
        B = real.shape[0]
        per = real.numel() // B
        if draws is None:
            draws = self.gp_draws(B)
        real = real.contiguous()
        xhat = torch.empty_like(real)
        if self.gradient_penalty_type == 'dragan':
            sums = torch.zeros(2, dtype=torch.float64, device=self.device)
            Fn.check(L.bg_bn_stats(Fn.f32(real), Fn.hip.ptr(sums), real.numel(), 1, Fn.stream()))
            if self._reduce_fn() is not None:          # the moments of the GLOBAL real batch (BigGAN.py:720)
                self._reduce_fn()(sums)
            Fn.check(L.bg_gp_interpolate(Fn.f32(real), Fn.f32(draws["eps"].contiguous()), Fn.f32(draws["alpha"]),
                                         Fn.hip.ptr(sums), float(real.numel() * self.world), Fn.f32(xhat), B, per,
                                         Fn.stream()))
        else:
            Fn.check(L.bg_gp_interpolate(Fn.f32(real), Fn.f32(fake.detach().contiguous()), Fn.f32(draws["alpha"]), None,
                                         0.0, Fn.f32(xhat), B, per, Fn.stream()))
        aug = draws.get("aug")
        if Fn.Precision.resident:          # bf16-resident model: the penalty's passes run on fp32 tensors (bf16 MFMA)
            with Fn.precision_scope("bf16-staged"):
                return self._gradient_penalty_passes(xhat, draws, aug, B, per)
        return self._gradient_penalty_passes(xhat, draws, aug, B, per)

    def _gradient_penalty_passes(self, xhat, draws, aug, B, per):
        from .ops import Dual
        L = Fn.lib()
        # (1) g = d sum(D(aug(x^))) / d x^
        xg = xhat.detach().requires_grad_(True)
        logit = self.discriminator(DiffAugment(xg, policy=self.da_policy, draws=aug), reuse=True)["real"]
        seed = torch.empty_like(logit)
        seed.fill_(1.0)
        with Fn.inputs_only_backward():
            g, = torch.autograd.grad(logit, xg, grad_outputs=seed)
        g = g.contiguous()
        ws = torch.empty(2 * B, dtype=torch.float64, device=self.device)
        value = torch.empty(1, dtype=torch.float32, device=self.device)
        v = torch.empty_like(g)
        Fn.check(L.bg_gp_penalty(Fn.f32(g), B, per, float(B * self.world), float(self.ld),
                                 int(self.gradient_penalty_type == 'wgan-lp'), Fn.hip.ptr(ws), Fn.f32(value), Fn.f32(v),
                                 Fn.stream()))
        red = self._reduce_fn()
        if red is not None:
            red(value)
        # (2) forward-mode pass; the tangent of DiffAugment is DiffAugment without the brightness offset
        if self.da_policy:
            lin = dict(aug)
            lin["u_b"] = torch.full_like(aug["u_b"], 0.5)
            dual = Dual(DiffAugment(xhat, policy=self.da_policy, draws=aug),
                        DiffAugment(v, policy=self.da_policy, draws=lin))
        else:
            dual = Dual(xhat, v)
        fdot = self.discriminator(dual, reuse=True)["real"].t
        return Fn.GpSurrogateFn.apply(fdot, value)

    def d_forward(self, real, z=None, draws_real=None, draws_fake=None, labels=None, cls_z=None, gp_draws=None):
        """BigGAN.py:806-808, 856-883: D(aug(real)), D(aug(G(z))), hinge + flood (+ the label loss on the
        real half when n_labels > 0, BigGAN.py:853).  G runs without a backward graph (d_loss is minimised
        over d_vars only); real and fake go through D as one batch."""
        B = real.shape[0]
        self._begin_run()
        if z is None:
            z = self.sample_z(B)
        if self.acgan and cls_z is None:
            cls_z = self.synthetic_labels(B)                                           # rnd_cls_feed_dict, BigGAN.py:1454
        with torch.no_grad():
            fake = self.generator(z, cls_z, is_training=True)
        real_aug = DiffAugment(real, policy=self.da_policy, draws=draws_real, generator=self.gen)
        fake_aug = DiffAugment(fake, policy=self.da_policy, draws=draws_fake, generator=self.gen)
        if self.bn_in_d:
            # batch norm couples the samples of a call: keep the reference's two instantiations
            # (BigGAN.py:807,857), each with its own batch statistics; moving statistics compound
            d_real, d_fake = self.discriminator(real_aug), self.discriminator(fake_aug, reuse=True)
            real_logits, fake_logits = d_real["real"], d_fake["real"]
            d_out = d_real
        else:
            d_out = self.discriminator(torch.cat([real_aug, fake_aug], dim=0))
            logits = d_out["real"]
            real_logits, fake_logits = logits[:B], logits[B:]
        d_loss = discriminator_loss(self.d_loss_func, real=real_logits, fake=fake_logits, flood_level=self.d_flood)
        out = {"real_logits": real_logits, "fake_logits": fake_logits, "fake": fake}
        if self.use_gradient_penalty:                                                  # BigGAN.py:867-868, 880
            out["gp"] = self.gradient_penalty(real, fake, gp_draws)
            d_loss = Fn.AddFn.apply(d_loss, out["gp"])
        if self.acgan:
            if labels is None:
                raise ValueError("n_labels > 0: d_forward needs the labels of the real batch")
            real_cls = d_out["cls"][:B]             # (a no-op slice when D ran on the real batch alone)
            d_cls = self._cls_loss()(labels, real_cls, self.d_cls_loss_weight, self._reduce_fn(), self.world)
            out["d_cls_loss"] = d_cls
            d_loss = Fn.AddFn.apply(d_loss, d_cls)
        out["d_loss"] = d_loss
        return out

    def _per_virtual_batch(self, k, v):
        if isinstance(v, (list, tuple)):
            if len(v) != self.virtual_batches:
                raise ValueError("expected %d virtual batches, got %d" % (self.virtual_batches, len(v)))
            return v[k]
        return v

    def d_step(self, real, z=None, draws_real=None, draws_fake=None, apply=True, labels=None, cls_z=None,
               defer=False, gp_draws=None):
        """One run of d_ops (utils.py:252-320): with --virtual_batches k, gradients of k forward/backward
        passes (each on its own real batch / z / draws; lists of k are accepted) are accumulated and
        applied once, scaled 1/k; reported losses are means over the k passes."""
        vb = self.virtual_batches
        self.store.begin_backward("discriminator")
        outs = []
        for k in range(vb):
            out = self.d_forward(*[self._per_virtual_batch(k, a) for a in (real, z, draws_real, draws_fake, labels)],
                                 cls_z=cls_z,                                        # one feed_dict for all k
                                 gp_draws=self._per_virtual_batch(k, gp_draws))
            out["d_loss"].backward()
            self._sn_backward("discriminator")
            outs.append(out)
        self.store.zero_untouched("discriminator")
        if self.shards:
            # data parallel: start the exchange of the D gradients (reduce-scatter, or all-reduce when the update is
            # replicated); with ``defer`` return at once - the generator forward of the G step does not read D, so it
            # runs while the collectives are in flight, and _finish_d() (wait, Adam on the owned part, all-gather of
            # the parameters) is called just before the G step's first use of the discriminator
            handles = self._exchange_begin("discriminator", 0, self.d_arena.size, apply)
            self._pending_d = (handles, bool(apply), 1.0 / vb)
            if not defer:
                self._finish_d()
        elif apply:
            self._adam(self.d_arena, self.d_learning_rate, with_ema=False, grad_scale=1.0 / vb)
        return self._mean_losses(outs, ("d_loss", "d_cls_loss", "gp"))

    def _finish_d(self):
        pending = getattr(self, "_pending_d", None)
        if pending is None:
            return
        handles, apply, gscale = pending
        self._pending_d = None
        if apply:
            self._adam_prepare(self.d_arena, self.d_learning_rate)
        self._exchange_end("discriminator", handles, self.d_learning_rate, False, apply, gscale)

    def _mean_losses(self, outs, keys):
        out = outs[-1]
        if len(outs) > 1:
            for key in keys:
                if key in out and out[key] is not None:
                    acc = outs[0][key].detach().clone()
                    for o in outs[1:]:
                        Fn.axpby(o[key].detach(), 1.0, acc, 1.0)
                    out[key] = Fn.axpby(acc, 0.0, acc, 1.0 / len(outs))
        return out

    def g_forward(self, B, z=None, draws_fake=None, cls_z=None, after_generator=None, real=None, draws_real=None):
        """BigGAN.py:896-898: -mean(D(aug(G(z)))) + flood (+ label loss, BigGAN.py:894) + regularisation losses."""
        self._begin_run()
        if z is None:
            z = self.sample_z(B)
        if self.acgan and cls_z is None:
            cls_z = self.synthetic_labels(B)
        fake = self.generator(z, cls_z, is_training=True)
        if after_generator is not None:
            after_generator()               # e.g. the deferred D update: must precede any use of the discriminator
        fake_aug = DiffAugment(fake, policy=self.da_policy, draws=draws_fake, generator=self.gen)
        real_logits = None
        if self.relativistic:
            # g_loss of the ra-* types reads D(aug(real)) too (BigGAN.py:806-808, 896); D's weights are frozen
            # in this op, so the real half needs no backward graph
            if real is None:
                raise ValueError("--gan_type %s: the G step needs a real batch" % self.gan_type)
            real_aug = DiffAugment(real, policy=self.da_policy, draws=draws_real, generator=self.gen)
            if self.bn_in_d:
                real_logits = self.discriminator(real_aug)["real"]
                d_out = self.discriminator(fake_aug, reuse=True)
                fake_logits = d_out["real"]
            else:
                d_out = self.discriminator(torch.cat([real_aug, fake_aug], dim=0))
                nr = real_aug.shape[0]
                real_logits, fake_logits = d_out["real"][:nr], d_out["real"][nr:]
                if self.acgan:
                    d_out = dict(d_out, cls=d_out["cls"][nr:])
        else:
            d_out = self.discriminator(fake_aug)
            fake_logits = d_out["real"]
        g_adv = generator_loss(self.gan_type, fake=fake_logits, real=real_logits, flood_level=self.g_flood)
        out = {"fake_logits": fake_logits, "fake": fake}
        if self.acgan:
            g_cls = self._cls_loss()(cls_z, d_out["cls"], self.g_cls_loss_weight, self._reduce_fn(), self.world)
            out["g_cls_loss"] = g_cls
            g_adv = Fn.AddFn.apply(g_adv, g_cls)
        out["g_adv"] = g_adv
        out["regs"] = ops.get_regularization_losses() if self.g_regularization_method != 'none' else []
        return out

    def g_step(self, B, z=None, draws_fake=None, apply=True, cls_z=None, after_generator=None, real=None,
               draws_real=None):
        vb = self.virtual_batches
        self._set_requires_grad(self.d_vars, False)        # g_loss is minimised over g_vars only
        outs = []
        # data parallel: exchange the generator gradients stage by stage while backward is still running
        overlap = (vb == 1 and self.device.type == "cuda" and bool(getattr(self, "shards", None))
                   and not getattr(self, "_capturing", False))
        self._g_overlap = {"hi": self.g_arena.size, "handles": [], "apply": bool(apply)} if overlap else None
        try:
            self.store.begin_backward("generator")
            for k in range(vb):
                out = self.g_forward(B, self._per_virtual_batch(k, z), self._per_virtual_batch(k, draws_fake), cls_z,
                                     after_generator if k == 0 else None, self._per_virtual_batch(k, real),
                                     self._per_virtual_batch(k, draws_real))
                roots = [out["g_adv"]] + out["regs"]
                ones = torch.ones(1, dtype=torch.float32, device=self.device)
                # each regularisation term is evaluated on exactly one rank (_shard_regularisers): weight 1,
                # the SUM all-reduce of the flat gradient arena then yields the single-process gradient
                torch.autograd.backward(roots, [ones] * len(roots))
                self._sn_backward("generator")
                out["g_reg"] = Fn.zeros(1, torch.float32, self.device)
                if out["regs"]:                                     # reported value: column sum of the terms
                    terms = torch.cat([r.detach().reshape(1) for r in out["regs"]]).reshape(-1, 1)
                    Fn._bias_grad(terms, out["g_reg"])
                if self.world > 1 and self.g_regularization_method != 'none':
                    self._reduce_fn()(out["g_reg"])                 # sum over the owners
                out["g_loss"] = Fn.axpby(out["g_reg"], 1.0, out["g_adv"].detach().clone().reshape(1), 1.0)
                outs.append(out)
        finally:
            self._set_requires_grad(self.d_vars, True)
            st, self._g_overlap = getattr(self, "_g_overlap", None), None
        if st is not None:
            self._g_overlap = st
            if st["hi"] > 0:
                self._g_exchange_range(0, st["hi"])         # first/* (and anything no marker covered)
            self._g_overlap = None
            if apply:
                self._adam_prepare(self.g_arena, self.g_learning_rate)
            # in completion order: the last exchange overlaps the optimiser (and the all-gathers) of the earlier ranges
            self._exchange_end("generator", st["handles"], self.g_learning_rate, True, apply, 1.0)
            return self._mean_losses(outs, ("g_adv", "g_reg", "g_loss", "g_cls_loss"))
        self.store.zero_untouched("generator")
        if self.shards:
            handles = self._exchange_begin("generator", 0, self.g_arena.size, apply)
            if apply:
                self._adam_prepare(self.g_arena, self.g_learning_rate)
            self._exchange_end("generator", handles, self.g_learning_rate, True, apply, 1.0 / vb)
        elif apply:
            self._adam(self.g_arena, self.g_learning_rate, with_ema=True, grad_scale=1.0 / vb)
        return self._mean_losses(outs, ("g_adv", "g_reg", "g_loss", "g_cls_loss"))

    def train_step(self, real, labels=None, real_g=None):
        """One iteration of BigGAN.py:1061-1084.  ``real`` (and ``labels`` when n_labels > 0) may be lists
        of --virtual_batches tensors.  ``real_g``: the real batch of the G op - the reference's g_ops pull a FRESH
        batch from the input iterator (BigGAN.py:1082), which matters for the relativistic losses whose generator
        loss reads D(real); None re-uses the D op's batch.  After ``capture_graphs()`` the iteration is replayed
        from HIP graphs."""
        if getattr(self, "_graphs_ready", False):
            if self.acgan and labels is None:
                labels = self.synthetic_labels(real.shape[0])
            return self._train_step_graph(real, labels, real_g)
        return self._train_step_eager(real, labels, real_g)

    def _train_step_eager(self, real, labels=None, real_g=None):
        losses = {}
        first = real[0] if isinstance(real, (list, tuple)) else real
        if self.acgan and labels is None:
            labels = [self.synthetic_labels(first.shape[0]) for _ in range(self.virtual_batches)]
        run_g = (self.counter - 1) % self.n_critic == 0                               # BigGAN.py:1080
        d = self.d_step(real, labels=labels, defer=run_g)
        losses["d_loss"] = d["d_loss"]
        if d.get("gp") is not None:
            losses["gp"] = d["gp"]
        if run_g:
            g = self.g_step(first.shape[0], after_generator=self._finish_d,
                            real=(real_g if real_g is not None else real) if self.relativistic else None)
            losses["g_loss"] = g["g_loss"]
        self._finish_d()
        self.counter += 1
        return losses

    # ---- HIP-graph replay of the iteration (launch-bound configurations) -------------------------
    def capture_graphs(self, B=None):
        """Capture the D op and the G op (forward, backward, spectral-norm batches, optimiser) into two HIP
        graphs (torch.cuda.CUDAGraph) with static input buffers.  ``train_step`` then copies the batch in,
        refreshes the two Adam step sizes on the host and replays: ~4000 kernel launches per iteration cost
        two graph launches, which is what bounds small configurations (BASELINE config 1 spends 6 ms of a
        16 ms step in kernels).  Single process only (collectives are not captured)."""
        if self.world != 1:
            raise NotImplementedError("graph capture is for single-process runs")
        if self.virtual_batches != 1:
            raise NotImplementedError("graph capture with --virtual_batches > 1")
        B = B or self.batch_size
        self._g_real = torch.zeros(B, self.img_size, self.img_size, self.c_dim, dtype=torch.float32, device=self.device)
        self._g_labels = (torch.zeros(B, self.n_labels, dtype=torch.float32, device=self.device) if self.acgan else None)
        self._g_real_g = torch.zeros_like(self._g_real) if self.relativistic else None     # the G op's own real batch
        if getattr(self.d_arena, "lr_dev", None) is None:
            for arena in (self.d_arena, self.g_arena):
                arena.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        # untimed eager warm-up on a side stream (allocator pools, lazy initialisation), state restored afterwards
        snap = self.state_tensors()
        saved = {k: v.detach().clone() for k, v in snap.items()}
        steps = (self.counter, self.d_arena.step, self.g_arena.step)
        rng = self.gen.get_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._train_step_eager(self._g_real, self._g_labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def restore():
            with torch.no_grad():
                for k, v in snap.items():
                    v.copy_(saved[k])
            self.counter, self.d_arena.step, self.g_arena.step = steps
            self.gen.set_state(rng)
        restore()
        self._capturing = True
        try:
            self._graph_d = torch.cuda.CUDAGraph()
            self._graph_d.register_generator_state(self.gen)
            with torch.cuda.graph(self._graph_d):
                self._g_out_d = self.d_step(self._g_real, labels=self._g_labels)
            self._graph_g = torch.cuda.CUDAGraph()
            self._graph_g.register_generator_state(self.gen)
            with torch.cuda.graph(self._graph_g, pool=self._graph_d.pool()):
                self._g_out_g = self.g_step(B, real=self._g_real_g if self.relativistic else None)
        finally:
            self._capturing = False
        torch.cuda.synchronize()
        restore()                      # capture does not execute kernels, but it advanced host-side bookkeeping
        self._graphs_ready = True
        return self

    def _train_step_graph(self, real, labels, real_g=None):
        self._g_real.copy_(real)
        if self._g_real_g is not None:
            self._g_real_g.copy_(real_g if real_g is not None else real)
        if self._g_labels is not None:
            self._g_labels.copy_(labels)
        losses = {}
        self._adam_prepare(self.d_arena, self.d_learning_rate)
        self._graph_d.replay()
        losses["d_loss"] = self._g_out_d["d_loss"]
        if (self.counter - 1) % self.n_critic == 0:
            self._adam_prepare(self.g_arena, self.g_learning_rate)
            self._graph_g.replay()
            losses["g_loss"] = self._g_out_g["g_loss"]
        self.counter += 1
        return losses

    def synthetic_labels(self, B):
        """Synthetic one-hot labels, uniform classes (the reference draws tags from its label file,
        BigGAN.py:1446-1455)."""
        idx = torch.randint(0, self.n_labels, (B,), device=self.device, generator=self.gen)
        lab = torch.zeros(B, self.n_labels, dtype=torch.float32, device=self.device)
        lab[torch.arange(B, device=self.device), idx] = 1.0
        return lab

    def synthetic_batch(self, B=None):
        """Synthetic images U(-1,1) [B,S,S,c_dim] on the device (the reference reads PNG files)."""
        B = B or self.batch_size
        return torch.rand(B, self.img_size, self.img_size, self.c_dim, device=self.device,
                          generator=self.gen) * 2.0 - 1.0

    @staticmethod
    def settle_host():
        """Take the long-lived object graph (variables, arenas, autograd Function classes, ctypes tables: hundreds of
        thousands of objects) out of Python's cyclic garbage collector: a generation-2 collection that has to traverse it
        stalls the enqueueing thread for 45-75 ms (measured r02 at iteration 6 of a config-3 run, tools/host_time.py) -
        under data parallelism every rank waits for the stalled one.  Call after build_model() (and a first iteration)."""
        import gc
        gc.collect()
        gc.freeze()

    def train(self, data_fn=None, iterations=None, resume=True):
        """BigGAN.py:1015-1118 training loop (synthetic data unless ``data_fn`` is given): resume from the
        latest checkpoint of ``checkpoint_dir`` if there is one, print the losses every iteration, save
        every ``save_freq`` iterations of an epoch (rank 0 writes; replicas are identical)."""
        loader = None
        if data_fn is None:
            loader = self.open_dataset()
            if loader is not None:
                data_fn = lambda: next(loader)                                    # noqa: E731
        try:
            return self._train_loop(data_fn, iterations, resume)
        finally:
            if loader is not None:
                loader.close()

    def open_dataset(self, root="./dataset"):
        """BigGAN.py:195-212, 768-787: the files of ``<root>/<--dataset>/`` (+ ``--label_file``) behind the
        reference's shuffle / decode / resize / flip / batch pipeline; None when that folder does not exist
        (the training loop then runs on synthetic batches)."""
        from . import data as D
        folder = os.path.join(root, self.dataset_name)
        if not os.path.isdir(folder):
            return None
        files, labels = D.load_data(self.dataset_name, self.args.label_file, self.args.weight_file,
                                    ignore_missing=self.args.ignore_missing_labels, n_labels=self.n_labels, root=root)
        if self.acgan and labels is None:
            raise ValueError("--n_labels > 0 needs --label_file")
        print("# dataset number:", len(files))
        image_data = D.ImageData(self.img_size, self.c_dim, True, self.args.random_flip, seed=1234 + self.rank)
        return D.BatchLoader(files, labels if self.acgan else None, self.batch_size, image_data, self.device,
                             seed=4321, rank=self.rank, world=self.world)

    def _train_loop(self, data_fn, iterations, resume):
        could_load, checkpoint_counter = (self.load(self.checkpoint_dir) if resume else (False, 0))
        if could_load:
            start_epoch = int(checkpoint_counter / self.iterations_per_epoch)
            start_batch_id = checkpoint_counter - start_epoch * self.iterations_per_epoch
            print(" [*] Load SUCCESS")
        else:
            start_epoch, start_batch_id = 0, 0
            if resume:
                print(" [!] Load failed...")
        start_time = time.time()
        done = 0
        self.settle_host()
        for epoch in range(start_epoch, self.epoch):
            for idx in range(start_batch_id, self.iterations_per_epoch):
                if iterations is not None and done >= iterations:
                    return
                batch = data_fn() if data_fn is not None else self.synthetic_batch()
                real_g = None
                if self.relativistic and (self.counter - 1) % self.n_critic == 0:  # g_ops take their own batch (BigGAN.py:1082)
                    nxt = data_fn() if data_fn is not None else self.synthetic_batch()
                    real_g = nxt[0] if isinstance(nxt, tuple) else nxt
                if isinstance(batch, tuple):                                       # (images, labels) with --n_labels
                    losses = self.train_step(batch[0], labels=batch[1], real_g=real_g)
                else:
                    losses = self.train_step(batch, real_g=real_g)
                done += 1
                vals = {k: float(v.item()) for k, v in losses.items()}
                print_str = "Step: %5d, time: %4.4f" % (self.counter, time.time() - start_time)   # BigGAN.py:1109-1116
                for name, val in vals.items():
                    print_str += ", " + name + ": %.4f" % val
                if self.rank == 0:
                    print(print_str, flush=True)
                if (idx + 1) % self.save_freq == 0:                                   # BigGAN.py:1121-1122
                    self.save(self.checkpoint_dir, self.counter)
            start_batch_id = 0                                                         # BigGAN.py:1164-1166
            self.save(self.checkpoint_dir, self.counter)

    # ---- sampling with the EMA weights (BigGAN.py:963-971) ------------------------------------
    def sample(self, z=None, cls_z=None, B=None, use_ema=True):
        """``self.fake_images``: generator(test_z, zero_cls_z, is_training=False, custom_getter=ema_getter).
        Trainables are read from their ExponentialMovingAverage shadows, non-trainables (``u``, moving
        statistics) from the live variables; batch norm uses the population statistics; the spectral-norm
        power iteration still advances ``u`` (its assign is a control dependency of w / sigma, ops.py:743)."""
        B = B or (z.shape[0] if z is not None else self.batch_size)
        self.sync_sharded_state()             # (data parallel with a sharded update: collective, every rank samples)
        if z is None:
            z = self.sample_z(B)
        if self.acgan and cls_z is None:
            cls_z = torch.zeros(B, self.n_labels, dtype=torch.float32, device=self.device)   # zero_cls_z
        arena = self.g_arena
        live = None
        if use_ema and arena.ema is not None:
            live = arena.params.clone()
            arena.params.copy_(arena.ema)
        try:
            S.set_default_store(self.store)
            ops.begin_run(None, 1)
            with torch.no_grad():
                img = self.generator(z, cls_z, is_training=False, reuse=True)
        finally:
            if live is not None:
                arena.params.copy_(live)
        return img

    def test(self):
        """--phase test (BigGAN.py:1372-1395): load the latest checkpoint and write ``test_num`` grids of
        floor(sqrt(min(sample_num, batch_size)))^2 EMA samples to ``<result_dir>/<model_dir>/``."""
        import numpy as np
        from .utils import save_images, check_folder
        could_load, _ = self.load(self.checkpoint_dir)
        result_dir = os.path.join(self.args.result_dir, self.model_dir)
        check_folder(result_dir)
        print(" [*] Load SUCCESS" if could_load else " [!] Load failed...")
        tot_num_samples = min(self.args.sample_num, self.batch_size)
        dim = int(np.floor(np.sqrt(tot_num_samples)))
        paths = []
        for i in range(self.args.test_num):
            samples = self.sample(B=self.batch_size).detach().cpu().numpy()
            paths.append(save_images(samples[:dim * dim, :, :, :], [dim, dim],
                                     result_dir + '/' + self.model_name + '_test_{}.png'.format(i)))
        return paths

    # ---- checkpoints (BigGAN.py:1255-1284) ---------------------------------------------------
    def _ckpt_dir(self, checkpoint_dir):
        return os.path.join(checkpoint_dir, self.model_dir)

    def state_tensors(self):
        """name -> tensor, keyed like a TF checkpoint of the reference graph: variables by their scope
        names, ``<var>/ExponentialMovingAverage`` shadows, Adam slots ``<var>/Adam`` (m) and ``<var>/Adam_1`` (v).

        Key convention (differs from the reference's swapping_saver, BigGAN.py:1018): here ``<var>`` is the LIVE
        weight and ``<var>/ExponentialMovingAverage`` its shadow, as in a plain tf.train.Saver.  The reference's
        swapping saver stores them the other way round (averages under the variable names), so a converter between
        the two formats must swap the pair for every generator trainable."""
        out = {}
        for k, v in self.store.vars.items():
            out[k] = v.detach()
        for group, arena in self.store.arenas.items():
            for name in arena.names:
                out[name + "/Adam"] = arena.view(arena.m, name)
                out[name + "/Adam_1"] = arena.view(arena.v, name)
                if arena.ema is not None:
                    out[name + "/ExponentialMovingAverage"] = arena.view(arena.ema, name)
        return out

    def save(self, checkpoint_dir, step):
        """``<checkpoint_dir>/<model_dir>/BigGAN.model-<step>.safetensors`` + a TF-style ``checkpoint`` index.
        Under data parallelism every rank calls it (the sharded optimiser state is gathered first); rank 0 writes."""
        from safetensors.torch import save_file
        self.sync_sharded_state()
        if self.rank != 0:
            return None
        d = self._ckpt_dir(checkpoint_dir)
        os.makedirs(d, exist_ok=True)
        tensors = {k: v.detach().to("cpu").contiguous().clone() for k, v in self.state_tensors().items()}
        tensors["_meta/steps"] = torch.tensor([self.counter, self.d_arena.step, self.g_arena.step], dtype=torch.int64)
        name = "%s.model-%d" % (self.model_name, int(step))
        save_file(tensors, os.path.join(d, name + ".safetensors"))
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % name)
        self._prune_checkpoints(d)
        return os.path.join(d, name + ".safetensors")

    def _prune_checkpoints(self, d):
        """swapping_saver(max_to_keep=keep_checkpoints) (BigGAN.py:1018): keep the newest N checkpoint files."""
        keep = int(getattr(self.args, "keep_checkpoints", 0) or 0)
        if keep <= 0:
            return
        pre, suf = self.model_name + ".model-", ".safetensors"
        found = []
        for f in os.listdir(d):
            if f.startswith(pre) and f.endswith(suf):
                try:
                    found.append((int(f[len(pre):-len(suf)]), f))
                except ValueError:
                    pass
        for _, f in sorted(found)[:-keep]:
            os.remove(os.path.join(d, f))

    def load_checkpoint(self, checkpoint_path):
        from safetensors.torch import load_file
        if not checkpoint_path.endswith(".safetensors"):
            checkpoint_path += ".safetensors"
        tensors = load_file(checkpoint_path)
        dst = self.state_tensors()
        missing = [k for k in dst if k not in tensors]
        if missing:
            raise KeyError("checkpoint %s lacks %d tensors, e.g. %s" % (checkpoint_path, len(missing), missing[:3]))
        with torch.no_grad():
            for k, t in dst.items():
                src = tensors[k]
                if tuple(src.shape) != tuple(t.shape):
                    raise ValueError("checkpoint tensor %s has shape %s, model expects %s"
                                     % (k, tuple(src.shape), tuple(t.shape)))
                t.copy_(src.to(t.device))
        steps = tensors.get("_meta/steps")
        ckpt_name = os.path.basename(checkpoint_path)[:-len(".safetensors")]
        counter = int(ckpt_name.split('-')[-1])
        if steps is not None:
            self.d_arena.step, self.g_arena.step = int(steps[1]), int(steps[2])
        self.counter = counter
        print(" [*] Successfully read {}".format(ckpt_name))
        return True, counter

    def load(self, checkpoint_dir):
        print(" [*] Reading checkpoints...")
        d = self._ckpt_dir(checkpoint_dir)
        if getattr(self.args, "checkpoint", ""):
            return self.load_checkpoint(self.args.checkpoint)
        index = os.path.join(d, "checkpoint")
        if os.path.exists(index):
            line = open(index).read().strip().splitlines()[0]
            name = line.split(":", 1)[1].strip().strip('"')
            return self.load_checkpoint(os.path.join(d, name))
        print(" [*] Failed to find a checkpoint")
        return False, 0

    @property
    def model_dir(self):
        """BigGAN.py:1246-1253."""
        sn = '_sn' if self.sn else ''
        return "{}_{}_{}_{}_{}{}".format(self.model_name, self.dataset_name, self.gan_type, self.img_size,
                                         self.z_dim, sn)
