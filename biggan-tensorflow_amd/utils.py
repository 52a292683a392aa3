"""Host-side helpers with the reference's names (``/root/reference/utils.py``) that the hot path
needs: flag parsing helper, channel rounding, directory creation, the generator weight
regularisers and the step-op glue."""
import os

import torch


def check_folder(log_dir):
    """utils.py:164-167."""
    if not os.path.exists(log_dir):
        os.makedirs(log_dir)
    return log_dir


def str2bool(x):
    """utils.py:173-174 - a SUBSTRING test, kept verbatim: '', 't', 'rue' parse as True."""
    return x.lower() in ('true')


def round_up(val, multiple):
    """utils.py:335-336."""
    return (int(val) + multiple - 1) // multiple * multiple


def parse_int_list(str):
    """utils.py:237-239."""
    if str == 'none' or str == '':
        return []
    return [int(x.strip()) for x in str.split(",")]


##################################################################################
# Regularization (utils.py:180-235)
##################################################################################
def orthogonal_regularizer(scale, type='ortho'):
    """utils.py:185-211.  Returns a callable w[k,k,a,c] -> scalar loss tensor (device)."""
    if type not in ('ortho', 'ortho_cosine'):
        raise ValueError("Unknown regularization method.")

    def ortho_reg(w):
        from . import functional as Fn
        return Fn.OrthoCosineRegFn.apply(w, scale, type)
    return ortho_reg


def l2_regularizer(scale):
    """tf.contrib.layers.l2_regularizer(scale) (BigGAN.py:268-270)."""
    def l2_reg(w):
        from . import functional as Fn
        return Fn.L2RegFn.apply(w, scale)
    return l2_reg


def orthogonal_regularizer_fc(scale, type='ortho'):
    """utils.py:213-235 (same arithmetic on a [Cin, units] kernel)."""
    return orthogonal_regularizer(scale, type)


##################################################################################
# Class-label loss (utils.py:323-377)
##################################################################################
def cls_loss_fn(type, cls_weights):
    """utils.py:366-369: loss(truth, answer) = mean(sigmoid_cross_entropy_with_logits(truth, answer) * w).
    ``cls_weights`` is a device tensor [n_labels] (or None = ones).  The returned callable takes an
    optional ``loss_weight`` folded into the kernel (BigGAN.py:853,894)."""
    if type != 'logistic':
        if type == 'euclidean' or '-' in str(type):
            raise NotImplementedError("cls_loss_type '%s' is outside the hot path (only 'logistic')" % type)
        raise ValueError("Invalid label loss type: " + str(type))

    def loss(truth, answer, loss_weight=1.0, reduce_fn=None, world=1):
        from . import functional as Fn
        return Fn.SigmoidCeLossFn.apply(truth, answer, cls_weights, loss_weight, reduce_fn, world)
    return loss


def add_n(tensors):
    """tf.add_n over 1-element device tensors (loss bookkeeping, BigGAN.py:898)."""
    out = tensors[0]
    for t in tensors[1:]:
        out = out + t
    return out
