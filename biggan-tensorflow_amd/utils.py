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
    if type == 'ortho':
        raise NotImplementedError("g_regularization 'ortho' is outside the default hot path (use ortho_cosine or none)")
    if type != 'ortho_cosine':
        raise ValueError("Unknown regularization method.")

    def ortho_reg(w):
        from . import functional as Fn
        return Fn.OrthoCosineRegFn.apply(w, scale)
    return ortho_reg


def orthogonal_regularizer_fc(scale, type='ortho'):
    """utils.py:213-235 (same arithmetic on a [Cin, units] kernel)."""
    return orthogonal_regularizer(scale, type)


def add_n(tensors):
    """tf.add_n over 1-element device tensors (loss bookkeeping, BigGAN.py:898)."""
    out = tensors[0]
    for t in tensors[1:]:
        out = out + t
    return out
