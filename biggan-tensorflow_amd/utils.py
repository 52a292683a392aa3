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
# Sample grids (utils.py:133-161)
##################################################################################
def inverse_transform(images):
    """utils.py:160-161: [-1, 1] -> [0, 1]."""
    return (images + 1.) / 2.


def merge(images, size):
    """utils.py:136-154: tile [n, h, w, c] images into a size[0] x size[1] grid (row-major)."""
    import numpy as np
    h, w = images.shape[1], images.shape[2]
    if images.shape[3] in (3, 4):
        c = images.shape[3]
        img = np.zeros((h * size[0], w * size[1], c))
        for idx, image in enumerate(images):
            i = idx % size[1]
            j = idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w, :] = image
        return img
    elif images.shape[3] == 1:
        img = np.zeros((h * size[0], w * size[1]))
        for idx, image in enumerate(images):
            i = idx % size[1]
            j = idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w] = image[:, :, 0]
        return img
    raise ValueError('in merge(images,size) images parameter must have dimensions: HxW or HxWx3 or HxWx4')


def imsave(images, size, path):
    """utils.py:156-157 (imageio.imwrite): 8-bit PNG written with zlib only (no imaging library here)."""
    import struct
    import zlib
    import numpy as np
    img = merge(images, size)
    a = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    color_type = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + a[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(png)
    return path


def save_images(images, size, image_path):
    """utils.py:133-134."""
    return imsave(inverse_transform(images), size, image_path)


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
