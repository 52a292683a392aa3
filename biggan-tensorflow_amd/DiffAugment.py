"""Differentiable augmentation with the reference's entry point
``DiffAugment(x, policy='', channels_first=False)`` (``/root/reference/DiffAugment_tf.py:8-17``),
executed as ONE fused gather kernel (plus a per-sample mean reduction for rand_contrast).

The seven random draws of a 'color,translation,cutout' call are made on the device with torch's
generator (or passed explicitly through ``draws`` for parity tests); the integer index math runs in
the kernel and is bit-exact with DiffAugment_tf.py:40-66.
"""
import torch

from . import functional as Fn

_BITS = {"color": 1, "translation": 2, "cutout": 4}
_ORDER = ["color", "translation", "cutout"]


def augment_params(S):
    """(shift, cutout_size, exclusive max of the cutout offset) for image size S
    (DiffAugment_tf.py:43, 56-58)."""
    shift = int(S * 0.125 + 0.5)
    cs = int(S * 0.5 + 0.5)
    return shift, cs, S + (1 - cs % 2)


def draw(B, S, device, generator=None):
    """The draws of one call in graph order: brightness, saturation, contrast (U[0,1) fp32),
    translation x/y (int32 in [-shift, shift]), cutout offset x/y (int32 in [0, S + 1 - cs%2))."""
    shift, cs, off_max = augment_params(S)
    kw = dict(device=device, generator=generator)
    return {
        "u_b": torch.rand(B, dtype=torch.float32, **kw),
        "u_s": torch.rand(B, dtype=torch.float32, **kw),
        "u_c": torch.rand(B, dtype=torch.float32, **kw),
        "t_x": torch.randint(-shift, shift + 1, (B,), dtype=torch.int32, **kw),
        "t_y": torch.randint(-shift, shift + 1, (B,), dtype=torch.int32, **kw),
        "o_x": torch.randint(0, off_max, (B,), dtype=torch.int32, **kw),
        "o_y": torch.randint(0, off_max, (B,), dtype=torch.int32, **kw),
    }


def draws_to_device(draws, device):
    out = {}
    for k, v in draws.items():
        t = torch.as_tensor(v)
        t = t.to(torch.float32 if k.startswith("u_") else torch.int32)
        out[k] = t.to(device).contiguous()
    return out


def policy_bits(policy):
    bits = 0
    last = -1
    for p in policy.split(','):
        if p not in _BITS:
            raise KeyError(p)                       # AUGMENT_FNS[p] in the reference
        idx = _ORDER.index(p)
        if idx <= last:
            raise NotImplementedError("DiffAugment policy order %r: the fused kernel applies "
                                      "color -> translation -> cutout" % policy)
        last = idx
        bits |= _BITS[p]
    return bits


def DiffAugment(x, policy='', channels_first=False, draws=None, generator=None):
    if not policy:
        return x
    if channels_first:
        raise NotImplementedError("channels_first=True is never used by the reference's caller")
    bits = policy_bits(policy)
    B, S = x.shape[0], x.shape[1]
    if draws is None:
        draws = draw(B, S, x.device, generator)
    d = draws
    return Fn.DiffAugmentFn.apply(x, d["u_b"], d["u_s"], d["u_c"], d["t_x"], d["t_y"], d["o_x"], d["o_y"], bits)


# ------------------------------------------------------------------------------------------
# The reference's per-transform names (DiffAugment_tf.py:20-73).  Each is the fused kernel restricted to one
# transform, with its own draws (same distributions, same order) unless ``draws`` is given.
# ------------------------------------------------------------------------------------------
def _single(x, bits, draws=None, generator=None):
    d = draws if draws is not None else draw(x.shape[0], x.shape[1], x.device, generator)
    return Fn.DiffAugmentFn.apply(x, d["u_b"], d["u_s"], d["u_c"], d["t_x"], d["t_y"], d["o_x"], d["o_y"], bits)


def _neutral_color(x, draws, generator, keep):
    """Draws of a colour-only call in which every colour transform except ``keep`` is the identity
    (brightness offset 0 at u_b = 0.5, saturation factor 1 at u_s = 0.5, contrast factor 1 at u_c = 0.5)."""
    d = dict(draws if draws is not None else draw(x.shape[0], x.shape[1], x.device, generator))
    for k in ("u_b", "u_s", "u_c"):
        if k != keep:
            d[k] = torch.full_like(d[k], 0.5)
    return d


def rand_brightness(x, draws=None, generator=None):
    """DiffAugment_tf.py:20-23: x + (u - 0.5)."""
    return _single(x, _BITS["color"], _neutral_color(x, draws, generator, "u_b"))


def rand_saturation(x, draws=None, generator=None):
    """DiffAugment_tf.py:26-30: (x - mean_C) * 2u + mean_C."""
    return _single(x, _BITS["color"], _neutral_color(x, draws, generator, "u_s"))


def rand_contrast(x, draws=None, generator=None):
    """DiffAugment_tf.py:33-37: (x - mean_HWC) * (u + 0.5) + mean_HWC."""
    return _single(x, _BITS["color"], _neutral_color(x, draws, generator, "u_c"))


def rand_translation(x, ratio=0.125, draws=None, generator=None):
    """DiffAugment_tf.py:40-50 (ratio is fixed at the reference's 0.125 inside the kernel)."""
    if ratio != 0.125:
        raise NotImplementedError("rand_translation ratio != 0.125")
    return _single(x, _BITS["translation"], draws, generator)


def rand_cutout(x, ratio=0.5, draws=None, generator=None):
    """DiffAugment_tf.py:53-66 (ratio is fixed at the reference's 0.5 inside the kernel)."""
    if ratio != 0.5:
        raise NotImplementedError("rand_cutout ratio != 0.5")
    return _single(x, _BITS["cutout"], draws, generator)


AUGMENT_FNS = {                                  # DiffAugment_tf.py:69-73
    'color': [rand_brightness, rand_saturation, rand_contrast],
    'translation': [rand_translation],
    'cutout': [rand_cutout],
}
