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
