"""ctypes binding of ``libbiggan_hip.so`` (C ABI: ``include/biggan_hip.h``).

The loader fails loudly: a missing library raises ImportError at first use, a non-zero return code
raises RuntimeError with ``bg_last_error()``; tensors must be CUDA/fp32/contiguous.  Nothing here
computes on the host.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_LIB = None
LIB_NAME = "libbiggan_hip.so"
ABI_VERSION = 3
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

PAD_REFLECT, PAD_ZERO = 0, 1
PAD_ZERO_FILL, PAD_SPLIT, PAD_DUP, PAD_FOLD = 0, 1, 2, 3      # bg_pad_channels modes
F32, BF16 = 0, 1                       # BG_F32 / BG_BF16: element types of activation tensors
COMPUTE_F32, COMPUTE_BF16 = 0, 1       # per-call arithmetic of the GEMM-shaped launches
SN_ALL, SN_POWER, SN_NORMALIZE = 0, 1, 2      # bg_spectral_norm_batch_phase


class BgConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in
                ("N", "H", "W", "Cin", "Ho", "Wo", "Cout", "k", "stride", "pad_lo", "pad_mode",
                 "compute", "x_dtype", "y_dtype", "w_packed")]


class BgGemmDesc(Structure):
    _fields_ = [("M", c_int32), ("N", c_int32), ("K", c_int32), ("transA", c_int32), ("transB", c_int32),
                ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("batch", c_int32),
                ("strideA", c_int64), ("strideB", c_int64), ("strideC", c_int64),
                ("compute", c_int32), ("reserved", c_int32)]


class BgSnItem(Structure):
    _fields_ = [("w", c_void_p), ("u", c_void_p), ("v", c_void_p), ("sigma", c_void_p), ("w_norm", c_void_p),
                ("g_wnorm", c_void_p), ("dw", c_void_p), ("ws_offset", c_int64), ("rows", c_int32), ("cols", c_int32),
                ("pack_p", c_void_p), ("pack_t", c_void_p), ("taps", c_int32), ("pack_p_ld", c_int32)]


class BgDenseItem(Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int64), ("w", c_void_p), ("bias", c_void_p), ("y", c_void_p), ("dw", c_void_p),
                ("db", c_void_p), ("K", c_int32), ("N", c_int32), ("acc_w", c_int32), ("acc_b", c_int32)]


DENSE_GROUP_MAX = 8


class BgAttn16Desc(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "N", "Nk", "d", "dv", "reserved")] + \
               [(n, c_int64) for n in ("ldq", "sq", "ldk", "sk", "ldv", "sv", "ldo", "so",
                                       "ldg", "sg", "lddq", "sdq", "lddk", "sdk", "lddv", "sdv")]


_P = c_void_p
_AD = POINTER(BgAttn16Desc)
_CD = POINTER(BgConvDesc)
_GD = POINTER(BgGemmDesc)

# name -> (restype, argtypes); must list every symbol declared in include/biggan_hip.h
SIGNATURES = {
    "bg_abi_version": (c_int, []),
    "bg_last_error": (c_char_p, []),
    "bg_target_arch": (c_char_p, []),
    "bg_png_unfilter": (c_int, [c_char_p, c_int, c_int, c_int, _P]),
    "bg_conv2d_fwd_workspace_bytes": (c_size_t, [_CD]),
    "bg_conv2d_fwd": (c_int, [_CD, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "bg_conv2d_dgrad_workspace_bytes": (c_size_t, [_CD]),
    "bg_conv2d_dgrad": (c_int, [_CD, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "bg_conv2d_wgrad_workspace_bytes": (c_size_t, [_CD]),
    "bg_conv2d_wgrad": (c_int, [_CD, _P, _P, _P, _P, c_size_t, _P]),
    "bg_deconv2d_fwd_workspace_bytes": (c_size_t, [_CD]),
    "bg_deconv2d_fwd": (c_int, [_CD, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "bg_deconv2d_fwd_stats_workspace_bytes": (c_size_t, [_CD]),
    "bg_deconv2d_fwd_stats": (c_int, [_CD, _P, _P, _P, _P, _P, c_int, _P, _P, c_size_t, _P, c_size_t, _P]),
    "bg_deconv2d_dgrad_workspace_bytes": (c_size_t, [_CD]),
    "bg_deconv2d_dgrad": (c_int, [_CD, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "bg_deconv2d_wgrad_workspace_bytes": (c_size_t, [_CD]),
    "bg_deconv2d_wgrad": (c_int, [_CD, _P, _P, _P, _P, c_size_t, _P]),
    "bg_rgbconv_supported": (c_int, [_CD]),
    "bg_rgbconv_fwd": (c_int, [_CD, _P, _P, _P, _P, c_int, _P]),
    "bg_rgbconv_dgrad": (c_int, [_CD, _P, _P, _P, c_int, _P]),
    "bg_rgbconv_wgrad_workspace_bytes": (c_size_t, [_CD]),
    "bg_rgbconv_wgrad": (c_int, [_CD, _P, _P, _P, _P, c_size_t, _P]),
    "bg_dense_group_fwd": (c_int, [POINTER(BgDenseItem), c_int, c_int, _P]),
    "bg_dense_group_wgrad": (c_int, [POINTER(BgDenseItem), c_int, c_int, _P]),
    "bg_gemm_workspace_bytes": (c_size_t, [_GD]),
    "bg_gemm": (c_int, [_GD, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "bg_attention2_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "bg_attention2_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bg_attention2_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bg_attention16_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "bg_attention16_fwd": (c_int, [_AD, _P, _P, _P, _P, _P, _P]),
    "bg_attention16_bwd": (c_int, [_AD, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bg_spectral_norm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "bg_spectral_norm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "bg_spectral_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "bg_spectral_norm_batch_fwd": (c_int, [_P, c_int, _P, c_size_t, _P]),
    "bg_spectral_norm_batch_phase": (c_int, [_P, c_int, _P, c_size_t, c_int, _P]),
    "bg_spectral_norm_batch_bwd": (c_int, [_P, c_int, _P, _P, _P, c_size_t, _P]),
    "bg_bn_stats": (c_int, [_P, _P, c_int64, c_int, _P]),
    "bg_bn_finalize": (c_int, [_P, c_double, c_float, c_float, c_int, _P, _P, _P, _P, c_int, _P]),
    "bg_bn_apply_act_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "bg_bn_apply_act_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "bg_bn_apply_act_bwd_dx": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "bg_bn_bwd_finalize": (c_int, [_P, _P, c_int, c_double, _P, _P, _P, _P, c_int, c_int, _P]),
    "bg_bn_population": (c_int, [_P, _P, c_float, _P, _P, c_int, _P]),
    "bg_renorm_coeffs": (c_int, [_P, c_double, _P, _P, c_int, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                                 c_int, _P, _P, c_int, _P]),
    "bg_renorm_affine_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_renorm_affine_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_symmetrize": (c_int, [_P, _P, c_int, _P]),
    "bg_gp_interpolate": (c_int, [_P, _P, _P, _P, c_double, _P, c_int, c_int64, _P]),
    "bg_gp_penalty": (c_int, [_P, c_int, c_int64, c_double, c_float, c_int, _P, _P, _P, _P]),
    "bg_maxpool2_gather": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bg_prelu_tangent_dalpha": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_softmax_tangent_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_prelu_fwd": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "bg_prelu_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_maxpool2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "bg_maxpool2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bg_box2_down": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "bg_box2_up": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "bg_softmax_fwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "bg_softmax_bwd": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "bg_sum_pool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "bg_sum_pool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "bg_axpby": (c_int, [_P, c_float, _P, c_float, c_int64, _P]),
    "bg_add": (c_int, [_P, _P, _P, c_int64, _P]),
    "bg_scale_add": (c_int, [_P, _P, _P, _P, c_int64, _P]),
    "bg_dot": (c_int, [_P, _P, _P, c_int64, _P]),
    "bg_scale_dev": (c_int, [_P, _P, _P, c_int64, _P]),
    "bg_tanh_fwd": (c_int, [_P, _P, c_int64, _P]),
    "bg_tanh_bwd": (c_int, [_P, _P, _P, c_int64, _P]),
    "bg_bias_grad": (c_int, [_P, _P, c_int64, c_int, _P]),
    "bg_diffaugment_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "bg_diffaugment_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "bg_hinge_d_sums": (c_int, [_P, _P, _P, c_int, _P]),
    "bg_hinge_d_grad": (c_int, [_P, _P, _P, c_double, c_float, _P, _P, _P, c_int, _P]),
    "bg_hinge_g_sums": (c_int, [_P, _P, c_int, _P]),
    "bg_hinge_g_grad": (c_int, [_P, c_double, c_float, _P, _P, c_int, _P]),
    "bg_gan_loss_means": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "bg_gan_loss_terms": (c_int, [c_int, c_int, _P, _P, _P, c_double, c_double, _P, c_int, c_int, _P]),
    "bg_gan_loss_grad": (c_int, [c_int, c_int, _P, _P, _P, _P, c_double, c_double, c_float, _P, _P, _P, c_int, c_int,
                                 _P]),
    "bg_sigmoid_ce": (c_int, [_P, _P, _P, c_float, _P, _P, c_int, c_int, _P]),
    "bg_ortho_cosine_fwd_bwd": (c_int, [_P, c_float, _P, _P, c_int, _P]),
    "bg_ortho_identity_fwd_bwd": (c_int, [_P, c_float, _P, _P, c_int, _P]),
    "bg_gemv_rows": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "bg_ortho_lowrank_cols": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P]),
    "bg_ortho_lowrank_finish": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "bg_adam_tf_ema_step": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                                    c_int64, _P]),
    "bg_adam_tf_ema_step_dev": (c_int, [_P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int64,
                                        _P]),
    "bg_chan_dots3": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_bn_tangent_fwd_coefs": (c_int, [_P, c_double, _P, _P, _P, _P, _P, c_int, _P]),
    "bg_bn_tangent_bwd_coefs": (c_int, [_P, c_double, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "bg_chan_lincomb3": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_gram16_workspace_bytes": (c_size_t, [c_int, c_int]),
    "bg_gram16": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "bg_cast": (c_int, [_P, c_int, _P, c_int, c_int64, _P]),
    "bg_pad_channels": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_int, c_int64, c_int, _P]),
    "bg_weight_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "bg_bn_stats_t": (c_int, [_P, c_int, _P, c_int64, c_int, _P]),
    "bg_bn_apply_act_fwd_t": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "bg_bn_apply_act_bwd_reduce_t": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int,
                                             _P]),
    "bg_bn_apply_act_bwd_dx_t": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int,
                                         _P]),
    "bg_prelu_fwd_t": (c_int, [_P, c_int, _P, _P, c_int, c_int64, c_int, _P]),
    "bg_prelu_bwd_t": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_int64, c_int, _P]),
    "bg_bias_grad_t": (c_int, [_P, c_int, _P, c_int64, c_int, _P]),
    "bg_maxpool2_fwd_t": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bg_maxpool2_bwd_t": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "bg_sum_pool_fwd_t": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "bg_sum_pool_bwd_t": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "bg_lincomb_t": (c_int, [_P, _P, c_float, _P, c_float, _P, c_int, c_int64, _P]),
    "bg_dot_t": (c_int, [_P, _P, c_int, _P, c_int64, _P]),
    "bg_prof_enable": (None, [c_int]),
    "bg_prof_reset": (None, []),
    "bg_prof_collect": (c_int, [POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "bg_prof_dump": (c_int, [c_char_p]),
}


def lib():
    """The loaded library (loads on first call).  Raises ImportError if it was not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % (LIB_NAME, LIB_PATH))
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L.bg_abi_version() != ABI_VERSION:
            raise ImportError("libbiggan_hip.so ABI version %d != %d (rebuild: python -c 'import __graft_entry__ as g; "
                              "g.build()')" % (L.bg_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError("libbiggan_hip: " + lib().bg_last_error().decode())


def ptr(t):
    """Device pointer of a CUDA fp32 / int32 contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("biggan_tensorflow_amd ops run on the MI355X only: got a %s tensor "
                           "(there is no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def f32(t):
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError("expected float32, got %s" % t.dtype)
    return ptr(t)


def dt(t):
    """BG_F32 / BG_BF16 code of a tensor's element type."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError("expected float32 or bfloat16, got %s" % t.dtype)


def act(t):
    """Device pointer of an activation tensor (fp32 or bf16)."""
    if t is not None:
        dt(t)
    return ptr(t)


def i32(t):
    if t is not None and t.dtype != torch.int32:
        raise RuntimeError("expected int32, got %s" % t.dtype)
    return ptr(t)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device as a void* (torch.cuda.current_stream() builds a Stream object per
    call: 9 us, 2600 calls per training iteration; the raw getter is 0.3 us)."""
    if _raw_stream is not None and _raw_device is not None:
        return c_void_p(_raw_stream(_raw_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad_lo, pad_mode, compute=COMPUTE_F32, x_dtype=F32, y_dtype=F32,
              w_packed=0):
    return BgConvDesc(N, H, W, Cin, Ho, Wo, Cout, k, stride, pad_lo, pad_mode, compute, x_dtype, y_dtype, w_packed)


def workspace(nbytes, device):
    """Scratch from PyTorch's caching allocator (no allocation inside the library)."""
    n = max(int(nbytes), 16)
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device)


def scratch(query, desc, device):
    """(ws tensor or None, nbytes) for a *_workspace_bytes query."""
    nb = int(query(desc))
    if nb == 0:
        return None, 0
    return workspace(nb, device), nb
