"""Known-answer formulations of the TensorFlow op semantics the reference relies on, written as
direct NumPy loops from TF's documentation (not from measurement: TensorFlow is absent).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  They pin the alignment conventions the fast
formulations in ``ref_ops.py`` (torch.nn.functional) and the HIP kernels must reproduce:

* ``conv2d_valid``            tf.nn.conv2d(padding='VALID')            (ops.py:94-95)
* ``reflect_pad``             tf.pad(mode='REFLECT')                   (ops.py:82)
* ``conv2d_transpose_same``   tf.nn.conv2d_transpose(padding='SAME') defined, as TF does, as the
                              gradient of a SAME conv2d w.r.t. its input (ops.py:128)
"""
import numpy as np


def reflect_pad(x, lo, hi):
    """x [B,H,W,C] -> [B,H+lo+hi,W+lo+hi,C]; REFLECT mirrors without repeating the edge pixel."""
    B, H, W, C = x.shape

    def src(i, n):
        j = i - lo
        if j < 0:
            j = -j
        if j >= n:
            j = 2 * (n - 1) - j
        return j
    out = np.empty((B, H + lo + hi, W + lo + hi, C), x.dtype)
    for i in range(H + lo + hi):
        for j in range(W + lo + hi):
            out[:, i, j] = x[:, src(i, H), src(j, W)]
    return out


def conv2d_valid(x, w, stride):
    """y[b,i,j,o] = sum_{p,q,c} x[b, i*s+p, j*s+q, c] * w[p,q,c,o]   (cross-correlation, HWIO)."""
    B, H, W, C = x.shape
    k = w.shape[0]
    Ho = (H - k) // stride + 1
    Wo = (W - k) // stride + 1
    y = np.zeros((B, Ho, Wo, w.shape[3]), np.float64)
    for i in range(Ho):
        for j in range(Wo):
            patch = x[:, i * stride:i * stride + k, j * stride:j * stride + k, :]
            y[:, i, j] = np.tensordot(patch, w, axes=([1, 2, 3], [0, 1, 2]))
    return y


def same_padding(n_in, k, s):
    """TF 'SAME': out = ceil(n/s); total = max((out-1)*s + k - n, 0); low = total//2."""
    out = -(-n_in // s)
    tot = max((out - 1) * s + k - n_in, 0)
    return out, tot // 2, tot - tot // 2


def conv2d_transpose_same(x, w, stride):
    """tf.nn.conv2d_transpose(x, w[k,k,Cout,Cin], [B, s*H, s*W, Cout], s, 'SAME').

    Definition: it is dL/d(input) of  y = conv2d(input[B,sH,sW,Cout], w, s, 'SAME')  evaluated with
    dL/dy = x.  The SAME forward reads input[a*s + p - lo] for output position a and tap p, so the
    gradient scatters  out[a*s + p - lo] += x[a] * w[p]  (out-of-range targets dropped)."""
    B, H, W, Cin = x.shape
    k = w.shape[0]
    Cout = w.shape[2]
    So = stride * H
    _, lo, _ = same_padding(So, k, stride)
    out = np.zeros((B, So, stride * W, Cout), np.float64)
    for a in range(H):
        for b_ in range(W):
            for p in range(k):
                for q in range(k):
                    i = a * stride + p - lo
                    j = b_ * stride + q - lo
                    if 0 <= i < So and 0 <= j < stride * W:
                        out[:, i, j] += x[:, a, b_] @ w[p, q].T     # [B,Cin] @ [Cin,Cout]
    return out
