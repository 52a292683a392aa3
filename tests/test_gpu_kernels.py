"""GPU parity tests: every HIP kernel family, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Tolerances are stated per test: bit-exact for DiffAugment's integer
indexing, fp32 accumulation-order noise (<= 2e-5 relative L2) for the MFMA GEMM families."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import kat
from oracle import ref_ops as R
from tests.common import rel_err, t2n

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _fn():
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import functional as Fn
    return Fn


def _hip():
    from biggan_tensorflow_amd import hip
    return hip


def cu(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    if grad:
        t.requires_grad_(True)
    return t


def test_library_loaded_and_arch():
    hip = _hip()
    L = hip.lib()
    assert L.bg_abi_version() == hip.ABI_VERSION
    assert L.bg_target_arch() == b"gfx950"
    with pytest.raises(RuntimeError):
        hip.f32(torch.zeros(4))          # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------
# conv (ops.py:49-113)
# ------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, Cin, Cout, k, s, pad, bias
    (2, 8, 8, 16, 3, 1, 1, False),
    (2, 8, 8, 16, 3, 2, 1, False),
    (3, 16, 3, 8, 3, 2, 1, False),      # image layer: Cin = 3 (scalar loads)
    (2, 16, 16, 3, 3, 1, 1, False),     # G_logit: Cout = 3
    (2, 8, 12, 6, 1, 1, 0, True),       # SA 1x1 conv with bias, C % 4 != 0 on the output
    (1, 4, 32, 32, 3, 1, 1, True),      # 4x4 map: every pixel touches the reflect border
    (5, 12, 20, 72, 3, 2, 1, True),     # ragged tiles in M, N and K
    (2, 32, 64, 128, 3, 1, 1, False),   # 128x128 tile path
    (2, 16, 64, 3, 3, 1, 1, True),      # RGB head (direct kernels), with bias
    (1, 8, 96, 3, 3, 1, 1, False),      # RGB head at ch=96: channel chunks of 64 + 32
    (3, 4, 32, 1, 3, 1, 1, False),      # single output channel, 4x4 map (all-border reflect)
    (2, 64, 16, 2, 1, 1, 0, False),     # 1x1, two output channels
    (64, 4, 1024, 1024, 3, 1, 1, False),  # small output, long K: split-K slabs
    (3, 16, 3, 64, 3, 2, 1, False),     # image layer (3 input channels): stride 2, reflect
    (2, 12, 3, 96, 3, 2, 1, True),      # ch = 96: 24 threads per pixel, bias
    (2, 10, 3, 32, 3, 1, 1, False),     # stride 1: both mirrored borders
    (2, 8, 4, 16, 3, 2, 1, False),      # 4 input channels
]


@pytest.mark.parametrize("N,H,Cin,Cout,k,s,pad,use_bias", CONV_CASES)
def test_conv2d_fwd_bwd(N, H, Cin, Cout, k, s, pad, use_bias):
    Fn, hip = _fn(), _hip()
    rng = np.random.default_rng(N * 1000 + H + Cin + Cout + k + s)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) * 0.2
    b = rng.standard_normal(Cout) if use_bias else None
    # oracle (float64)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    xin = xt.permute(0, 3, 1, 2)
    if pad:
        xin = F.pad(xin, (pad, pad, pad, pad), mode="reflect")
    yr = F.conv2d(xin.contiguous(), wt.permute(3, 2, 0, 1).contiguous(), stride=s).permute(0, 2, 3, 1)
    bt = None
    if use_bias:
        bt = torch.tensor(b, requires_grad=True)
        yr = yr + bt
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    Ho = yr.shape[1]
    # HIP
    xc, wc = cu(x, True), cu(w, True)
    bc = cu(b, True) if use_bias else None
    y = Fn.Conv2dFn.apply(xc, wc, bc, s, pad, Ho, Ho, hip.PAD_REFLECT)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < TOL
    if use_bias:
        assert rel_err(t2n(bc.grad), bt.grad.numpy()) < TOL


def test_conv_matches_direct_loop_kat():
    """Alignment KAT against the direct-loop TF semantics (reflect 1+1, k3 s2, even H)."""
    Fn, hip = _fn(), _hip()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 8, 5))
    w = rng.standard_normal((3, 3, 5, 7))
    ref = kat.conv2d_valid(kat.reflect_pad(x, 1, 1), w, 2)
    y = Fn.Conv2dFn.apply(cu(x), cu(w), None, 2, 1, 4, 4, hip.PAD_REFLECT)
    assert rel_err(t2n(y), ref) < TOL


def test_conv_zero_padding_same():
    Fn, hip = _fn(), _hip()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 9, 9, 8))
    w = rng.standard_normal((3, 3, 8, 8))
    ref = kat.conv2d_valid(np.pad(x, [(0, 0), (1, 1), (1, 1), (0, 0)]), w, 1)
    xc, wc = cu(x, True), cu(w, True)
    y = Fn.Conv2dFn.apply(xc, wc, None, 1, 1, 9, 9, hip.PAD_ZERO)
    assert rel_err(t2n(y), ref) < TOL
    g = rng.standard_normal(ref.shape)
    y.backward(cu(g))
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yr = F.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    yr.backward(torch.tensor(g))
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < TOL


@pytest.mark.parametrize("H,s", [(8, 2), (12, 2), (8, 1), (9, 1)])      # (every BigGAN map is a multiple of its stride)
def test_thin_input_conv_zero_padding_tf_same(H, s):
    """Image layer with --conv_padding zero: TF 'SAME' (pad_lo = total // 2, asymmetric for even H at
    stride 2) through the implicit-GEMM kernels with a 3-channel input."""
    Fn, hip = _fn(), _hip()
    rng = np.random.default_rng(H + s)
    x = rng.standard_normal((2, H, H, 3))
    w = rng.standard_normal((3, 3, 3, 32))
    out = -(-H // s)
    tot = max((out - 1) * s + 3 - H, 0)
    lo, hi = tot // 2, tot - tot // 2
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yr = F.conv2d(F.pad(xt.permute(0, 3, 1, 2), (lo, hi, lo, hi)), wt.permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc, wc = cu(x, True), cu(w, True)
    y = Fn.Conv2dFn.apply(xc, wc, None, s, lo, out, out, hip.PAD_ZERO)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < TOL


# ------------------------------------------------------------------------------------------
# transposed conv (ops.py:116-139)
# ------------------------------------------------------------------------------------------
DECONV_CASES = [
    # N, H, Cin, Cout, k, s
    (2, 4, 16, 8, 4, 2),
    (2, 4, 16, 8, 3, 1),
    (3, 8, 24, 40, 4, 2),     # ragged tiles
    (1, 16, 8, 8, 3, 1),
    (2, 8, 6, 10, 4, 2),      # C % 4 != 0
    (2, 16, 128, 64, 4, 2),   # big-tile path
    (2, 16, 64, 128, 3, 1),
]


@pytest.mark.parametrize("N,H,Cin,Cout,k,s", DECONV_CASES)
def test_deconv2d_fwd_bwd(N, H, Cin, Cout, k, s):
    Fn = _fn()
    rng = np.random.default_rng(N * 100 + H + Cin + Cout + k)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cout, Cin)) * 0.2
    b = rng.standard_normal(Cout)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    yr = F.conv_transpose2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), stride=s, padding=1).permute(0, 2, 3, 1) + bt
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc, wc, bc = cu(x, True), cu(w, True), cu(b, True)
    y = Fn.Deconv2dFn.apply(xc, wc, bc, s, 1, None)
    y.backward(cu(g))
    assert tuple(y.shape) == (N, s * H, s * H, Cout)
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < TOL
    assert rel_err(t2n(bc.grad), bt.grad.numpy()) < TOL


@pytest.mark.parametrize("k,s", [(4, 2), (3, 1), (3, 2), (6, 2), (4, 1)])
def test_deconv_matches_tf_gradient_definition(k, s):
    """KAT: tf.nn.conv2d_transpose(SAME) as the input-gradient of a SAME conv (no flip, low crop 1)."""
    Fn = _fn()
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 5, 5, 4))
    w = rng.standard_normal((k, k, 3, 4))
    ref = kat.conv2d_transpose_same(x, w, s)
    pad_lo = max((5 - 1) * s + k - s * 5, 0) // 2                  # ops.py:128 'SAME': 1 for (4,2),(3,1); 0, 2, 1 for the rest
    xc, wc = cu(x, True), cu(w, True)
    y = Fn.Deconv2dFn.apply(xc, wc, None, s, pad_lo, None)
    assert rel_err(t2n(y), ref) < TOL
    # gradients of the non-default (asymmetric) alignments too: adjoint identities against the KAT forward
    g = rng.standard_normal(ref.shape)
    y.backward(cu(g))
    eps_x, eps_w = rng.standard_normal(x.shape), rng.standard_normal(w.shape)
    lhs_x = float((t2n(xc.grad) * eps_x).sum())
    rhs_x = float((kat.conv2d_transpose_same(eps_x, w, s) * g).sum())
    lhs_w = float((t2n(wc.grad) * eps_w).sum())
    rhs_w = float((kat.conv2d_transpose_same(x, eps_w, s) * g).sum())
    assert abs(lhs_x - rhs_x) < 1e-4 * max(abs(rhs_x), 1.0) and abs(lhs_w - rhs_w) < 1e-4 * max(abs(rhs_w), 1.0)


@pytest.mark.parametrize("stride", [1, 2])
def test_deconv_reproduces_tensorflow_unit_test_closed_forms(stride):
    """The all-ones cases of TensorFlow's conv2d_transpose_test.py (testConv2DTransposeSingleStride: 12 / 18 / 27;
    testConv2DTransposeSame: 3 / 6 / 12 by parity of the output index) through the HIP kernel - a published
    known answer for the SAME transposed-conv alignment, including the asymmetric k3 s2 case (pad_lo 0)."""
    Fn = _fn()
    x = torch.ones(2, 6, 4, 3, device="cuda")
    f = torch.ones(3, 3, 2, 3, device="cuda")
    tot = max((6 - 1) * stride + 3 - stride * 6, 0)
    y = t2n(Fn.Deconv2dFn.apply(x, f, None, stride, tot // 2, None))

    def counts(n):
        if stride == 1:
            return np.array([3.0 if 0 < i < n - 1 else 2.0 for i in range(n)])
        return np.array([2.0 if (i % 2 == 0 and 0 < i < n - 1) else 1.0 for i in range(n)])
    expect = 3.0 * np.outer(counts(6 * stride), counts(4 * stride))
    assert y.shape == (2, 6 * stride, 4 * stride, 2)
    assert np.array_equal(y[0, :, :, 0], expect) and np.array_equal(y[1, :, :, 1], expect)


def test_deconv_accumulate_into_epilogue():
    Fn = _fn()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 4, 4, 8))
    w = rng.standard_normal((3, 3, 8, 8))
    base = rng.standard_normal((2, 4, 4, 8))
    xc, wc, bc = cu(x, True), cu(w, True), cu(base, True)
    y = Fn.Deconv2dFn.apply(xc, wc, None, 1, 1, bc * 1.0)
    ref = kat.conv2d_transpose_same(x, w, 1) + base
    assert rel_err(t2n(y), ref) < TOL
    y.sum().backward()
    assert np.allclose(t2n(bc.grad), 1.0)


# ------------------------------------------------------------------------------------------
# dense / attention (ops.py:148-175, 481-485)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(16, 96, 184), (16, 184, 1024), (8, 32, 64), (64, 128, 1), (5, 7, 3)])
def test_dense_fwd_bwd(M, K, N):
    Fn = _fn()
    rng = np.random.default_rng(M + K + N)
    x, w, b = rng.standard_normal((M, K)), rng.standard_normal((K, N)) * 0.1, rng.standard_normal(N)
    xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
    yr = xt @ wt + bt
    g = rng.standard_normal((M, N))
    yr.backward(torch.tensor(g))
    xc, wc, bc = cu(x, True), cu(w, True), cu(b, True)
    y = Fn.DenseFn.apply(xc, wc, bc)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < TOL
    assert rel_err(t2n(bc.grad), bt.grad.numpy()) < TOL


@pytest.mark.parametrize("B,K,Ns", [(4, 32, (96, 96, 48, 48)), (37, 36, (1536, 1536, 768, 768)), (256, 32, (24, 200)),
                                    (3, 1032, (64,)), (9, 5, (7, 3, 300))])
def test_grouped_dense_projections_match_the_separate_ones(B, K, Ns):
    """functional.GroupedDenseFn (csrc/dense_group.hip): the beta / gamma projections of a generator block's conditional
    batch norms (ops.py:623-624) in one launch per direction - same values as one fully_connected each (float64
    reference), weight and bias gradients included, on column slices of a wider z, with and without bias, and
    accumulating into a gradient that already holds a contribution."""
    Fn = _fn()
    rng = np.random.default_rng(B + K + len(Ns))
    z = rng.standard_normal((B, K + 40))
    zc = cu(z)
    xs = zc[:, 8:8 + K]
    ws = [rng.standard_normal((K, n)) * 0.1 for n in Ns]
    bs = [rng.standard_normal(n) if i % 3 != 2 else None for i, n in enumerate(Ns)]
    gs = [rng.standard_normal((B, n)) for n in Ns]
    wc = [cu(w, True) for w in ws]
    bc = [None if b is None else cu(b, True) for b in bs]
    args = []
    for w_, b_ in zip(wc, bc):
        args += [xs, w_, b_]
    ys = Fn.GroupedDenseFn.apply(len(Ns), *args)
    torch.autograd.backward(list(ys), [cu(g) for g in gs])
    for i, n in enumerate(Ns):
        xr = z[:, 8:8 + K]
        assert rel_err(t2n(ys[i]), xr @ ws[i] + (0 if bs[i] is None else bs[i])) < TOL, i
        assert rel_err(t2n(wc[i].grad), xr.T @ gs[i]) < TOL, i
        if bs[i] is not None:
            assert rel_err(t2n(bc[i].grad), gs[i].sum(0)) < TOL, i
    # a second backward pass accumulates (the emit_grad protocol: first writer overwrites, later ones add)
    ys = Fn.GroupedDenseFn.apply(len(Ns), *args)
    torch.autograd.backward(list(ys), [cu(g) for g in gs])
    assert rel_err(t2n(wc[0].grad), 2 * (z[:, 8:8 + K].T @ gs[0])) < TOL


def test_dense_on_column_slice_without_copy():
    Fn = _fn()
    rng = np.random.default_rng(5)
    z = rng.standard_normal((8, 256))
    w = rng.standard_normal((32, 48)) * 0.1
    zc = cu(z)
    y = Fn.DenseFn.apply(zc[:, 96:128], cu(w), None)
    assert rel_err(t2n(y), z[:, 96:128] @ w) < TOL


@pytest.mark.parametrize("B,N,Nk,dq,dv,fused", [
    (2, 256, 64, 2, 8, False),            # unsupported by the fused kernels: materialised form
    (3, 1024, 256, 8, 32, True), (3, 1024, 256, 8, 32, False),
    (1, 4096, 1024, 16, 64, True), (1, 4096, 1024, 16, 64, False),
    (2, 128, 128, 4, 16, True), (2, 512, 128, 12, 48, True), (1, 256, 256, 24, 96, True),
    (2, 256, 128, 32, 128, True), (1, 384, 128, 20, 72, True)])
def test_attention_fwd_bwd(B, N, Nk, dq, dv, fused):
    Fn = _fn()
    from biggan_tensorflow_amd import hip
    assert bool(hip.lib().bg_attention2_supported(N, Nk, dq, dv)) == (fused or (N, Nk) != (256, 64))
    Fn.AttentionFn.flash = fused
    try:
        _attention_case(Fn, B, N, Nk, dq, dv)
    finally:
        Fn.AttentionFn.flash = True


def _attention_case(Fn, B, N, Nk, dq, dv):
    rng = np.random.default_rng(N + dq)
    q, k, v = rng.standard_normal((B, N, dq)), rng.standard_normal((B, Nk, dq)), rng.standard_normal((B, Nk, dv))
    qt, kt, vt = (torch.tensor(a, requires_grad=True) for a in (q, k, v))
    o_ref = torch.softmax(qt @ kt.transpose(1, 2), -1) @ vt
    g = rng.standard_normal((B, N, dv))
    o_ref.backward(torch.tensor(g))
    qc, kc, vc = cu(q, True), cu(k, True), cu(v, True)
    o = Fn.AttentionFn.apply(qc, kc, vc)
    o.backward(cu(g))
    assert rel_err(t2n(o), o_ref.detach().numpy()) < TOL
    assert rel_err(t2n(qc.grad), qt.grad.numpy()) < 5e-5
    assert rel_err(t2n(kc.grad), kt.grad.numpy()) < 5e-5
    assert rel_err(t2n(vc.grad), vt.grad.numpy()) < TOL


# ------------------------------------------------------------------------------------------
# spectral norm (ops.py:718-747)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 3, 16, 32), (32, 64), (4, 4, 8, 24), (184, 1024), (3, 3, 8, 3), (16, 1)])
def test_spectral_norm_fwd_bwd(shape):
    Fn = _fn()
    rng = np.random.default_rng(sum(shape))
    w = rng.standard_normal(shape) * 0.1
    cols = shape[-1]
    u = rng.standard_normal((1, cols))
    W = torch.tensor(w).reshape(-1, cols)
    ut = torch.tensor(u)
    v_hat = R.l2_normalize(ut @ W.t())
    u_hat = R.l2_normalize(v_hat @ W)
    sigma = (v_hat @ W @ u_hat.t()).item()
    wc = cu(w, True)
    uc = cu(u)
    wn = Fn.SpectralNormFn.apply(wc, uc)
    assert rel_err(t2n(wn), w / sigma) < TOL
    assert rel_err(t2n(uc), u_hat.numpy()) < TOL            # u <- u_hat in place (ops.py:743)
    G = rng.standard_normal(shape)
    wn.backward(cu(G))
    Gm = torch.tensor(G).reshape(-1, cols)
    exp = (Gm - (Gm * (W / sigma)).sum() * (v_hat.t() @ u_hat)) / sigma
    assert rel_err(t2n(wc.grad).reshape(-1, cols), exp.numpy()) < 5e-5


def test_spectral_norm_batch_sharded_by_weight_matches_the_replicated_batch():
    """SnBatch(shard=...): two owners iterate their own weights, exchange their sigma | u | v_hat segments (what the
    all-gather does, done by hand here) and normalise everything - same w / sigma, u, v_hat, sigma as one batch."""
    Fn = _fn()
    shapes = [(3, 3, 16, 32), (32, 64), (4, 4, 8, 24), (184, 1024), (3, 3, 8, 3), (16, 1), (1, 1, 64, 8), (20, 2050)]
    rng = np.random.default_rng(78)
    ws = [rng.standard_normal(s) * 0.1 for s in shapes]
    us = [rng.standard_normal((1, s[-1])) for s in shapes]
    plain = Fn.SnBatch([(cu(w, True), cu(u)) for w, u in zip(ws, us)])
    ref = plain.forward()
    ranks = [Fn.SnBatch([(cu(w, True), cu(u)) for w, u in zip(ws, us)], None, (r, 2, None)) for r in (0, 1)]
    assert ranks[0].owner == ranks[1].owner and set(ranks[0].owner) == {0, 1}
    assert ranks[0].n_own + ranks[1].n_own == len(shapes)
    for sb in ranks:
        sb.power_owned()
    seg = ranks[0].seg
    for r in (0, 1):                                       # rank r's segment to the other rank
        ranks[1 - r].state_flat[r * seg:(r + 1) * seg].copy_(ranks[r].state_flat[r * seg:(r + 1) * seg])
    for sb in ranks:
        sb.normalize_all()
        for i in range(len(shapes)):
            assert rel_err(t2n(sb.wn[i]), t2n(ref[i])) < 1e-6, shapes[i]
            assert rel_err(t2n(sb.u[i]), t2n(plain.u[i])) < 1e-6, shapes[i]
            assert rel_err(t2n(sb.v[i]), t2n(plain.v[i])) < 1e-6, shapes[i]
            assert abs(sb.wn[i].bg_sigma.item() - plain.sigma[i].item()) < 1e-6 * abs(plain.sigma[i].item())
    assert torch.equal(ranks[0].state_flat, ranks[1].state_flat)


def test_spectral_norm_multi_tensor_batch():
    """functional.SnBatch: all weights of a network in one call; bwd with skip / overwrite / accumulate."""
    Fn = _fn()
    shapes = [(3, 3, 16, 32), (32, 64), (4, 4, 8, 24), (184, 1024), (3, 3, 8, 3), (16, 1), (1, 1, 64, 8),
              (3, 3, 256, 256), (20, 2050)]
    rng = np.random.default_rng(77)
    ws = [rng.standard_normal(s) * 0.1 for s in shapes]
    us = [rng.standard_normal((1, s[-1])) for s in shapes]
    wc = [cu(w, True) for w in ws]
    uc = [cu(u) for u in us]
    batch = Fn.SnBatch(list(zip(wc, uc)))
    wns = batch.forward()
    Gs = [rng.standard_normal(s) for s in shapes]
    prior = [rng.standard_normal(s) for s in shapes]
    exp_dw = []
    for i, (w, u) in enumerate(zip(ws, us)):
        cols = w.shape[-1]
        W = torch.tensor(w).reshape(-1, cols)
        v_hat = R.l2_normalize(torch.tensor(u) @ W.t())
        u_hat = R.l2_normalize(v_hat @ W)
        sigma = (v_hat @ W @ u_hat.t()).item()
        assert rel_err(t2n(wns[i]), w / sigma) < TOL, shapes[i]
        assert rel_err(t2n(uc[i]), u_hat.numpy()) < TOL, shapes[i]
        assert rel_err(t2n(batch.v[i]), v_hat.numpy().ravel()) < TOL, shapes[i]
        assert abs(batch.sigma[i].item() - sigma) < 1e-5 * abs(sigma)
        Gm = torch.tensor(Gs[i]).reshape(-1, cols)
        exp_dw.append(((Gm - (Gm * (W / sigma)).sum() * (v_hat.t() @ u_hat)) / sigma).numpy().reshape(w.shape))
    # item 1: no gradient arrives (skipped, slot untouched); item 2: slot already holds a gradient (accumulate)
    for i, wn in enumerate(wns):
        assert wn.requires_grad
        wc[i].bg_grad.copy_(cu(prior[i]))
        if i == 1:
            continue
        Fn.emit_grad(wn, lambda out, i=i: out.copy_(cu(Gs[i])))
        wc[i].bg_touched = (i == 2)
    batch.backward()
    for i in range(len(shapes)):
        got = t2n(wc[i].bg_grad)
        if i == 1:
            assert np.array_equal(got, prior[i].astype(np.float32)) and not wc[i].bg_touched
        elif i == 2:
            assert rel_err(got, exp_dw[i] + prior[i]) < 5e-5
        else:
            assert rel_err(got, exp_dw[i]) < 5e-5, shapes[i]
            assert wc[i].bg_touched
    # second forward continues the power iteration from the updated u
    wns = batch.forward()
    W = torch.tensor(ws[3]).reshape(-1, shapes[3][-1])
    u1 = torch.tensor(t2n(uc[3]).astype(np.float64))
    v_hat = R.l2_normalize(torch.tensor(us[3]) @ W.t())
    u_hat = R.l2_normalize(v_hat @ W)
    v2 = R.l2_normalize(u_hat @ W.t())
    u2 = R.l2_normalize(v2 @ W)
    assert rel_err(t2n(uc[3]), u2.numpy()) < TOL and u1 is not None


# ------------------------------------------------------------------------------------------
# batch norm + PReLU (ops.py:532-537, 580-585, 611-643)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,C,per_sample,with_alpha", [
    (4, 8, 16, True, True), (3, 4, 40, True, False), (2, 16, 8, False, True), (5, 2, 1024, True, True),
    (2, 4, 6, False, False)])
def test_bn_act_fwd_bwd(N, H, C, per_sample, with_alpha):
    Fn = _fn()
    rng = np.random.default_rng(N + H + C)
    x = rng.standard_normal((N, H, H, C)) * 1.5 + 0.3
    gshape = (N, C) if per_sample else (C,)
    gamma, beta = rng.standard_normal(gshape), rng.standard_normal(gshape)
    alpha = rng.uniform(0.05, 0.4, C) if with_alpha else None
    mm0, mv0 = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    xt, gt, bt = (torch.tensor(a, requires_grad=True) for a in (x, gamma, beta))
    at = torch.tensor(alpha, requires_grad=True) if with_alpha else None
    mean = xt.mean(dim=(0, 1, 2))
    var = ((xt - mean) ** 2).mean(dim=(0, 1, 2))
    gb = gt.reshape(-1, 1, 1, C) if per_sample else gt
    bb = bt.reshape(-1, 1, 1, C) if per_sample else bt
    inv = torch.rsqrt(var + 1e-5) * gb
    pre = xt * inv + (bb - mean * inv)
    yr = torch.relu(pre) + at * (pre - pre.abs()) * 0.5 if with_alpha else pre
    g = rng.standard_normal(x.shape)
    yr.backward(torch.tensor(g))
    xc, gc, bc = cu(x, True), cu(gamma, True), cu(beta, True)
    ac = cu(alpha, True) if with_alpha else None
    mm, mv = cu(mm0), cu(mv0)
    y = Fn.BnActFn.apply(xc, gc, bc, ac, mm, mv, 0.98, 1e-5, not per_sample, True, None, 1)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < 1e-4
    assert rel_err(t2n(gc.grad), gt.grad.numpy()) < 5e-5
    assert rel_err(t2n(bc.grad), bt.grad.numpy()) < 5e-5
    if with_alpha:
        assert rel_err(t2n(ac.grad), at.grad.numpy()) < 5e-5
    n = N * H * H
    vfac = n / (n - 1) if not per_sample else 1.0          # tf.layers BN: Bessel-corrected moving variance
    assert rel_err(t2n(mm), mm0 * 0.98 + mean.detach().numpy() * 0.02) < TOL
    assert rel_err(t2n(mv), mv0 * 0.98 + var.detach().numpy() * vfac * 0.02) < TOL


@pytest.mark.parametrize("N,H,C,per_sample,scale_is_var,weight", [
    (4, 8, 16, True, 1, 0.7), (3, 4, 40, True, 1, None), (2, 8, 8, False, 0, None), (5, 2, 1024, True, 1, 0.0)])
def test_bn_renorm_fwd_bwd(N, H, C, per_sample, scale_is_var, weight):
    """bg_renorm_coeffs / bg_renorm_affine_* inside BnActFn (ops.py:600-609, 645-715) against a float64 torch
    restatement: clipped and unclipped r / d, stop-gradient, running-statistics and fade-in updates."""
    Fn = _fn()
    rng = np.random.default_rng(N + H + C)
    x = rng.standard_normal((N, H, H, C)) * 1.5 + 0.3
    gshape = (N, C) if per_sample else (C,)
    gamma, beta = rng.standard_normal(gshape), rng.standard_normal(gshape)
    alpha = rng.uniform(0.05, 0.4, C)
    mm0, mv0 = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    rm0 = rng.normal(0.3, 0.6, C)
    rs0 = rng.uniform(0.3, 6.0, C) if scale_is_var else rng.uniform(0.7, 3.0, C)      # batch sigma is about 1.5
    if not scale_is_var:
        rs0[0] = 1e-4                                        # below sqrt(eps): the Keras floor
    rmin, rmax, dmax, decay, fade, eps = 1 / 1.5, 1.5, 0.5, 0.9, 0.9999, 1e-5
    xt, gt, bt, at = (torch.tensor(a, requires_grad=True) for a in (x, gamma, beta, alpha))
    mean = xt.mean(dim=(0, 1, 2))
    var = ((xt - mean) ** 2).mean(dim=(0, 1, 2))
    sigma = torch.sqrt(var + eps)
    w = 1.0 if weight is None else weight
    sref = torch.sqrt(torch.tensor(rs0) + eps) if scale_is_var else torch.clamp(torch.tensor(rs0), min=eps ** 0.5)
    sw = w * sref + (1 - w) * sigma
    mw = w * torch.tensor(rm0) + (1 - w) * mean
    r = torch.clamp(sigma / sw, rmin, rmax).detach()
    d = torch.clamp((mean - mw) / sw, -dmax, dmax).detach()
    if w > 0:
        assert 0 < int((r == rmax).sum() + (r == rmin).sum()) < C and 0 < int((d.abs() == dmax).sum()) < C
    gb = gt.reshape(-1, 1, 1, C) if per_sample else gt
    bb = bt.reshape(-1, 1, 1, C) if per_sample else bt
    inv = torch.rsqrt(var + eps) * (r * gb)
    pre = xt * inv + ((bb + d * gb) - mean * inv)
    yr = torch.relu(pre) + at * (pre - pre.abs()) * 0.5
    g = rng.standard_normal(x.shape)
    yr.backward(torch.tensor(g))
    xc, gc, bc, ac = cu(x, True), cu(gamma, True), cu(beta, True), cu(alpha, True)
    mm, mv, rm, rs = cu(mm0), cu(mv0), cu(rm0), cu(rs0)
    wt = None if weight is None else torch.full((), weight, dtype=torch.float32, device="cuda")
    renorm = dict(ref_mean=rm, ref_scale=rs, scale_is_var=scale_is_var, weight=wt, update=1, rmin=rmin, rmax=rmax,
                  dmax=dmax, decay=decay, fadein_decay=fade)
    y = Fn.BnActFn.apply(xc, gc, bc, ac, mm, mv, 0.98, eps, False, True, None, 1, renorm)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < 1e-4
    assert rel_err(t2n(gc.grad), gt.grad.numpy()) < 5e-5
    assert rel_err(t2n(bc.grad), bt.grad.numpy()) < 5e-5
    assert rel_err(t2n(ac.grad), at.grad.numpy()) < 5e-5
    assert rel_err(t2n(mm), mm0 * 0.98 + mean.detach().numpy() * 0.02) < TOL
    assert rel_err(t2n(mv), mv0 * 0.98 + var.detach().numpy() * 0.02) < TOL
    assert rel_err(t2n(rm), rm0 * decay + mean.detach().numpy() * (1 - decay)) < TOL
    moved = var.detach().numpy() if scale_is_var else sigma.detach().numpy()
    assert rel_err(t2n(rs), rs0 * decay + moved * (1 - decay)) < TOL
    if weight is not None:
        assert abs(float(wt) - (weight * fade + (1 - fade))) < 1e-6
    # inference: population statistics, no corrections, nothing moves
    before = [t.clone() for t in (mm, mv, rm, rs)]
    yi = Fn.BnActFn.apply(xc.detach(), gc.detach(), bc.detach(), None, mm, mv, 0.98, eps, False, False, None, 1, renorm)
    gi = gamma.reshape(-1, 1, 1, C) if per_sample else gamma
    bi = beta.reshape(-1, 1, 1, C) if per_sample else beta
    ref_i = (x - t2n(mm).astype(np.float64)) / np.sqrt(t2n(mv).astype(np.float64) + eps) * gi + bi
    assert rel_err(t2n(yi), ref_i) < TOL
    assert all(torch.equal(a, b) for a, b in zip(before, (mm, mv, rm, rs)))


# ---------------------------------------------------------------- second-order pieces of the gradient penalty
def test_prelu_tangent_fn():
    """ydot = xdot * prelu'(x): value and gradients w.r.t. xdot and alpha (BigGAN.py:731: tf.gradients through PReLU)."""
    Fn = _fn()
    rng = np.random.default_rng(3)
    x, xd, g = (rng.standard_normal((3, 5, 5, 8)) for _ in range(3))
    alpha = rng.uniform(0.05, 0.4, 8)
    xt, xdt, at = torch.tensor(x), torch.tensor(xd, requires_grad=True), torch.tensor(alpha, requires_grad=True)
    slope = torch.where(xt > 0, torch.ones_like(xt), at.expand_as(xt))
    yr = xdt * slope
    (yr * torch.tensor(g)).sum().backward()
    xdc, ac = cu(xd, True), cu(alpha, True)
    y = Fn.PReluTangentFn.apply(xdc, cu(x), ac)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xdc.grad), xdt.grad.numpy()) < TOL
    assert rel_err(t2n(ac.grad), at.grad.numpy()) < 5e-5


def test_maxpool_tangent_fn():
    Fn = _fn()
    rng = np.random.default_rng(4)
    x, xd = rng.standard_normal((2, 6, 8, 5)), rng.standard_normal((2, 6, 8, 5))
    g = rng.standard_normal((2, 3, 4, 5))
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    _, idx = F.max_pool2d(xt, 2, 2, return_indices=True)
    xdt = torch.tensor(xd, requires_grad=True)
    yr = xdt.permute(0, 3, 1, 2).reshape(2, 5, -1).gather(2, idx.reshape(2, 5, -1)).reshape(2, 5, 3, 4).permute(0, 2, 3, 1)
    (yr * torch.tensor(g)).sum().backward()
    xdc = cu(xd, True)
    y = Fn.MaxPool2TangentFn.apply(xdc, cu(x))
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(xdc.grad), xdt.grad.numpy()) < TOL


@pytest.mark.parametrize("trans_b", [True, False])
def test_bmm_fn(trans_b):
    Fn = _fn()
    rng = np.random.default_rng(5)
    a = rng.standard_normal((3, 40, 12))
    b = rng.standard_normal((3, 24, 12) if trans_b else (3, 12, 24))
    g = rng.standard_normal((3, 40, 24))
    at, bt = torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)
    yr = at @ (bt.transpose(1, 2) if trans_b else bt)
    (yr * torch.tensor(g)).sum().backward()
    ac, bc = cu(a, True), cu(b, True)
    y = Fn.BmmFn.apply(ac, bc, trans_b)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL
    assert rel_err(t2n(ac.grad), at.grad.numpy()) < TOL
    assert rel_err(t2n(bc.grad), bt.grad.numpy()) < TOL


@pytest.mark.parametrize("cols", [48, 256, 1024])
def test_softmax_and_softmax_tangent_fn(cols):
    """p = softmax(s); pdot = p * (sdot - <p, sdot>): gradients w.r.t. s (through p) and sdot."""
    Fn = _fn()
    rng = np.random.default_rng(cols)
    s_, sd, g, g2 = (rng.standard_normal((2, 7, cols)) for _ in range(4))
    st, sdt = torch.tensor(s_, requires_grad=True), torch.tensor(sd, requires_grad=True)
    pr = torch.softmax(st, dim=-1)
    pdr = pr * (sdt - (pr * sdt).sum(-1, keepdim=True))
    ((pdr * torch.tensor(g)).sum() + (pr * torch.tensor(g2)).sum()).backward()
    sc, sdc = cu(s_, True), cu(sd, True)
    p = Fn.SoftmaxFn.apply(sc)
    p1, p2 = Fn.ForkFn.apply(p, 2)
    pd = Fn.SoftmaxTangentFn.apply(p1, sdc)
    torch.autograd.backward([pd, p2], [cu(g), cu(g2)])
    assert rel_err(t2n(p), pr.detach().numpy()) < TOL
    assert rel_err(t2n(pd), pdr.detach().numpy()) < 5e-5
    assert rel_err(t2n(sdc.grad), sdt.grad.numpy()) < 5e-5
    assert rel_err(t2n(sc.grad), st.grad.numpy()) < 5e-5


@pytest.mark.parametrize("lp,dragan", [(0, False), (1, False), (0, True)])
def test_gp_interpolate_and_penalty(lp, dragan):
    """bg_gp_interpolate / bg_gp_penalty (BigGAN.py:718-740) against numpy float64."""
    from biggan_tensorflow_amd import hip
    L = hip.lib()
    rng = np.random.default_rng(7 + lp)
    B, per = 5, 3 * 16 * 16
    real, other = rng.uniform(-1, 1, (B, per)), rng.uniform(0, 1, (B, per))
    alpha = rng.uniform(0, 1, B)
    rc, oc, alc = cu(real), cu(other), cu(alpha)
    out = torch.empty_like(rc)
    if dragan:
        sums = torch.zeros(2, dtype=torch.float64, device="cuda")
        hip.check(L.bg_bn_stats(hip.f32(rc), hip.ptr(sums), B * per, 1, hip.stream()))
        hip.check(L.bg_gp_interpolate(hip.f32(rc), hip.f32(oc), hip.f32(alc), hip.ptr(sums), float(B * per), hip.f32(out),
                                      B, per, hip.stream()))
        ref = real + alpha[:, None] * (0.5 * real.std() * other)
    else:
        hip.check(L.bg_gp_interpolate(hip.f32(rc), hip.f32(oc), hip.f32(alc), None, 0.0, hip.f32(out), B, per,
                                      hip.stream()))
        ref = real + alpha[:, None] * (other - real)
    assert rel_err(t2n(out), ref) < TOL
    g = rng.standard_normal((B, per)) * np.array([0.001, 0.02, 0.03, 0.05, 0.0201])[:, None]    # norms around 1
    gc = cu(g)
    ws = torch.empty(2 * B, dtype=torch.float64, device="cuda")
    loss = torch.empty(1, dtype=torch.float32, device="cuda")
    v = torch.empty_like(gc)
    hip.check(L.bg_gp_penalty(hip.f32(gc), B, per, float(2 * B), 10.0, lp, hip.ptr(ws), hip.f32(loss), hip.f32(v),
                              hip.stream()))
    n = np.sqrt((g ** 2).sum(1))
    e = np.maximum(n - 1, 0) if lp else n - 1
    assert (n > 1).any() and (n < 1).any()
    assert abs(loss.item() - 10.0 * (e ** 2).sum() / (2 * B)) < 1e-5 * max(1.0, loss.item())
    assert rel_err(t2n(v), (10.0 * 2 * e / n / (2 * B))[:, None] * g) < 5e-5


@pytest.mark.parametrize("c", [8, 33, 96, 200])
def test_symmetrize(c):
    from biggan_tensorflow_amd import hip
    rng = np.random.default_rng(c)
    a = rng.standard_normal((c, c))
    ac = cu(a)
    out = torch.empty_like(ac)
    hip.check(hip.lib().bg_symmetrize(hip.f32(ac), hip.f32(out), c, hip.stream()))
    assert np.array_equal(t2n(out), (ac + ac.t()).cpu().numpy())


@pytest.mark.parametrize("shape", [(2, 8, 8, 3), (4, 4, 4, 64), (7, 33), (5, 128, 128, 3), (70001, 1)])
def test_prelu(shape):
    Fn = _fn()
    rng = np.random.default_rng(len(shape))
    x = rng.standard_normal(shape)
    a = rng.uniform(0.05, 0.3, shape[-1])
    xt, at = torch.tensor(x, requires_grad=True), torch.tensor(a, requires_grad=True)
    yr = torch.relu(xt) + at * (xt - xt.abs()) * 0.5
    g = rng.standard_normal(shape)
    yr.backward(torch.tensor(g))
    xc, ac = cu(x, True), cu(a, True)
    y = Fn.PReluFn.apply(xc, ac)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < 1e-6
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < 1e-6
    assert rel_err(t2n(ac.grad), at.grad.numpy()) < 5e-5


@pytest.mark.parametrize("cs", [1, 3, 4])
def test_pad_channels_pixel_forms_match_the_generic_kernel(cs):
    """bg_pad_channels: the one-pixel-per-thread kernels of the image layers (fp32 [pixels, C <= 4] -> bf16 [pixels, 8] in
    the zero-fill / hi | lo split / duplicate modes; fold of an fp32 [pixels, 8] result to C channels, fp32 or bf16) against
    the definition - and against the generic kernel, which a mis-aligned view still takes."""
    from biggan_tensorflow_amd import hip
    L = hip.lib()
    rng = np.random.default_rng(cs)
    pixels = 3 * 37 * 41
    x = torch.tensor(rng.standard_normal((pixels + 1, cs)), dtype=torch.float32, device="cuda")
    for mode in (hip.PAD_ZERO_FILL, hip.PAD_SPLIT, hip.PAD_DUP):
        if mode != hip.PAD_ZERO_FILL and 2 * cs > 8:
            continue
        xs = x[:pixels]
        ref = torch.zeros(pixels, 8, dtype=torch.float32, device="cuda")
        hi = xs.to(torch.bfloat16).float()
        if mode == hip.PAD_ZERO_FILL:
            ref[:, :cs] = xs
        elif mode == hip.PAD_SPLIT:
            ref[:, :cs] = hi
            ref[:, cs:2 * cs] = xs - hi
        else:
            ref[:, :cs] = xs
            ref[:, cs:2 * cs] = xs
        ref = ref.to(torch.bfloat16)
        got = torch.empty(pixels, 8, dtype=torch.bfloat16, device="cuda")
        hip.check(L.bg_pad_channels(hip.ptr(xs), hip.F32, hip.ptr(got), hip.BF16, pixels, cs, 8, 1, mode, hip.stream()))
        assert torch.equal(got, ref), mode
        if cs % 4:                              # a view that starts one row later is not 16-byte aligned: generic kernel
            xo = x[1:]
            got2 = torch.empty(pixels, 8, dtype=torch.bfloat16, device="cuda")
            hip.check(L.bg_pad_channels(hip.ptr(xo), hip.F32, hip.ptr(got2), hip.BF16, pixels, cs, 8, 1, mode, hip.stream()))
            got3 = torch.empty(pixels, 8, dtype=torch.bfloat16, device="cuda")
            xa = xo.clone()
            hip.check(L.bg_pad_channels(hip.ptr(xa), hip.F32, hip.ptr(got3), hip.BF16, pixels, cs, 8, 1, mode, hip.stream()))
            assert torch.equal(got2, got3), mode
    if 2 * cs <= 8:
        y8 = torch.tensor(rng.standard_normal((pixels, 8)), dtype=torch.float32, device="cuda")
        for dt_t, dt_h in ((torch.float32, hip.F32), (torch.bfloat16, hip.BF16)):
            out = torch.empty(pixels, cs, dtype=dt_t, device="cuda")
            hip.check(L.bg_pad_channels(hip.ptr(y8), hip.F32, hip.ptr(out), dt_h, pixels, 8, cs, 1, hip.PAD_FOLD, hip.stream()))
            assert torch.equal(out, (y8[:, :cs] + y8[:, cs:2 * cs]).to(dt_t))


@pytest.mark.parametrize("shape", [(2, 8, 8, 16), (3, 4, 6, 5), (1, 2, 2, 64)])
def test_avg_pool_and_nearest_upsample(shape):
    """ops.py:512-519: avg_pooling (2x2, stride 2) and up_sample (nearest x2), forward and backward."""
    Fn = _fn()
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape)
    xt = torch.tensor(x, requires_grad=True)
    yr = F.avg_pool2d(xt.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc = cu(x, True)
    y = Fn.AvgPool2Fn.apply(xc)
    y.backward(cu(g))
    assert rel_err(t2n(y), yr.detach().numpy()) < TOL and rel_err(t2n(xc.grad), xt.grad.numpy()) < TOL
    xt2 = torch.tensor(x, requires_grad=True)
    ur = xt2.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    g2 = rng.standard_normal(tuple(ur.shape))
    ur.backward(torch.tensor(g2))
    xc2 = cu(x, True)
    u = Fn.UpSample2Fn.apply(xc2)
    u.backward(cu(g2))
    assert np.array_equal(t2n(u), ur.detach().numpy().astype(np.float32))
    assert rel_err(t2n(xc2.grad), xt2.grad.numpy()) < TOL


def test_pooling_tanh_scaleadd():
    Fn = _fn()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 8, 8, 12))
    xt = torch.tensor(x, requires_grad=True)
    yr = F.max_pool2d(xt.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc = cu(x, True)
    y = Fn.MaxPool2Fn.apply(xc)
    y.backward(cu(g))
    assert np.array_equal(t2n(y), yr.detach().numpy().astype(np.float32))
    assert np.array_equal(t2n(xc.grad), xt.grad.numpy().astype(np.float32))
    # global sum pool
    xc2 = cu(x, True)
    s = Fn.SumPoolFn.apply(xc2)
    assert rel_err(t2n(s), x.sum((1, 2))) < 1e-6
    s.backward(cu(np.ones((2, 12))))
    assert np.allclose(t2n(xc2.grad), 1.0)
    # tanh
    xc3 = cu(x, True)
    t = Fn.TanhFn.apply(xc3)
    t.backward(cu(g_ := rng.standard_normal(x.shape)))
    assert rel_err(t2n(t), np.tanh(x)) < 1e-6
    assert rel_err(t2n(xc3.grad), g_ * (1 - np.tanh(x) ** 2)) < 1e-5
    # gamma * o + x
    o = rng.standard_normal(x.shape)
    oc, gc, xc4 = cu(o, True), cu([0.37], True), cu(x, True)
    r = Fn.ScaleAddFn.apply(oc, gc, xc4)
    r.backward(cu(g_))
    assert rel_err(t2n(r), 0.37 * o + x) < 1e-6
    assert rel_err(t2n(oc.grad), 0.37 * g_) < 1e-6
    assert rel_err(t2n(gc.grad), [(g_ * o).sum()]) < 5e-5
    assert rel_err(t2n(xc4.grad), g_) < 1e-7


# ------------------------------------------------------------------------------------------
# DiffAugment (DiffAugment_tf.py)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [8, 32, 64, 128, 256, 512])      # 256 / 512: BASELINE configs 4 and 5
def test_diffaugment_translation_cutout_bit_exact(S):
    from biggan_tensorflow_amd.DiffAugment import DiffAugment, draws_to_device
    rng = np.random.default_rng(S)
    B = 6
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    d = R.draw_diffaugment(rng, B, S)
    shift, cs, off_max = R.diffaugment_params(S)
    d["t_x"][0], d["t_y"][0] = -shift, shift            # extreme draws (edge cases)
    d["t_x"][1], d["t_y"][1] = shift, -shift
    d["o_x"][2], d["o_y"][2] = 0, off_max - 1
    d["o_x"][3], d["o_y"][3] = off_max - 1, 0
    lit = R.cutout_literal(R.translation_literal(x, d["t_x"], d["t_y"]), d["o_x"], d["o_y"])
    y = DiffAugment(cu(x), "translation,cutout", draws=draws_to_device(d, "cuda"))
    assert np.array_equal(t2n(y), lit)                   # bit-exact integer indexing
    y1 = DiffAugment(cu(x), "translation", draws=draws_to_device(d, "cuda"))
    assert np.array_equal(t2n(y1), R.translation_literal(x, d["t_x"], d["t_y"]))
    y2 = DiffAugment(cu(x), "cutout", draws=draws_to_device(d, "cuda"))
    assert np.array_equal(t2n(y2), R.cutout_literal(x, d["o_x"], d["o_y"]))


@pytest.mark.parametrize("policy", ["color,translation,cutout", "color", "color,cutout"])
def test_diffaugment_full_policy_fwd_bwd(policy):
    from biggan_tensorflow_amd.DiffAugment import DiffAugment, draws_to_device
    rng = np.random.default_rng(11)
    B, S = 4, 16
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    d = R.draw_diffaugment(rng, B, S)
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    yr = R.diffaugment(xt, d, policy)
    g = rng.standard_normal(x.shape)
    yr.backward(torch.tensor(g))
    xc = cu(x, True)
    y = DiffAugment(xc, policy, draws=draws_to_device(d, "cuda"))
    y.backward(cu(g))
    assert np.abs(t2n(y) - yr.detach().numpy()).max() < 2e-6          # colour ops: <= 1e-6-scale
    assert rel_err(t2n(xc.grad), xt.grad.numpy()) < 1e-5
    # zero pattern (translation fill + cutout box) must be identical
    assert np.array_equal(t2n(y) == 0, yr.detach().numpy() == 0)


def test_diffaugment_identity_and_bad_policy():
    from biggan_tensorflow_amd.DiffAugment import DiffAugment
    x = cu(np.zeros((1, 8, 8, 3)))
    assert DiffAugment(x, "") is x
    with pytest.raises(KeyError):
        DiffAugment(x, "bogus")


# ------------------------------------------------------------------------------------------
# losses, regulariser, optimiser
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scale", [1.0, -0.02, 3.0])
def test_hinge_losses_with_flood(scale):
    Fn = _fn()
    rng = np.random.default_rng(13)
    real, fake = rng.standard_normal((16, 1)) * scale, rng.standard_normal((16, 1)) * scale + 0.01
    rt, ft = torch.tensor(real, requires_grad=True), torch.tensor(fake, requires_grad=True)
    Ld = R.discriminator_loss("hinge", rt, ft, 0.1)
    Ld.backward()
    rc, fc = cu(real, True), cu(fake, True)
    L = Fn.HingeDLossFn.apply(rc, fc, 0.1, None, 1)
    L.backward()
    assert abs(L.item() - Ld.item()) < 1e-6
    assert rel_err(t2n(rc.grad), rt.grad.numpy()) < 1e-6
    assert rel_err(t2n(fc.grad), ft.grad.numpy()) < 1e-6
    ft2 = torch.tensor(fake, requires_grad=True)
    Lg = R.generator_loss("hinge", ft2, None, 0.05)
    Lg.backward()
    fc2 = cu(fake, True)
    L2 = Fn.HingeGLossFn.apply(fc2, 0.05, None, 1)
    L2.backward()
    assert abs(L2.item() - Lg.item()) < 1e-6
    assert rel_err(t2n(fc2.grad), ft2.grad.numpy()) < 1e-6     # includes the flood sign flip


@pytest.mark.parametrize("kind", ["hinge", "lsgan", "gan", "ra-lsgan", "ra-gan", "ra-hinge"])
@pytest.mark.parametrize("flood", [0.0, 0.3, 5.0])
def test_general_gan_losses(kind, flood):
    """ops.py:753-840 through bg_gan_loss_*: values and gradients of D and G losses, with the flood sign flip."""
    Fn = _fn()
    rng = np.random.default_rng(len(kind))
    real, fake = rng.standard_normal((7, 1)) * 1.5 + 0.2, rng.standard_normal((5, 1)) * 1.5 - 0.1
    for gen in (0, 1):
        rt, ft = torch.tensor(real, requires_grad=True), torch.tensor(fake, requires_grad=True)
        ref = (R.generator_loss(kind, ft, rt, flood) if gen else R.discriminator_loss(kind, rt, ft, flood))
        ref.backward()
        rc, fc = cu(real, True), cu(fake, True)
        use_real = (not gen) or kind.startswith("ra-")
        loss = Fn.GanLossFn.apply(rc if use_real else None, fc, Fn.GAN_LOSS_KINDS[kind], gen, flood, None, 1)
        loss.backward()
        assert abs(loss.item() - ref.item()) <= 2e-6 * max(abs(ref.item()), 1.0), (kind, gen, loss.item(), ref.item())
        assert rel_err(t2n(fc.grad), ft.grad.numpy()) < 2e-5, (kind, gen)
        if use_real and rt.grad is not None:
            assert rel_err(t2n(rc.grad), rt.grad.numpy()) < 2e-5, (kind, gen)


@pytest.mark.parametrize("shape", [(3, 3, 8, 16), (96, 184), (4, 4, 32, 8), (3, 3, 8, 3), (32, 320), (184, 1024),
                                   (7, 33)])
def test_ortho_cosine_regulariser(shape):
    Fn = _fn()
    rng = np.random.default_rng(sum(shape))
    w = rng.standard_normal(shape) * 0.05
    wt = torch.tensor(w, requires_grad=True)
    Lr = R.ortho_reg_loss(wt, 1e-4, "ortho_cosine")
    Lr.backward()
    wc = cu(w, True)
    L = Fn.OrthoCosineRegFn.apply(wc, 1e-4)
    L.backward()
    assert abs(L.item() - Lr.item()) < 2e-5 * abs(Lr.item())
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < 1e-4


@pytest.mark.parametrize("kind,shape", [("ortho", (3, 3, 8, 16)), ("ortho", (96, 184)), ("l2", (4, 4, 32, 8))])
def test_ortho_identity_and_l2_regularisers(kind, shape):
    """--g_regularization ortho (utils.py:199-200: l2_loss(W^T W - I)) and l2 (BigGAN.py:268-270)."""
    Fn = _fn()
    rng = np.random.default_rng(len(shape) + shape[-1])
    w = rng.standard_normal(shape) * 0.1
    wt = torch.tensor(w, requires_grad=True)
    ref = R.ortho_reg_loss(wt, 1e-2, "ortho") if kind == "ortho" else 1e-2 * 0.5 * (wt * wt).sum()
    ref.backward()
    wc = cu(w, True)
    loss = Fn.OrthoCosineRegFn.apply(wc, 1e-2, "ortho") if kind == "ortho" else Fn.L2RegFn.apply(wc, 1e-2)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_err(t2n(wc.grad), wt.grad.numpy()) < 5e-5


@pytest.mark.parametrize("n,b1,with_ema", [(10007, 0.0, True), (40000, 0.0, True), (40000, 0.5, True), (4096, 0.0, False)])
def test_adam_tf_ema_step(n, b1, with_ema):
    """bg_adam_tf_ema_step: the scalar kernel (ragged n) and the four-elements-per-thread kernel (n % 4 == 0; with
    beta1 = 0 the first moment is written without being read) against the documented TF update."""
    hip = _hip()
    rng = np.random.default_rng(17)
    p, g = rng.standard_normal(n), rng.standard_normal(n)
    m, v, ema = rng.standard_normal(n) * 0.1, rng.uniform(0, 1, n), rng.standard_normal(n)
    pc, gc, mc, vc, ec = cu(p), cu(g), cu(m), cu(v), cu(ema)
    lr_t = 2e-4 * np.sqrt(1 - 0.9 ** 3)
    hip.check(hip.lib().bg_adam_tf_ema_step(hip.f32(pc), hip.f32(gc), hip.f32(mc), hip.f32(vc), hip.f32(ec) if with_ema else None,
                                            lr_t, b1, 0.9, 1e-8, 0.999, 1.0, n, hip.stream()))
    m2 = b1 * m + (1.0 - b1) * g
    v2 = 0.9 * v + 0.1 * g * g
    p2 = p - lr_t * m2 / (np.sqrt(v2) + 1e-8)
    assert rel_err(t2n(pc), p2) < 1e-6
    assert rel_err(t2n(mc), m2) < 1e-6
    assert rel_err(t2n(vc), v2) < 1e-6
    if with_ema:
        assert rel_err(t2n(ec), 0.999 * ema + 0.001 * p2) < 1e-6
    else:
        assert np.array_equal(t2n(ec), ema.astype(np.float32))


# ------------------------------------------------------------------------------------------
# error behaviour of the C ABI: return codes + bg_last_error(), never a crash
# ------------------------------------------------------------------------------------------
def test_abi_rejects_bad_arguments_with_error_codes():
    Fn, hip = _fn(), _hip()
    L = hip.lib()
    x = torch.zeros(2, 8, 8, 16, device="cuda")
    w = torch.zeros(3, 3, 16, 32, device="cuda")
    y = torch.zeros(2, 8, 8, 32, device="cuda")
    good = hip.conv_desc(2, 8, 8, 16, 8, 8, 32, 3, 1, 1, hip.PAD_REFLECT)
    assert L.bg_conv2d_fwd(good, hip.f32(x), hip.f32(w), None, None, hip.f32(y), 0, None, 0, hip.stream()) == 0
    # null pointers, empty batch, zero channels, bad padding mode
    assert L.bg_conv2d_fwd(good, None, hip.f32(w), None, None, hip.f32(y), 0, None, 0, hip.stream()) == 1
    assert b"null tensor pointer" in L.bg_last_error()
    for bad in (hip.conv_desc(0, 8, 8, 16, 8, 8, 32, 3, 1, 1, hip.PAD_REFLECT),
                hip.conv_desc(2, 8, 8, 0, 8, 8, 32, 3, 1, 1, hip.PAD_REFLECT),
                hip.conv_desc(2, 8, 8, 16, 8, 8, 32, 3, 1, 1, 7)):
        assert L.bg_conv2d_fwd(bad, hip.f32(x), hip.f32(w), None, None, hip.f32(y), 0, None, 0, hip.stream()) != 0
        with pytest.raises(RuntimeError):
            hip.check(1)
    # attention: unsupported shape is refused by the fused entry point (the Function falls back instead)
    q = torch.zeros(1, 100, 8, device="cuda")
    o = torch.zeros(1, 100, 8, device="cuda")
    lse = torch.zeros(1, 100, device="cuda")
    assert not L.bg_attention2_supported(100, 100, 8, 8)
    assert L.bg_attention2_fwd(hip.f32(q), hip.f32(q), hip.f32(q), hip.f32(o), hip.f32(lse), 1, 100, 100, 8, 8,
                               hip.stream()) != 0
    assert b"unsupported shape" in L.bg_last_error()
    # spectral-norm batch: more than 256 items per call is refused
    assert L.bg_spectral_norm_batch_bwd(hip.f32(x), 300, None, None, hip.f32(x), 8 * 300, hip.stream()) != 0
    # a workspace that is too small
    d = hip.conv_desc(64, 4, 4, 1024, 4, 4, 1024, 3, 1, 1, hip.PAD_REFLECT)
    need = L.bg_conv2d_wgrad_workspace_bytes(d)
    assert need > 0
    xx = torch.zeros(64, 4, 4, 1024, device="cuda")
    dw = torch.zeros(3, 3, 1024, 1024, device="cuda")
    small = torch.zeros(4, device="cuda")
    rc = L.bg_conv2d_wgrad(d, hip.f32(xx), hip.f32(xx), hip.f32(dw), hip.f32(small), 16, hip.stream())
    torch.cuda.synchronize()
    assert rc == 0 or b"workspace" in L.bg_last_error()      # falls back to an unsplit launch or refuses; never overruns
