"""Size-independent properties at BASELINE config 2's FULL layer sizes (BigGAN-128, ch 64, batch 64),
where the float64 oracle is too slow to run: adjoint identities tying fwd / dgrad / wgrad of every
conv and transposed-conv family together, partition-of-unity and linearity of the fused attention,
fused vs materialised attention, and reproducibility / sanity of one full-size training iteration.
Everything goes through the C ABI; torch is used only to form the checking inner products (float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import functional as Fn, hip
    return Fn, hip


def _dot(a, b):
    return float((a.double() * b.double()).sum().item())


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g) * scale


# (kind, N, H, Cin, Cout, k, s): the layer shapes of G-128 / D-128 at ch = 64, batch 64 (D sees 2B = 128)
FULL_LAYERS = [
    ("deconv", 64, 4, 1024, 1024, 4, 2), ("deconv", 64, 8, 1024, 1024, 3, 1), ("deconv", 64, 32, 256, 128, 4, 2),
    ("deconv", 64, 128, 64, 64, 3, 1), ("deconv", 64, 64, 128, 64, 4, 2),
    ("conv", 128, 128, 3, 64, 3, 2), ("conv", 128, 64, 64, 64, 3, 1), ("conv", 128, 64, 64, 128, 3, 2),
    ("conv", 128, 8, 512, 1024, 3, 2), ("conv", 128, 4, 1024, 1024, 3, 1), ("conv", 64, 128, 64, 3, 3, 1),
    ("conv", 128, 64, 64, 8, 1, 1),
]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("kind,N,H,Cin,Cout,k,s", FULL_LAYERS)
def test_adjoint_identities_full_size(kind, N, H, Cin, Cout, k, s, mode):
    """<y, g> with y = op(x; w) must equal <x, dgrad(g; w)> and <w, wgrad(x, g)> (the op is bilinear):
    ties the three kernels of a layer together at the real sizes.  Tolerance 2e-4 relative to
    |y||g| in fp32 (fp32 accumulation over up to 9216-long reductions), 2e-2 in bf16-compute mode."""
    Fn, hip = _mods()
    L = hip.lib()
    Fn.set_precision("bf16-staged" if mode == "bf16" else "fp32")
    try:
        x = _rnd((N, H, H, Cin), 1).requires_grad_(True)
        if kind == "conv":
            w = _rnd((k, k, Cin, Cout), 2, 0.05).requires_grad_(True)
            Ho = H // s
            pad = 1 if k == 3 else 0
            y = Fn.Conv2dFn.apply(x, w, None, s, pad, Ho, Ho, hip.PAD_REFLECT)
        else:
            w = _rnd((k, k, Cout, Cin), 2, 0.05).requires_grad_(True)
            y = Fn.Deconv2dFn.apply(x, w, None, s, 1, None)
        g = _rnd(tuple(y.shape), 3)
        y.backward(g)
        lhs = _dot(y.detach(), g)
        scale = float(y.detach().double().norm() * g.double().norm())
        tol = (2e-2 if mode == "bf16" else 2e-4) * scale
        assert abs(lhs - _dot(x.detach(), x.grad)) <= tol, (lhs, _dot(x.detach(), x.grad), scale)
        assert abs(lhs - _dot(w.detach(), w.grad)) <= tol, (lhs, _dot(w.detach(), w.grad), scale)
        assert torch.isfinite(y).all()
    finally:
        Fn.set_precision("fp32")


def test_fused_attention_properties_full_size():
    """B=64, N=4096 queries, Nk=1024 keys, d=16, dv=64 (G's attention block at config 2): rows of the
    softmax sum to one (v = 1 -> o = 1), o is linear in v, the fused kernels agree with the materialised
    form (two GEMMs + softmax) in forward and in all three gradients."""
    Fn, hip = _mods()
    B, N, Nk, d, dv = 16, 4096, 1024, 16, 64           # 16 of the 64 samples: the materialised P is 1 GiB at 64
    q, k = _rnd((B, N, d), 1, 0.7), _rnd((B, Nk, d), 2, 0.7)
    v1, v2 = _rnd((B, Nk, dv), 3), _rnd((B, Nk, dv), 4)
    ones = torch.ones((B, Nk, dv), device="cuda")
    assert hip.lib().bg_attention2_supported(N, Nk, d, dv)
    o1 = Fn.AttentionFn.apply(q, k, ones)
    assert float((o1 - 1).abs().max()) < 1e-5        # 1024 fp32 probabilities per row
    oa, ob = Fn.AttentionFn.apply(q, k, v1), Fn.AttentionFn.apply(q, k, v2)
    oc = Fn.AttentionFn.apply(q, k, 0.5 * v1 - 2.0 * v2)
    lin = 0.5 * oa - 2.0 * ob
    assert float((oc - lin).double().norm() / lin.double().norm()) < 2e-6
    g = _rnd((B, N, dv), 5)
    outs = {}
    for fused in (True, False):
        Fn.AttentionFn.flash = fused
        try:
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v1))
            o = Fn.AttentionFn.apply(qq, kk, vv)
            o.backward(g)
            outs[fused] = (o.detach(), qq.grad, kk.grad, vv.grad)
        finally:
            Fn.AttentionFn.flash = True
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).double().norm() / b.double().norm()) < 2e-5


def test_full_size_iteration_is_reproducible_and_sane():
    """One D + G iteration of BASELINE config 2 (128^2, ch 64, batch 64): finite losses in the expected
    range for a random-init hinge GAN, forward pass bit-reproducible from identical state and inputs
    (fp64 accumulators in every forward reduction), every gradient finite, u / EMA / Adam state advance."""
    from tests.common import make_args
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import model, scope as S
    from biggan_tensorflow_amd.DiffAugment import draw
    args = make_args(img_size=128, ch=64, batch_size=64)
    gan = model.BigGAN(args, store=S.VariableStore("cuda", seed=42)).build_model()
    B = 64
    real = gan.synthetic_batch(B)
    z = gan.sample_z(B)
    dr, df = draw(B, 128, generator=gan.gen, device="cuda"), draw(B, 128, generator=gan.gen, device="cuda")
    state = gan.store.export_arrays()
    a = gan.d_step(real, z, dr, df, apply=False)
    la, fa = a["d_loss"].item(), a["fake"].clone()
    ga = gan.d_arena.grads.clone()
    gan.store.load_arrays(state, reset_ema=False)
    b = gan.d_step(real, z, dr, df, apply=False)
    assert b["d_loss"].item() == la and torch.equal(b["fake"], fa)          # forward: bit-identical
    assert torch.isfinite(ga).all() and float(ga.abs().max()) > 0
    rel = float((gan.d_arena.grads - ga).double().norm() / ga.double().norm())
    assert rel < 1e-4, rel                                                  # backward: split-K slabs are deterministic
    assert 0.2 < la < 20.0, la             # hinge at random init (the first power iteration underestimates sigma)
    gan.store.load_arrays(state, reset_ema=False)
    u0 = gan.store.vars["generator/first/dense2/u"].clone()
    w0 = gan.g_arena.params.clone()
    losses = gan.train_step(real)
    assert all(np.isfinite(v.item()) for v in losses.values())
    assert not torch.equal(gan.store.vars["generator/first/dense2/u"], u0)
    assert not torch.equal(gan.g_arena.params, w0) and not torch.equal(gan.g_arena.ema, gan.g_arena.params)
    assert torch.isfinite(gan.g_arena.grads).all() and torch.isfinite(gan.g_arena.params).all()


# ------------------------------------------------------------------------------------------
# BASELINE configs 3, 4, 5 at their real sizes, bf16-resident precision (the oracle is far too slow here: properties)
# ------------------------------------------------------------------------------------------
# (kind, N, H, Cin, Cout, k, s): layer shapes of G-128 / D-128 at ch = 96, per-GPU batch 32 (D sees 2B = 64)
C3_LAYERS = [
    ("deconv", 32, 4, 1536, 1536, 4, 2), ("deconv", 32, 16, 768, 384, 4, 2), ("deconv", 32, 64, 192, 96, 4, 2),
    ("deconv", 32, 128, 96, 96, 3, 1), ("deconv", 32, 32, 384, 384, 3, 1),
    ("conv", 64, 64, 96, 96, 3, 1), ("conv", 64, 64, 96, 192, 3, 2), ("conv", 64, 8, 768, 1536, 3, 2),
    ("conv", 64, 4, 1536, 1536, 3, 1), ("conv", 32, 64, 192, 144, 1, 1),
]


@pytest.mark.parametrize("kind,N,H,Cin,Cout,k,s", C3_LAYERS)
def test_adjoint_identities_config3_bf16_resident(kind, N, H, Cin, Cout, k, s):
    """<y, g> = <x, dgrad(g)> = <w, wgrad(x, g)> on the bf16-resident kernels at BASELINE config 3's layer shapes
    (ch = 96: 96 ... 1536 channels, batch 32 / 64).  Each product is rounded to bf16 once more than the other side of
    its identity, so the tolerance is 2e-2 of |y||g| (as for the staged mode)."""
    Fn, hip = _mods()
    Fn.set_precision("bf16")
    try:
        x = _rnd((N, H, H, Cin), 1).bfloat16().requires_grad_(True)
        if kind == "conv":
            w = _rnd((k, k, Cin, Cout), 2, 0.05).requires_grad_(True)
            Ho = H // s
            pad = 1 if k == 3 else 0
            y = Fn.Conv2dFn.apply(x, w, None, s, pad, Ho, Ho, hip.PAD_REFLECT)
        else:
            w = _rnd((k, k, Cout, Cin), 2, 0.05).requires_grad_(True)
            y = Fn.Deconv2dFn.apply(x, w, None, s, 1, None)
        assert y.dtype == torch.bfloat16
        g = _rnd(tuple(y.shape), 3).bfloat16()
        y.backward(g)
        lhs = _dot(y.detach().float(), g.float())
        scale = float(y.detach().double().norm() * g.double().norm())
        tol = 2e-2 * scale
        wb = w.detach().bfloat16().float()          # the kernels see the packed bf16 copy of w
        assert abs(lhs - _dot(x.detach().float(), x.grad.float())) <= tol
        assert abs(lhs - _dot(wb, w.grad)) <= tol
        assert torch.isfinite(y.float()).all()
    finally:
        Fn.set_precision("fp32")


@pytest.mark.parametrize("img,ch,B", [(128, 96, 32), (256, 96, 32), (512, 128, 64)])
def test_full_size_bf16_iteration_configs_3_4_5(img, ch, B):
    """One D + G iteration of BASELINE configs 3, 4 and 5 at the per-GPU batch BASELINE.json states (32 / 32 / 64),
    --precision bf16: finite losses in the range of a random-init hinge GAN, the D op's forward bit-reproducible from
    identical state and inputs, gradients reproducible (no atomics in any GEMM), every gradient and parameter finite,
    u / Adam / EMA state advancing.  (Config 5 needs ~150 GB of HBM for its activations at batch 64.)"""
    from tests.common import make_args
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import model, scope as S, functional as Fn
    from biggan_tensorflow_amd.DiffAugment import draw
    args = make_args(img_size=img, ch=ch, batch_size=B, precision="bf16")
    try:
        gan = model.BigGAN(args, store=S.VariableStore("cuda", seed=42)).build_model()
        real = gan.synthetic_batch(B)
        z = gan.sample_z(B)
        dr, df = draw(B, img, generator=gan.gen, device="cuda"), draw(B, img, generator=gan.gen, device="cuda")
        state = gan.store.export_arrays()
        a = gan.d_step(real, z, dr, df, apply=False)
        la, fa = a["d_loss"].item(), a["fake"].clone()
        ga = gan.d_arena.grads.clone()
        del a
        gan.store.load_arrays(state, reset_ema=False)
        b = gan.d_step(real, z, dr, df, apply=False)
        assert b["d_loss"].item() == la and torch.equal(b["fake"], fa)          # forward: bit-identical
        del b
        assert torch.isfinite(ga).all() and float(ga.abs().max()) > 0
        rel = float((gan.d_arena.grads - ga).double().norm() / ga.double().norm())
        assert rel < 1e-4, rel
        assert 0.2 < la < 20.0, la
        del ga, fa
        gan.store.load_arrays(state, reset_ema=False)
        u0 = gan.store.vars["generator/first/dense2/u"].clone()
        w0 = gan.g_arena.params[:4096].clone()
        losses = gan.train_step(real)
        assert all(np.isfinite(v.item()) for v in losses.values())
        assert not torch.equal(gan.store.vars["generator/first/dense2/u"], u0)
        assert not torch.equal(gan.g_arena.params[:4096], w0)
        assert torch.isfinite(gan.g_arena.grads).all() and torch.isfinite(gan.g_arena.params).all()
        assert torch.isfinite(gan.d_arena.grads).all() and torch.isfinite(gan.d_arena.params).all()
    finally:
        Fn.set_precision("fp32")
        gan = None
        torch.cuda.empty_cache()
