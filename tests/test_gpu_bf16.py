"""bf16-compute mode of the conv / transposed-conv kernels (bg_set_gemm_compute(1)): operands are
rounded to bf16 while staged into LDS, accumulation is fp32.  Stated tolerance: relative L2 error
<= 1e-2 against the float64 reference (bf16 has an 8-bit mantissa: unit round-off 2^-9 = 2e-3 per
operand; measured errors are 2e-3..4e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.common import rel_err, t2n

pytestmark = pytest.mark.gpu
TOL = 1e-2


def cu(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    if grad:
        t.requires_grad_(True)
    return t


@pytest.fixture(params=[1, 3])
def bf16_mode(request):
    """1: operands rounded while they are staged into LDS; 3: whole-operand bf16 copies + the bf16-source kernel."""
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import hip
    L = hip.lib()
    L.bg_set_gemm_compute(request.param)
    assert L.bg_get_gemm_compute() == request.param
    yield
    L.bg_set_gemm_compute(0)
    assert L.bg_get_gemm_compute() == 0


def test_bf16_source_kernels_equal_staged_rounding():
    """Mode 3 rounds the same values the same way (RNE) and accumulates in the same order as mode 1: bit-identical."""
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import hip, functional as Fn
    L = hip.lib()
    rng = np.random.default_rng(0)
    x = cu(rng.standard_normal((4, 16, 16, 128)))
    wc = cu(rng.standard_normal((3, 3, 128, 256)) * 0.05)
    wd = cu(rng.standard_normal((4, 4, 64, 128)) * 0.05)
    outs = {}
    try:
        for mode in (1, 3):
            L.bg_set_gemm_compute(mode)
            outs[mode] = (Fn.Conv2dFn.apply(x, wc, None, 1, 1, 16, 16, hip.PAD_REFLECT).clone(),
                          Fn.Conv2dFn.apply(x, wc, None, 2, 1, 8, 8, hip.PAD_ZERO).clone(),
                          Fn.Deconv2dFn.apply(x, wd, None, 2, 1, None).clone())
    finally:
        L.bg_set_gemm_compute(0)
    for a, b in zip(outs[1], outs[3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,Cin,Cout,k,s", [(2, 16, 64, 128, 3, 1), (2, 16, 64, 64, 3, 2), (4, 8, 128, 256, 3, 1),
                                              (2, 32, 64, 64, 3, 1), (3, 8, 96, 192, 3, 2), (16, 4, 256, 256, 3, 1),
                                              (4, 16, 128, 96, 3, 1),      # 128-wide tile, ragged N = 96 (ch = 96 nets)
                                              (4, 16, 160, 200, 3, 1)])    # ragged N = 200 and Ca = 160 in the wgrad
def test_conv_bf16(bf16_mode, N, H, Cin, Cout, k, s):
    from biggan_tensorflow_amd import functional as Fn, hip
    rng = np.random.default_rng(N + H + Cin + Cout)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) * 0.1
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    xin = F.pad(xt.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
    yr = F.conv2d(xin.contiguous(), wt.permute(3, 2, 0, 1).contiguous(), stride=s).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc, wc = cu(x, True), cu(w, True)
    y = Fn.Conv2dFn.apply(xc, wc, None, s, 1, yr.shape[1], yr.shape[1], hip.PAD_REFLECT)
    y.backward(cu(g))
    e = (rel_err(t2n(y), yr.detach().numpy()), rel_err(t2n(xc.grad), xt.grad.numpy()),
         rel_err(t2n(wc.grad), wt.grad.numpy()))
    assert max(e) < TOL, e
    assert min(e) > 1e-5, ("bf16 path does not seem to be active", e)


@pytest.mark.parametrize("N,H,Cin,Cout,k,s", [(2, 8, 128, 64, 4, 2), (2, 8, 64, 64, 3, 1), (2, 4, 256, 128, 4, 2),
                                              (2, 16, 96, 96, 3, 1), (8, 4, 192, 192, 4, 2),
                                              (4, 16, 96, 128, 4, 2)])     # dgrad: 128-wide tile over N = Cin = 96
def test_deconv_bf16(bf16_mode, N, H, Cin, Cout, k, s):
    from biggan_tensorflow_amd import functional as Fn
    rng = np.random.default_rng(N + H + Cin + Cout + k)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cout, Cin)) * 0.1
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    yr = F.conv_transpose2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), stride=s, padding=1).permute(0, 2, 3, 1)
    g = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(g))
    xc, wc = cu(x, True), cu(w, True)
    y = Fn.Deconv2dFn.apply(xc, wc, None, s, 1, None)
    y.backward(cu(g))
    e = (rel_err(t2n(y), yr.detach().numpy()), rel_err(t2n(xc.grad), xt.grad.numpy()),
         rel_err(t2n(wc.grad), wt.grad.numpy()))
    assert max(e) < TOL, e
    assert min(e) > 1e-5, ("bf16 path does not seem to be active", e)


def test_bf16_mode2_regulariser_and_large_gemm():
    """bg_set_gemm_compute(2): the large plain GEMMs (Gram matrices of the ortho-cosine regulariser and
    their gradients) run on the bf16 MFMA too; loss <= 2e-2, gradient <= 3e-2 relative to float64."""
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import functional as Fn, hip
    from oracle import ref_ops as R
    L = hip.lib()
    L.bg_set_gemm_compute(2)
    try:
        assert L.bg_get_gemm_compute() == 2
        rng = np.random.default_rng(5)
        for shape in [(3, 3, 256, 256), (4, 4, 128, 192), (184, 1024)]:
            w = rng.standard_normal(shape) * 0.05
            wt = torch.tensor(w, requires_grad=True)
            ref = R.ortho_reg_loss(wt, 1e-4, "ortho_cosine") if shape[-1] <= 256 else R.ortho_cosine_closed_form(wt, 1e-4)
            ref.backward()
            wc = cu(w, True)
            loss = Fn.OrthoCosineRegFn.apply(wc, 1e-4)
            loss.backward()
            assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item()), (shape, loss.item(), ref.item())
            e = rel_err(t2n(wc.grad), wt.grad.numpy())
            assert 1e-6 < e < 3e-2, (shape, e)
        # plain GEMM, both kernels (NN and TN)
        A, B = rng.standard_normal((512, 384)), rng.standard_normal((384, 256))
        C = torch.empty(512, 256, device="cuda")
        Fn.gemm(cu(A), cu(B), C, 512, 256, 384, 384, 256, 256)
        assert 1e-5 < rel_err(t2n(C), A @ B) < 1e-2
        C2 = torch.empty(384, 256, device="cuda")
        A2 = rng.standard_normal((512, 384))
        B2 = rng.standard_normal((512, 256))
        Fn.gemm(cu(A2), cu(B2), C2, 384, 256, 512, 384, 256, 256, transA=True)
        assert 1e-5 < rel_err(t2n(C2), A2.T @ B2) < 1e-2
    finally:
        L.bg_set_gemm_compute(0)


def test_bf16_step_losses_close_to_fp32(bf16_mode):
    """Whole D+G iteration in bf16-compute mode: losses within 2e-2 relative of the float64 oracle
    (bf16 tolerance stated separately from the fp32 gate)."""
    from oracle import ref_model as RM
    from tests.common import oracle_trainer, hip_model_like, dev_draws
    tr = oracle_trainer(64, 16, 64, 4)
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 29, 4)
    ro = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False)
    ho = gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]),
                    apply=False)
    assert abs(ho["d_loss"].item() - ro["d_loss"].item()) <= 2e-2 * abs(ro["d_loss"].item())
    assert rel_err(t2n(ho["fake"]), ro["fake"].detach().numpy()) < 2e-2
    k = "discriminator/resblock_down_4/res2/conv_0/kernel"
    assert rel_err(t2n(gan.store.vars[k].bg_grad), ro["grads"][k].numpy()) < 5e-2
