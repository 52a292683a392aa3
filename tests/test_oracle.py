"""CPU tests that pin the oracle (there is no TensorFlow and no reference fixture: parity unpinned).

Known-answer tests of TF op semantics, dual formulations, float64 finite differences and the
algebraic invariants listed in SURVEY.md section 8(c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import kat
from oracle import ref_ops as R
from oracle import ref_model as M

F64 = torch.float64


def _rand(rng, *shape):
    return rng.standard_normal(shape)


# ---------------------------------------------------------------- alignment KATs
@pytest.mark.parametrize("k,s,H", [(3, 2, 8), (3, 1, 6), (1, 1, 4), (3, 2, 6)])
def test_conv_reflect_matches_direct_loops(k, s, H):
    """ops.py:65-95: reflect pad (2*pad split lo/hi) + VALID conv, vs loops."""
    rng = np.random.default_rng(0)
    x = _rand(rng, 2, H, H, 3)
    vs = R.VarStore(F64, 1)
    opt = {"sn": False}
    pad = (k - 1) // 2
    y = R.conv(vs, "d/c", torch.tensor(x), 5, opt, kernel=k, stride=s, pad=pad, use_bias=False)
    w = vs.vars["d/c/kernel"].detach().numpy()
    xp = kat.reflect_pad(x, pad, pad) if pad else x
    y_ref = kat.conv2d_valid(xp, w, s)
    assert y.shape == y_ref.shape == (2, H // s, H // s, 5)
    np.testing.assert_allclose(y.detach().numpy(), y_ref, rtol=1e-12, atol=1e-12)


def test_conv_stride2_window_centres():
    """k3 s2 even H: reflect 1+1, output i reads padded rows 2i..2i+2 = input rows 2i-1..2i+1."""
    H = 8
    x = np.zeros((1, H, H, 1))
    x[0, 3, 5, 0] = 1.0
    vs = R.VarStore(F64, 1)
    vs.load({"d/c/kernel": np.arange(9.0).reshape(3, 3, 1, 1) + 1})
    y = R.conv(vs, "d/c", torch.tensor(x), 1, {"sn": False}, kernel=3, stride=2, pad=1, use_bias=False)
    y = y.detach().numpy()[0, :, :, 0]
    # input row 3 = 2*i - 1 + p  ->  (i=1,p=2), (i=2,p=0); col 5 -> (j=2,q=2), (j=3,q=0)
    exp = np.zeros((4, 4))
    exp[1, 2] = 9.0
    exp[1, 3] = 7.0
    exp[2, 2] = 3.0
    exp[2, 3] = 1.0
    np.testing.assert_array_equal(y, exp)


@pytest.mark.parametrize("k,s", [(4, 2), (3, 1)])
def test_deconv_same_matches_gradient_definition(k, s):
    """ops.py:128: conv2d_transpose SAME == scatter definition (no kernel flip, pad_lo = 1)."""
    rng = np.random.default_rng(1)
    x = _rand(rng, 2, 5, 5, 4)
    vs = R.VarStore(F64, 2)
    y = R.deconv(vs, "generator/d", torch.tensor(x), 3, {"sn": False}, kernel=k, stride=s, use_bias=False)
    w = vs.vars["generator/d/kernel"].detach().numpy()
    assert w.shape == (k, k, 3, 4)
    y_ref = kat.conv2d_transpose_same(x, w, s)
    np.testing.assert_allclose(y.detach().numpy(), y_ref, rtol=1e-12, atol=1e-12)


def test_deconv_is_adjoint_of_same_conv():
    """<conv_same(a), b> == <a, conv2d_transpose(b)> for k4 s2 - the property TF defines it by."""
    rng = np.random.default_rng(2)
    a = _rand(rng, 1, 8, 8, 2)          # conv input  [B, sH, sW, Cout]
    b = _rand(rng, 1, 4, 4, 3)          # conv output [B, H, W, Cin]
    w = _rand(rng, 4, 4, 2, 3)          # [k,k,Cout,Cin]
    _, lo, hi = kat.same_padding(8, 4, 2)
    ap = np.pad(a, [(0, 0), (lo, hi), (lo, hi), (0, 0)])
    fwd = kat.conv2d_valid(ap, w, 2)
    bt = kat.conv2d_transpose_same(b, w, 2)
    np.testing.assert_allclose((fwd * b).sum(), (a * bt).sum(), rtol=1e-12)


# ---------------------------------------------------------------- DiffAugment closed forms
@pytest.mark.parametrize("S", [7, 8, 32, 64])
def test_diffaugment_closed_forms_bit_identical_to_literal(S):
    rng = np.random.default_rng(S)
    B = 6
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    d = R.draw_diffaugment(rng, B, S)
    shift, cs, off_max = R.diffaugment_params(S)
    # force the extreme draws too
    d["t_x"][0], d["t_y"][0] = -shift, shift
    d["o_x"][1], d["o_y"][1] = 0, off_max - 1
    lit = R.translation_literal(x, d["t_x"], d["t_y"])
    ix = R.translation_index(S, d["t_x"])
    iy = R.translation_index(S, d["t_y"])
    closed = np.zeros_like(x)
    for b in range(B):
        for i in range(S):
            for j in range(S):
                if ix[b, i] >= 0 and iy[b, j] >= 0:
                    closed[b, i, j] = x[b, ix[b, i], iy[b, j]]
    assert np.array_equal(lit, closed)
    litc = R.cutout_literal(x, d["o_x"], d["o_y"])
    closedc = x * R.cutout_mask(S, d["o_x"], d["o_y"])[..., None]
    assert np.array_equal(litc, closedc)
    # torch formulation used by the model oracle
    y = R.diffaugment(torch.tensor(x), d, "translation,cutout").numpy()
    assert np.array_equal(y, closedc * 0 + R.cutout_mask(S, d["o_x"], d["o_y"])[..., None] * closed)


def test_diffaugment_constants():
    assert R.diffaugment_params(128) == (16, 64, 129)
    assert R.diffaugment_params(256) == (32, 128, 257)
    assert R.diffaugment_params(64) == (8, 32, 65)
    assert R.diffaugment_params(32) == (4, 16, 33)


# ---------------------------------------------------------------- spectral norm
def test_spectral_norm_sigma_and_gradient():
    rng = np.random.default_rng(3)
    vs = R.VarStore(F64, 3)
    w = torch.tensor(_rand(rng, 3, 3, 4, 6), requires_grad=True)
    wn = R.spectral_norm(vs, "s", w)
    W = w.detach().reshape(-1, 6)
    u = vs.vars["s/u"]
    v_hat = R.l2_normalize(u @ W.t())
    u_hat = R.l2_normalize(v_hat @ W)
    sigma = (v_hat @ W @ u_hat.t()).item()
    assert abs(sigma - (v_hat @ W).norm().item()) < 1e-12            # sigma = ||v_hat W||
    np.testing.assert_allclose(wn.detach().numpy(), (w / sigma).detach().numpy(), rtol=1e-13)
    # closed-form backward: dW = (G - <G, Wn> v_hat^T u_hat) / sigma
    G = torch.tensor(_rand(rng, 3, 3, 4, 6))
    (gw,) = torch.autograd.grad((wn * G).sum(), w)
    Gm = G.reshape(-1, 6)
    exp = (Gm - (Gm * (W / sigma)).sum() * (v_hat.t() @ u_hat)) / sigma
    np.testing.assert_allclose(gw.reshape(-1, 6).numpy(), exp.numpy(), rtol=1e-10, atol=1e-12)
    assert torch.allclose(vs.state_updates["s/u"], u_hat)


# ---------------------------------------------------------------- regulariser identity
@pytest.mark.parametrize("c", [3, 8, 33])
def test_ortho_cosine_closed_form(c):
    rng = np.random.default_rng(c)
    w = torch.tensor(_rand(rng, 3, 3, 5, c))
    a = R.ortho_reg_loss(w, 1e-4, "ortho_cosine")
    b = R.ortho_cosine_closed_form(w, 1e-4)
    assert abs(a.item() - b.item()) <= 1e-14 * max(1.0, abs(a.item()))


# ---------------------------------------------------------------- losses
def test_hinge_flood_values_at_zero_logits():
    z = torch.zeros(4, 1, dtype=F64)
    assert R.discriminator_loss("hinge", z, z, 0).item() == 2.0
    assert R.discriminator_loss("hinge", z, z, 0.1).item() == pytest.approx(2.0)
    assert R.generator_loss("hinge", z, None, 0).item() == 0.0
    assert R.generator_loss("hinge", z, None, 0.05).item() == pytest.approx(0.1)   # |0-0.05|+0.05
    f = torch.full((4, 1), 1.0, dtype=F64, requires_grad=True)
    L = R.generator_loss("hinge", f, None, 0.05)                   # -1 < 0.05 -> sign flips
    (g,) = torch.autograd.grad(L, f)
    assert torch.all(g > 0)


# ---------------------------------------------------------------- model-level checks
def _small_cfg(**kw):
    base = dict(img_size=64, ch=8, batch_size=2, z_dim=64)
    base.update(kw)
    return M.Config(**base)


def test_manifest_names_biggan128():
    """Variable manifest of SURVEY R16/R17 (TF scope names)."""
    cfg = M.Config(img_size=128, ch=8, batch_size=2)
    tr = M.Trainer(cfg, torch.float32).build()
    names = list(tr.vs.vars.keys())
    assert cfg.z_split_sizes() == [96, 32, 32, 32, 32, 32]
    for n, shape in [
        ("generator/first/dense1/kernel", (96, 184)),
        ("generator/first/dense1/u", (1, 184)),
        ("generator/first/prelu/alpha", (184,)),
        ("generator/first/dense2/kernel", (184, 16 * 16 * 8)),
        ("generator/resblock_up_16/res1/batch_norm/pop_mean", (128,)),
        ("generator/resblock_up_16/res1/batch_norm/beta/kernel", (32, 128)),
        ("generator/resblock_up_16/res1/batch_norm/gamma/u", (1, 128)),
        ("generator/resblock_up_16/res1/deconv_0/kernel", (4, 4, 128, 128)),
        ("generator/resblock_up_16/res1/deconv_0/u", (1, 128)),
        ("generator/resblock_up_8/res2/deconv_0/kernel", (3, 3, 64, 64)),
        ("generator/resblock_up_8/skip/deconv_0/kernel", (4, 4, 64, 128)),
        ("generator/self_attention/f_conv/kernel", (1, 1, 16, 2)),
        ("generator/self_attention/h_conv/kernel", (1, 1, 16, 8)),
        ("generator/self_attention/attn_conv/kernel", (1, 1, 8, 16)),
        ("generator/self_attention/gamma", (1,)),
        ("generator/batch_norm/moving_variance", (8,)),
        ("generator/prelu/alpha", (8,)),
        ("generator/G_logit/kernel", (3, 3, 8, 3)),
        ("generator/G_logit/u", (1, 3)),
        ("discriminator/resblock_down_1/res1/prelu/alpha", (3,)),
        ("discriminator/resblock_down_1/res1/conv_0/kernel", (3, 3, 3, 8)),
        ("discriminator/resblock_down_1/skip/conv_0/u", (1, 8)),
        ("discriminator/self_attention/g_conv/bias", (1,)),
        ("discriminator/resblock_down_16/res2/conv_0/kernel", (3, 3, 128, 128)),
        ("discriminator/resblock/res1/conv_0/kernel", (3, 3, 128, 128)),
        ("discriminator/prelu/alpha", (128,)),
        ("discriminator/D_logit/kernel", (128, 1)),
        ("discriminator/D_logit/bias", (1,)),
        ("discriminator/D_logit/u", (1, 1)),
    ]:
        assert n in tr.vs.vars, n
        assert tuple(tr.vs.vars[n].shape) == shape, (n, tuple(tr.vs.vars[n].shape))
    assert not any("bias" in n for n in names if "resblock_down" in n)      # bias_in_d False
    # SN'd weight counts: 42 in G-128, 22 in D-128
    assert sum(n.startswith("generator") and n.endswith("/u") for n in names) == 42
    assert sum(n.startswith("discriminator") and n.endswith("/u") for n in names) == 22


def test_cumulative_scope_names_256():
    cfg = M.Config(img_size=256, ch=8, batch_size=1)
    tr = M.Trainer(cfg, torch.float32).build()
    names = tr.vs.vars.keys()
    assert "generator/resblock_up_8_0/res1/deconv_0/kernel" in names
    assert "generator/resblock_up_8_0_1/res1/deconv_0/kernel" in names      # BigGAN.py:455 quirk
    assert "discriminator/resblock_down_8_0_1/res1/conv_0/kernel" in names
    assert cfg.z_split_sizes() == [88] + [28] * 6


def test_sa_block_is_identity_at_gamma_zero():
    cfg = _small_cfg()
    vs = R.VarStore(F64, 0)
    x = torch.tensor(np.random.default_rng(0).standard_normal((2, 8, 8, 16)))
    y = R.self_attention_2(vs, "generator/self_attention", x, 16, {"sn": True, "self_attention_bias": True})
    assert torch.equal(x, y)


def test_step_runs_and_updates_expected_state():
    cfg = _small_cfg()
    tr = M.Trainer(cfg, F64).build()
    M.perturb_for_parity(tr.vs)
    batch = M.synthetic_batch(cfg, 5)
    before = tr.vs.export()
    out = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])
    after = tr.vs.export()
    assert np.isfinite(out["d_loss"].item())
    changed = {k for k in before if not np.array_equal(before[k], after[k])}
    # D step: every D trainable + u of BOTH nets + G's pop/moving stats; no G trainable
    for k in before:
        leaf = k.rsplit("/", 1)[-1]
        if k.startswith("discriminator"):
            assert k in changed, k
        elif leaf in ("u", "pop_mean", "pop_var", "moving_mean", "moving_variance"):
            assert k in changed, k
        else:
            assert k not in changed, k
    out = tr.g_step(batch["z_g"], batch["aug_fake_g"])
    after2 = tr.vs.export()
    changed2 = {k for k in after if not np.array_equal(after[k], after2[k])}
    for k in after:
        leaf = k.rsplit("/", 1)[-1]
        if k.startswith("generator"):
            assert k in changed2, k
        elif leaf == "u":
            if after[k].size > 1:          # a 1-element u is already +-1 after the first step
                assert k in changed2, k
        else:
            assert k not in changed2, k
    assert out["g_reg"].item() > 0


def test_finite_difference_gradients():
    """float64 central differences on a handful of D and G parameters."""
    cfg = _small_cfg(ch=8, z_dim=64)
    tr = M.Trainer(cfg, F64).build()
    M.perturb_for_parity(tr.vs)
    batch = M.synthetic_batch(cfg, 11)

    def d_loss():
        with torch.no_grad():
            return tr.d_forward(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])["d_loss"].item()

    def g_loss():
        with torch.no_grad():
            return tr.g_forward(batch["z_g"], batch["aug_fake_g"])["g_loss"].item()

    gd = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False)["grads"]
    gg = tr.g_step(batch["z_g"], batch["aug_fake_g"], apply=False)["grads"]
    tr.vs.freeze_uv = True        # u_hat / v_hat are stop-gradient constants of the run (ops.py:738-739)
    rng = np.random.default_rng(0)
    eps = 1e-6
    checks = [
        (d_loss, gd, "discriminator/resblock_down_2/res1/conv_0/kernel"),
        (d_loss, gd, "discriminator/self_attention/f_conv/kernel"),
        (d_loss, gd, "discriminator/resblock_down_1/res1/prelu/alpha"),
        (d_loss, gd, "discriminator/D_logit/kernel"),
        (g_loss, gg, "generator/resblock_up_4/res1/deconv_0/kernel"),
        (g_loss, gg, "generator/resblock_up_2/res2/batch_norm/gamma/kernel"),
        (g_loss, gg, "generator/self_attention/gamma"),
        (g_loss, gg, "generator/first/dense2/kernel"),
        (g_loss, gg, "generator/batch_norm/gamma"),
        (g_loss, gg, "generator/G_logit/kernel"),
    ]
    for fn, grads, name in checks:
        p = tr.vs.vars[name]
        flat = p.detach().view(-1)
        for idx in rng.integers(0, flat.numel(), 2):
            old = flat[idx].item()
            with torch.no_grad():
                flat[idx] = old + eps
            lp = fn()
            with torch.no_grad():
                flat[idx] = old - eps
            lm = fn()
            with torch.no_grad():
                flat[idx] = old
            fd = (lp - lm) / (2 * eps)
            an = grads[name].reshape(-1)[idx].item()
            assert abs(fd - an) <= 1e-5 * max(1e-3, abs(an), abs(fd)) + 1e-9, (name, idx, fd, an)


def test_adam_tf_form():
    opt = M.AdamTF(2e-4, 0.0, 0.9)
    p = {"a": torch.tensor([1.0, -2.0], dtype=F64)}
    g = {"a": torch.tensor([0.5, -0.25], dtype=F64)}
    opt.step(p, g)
    lr_t = 2e-4 * np.sqrt(1 - 0.9)
    v = 0.1 * np.array([0.25, 0.0625])
    exp = np.array([1.0, -2.0]) - lr_t * np.array([0.5, -0.25]) / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(p["a"].numpy(), exp, rtol=1e-14)


# ----------------------------------------------------------------------------------
# known answers PUBLISHED by TensorFlow itself (documentation examples and the closed forms asserted by its
# own unit tests), restated here because TensorFlow cannot be installed: they pin the two alignment
# conventions the whole path rests on independently of this repository's own derivations
# ----------------------------------------------------------------------------------
def test_tf_pad_reflect_documentation_example():
    """tf.pad docs: t = [[1,2,3],[4,5,6]], paddings [[1,1],[2,2]], mode REFLECT."""
    from oracle import kat
    t = np.array([[1, 2, 3], [4, 5, 6]], np.float64)
    expect = np.array([[6, 5, 4, 5, 6, 5, 4],
                       [3, 2, 1, 2, 3, 2, 1],
                       [6, 5, 4, 5, 6, 5, 4],
                       [3, 2, 1, 2, 3, 2, 1]], np.float64)
    # kat.reflect_pad pads H and W by the same (lo, hi): pad by (2, 2) and crop the row axis back to (1, 1)
    got = kat.reflect_pad(t[None, :, :, None], 2, 2)[0, 1:-1, :, 0]
    assert np.array_equal(got, expect)
    assert np.array_equal(F.pad(torch.tensor(t)[None, None], (2, 2, 1, 1), mode="reflect")[0, 0].numpy(), expect)


def _tf_transpose_conv_counts(n_out, stride, k, single_stride):
    """Closed forms asserted by TensorFlow's conv2d_transpose_test.py for all-ones inputs and filters."""
    v = np.zeros(n_out)
    for i in range(n_out):
        if single_stride:              # testConv2DTransposeSingleStride: k 3, stride 1: 2 taps at the border, 3 inside
            v[i] = 3 if 0 < i < n_out - 1 else 2
        else:                          # testConv2DTransposeSame: k 3, stride 2: 2 taps at even interior rows, else 1
            v[i] = 2 if (i % stride == 0 and 0 < i < n_out - 1) else 1
    return v


def test_tf_conv2d_transpose_unit_test_closed_forms():
    """testConv2DTransposeSingleStride: x ones [2,6,4,3], f ones [3,3,2,3], strides 1, SAME -> 12 at the corners,
    18 on the edges, 27 inside (target 4*3, +2*3, +5*3).  testConv2DTransposeSame: strides 2, output [2,12,8,2] ->
    3 everywhere, +3 where exactly one of (h, w) is an even interior index, +9 where both are."""
    from oracle import kat
    x = np.ones((2, 6, 4, 3))
    f = np.ones((3, 3, 2, 3))
    for stride, single in ((1, True), (2, False)):
        y = kat.conv2d_transpose_same(x, f, stride)
        assert y.shape == (2, 6 * stride, 4 * stride, 2)
        ch = _tf_transpose_conv_counts(6 * stride, stride, 3, single)
        cw = _tf_transpose_conv_counts(4 * stride, stride, 3, single)
        expect = 3.0 * np.outer(ch, cw)
        assert np.array_equal(y[0, :, :, 0], expect) and np.array_equal(y[1, :, :, 1], expect)
        if single:
            assert (expect[0, 0], expect[0, 1], expect[1, 1]) == (12.0, 18.0, 27.0)
        else:
            assert (expect[1, 1], expect[2, 1], expect[2, 2]) == (3.0, 6.0, 12.0)
        # the fast formulation used by the oracle model agrees on this case for the shapes the path uses
        if (3, stride) == (3, 1):
            w = torch.tensor(f)
            yt = F.conv_transpose2d(torch.tensor(x).permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=1, padding=1)
            assert np.array_equal(yt.permute(0, 2, 3, 1).numpy(), y)


def test_batch_renorm_reductions_and_hand_case():
    """ops.py:645-715 / 600-609: (1) at its initial state (renorm_weight = 0) condition_batch_renorm IS
    condition_batch_norm; (2) the Keras form with renorm statistics equal to the batch's is plain batch norm;
    (3) a hand-computed channel: batch mean 1, var 4 (sigma 2), renorm mean 0, renorm var 1, weight 1 ->
    r = clip(2, 2/3, 1.5) = 1.5, d = clip(1, -.5, .5) = 0.5, y = xhat * 1.5 * gamma + beta + 0.5 * gamma;
    (4) r and d carry no gradient."""
    torch.manual_seed(0)
    opt = {"sn": False, "bn_momentum": 0.98}
    x = torch.randn(3, 4, 4, 5, dtype=torch.float64)
    z = torch.randn(3, 7, dtype=torch.float64)
    vs = R.VarStore(torch.float64, 1)
    a = R.condition_batch_norm(vs, "g/batch_norm", x, z, opt)
    vr = R.VarStore(torch.float64, 1)
    b = R.condition_batch_renorm(vr, "g/batch_renorm", x, z, opt)
    assert torch.allclose(a, b, atol=1e-14)
    assert float(vr.state_updates["g/batch_renorm/renorm_weight"]) == pytest.approx(1e-4)
    # the population statistics of the non-shared form decay with renorm_momentum (0.9), ops.py:658-659
    mean = x.mean(dim=(0, 1, 2))
    assert torch.allclose(vr.state_updates["g/batch_renorm/pop_mean"], 0.1 * mean)

    vk = R.VarStore(torch.float64, 1)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    vk.get("d/batch_renorm/renorm_mean", (5,), 0.0, trainable=False)
    vk.get("d/batch_renorm/renorm_stddev", (5,), 1.0, trainable=False)
    with torch.no_grad():
        vk.vars["d/batch_renorm/renorm_mean"].copy_(mean)
        vk.vars["d/batch_renorm/renorm_stddev"].copy_(torch.sqrt(var + 1e-5))
    k = R.batch_renorm(vk, "d/batch_renorm", x, opt)
    plain = R.batch_norm(R.VarStore(torch.float64, 1), "d/batch_norm", x, opt)
    assert torch.allclose(k, plain, atol=1e-12)

    xs = torch.tensor([-1.0, 3.0], dtype=torch.float64).reshape(2, 1, 1, 1)          # mean 1, var 4
    vh = R.VarStore(torch.float64, 1)
    vh.get("h/batch_renorm/renorm_weight", (), 0.0, trainable=False)
    with torch.no_grad():
        vh.vars["h/batch_renorm/renorm_weight"].fill_(1.0)
    zh = torch.ones(2, 1, dtype=torch.float64)
    y = R.condition_batch_renorm(vh, "h/batch_renorm", xs, zh, opt)
    gam = (zh @ vh.vars["h/batch_renorm/gamma/kernel"] + vh.vars["h/batch_renorm/gamma/bias"]).reshape(2, 1, 1, 1)
    bet = (zh @ vh.vars["h/batch_renorm/beta/kernel"] + vh.vars["h/batch_renorm/beta/bias"]).reshape(2, 1, 1, 1)
    xhat = (xs - 1.0) / np.sqrt(4.0 + 1e-5)
    assert torch.allclose(y, xhat * 1.5 * gam + bet + 0.5 * gam, atol=1e-12)

    # stop-gradient: d(sum y)/dx through a frozen-(r,d) restatement equals autograd's
    xr = x.clone().requires_grad_(True)
    v4 = R.VarStore(torch.float64, 1)
    v4.get("q/batch_renorm/renorm_weight", (), 0.0, trainable=False)
    with torch.no_grad():
        v4.vars["q/batch_renorm/renorm_weight"].fill_(0.5)
    y4 = R.condition_batch_renorm(v4, "q/batch_renorm", xr, z, opt)
    g_auto, = torch.autograd.grad((y4 * y4).sum(), xr)
    with torch.no_grad():
        m0 = x.mean(dim=(0, 1, 2)); v0 = ((x - m0) ** 2).mean(dim=(0, 1, 2)); s0 = torch.sqrt(v0 + 1e-5)
        sw = 0.5 * np.sqrt(1 + 1e-5) + 0.5 * s0
        r0 = torch.clamp(s0 / sw, 1 / 1.5, 1.5); d0 = torch.clamp((m0 - 0.5 * m0) / sw, -0.5, 0.5)
    xq = x.clone().requires_grad_(True)
    gam = (z @ v4.vars["q/batch_renorm/gamma/kernel"] + v4.vars["q/batch_renorm/gamma/bias"]).reshape(3, 1, 1, 5)
    bet = (z @ v4.vars["q/batch_renorm/beta/kernel"] + v4.vars["q/batch_renorm/beta/bias"]).reshape(3, 1, 1, 5)
    mq = xq.mean(dim=(0, 1, 2)); vq = ((xq - mq) ** 2).mean(dim=(0, 1, 2))
    yq = (xq - mq) * torch.rsqrt(vq + 1e-5) * (r0 * gam) + bet + d0 * gam
    g_frozen, = torch.autograd.grad((yq * yq).sum(), xq)
    assert torch.allclose(g_auto, g_frozen, atol=1e-10)


@pytest.mark.parametrize("gan_type", ["wgan-gp", "wgan-lp", "ra-dragan"])
def test_gradient_penalty_finite_differences(gan_type):
    """BigGAN.py:717-742: float64 central differences of d_loss INCLUDING the gradient penalty (whose parameter
    gradient needs the second derivative of D) on a few discriminator parameters."""
    cfg = _small_cfg(ch=8, z_dim=64, gan_type=gan_type)
    tr = M.Trainer(cfg, F64).build()
    M.perturb_for_parity(tr.vs)
    batch = M.synthetic_batch(cfg, 11)
    args = (batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])
    out = tr.d_step(*args, apply=False, gp=batch["gp"])
    assert float(out["gp"].detach()) > 0
    gd = out["grads"]
    tr.vs.freeze_uv = True
    rng = np.random.default_rng(0)
    eps = 1e-6
    for name in ("discriminator/resblock_down_2/res1/conv_0/kernel", "discriminator/self_attention/f_conv/kernel",
                 "discriminator/self_attention/gamma", "discriminator/resblock_down_1/res1/prelu/alpha",
                 "discriminator/D_logit/kernel"):
        flat = tr.vs.vars[name].detach().view(-1)
        for idx in rng.integers(0, flat.numel(), 2):
            old = flat[idx].item()
            vals = []
            for sgn in (1, -1):
                with torch.no_grad():
                    flat[idx] = old + sgn * eps
                vals.append(tr.d_forward(*args, gp=batch["gp"])["d_loss"].item())
            with torch.no_grad():
                flat[idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            an = gd[name].reshape(-1)[idx].item()
            assert abs(fd - an) <= 2e-5 * max(1e-3, abs(an), abs(fd)) + 1e-8, (name, idx, fd, an)


# ----------------------------------------------------------------------------------
# More known answers PUBLISHED by TensorFlow's API documentation, restated ("the remaining lever is widening
# the TF-published KAT set").  Each checks the oracle function a reference call site maps to against the formula or
# worked example the TF docs give for that op, written independently of the oracle's own (numerically stable) form.
# ----------------------------------------------------------------------------------
def test_tf_sigmoid_cross_entropy_documented_definition():
    """tf.nn.sigmoid_cross_entropy_with_logits docs: loss = z * -log(sigmoid(x)) + (1 - z) * -log(1 - sigmoid(x)), and the
    stable form max(x, 0) - x * z + log(1 + exp(-abs(x))) 'to ensure stability and avoid overflow' (utils.py:366-369 via
    cls_loss_fn; ops.py:763-777 for the gan / dragan losses with labels of ones / zeros)."""
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((5, 7)) * 3.0, dtype=F64)
    z = torch.tensor(rng.integers(0, 2, (5, 7)).astype(np.float64))
    w = torch.tensor(rng.uniform(0.5, 2.0, (7,)), dtype=F64)
    sig = 1.0 / (1.0 + np.exp(-x.numpy()))
    definition = z.numpy() * -np.log(sig) + (1 - z.numpy()) * -np.log(1 - sig)
    assert abs(R.cls_loss_logistic(z, x, w).item() - (definition * w.numpy()).mean()) < 1e-12
    assert np.allclose(R._sce(True, x).numpy(), -np.log(sig), atol=1e-12)
    assert np.allclose(R._sce(False, x).numpy(), -np.log(1 - sig), atol=1e-12)
    big = torch.tensor([[80.0, -80.0]], dtype=F64)                 # the documented reason for the stable form
    assert torch.isfinite(R._sce(True, big)).all() and torch.isfinite(R._sce(False, big)).all()


def test_tf_l2_normalize_documented_formula():
    """tf.nn.l2_normalize docs: output = x / sqrt(max(sum(x**2), epsilon)), epsilon = 1e-12 (ops.py:733,736)."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 9))
    assert np.allclose(R.l2_normalize(torch.tensor(x)).numpy(), x / np.sqrt(max((x ** 2).sum(), 1e-12)), atol=1e-15)
    tiny = np.full((1, 4), 1e-9)                                   # sum of squares 4e-18 < epsilon: divided by 1e-6
    assert np.allclose(R.l2_normalize(torch.tensor(tiny)).numpy(), tiny / 1e-6, rtol=1e-12)


def test_tf_moments_and_batch_normalization_documented_formulas():
    """tf.nn.moments docs: 'the mean and variance of x' over the axes (population variance); tf.nn.batch_normalization
    docs: (x - mean) / sqrt(variance + epsilon) * scale + offset (ops.py:630-638), and tf.layers.batch_normalization's
    moving averages moving = moving * momentum + batch * (1 - momentum) (ops.py:581-585)."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((4, 3, 3, 5)) * 2.0 + 1.0
    vs = R.VarStore()
    opt = {"bn_momentum": 0.9}
    y = R.batch_norm(vs, "bn", torch.tensor(x), opt, is_training=True).detach().numpy()
    mean, var = x.mean(axis=(0, 1, 2)), x.var(axis=(0, 1, 2))        # numpy var = population variance
    assert np.allclose(y, (x - mean) / np.sqrt(var + 1e-5), atol=1e-12)   # gamma = 1, beta = 0 at initialisation
    vs.commit()
    n = 4 * 3 * 3
    assert np.allclose(vs.vars["bn/moving_mean"].numpy(), 0.0 * 0.9 + mean * 0.1, atol=1e-14)
    assert np.allclose(vs.vars["bn/moving_variance"].numpy(), 1.0 * 0.9 + var * n / (n - 1) * 0.1, atol=1e-14)
    y_inf = R.batch_norm(vs, "bn", torch.tensor(x), opt, is_training=False).detach().numpy()
    mm, mv = vs.vars["bn/moving_mean"].numpy(), vs.vars["bn/moving_variance"].numpy()
    assert np.allclose(y_inf, (x - mm) / np.sqrt(mv + 1e-5), atol=1e-12)


def test_tf_same_padding_documented_arithmetic():
    """TF 'SAME' padding as documented for tf.nn.conv2d: out = ceil(in / stride), pad_along = max((out - 1) * stride +
    filter - in, 0), pad_top = pad_along // 2, pad_bottom = pad_along - pad_top - the odd pixel goes to the bottom /
    right (--conv_padding zero, ops.py:77-80); checked on the documented worked sizes and against a direct-loop VALID
    convolution of the explicitly padded input."""
    assert kat.same_padding(13, 6, 5) == (3, 1, 2)      # TF docs' own example: in 13, filter 6, stride 5: 1 left, 2 right
    assert kat.same_padding(8, 3, 2) == (4, 0, 1)
    assert kat.same_padding(7, 3, 2) == (4, 1, 1)
    rng = np.random.default_rng(3)
    for H, k, s in ((8, 3, 2), (7, 3, 2), (6, 3, 1), (8, 4, 2)):
        x = rng.standard_normal((2, H, H, 3))
        vs = R.VarStore(seed=H)
        y = R.conv(vs, "c", torch.tensor(x), 4, {"padding_type": "zero", "sn": False}, kernel=k, stride=s, pad=1,
                   use_bias=False).detach().numpy()
        out, lo, hi = kat.same_padding(H, k, s)
        xp = np.pad(x, ((0, 0), (lo, hi), (lo, hi), (0, 0)))
        ref = kat.conv2d_valid(xp, vs.vars["c/kernel"].detach().numpy(), s)
        assert y.shape == (2, out, out, 4) and np.allclose(y, ref, atol=1e-12)


def test_tf_pooling_and_resize_documented_semantics():
    """tf.layers.max_pooling2d / average_pooling2d(pool 2, strides 2, VALID) take each non-overlapping 2 x 2 window
    (ops.py:508-514); tf.image.resize_nearest_neighbor with align_corners = False maps output pixel i to input pixel
    floor(i * in / out) (ops.py:516-519): for a factor of 2 every input pixel is repeated 2 x 2."""
    x = np.arange(2 * 4 * 4 * 1, dtype=np.float64).reshape(2, 4, 4, 1)
    mp = R.max_pooling(torch.tensor(x)).numpy()
    ap = R.avg_pooling(torch.tensor(x)).numpy()
    up = R.up_sample(torch.tensor(x)).numpy()
    for b in range(2):
        for i in range(2):
            for j in range(2):
                win = x[b, 2 * i:2 * i + 2, 2 * j:2 * j + 2, 0]
                assert mp[b, i, j, 0] == win.max() and ap[b, i, j, 0] == win.mean()
        for i in range(8):
            for j in range(8):
                assert up[b, i, j, 0] == x[b, (i * 4) // 8, (j * 4) // 8, 0]


def test_tf_adam_documented_update_three_steps():
    """tf.train.AdamOptimizer docs: lr_t = learning_rate * sqrt(1 - beta2^t) / (1 - beta1^t); m_t = beta1 * m + (1 - beta1) * g;
    v_t = beta2 * v + (1 - beta2) * g * g; variable -= lr_t * m_t / (sqrt(v_t) + epsilon) - note epsilon OUTSIDE the bias
    correction ('epsilon hat' of the paper), three consecutive steps on a scalar (BigGAN.py:923-927)."""
    lr, b1, b2, eps = 2e-4, 0.0, 0.9, 1e-8
    opt = M.AdamTF(lr, b1, b2, eps)
    p = {"w": torch.tensor([0.5, -1.0], dtype=F64)}
    grads = [np.array([0.3, -0.2]), np.array([-0.1, 0.4]), np.array([0.05, 0.05])]
    w, m, v = np.array([0.5, -1.0]), np.zeros(2), np.zeros(2)
    for t, g in enumerate(grads, 1):
        opt.step(p, {"w": torch.tensor(g, dtype=F64)})
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        w = w - lr_t * m / (np.sqrt(v) + eps)
        assert np.allclose(p["w"].detach().numpy(), w, rtol=0, atol=1e-15)


def test_batch_norm_tangent_backward_formulas_finite_differences():
    """The closed forms behind functional.BnTangentFn / bg_bn_tangent_bwd_coefs (gradient penalty through --bn_in_d):
    T(x, xd, g) = g r (xd - mean xd - xh mean(xd xh)); for s = dL/dT the derivatives w.r.t. xd, g and x (through the
    batch statistics) written in include/biggan_hip.h, against central differences in float64."""
    rng = np.random.default_rng(0)
    n, eps = 50, 1e-5
    x, xd, s, g = rng.standard_normal(n) * 1.7 + 0.3, rng.standard_normal(n), rng.standard_normal(n), 1.3

    def T(x, xd, g):
        mu = x.mean()
        r = 1.0 / np.sqrt(((x - mu) ** 2).mean() + eps)
        xh = (x - mu) * r
        return g * r * (xd - xd.mean() - xh * (xd * xh).mean())

    def L(x, xd, g):
        return float((T(x, xd, g) * s).sum())

    mu = x.mean()
    r = 1.0 / np.sqrt(((x - mu) ** 2).mean() + eps)
    xh = (x - mu) * r
    m1 = xd.mean()
    u = xd - m1
    m2 = (u * xh).mean()
    d_xd = g * r * (s - s.mean() - xh * (s * xh).mean())
    d_g = (s * r * (u - xh * m2)).sum()
    w = g * r * r * s
    Sw, Swu, Swx = w.sum(), (w * u).sum(), (w * xh).sum()
    d_x = xh * (3 * m2 * Swx - Swu) / n - m2 * w - (Swx / n) * u + m2 * Sw / n

    def fd(f, v, h=1e-6):
        out = np.zeros_like(v)
        for i in range(len(v)):
            vp, vm = v.copy(), v.copy()
            vp[i] += h
            vm[i] -= h
            out[i] = (f(vp) - f(vm)) / (2 * h)
        return out
    assert np.abs(d_xd - fd(lambda v: L(x, v, g), xd)).max() < 1e-7
    assert np.abs(d_x - fd(lambda v: L(v, xd, g), x)).max() < 1e-7
    assert abs(d_g - (L(x, xd, g + 1e-6) - L(x, xd, g - 1e-6)) / 2e-6) < 1e-7
