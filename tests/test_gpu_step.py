"""End-to-end parity of the HIP training step against the CPU oracle (float64) on identical
synthetic images / latents / weights / DiffAugment draws (SURVEY.md section 8d parity gate):

    losses           |L - L_ref| / max(|L_ref|, 1e-6) <= 1e-4
    logits, grads    relative L2 error per tensor    <= 1e-3
    post-step state  relative L2 error per tensor    <= 1e-4
"""
import numpy as np
import pytest
import torch

from oracle import ref_model as RM
from tests.common import oracle_trainer, hip_model_like, dev_draws, rel_err, t2n

pytestmark = pytest.mark.gpu

LOSS_TOL, GRAD_TOL, STATE_TOL = 1e-4, 1e-3, 1e-4


def cu(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def _loss_close(a, b):
    return abs(a - b) / max(abs(b), 1e-6) <= LOSS_TOL


def _run_parity(tr, gan, batch, check_state=True):
    cfg = tr.cfg
    before = tr.vs.export()
    # ---------------- D step ----------------
    ro = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])
    ho = gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]))
    assert _loss_close(ho["d_loss"].item(), ro["d_loss"].item()), (ho["d_loss"].item(), ro["d_loss"].item())
    assert rel_err(t2n(ho["real_logits"]), ro["real_logits"].detach().numpy()) < GRAD_TOL
    assert rel_err(t2n(ho["fake_logits"]), ro["fake_logits"].detach().numpy()) < GRAD_TOL
    assert rel_err(t2n(ho["fake"]), ro["fake"].detach().numpy()) < GRAD_TOL
    worst = ("", 0.0)
    for k, g in ro["grads"].items():
        hg = t2n(gan.store.vars[k].bg_grad)
        e = rel_err(hg, g.numpy())
        if e > worst[1]:
            worst = (k, e)
        assert e < GRAD_TOL, ("d grad", k, e)
    after = tr.vs.export()
    if check_state:
        for k in after:
            if np.array_equal(before[k], after[k]):
                assert np.array_equal(t2n(gan.store.vars[k]), before[k].astype(np.float32)), ("unchanged", k)
            else:
                e = rel_err(t2n(gan.store.vars[k]), after[k])
                assert e < STATE_TOL, ("d-step state", k, e)
    # ---------------- G step ----------------
    B = batch["real"].shape[0]
    ro = tr.g_step(batch["z_g"], batch["aug_fake_g"])
    ho = gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]))
    assert _loss_close(ho["g_adv"].item(), ro["g_adv"].item()), (ho["g_adv"].item(), ro["g_adv"].item())
    assert _loss_close(ho["g_loss"].item(), ro["g_loss"].item()), (ho["g_loss"].item(), ro["g_loss"].item())
    if cfg.g_regularization != "none":
        assert _loss_close(ho["g_reg"].item(), ro["g_reg"].item())
    assert rel_err(t2n(ho["fake_logits"]), ro["fake_logits"].detach().numpy()) < GRAD_TOL
    for k, g in ro["grads"].items():
        hg = t2n(gan.store.vars[k].bg_grad)
        e = rel_err(hg, g.numpy())
        assert e < GRAD_TOL, ("g grad", k, e)
    after2 = tr.vs.export()
    if check_state:
        for k in after2:
            e = rel_err(t2n(gan.store.vars[k]), after2[k]) if np.linalg.norm(after2[k]) > 0 else 0.0
            assert e < STATE_TOL, ("g-step state", k, e)
        for k, s in tr.ema.items():
            e = rel_err(t2n(gan.g_arena.view(gan.g_arena.ema, k)), s.numpy())
            assert e < STATE_TOL, ("ema", k, e)


@pytest.mark.parametrize("img,ch,zd,B", [(64, 8, 64, 4), (128, 8, 256, 2)])
def test_step_parity_small(img, ch, zd, B):
    tr = oracle_trainer(img, ch, zd, B)
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 5, B)
    _run_parity(tr, gan, batch)


def test_step_parity_no_regulariser_no_augment():
    tr = oracle_trainer(64, 8, 64, 4, g_regularization="none", da_policy="")
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 6, 4)
    _run_parity(tr, gan, batch)


def test_step_parity_default_init_attention_bypassed():
    """Reference initialisation: SA gamma = 0 and PReLU alpha = 0 (ops.py:486,534): attention is the
    identity and only d(gamma) is non-zero inside the block."""
    tr = oracle_trainer(64, 8, 64, 2, perturb=False)
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 7, 2)
    _run_parity(tr, gan, batch)


def test_two_consecutive_iterations():
    """State carried across iterations: Adam slots, u, BN statistics, EMA (two D+G iterations)."""
    tr = oracle_trainer(64, 8, 64, 2)
    gan = hip_model_like(tr)
    for it in range(2):
        batch = RM.synthetic_batch(tr.cfg, 20 + it, 2)
        _run_parity(tr, gan, batch)


def test_plumbing_config_img64_ch32_batch16():
    """BASELINE config 1 (plumbing): smallest reference-supported size, ch=32, batch=16, fp32."""
    tr = oracle_trainer(64, 32, 256, 16, dtype=torch.float32)
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 9, 16)
    _run_parity(tr, gan, batch, check_state=False)


def test_extension_32px():
    tr = oracle_trainer(32, 16, 64, 4, extension_32=True)
    gan = hip_model_like(tr)
    batch = RM.synthetic_batch(tr.cfg, 10, 4)
    _run_parity(tr, gan, batch)


def test_train_loop_runs_and_loss_is_finite():
    from tests.common import make_args
    from biggan_tensorflow_amd import model, scope as S
    args = make_args(img_size=64, ch=16, batch_size=4, z_dim=128)
    gan = model.BigGAN(args, store=S.VariableStore("cuda")).build_model()
    for _ in range(3):
        losses = gan.train_step(gan.synthetic_batch())
    assert all(np.isfinite(v.item()) for v in losses.values())
    assert gan.counter == 3
