"""Golden fixture tests.  tests/golden/step_img64_ch8.npz holds outputs of the float64 oracle
(tests/golden/make_golden.py); the CPU test guards the oracle against regressions, the GPU test
checks the HIP path against the frozen vectors (no oracle arithmetic involved in the comparison)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "step_img64_ch8.npz"))
GOLD_GP = np.load(os.path.join(HERE, "golden", "step_img64_ch8_radragan.npz"))


def test_oracle_reproduces_golden():
    res = MG.compute()
    assert set(res.keys()) == set(GOLD.files)
    for k in GOLD.files:
        np.testing.assert_allclose(res[k], GOLD[k], rtol=1e-9, atol=1e-12, err_msg=k)


def test_oracle_reproduces_gradient_penalty_golden():
    res = MG.compute_gp()
    assert set(res.keys()) == set(GOLD_GP.files)
    for k in GOLD_GP.files:
        np.testing.assert_allclose(res[k], GOLD_GP[k], rtol=1e-9, atol=1e-12, err_msg=k)


@pytest.mark.gpu
def test_hip_gradient_penalty_step_matches_golden():
    """The D op of --gan_type ra-dragan (the reference's default) against the frozen float64 vectors."""
    from tests.common import hip_model_like, dev_draws, t2n
    tr, batch = MG.build(gan_type="ra-dragan")
    gan = hip_model_like(tr, gan_type="ra-dragan")

    def cu(a):
        return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    gp = {"alpha": cu(batch["gp"]["alpha"]), "eps": cu(batch["gp"]["eps"]), "aug": dev_draws(batch["gp"]["aug"])}
    d = gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]),
                   apply=False, gp_draws=gp)
    assert abs(d["d_loss"].item() - GOLD_GP["d_loss"]) <= 1e-4 * abs(GOLD_GP["d_loss"])
    assert abs(d["gp"].item() - GOLD_GP["gp"]) <= 1e-4 * abs(GOLD_GP["gp"])
    np.testing.assert_allclose(t2n(d["real_logits"]), GOLD_GP["real_logits"], rtol=1e-3, atol=1e-5)
    worst = 0.0
    for k in [f[len("dgrad_norm/"):] for f in GOLD_GP.files if f.startswith("dgrad_norm/")]:
        g = t2n(gan.store.vars[k].bg_grad).reshape(-1).astype(np.float64)
        gn = float(GOLD_GP["dgrad_norm/" + k])
        if gn < 1e-9:
            continue
        tol = 5e-2 if k.endswith("self_attention/gamma") else 2e-3
        assert abs(np.linalg.norm(g) - gn) <= tol * gn + 1e-7, (k, np.linalg.norm(g), gn)
        idx = MG.stable_indices(k, g.size)
        assert np.abs(g[idx] - GOLD_GP["dgrad_samp/" + k]).max() <= tol * gn + 1e-7, k


@pytest.mark.gpu
def test_hip_step_matches_golden():
    from tests.common import hip_model_like, dev_draws, t2n
    tr, batch = MG.build()
    gan = hip_model_like(tr)

    def cu(a):
        return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    hip0 = gan.store.export_arrays()
    B = batch["real"].shape[0]
    d = gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]),
                   apply=False)
    assert abs(d["d_loss"].item() - GOLD["d_loss"]) <= 1e-4 * abs(GOLD["d_loss"])
    np.testing.assert_allclose(t2n(d["real_logits"]), GOLD["real_logits"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(t2n(d["fake_logits"]), GOLD["fake_logits_d"], rtol=1e-3, atol=1e-5)

    def check(prefix, names):
        for k in names:
            if k.endswith("self_attention/f_conv/bias"):
                continue                                   # exactly zero in exact arithmetic
            g = t2n(gan.store.vars[k].bg_grad).reshape(-1).astype(np.float64)
            gn = float(GOLD[prefix + "_norm/" + k])
            tol = 5e-2 if k.endswith("self_attention/gamma") else 1e-3
            assert abs(np.linalg.norm(g) - gn) <= tol * gn + 1e-7, (k, np.linalg.norm(g), gn)
            idx = MG.stable_indices(k, g.size)
            assert np.abs(g[idx] - GOLD[prefix + "_samp/" + k]).max() <= tol * gn + 1e-7, k
    check("dgrad", [k[len("dgrad_norm/"):] for k in GOLD.files if k.startswith("dgrad_norm/")])
    gan.store.load_arrays(hip0, reset_ema=False)
    g = gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), apply=False)
    assert abs(g["g_loss"].item() - GOLD["g_loss"]) <= 1e-4 * abs(GOLD["g_loss"])
    assert abs(g["g_reg"].item() - GOLD["g_reg"]) <= 1e-4 * abs(GOLD["g_reg"])
    check("ggrad", [k[len("ggrad_norm/"):] for k in GOLD.files if k.startswith("ggrad_norm/")])
    gan.store.load_arrays(hip0, reset_ema=False)
    # one applied iteration
    gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]))
    gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]))
    st = gan.store.export_arrays()
    for k in [f[len("state_norm/"):] for f in GOLD.files if f.startswith("state_norm/")]:
        if k.endswith("self_attention/f_conv/bias"):
            continue
        v = st[k].reshape(-1).astype(np.float64)
        gn = float(GOLD["state_norm/" + k])
        assert abs(np.linalg.norm(v) - gn) <= 1e-4 * gn + 1e-7, k
        idx = MG.stable_indices(k, v.size)
        assert np.abs(v[idx] - GOLD["state_samp/" + k]).max() <= 1e-4 * max(gn, 1e-3) + 1e-6, k
