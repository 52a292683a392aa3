"""Generates tests/golden/step_img64_ch8.npz: outputs of the float64 oracle for one D op and one G op
(first-step gradients from the seeded initial state) and for one applied D+G iteration.

The reference itself cannot be run (TensorFlow is absent; see oracle/__init__.py), so these vectors
pin the ORACLE (regression fixture) and give the HIP path a second, frozen target.  Inputs are not
stored: they are regenerated from the seeds below (numpy PCG64 streams are stable across versions).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model as RM  # noqa: E402

CFG = dict(img_size=64, ch=8, z_dim=64, batch=2)
WEIGHT_SEED, BATCH_SEED = 42, 5
N_SAMPLE = 8


def build(**cfg_kw):
    cfg = RM.Config(img_size=CFG["img_size"], ch=CFG["ch"], z_dim=CFG["z_dim"], batch_size=CFG["batch"], **cfg_kw)
    tr = RM.Trainer(cfg, torch.float64, WEIGHT_SEED).build()
    RM.perturb_for_parity(tr.vs)
    for k, p in tr.g_params().items():
        tr.ema[k] = p.detach().clone()
    batch = RM.synthetic_batch(cfg, BATCH_SEED, CFG["batch"])
    return tr, batch


def sample_idx(name, n):
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))   # not used: hash() is salted
    return rng


def stable_indices(name, n, k=N_SAMPLE):
    seed = sum((i + 1) * ord(c) for i, c in enumerate(name)) % (2 ** 31)
    return np.random.default_rng(seed).integers(0, n, k)


def compute():
    tr, batch = build()
    out = {}
    d = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False)
    out["d_loss"] = np.array(d["d_loss"].item())
    out["real_logits"] = d["real_logits"].detach().numpy()
    out["fake_logits_d"] = d["fake_logits"].detach().numpy()
    for k, g in d["grads"].items():
        g = g.numpy().reshape(-1)
        out["dgrad_norm/" + k] = np.array(np.linalg.norm(g))
        out["dgrad_samp/" + k] = g[stable_indices(k, g.size)]
    tr.vs.state_updates.clear()
    g_ = tr.g_step(batch["z_g"], batch["aug_fake_g"], apply=False)
    out["g_loss"] = np.array(g_["g_loss"].item())
    out["g_adv"] = np.array(g_["g_adv"].item())
    out["g_reg"] = np.array(g_["g_reg"].item())
    out["fake_logits_g"] = g_["fake_logits"].detach().numpy()
    for k, g in g_["grads"].items():
        g = g.numpy().reshape(-1)
        out["ggrad_norm/" + k] = np.array(np.linalg.norm(g))
        out["ggrad_samp/" + k] = g[stable_indices(k, g.size)]
    tr.vs.state_updates.clear()
    # one applied iteration
    tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])
    it = tr.g_step(batch["z_g"], batch["aug_fake_g"])
    out["iter_g_loss"] = np.array(it["g_loss"].item())
    for k, v in tr.vs.export().items():
        v = v.reshape(-1)
        out["state_norm/" + k] = np.array(np.linalg.norm(v))
        out["state_samp/" + k] = v[stable_indices(k, v.size)]
    return out


def compute_gp(gan_type="ra-dragan"):
    """step_img64_ch8_radragan.npz: the D op of the reference's DEFAULT --gan_type (relativistic loss + DRAGAN gradient
    penalty, BigGAN.py:717-742): loss, penalty, logits and every first-step D gradient (norm + sampled elements)."""
    tr, batch = build(gan_type=gan_type)
    out = {}
    d = tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False, gp=batch["gp"])
    out["d_loss"] = np.array(d["d_loss"].item())
    out["gp"] = np.array(d["gp"].item())
    out["real_logits"] = d["real_logits"].detach().numpy()
    out["fake_logits_d"] = d["fake_logits"].detach().numpy()
    for k, g in d["grads"].items():
        g = g.numpy().reshape(-1)
        out["dgrad_norm/" + k] = np.array(np.linalg.norm(g))
        out["dgrad_samp/" + k] = g[stable_indices(k, g.size)]
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, res in (("step_img64_ch8.npz", compute()), ("step_img64_ch8_radragan.npz", compute_gp())):
        path = os.path.join(here, name)
        np.savez_compressed(path, **res)
        print("wrote", path, os.path.getsize(path), "bytes,", len(res), "arrays")
