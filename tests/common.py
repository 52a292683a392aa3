"""Shared helpers for the parity tests: build the oracle and the HIP model from identical weights."""
import numpy as np
import torch

from oracle import ref_model as RM


def make_args(**kw):
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import main as M
    argv = ["--gan_type", "hinge"]
    for k, v in kw.items():
        argv += ["--" + k, str(v)]
    return M.parse_args(argv, make_dirs=False)


def oracle_trainer(img_size, ch, z_dim=256, batch=2, dtype=torch.float64, seed=42, perturb=True, **cfg_kw):
    cfg = RM.Config(img_size=img_size, ch=ch, z_dim=z_dim, batch_size=batch, **cfg_kw)
    tr = RM.Trainer(cfg, dtype, seed).build()
    if perturb:
        RM.perturb_for_parity(tr.vs)
        for k, p in tr.g_params().items():
            tr.ema[k] = p.detach().clone()
    return tr


def hip_model_like(tr, **flag_kw):
    """HIP model with the oracle trainer's weights / state loaded by TF variable name."""
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import model, scope as S
    cfg = tr.cfg
    flags = dict(img_size=cfg.img_size, ch=cfg.ch, z_dim=cfg.z_dim, batch_size=cfg.batch_size,
                 g_regularization=cfg.g_regularization, da_policy=cfg.da_policy or "none_")
    flags.update(flag_kw)
    if not cfg.da_policy:
        flags["da_policy"] = ""
    argv = ["--gan_type", "hinge"]
    for k, v in flags.items():
        argv += ["--" + k, str(v)]
    from biggan_tensorflow_amd import main as M
    args = M.parse_args(argv, make_dirs=False)
    if cfg.extension_32:
        args.extension_32 = True
    store = S.VariableStore("cuda")
    gan = model.BigGAN(args, device="cuda", store=store)
    gan.build_model()
    store.load_arrays({k: v.astype(np.float32) for k, v in tr.vs.export().items()})
    return gan


def dev_draws(d):
    from biggan_tensorflow_amd.DiffAugment import draws_to_device
    return draws_to_device(d, "cuda")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def t2n(t):
    return t.detach().cpu().numpy()
