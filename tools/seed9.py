"""fp32 plumbing config (64^2 / ch 32 / batch 16), one batch seed: G-op gradients of product vs float64 oracle after kink
synchronisation - the tensors that differ most, and the max-pool near-ties of the attention blocks (oracle side).

    python tools/seed9.py [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_model as RM, ref_ops as R  # noqa: E402
from tests.common import oracle_trainer, hip_model_like, dev_draws, t2n  # noqa: E402
from tests import test_gpu_step as TS  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9
tr = oracle_trainer(64, 32, 256, 16)
gan = hip_model_like(tr)
batch = RM.synthetic_batch(tr.cfg, seed, 16)


def cu(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


# near-ties of the oracle's 2x2 max pools: gap between the two largest values of a window relative to the tensor's rms
gaps = []
orig = R.max_pooling


def probe(x):
    n, h, w, c = x.shape
    v = x.detach().reshape(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c, 4)
    top = torch.topk(v, 2, dim=-1).values
    gap = (top[..., 0] - top[..., 1]) / x.detach().pow(2).mean().sqrt()
    gaps.append((float(gap.min()), int((gap < 1e-6).sum()), int((gap < 1e-5).sum()), gap.numel()))
    return orig(x)


R.max_pooling = probe
for tag, ro_fn, ho_fn in (
        ("D op", lambda: tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False),
         lambda: gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]),
                            dev_draws(batch["aug_fake_d"]), apply=False)),
        ("G op", lambda: tr.g_step(batch["z_g"], batch["aug_fake_g"], apply=False),
         lambda: gan.g_step(16, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), apply=False))):
    hip0 = gan.store.export_arrays()
    del gaps[:]
    ro, ho, flips = TS._kink_sync(tr, ro_fn, ho_fn)
    print(tag, "kink elements flipped:", TS.EXEMPT["kink_elements"], " max-pool windows (min gap / rms, < 1e-6, < 1e-5, of):",
          gaps[:4])
    errs = {}
    for k, g in ro["grads"].items():
        gr = g.numpy().astype(np.float64)
        if np.linalg.norm(gr) < 1e-12:
            continue
        got = t2n(gan.store.vars[k].bg_grad).astype(np.float64)
        errs[k] = (float(np.linalg.norm(got - gr) / np.linalg.norm(gr)), float((got * gr).sum() / (gr * gr).sum()), gr.size)
    for k, (e, p, n) in sorted(errs.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %-60s rel %.3e  projection %.6f  n=%d" % (k, e, p, n))
    tr.vs.state_updates.clear()
    gan.store.load_arrays(hip0, reset_ema=False)
