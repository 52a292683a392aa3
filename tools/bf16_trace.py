"""Where does the bf16-resident product leave the bf16-rounded oracle?  Runs the D op (or G op) on both sides with the
activation probes on and prints, per activation site in call order, the relative L2 distance of the pre-activations
(oracle.ref_ops.ROUND on / off).  A clean emulation of the product's rounding points shows ~1e-3 everywhere; the first
site that jumps names the layer whose arithmetic differs.

    python tools/bf16_trace.py [img] [ch] [batch] [d|g]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_model as RM, ref_ops as R  # noqa: E402
from tests.common import oracle_trainer, hip_model_like, dev_draws, t2n  # noqa: E402
from tests.test_gpu_step import _hip_pre  # noqa: E402
from biggan_tensorflow_amd import functional as Fn  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
which = sys.argv[4] if len(sys.argv) > 4 else "d"


def cu(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


tr = oracle_trainer(img, ch, 64, B)
gan = hip_model_like(tr, precision="bf16")
batch = RM.synthetic_batch(tr.cfg, 29, B)


def run_o():
    if which == "d":
        return tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"], apply=False)
    return tr.g_step(batch["z_g"], batch["aug_fake_g"], apply=False)


def run_h():
    if which == "d":
        return gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]),
                          apply=False)
    return gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), apply=False)


Fn.KinkProbe.sites = []
ho = run_h()
sites, Fn.KinkProbe.sites = Fn.KinkProbe.sites, None
recs = {}
for mode in (True, False):
    tr.vs.state_updates.clear()
    R.ROUND.on = mode
    R.KINK.record = []
    ro = run_o()
    recs[mode], R.KINK.record = R.KINK.record, None
    R.ROUND.on = False
    if "fake" in ro and "fake" in ho:
        a, b = t2n(ho["fake"]).astype(np.float64), ro["fake"].detach().numpy()
        print("ROUND=%s  fake image rel err %.3e" % (mode, np.linalg.norm(a - b) / np.linalg.norm(b)))
    for k in ("d_loss", "g_loss"):
        if k in ro:
            print("ROUND=%s  %s product %.6f oracle %.6f" % (mode, k, ho[k].item(), ro[k].item()))
print("%-58s %12s %12s" % ("activation site (call order)", "vs ROUNDED", "vs float64"))
h_by, order = {}, []
for st in sites:
    name = st[0][:-len("/alpha")] if st[0] else "?"
    if name not in h_by:
        order.append(name)
    h_by.setdefault(name, []).append(_hip_pre(st))
for name in order:
    ph = np.concatenate(h_by[name], axis=0)
    row = []
    for mode in (True, False):
        calls = [x.double().numpy() for sc, x in recs[mode] if sc == name]
        po = np.concatenate(calls, axis=0) if calls else np.zeros(0)
        row.append(float(np.linalg.norm(ph - po) / max(np.linalg.norm(po), 1e-30)) if po.shape == ph.shape else float("nan"))
    print("%-58s %12.3e %12.3e" % (name, row[0], row[1]))

# sign disagreements per site (what the kink synchronisation of the parity tests flips in the oracle)
print("sign disagreements (ROUNDED oracle):")
tot = 0
for name in order:
    ph = np.concatenate(h_by[name], axis=0)
    calls = [x.double().numpy() for sc, x in recs[True] if sc == name]
    po = np.concatenate(calls, axis=0)
    m = (ph > 0) != (po > 0)
    n = int(m.sum())
    tot += n
    if n:
        rms = float(np.sqrt(np.mean(po * po)))
        print("  %-56s %7d of %9d   worst |pre| / rms %.3e" % (name, n, m.size, float(np.abs(po[m]).max()) / rms))
print("  total", tot)

# gradient tensors in creation (= forward) order: relative L2 distance and projection on the rounded oracle's gradient
# (kinks NOT synchronised here: sign disagreements above show up as a slightly lower projection)
R.ROUND.on = True
tr.vs.state_updates.clear()
ro = run_o()
R.ROUND.on = False
print("%-66s %9s %9s %9s" % ("gradient tensor", "elements", "rel L2", "projection"))
for k in gan.store.vars:
    if k not in ro["grads"]:
        continue
    gr = ro["grads"][k].numpy().astype(np.float64)
    if np.linalg.norm(gr) < 1e-12:
        continue
    got = t2n(gan.store.vars[k].bg_grad).astype(np.float64)
    print("%-66s %9d %9.3e %9.4f" % (k, gr.size, np.linalg.norm(got - gr) / np.linalg.norm(gr), (got * gr).sum() / (gr * gr).sum()))
