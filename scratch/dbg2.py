import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_model as RM
from tests.common import oracle_trainer, hip_model_like, dev_draws, rel_err, t2n
def cu(a): return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
img, ch, zd, B = [int(a) for a in sys.argv[1:5]]
tr = oracle_trainer(img, ch, zd, B)
gan = hip_model_like(tr)
batch = RM.synthetic_batch(tr.cfg, 9, B)
ro = tr.g_step(batch["z_g"], batch["aug_fake_g"], apply=False)
ho = gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), apply=False)
print("g_adv", ho["g_adv"].item(), ro["g_adv"].item(), "reg", ho["g_reg"].item(), ro["g_reg"].item())
print("fake", rel_err(t2n(ho["fake"]), ro["fake"].detach().numpy()), "logits", rel_err(t2n(ho["fake_logits"]), ro["fake_logits"].detach().numpy()))
for k, g in ro["grads"].items():
    e = rel_err(t2n(gan.store.vars[k].bg_grad), g.numpy())
    print("%-70s %.3e  |g|=%.3e" % (k, e, np.linalg.norm(g.numpy())))
