import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd
from biggan_tensorflow_amd import main as M, model, scope as S, hip
mode = int(sys.argv[1]); iters = int(sys.argv[2])
from biggan_tensorflow_amd import functional as Fn; Fn.set_precision({0: "fp32", 1: "bf16-staged", 2: "bf16-staged", 3: "bf16"}[mode])
args = M.parse_args(["--gan_type", "hinge", "--img_size", "64", "--ch", "32", "--batch_size", "16"] + sys.argv[3:], make_dirs=False)
gan = model.BigGAN(args, store=S.VariableStore("cuda")).build_model()
real = gan.synthetic_batch(16)
t0 = time.time()
for i in range(iters):
    l = gan.train_step(real)
    if i % 25 == 0 or i == iters - 1:
        print("it %4d d %.4f g %.4f  |G| %.3f  (%.1fs)" % (i, l["d_loss"].item(), l["g_loss"].item(), float(gan.g_arena.params.norm()), time.time() - t0), flush=True)
assert torch.isfinite(gan.g_arena.params).all() and torch.isfinite(gan.d_arena.params).all()
img = gan.sample(B=4)
print("sample range", float(img.min()), float(img.max()), "finite", bool(torch.isfinite(img).all()))
