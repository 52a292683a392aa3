import sys, time, torch, ctypes, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd
from biggan_tensorflow_amd import main as M, model, scope as S, ops, hip
img, ch, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
out = sys.argv[4]
import os
if os.environ.get("BF16"):
    from biggan_tensorflow_amd import functional as Fn; Fn.set_precision("bf16-staged")
args = M.parse_args(["--gan_type","hinge","--img_size",str(img),"--ch",str(ch),"--batch_size",str(B)] + sys.argv[5:], make_dirs=False)
gan = model.BigGAN(args, store=S.VariableStore("cuda")).build_model()
real = gan.synthetic_batch(B)
for _ in range(2): gan.train_step(real)
torch.cuda.synchronize()
L = hip.lib(); L.bg_prof_reset(); L.bg_prof_enable(1)
gan.train_step(real); torch.cuda.synchronize()
L.bg_prof_dump(out.encode()); L.bg_prof_enable(0)
agg = collections.OrderedDict()
for line in open(out).read().splitlines()[1:]:
    tag, fl, ms = line.rsplit(",", 2)
    a = agg.setdefault(tag, [0, 0.0, 0.0]); a[0]+=1; a[1]+=float(fl); a[2]+=float(ms)
tot = sum(a[2] for a in agg.values())
print("total gemm ms %.2f  flops %.3e -> %.1f TF/s" % (tot, sum(a[1] for a in agg.values()), sum(a[1] for a in agg.values())/tot/1e9))
for tag, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-62s n=%2d ms=%7.3f (%4.1f%%) %6.1f TF/s" % (tag, a[0], a[2], 100*a[2]/tot, a[1]/a[2]/1e9 if a[2] else 0))
