"""HIP-event time of every implicit-GEMM launch of ONE training step, grouped by shape (bg_prof_dump).

    python tools/shapes.py img ch batch dump.tsv [main.py flags ...]      e.g.  128 96 256 /tmp/s.tsv --precision bf16

Prints the per-shape table (kernel symbol the launcher chose, launches, ms, share, TFLOP/s, algorithmic GB/s) and the
totals per op class (conv2d_fwd / dgrad / wgrad, deconv2d_*, attention)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import main as M, model, scope as S, hip  # noqa: E402

img, ch, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
out = sys.argv[4]
args = M.parse_args(["--gan_type", "hinge", "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B)] + sys.argv[5:],
                    make_dirs=False)
gan = model.BigGAN(args, store=S.VariableStore("cuda")).build_model()
real = gan.synthetic_batch(B)
for _ in range(2):
    gan.train_step(real)
torch.cuda.synchronize()
L = hip.lib()
L.bg_prof_reset()
L.bg_prof_enable(1)
gan.train_step(real)
torch.cuda.synchronize()
L.bg_prof_dump(out.encode())
L.bg_prof_enable(0)
agg = collections.OrderedDict()
cls = collections.OrderedDict()
for line in open(out).read().splitlines()[1:]:
    tag, kernel, nbytes, fl, ms = line.split("\t")
    a = agg.setdefault(tag, [0, 0.0, 0.0, 0.0, kernel])
    a[0] += 1
    a[1] += float(fl)
    a[2] += float(ms)
    a[3] += float(nbytes)
    c = cls.setdefault(tag.split(" ")[0], [0, 0.0, 0.0, 0.0])
    c[0] += 1
    c[1] += float(fl)
    c[2] += float(ms)
    c[3] += float(nbytes)
tot = sum(a[2] for a in agg.values())
print("total gemm ms %.2f  flops %.3e -> %.1f TF/s" % (tot, sum(a[1] for a in agg.values()),
                                                       sum(a[1] for a in agg.values()) / tot / 1e9))
for tag, c in sorted(cls.items(), key=lambda kv: -kv[1][2]):
    print("CLASS %-24s n=%3d ms=%7.3f (%4.1f%%) %7.1f TF/s %7.1f alg GB/s" % (tag, c[0], c[2], 100 * c[2] / tot,
                                                                             c[1] / c[2] / 1e9 if c[2] else 0,
                                                                             c[3] / c[2] / 1e6 if c[2] else 0))
for tag, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-62s n=%2d ms=%7.3f (%4.1f%%) %6.1f TF/s %6.0f GB/s  %s" % (tag, a[0], a[2], 100 * a[2] / tot,
                                                                      a[1] / a[2] / 1e9 if a[2] else 0,
                                                                      a[3] / a[2] / 1e6 if a[2] else 0, a[4]))
