"""Which Python lines launch torch's own fill / copy kernels inside one training iteration?

    python tools/launch_sites.py [workload] [batch]

Runs two iterations of a bench workload under torch.profiler (with stacks) and prints, per source line of this package,
how many aten fill / zero / copy / cat / clone operators it issued in the second iteration (each is one small kernel
launch the HIP path does not need in principle: DESIGN section 7 item 2)."""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
img, ch, _, _ = bench.WORKLOADS[name]
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import main as M, model, scope as S  # noqa: E402
precision = "bf16" if name in bench.BF16_WORKLOADS else "fp32"
args = M.parse_args(["--gan_type", "hinge", "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B),
                     "--precision", precision], make_dirs=False)
gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=42)).build_model()
real = gan.synthetic_batch(B)
for _ in range(2):
    gan.train_step(real)
torch.cuda.synchronize()
# Python-level hooks (the profiler's stacks are empty for these operators on this build): every call of the wrapped
# functions from a line of this package is counted at that line
import traceback  # noqa: E402

count = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "biggan" in fr.filename and "tools" not in fr.filename:
            return "%s:%d" % (fr.filename.replace(root + "/", ""), fr.lineno)
    return "?"


def wrap(obj, name, label, cond=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if cond is None or cond(*a, **k):
            count[(label, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for fn in ("zeros", "zeros_like", "ones", "full", "full_like", "cat"):
    wrap(torch, fn, "torch." + fn)
for fn in ("zero_", "fill_", "copy_", "clone"):
    wrap(torch.Tensor, fn, "Tensor." + fn)
wrap(torch.Tensor, "contiguous", "Tensor.contiguous (copy)", lambda t, *a, **k: not t.is_contiguous())
gan.train_step(real)
torch.cuda.synchronize()
tot = collections.Counter()
for (op, st), n in count.items():
    tot[op] += n
print("calls per iteration:", dict(tot))
for (op, st), n in count.most_common(45):
    print("%4d  %-26s %s" % (n, op, st))
sys.exit(0)
