"""Which call sites still run a stand-alone elementwise sum (functional.add) in one training iteration, by tensor shape.
    python tools/who_adds.py [img] [ch] [batch]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import main as M, model, scope as S, functional as Fn  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
args = M.parse_args(["--gan_type", "hinge", "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B), "--precision", "bf16"],
                    make_dirs=False)
gan = model.BigGAN(args, store=S.VariableStore("cuda", seed=5)).build_model()
real = gan.synthetic_batch(B)
gan.train_step(real)
seen = collections.Counter()
orig = Fn.add


def add(a, b, out=None):
    fr = [f for f in traceback.extract_stack()[:-1] if "biggan" in f.filename][-3:]
    seen[(tuple(a.shape), str(a.dtype), " < ".join("%s:%d" % (f.name, f.lineno) for f in reversed(fr)))] += 1
    return orig(a, b, out)


Fn.add = add
gan.train_step(real)
torch.cuda.synchronize()
for (shape, dtp, who), n in sorted(seen.items(), key=lambda kv: -kv[1] * torch.tensor(kv[0][0]).prod().item()):
    print(n, shape, dtp, who)
