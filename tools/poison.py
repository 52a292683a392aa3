"""Uninitialised-read probe: fill PyTorch's caching allocator with NaN-filled blocks of many sizes, free them, then run
training iterations - any kernel that reads memory it (or a predecessor) never wrote now meets NaNs instead of the zeros
of a fresh allocation.  Reports the first non-finite state tensors / losses.

    python tools/poison.py [precision] [img] [ch] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import main as M, model, scope as S  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4


def poison(total_mb=3000):
    blocks = []
    sizes = [256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 4 << 20, 16 << 20, 64 << 20]
    per = total_mb * (1 << 20) // len(sizes)
    for sz in sizes:
        n = max(1, min(4096, per // (sz * 4)))
        for _ in range(n):
            blocks.append(torch.full((sz,), float("nan"), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    del blocks


args = M.parse_args(["--gan_type", "hinge", "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B), "--z_dim", "64",
                     "--precision", prec], make_dirs=False)
gan = model.BigGAN(args, store=S.VariableStore("cuda", seed=5)).build_model()
real = gan.synthetic_batch(B)
bad_any = False
for it in range(3):
    poison()
    losses = gan.train_step(real)
    torch.cuda.synchronize()
    vals = {k: float(v.item()) for k, v in losses.items()}
    bad = [k for k, v in gan.state_tensors().items() if not bool(torch.isfinite(v.float()).all())]
    print("iteration", it, vals, "non-finite state tensors:", len(bad), bad[:12], flush=True)
    bad_any = bad_any or bool(bad) or any(v != v for v in vals.values())
print("RESULT", "UNINITIALISED READ SUSPECTED" if bad_any else "clean")
