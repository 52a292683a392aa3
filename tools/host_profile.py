"""Where does the Python side of one training iteration spend its time?

    python tools/host_profile.py [batch]

cProfile over five eager iterations of config 3 (bf16-resident); prints the top functions by own time.  The enqueueing
thread must stay ahead of the GPU (27 ms per 32-image iteration): DESIGN section 6."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd  # noqa: E402,F401
from biggan_tensorflow_amd import main as M, model, scope as S  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = M.parse_args(["--gan_type", "hinge", "--img_size", "128", "--ch", "96", "--batch_size", str(B), "--precision", "bf16"],
                    make_dirs=False)
gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=42)).build_model()
real = gan.synthetic_batch(B)
for _ in range(3):
    gan.train_step(real)
torch.cuda.synchronize()
gan.settle_host()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    gan.train_step(real)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
