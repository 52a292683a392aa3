import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import biggan_tensorflow_amd
from biggan_tensorflow_amd import main as M, model, scope as S
args = M.parse_args(["--gan_type", "hinge", "--img_size", "128", "--ch", "96", "--batch_size", "32", "--precision", "bf16"], make_dirs=False)
gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=42)).build_model()
real = gan.synthetic_batch(32)
import gc
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
print("gc mode:", mode, "thresholds", gc.get_threshold())
gc.callbacks.append(lambda phase, info: print("   gc %s gen %d collected %s" % (phase, info["generation"], info.get("collected"))) if phase == "stop" and info["generation"] == 2 else None)
for i in range(24):
    if i == 2 and mode == "freeze":
        gc.collect(); gc.freeze()
    if i == 2 and mode == "off":
        gc.disable()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gan.train_step(real)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step %2d host enqueue %.2f ms, total %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
