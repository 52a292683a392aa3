import sys, os, time, tempfile, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biggan_tensorflow_amd
from biggan_tensorflow_amd import data as D, utils
tmp = tempfile.mkdtemp()
folder = os.path.join(tmp, "dataset", "toy"); os.makedirs(folder)
rng = np.random.default_rng(0)
base = rng.uniform(-1, 1, (1, 160, 160, 3)).astype(np.float32)
for i in range(256):
    utils.save_images(np.clip(base + rng.normal(0, 0.05, base.shape), -1, 1).astype(np.float32), [1, 1], os.path.join(folder, "%03d.png" % i))
files, _ = D.load_data("toy", "", root=os.path.join(tmp, "dataset"))
print("file size", os.path.getsize(files[0]))
t0 = time.time(); img = D.ImageData(128, 3, True, True).image_processing(files[0]); print("one image %.2f ms" % ((time.time() - t0) * 1e3))
for workers in (1, 4, 8, 16):
    ld = D.BatchLoader(files, None, 64, D.ImageData(128, 3, True, True), "cuda" if torch.cuda.is_available() else "cpu", workers=workers)
    next(ld); t0 = time.time(); n = 0
    for _ in range(12): b = next(ld); n += b.shape[0]
    dt = time.time() - t0; ld.close()
    print("workers %2d: %.0f images/s" % (workers, n / dt))
