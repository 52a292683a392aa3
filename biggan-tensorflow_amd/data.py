"""Input pipeline of the reference (``utils.py:12-121``, ``BigGAN.py:768-787``) without TensorFlow:
file list + optional label file, PNG decoding, TF1 ``resize_images`` (legacy bilinear), random flip,
``x / 127.5 - 1``, shuffle-and-repeat batching with a prefetching worker thread that hands pinned host
batches to the GPU.  Host-side data preparation only: no arithmetic of the training step runs here.

Only what the reference's custom-dataset branch needs is provided: 8-bit non-interlaced PNG files (grey,
RGB, palette, with or without alpha) and ``.npy`` arrays ``[H, W, C]`` uint8; ``mnist`` / ``cifar10`` (Keras
downloads) and per-file sampling weights raise ``NotImplementedError``.
"""
import csv
import os
import queue
import struct
import threading
import zlib
from glob import glob

import numpy as np
import torch


# ------------------------------------------------------------------------------------------
# PNG (ISO/IEC 15948): 8-bit, non-interlaced
# ------------------------------------------------------------------------------------------
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _unfilter(raw, h, stride, bpp):
    try:                                                # C helper of libbiggan_hip.so (host code, no GPU involved)
        from . import hip
        out = np.empty((h, stride), np.uint8)
        if len(raw) < h * (stride + 1):
            raise ValueError("PNG: truncated image data")
        rc = hip.lib().bg_png_unfilter(bytes(raw), h, stride, bpp, out.ctypes.data_as(hip.c_void_p))
        if rc != 0:
            raise ValueError("PNG: " + hip.lib().bg_last_error().decode())
        return out
    except ImportError:                                 # library not built: pure-Python fallback (slow)
        return _unfilter_py(raw, h, stride, bpp)


def _unfilter_py(raw, h, stride, bpp):
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for r in range(h):
        ft = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32)
        pos += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 2:                                   # Up
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):                           # Sub / Average / Paeth: sequential along the row
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                left = cur[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = left
                elif ft == 3:
                    pred = (left + prev[i]) >> 1
                else:
                    ul = prev[i - bpp] if i >= bpp else 0
                    pred = _paeth(int(left), int(prev[i]), int(ul))
                cur[i] = (line[i] + pred) & 255
        else:
            raise ValueError("PNG: bad filter type %d" % ft)
        out[r] = cur
        prev = cur
    return out


def decode_png(data, channels=3):
    """tf.image.decode_png(contents, channels): uint8 [H, W, channels] (channels 1, 3 or 4)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, plte, trns = 8, [], None, None
    w = h = depth = ctype = interlace = None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if depth != 8 or interlace != 0:
        raise NotImplementedError("PNG: only 8-bit non-interlaced files are supported (depth %s, interlace %s)"
                                  % (depth, interlace))
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    px = _unfilter(zlib.decompress(b"".join(idat)), h, w * nch, nch).reshape(h, w, nch)
    if ctype == 3:
        rgb = plte[px[:, :, 0]]
        alpha = np.full((h, w, 1), 255, np.uint8)
        if trns is not None:
            lut = np.full(256, 255, np.uint8)
            lut[:len(trns)] = trns
            alpha = lut[px[:, :, 0]][:, :, None]
        px = np.concatenate([rgb, alpha], axis=2)
    elif ctype == 0:
        px = np.concatenate([px, px, px, np.full((h, w, 1), 255, np.uint8)], axis=2)
    elif ctype == 4:
        px = np.concatenate([px[:, :, :1]] * 3 + [px[:, :, 1:2]], axis=2)
    elif ctype == 2:
        px = np.concatenate([px, np.full((h, w, 1), 255, np.uint8)], axis=2)
    if channels == 4:
        return px
    if channels == 3:
        return px[:, :, :3]
    if channels == 1:                                   # TF converts RGB to grey with the Rec. 601 weights
        g = (0.299 * px[:, :, 0] + 0.587 * px[:, :, 1] + 0.114 * px[:, :, 2])
        return np.clip(np.rint(g), 0, 255).astype(np.uint8)[:, :, None]
    raise ValueError("decode_png: channels must be 1, 3 or 4")


def resize_bilinear_legacy(img, size):
    """tf.image.resize_images(img, [size, size]) of TF 1.x: bilinear, align_corners=False, no half-pixel
    centres: source coordinate = destination index * (in / out)."""
    img = np.asarray(img, np.float32)
    H, W = img.shape[:2]

    def axis(n_in, n_out):
        src = np.arange(n_out, dtype=np.float32) * (n_in / float(n_out))
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return lo, hi, (src - lo).astype(np.float32)
    y0, y1, fy = axis(H, size)
    x0, x1, fx = axis(W, size)
    top = img[y0][:, x0] * (1 - fx)[None, :, None] + img[y0][:, x1] * fx[None, :, None]
    bot = img[y1][:, x0] * (1 - fx)[None, :, None] + img[y1][:, x1] * fx[None, :, None]
    return top * (1 - fy)[:, None, None] + bot * fy[:, None, None]


# ------------------------------------------------------------------------------------------
# utils.py:12-121
# ------------------------------------------------------------------------------------------
class ImageData:
    """utils.py:12-38."""

    def __init__(self, load_size, channels, custom_dataset, flip, seed=0):
        self.load_size = load_size
        self.channels = channels
        self.custom_dataset = custom_dataset
        self.flip = flip
        self.rng = np.random.default_rng(seed)
        self._lock = threading.Lock()                   # image_processing runs on the loader's decode threads

    def image_processing(self, filename):
        if not self.custom_dataset:
            x_decode = np.asarray(filename)             # in-memory dataset: the array itself
        elif str(filename).endswith(".npy"):
            x_decode = np.load(filename)
        else:
            with open(filename, "rb") as f:
                x_decode = decode_png(f.read(), channels=self.channels)
        img = resize_bilinear_legacy(x_decode, self.load_size)
        if self.flip:                                    # tf.image.random_flip_left_right
            with self._lock:
                flip = self.rng.random() < 0.5
            if flip:
                img = img[:, ::-1]
        return (img / 127.5 - 1).astype(np.float32)

    def image_processing_with_labels(self, filename, label):
        return self.image_processing(filename), np.asarray(label, np.float32)


def read_labels(path):
    """utils.py:41-51: tab-separated ``filename  tag tag ...``."""
    labels = {}
    with open(path, 'r') as csvFile:
        reader = csv.reader(csvFile, delimiter='\t', quotechar='"')
        for row in reader:
            filename = row.pop(0)
            labels[filename] = list(map(float, row))
    return labels


def load_data(dataset_name, label_file, weight_file=None, ignore_missing=False, n_labels=None, root="./dataset"):
    """utils.py:79-121 (custom datasets)."""
    if dataset_name in ('mnist', 'cifar10'):
        raise NotImplementedError("dataset '%s' is a Keras download in the reference; no network here" % dataset_name)
    if weight_file:
        raise NotImplementedError("per-file sampling weights (--weight_file) are not implemented")
    x = sorted(glob(os.path.join(root, dataset_name, '*.*')))
    if label_file:
        labels = read_labels(label_file)
        used_labels = []
        for full in x:
            fn = os.path.basename(full)
            if fn not in labels:
                if ignore_missing:
                    used_labels.append([0.0] * n_labels)
                else:
                    raise RuntimeError("No label found for file " + fn)
            else:
                used_labels.append(labels[fn])
    else:
        used_labels = None
    return x, used_labels


class BatchLoader:
    """shuffle_and_repeat(dataset_num) + map_and_batch(batch_size, drop_remainder=True) +
    prefetch_to_device (BigGAN.py:776-781): an endless iterator of device batches.  One worker thread
    decodes ahead (queue depth 4); ``rank`` / ``world`` give each data-parallel rank a disjoint shard of
    every shuffled epoch."""

    def __init__(self, files, labels, batch_size, image_data, device, seed=0, rank=0, world=1, depth=4, workers=8):
        if len(files) < batch_size * world:
            raise ValueError("dataset has %d files, fewer than one global batch (%d)" % (len(files), batch_size * world))
        self.files, self.labels = list(files), labels
        self.batch_size, self.image_data, self.device = batch_size, image_data, torch.device(device)
        self._dev_index = 0
        if self.device.type == "cuda":
            self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.rank, self.world = rank, world
        self.rng = np.random.default_rng(seed)          # same seed on every rank: identical permutations
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)     # zlib, the C unfilter and numpy release the GIL
        self.q = queue.Queue(maxsize=depth)
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            if self.device.type == "cuda":
                # the current CUDA/HIP device is per THREAD and defaults to 0: without this, ranks 1..N-1 would create
                # a context (and pinned-allocator state) on GPU 0 from their loader threads
                torch.cuda.set_device(self._dev_index)
            while not self.stop.is_set():
                order = self.rng.permutation(len(self.files))
                per_step = self.batch_size * self.world
                for s in range(0, len(order) - per_step + 1, per_step):
                    idx = order[s + self.rank * self.batch_size: s + (self.rank + 1) * self.batch_size]
                    imgs = np.stack(list(self.pool.map(lambda i: self.image_data.image_processing(self.files[i]), idx)))
                    item = [torch.from_numpy(imgs)]
                    if self.labels is not None:
                        item.append(torch.tensor(np.asarray([self.labels[i] for i in idx], np.float32)))
                    if self.device.type == "cuda":
                        item = [t.pin_memory() for t in item]
                    while not self.stop.is_set():
                        try:
                            self.q.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if self.stop.is_set():
                        return
        except Exception as e:                          # surface worker failures in the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if isinstance(item, Exception):
            raise item
        out = [t.to(self.device, non_blocking=True) for t in item]
        return out[0] if self.labels is None else tuple(out)

    def close(self):
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(2)
        self.pool.shutdown(wait=False)
