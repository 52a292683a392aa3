#!/usr/bin/env python
"""bench.py - BigGAN train-step images/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one D update + one G update (BigGAN.py:1061-1084, n_critic = 1) on a synthetic batch already resident
in HBM.

  Every N          the SAME workload: BASELINE config 3 (BigGAN-128, ch 96, bf16 - the north star's target) at its FIXED
                   global batch 256, 256 / N images per rank, "scaling": "strong" - so the driver's N = 1, 2, 4, 8 lines
                   are one series.  (--scaling weak keeps 32 images per rank instead.)
  N = 1 (default)  additionally carries `fp32_config2` (BASELINE config 2: ch 64, batch 64, fp32 - the reference's
                   precision - with its own roofline), the --g_regularization none / --da_policy "" companions of
                   SURVEY.md section 8(d), the 32-image share one of 8 ranks runs, and the CPU baseline.
  `--gpus N` without torchrun's environment starts the N ranks itself (one process per GPU, RCCL) before any GPU call
  and relays rank 0's line; under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0 with the throughput, the MFMA roofline of the convolution / attention GEMM launches
(timed with HIP events on their stream in a second pass over the same steps; algorithmic FLOPs only - the
regulariser's Gram matrices and dense layers are excluded) and a CPU baseline: the oracle (a torch-CPU restatement,
"port" - TensorFlow is not available) timed on this host's cores on a bounded sample.
"""
import argparse
import collections
import ctypes
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (img_size, ch, per-GPU batch, description)
    "c2": (128, 64, 64, "BigGAN-128 ch=64 batch=64/GPU fp32 (BASELINE config 2)"),
    "c1": (64, 32, 16, "plumbing: BigGAN-64 ch=32 batch=16 fp32 (BASELINE config 1)"),
    "c3fp32": (128, 96, 32, "BigGAN-128 ch=96 batch=32/GPU fp32 (config 3 shape, fp32 kernels)"),
    # bf16: bf16-resident activations + packed bf16 weights, fp32 accumulate / master weights / optimiser (--precision bf16)
    "c3": (128, 96, 32, "BigGAN-128 ch=96 bf16 (BASELINE config 3: global batch 256, 32/GPU over 8 GPUs)"),
    "c2bf16": (128, 64, 64, "BigGAN-128 ch=64 batch=64/GPU bf16 (config 2 shape)"),
    "c4": (256, 96, 32, "BigGAN-256 ch=96 bf16 + DiffAugment (BASELINE config 4: global batch 256, 32/GPU over 8 GPUs)"),
    "c5": (512, 128, 64, "BigGAN-512 ch=128 bf16 (BASELINE config 5: global batch 512, 64/GPU over 8 GPUs)"),
    # round-1 arithmetic for comparison: fp32 tensors, operands rounded to bf16 while staged (--precision bf16-staged)
    "c3staged": (128, 96, 32, "BigGAN-128 ch=96 batch=32/GPU, fp32 tensors with bf16-staged conv operands"),
}
BF16_WORKLOADS = ("c3", "c2bf16", "c4", "c5", "c3staged")
GLOBAL_BATCH = {"c3": 256, "c4": 256, "c5": 512}      # BASELINE.json: fixed global batch of the 8-GPU configs

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6     # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)


def step_flops_per_image(img, ch):
    """4*F_G + 8*F_D: F = 2 x MACs of every conv / transposed conv / dense /
    attention matmul of one forward pass."""
    def ru8(v):
        return (int(v) + 7) // 8 * 8
    counts_g = {64: [1, 1, 1, 1], 128: [1, 1, 1, 1, 1], 256: [1, 2, 1, 1, 1], 512: [1, 2, 1, 1, 2]}[img]
    sa_g = {64: 3, 128: 4, 256: 3, 512: 3}[img]
    counts_d = {64: [1, 1, 1, 1], 128: [1, 1, 1, 1, 1], 256: [1, 1, 1, 2, 1], 512: [1, 2, 1, 1, 2]}[img]
    sa_d = {64: 1, 128: 1, 256: 2, 512: 2}[img]
    depth = img.bit_length() - 2
    zd = 256
    split = zd // (depth - 1 + 3)
    first = zd - (depth - 1) * split

    def attn(C, hw):
        n = hw * hw
        m = n * (C * (C // 8) * 2 + C * (C // 2) + (C // 2) * C)       # four 1x1 convs
        m += n * (n // 4) * (C // 8 + C // 2)                            # QK^T + PV
        return m
    # generator
    nb = len(counts_g)
    c = ru8(int(ch * 2 ** (nb - 1)))
    fw = ru8((first) * 1.85)
    macs_g = first * fw + fw * 16 * c
    h = 4
    for bi, cnt in enumerate(counts_g):
        cout = ru8(int(ch * 2 ** (nb - bi - 1)))
        for _ in range(cnt):
            macs_g += 2 * split * c + 2 * split * cout                   # cond-BN beta/gamma dense
            macs_g += h * h * 16 * c * cout * 2                          # res1 k4s2 + skip k4s2
            h *= 2
            macs_g += h * h * 9 * cout * cout                            # res2 k3s1
            c = cout
        if bi + 1 == sa_g:
            macs_g += attn(c, h)
    macs_g += h * h * 9 * c * 3
    # discriminator
    macs_d = 0
    c = 3
    h = img
    for bi, cnt in enumerate(counts_d):
        cout = ru8(int(ch * 2 ** bi))
        for _ in range(cnt):
            h //= 2
            macs_d += h * h * 9 * c * cout * 2 + h * h * 9 * cout * cout
            c = cout
        if bi + 1 == sa_d:
            macs_d += attn(c, h)
    macs_d += 2 * h * h * 9 * c * c + c
    return 2.0 * (4 * macs_g + 8 * macs_d), 2.0 * macs_g, 2.0 * macs_d


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(img, ch, sample_batch, steps, note=lambda m: None):
    """Oracle (torch-CPU fp32 restatement) timed on this host's cores: images/sec."""
    import torch
    from oracle import ref_model as RM
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = RM.Config(img_size=img, ch=ch, batch_size=sample_batch)
    tr = RM.Trainer(cfg, torch.float32).build()
    note("cpu baseline: oracle built, %d threads" % cores)
    batch = RM.synthetic_batch(cfg, 1, sample_batch)
    t0 = time.time()
    for i in range(steps):
        tr.d_step(batch["real"], batch["z_d"], batch["aug_real"], batch["aug_fake_d"])
        note("cpu baseline: D step %d done (%.1fs)" % (i, time.time() - t0))
        tr.g_step(batch["z_g"], batch["aug_fake_g"])
        note("cpu baseline: G step %d done (%.1fs)" % (i, time.time() - t0))
    dt = time.time() - t0
    return {"value": round(sample_batch * steps / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d D+G iteration(s) of the same workload at batch %d, %.1f s (torch-CPU fp32 oracle; "
                      "TensorFlow, the reference's substrate, is not installed)" % (steps, sample_batch, dt)}


def resolve_workload(workload, world, scaling="", batch=0):
    """(workload name, per-GPU batch, "strong" | "weak") of a run.  The default workload at EVERY world size is BASELINE
    config 3 with its FIXED global batch 256 split into equal shards (strong scaling): N = 1 runs all 256
    images on one GPU, so the driver's N = 1, 2, 4, 8 lines form one series.  --scaling weak keeps the per-GPU batch of
    the workload table; workloads without a BASELINE global batch (c1, c2, ...) are weak by nature; --batch overrides
    the per-GPU batch."""
    name = workload or "c3"
    B = WORKLOADS[name][2]
    mode = scaling or ("strong" if name in GLOBAL_BATCH else "weak")
    if mode == "strong":
        gb = GLOBAL_BATCH.get(name, B * 8)
        if gb % world:
            raise SystemExit("global batch %d is not divisible by %d ranks" % (gb, world))
        B = gb // world
    if batch:
        B = batch
    return name, B, mode


def pmc_traffic(workload, B=None):
    """HBM bytes per implicit-GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    over this same command (profiles/README.md; counters cannot be read from inside the process): the passes taken at
    this batch size (`r*_pmc_<workload>_b<B>.json`), or at the workload's own batch (`r*_pmc_<workload>.json`); None when
    there is no counter run of this batch."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_b%d.json" % (workload, B)))) if B else []
    if not files and (B is None or B == WORKLOADS[workload][2]):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        d = json.load(fh)
    n = b = 0.0
    for k, e in d.items():
        if k == "_summary" or re.search(r"tn_kernel_bf16_tr<2>|tn16x?_kernel<2[,>]", k):
            continue                                   # (the regulariser's Gram launches are not conv / attention launches)
        if re.search(r"(\b(nn|tn)(16[xh]?)?_kernel|attn(16)?_(fwd|bwd))", k) and "hbm_read_bytes_per_launch" in e:
            n += e["launches"]
            b += e["launches"] * (e["hbm_read_bytes_per_launch"] + e.get("hbm_write_bytes_per_launch", 0.0))
    return (b / n if n else None), os.path.relpath(files[-1], ROOT)


ALGORITHMIC_TAGS = ("conv2d_", "deconv2d_", "attention")     # launches whose FLOPs SURVEY 8(d) counts


def read_prof_dump(path):
    """Rows of bg_prof_dump (tab-separated: tag, kernel, bytes, flops, ms)."""
    rows = []
    with open(path) as fh:
        next(fh)
        for line in fh:
            tag, kernel, nbytes, flops, ms = line.rstrip("\n").split("\t")
            rows.append((tag, kernel, float(nbytes), float(flops), float(ms)))
    return rows


def roofline_from_rows(rows, nsteps, peak, fpi, B, ms_per_step):
    """The roofline object of one workload from the per-launch HIP-event records of `nsteps` iterations.
    achieved = ALGORITHMIC FLOPs of one iteration (4 F_G + 8 F_D per image, SURVEY 8d - padded channels and recomputation
    are not counted) / the time its conv / transposed-conv / attention launches took; `dominant` = the kernel SYMBOL
    (what rocprofv3 lists) with the most time among them."""
    by_kernel = collections.OrderedDict()
    alg_ms = lib_fl = alg_bytes = all_ms = 0.0
    n_alg = 0
    for tag, kernel, nbytes, f, t in rows:
        all_ms += t
        if tag.startswith(ALGORITHMIC_TAGS):
            alg_ms += t
            lib_fl += f
            alg_bytes += nbytes
            n_alg += 1
            e = by_kernel.setdefault(kernel or tag.split(" ")[0], [0.0, 0.0, 0, 0.0])
            e[0] += t
            e[1] += f
            e[2] += 1
            e[3] += nbytes
    step_fl = fpi * B
    gemm_ms = alg_ms / nsteps if nsteps else 0.0
    achieved = step_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    dom_k, dom = max(by_kernel.items(), key=lambda kv: kv[1][0]) if by_kernel else ("", [0.0, 0.0, 0, 0.0])
    dom_tf = dom[1] / (dom[0] * 1e-3) / 1e12 if dom[0] > 0 else 0.0
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "kernel": "conv / transposed-conv implicit-GEMM launches (forward, input gradient, weight gradient) and the "
                      "fused attention: the launches whose FLOPs SURVEY 8(d) counts",
            "launches_per_step": int(n_alg // max(nsteps, 1)),
            "gemm_ms_per_step": round(gemm_ms, 3), "step_algorithmic_flops": step_fl,
            "gemm_flops_per_step_as_launched": lib_fl / max(nsteps, 1),
            "algorithmic_bytes_per_launch": round(alg_bytes / n_alg, 1) if n_alg else None,
            "algorithmic_bytes_per_step": alg_bytes / max(nsteps, 1),
            "all_gemm_family_ms_per_step": round(all_ms / max(nsteps, 1), 3),
            "dominant": {"kernel": dom_k, "launches_per_step": dom[2] // max(nsteps, 1),
                         "ms_per_step": round(dom[0] / max(nsteps, 1), 3), "achieved": round(dom_tf, 2),
                         "frac": round(dom_tf / peak, 4),
                         "algorithmic_bytes_per_launch": round(dom[3] / dom[2], 1) if dom[2] else None},
            "step_frac_of_peak": round(step_fl / (ms_per_step * 1e-3) / 1e12 / peak, 4)}


def roofline_pass(gan, real, nsteps, peak, fpi, B, ms_per_step, workload):
    """Second pass over the same steps with every GEMM-family launch bracketed by HIP events on its stream."""
    import torch
    from biggan_tensorflow_amd import hip
    L = hip.lib()
    L.bg_prof_reset()
    L.bg_prof_enable(1)
    for _ in range(nsteps):
        gan.train_step(real)
    torch.cuda.synchronize()
    fd, path = tempfile.mkstemp(suffix=".tsv")
    os.close(fd)
    L.bg_prof_dump(path.encode())
    L.bg_prof_reset()
    L.bg_prof_enable(0)
    rows = read_prof_dump(path)
    os.unlink(path)
    out = roofline_from_rows(rows, nsteps, peak, fpi, B, ms_per_step)
    traffic, traffic_src = pmc_traffic(workload, B)
    out["traffic"] = traffic
    out["traffic_unit"] = "HBM bytes per launch (rocprofv3 --pmc passes of this command; algorithmic_bytes_per_launch beside it)"
    out["traffic_source"] = traffic_src
    return out


def run_workload(name, B, a, world, rank, note, extra=(), steps=None, warmup=None, roofline=True, graph=False):
    """Build the model for one workload, time `steps` iterations, optionally the roofline pass.  Returns a dict."""
    import gc
    import torch
    from biggan_tensorflow_amd import main as M, model, scope as S
    img, ch, _, desc = WORKLOADS[name]
    bf16 = name in BF16_WORKLOADS
    precision = a.precision or ("bf16-staged" if name == "c3staged" else "bf16" if bf16 else "fp32")
    kw = dict(da_policy=a.da_policy, g_regularization=a.g_regularization)
    kw.update(dict(extra))
    argv = ["--gan_type", a.gan_type, "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B),
            "--da_policy", kw["da_policy"], "--g_regularization", kw["g_regularization"], "--n_labels", str(a.n_labels),
            "--precision", precision]
    args = M.parse_args(argv, make_dirs=False)
    # arithmetic type and peak follow the precision actually run, not the workload's name
    peak = FP32_MFMA_PEAK_TFLOPS if precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
    gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=42)).build_model()
    real = gan.synthetic_batch(B)
    steps = steps or a.steps
    warmup = a.warmup if warmup is None else warmup

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    if graph:
        gan.capture_graphs(B)
    note("model built: %s, batch %d/GPU%s" % (desc, B, " (HIP-graph replay)" if graph else ""))
    gan.settle_host()                  # (the training loop of main.py does the same: no generation-2 GC pauses mid-run)
    for i in range(warmup):
        gan.train_step(real)
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = gan.train_step(real)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / steps * 1e3
    value = B * world * steps / dt
    note("%s: timed %d steps: %.1f ms/step, %.1f images/sec" % (name, steps, ms_per_step, value))
    fpi, _, _ = step_flops_per_image(img, ch)
    out = {"workload": desc, "img_size": img, "ch": ch, "per_gpu_batch": B, "global_batch": B * world,
           "precision": precision, "dtype": "f32" if precision == "fp32" else "bf16",
           "storage_dtype": ("bf16 activations + packed bf16 conv weights; fp32 master weights, optimiser, statistics"
                             if precision == "bf16" else "fp32"),
           "da_policy": kw["da_policy"], "g_regularization": kw["g_regularization"],
           "value": round(value, 2), "ms_per_step": round(ms_per_step, 3), "steps": steps,
           "step_frac_of_peak": round(fpi * B / (ms_per_step * 1e-3) / 1e12 / peak, 4),
           "losses": {k: round(float(v.item()), 5) for k, v in losses.items()}}
    if roofline and not graph:
        out["roofline"] = roofline_pass(gan, real, min(steps, 3), peak, fpi, B, ms_per_step, name)
        note("%s: roofline pass done" % name)
    del gan, real, losses
    gc.unfreeze()                      # the frozen object graph of THIS model must be collectable before the next one
    gc.collect()
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside torchrun: start the N ranks (one process per GPU) through torch.distributed.run BEFORE
    this process touches a GPU, relay rank 0's JSON line (the children share our stdout) and return the launcher's exit
    code, non-zero if any rank failed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] --gpus %d without WORLD_SIZE: starting %d ranks: %s" % (n, n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(a, rank, world):
    """--dry_run: the rank plumbing of a run without the model (no GPU needed: gloo) - process group from the
    environment, workload resolution, the barrier / MAX-over-ranks timing and rank 0's line.  `value` is null: nothing
    was measured.  tests/test_parallel.py drives `bench.py --gpus 2 --dry_run` through the self-spawning launcher."""
    import torch
    name, B, scaling = resolve_workload(a.workload, world, a.scaling, a.batch)
    img, ch, _, desc = WORKLOADS[name]
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        torch.distributed.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "BigGAN-%d train-step images/sec" % img, "value": None, "unit": "images/sec",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dry_run": True,
                          "max_over_ranks_s": round(float(dt.item()), 4),
                          "config": {"workload": desc, "img_size": img, "ch": ch, "per_gpu_batch": B,
                                     "global_batch": B * world, "parallelism": "dp%d" % world}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", type=str, default="", choices=[""] + sorted(WORKLOADS),
                    help="default: c3 = BASELINE config 3 at its fixed global batch 256 split over the ranks")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--scaling", type=str, default="", choices=["", "strong", "weak"],
                    help="strong (default for c3 / c4 / c5) = BASELINE's fixed global batch split over the ranks; "
                         "weak = per-GPU batch fixed")
    ap.add_argument("--da_policy", type=str, default="full")
    ap.add_argument("--g_regularization", type=str, default="ortho_cosine")
    ap.add_argument("--n_labels", type=int, default=0, help="class-conditional variant: synthetic one-hot labels")
    ap.add_argument("--gan_type", type=str, default="hinge",
                    help="BASELINE's configs use hinge; e.g. ra-dragan (the reference's default) adds the gradient penalty")
    ap.add_argument("--precision", type=str, default="", help="override: fp32 | bf16-staged | bf16")
    ap.add_argument("--graph", action="store_true", help="replay the iteration from captured HIP graphs (N=1 only)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_companions", action="store_true",
                    help="N = 1 default run: only the headline (skip fp32_config2, the two SURVEY 8(d) companions and the "
                         "32-image share)")
    ap.add_argument("--dry_run", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))      # (nothing in this process has touched a GPU yet)

    t_start = time.perf_counter()
    import torch
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import parallel

    rank, world, local = parallel.init_from_env(backend="gloo" if a.dry_run else None)
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.dry_run:
        return dry_run(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local % torch.cuda.device_count())   # (ranks share a card only in gloo rehearsals)

    def note(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    default_run = a.workload == "" and not a.batch and not a.scaling and not a.precision
    name, B, scaling = resolve_workload(a.workload, world, a.scaling, a.batch)
    img, ch, _, desc = WORKLOADS[name]
    if a.graph and world != 1:
        raise SystemExit("--graph is a single-process option (RCCL collectives cannot be captured on this stack: "
                         "tools/rccl_graph_probe.py aborts)")
    res = run_workload(name, B, a, world, rank, note, roofline=not a.no_roofline, graph=a.graph)

    companions = fp32_c2 = per32 = None
    if default_run and world == 1 and not a.no_companions and not a.graph:
        # the two companions SURVEY 8(d) asks for, one rank's 32-image share of the 8-GPU run, and BASELINE config 2
        # (the reference's precision) with its own roofline
        companions = {}
        for key, extra in (("g_regularization_none", {"g_regularization": "none"}), ("da_policy_empty", {"da_policy": ""})):
            c = run_workload("c3", B, a, world, rank, note, extra=extra.items(), steps=min(a.steps, 5), warmup=1,
                             roofline=False)
            companions[key] = {"value": c["value"], "ms_per_step": c["ms_per_step"]}
        p32 = run_workload("c3", 32, a, world, rank, note, steps=min(a.steps, 10), warmup=2, roofline=False)
        per32 = {"value": p32["value"], "ms_per_step": p32["ms_per_step"],
                 "note": "one rank's work of the 8-GPU run (32 images per step), without collectives"}
        note("fp32_config2: BASELINE config 2 (BigGAN-128 ch=64 batch=64 fp32) ...")
        c2 = run_workload("c2", 64, a, world, rank, note, steps=min(a.steps, 10), warmup=2, roofline=not a.no_roofline)
        fp32_c2 = {"metric": "BigGAN-128 ch=64 batch=64 fp32 train-step images/sec (BASELINE config 2)", "n_gpus": 1,
                   "value": c2["value"], "unit": "images/sec", "ms_per_step": c2["ms_per_step"], "steps": c2["steps"],
                   "dtype": "f32", "step_frac_of_peak": c2["step_frac_of_peak"], "roofline": c2.get("roofline"),
                   "losses": c2["losses"]}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sb = {64: 32, 128: 8 if ch <= 64 else 4, 256: 2}.get(img, 1)      # ~10-30 s of CPU work on 16 cores
        note("cpu baseline (oracle, batch %d) ..." % sb)
        cpu = cpu_baseline(img, ch, sb, 2 if img <= 128 else 1, note)
        note("cpu baseline done")

    if rank == 0:
        out = {
            "metric": "BigGAN-128 train-step images/sec" if img == 128 else "BigGAN-%d train-step images/sec" % img,
            "value": res["value"], "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic",
            "config": {"workload": desc, "img_size": img, "ch": ch, "per_gpu_batch": B, "global_batch": B * world,
                       "da_policy": a.da_policy, "g_regularization": a.g_regularization, "n_labels": a.n_labels,
                       "gan_type": a.gan_type, "hip_graph": bool(a.graph), "precision": res["precision"],
                       "storage_dtype": res["storage_dtype"], "parallelism": "dp%d" % world},
            "losses": res["losses"], "step_frac_of_peak": res["step_frac_of_peak"],
        }
        if scaling == "strong":
            out["config"]["scaling_series"] = ("every --gpus N line of this workload keeps the global batch %d: "
                                               "N = 1 runs it on one GPU" % (B * world))
        if "roofline" in res:
            out["roofline"] = res["roofline"]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if companions is not None:
            out["companions"] = companions
        if per32 is not None:
            out["per_gpu_share_32"] = per32
        if fp32_c2 is not None:
            out["fp32_config2"] = fp32_c2
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
