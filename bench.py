#!/usr/bin/env python
"""bench.py - BigGAN train-step images/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one D update + one G update (BigGAN.py:1061-1084, n_critic = 1) on a synthetic
class-free batch already resident in HBM.  The default workload is BASELINE config 2
(BigGAN-128, ch=64, batch 64 per GPU, fp32); per-GPU work is fixed as N grows (weak scaling).

Prints ONE JSON line on rank 0 with the throughput, the MFMA roofline of the dominant kernel
family (conv / transposed conv / matmul implicit-GEMM launches, timed with HIP events on their
stream in a second pass over the same steps) and a CPU baseline: the oracle (a torch-CPU
restatement, "port" - TensorFlow is not available) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
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
    "c3": (128, 96, 32, "BigGAN-128 ch=96 batch=32/GPU (256 over 8 GPUs) bf16 (BASELINE config 3)"),
    "c2bf16": (128, 64, 64, "BigGAN-128 ch=64 batch=64/GPU bf16 (config 2 shape)"),
    "c4": (256, 96, 32, "BigGAN-256 ch=96 batch=32/GPU (256 over 8 GPUs) bf16 + DiffAugment (BASELINE config 4)"),
    "c5": (512, 128, 64, "BigGAN-512 ch=128 batch=64/GPU (512 over 8 GPUs) bf16 (BASELINE config 5)"),
    # round-1 arithmetic for comparison: fp32 tensors, operands rounded to bf16 while staged (--precision bf16-staged)
    "c3staged": (128, 96, 32, "BigGAN-128 ch=96 batch=32/GPU, fp32 tensors with bf16-staged conv operands"),
}
BF16_WORKLOADS = ("c3", "c2bf16", "c4", "c5", "c3staged")

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)


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


def pmc_traffic(workload):
    """HBM bytes per implicit-GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    over this same command (profiles/README.md; counters cannot be read from inside the process)."""
    import glob
    import re
    base = {"c2bf16": "c2", "c3fp32": "c3"}.get(workload, workload)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)) or
                   glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % base)))
    if not files or not files[-1].endswith("_pmc_%s.json" % workload):
        return None, None
    with open(files[-1]) as fh:
        d = json.load(fh)
    n = b = 0.0
    for k, e in d.items():
        if re.search(r"\b((nn|tn)_kernel|attn_(fwd|bwd))", k) and "hbm_read_bytes_per_launch" in e:
            n += e["launches"]
            b += e["launches"] * (e["hbm_read_bytes_per_launch"] + e.get("hbm_write_bytes_per_launch", 0.0))
    return (b / n if n else None), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", type=str, default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--da_policy", type=str, default="full")
    ap.add_argument("--g_regularization", type=str, default="ortho_cosine")
    ap.add_argument("--n_labels", type=int, default=0, help="class-conditional variant: synthetic one-hot labels")
    ap.add_argument("--gan_type", type=str, default="hinge",
                    help="BASELINE's configs use hinge; e.g. ra-dragan (the reference's default) adds the gradient penalty")
    ap.add_argument("--precision", type=str, default="", help="override: fp32 | bf16-staged | bf16")
    ap.add_argument("--graph", action="store_true", help="replay the iteration from captured HIP graphs (N=1 only)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    a = ap.parse_args()

    t_start = time.perf_counter()
    import torch
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import hip, main as M, model, parallel, scope as S
    import ctypes

    rank, world, local = parallel.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local % torch.cuda.device_count())   # (ranks share a card only in gloo rehearsals)

    img, ch, B, desc = WORKLOADS[a.workload]
    if a.batch:
        B = a.batch
    argv = ["--gan_type", a.gan_type, "--img_size", str(img), "--ch", str(ch), "--batch_size", str(B),
            "--da_policy", a.da_policy, "--g_regularization", a.g_regularization, "--n_labels", str(a.n_labels)]
    bf16 = a.workload in BF16_WORKLOADS
    precision = a.precision or ("bf16-staged" if a.workload == "c3staged" else "bf16" if bf16 else "fp32")
    argv += ["--precision", precision]
    args = M.parse_args(argv, make_dirs=False)
    peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=42)).build_model()
    real = gan.synthetic_batch(B)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    if a.graph:
        if world != 1:
            raise SystemExit("--graph is a single-process option")
        gan.capture_graphs(B)
        a.no_roofline = True            # per-launch HIP events cannot be recorded inside a replayed graph
    note("model built: %s%s" % (desc, " (HIP-graph replay)" if a.graph else ""))
    for i in range(a.warmup):
        gan.train_step(real)
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = gan.train_step(real)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = B * world * a.steps / dt
    note("timed %d steps: %.1f ms/step, %.1f images/sec" % (a.steps, ms_per_step, value))

    roof = None
    if not a.no_roofline:
        L = hip.lib()
        L.bg_prof_reset()
        L.bg_prof_enable(1)
        nprof = min(a.steps, 3)
        for _ in range(nprof):
            gan.train_step(real)
        torch.cuda.synchronize()
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        L.bg_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        L.bg_prof_enable(0)
        note("roofline pass done")
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        fpi, fg, fd = step_flops_per_image(img, ch)
        traffic, traffic_src = pmc_traffic(a.workload)
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src,
                "kernel": ("bg::nn16_kernel / bg::tn16_kernel (bf16-resident MFMA implicit GEMM: conv, deconv) + "
                           "the fp32-tensor kernels for the image layers, dense layers and attention") if bf16 else
                          "bg::nn_kernel / bg::tn_kernel (fp32 MFMA implicit GEMM: conv, deconv, dense) + bg::attn_* (fused attention)",
                "launches_per_step": int(n.value // nprof), "gemm_ms_per_step": round(ms.value / nprof, 3),
                "gemm_flops_per_step": fl.value / nprof,
                "step_algorithmic_flops": fpi * B,
                "step_frac_of_peak": round(fpi * B / (ms_per_step * 1e-3) / 1e12 / peak, 4)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sb = {64: 32, 128: 8, 256: 2}.get(img, 1)      # ~10-20 s of CPU work on 16 cores
        note("cpu baseline (oracle, batch %d) ..." % sb)
        cpu = cpu_baseline(img, ch, sb, 2 if img <= 128 else 1, note)
        note("cpu baseline done")

    if rank == 0:
        out = {
            "metric": "BigGAN-128 train-step images/sec" if img == 128 else "BigGAN-%d train-step images/sec" % img,
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": desc, "img_size": img, "ch": ch, "per_gpu_batch": B, "global_batch": B * world,
                       "da_policy": a.da_policy, "g_regularization": a.g_regularization, "n_labels": a.n_labels, "gan_type": a.gan_type, "hip_graph": bool(a.graph), "precision": precision,
                       "parallelism": "dp%d" % world},
            "losses": {k: round(float(v.item()), 5) for k, v in losses.items()},
        }
        if roof is not None:
            out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
