#!/usr/bin/env python
"""bench.py - BigGAN train-step images/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one D update + one G update (BigGAN.py:1061-1084, n_critic = 1) on a synthetic batch already resident
in HBM.

  N = 1 (default)  headline = BASELINE config 2 (BigGAN-128, ch 64, batch 64, fp32: the reference's precision), plus a
                   "target" object: BASELINE config 3 (BigGAN-128, ch 96, bf16) at its GLOBAL batch 256 on this one GPU -
                   the 1-GPU point of the strong-scaling curve - with its --g_regularization none and --da_policy ""
                   companions.
  N > 1            config 3 at FIXED global batch 256 (256 / N images per rank), "scaling": "strong"
                   (--scaling weak keeps 32 images per rank instead).

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
    """(workload name, per-GPU batch, "strong" | "weak") of a run.  One GPU: config 2 unless told otherwise.  N > 1:
    BASELINE config 3 with its FIXED global batch 256 split into equal shards (strong scaling); --scaling
    weak keeps the per-GPU batch of the workload table; --batch overrides the per-GPU batch."""
    name = workload or ("c2" if world == 1 else "c3")
    B = WORKLOADS[name][2]
    mode = "weak"
    if world > 1:
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
    fd, path = tempfile.mkstemp(suffix=".csv")
    os.close(fd)
    L.bg_prof_dump(path.encode())
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    L.bg_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    L.bg_prof_enable(0)
    by_tag = collections.OrderedDict()
    alg_ms = alg_fl = all_ms = 0.0
    with open(path) as fh:
        next(fh)
        for line in fh:
            tag, f, t = line.rsplit(",", 2)
            f, t = float(f), float(t)
            all_ms += t
            if tag.startswith(ALGORITHMIC_TAGS):
                alg_ms += t
                alg_fl += f
                e = by_tag.setdefault(tag, [0.0, 0.0, 0])
                e[0] += t
                e[1] += f
                e[2] += 1
    os.unlink(path)
    achieved = alg_fl / (alg_ms * 1e-3) / 1e12 if alg_ms > 0 else 0.0
    dom_tag, dom = max(by_tag.items(), key=lambda kv: kv[1][0]) if by_tag else ("", [0.0, 0.0, 0])
    dom_tf = dom[1] / (dom[0] * 1e-3) / 1e12 if dom[0] > 0 else 0.0
    traffic, traffic_src = pmc_traffic(workload, B)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
            "traffic_source": traffic_src,
            "kernel": "conv / transposed-conv implicit-GEMM launches (forward, input gradient, weight gradient) and the "
                      "fused attention: the launches whose FLOPs SURVEY 8(d) counts",
            "launches_per_step": int(sum(e[2] for e in by_tag.values()) // nsteps),
            "gemm_ms_per_step": round(alg_ms / nsteps, 3), "gemm_flops_per_step": alg_fl / nsteps,
            "all_gemm_family_ms_per_step": round(all_ms / nsteps, 3),
            "dominant": {"launch": dom_tag, "launches_per_step": dom[2] // nsteps, "ms_per_step": round(dom[0] / nsteps, 3),
                         "achieved": round(dom_tf, 2), "frac": round(dom_tf / peak, 4)},
            "step_algorithmic_flops": fpi * B,
            "step_frac_of_peak": round(fpi * B / (ms_per_step * 1e-3) / 1e12 / peak, 4)}


def run_workload(name, B, a, world, rank, note, extra=(), steps=None, warmup=None, roofline=True, graph=False):
    """Build the model for one workload, time `steps` iterations, optionally the roofline pass.  Returns a dict."""
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
    peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
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
           "precision": precision,
           "storage_dtype": ("bf16 activations + packed bf16 conv weights; fp32 master weights, optimiser, statistics"
                             if precision == "bf16" else "fp32"),
           "da_policy": kw["da_policy"], "g_regularization": kw["g_regularization"],
           "value": round(value, 2), "ms_per_step": round(ms_per_step, 3), "steps": steps,
           "step_frac_of_peak": round(fpi * B / (ms_per_step * 1e-3) / 1e12 / peak, 4),
           "losses": {k: round(float(v.item()), 5) for k, v in losses.items()}}
    if roofline and not graph:
        out["roofline"] = roofline_pass(gan, real, min(steps, 3), peak, fpi, B, ms_per_step, name)
        note("%s: roofline pass done" % name)
    del gan, real
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", type=str, default="", choices=[""] + sorted(WORKLOADS),
                    help="default: c2 on one GPU (+ the config-3 target object), c3 at fixed global batch 256 on N > 1")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--scaling", type=str, default="", choices=["", "strong", "weak"],
                    help="N > 1: strong (default) = BASELINE's fixed global batch split over the ranks; weak = per-GPU batch fixed")
    ap.add_argument("--da_policy", type=str, default="full")
    ap.add_argument("--g_regularization", type=str, default="ortho_cosine")
    ap.add_argument("--n_labels", type=int, default=0, help="class-conditional variant: synthetic one-hot labels")
    ap.add_argument("--gan_type", type=str, default="hinge",
                    help="BASELINE's configs use hinge; e.g. ra-dragan (the reference's default) adds the gradient penalty")
    ap.add_argument("--precision", type=str, default="", help="override: fp32 | bf16-staged | bf16")
    ap.add_argument("--graph", action="store_true", help="replay the iteration from captured HIP graphs (N=1 only)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_target", action="store_true", help="N = 1 default run: skip the config-3 target object")
    a = ap.parse_args()

    t_start = time.perf_counter()
    import torch
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import parallel

    rank, world, local = parallel.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local % torch.cuda.device_count())   # (ranks share a card only in gloo rehearsals)

    def note(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    default_run = a.workload == ""
    name, B, scaling = resolve_workload(a.workload, world, a.scaling, a.batch)
    img, ch, _, desc = WORKLOADS[name]
    if a.graph and world != 1:
        raise SystemExit("--graph is a single-process option (RCCL collectives cannot be captured on this stack: "
                         "tools/rccl_graph_probe.py aborts)")
    res = run_workload(name, B, a, world, rank, note, roofline=not a.no_roofline, graph=a.graph)

    target = None
    if default_run and world == 1 and not a.no_target:
        # BASELINE config 3 (the north-star target) at its global batch on this ONE GPU = the 1-GPU point of the
        # strong-scaling curve, plus the two companions SURVEY 8(d) asks for
        note("target: BASELINE config 3 (BigGAN-128 ch=96 bf16) at global batch 256 on one GPU ...")
        t = run_workload("c3", 256, a, world, rank, note, steps=4, warmup=2, roofline=not a.no_roofline)
        comp = {}
        for key, extra in (("g_regularization_none", {"g_regularization": "none"}), ("da_policy_empty", {"da_policy": ""})):
            c = run_workload("c3", 256, a, world, rank, note, extra=extra.items(), steps=3, warmup=1, roofline=False)
            comp[key] = {"value": c["value"], "ms_per_step": c["ms_per_step"]}
        per32 = run_workload("c3", 32, a, world, rank, note, steps=6, warmup=2, roofline=False)
        target = {"metric": "BigGAN-128 ch=96 bf16 train-step images/sec (BASELINE config 3)", "n_gpus": 1,
                  "value": t["value"], "unit": "images/sec", "ms_per_step": t["ms_per_step"], "global_batch": 256,
                  "dtype": "bf16", "storage_dtype": t["storage_dtype"], "steps": t["steps"],
                  "step_frac_of_peak": t["step_frac_of_peak"], "roofline": t.get("roofline"),
                  "scaling_series": "this object is the N = 1 point of the `--gpus N` strong-scaling series (same workload, "
                                    "same global batch 256); the headline value above is config 2 in fp32",
                  "companions": comp, "losses": t["losses"],
                  "per_gpu_share_32": {"value": per32["value"], "ms_per_step": per32["ms_per_step"],
                                       "note": "one rank's work of the 8-GPU run (32 images per step), without collectives"}}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sb = {64: 32, 128: 8, 256: 2}.get(img, 1)      # ~10-20 s of CPU work on 16 cores
        note("cpu baseline (oracle, batch %d) ..." % sb)
        cpu = cpu_baseline(img, ch, sb, 2 if img <= 128 else 1, note)
        note("cpu baseline done")

    if rank == 0:
        bf16 = name in BF16_WORKLOADS
        out = {
            "metric": "BigGAN-128 train-step images/sec" if img == 128 else "BigGAN-%d train-step images/sec" % img,
            "value": res["value"], "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": desc, "img_size": img, "ch": ch, "per_gpu_batch": B, "global_batch": B * world,
                       "da_policy": a.da_policy, "g_regularization": a.g_regularization, "n_labels": a.n_labels,
                       "gan_type": a.gan_type, "hip_graph": bool(a.graph), "precision": res["precision"],
                       "storage_dtype": res["storage_dtype"], "parallelism": "dp%d" % world},
            "losses": res["losses"],
        }
        if world > 1 and scaling == "strong":
            out["config"]["scaling_series"] = ("N = 1 point of this series: the 'target' object of the --gpus 1 line "
                                               "(this workload at global batch %d on one GPU)" % (B * world))
        if "roofline" in res:
            out["roofline"] = res["roofline"]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if target is not None:
            out["target"] = target
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
