"""Data-parallel plumbing on CPU: world_size-2 gloo process group (the GPU path uses the same code
with backend "nccl" = RCCL).  Checks the flat-arena all-reduce, shard arithmetic and the small
statistics all-reduce hook; no HIP arithmetic is involved."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import biggan_tensorflow_amd  # noqa: F401
from biggan_tensorflow_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # flat gradient arena: SUM all-reduce in chunks
    n = 1000 + 7
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_flat(flat, chunk_elems=256)
    exp = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok1 = torch.equal(flat, exp)
    # cross-replica BN statistics / loss sums: tiny in-place SUM
    sums = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
    dist.all_reduce(sums)
    ok2 = sums.tolist() == [sum(1.0 + i for i in range(world)), sum(10.0 * (i + 1) for i in range(world))]
    # replicas start identical
    p = torch.full((5,), float(rank))
    parallel.broadcast_flat(p, src=0)
    ok3 = p.tolist() == [0.0] * 5
    lo, hi = parallel.shard_batch(64, rank, world)
    # sharded update on a flat arena (SURVEY 8e last row): reduce-scatter per exchange range -> update of the owned
    # half -> all-gather, on a second process group; must equal all-reduce + replicated update
    pg2 = parallel.new_gradient_group()
    size = 64 * 37
    sh = parallel.ShardedRanges(size, [64 * 5, 64 * 20], world, rank, max_elems=64 * 8)
    gen = torch.Generator().manual_seed(3)
    p0 = torch.randn(size, generator=gen)
    g_all = [torch.randn(size, generator=gen) for _ in range(world)]
    p, g = p0.clone(), g_all[rank].clone()
    works = [(a, b, parallel.reduce_scatter_range(g, a, b, sh, pg2, async_op=True)) for a, b in sh.ranges]
    gathers = []
    for a, b, wk in works:
        wk.wait()
        oa, ob = sh.owned(a, b)
        p[oa:ob] -= 0.5 * g[oa:ob]                                   # (stands for TF-Adam on the owned part)
        gathers.append(parallel.all_gather_range(p, a, b, sh, pg2, async_op=True))
    for wk in gathers:
        wk.wait()
    ok5 = (pg2 is not None and sh.sharded and sh.ranges[0] == (0, 320) and all(b - a <= 512 for a, b in sh.ranges)
           and sum(b - a for a, b in sh.ranges) == size and sh.containing(320, 1280) == [r for r in sh.ranges if 320 <= r[0] < 1280]
           and torch.allclose(p, p0 - 0.5 * sum(g_all), atol=1e-6)
           and not parallel.ShardedRanges(64 * 4, [], 3, 0).sharded)
    # regulariser sharding: identical owner maps on all ranks, every kernel owned once, loads balanced
    from biggan_tensorflow_amd import main as M, model, scope as S
    g = model.BigGAN(M.parse_args(["--gan_type", "hinge", "--img_size", "128", "--ch", "16"], make_dirs=False),
                     device="cpu", store=S.VariableStore("cpu")).build_model()
    owner = g.reg_owner
    ok4 = (g.world == world and g.rank == rank and owner is not None and
           set(owner) == set(g.store.reg_shapes) and len(owner) == 42 and set(owner.values()) == set(range(world)))
    # bench.py under this very process group: N > 1 runs BASELINE config 3 at its FIXED global batch
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    name, per_rank, mode = bench.resolve_workload("", dist.get_world_size())
    total = torch.tensor([float(per_rank)])
    dist.all_reduce(total)
    ok4 = ok4 and name == "c3" and mode == "strong" and int(total.item()) == bench.GLOBAL_BATCH["c3"] == 256
    q.put((rank, ok1, ok2, ok3 and ok5, lo, hi, ok4, sorted(owner.items())))
    dist.destroy_process_group()


def _worker_exchange(rank, world, port, q):
    """The sharded update's data movement alone, for a larger world: in-place reduce-scatter of every exchange range, update
    of the owned 1 / world, in-place all-gather - on the dedicated process group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_from_env(backend="gloo")
    pg2 = parallel.new_gradient_group()
    size = 64 * 203
    sh = parallel.ShardedRanges(size, [64 * 40, 64 * 41, 64 * 150], world, rank, max_elems=64 * 32)
    gen = torch.Generator().manual_seed(11)
    p0 = torch.randn(size, generator=gen)
    g_all = [torch.randn(size, generator=gen) for _ in range(world)]
    p, g = p0.clone(), g_all[rank].clone()
    works = [(a, b, parallel.reduce_scatter_range(g, a, b, sh, pg2, async_op=True)) for a, b in sh.ranges]
    gathers = []
    for a, b, wk in works:
        wk.wait()
        oa, ob = sh.owned(a, b)
        p[oa:ob] -= 0.5 * g[oa:ob]
        gathers.append(parallel.all_gather_range(p, a, b, sh, pg2, async_op=True))
    for wk in gathers:
        wk.wait()
    ok = sh.sharded and torch.allclose(p, p0 - 0.5 * sum(g_all), atol=1e-5)
    # every rank ends with the SAME parameters, bit for bit (they all received the owners' bytes)
    digest = float(p.double().sum())
    q.put((rank, bool(ok), digest))
    dist.destroy_process_group()


def test_gloo_world8_sharded_exchange():
    """Eight ranks (the driver's scaling run) over gloo on CPU: the in-place reduce-scatter / all-gather on arena ranges that
    divide by 8, on the second process group."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world)) and all(r[1] for r in res)
    assert len({r[2] for r in res}) == 1


def test_gloo_world2_allreduce_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[:7] for r in res] == [(0, True, True, True, 0, 32, True), (1, True, True, True, 32, 64, True)]
    assert res[0][7] == res[1][7]                      # the same owner map on every rank


def test_bench_gpus_n_starts_the_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT torchrun's environment (how the driver calls it) must run 2 ranks, not one:
    the launcher spawns them before touching a GPU and relays rank 0's line.  --dry_run = rank plumbing only (gloo, no
    model, value null) so this runs on the CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry_run"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 256
    assert d["config"]["per_gpu_batch"] == 128 and d["config"]["parallelism"] == "dp2" and d["dry_run"] is True
    # a failing rank makes the launcher exit non-zero
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry_run"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0                      # 256 is not divisible by 3 ranks


def test_shard_batch_rejects_ragged():
    with pytest.raises(ValueError):
        parallel.shard_batch(65, 0, 2)
    assert parallel.shard_batch(256, 7, 8) == (224, 256)


def test_dp_gradient_identity_numpy():
    """Why no 1/world factor is applied: with per-sample gradients already scaled by 1/global_batch,
    the SUM over ranks of shard gradients equals the single-process gradient."""
    rng = np.random.default_rng(0)
    per_sample = rng.standard_normal((8, 5))
    full = (per_sample / 8).sum(0)
    shards = [(per_sample[lo:hi] / 8).sum(0) for lo, hi in ((0, 4), (4, 8))]
    np.testing.assert_allclose(sum(shards), full, rtol=1e-13)


def test_eight_rank_partitions_are_complete_and_disjoint():
    """The 8-GPU run cannot be rehearsed here; what it partitions can: (a) ShardedRanges for every rank of world 8 on an
    arena cut at stage boundaries - the owned sub-ranges tile every exchange range exactly once, in rank order (what the
    in-place reduce-scatter / all-gather require: rank r's piece starts at lo + r * len / world); (b) the spectral-norm
    ownership - every weight has exactly one owner, the segments of a rank's sigma | u | v_hat do not overlap, loads are
    balanced to within the largest weight."""
    import random
    from biggan_tensorflow_amd.parallel import ShardedRanges
    from biggan_tensorflow_amd.functional import sn_shard_layout
    world = 8
    size = 64 * 1237
    bounds = [64 * 100, 64 * 400, 64 * 401, 64 * 900]
    per_rank = [ShardedRanges(size, bounds, world, r, max_elems=64 * 256) for r in range(world)]
    assert all(s.sharded and s.ranges == per_rank[0].ranges for s in per_rank)
    covered = 0
    for lo, hi in per_rank[0].ranges:
        assert (hi - lo) % world == 0 and hi - lo <= 64 * 256
        pieces = [s.owned(lo, hi) for s in per_rank]
        step = (hi - lo) // world
        assert pieces == [(lo + r * step, lo + (r + 1) * step) for r in range(world)]
        covered += hi - lo
    assert covered == size and per_rank[0].ranges[0][0] == 0 and per_rank[0].ranges[-1][1] == size
    # spectral norm: config-3-like weight list (k*k*Cin x Cout kernels, dense layers, 1 x 1 attention projections)
    rnd = random.Random(3)
    shapes = [(9 * cin, cout) for cin, cout in ((1536, 1536), (1536, 768), (768, 768), (768, 384), (384, 384), (384, 192),
                                                (192, 192), (192, 96), (96, 96))] * 3
    shapes += [(16 * 1536, 768), (16 * 768, 384), (20, 24576), (148, 1536), (96, 12), (96, 48), (1536, 1)] * 2
    rnd.shuffle(shapes)
    rows, cols = [s[0] for s in shapes], [s[1] for s in shapes]
    numels = [a * b for a, b in shapes]
    owner, place, fill = sn_shard_layout(numels, rows, cols, world)
    assert len(owner) == len(shapes) and set(owner) == set(range(world))
    load = [sum(n for n, o in zip(numels, owner) if o == r) for r in range(world)]
    assert max(load) - min(load) <= max(numels)
    for r in range(world):
        spans = sorted((o_s, o_v + (rows[i] + 3) // 4 * 4) for i, (rr, o_s, o_u, o_v) in enumerate(place) if rr == r)
        assert spans[0][0] == 0 and spans[-1][1] == fill[r]
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    for i, (rr, o_s, o_u, o_v) in enumerate(place):
        assert rr == owner[i] and o_u == o_s + 4 and o_v == o_u + (cols[i] + 3) // 4 * 4
