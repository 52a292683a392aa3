"""Data-parallel correctness on the GPU: two ranks (both on cuda:0, process group "gloo" - the only
backend two processes can share one card with; the 8-GPU run uses "nccl" = RCCL with the same code)
each take half of the global batch and must reproduce the single-process full-batch losses and
gradients: cross-replica batch-norm statistics, globally averaged hinge loss (flood sign), SUM
all-reduce of the flat gradient arenas, 1/world weighting of the replicated regulariser."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

IMG, CH, ZD, B = 64, 8, 64, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice_draws(d, lo, hi):
    return {k: v[lo:hi] for k, v in d.items()}


def _extra(batch, tr, cu, dev_draws, lo, hi):
    """Keyword arguments of the D op (gradient-penalty draws) and the G op (real batch of the relativistic types)."""
    dkw, gkw = {}, {}
    if "gp" in batch:
        gp = {"alpha": cu(batch["gp"]["alpha"][lo:hi]), "aug": dev_draws(_slice_draws(batch["gp"]["aug"], lo, hi))}
        if "eps" in batch["gp"]:
            gp["eps"] = cu(batch["gp"]["eps"][lo:hi])
        dkw["gp_draws"] = gp
    if tr.cfg.gan_type.startswith("ra-"):
        gkw = dict(real=cu(batch["real"][lo:hi]), draws_real=dev_draws(_slice_draws(batch["aug_real"], lo, hi)))
    return dkw, gkw


def _worker(rank, world, port, q, gan_type="hinge", shard="1"):
    import sys
    os.environ["BG_SHARD_OPT"] = shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from biggan_tensorflow_amd import parallel
    from oracle import ref_model as RM
    from tests.common import oracle_trainer, hip_model_like, dev_draws, t2n
    torch.cuda.set_device(0)
    parallel.init_from_env(backend="gloo")
    tr = oracle_trainer(IMG, CH, ZD, B, gan_type=gan_type)     # same seed on both ranks: identical replicas
    gan = hip_model_like(tr, gan_type=gan_type)
    assert gan.world == world and gan.rank == rank
    batch = RM.synthetic_batch(tr.cfg, 5, B)
    lo, hi = parallel.shard_batch(B, rank, world)

    def cu(a):
        return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    out = {}
    dkw, gkw = _extra(batch, tr, cu, dev_draws, lo, hi)
    d = gan.d_step(cu(batch["real"][lo:hi]), cu(batch["z_d"][lo:hi]), dev_draws(_slice_draws(batch["aug_real"], lo, hi)),
                   dev_draws(_slice_draws(batch["aug_fake_d"], lo, hi)), apply=False, **dkw)
    out["d_loss"] = d["d_loss"].item()
    out["d_grads"] = t2n(gan.d_arena.grads).copy()
    hip0 = {k: v for k, v in tr.vs.export().items()}
    gan.store.load_arrays({k: v.astype(np.float32) for k, v in hip0.items()}, reset_ema=False)
    g = gan.g_step(hi - lo, cu(batch["z_g"][lo:hi]), dev_draws(_slice_draws(batch["aug_fake_g"], lo, hi)), apply=False,
                   **gkw)
    out["g_adv"] = g["g_adv"].item()
    out["g_grads"] = t2n(gan.g_arena.grads).copy()
    # one full iteration the way train_step runs it under data parallelism: the D all-reduce is started
    # asynchronously and finished (wait + Adam) after the G step's generator forward
    gan.store.load_arrays({k: v.astype(np.float32) for k, v in hip0.items()}, reset_ema=True)
    gan.d_step(cu(batch["real"][lo:hi]), cu(batch["z_d"][lo:hi]), dev_draws(_slice_draws(batch["aug_real"], lo, hi)),
               dev_draws(_slice_draws(batch["aug_fake_d"], lo, hi)), defer=True, **dkw)
    assert gan._pending_d is not None
    gan.g_step(hi - lo, cu(batch["z_g"][lo:hi]), dev_draws(_slice_draws(batch["aug_fake_g"], lo, hi)),
               after_generator=gan._finish_d, **gkw)
    assert gan._pending_d is None
    assert gan.shard_opt == (shard != "0") and gan.pg_grad is not None
    gan.sync_sharded_state()            # collective: parameters whole again, Adam moments / EMA gathered from their owners
    out["d_params"] = t2n(gan.d_arena.params).copy()
    out["g_params"] = t2n(gan.g_arena.params).copy()
    out["g_ema"] = t2n(gan.g_arena.ema).copy()
    out["g_v"] = t2n(gan.g_arena.v).copy()
    out["d_v"] = t2n(gan.d_arena.v).copy()
    out["n_ranges"] = (len(gan.shards["generator"].ranges), len(gan.shards["discriminator"].ranges))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _single(gan_type="hinge"):
    from oracle import ref_model as RM
    from tests.common import oracle_trainer, hip_model_like, dev_draws, t2n
    tr = oracle_trainer(IMG, CH, ZD, B, gan_type=gan_type)
    gan = hip_model_like(tr, gan_type=gan_type)
    batch = RM.synthetic_batch(tr.cfg, 5, B)

    def cu(a):
        return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    out = {}
    dkw, gkw = _extra(batch, tr, cu, dev_draws, 0, B)
    d = gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]),
                   apply=False, **dkw)
    out["d_loss"] = d["d_loss"].item()
    out["d_grads"] = t2n(gan.d_arena.grads).copy()
    gan.store.load_arrays({k: v.astype(np.float32) for k, v in tr.vs.export().items()}, reset_ema=False)
    g = gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), apply=False, **gkw)
    out["g_adv"] = g["g_adv"].item()
    out["g_grads"] = t2n(gan.g_arena.grads).copy()
    gan.store.load_arrays({k: v.astype(np.float32) for k, v in tr.vs.export().items()}, reset_ema=True)
    gan.d_step(cu(batch["real"]), cu(batch["z_d"]), dev_draws(batch["aug_real"]), dev_draws(batch["aug_fake_d"]), **dkw)
    gan.g_step(B, cu(batch["z_g"]), dev_draws(batch["aug_fake_g"]), **gkw)
    out["d_params"] = t2n(gan.d_arena.params).copy()
    out["g_params"] = t2n(gan.g_arena.params).copy()
    out["g_ema"] = t2n(gan.g_arena.ema).copy()
    out["g_v"] = t2n(gan.g_arena.v).copy()
    out["d_v"] = t2n(gan.d_arena.v).copy()
    return out


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


@pytest.mark.parametrize("gan_type,shard", [("hinge", "1"), ("hinge", "0"), ("ra-dragan", "1")])
def test_two_rank_data_parallel_matches_single_process(gan_type, shard):
    """hinge: the BASELINE path.  ra-dragan (the reference's default --gan_type): relativistic batch means, the
    global moments of the real batch behind the DRAGAN perturbation and the gradient penalty's global mean all cross
    the rank boundary.  shard "1" (default): reduce-scatter -> TF-Adam + EMA on the owned half of every exchange range
    -> all-gather of the parameters (last row); "0": all-reduce and a replicated update.  Both must give the
    single-process parameters, Adam second moments and EMA shadows."""
    ref = _single(gan_type)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, gan_type, shard)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert abs(o["d_loss"] - ref["d_loss"]) <= 1e-5 * abs(ref["d_loss"]), (r, o["d_loss"], ref["d_loss"])
        assert abs(o["g_adv"] - ref["g_adv"]) <= 1e-5 * abs(ref["g_adv"]), (r, o["g_adv"], ref["g_adv"])
        # flat arenas after the SUM all-reduce: equal on both ranks and equal to the single-process gradient
        assert _rel(o["d_grads"], ref["d_grads"]) < 1e-4, (r, _rel(o["d_grads"], ref["d_grads"]))
        assert _rel(o["g_grads"], ref["g_grads"]) < 1e-4, (r, _rel(o["g_grads"], ref["g_grads"]))
    assert np.array_equal(res[0]["d_grads"], res[1]["d_grads"])
    assert np.array_equal(res[0]["g_grads"], res[1]["g_grads"])
    # full iteration with the deferred D update: replicas stay identical and follow the single-process weights.
    # (TF-Adam with beta1 = 0 moves every element by ~lr * sign(g): compare the UPDATE, not the weights, and
    # allow the sign noise of near-zero gradient elements)
    for name in ("d_params", "g_params", "g_ema"):
        assert np.array_equal(res[0][name], res[1][name]), name
        same = np.mean(np.abs(res[0][name].astype(np.float64) - ref[name]) < 1e-7)
        assert same > 0.995, (name, same)
    for name in ("g_v", "d_v"):          # Adam's second moment = 0.1 g^2 after one step: smooth in the gradient
        assert np.array_equal(res[0][name], res[1][name]), name
        assert _rel(res[0][name], ref[name]) < 1e-3, (name, _rel(res[0][name], ref[name]))
    assert res[0]["n_ranges"][0] >= 5 and res[0]["n_ranges"] == res[1]["n_ranges"]     # generator: one range per stage


def _rccl_worker(port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import biggan_tensorflow_amd  # noqa: F401
    from biggan_tensorflow_amd import model, scope as S, main as M
    out = {}
    for gt in ("hinge", "ra-dragan"):
        args = M.parse_args(["--gan_type", gt, "--img_size", "64", "--ch", "8", "--z_dim", "64", "--batch_size", "4"],
                            make_dirs=False)
        os.environ["BG_SHARD_OPT"] = "0"
        gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=1), process_group=dist.group.WORLD)
        gan.build_model()
        gan.world = 2                           # every `world > 1` branch runs; the collectives go through RCCL
        gan.reg_owner = gan._shard_regularisers()
        gan._setup_exchange()                   # all-reduce form of the gradient exchange (the group has one rank)
        assert gan.shards and not gan.shard_opt
        real = gan.synthetic_batch(4)
        for _ in range(3):
            losses = gan.train_step(real)
        torch.cuda.synchronize()
        out[gt] = {k: float(v.item()) for k, v in losses.items()}
    # the sharded update's call sites (reduce_scatter_tensor / all_gather_into_tensor in place on arena ranges, the
    # deferred waits, sync_sharded_state) on the 1-rank RCCL group: must reproduce the plain single-process iterations
    # (up to the run-to-run noise of atomically accumulated reductions, which TF-Adam with beta1 = 0 turns into +-lr on
    # elements whose gradient is ~0: compare like the 2-rank test does)
    args = M.parse_args(["--gan_type", "hinge", "--img_size", "64", "--ch", "8", "--z_dim", "64", "--batch_size", "4"],
                        make_dirs=False)
    runs = []
    for mode in ("force", "off"):
        os.environ["BG_SHARD_OPT"] = "force" if mode == "force" else "0"
        gan = model.BigGAN(args, device="cuda", store=S.VariableStore("cuda", seed=1),
                           process_group=dist.group.WORLD if mode == "force" else None, seed=7)
        gan.build_model()
        assert gan.shard_opt == (mode == "force")
        real = gan.synthetic_batch(4)
        for _ in range(2):
            gan.train_step(real)
        gan.sync_sharded_state()
        torch.cuda.synchronize()
        runs.append([t.detach().cpu().clone() for t in (gan.g_arena.params, gan.d_arena.params, gan.g_arena.ema,
                                                        gan.g_arena.v)])
    same = [float(((a - b).abs() < 1e-7).double().mean()) for a, b in zip(runs[0][:3], runs[1][:3])]
    relv = float((runs[0][3] - runs[1][3]).norm() / runs[1][3].norm())
    out["sharded_equals_plain"] = (min(same) > 0.995 and relv < 1e-3, same, relv)
    dist.barrier()
    dist.destroy_process_group()
    q.put(out)


def test_collective_call_sites_under_rccl():
    """Two processes cannot share one card under RCCL, so the numerical 2-rank tests above use gloo.  This one runs
    every collective call site of the data-parallel path (fp64 batch-norm sums, loss sums, the regulariser share,
    chunked and deferred-asynchronous gradient all-reduces, DRAGAN moments, penalty mean, barrier) through the real
    "nccl" (= RCCL) backend on a world-size-1 group with the model told it has two ranks: dtypes, async handles and
    stream ordering are exercised; the values only have to stay finite."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    eq = out.pop("sharded_equals_plain")
    assert eq[0], eq
    for gt, losses in out.items():
        assert losses and all(np.isfinite(v) for v in losses.values()), (gt, losses)
    assert "gp" in out["ra-dragan"]
