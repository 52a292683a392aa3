"""Data parallelism over the 8 MI355X of one node: one process per GPU, ``torch.distributed``
backend "nccl" (= RCCL over xGMI).  The reference is single-device (BigGAN.py:776); sharding the
minibatch adds exactly these exchanges:

* G / D weight gradients: SUM all-reduce of the network's FLAT gradient arena, issued as a few
  large chunks (xGMI is a 7-link point-to-point mesh: few big messages beat many small ones);
* batch-norm statistics and the hinge-loss sums: tiny SUM all-reduces inside forward/backward
  (``ops._run.reduce_fn``), so the maths equals the single-process run on the global batch.

Per-sample gradients are already scaled by 1/global_batch inside the loss kernels, so no 1/world
factor is applied to the data terms.
"""
import os

import torch
import torch.distributed as dist

# elements per all-reduce call: 64 Mi floats = 256 MiB
CHUNK_ELEMS = 64 * 1024 * 1024


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("BG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_flat(flat, group=None, chunk_elems=CHUNK_ELEMS, async_op=False):
    """SUM all-reduce of a flat fp32 buffer in place, in chunks of ``chunk_elems``."""
    n = flat.numel()
    works = []
    for off in range(0, n, chunk_elems):
        part = flat.narrow(0, off, min(chunk_elems, n - off))
        w = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard_batch(global_batch, rank, world):
    """Per-rank slice [lo, hi) of a global minibatch (equal shards; the remainder is rejected so that
    every rank runs identical shapes)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def broadcast_flat(flat, src=0, group=None):
    """Make replicas identical (used once after initialisation)."""
    dist.broadcast(flat, src=src, group=group)
