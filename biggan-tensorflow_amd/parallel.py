"""Data parallelism over the 8 MI355X of one node: one process per GPU, ``torch.distributed``
backend "nccl" (= RCCL over xGMI).  The reference is single-device (BigGAN.py:776); sharding the
minibatch adds exactly these exchanges:

* G / D weight gradients: SUM all-reduce of the network's FLAT gradient arena, issued as a few
  large chunks (xGMI is a 7-link point-to-point mesh: few big messages beat many small ones);
* batch-norm statistics and the hinge-loss sums: tiny SUM all-reduces inside forward/backward
  (``ops._run.reduce_fn``), so the maths equals the single-process run on the global batch;
* the batch-independent work of a step is SHARDED instead of replicated (last row): the
  gradient exchange of an arena range is a reduce-scatter, each rank runs TF-Adam (+ EMA) on the 1/N of the range it
  owns, and an all-gather returns the updated parameters (``ShardedRanges``) - the same wire bytes as the all-reduce.
  Gradient traffic runs on its OWN process group (``new_gradient_group``): under RCCL a group is a communicator with
  its own stream, so the latency-bound batch-norm reductions issued from inside backward do not queue behind a 150 MB
  bucket that was started a moment earlier.

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


def new_gradient_group(group=None):
    """A second process group over the same ranks for the large asynchronous exchanges (every rank must call this at
    the same point: model.BigGAN.build_model does).  None when there is nothing to communicate."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
    return dist.new_group(ranks=ranks, backend=dist.get_backend(group))


class ShardedRanges:
    """Static partition of a flat arena into exchange ranges, each split evenly over the ranks.

    ``ranges`` = [(lo, hi), ...] cover [0, size) without overlap; rank r OWNS [lo + r * s, lo + (r + 1) * s) of every
    range, s = (hi - lo) / world.  The partition is fixed for the life of the model: the Adam moments and the EMA shadow
    of an element live only on its owner, so every step must exchange exactly these ranges.  Arena slots are aligned to
    64 floats, so range lengths divide by any world size in {1, 2, 4, 8, 16, 32, 64}; otherwise ``sharded`` is False
    and callers fall back to the all-reduce."""

    def __init__(self, size, boundaries, world, rank, max_elems=CHUNK_ELEMS):
        cuts = sorted(set(int(b) for b in boundaries if 0 < b < size) | {0, int(size)})
        ranges = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            while hi - lo > max_elems:                      # cap the size of one collective (256 MiB)
                ranges.append((lo, lo + max_elems))
                lo += max_elems
            ranges.append((lo, hi))
        self.ranges = ranges
        self.world, self.rank, self.size = world, rank, int(size)
        self.sharded = world > 1 and all((hi - lo) % world == 0 for lo, hi in ranges)

    def owned(self, lo, hi):
        s = (hi - lo) // self.world
        return lo + self.rank * s, lo + (self.rank + 1) * s

    def containing(self, lo, hi):
        """The ranges inside [lo, hi) (which must be a union of whole ranges)."""
        out = [r for r in self.ranges if r[0] >= lo and r[1] <= hi]
        if sum(b - a for a, b in out) != hi - lo:
            raise ValueError("[%d, %d) is not a union of exchange ranges" % (lo, hi))
        return out


def reduce_scatter_range(flat, lo, hi, shards, group=None, async_op=False):
    """SUM reduce-scatter of flat[lo:hi] IN PLACE: afterwards this rank's owned sub-range holds the sum over ranks
    (the rest of the range is stale).  Returns the work handle (async) or None."""
    a, b = shards.owned(lo, hi)
    return dist.reduce_scatter_tensor(flat.narrow(0, a, b - a), flat.narrow(0, lo, hi - lo), op=dist.ReduceOp.SUM,
                                      group=group, async_op=async_op)


def all_gather_range(flat, lo, hi, shards, group=None, async_op=False):
    """All-gather of the owned sub-ranges of flat[lo:hi] IN PLACE (the inverse data movement of reduce_scatter_range)."""
    a, b = shards.owned(lo, hi)
    return dist.all_gather_into_tensor(flat.narrow(0, lo, hi - lo), flat.narrow(0, a, b - a), group=group,
                                       async_op=async_op)
