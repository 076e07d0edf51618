"""Data-parallel plumbing (the reference has none: SURVEY.md 8e).  One process per GPU;
`torch.distributed` backend "nccl" is RCCL over xGMI on the box, "gloo" in the CPU tests.

* training: sentences are independent and every loss term is a batch mean
  (CV/Experiment.py:730-751), so with equal per-rank batches the mean of the per-rank
  gradients IS the global-batch gradient.  One collective per step, on the flat gradient
  arena, completed BEFORE the clip (Experiment.py:515 clips the full-batch gradient).
* IW-NLL / MI: whole batches are sharded over ranks (MI depends on batch composition);
  only five (two) sums are reduced at the end.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat, n_buckets=1):
    """In-place mean over ranks of a flat fp32 buffer.  n_buckets > 1 issues the buckets as
    separate async collectives (RCCL pipelines them over the xGMI links)."""
    w = world()
    if w == 1:
        return flat
    if n_buckets <= 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    else:
        step = (flat.numel() + n_buckets - 1) // n_buckets
        hs = [dist.all_reduce(flat[i:i + step], op=dist.ReduceOp.SUM, async_op=True)
              for i in range(0, flat.numel(), step)]
        for h in hs:
            h.wait()
    flat.mul_(1.0 / w)
    return flat


def shard_batches(n_batches, r=None, w=None):
    """Indices of the batches rank r evaluates (round-robin)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n_batches, w))


def reduce_sums_(acc):
    """Sum a small tensor of partial sums over ranks (in place)."""
    if world() > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc
