"""Data-parallel plumbing (the reference has none: SURVEY.md 8e).  One process per GPU;
`torch.distributed` backend "nccl" is RCCL over xGMI on the box, "gloo" in the CPU tests.

* training: sentences are independent and every loss term is a batch mean
  (CV/Experiment.py:730-751), so with equal per-rank batches the mean of the per-rank
  gradients IS the global-batch gradient.  One collective per step, on the flat gradient
  arena, completed BEFORE the clip (Experiment.py:515 clips the full-batch gradient).
* IW-NLL / MI: whole batches are sharded over ranks (MI depends on batch composition);
  only five (two) sums are reduced at the end.
* IW-NLL, sample-sharded (`settings.shard_samples`, SURVEY.md 8e "preferred"): every rank sees the
  same batch and the same eps (broadcast from rank 0), decodes K/N of the K samples, and the
  (K/N, B) cross-entropy rows are all-gathered; every rank then reduces the full (K, B) table,
  so test_lm returns identical (xent, nll, kl) everywhere and its caller is unchanged.
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
    if flat.is_cuda and dist.get_backend() == 'gloo':   # CPU-side rehearsal / tests: stage through the host
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h.mul_(1.0 / w))
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


def allreduce_sum_async(flat):
    """Starts the SUM all-reduce of a flat buffer and returns a handle to wait on (None when it already completed:
    gloo on device buffers, staged through the host).  With nccl (RCCL) the collective runs on its own stream behind
    the work queued so far on the current stream, i.e. it overlaps whatever the caller enqueues next."""
    if flat.is_cuda and dist.get_backend() == 'gloo':
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def shard_batches(n_batches, r=None, w=None):
    """Indices of the batches rank r evaluates (round-robin)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n_batches, w))


def reduce_sums_(acc):
    """Sum a small tensor of partial sums over ranks (in place)."""
    if world() > 1:
        if acc.is_cuda and dist.get_backend() == 'gloo':
            h = acc.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            acc.copy_(h)
        else:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


def sample_slice(n_sample, r=None, w=None):
    """[lo, hi) of the K importance samples rank r decodes (equal contiguous slices)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    assert n_sample % w == 0, 'n_sample (%d) must be a multiple of the world size (%d)' % (n_sample, w)
    k = n_sample // w
    return r * k, (r + 1) * k


def broadcast_(t, src=0):
    if world() > 1:
        if t.is_cuda and dist.get_backend() == 'gloo':
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src)
    return t


def allgather_rows(x):
    """(k, ...) per rank -> (world*k, ...) in rank order, identical on every rank."""
    w = world()
    if w == 1:
        return x
    x = x.contiguous()
    if x.is_cuda and dist.get_backend() == 'gloo':      # gloo gathers host tensors only (CPU-side tests)
        return allgather_rows(x.cpu()).to(x.device)
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather(list(out.chunk(w, 0)), x)
    return out
