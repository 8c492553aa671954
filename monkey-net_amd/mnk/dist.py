"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in the
CPU tests).  Replaces the reference's single-process DataParallel + SyncMaster threads
(sync_batchnorm/replicate.py:50-67, comm.py, batchnorm.py:90-111):

* BatchNorm sufficient statistics  [sum x, sum x^2]  (forward) and  [sum g, sum g*xhat]  (backward) are summed over
  ranks with one small all-reduce per layer (`all_reduce_sum_`); every rank then finalises mean / inv-std itself,
  so the reference's reduce-to-master + broadcast pair (batchnorm.py:102,105) collapses into one collective.
  Ranks are assumed to hold equally sized shards (count = local count * world size).
* Gradients are averaged with a bucketed flat all-reduce after backward (`GradAverager`), replacing the implicit
  reduce-add of DataParallel's backward and its per-forward parameter broadcast.
"""
import os

import torch
import torch.distributed as tdist

_SYNC_BN = True


def set_sync_bn(flag):
    """Disable to get per-rank BatchNorm statistics (plain DDP semantics)."""
    global _SYNC_BN
    _SYNC_BN = bool(flag)


def initialized():
    return tdist.is_available() and tdist.is_initialized()


def world_size():
    return tdist.get_world_size() if initialized() else 1


def rank():
    return tdist.get_rank() if initialized() else 0


_FORCE = os.environ.get("MNK_DIST_FORCE", "") == "1"   # exercise the collectives even with a single rank (tests)


def active():
    """True when BatchNorm statistics must be exchanged."""
    return _SYNC_BN and initialized() and (tdist.get_world_size() > 1 or _FORCE)


def all_reduce_sum_(t):
    tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
    return t


def combine_bn_stats(sums, count):
    """Host-side statement of the SyncBN exchange for tests: returns (global sums, global count)."""
    if active():
        sums = sums.clone()
        all_reduce_sum_(sums)
        count = count * world_size()
    return sums, count


class GradAverager:
    """Bucketed gradient averaging over ranks.

    Gradients are copied into flat fp32 buckets of ~`bucket_mb` MB (fewer, larger collectives: xGMI rings are
    per-link bound), each bucket is all-reduced asynchronously as soon as it is filled, and the averaged values are
    copied back.  Parameters without a gradient are skipped (e.g. the discriminator in an inference-only step)."""

    def __init__(self, params, bucket_mb=64.0):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)

    def average(self):
        if not initialized() or (world_size() == 1 and not _FORCE):
            return 0
        ws = float(world_size())
        pending = []
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            work = tdist.all_reduce(flat, op=tdist.ReduceOp.SUM, async_op=True)
            pending.append((work, flat, bucket))
            bucket, size = [], 0

        n = 0
        for p in reversed(self.params):          # roughly the order in which backward produced them
            if p.grad is None:
                continue
            bucket.append(p)
            size += p.grad.numel()
            n += 1
            if size >= self.bucket_elems:
                flush()
        flush()
        for work, flat, ps in pending:
            work.wait()
            flat.div_(ws)
            off = 0
            for p in ps:
                k = p.grad.numel()
                p.grad.copy_(flat[off:off + k].view_as(p.grad))
                off += k
        return n


def shard_batch(x, dim=0):
    """This rank's equal slice of a global batch tensor (the DataParallel scatter, replicate.py:64-67)."""
    ws, r = world_size(), rank()
    if ws == 1:
        return x
    n = x.shape[dim]
    assert n % ws == 0, "global batch %d is not divisible by world size %d" % (n, ws)
    per = n // ws
    return x.narrow(dim, r * per, per)
