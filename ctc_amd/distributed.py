"""Batch-sharded CTC loss: one process per GPU, one all-reduce of the scalar loss.

Every sample's lattice is independent (computes_transition has no cross-b term,
NoBlankCTC.py:71-87); the only cross-sample operation of the reference is the final
``torch.mean(loss)`` (NoBlankCTC.py:140).  Rank r of N therefore owns a slice of the
batch, computes ``sum_{b in shard} nll_b / B_global`` and the input gradient of its own
slice scaled by ``1/B_global`` (no communication: d loss / d x_shard depends on the shard
only), and the global mean is ONE all-reduce(SUM) of a 4-byte value per step -- RCCL
over xGMI with backend "nccl", gloo on CPU for tests.

The 4-byte message is pure latency, so the collective must stay off the compute
stream's critical path: ``ShardedCTCLoss`` launches it asynchronously and hands back
a tensor whose value is complete once ``.wait()`` (or any stream-ordered use through
``ShardedLoss.value``) has run; ``ShardedCTCLoss(..., bucket=M)`` lets M consecutive steps
share one all-reduce of an M-vector (one RCCL call carrying M step losses).
"""
import torch
import torch.distributed as dist

from . import functional as F


def shard_bounds(global_batch, rank, world):
    """[lo, hi) of rank's contiguous slice; the first (global_batch % world) ranks get one more."""
    base, rem = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _ShardFn(torch.autograd.Function):
    """local contribution sum_shard(nll)/B_global, differentiable w.r.t. the shard's logits."""

    @staticmethod
    def forward(ctx, x, targets, in_len, tgt_len, variant, batch_total, blank):
        want = ctx.needs_input_grad[0]
        loss, _nll, grad = F._launch(variant, x, targets, in_len, tgt_len, want, batch_total, blank)
        ctx.grad = grad
        ctx.meta = (variant, batch_total, blank)
        if want:
            ctx.save_for_backward(x, targets)
            ctx.lens = (in_len, tgt_len)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        return F._scaled_grad(ctx, gout), None, None, None, None, None, None


class _Bucket:
    """M consecutive steps' local contributions and the ONE all-reduce that sums them over the ranks"""

    def __init__(self, m, like, group, async_op):
        self.vec = torch.zeros(m, dtype=torch.float32, device=like.device)
        self.n, self.work, self.reduced = 0, None, False
        self.group, self.async_op = group, async_op

    def add(self, local):
        i = self.n
        self.vec[i] = local.detach()
        self.n += 1
        if self.n == self.vec.numel():
            self.launch()
        return i

    def launch(self):
        if self.reduced:
            return
        self.reduced = True
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            w = dist.all_reduce(self.vec[:self.n], op=dist.ReduceOp.SUM, group=self.group, async_op=self.async_op)
            self.work = w if self.async_op else None

    def wait(self):
        self.launch()                                  # (a bucket read before it is full goes out as it is)
        if self.work is not None:
            self.work.wait()
            self.work = None


class ShardedLoss:
    """Result of one sharded step: ``local`` carries the autograd graph of this rank's shard,
    ``value`` is the all-reduced global mean (waits for the collective, stream-ordered)."""

    def __init__(self, local, bucket, index):
        self.local, self._bucket, self._index = local, bucket, index

    def wait(self):
        self._bucket.wait()
        return self

    @property
    def value(self):
        self._bucket.wait()
        return self._bucket.vec[self._index]

    def backward(self, *a, **k):
        # d(global mean)/d(x_shard) == d(local contribution)/d(x_shard)
        return self.local.backward(*a, **k)


class ShardedCTCLoss:
    """``loss = ShardedCTCLoss(global_batch)(x_shard, targets_shard, in_len_shard, tgt_len_shard)``.

    variant: "noblank" | "binary" | "blank" | None (from the targets, as CTCLoss.apply).
    bucket:  steps per all-reduce.  1 (default): one all-reduce of the 4-byte scalar per step (BASELINE north_star).
             M > 1: the local contributions of M consecutive calls travel as ONE all-reduce of an M-vector, launched by the
             M-th call (``flush()`` sends a partial bucket; reading ``.value`` of a step in an open bucket does so too).
             The gradient never waits for a collective: it depends on the shard alone.
    """

    _VARIANTS = {"noblank": 0, "binary": 1, "blank": 2}
    _fn = _ShardFn                                    # the local launch (tests substitute a CPU stand-in by subclassing)

    def __init__(self, global_batch, variant=None, group=None, blank=0, async_op=True, bucket=1):
        if int(bucket) < 1:
            raise ValueError("bucket must be >= 1")
        self.global_batch = int(global_batch)
        self.variant = self._VARIANTS[variant] if isinstance(variant, str) else variant
        self.group, self.blank, self.async_op, self.bucket = group, int(blank), async_op, int(bucket)
        self._open = None

    def __call__(self, x, targets, in_len, tgt_len):
        variant = F._variant_of(targets) if self.variant is None else self.variant
        local = self._fn.apply(x, targets, in_len, tgt_len, variant, self.global_batch, self.blank)
        if self._open is None:
            self._open = _Bucket(self.bucket, local, self.group, self.async_op)
        b = self._open
        i = b.add(local)
        if b.n == self.bucket:
            self._open = None
        return ShardedLoss(local, b, i)

    def flush(self):
        """send the open (partial) bucket now"""
        if self._open is not None:
            self._open.launch()
            self._open = None


def all_reduce_losses(loss_vec, group=None, async_op=True):
    """Bucketed form for loops that keep their own loss vector (a captured graph's ring): one all-reduce(SUM) of an
    M-vector of per-step local contributions."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(loss_vec, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
