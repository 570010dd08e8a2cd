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
``ShardedLoss.value``) has run; ``bucket=M`` lets M consecutive steps share one
all-reduce of an M-vector (one RCCL call carrying M step losses).
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
    def forward(ctx, x, targets, in_len, tgt_len, variant, batch_total, blank, local_fn):
        if local_fn is not None:                      # test hook (gloo/CPU): injected local compute
            loss, grad = local_fn(x, targets, in_len, tgt_len, batch_total)
            ctx.grad, ctx.injected = grad, True
            return loss
        want = ctx.needs_input_grad[0]
        loss, _nll, grad = F._launch(variant, x, targets, in_len, tgt_len, want, batch_total, blank)
        ctx.grad, ctx.injected = grad, False
        ctx.meta = (variant, batch_total, blank)
        if want:
            ctx.save_for_backward(x, targets)
            ctx.lens = (in_len, tgt_len)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        if ctx.injected:
            return ctx.grad * gout, None, None, None, None, None, None, None
        return F._scaled_grad(ctx, gout), None, None, None, None, None, None, None


class ShardedLoss:
    """Result of one sharded step: ``local`` carries the autograd graph of this rank's shard,
    ``value`` is the all-reduced global mean (waits for the collective, stream-ordered)."""

    def __init__(self, local, reduced, work):
        self.local, self._reduced, self._work = local, reduced, work

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self

    @property
    def value(self):
        self.wait()
        return self._reduced

    def backward(self, *a, **k):
        # d(global mean)/d(x_shard) == d(local contribution)/d(x_shard)
        return self.local.backward(*a, **k)


class ShardedCTCLoss:
    """``loss = ShardedCTCLoss(global_batch)(x_shard, targets_shard, in_len_shard, tgt_len_shard)``.

    variant: "noblank" | "binary" | "blank" | None (from the targets, as CTCLoss.apply).
    """

    _VARIANTS = {"noblank": 0, "binary": 1, "blank": 2}

    def __init__(self, global_batch, variant=None, group=None, blank=0, async_op=True, local_fn=None):
        self.global_batch = int(global_batch)
        self.variant = self._VARIANTS[variant] if isinstance(variant, str) else variant
        self.group, self.blank, self.async_op, self.local_fn = group, int(blank), async_op, local_fn

    def __call__(self, x, targets, in_len, tgt_len):
        variant = F._variant_of(targets) if self.variant is None else self.variant
        local = _ShardFn.apply(x, targets, in_len, tgt_len, variant, self.global_batch, self.blank,
                               self.local_fn)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return ShardedLoss(local, local.detach(), None)
        reduced = local.detach().clone()
        work = dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group, async_op=self.async_op)
        return ShardedLoss(local, reduced, work if self.async_op else None)


def all_reduce_losses(loss_vec, group=None, async_op=True):
    """Bucketed form: one all-reduce(SUM) of an M-vector of per-step local contributions."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(loss_vec, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
