"""Batch-sharded loss (config 4 logic) on CPU: world_size 2, gloo.  The local compute is
substituted by subclassing (the oracle stands in for the HIP launch, which needs a GPU); what is under test
is the sharding, the 1/B_global scaling, the single all-reduce and the autograd plumbing
of ctc_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import np_, synth_noblank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_local(x, targets, in_len, tgt_len, batch_total):
    from oracle import ctc_numpy
    r = ctc_numpy.noblank_ctc(np_(x), np_(targets), np_(in_len), np_(tgt_len), np.float64, scale=1.0 / batch_total)
    loss = torch.tensor(r["nll"].sum() / batch_total, dtype=torch.float32)
    return loss, torch.tensor(r["grad"], dtype=torch.float32)


class _OracleShardFn(torch.autograd.Function):
    """the local launch replaced by the oracle (the HIP launch needs a GPU): same signature as ctc_amd.distributed._ShardFn"""

    @staticmethod
    def forward(ctx, x, targets, in_len, tgt_len, variant, batch_total, blank):
        loss, ctx.grad = _oracle_local(x, targets, in_len, tgt_len, batch_total)
        return loss

    @staticmethod
    def backward(ctx, gout):
        return ctx.grad * gout, None, None, None, None, None, None


def _sharded(**kw):
    from ctc_amd.distributed import ShardedCTCLoss

    class _OracleSharded(ShardedCTCLoss):
        _fn = _OracleShardFn
    return _OracleSharded(**kw)


def _worker(rank, world, port, B, out_dir, bucket):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ctc_amd.distributed import all_reduce_losses, shard_bounds
    x, lab, Tb, L = synth_noblank(0, 12, B, 9, 4, var_T=True)          # same global batch on every rank
    lo, hi = shard_bounds(B, rank, world)
    xs = x[:, lo:hi].clone().requires_grad_(True)
    crit = _sharded(global_batch=B, variant="noblank")
    res = crit(xs, lab[lo:hi], Tb[lo:hi], L[lo:hi])
    (2.0 * res.local).backward()
    val = float(res.value)
    # bucketed form: M per-step contributions in one all-reduce
    vec = torch.stack([res.local.detach() * (k + 1) for k in range(bucket)])
    w = all_reduce_losses(vec, async_op=True)
    w.wait()
    # ShardedCTCLoss(bucket=M): M calls, ONE all-reduce launched by the M-th; a partial bucket goes out on flush / read
    critb = _sharded(global_batch=B, variant="noblank", bucket=bucket)
    steps = [critb(xs.detach(), lab[lo:hi], Tb[lo:hi], L[lo:hi]) for _ in range(bucket + 1)]
    assert steps[0]._bucket is steps[bucket - 1]._bucket and steps[0]._bucket.reduced and not steps[bucket]._bucket.reduced
    bvals = [float(s_.value) for s_ in steps]                           # (the last read sends the open bucket of one)
    assert critb._open is not None and critb._open.reduced
    critb.flush()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, grad=xs.grad.numpy(), value=val,
             local=float(res.local), vec=vec.numpy(), bvals=np.array(bvals))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])
def test_sharded_loss_matches_full_batch(tmp_path, B):
    from oracle import ctc_numpy
    world, bucket = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path), bucket), nprocs=world, join=True)
    x, lab, Tb, L = synth_noblank(0, 12, B, 9, 4, var_T=True)
    full = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    parts = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    assert [int(p["lo"]) for p in parts] == [0, (B + 1) // 2] and int(parts[-1]["hi"]) == B
    for p in parts:
        # every rank sees the global mean after the single all-reduce
        assert abs(float(p["value"]) - float(full["loss"])) < 1e-5
        # local gradients are the slices of the full-batch gradient (x2: upstream gradient)
        assert np.abs(p["grad"] - 2.0 * full["grad"][:, int(p["lo"]):int(p["hi"])]).max() < 1e-6
        assert np.allclose(p["vec"], [float(full["loss"]) * (k + 1) for k in range(bucket)], atol=1e-5)
        assert np.allclose(p["bvals"], float(full["loss"]), atol=1e-5)         # every step of every bucket: the global mean
    assert abs(sum(float(p["local"]) for p in parts) - float(full["loss"])) < 1e-5


def test_shard_bounds_cover_batch():
    from ctc_amd.distributed import shard_bounds
    for B in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_degenerate_mode():
    x, lab, Tb, L = synth_noblank(1, 8, 3, 5, 3)
    xs = x.clone().requires_grad_(True)
    res = _sharded(global_batch=3, variant="noblank")(xs, lab, Tb, L)
    res.backward()
    from oracle import ctc_numpy
    full = ctc_numpy.noblank_ctc(np_(x), np_(lab), np_(Tb), np_(L), np.float64)
    assert abs(float(res.value) - float(full["loss"])) < 1e-5
    assert np.abs(np_(xs.grad) - full["grad"]).max() < 1e-6
