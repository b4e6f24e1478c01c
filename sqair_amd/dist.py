"""Data-parallel sharding of the SQAIR hot path over the GPUs of one node (one process per GPU,
``torch.distributed`` over RCCL/xGMI — backend "nccl" on ROCm).

The reference is single-device; every sequence b — and every particle k — is
independent through the whole forward pass (the only cross-row ops are the final means, reference
sqair/model.py:91-93, sqair/targets.py:75), so the path shards by sequences with NO collective on the
data path.  What crosses GPUs:
  * evaluation: one tiny all-reduce of the scalar metrics (``reduce_scalars``);
  * training: one all-reduce(sum) of the flat fp32 gradient buffer per step
    (2 951 522 floats = 11.8 MB at 50x50), divided by the world size so that the result equals the
    reference's ``reduce_mean`` over the global batch (``allreduce_flat_grads``).  11.8 MB over 7 xGMI
    links is latency-bound, one bucket, nothing to overlap.
All K particles of a sequence stay on one rank (IWAE / VIMCO reduce over K locally).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world):
    """Contiguous, equal shards of sequences (the global batch must divide evenly: the mean of shard means
    is then the global mean)."""
    if global_batch % world != 0:
        raise ValueError("global batch {} is not divisible by world size {}".format(global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(x, rank, world, batch_axis=1):
    """Slices [T, B, ...] inputs (obs, nums, coords) to this rank's sequences."""
    lo, hi = shard_bounds(x.shape[batch_axis], rank, world)
    idx = [slice(None)] * x.ndim
    idx[batch_axis] = slice(lo, hi)
    return x[tuple(idx)]


def shard_noise(noise, k_particles, rank, world):
    """noise [T, B*K, 2, N, w] of the GLOBAL batch -> this rank's rows (b' = b*K + k keeps a sequence's particles
    contiguous, reference sqair/index.py:106-129), so 1 and N GPUs consume identical draws."""
    B = noise.shape[1] // k_particles
    lo, hi = shard_bounds(B, rank, world)
    return noise[:, lo * k_particles:hi * k_particles]


def reduce_scalars(values, world=None):
    """Mean over ranks of a small vector of per-shard means (elbo_iwae, elbo_vae, data_ll, ...)."""
    t = torch.as_tensor(values).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t / dist.get_world_size()
    return t


def allreduce_flat_grads(flat_grad, comm=None, stream=None):
    """The single collective of a training step: sum the flat gradient buffer over the ranks, in place.  Returns the
    factor 1 / world that turns the sum into the reference's ``reduce_mean`` over the global batch — the caller folds it
    into the fused optimiser kernel (``sqair_rmsprop_step``'s ``grad_scale``) instead of spending a kernel on a division.

    ``comm`` (a ``sqair_amd.rccl.RcclComm``) enqueues ``ncclAllReduce`` on ``stream`` — the library's own launch stream,
    so the step stays on ONE hardware queue; without it the default ``torch.distributed`` group is used (gloo in the CPU
    tests and on boxes with fewer devices than ranks)."""
    if comm is not None:
        if comm.world > 1:
            comm.all_reduce_sum_(flat_grad, stream)
        return 1.0 / comm.world
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0
