"""Env-sharded data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  Environments are independent units, so each rank owns a contiguous block of envs,
its rollout buffers, its CaT per-env statistics and its GAE; the exchange points are

    * flat gradient           SUM   once per optimiser step (one buffer, one call)
    * CaT column maxima       MAX   K floats per env step      (exact => masks stay bit-exact)
    * normaliser moments      SUM   fp64 [sum x | sum x^2]     per normaliser update
    * advantage mean / std    SUM   2 fp64 per minibatch

The last three make N ranks reproduce ONE process on the union of the shards; the
helpers are device agnostic: on the GPUs they call RCCL directly through libcatppo's C ABI (catppo_comm_init /
catppo_allreduce, see init_native_comm) - torch.distributed is only the rendezvous that ships the 128-byte unique id -
and on CPU tensors (the gloo tests) they fall back to torch.distributed.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

_FORCE = os.environ.get("CATPPO_FORCE_DIST", "0") == "1"
#: libcatppo context whose RCCL communicator carries the device-side exchange points (None: torch.distributed does)
_native = None


_native_error = None


def init_native_comm(nat, group=None) -> bool:
    """Create libcatppo's own RCCL communicator (catppo_comm_init) for the ranks of the initialised
    torch.distributed world: rank 0 makes the unique id, the launcher's process group only ships its 128 bytes.
    Afterwards every device-tensor exchange point below is a catppo_allreduce / catppo_broadcast on the caller's
    HIP stream - capturable in a hipGraph, no torch.distributed on the data path.  CATPPO_NATIVE_COMM=0 keeps
    torch.distributed (RCCL through PyTorch) instead."""
    global _native
    if _native is not None:
        return True
    if not active(group) or os.environ.get("CATPPO_NATIVE_COMM", "1") == "0":
        return False
    # Every rank must end up on the same transport: a rank that could not join (librccl not loadable, communicator
    # set-up refused) votes no, and then ALL ranks keep torch.distributed (RCCL through PyTorch) - said on stderr
    # and in native_comm_error(), never silently.
    global _native_error
    err = None
    try:
        box = [nat.comm_unique_id() if rank(group) == 0 else None]
    except RuntimeError as e:      # rank 0 could not even make an id: the others must not wait inside RCCL
        box, err = [None], str(e)
    dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        err = err or "rank 0 could not create an RCCL unique id"
    else:
        try:
            nat.comm_init(rank(group), world_size(group), box[0])
        except RuntimeError as e:
            err = str(e)
    votes = [None] * world_size(group)
    dist.all_gather_object(votes, err, group=group)
    failed = [(r, v) for r, v in enumerate(votes) if v is not None]
    if failed:
        if err is None:
            nat.comm_destroy()
        _native_error = "; ".join(f"rank {r}: {v}" for r, v in failed)
        if rank(group) == 0:
            import sys
            print(f"[catppo] native RCCL communicator unavailable, collectives stay on torch.distributed: "
                  f"{_native_error}", file=sys.stderr)
        return False
    _native = nat
    return True


def native_comm_error():
    """why init_native_comm() fell back to torch.distributed (None when it did not)"""
    return _native_error


def shutdown_native_comm():
    global _native
    if _native is not None:
        _native.comm_destroy()
        _native = None


def native_comm_active() -> bool:
    return _native is not None


def _use_native(t: torch.Tensor, group) -> bool:
    return _native is not None and t.is_cuda and group in (None, dist.group.WORLD) and t.is_contiguous() and \
        t.dtype in (torch.float32, torch.float64, torch.float16)


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def active(group=None) -> bool:
    """True when the exchange points must run.  CATPPO_FORCE_DIST=1 turns them on for an initialised
    world of size 1 too, so the two-phase kernels + collectives can be exercised on a single GPU."""
    if world_size(group) > 1:
        return True
    return _FORCE and dist.is_available() and dist.is_initialized()


def shard_slice(n_total: int, r: int, w: int) -> slice:
    """contiguous block of rank ``r`` of ``w`` (sizes differ by at most one)"""
    base, rem = divmod(n_total, w)
    start = r * base + min(r, rem)
    return slice(start, start + base + (1 if r < rem else 0))


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    if active(group):
        if _use_native(t, group):
            _native.allreduce(t, 0)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t: torch.Tensor, group=None) -> torch.Tensor:
    if active(group):
        if _use_native(t, group):
            _native.allreduce(t, 1)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if active(group):
        if _use_native(t, group):
            _native.broadcast(t, src)
        else:
            dist.broadcast(t, src=src, group=group)
    return t


def global_moment_sums(sums_and_count: torch.Tensor, group=None) -> torch.Tensor:
    """``[sum x (D) | sum x^2 (D) | n]`` in fp64, summed over ranks in place"""
    assert sums_and_count.dtype == torch.float64
    return allreduce_sum_(sums_and_count, group)


def global_adv_stats(adv_local: torch.Tensor, group=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """mean and (unbiased std + 1e-8) of the minibatch advantages over ALL ranks
    (reference ppo.py:316-318 on the union of the per-rank minibatch shards)"""
    a = adv_local.double()
    s = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64, device=a.device)])
    allreduce_sum_(s, group)
    n = s[2]
    mean = s[0] / n
    var = ((s[1] - n * mean * mean) / (n - 1)).clamp_min(0)
    res = torch.stack([mean, var.sqrt() + 1e-8]).float()
    if out is not None:
        out.copy_(res)
        return out
    return res
