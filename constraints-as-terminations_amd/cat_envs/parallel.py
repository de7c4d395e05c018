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

Round 5: the rendezvous is a GLOO (CPU) process group (``init_rendezvous``), so a rank creates exactly ONE RCCL
communicator - libcatppo's - instead of torch's "nccl" group (a whole second RCCL bootstrap with its own channel
buffers, created only to ship 128 bytes and a few votes) plus ours.  If the native communicator cannot be set up the
ranks agree on that (votes over gloo) and only THEN create a torch "nccl" group for the device operands
(``_device_fallback_group``); ranks that share one GPU (launcher proof on a one-GPU box) stage through host memory.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

_FORCE = os.environ.get("CATPPO_FORCE_DIST", "0") == "1"
#: libcatppo context whose RCCL communicator carries the device-side exchange points (None: torch.distributed does)
_native = None


_native_error = None
#: torch "nccl" group created lazily for device operands when the rendezvous group is gloo and libcatppo's own
#: communicator is unavailable (None: not needed / ranks share a device: host staging)
_device_fallback_group = None
#: what the communicator set-up saw of the environment (bench.py prints it as ``config.comm_env``)
_comm_env = {}


def init_rendezvous(local_device: int | None = None, timeout_s: float = 600.0):
    """Process group of the launcher's ranks for everything that is NOT the data path: the unique id of libcatppo's RCCL
    communicator, the set-up votes, barriers, Python-object gathers.  Backend gloo (CPU, TCP on MASTER_ADDR): no RCCL
    communicator is created by torch.  Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like torch.distributed.run
    sets them (reference precedent: scripts/skrl/train.py:116-117 initialises its process group from the same
    variables).  ``local_device`` selects the HIP device of this rank first."""
    import datetime
    if local_device is not None and torch.cuda.is_available():
        torch.cuda.set_device(local_device)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return dist.group.WORLD


def comm_env() -> dict:
    """environment facts of the communicator set-up (for the bench line): the dmabuf IPC switch RCCL needs on this
    platform, who set it, the rendezvous backend, and which transport carries device operands"""
    d = dict(_comm_env)
    d["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    d["rendezvous_backend"] = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
    d["device_transport"] = ("libcatppo RCCL communicator" if _native is not None else
                             "torch nccl group (fallback)" if _device_fallback_group is not None else
                             "host staging over gloo" if d["rendezvous_backend"] == "gloo" else
                             "torch.distributed" if d["rendezvous_backend"] else "none")
    return d


def _ranks_share_device() -> bool:
    return os.environ.get("CATPPO_RANKS_SHARE_GPU", "0") == "1"


def init_native_comm(nat, group=None) -> bool:
    """Create libcatppo's own RCCL communicator (catppo_comm_init) for the ranks of the initialised
    torch.distributed world: rank 0 makes the unique id, the launcher's process group only ships its 128 bytes.
    Afterwards every device-tensor exchange point below is a catppo_allreduce / catppo_broadcast on the caller's
    HIP stream - capturable in a hipGraph, no torch.distributed on the data path.  CATPPO_NATIVE_COMM=0 keeps
    torch.distributed (RCCL through PyTorch) instead."""
    global _native
    if _native is not None:
        return True
    if not active(group):
        return False
    if os.environ.get("CATPPO_NATIVE_COMM", "1") == "0":
        _ensure_device_fallback(group)      # the switch is an environment variable of the job: every rank sees it
        return False
    # Every rank must end up on the same transport, and nobody may enter the (blocking) communicator set-up unless
    # everybody can: ncclCommInitRank waits for all ranks, so a rank that failed BEFORE it (librccl not loadable, a
    # communicator already present) would leave the healthy ones stuck inside RCCL.  Hence two votes over the
    # launcher's process group: (1) preconditions + the unique id, (2) the outcome of the set-up itself.  A "no" in
    # either keeps ALL ranks on torch.distributed - said on stderr and in native_comm_error(), never silently.
    global _native_error
    r, w = rank(group), world_size(group)
    pre_err, uid = None, None
    try:
        if nat.comm_world:
            pre_err = "a communicator already exists on this context"
        elif r == 0:
            uid = nat.comm_unique_id()          # dlopens librccl and starts RCCL's bootstrap root: rank 0 only
        else:
            nat.comm_probe()                    # dlopens librccl, nothing else
    except (RuntimeError, OSError) as e:
        pre_err = str(e)
    pre = [None] * w
    dist.all_gather_object(pre, (pre_err, uid if r == 0 else None), group=group)
    failed = [(i, v[0]) for i, v in enumerate(pre) if v[0] is not None]
    err = None
    if not failed:
        try:
            nat.comm_init(r, w, pre[0][1])
        except RuntimeError as e:
            err = str(e)
        votes = [None] * w
        dist.all_gather_object(votes, err, group=group)
        failed = [(i, v) for i, v in enumerate(votes) if v is not None]
        if failed and err is None:
            nat.comm_destroy()
    if not failed:
        # (3) known-answer pass of every collective the data path uses, on this communicator: the first real N-rank run
        # must not also be the first time anybody looks at what these calls return
        err = _selfcheck(nat, r, w)
        votes = [None] * w
        dist.all_gather_object(votes, err, group=group)
        failed = [(i, v) for i, v in enumerate(votes) if v is not None]
        if failed:
            nat.comm_destroy()
    if failed:
        _native_error = "; ".join(f"rank {i}: {v}" for i, v in failed)
        if r == 0:
            import sys
            print(f"[catppo] native RCCL communicator unavailable, collectives stay on torch.distributed: "
                  f"{_native_error}", file=sys.stderr)
        _ensure_device_fallback(group)
        return False
    _native = nat
    _native_error = None                  # (a set-up that succeeds after an earlier failure: the old reason is history)
    return True


def _ensure_device_fallback(group=None):
    """Rendezvous on gloo and no native communicator: device operands need a device transport after all.  Every rank
    reaches this point together (it follows a collective vote), so the collective ``new_group`` is safe.  Ranks that
    share a GPU keep the host-staged path (RCCL refuses two ranks on one device)."""
    global _device_fallback_group
    if _device_fallback_group is not None or not dist.is_initialized() or world_size(group) < 2:
        return
    if dist.get_backend(group) != "gloo" or not torch.cuda.is_available():
        return
    # two ranks on one device (a one-GPU box: the 2-rank trainer tests, bench.py's launcher proof) cannot form an RCCL
    # group: every rank publishes which physical device it sits on, and any duplicate keeps ALL ranks on host staging
    import socket
    dev = torch.cuda.current_device()
    ident = (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")),
             str(getattr(torch.cuda.get_device_properties(dev), "uuid", dev)), dev)
    idents = [None] * world_size(group)
    dist.all_gather_object(idents, ident, group=group)
    _comm_env["ranks_share_a_device"] = len(set(idents)) < len(idents)
    if _ranks_share_device() or _comm_env["ranks_share_a_device"]:
        return
    try:
        _device_fallback_group = dist.new_group(backend="nccl")
    except Exception as e:                                      # stay on host staging: slow, correct
        import sys
        print(f"[catppo] rank {rank(group)}: torch nccl fallback group unavailable ({e}); device operands are staged "
              f"through host memory", file=sys.stderr)


#: seconds the known-answer pass may take before the communicator is declared unusable (first RCCL call: lazy channel set-up)
_SELFCHECK_TIMEOUT_S = float(os.environ.get("CATPPO_COMM_SELFCHECK_TIMEOUT", "180"))


def _selfcheck(nat, r: int, w: int):
    """SUM (fp32, fp64, fp16), MAX, broadcast and all-gather with closed-form answers on the fresh communicator; returns
    None or what went wrong.  Waits with a deadline (stream query, no blocking synchronise): a collective that never
    completes is reported instead of hanging the job before its first line of output."""
    import time
    try:
        dev = nat.device
        n = 1027                                                   # not a multiple of anything RCCL slices by
        i = torch.arange(n, device=dev, dtype=torch.float64)
        s32 = ((i % 7) + 1 + r).float()
        s64 = (i + 1) * (r + 1) + 1e-9 * r
        s16 = torch.full((n,), float(r % 3), device=dev, dtype=torch.float16)
        m32 = ((i * 31 + 17 * r) % 101).float()
        b32 = torch.full((n,), float(r + 5), device=dev)
        rec = torch.full((24,), r, device=dev, dtype=torch.uint8)
        got = torch.empty(24 * w, device=dev, dtype=torch.uint8)
        nat.allreduce(s32, 0), nat.allreduce(s64, 0), nat.allreduce(s16, 0), nat.allreduce(m32, 1)
        nat.broadcast(b32, w - 1), nat.allgather(rec, got)
        if dev.type == "cuda":
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            t0 = time.monotonic()
            while not done.query():
                if time.monotonic() - t0 > _SELFCHECK_TIMEOUT_S:
                    return f"known-answer collectives did not complete within {_SELFCHECK_TIMEOUT_S:.0f} s"
                time.sleep(0.002)
        ranks = torch.arange(w, device=dev, dtype=torch.float64)
        exp32 = (w * ((i % 7) + 1) + ranks.sum()).float()
        exp64 = (i + 1) * (ranks + 1).sum() + 1e-9 * ranks.sum()      # the per-rank perturbation is part of the answer
        exp16 = float(sum(k % 3 for k in range(w)))
        expm = torch.stack([(i * 31 + 17 * k) % 101 for k in range(w)]).max(0).values.float()
        bad = []
        if not torch.equal(s32, exp32):
            bad.append("SUM fp32")
        if not torch.allclose(s64, exp64, rtol=1e-12, atol=1e-10):
            bad.append("SUM fp64")
        if not bool((s16.float() == exp16).all()):
            bad.append("SUM fp16")
        if not torch.equal(m32, expm):
            bad.append("MAX fp32")
        if not bool((b32 == float(w - 1 + 5)).all()):
            bad.append("broadcast")
        if not torch.equal(got.view(w, 24), torch.arange(w, device=dev, dtype=torch.uint8)[:, None].expand(w, 24)):
            bad.append("all-gather")
        return ("wrong result of " + ", ".join(bad)) if bad else None
    except RuntimeError as e:
        return str(e)


def native_comm_error():
    """why init_native_comm() fell back to torch.distributed (None when it did not)"""
    return _native_error


def shutdown_native_comm():
    global _native, _device_fallback_group
    if _native is not None:
        _native.comm_destroy()
        _native = None
    _device_fallback_group = None          # destroyed with the process group


def reinit_native_comm(group=None) -> bool:
    """Destroy libcatppo's communicator and build a fresh one (collective: every rank must call it).  Used after a
    stream capture that enqueued collectives was aborted on some rank: that rank's communicator state may be ahead of
    its peers' (ADVICE r4), so nobody keeps using the old one."""
    global _native
    nat = _native
    if nat is None:
        return False
    nat.comm_destroy()
    _native = None
    return init_native_comm(nat, group)


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


def _host_staged(t: torch.Tensor, group) -> bool:
    """device tensor on a process group whose backend has no device transport here (gloo: the CPU tests' backend,
    and the way two ranks share ONE GPU in the 2-rank trainer test - RCCL refuses two ranks on one device): the
    operand takes a round trip through host memory.  Correct, deterministic and slow; never the production transport."""
    # decided per group: stage through host memory unless the group that will actually carry `t` - the
    # caller's, or the nccl fallback that stands in for the gloo WORLD group - has a device transport.  (A device operand
    # on a gloo SUB-group, e.g. a custom obs_group / dist_group, used to go to gloo unstaged once the fallback existed.)
    return t.is_cuda and dist.get_backend(_torch_group(t, group)) == "gloo"


def _torch_group(t: torch.Tensor, group):
    """the torch.distributed group that carries ``t``: the nccl fallback group for device operands of the (gloo) world
    group when it exists, else the caller's group"""
    if t.is_cuda and _device_fallback_group is not None and group in (None, dist.group.WORLD):
        return _device_fallback_group
    return group


#: [(kind, bytes, start event, end event)] while comm_timing_begin() ... comm_timing_end() brackets a region (bench.py's
#: ``comm_ms_per_iteration``), else None.  HIP events on the stream the collective is enqueued on.
_timing = None


def comm_timing_begin():
    """start recording one HIP-event pair around every device-tensor exchange point issued from Python"""
    global _timing
    _timing = []


def comm_timing_end():
    """stop recording; returns {"ms": total device milliseconds between the event pairs, "calls": n, "bytes": operand
    bytes, "by_kind": {kind: [ms, calls]}} (synchronises)"""
    global _timing
    rec, _timing = _timing or [], None
    if rec:
        torch.cuda.synchronize()
    out = {"ms": 0.0, "calls": len(rec), "bytes": 0, "by_kind": {}}
    for kind, nbytes, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        out["ms"] += ms
        out["bytes"] += nbytes
        k = out["by_kind"].setdefault(kind, [0.0, 0])
        k[0] += ms
        k[1] += 1
    return out


class _timed:
    """brackets one exchange point with HIP events when comm timing is on (device operands only)"""

    def __init__(self, kind, t):
        # (an event recorded inside a stream capture belongs to the graph: it cannot be timed - the captured update phase
        # is measured by bench.py's un-graphed pass instead)
        self.on = _timing is not None and t.is_cuda and not torch.cuda.is_current_stream_capturing()
        self.kind, self.nbytes = kind, t.numel() * t.element_size()

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            if _timing is not None:
                _timing.append((self.kind, self.nbytes, self.e0, self.e1))
        return False


def _collective(t: torch.Tensor, group, native_call, torch_call, kind="allreduce"):
    if not active(group):
        return t
    with _timed(kind, t):
        if _use_native(t, group):
            native_call(t)
        elif _host_staged(t, group):
            h = t.detach().cpu()
            torch_call(h)
            t.copy_(h)
        else:
            torch_call(t)
    return t


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    return _collective(t, group, lambda x: _native.allreduce(x, 0),
                       lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=_torch_group(x, group)))


def allreduce_max_(t: torch.Tensor, group=None) -> torch.Tensor:
    return _collective(t, group, lambda x: _native.allreduce(x, 1),
                       lambda x: dist.all_reduce(x, op=dist.ReduceOp.MAX, group=_torch_group(x, group)))


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    return _collective(t, group, lambda x: _native.broadcast(x, src),
                       lambda x: dist.broadcast(x, src=src, group=_torch_group(x, group)), kind="broadcast")


def allgather_bytes_(send: torch.Tensor, recv: torch.Tensor, group=None) -> torch.Tensor:
    """recv (uint8 [world * n]) <- the uint8 [n] record of every rank, in rank order: ONE collective for a record that
    mixes reductions (the fused env step's {column maxima: MAX | moment sums: SUM}); the consumer folds the records
    itself, in rank order, so every rank computes bit-identical results"""
    if not active(group):
        recv[:send.numel()].copy_(send)
        return recv
    with _timed("allgather", recv):
        if _native is not None and send.is_cuda and group in (None, dist.group.WORLD):
            _native.allgather(send, recv)
        elif _host_staged(send, group):
            h = torch.empty(recv.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(h, send.detach().cpu(), group=group)
            recv.copy_(h)
        else:
            dist.all_gather_into_tensor(recv, send, group=_torch_group(send, group))
    return recv


def gather_counts(n_local: int, device=None, group=None) -> list:
    """[n of rank 0, n of rank 1, ...] as Python ints (one small SUM all-reduce; world of one: [n_local])"""
    w, r = world_size(group), rank(group)
    if not active(group):
        return [int(n_local)]
    v = torch.zeros(w, dtype=torch.float64, device=device)
    v[r] = float(n_local)
    allreduce_sum_(v, group)
    return [int(round(x)) for x in v.cpu().tolist()]


def minibatch_plan(batch_rows, minibatch: int):
    """Minibatch schedule of an env-sharded update that every rank can derive from the same inputs.

    ``batch_rows`` = rows (T x N_r) of every rank, ``minibatch`` = nominal per-rank minibatch size.  Shards may differ
    (ragged env counts), so ``ceil(B_r / minibatch)`` can differ by one between ranks - and a rank that issues one
    more gradient all-reduce than its peers hangs the job.  When the per-rank counts agree the configured size is kept
    (a single process gets the reference's schedule); otherwise the plan fixes ONE minibatch count for all ranks (the
    smallest per-rank count), gives rank r minibatches of ``M_r = ceil(B_r / n_mb)`` rows (the last one may be
    shorter) and returns the TRUE number of rows of every global minibatch k (sum over ranks) for the 1/M_global loss
    scaling.  Returns (n_mb, [M_r], [[rows of minibatch k on rank r]], [global rows of minibatch k])."""
    rows = [int(b) for b in batch_rows]
    if min(rows) < 1 or minibatch < 1:
        raise ValueError(f"minibatch_plan: every rank needs at least one row (rows per rank {rows}, minibatch {minibatch})")
    nominal = [min(minibatch, b) for b in rows]
    counts = [(b + m - 1) // m for b, m in zip(rows, nominal)]
    n_mb = min(counts)
    if all(c == n_mb for c in counts):
        m_r = nominal                      # every rank agrees: the configured schedule (single process: the reference's)
    else:
        m_r = [(b + n_mb - 1) // n_mb for b in rows]
    per_rank = [[max(0, min(m, b - k * m)) for k in range(n_mb)] for b, m in zip(rows, m_r)]
    if any(x == 0 for pr in per_rank for x in pr):
        raise ValueError(f"minibatch_plan: shards {rows} cannot be cut into {n_mb} non-empty minibatches each; "
                         "use env counts per rank that differ by at most one, or a smaller minibatch_size")
    glob = [sum(pr[k] for pr in per_rank) for k in range(n_mb)]
    return n_mb, m_r, per_rank, glob


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
