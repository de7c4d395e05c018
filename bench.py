"""bench.py - env-steps/s of the CaT-PPO hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps 20 --warmup 5          (N > 1 and no launcher around it: bench.py starts the N ranks
                                                             itself, `torch.distributed.run --nnodes=1 --nproc-per-node N`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (the driver's form; same result)
A line is only printed when the ranks that ran equal --gpus.  More ranks than visible GPUs (a one-GPU box) is allowed as
a LAUNCHER PROOF, not a scaling number: the ranks then share devices and exchange through gloo with host staging (RCCL
refuses two ranks on one device), and the line says so (`physical_gpus`, `ranks_share_gpus`, `collectives`).

A "step" is ONE full CaT-PPO iteration (BASELINE.json metric): T x [policy/value forward + Philox action
sample, constraint-term evaluation + CaT step + reward / dones epilogue + reset statistics, rollout-buffer rows, obs
normaliser update + apply] + bootstrap value + GAE + value normaliser x2 + E epochs x ceil(N*T/M) minibatch steps
(keyed-permutation gather, forward, losses, backward, [RCCL gradient all-reduce], global-norm clip, Adam) + the
per-iteration diagnostics read-back the reference's logging performs (ppo.py:356-367).  Synthetic streams are generated
before the timed region and are device resident.  Workload (default) = BASELINE.json configs[1]: 4096 envs x 24,
Solo12, 6 constraint terms (42 columns), 48-d obs, 3x256 MLP;  --workload reference runs the reference's own shapes
(45-d obs, 13 terms / 78 columns, 512/256/128 MLPs).

Scaling (--gpus N > 1): default = WEAK (every rank owns the workload's envs and minibatch share: global minibatch =
16384 x N) - and the same line carries a `strong` record: cfg3 measured by the same process group right behind the weak
timed region (value, per-rank times, wire time, RCCL world, rank 0's roofline fraction; --no-strong skips it).  --workload cfg3 is BASELINE configs[2] as written - 16384 envs and a 16384-sample minibatch GLOBAL, sharded
over the ranks - i.e. STRONG scaling of a fixed job.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL / device-tensor sharing fail with
# "hipIpcGetMemHandle: invalid argument" otherwise): in the environment BEFORE the HIP runtime comes up in this process
if os.environ.get("CATPPO_BENCH_IPC_UNSET") != "1":
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        os.environ["CATPPO_BENCH_IPC_DEFAULTED"] = "1"        # (self_launch: the value is ours, not the user's)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(num_envs=4096, num_steps=24, obs_dim=48, hidden=(256, 256, 256), six_terms=True,
                 desc="4096 envs x 24, Solo12 48-d obs, 6 ConstraintTerms (42 cols), 3x256 MLP, 5 epochs x 6 minibatches of 16384"),
    # the reference's real shapes
    "reference": dict(num_envs=4096, num_steps=24, obs_dim=45, hidden=(512, 256, 128), six_terms=False,
                      desc="4096 envs x 24, Solo12 45-d obs, 13 ConstraintTerms (78 cols), 512/256/128 MLPs, 5 epochs x 6 minibatches of 16384"),
    # the other BASELINE.json configs (parity-test cases; measured for orientation, never the headline line)
    "cfg1": dict(num_envs=64, num_steps=24, obs_dim=48, hidden=(512, 256, 128), six_terms="two", minibatch=512,
                 desc="64 envs x 24, 48-d obs, 2 ConstraintTerms, reference MLP (plumbing case)"),
    "cfg3": dict(num_envs=16384, num_steps=24, obs_dim=45, hidden=(512, 256, 128), six_terms=False, strong=True,
                 desc="BASELINE configs[2] as written: 16384 envs x 24 GLOBAL, full ConstraintsCfg, 512/256/128 MLPs, "
                      "global minibatch 16384 - envs and minibatch sharded over the ranks (strong scaling)"),
    "cfg3_shard": dict(num_envs=2048, num_steps=24, obs_dim=45, hidden=(512, 256, 128), six_terms=False, minibatch=2048,
                       desc="one rank's share of config 3 at 8 GPUs: 2048 envs x 24, full ConstraintsCfg, 512/256/128 MLPs, minibatches of 2048"),
    "cfg4": dict(num_envs=4096, num_steps=48, obs_dim=235, hidden=(256, 256, 256), six_terms=True,
                 desc="4096 envs x 48, 235-d obs (48 + 187 height scan), 6 ConstraintTerms, 3x256 MLP, minibatches of 16384"),
    "cfg5_envs": dict(num_envs=32768, num_steps=24, obs_dim=48, hidden=(256, 256, 256), six_terms=False,
                      desc="32768 envs x 24, 48-d obs, 13 ConstraintTerms with mixed hard/soft max_p, 3x256 MLP, minibatches of 16384 (fp32 planes / fp32 MFMA)"),
    "cfg5": dict(num_envs=32768, num_steps=24, obs_dim=48, hidden=(256, 256, 256), six_terms=False,
                 rollout_dtype="fp16", mlp_precision="bf16",
                 desc="BASELINE configs[4]: 32768 envs x 24, fp16 rollout planes + bf16-operand MLP MFMA, 13 ConstraintTerms "
                      "with mixed hard/soft max_p, 3x256 MLP, minibatches of 16384 (reduced precision: not the headline)"),
}
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (no 2:1 sparsity)


def fwd_macs(obs_dim, hidden, act_dim=12):
    dims = [obs_dim, *hidden]
    body = sum(i * o for i, o in zip(dims[:-1], dims[1:]))
    return 2 * body + hidden[-1] * (act_dim + 1)


def executed_macs(obs_pad, hidden, head_pad=16):
    """multiply-accumulates per sample the MATRIX PIPE executes in one forward + backward of both networks, as opposed to
    the algorithmic 3 x forward of SURVEY 8(d): observations padded to a multiple of 16, the first layer's data gradient
    never computed (nobody needs d loss / d obs), head products run as 16-output MFMA tiles inside fwd_head_kernel
    (forward, head weight gradient, dZ of the last layer)"""
    dims = [obs_pad, *hidden]
    body = sum(i * o for i, o in zip(dims[:-1], dims[1:]))          # one network
    heads = head_pad * hidden[-1]                                    # one network, one of the three head products
    fwd = 2 * (body + heads)
    dw = 2 * (body + heads)
    dx = 2 * (body - obs_pad * hidden[0] + heads)
    return fwd + dw + dx


def csrc_hash():
    """identity of the kernel sources: PMC traffic summaries under profiles/ are stamped with it"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "constraints-as-terminations_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def shard_of(w, world, rank):
    """(envs, minibatch) of this rank"""
    from cat_envs import parallel
    mb = w.get("minibatch", 16384)
    if not w.get("strong"):
        return w["num_envs"], mb
    sl = parallel.shard_slice(w["num_envs"], rank, world)
    return sl.stop - sl.start, max(mb // world, 1)


def build(workload, seed, device_index, mlp_precision=None, world=1, rank=0, overrides=None):
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    w = WORKLOADS[workload]
    n_envs, mb = shard_of(w, world, rank)
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(n_envs, w["num_steps"], mb, 5, 2000, w["hidden"],
                                                    w["six_terms"], obs_dim=w["obs_dim"], stream_steps=48, seed=seed)
    env_cfg.sim.device = f"cuda:{device_index}"
    agent_cfg.mlp_precision = mlp_precision or w.get("mlp_precision", "fp32")
    agent_cfg.rollout_dtype = w.get("rollout_dtype", "fp32")
    for k, v in (overrides or {}).items():
        setattr(agent_cfg, k, v)
    env = make(task, cfg=env_cfg)
    trainer = PPOTrainer(env, agent_cfg)
    return env, trainer, agent_cfg


def pick_cpu_threads(obs_dim, hidden):
    """torch-CPU thread count with the best throughput on this host for the dominant op (an MLP
    forward/backward on 16384 rows); one thread per core is far from optimal on a 256-core host."""
    import torch.nn as nn
    dims = [obs_dim, *hidden]
    net = nn.Sequential(*[m for i, o in zip(dims[:-1], dims[1:]) for m in (nn.Linear(i, o), nn.ELU())])
    x = torch.randn(16384, obs_dim)
    best, best_t = 1, float("inf")
    for n in (8, 16, 32, 64, 128):
        if n > (os.cpu_count() or 1):
            break
        torch.set_num_threads(n)
        net(x).sum().backward()
        t0 = time.perf_counter()
        for _ in range(2):
            net(x).sum().backward()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = n, dt
    return best


def cpu_baseline(workload, trainer, env, agent_cfg, budget_s=15.0):
    """the oracle (torch-CPU restatement of PPO() + numpy CaT) timed on the host cores, rank 0 / N=1 only,
    on a bounded sample of the same workload: whole iterations (24 steps + GAE + 5 epochs x 6 minibatches)
    after one untimed warm-up iteration, until ~budget_s seconds of CPU work have been measured"""
    from oracle import env_oracle, ppo_oracle
    w = WORKLOADS[workload]
    cores = pick_cpu_threads(trainer.D, w["hidden"])
    torch.set_num_threads(cores)
    cpu_env = env_oracle.from_device_env(env)
    ag = ppo_oracle.AgentOracle(trainer.D, trainer.A, w["hidden"], seed=0)
    cfg = {k: getattr(agent_cfg, k) for k in ppo_oracle.PPOOracle.DEFAULT_CFG}
    orc = ppo_oracle.PPOOracle(cpu_env, trainer.N, trainer.D, trainer.A, cfg=cfg, hidden=w["hidden"], agent=ag)
    orc.run_iteration()                                   # warm-up (allocator, thread pool)
    for k in orc.timers:
        orc.timers[k] = 0.0
    n, t0 = 0, time.perf_counter()
    while n < 8 and (time.perf_counter() - t0) < budget_s:
        orc.run_iteration()
        n += 1
    dt = time.perf_counter() - t0
    tm = orc.timers
    steps = trainer.N * w["num_steps"] * n
    # "cores" = the torch threads this baseline actually used (the contract's meaning); "host_cores" = what the box has
    return {"value": steps / dt, "unit": "env-steps/s", "cores": cores, "threads": cores,
            "host_cores": os.cpu_count(), "thread_choice": "best of {8,16,32,64,128} torch threads on a 16384-row MLP "
                                                           "forward/backward probe (one thread per core is slower on "
                                                           "this host)", "kind": "port",
            "sample": f"{n} full iterations of {trainer.N}x{w['num_steps']} (5 epochs, minibatches of "
                      f"{trainer.M}) after 1 warm-up iteration, {dt:.1f} s wall",
            "phase_ms_per_iteration": {"rollout_fwd_and_env": 1e3 * tm["rollout"] / n, "cat_env_step": 1e3 * tm["env"] / n,
                                       "gae": 1e3 * tm["gae"] / n, "update": 1e3 * tm["update"] / n}}


def pmc_traffic(workload, tag):
    """HBM bytes per launch of the dominant kernel group from the committed PMC summary (separate rocprofv3 --pmc
    passes cannot run inside the timed process).  The summary carries the hash of the kernel sources it was measured
    on: a stale one is reported as null, not as a number."""
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_{workload}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None, f"no PMC summary at profiles/{os.path.basename(path)}"
    if d.get("csrc_hash") != csrc_hash():
        return None, None, (f"profiles/{os.path.basename(path)} was measured on kernel sources {d.get('csrc_hash')}, the "
                            f"tree is {csrc_hash()}: re-run tools/profile_bench.sh")
    return d.get("hbm_bytes_per_launch"), os.path.basename(path), None


GROUP_KERNELS = ("gemm_f32_kernel", "gemm_pair_kernel", "head_loss_kernel", "fwd_head_kernel", "seg_reduce_kernel",
                 "rows_fwd_kernel", "dw_fold_kernel", "rows_fwd_wide_kernel", "step16_kernel", "dw_multi_kernel")


def profiled_group_us(workload, tag):
    """average duration of the dominant kernel group per minibatch from the COMMITTED rocprofv3 kernel trace of the same
    command (profiles/<tag>_bench_<workload>_kernel_stats.csv): sum over the group's kernels of total time / number of
    minibatches (one fold launch per minibatch), the rollout-only GEMM shapes excluded like tools/pmc_group_traffic.py
    does.  Lets the bench line and profiles/ be diffed mechanically (should agree with avg_launch_us within a few %)."""
    import csv
    path = os.path.join(ROOT, "profiles", f"{tag}_bench_{workload}_kernel_stats.csv")
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError:
        return None, None
    try:
        n_mb = next(int(r["calls"]) for r in rows if "seg_reduce" in r["kernel"])      # one (final) fold launch per minibatch
    except StopIteration:
        return None, os.path.basename(path)
    def blocks(r):
        x, y, z = (int(v) for v in r["blocks"].split("x"))
        return x * y * z
    us = 0.0
    for r in rows:
        if not any(k in r["kernel"] for k in GROUP_KERNELS):
            continue
        if "<64, 64, true, true, 0, 64" in r["kernel"] or "policy_fwd" in r["kernel"] or "rows_fwd_kernel<32" in r["kernel"] \
                or "rows_fwd_wide_kernel<32" in r["kernel"]:
            continue                                   # rollout-only launches (<= 4096 rows)
        if "<64, 64, true, true, 0" in r["kernel"]:    # first-layer forward: the rollout uses the same kernel on fewer rows
            if blocks(r) < max(blocks(q) for q in rows if q["kernel"] == r["kernel"]):
                continue
        us += float(r["total_us"])
    return us / n_mb, os.path.basename(path)


def gae_roofline(nat, T, N, reps=50, mode=None):
    from cat_envs import native
    dev = "cuda"
    x = [torch.rand(T, N, device=dev) for _ in range(4)]
    nv, nd, ntd = (torch.rand(N, device=dev) for _ in range(3))
    adv, ret = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
    if mode is None:
        f = lambda: nat.gae(x[0], x[1], x[2], x[3], nv, nd, ntd, 0.99, 0.95, adv, ret)
    else:
        f = lambda: nat.gae_mode(mode, x[0], x[1], x[2], x[3], nv, nd, ntd, 0.99, 0.95, adv, ret)
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = 24.0 * T * N + 12.0 * N
    return {"T": T, "N": N, "us": us, "GBps": byt / us / 1e3, "frac": byt / us / 1e3 / HBM_PEAK_GBPS,
            "mode": {None: "serial_exact", native.GAE_SCAN: "scan", native.GAE_SERIAL: "serial_exact"}[mode]}


def time_group_eager(trainer, reps=12, skip=2):
    """average device time (us) of catppo_ppo_minibatch_grad_packed on the data of the trainer's last iteration: HIP events
    around `reps` eager calls on the current stream, the first `skip` dropped"""
    nat = trainer.nat
    trainer._update_buffers()
    # (env-sharded runs normalise advantages with the statistics of the GLOBAL minibatch: hand the kernel the first pair
    # the last iteration computed)
    adv_stats = trainer._adv_stats_all[0] if trainer.hp.adv_stats_external else None
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.ppo_minibatch_grad_packed(trainer.agent.shape, trainer.hp, trainer.agent.flat, trainer._x_g, trainer._act_g,
                                      trainer._scal_g, trainer._advp_g, trainer.M, trainer.agent.value_rms.running_mean,
                                      trainer.agent.value_rms.running_var, adv_stats, trainer.grad, trainer.diag)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ev = ev[skip:]
    return float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3, len(ev)


def strong_record(a, world, rank, dev_index, shared, steps, warmup):
    """`--gpus N` (N > 1) prints the WEAK line the driver's contract asks for; BASELINE configs[2] - 16384 envs and a
    16384-row minibatch GLOBAL, sharded over the ranks (solo12/agents/clean_rl_ppo_cfg.py:20; exchange semantics of the
    reference's only distributed trainer, skrl/ppo.py:534-537) - is a different hyper-parameter set, so the same process
    group measures it right behind the weak timed region and the line carries it as `strong`: value, per-rank times,
    serialised wire time, the RCCL world, rank 0's minibatch-group time and roofline fraction.  Every rank runs this."""
    import contextlib
    from cat_envs import parallel
    w = WORKLOADS["cfg3"]
    with contextlib.redirect_stdout(sys.stderr):
        env, tr, cfg = build("cfg3", a.seed + rank, dev_index, None, world, rank, {})

    def barrier():
        torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        tr.run_iteration(log=True)
    tr.time_phases = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.run_iteration(log=True)
    barrier()
    dt = time.perf_counter() - t0
    phases = tr.phase_summary()
    tr.time_phases = False
    times = [None] * world
    torch.distributed.all_gather_object(times, dt)
    dt = max(times)
    # serialised wire time of one iteration (as the weak record's `comm`): un-graphed, un-overlapped pass
    g_was, ov_was = tr.graph_update, tr.grad_overlap
    tr.graph_update = False
    if ov_was:
        tr.grad_overlap = tr.nat.set_grad_overlap(False)
    tr.run_iteration(log=True)
    barrier()
    parallel.comm_timing_begin()
    tr.run_iteration(log=True)
    comm = parallel.comm_timing_end()
    barrier()
    tr.graph_update = g_was
    rec = None
    if rank == 0:
        grad_us, n_timed = time_group_eager(tr)        # (gradient buckets are off: no collective inside the call)
        macs = fwd_macs(w["obs_dim"], w["hidden"])
        flops = 3 * 2 * macs * tr.M
        rec = {"workload": "cfg3: " + w["desc"], "scaling": "strong", "value": tr.n_envs_global * w["num_steps"] * steps / dt,
               "unit": "env-steps/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
               "per_rank_ms": {"min": 1e3 * min(times) / steps, "max": 1e3 * max(times) / steps,
                               "by_rank": [1e3 * x / steps for x in times]},
               "envs_per_gpu": tr.N, "envs_total": tr.n_envs_global, "minibatch_per_gpu": tr.M, "global_minibatch": tr.M * world,
               "rccl_world": tr.nat.comm_world if parallel.native_comm_active() else (world if not shared else 0),
               "comm_ms_per_iteration": comm["ms"], "collectives_per_iteration": comm["calls"],
               "phases_device_ms": phases, "graph_update": tr.graph_update, "graph_fallback": tr.graph_fallback,
               "roofline": {"bound": "mfma", "rank": 0, "avg_launch_us": grad_us, "launches_timed": n_timed,
                            "achieved": flops / grad_us / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": flops / grad_us / 1e6 / MFMA_F32_PEAK_TFLOPS,
                            "kernel": f"catppo_ppo_minibatch_grad_packed per {tr.M}-sample minibatch (rank 0; every rank runs the same shapes)"}}
    if ov_was:
        tr.grad_overlap = tr.nat.set_grad_overlap(getattr(tr, "_grad_overlap_mode", True))
    barrier()
    return rec


def secondary_record(a, w, dev_index, steps, warmup):
    """The same workload with split-bf16 GEMM operands (`mlp_precision="bf16x3"`: hi/lo bf16 planes, three bf16 MFMAs per
    product, ~16 mantissa bits - wider than the TF32 arithmetic the reference enables for these GEMMs,
    /root/reference/scripts/clean_rl/train.py:86-87; passes the fp32 parity bars of tests/test_gpu_r2_features.py), as a
    SECONDARY record next to the fp32 headline: a fresh trainer built and timed after the headline's timed region."""
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        env, tr, cfg = build(a.workload, a.seed, dev_index, "bf16x3", 1, 0, {})
    for _ in range(warmup):
        tr.run_iteration(log=True)
    tr.time_phases = True
    # two timed passes, their MEAN is the record (both are listed): the process holds a second trainer and has just
    # run the headline, and a host-side pause (a generation-2 garbage collection over two trainers' objects) inside one
    # pass of a few steps showed up as 10.1 against 8.1 ms per iteration between two runs of the same tree
    import gc
    gc.collect()
    passes = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.run_iteration(log=True)
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t0)
    dt = sum(passes) / len(passes)
    phases = tr.phase_summary()
    grp_us, n_ev = time_group_eager(tr)
    macs = fwd_macs(w["obs_dim"], w["hidden"])
    flops = 3 * 2 * macs * tr.M
    ach = flops / grp_us / 1e6
    traffic, src, note = pmc_traffic(a.workload + "_bf16x3", a.profile_tag)
    prof_us, prof_src = profiled_group_us(a.workload + "_bf16x3", a.profile_tag)
    return {"dtype": "bf16x3 (split-bf16 GEMM operands = 16 mantissa bits, f32 accumulate/params/activations; meets the "
                     "f32 parity tolerances)",
            "value": tr.N * w["num_steps"] * steps / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / steps,
            "steps": steps, "warmup": warmup, "phases_device_ms": phases,
            "passes_ms_per_step": [1e3 * p / steps for p in passes],
            "measured": "after the timed region of the headline, fresh trainer, same workload / seed; the mean of two "
                        "timed passes of `steps` iterations (both in passes_ms_per_step)",
            # `achieved` / `frac` count what the matrix pipe executes (three bf16 MFMAs per product) against the bf16
            # peak; `algorithmic_tflops` / `algorithmic_frac_of_f32_peak` are the figures comparable with the fp32 headline
            "roofline": {"bound": "mfma", "achieved": 3.0 * ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": 3.0 * ach / MFMA_BF16_PEAK_TFLOPS, "algorithmic_tflops": ach,
                         "algorithmic_frac_of_f32_peak": ach / MFMA_F32_PEAK_TFLOPS,
                         "executed_over_algorithmic_flops": 3.0, "avg_launch_us": grp_us, "launches_timed": n_ev,
                         "traffic": traffic, "traffic_source": src, "traffic_note": note,
                         "hbm_GBps": None if not traffic else traffic / grp_us / 1e3,
                         "hbm_frac": None if not traffic else traffic / grp_us / 1e3 / HBM_PEAK_GBPS,
                         "dominant_kernel_us_profiled": prof_us, "profiled_source": prof_src,
                         "frac_profiled": None if not prof_us else 3.0 * flops / prof_us / 1e6 / MFMA_BF16_PEAK_TFLOPS}}


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks of this node here -
    the same command line under torch.distributed.run, one process per rank, rendezvous on 127.0.0.1 (the container
    hostname may not resolve).  Rank 0's JSON line passes through on stdout; returns the launcher's exit code.
    Reference precedent for a self-contained distributed entry point: scripts/skrl/train.py:28-30,116-117."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC: this image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only supports dmabuf IPC; RCCL /
    # device-tensor sharing otherwise fail with "hipIpcGetMemHandle: invalid argument").  No run with more than one
    # RCCL rank has been possible on the one-GPU development boxes, so the value is a documented platform requirement,
    # not something measured here: a user's explicit setting always wins, and if the ranks fail with the default the
    # launch is retried ONCE with the variable removed - the line says which attempt produced it (config.comm_env).
    user_set = "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ and os.environ.get("CATPPO_BENCH_IPC_DEFAULTED") != "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env["CATPPO_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n_ranks} without a launcher: starting {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr)
    env["CATPPO_BENCH_LAUNCH_ATTEMPT"] = "1"
    rc = subprocess.call(cmd, env=env)
    if rc != 0 and not user_set and os.environ.get("CATPPO_BENCH_NO_RETRY") != "1":
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd[cmd.index("--master-port") + 1] = str(port)
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        env["CATPPO_BENCH_LAUNCH_ATTEMPT"] = "2"
        env["CATPPO_BENCH_IPC_UNSET"] = "1"
        print(f"[bench] the ranks exited with code {rc}; retrying once WITHOUT HSA_ENABLE_IPC_MODE_LEGACY", file=sys.stderr)
        rc = subprocess.call(cmd, env=env)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default cfg2 (BASELINE configs[1]); with --scaling strong: cfg3")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak (default): every rank owns the workload's envs and minibatch.  strong = --workload cfg3: "
                         "BASELINE configs[2] as written, 16384 envs and a 16384-row minibatch GLOBAL, sharded over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mlp-precision", choices=("fp32", "bf16", "bf16x3"), default=None,
                    help="fp32 = fp32-input MFMA (the headline metric).  bf16x3 = split-bf16 operands (3 bf16 MFMAs per "
                         "product, ~16 mantissa bits: meets the fp32 parity tolerances; reported as its own dtype against "
                         "the bf16 peak).  bf16 = operands rounded to bf16 (BASELINE config 5): NOT a parity mode")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="PPO cfg override, e.g. --set graph_update=True --set rng=torch --set fused_rollout=False")
    ap.add_argument("--profile-tag", default="r6", help="prefix of the PMC / kernel-trace summaries under profiles/")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary record (the same workload with split-bf16 GEMM operands, measured AFTER the "
                         "timed region of the fp32 headline; single process, fp32 workloads only)")
    ap.add_argument("--no-strong", action="store_true",
                    help="--gpus N > 1 with a weak workload: skip the `strong` record (cfg3 measured in the same process group "
                         "behind the weak timed region)")
    ap.add_argument("--shard-of", type=int, default=0, metavar="W",
                    help="single process, no collectives: run ONE rank's share of a strong-scaling workload as if the "
                         "world had W ranks (compute side of the scaling curve on a one-GPU box)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--print-launch", action="store_true",
                    help="print the torch.distributed.run command a bare `--gpus N` would start, and exit")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.workload is None:
        a.workload = "cfg3" if a.scaling == "strong" else "cfg2"
    if a.scaling is not None and (a.scaling == "strong") != bool(WORKLOADS[a.workload].get("strong")):
        ap.error(f"--scaling {a.scaling} contradicts --workload {a.workload} (strong scaling is --workload cfg3)")

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        if a.print_launch:
            print(" ".join([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                            "--master-addr", "127.0.0.1", "--master-port", "<free port>", os.path.abspath(__file__),
                            *[x for x in sys.argv[1:] if x != "--print-launch"]]))
            return 0
        return self_launch(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        # never a line whose n_gpus differs from what was asked for
        if rank == 0:
            print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to run "
                  f"(use `python bench.py --gpus {a.gpus}` or --nproc-per-node {a.gpus})", file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        print("[bench] no HIP device visible: the hot path runs on MI355X only (no CPU fallback)", file=sys.stderr)
        return 3
    shared = world > n_dev                # more ranks than GPUs: launcher proof on a small box, ranks share devices
    dev_index = local % n_dev
    if world > 1 or os.environ.get("CATPPO_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(dev_index)
        if shared:
            # RCCL refuses two ranks on one device: device operands staged through host memory by cat_envs.parallel (the
            # transport of tests/test_gpu_two_rank_trainer.py) - correct, slow, never production
            os.environ["CATPPO_NATIVE_COMM"] = "0"
            os.environ["CATPPO_RANKS_SHARE_GPU"] = "1"
        # The rendezvous (unique-id exchange, votes, barriers) is a gloo (CPU) process group: the ONLY RCCL communicator
        # of this process is libcatppo's own (catppo_comm_init), which carries every data-path collective.  (Round 4
        # initialised torch's "nccl" group here just to ship 128 bytes: a second RCCL bootstrap on the first real run.)
        from cat_envs import parallel as _par
        _par.init_rendezvous(dev_index)
    else:
        torch.cuda.set_device(0)

    import ast
    overrides = {}
    for kv in a.set:
        k, _, v = kv.partition("=")
        try:
            overrides[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            overrides[k] = v
    w = WORKLOADS[a.workload]
    shard_world = a.shard_of if (a.shard_of > 0 and world == 1) else world
    if (shard_world > 1 and world == 1) or a.workload.endswith("_shard"):
        # one rank's share of an env-sharded job run as a single process: a real rank has its gradient all-reduce between
        # the fold and the clip, so it takes the two-call optimiser step - and so does its stand-in
        overrides.setdefault("one_call_step", False)
    # stdout carries ONE line (the JSON record): what building the env prints (the reference's "[INFO] Constraint
    # Manager" table) goes to stderr
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        env, trainer, agent_cfg = build(a.workload, a.seed + rank, dev_index, a.mlp_precision, shard_world, rank, overrides)
    nat = trainer.nat
    from cat_envs import parallel

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.run_iteration(log=True)

    # ---- timed region: exactly K iterations between barrier + synchronize
    # HIP events bracket every catppo_ppo_minibatch_grad_packed call (the dominant kernel group) on the stream it is
    # enqueued on (the trainer's current stream).  When the update phase is replayed from a hipGraph there are no
    # per-call host launches to bracket: the group is then timed by an eager replay right after the timed region.
    ev = []
    orig = nat.ppo_minibatch_grad_packed

    def timed_grad(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(*args, **kw)
        e1.record()
        ev.append((e0, e1))
    if not trainer.graph_update:
        nat.ppo_minibatch_grad_packed = timed_grad
    trainer.time_phases = True
    barrier()
    t0 = time.perf_counter()
    # (CATPPO_BENCH_TIMED_LOG=0: A/B of what the read-back costs - never the reported configuration)
    timed_log = os.environ.get("CATPPO_BENCH_TIMED_LOG", "1") != "0"
    for _ in range(a.steps):
        trainer.run_iteration(log=timed_log)      # log=True: the per-iteration diagnostics read-back is inside the metric
    barrier()
    dt = time.perf_counter() - t0
    nat.ppo_minibatch_grad_packed = orig
    phases = trainer.phase_summary()
    trainer.time_phases = False
    per_rank_ms = None
    if world > 1:
        # MAX over ranks is the job's time; min / max / every rank's own time make stragglers visible (CPU gather over
        # the rendezvous group: not on the data path)
        times = [None] * world
        torch.distributed.all_gather_object(times, dt)
        per_rank_ms = {"min": 1e3 * min(times) / a.steps, "max": 1e3 * max(times) / a.steps,
                       "by_rank": [1e3 * x / a.steps for x in times]}
        dt = max(times)
    env_total = trainer.n_envs_global if world > 1 else float(trainer.N)
    steps_total = env_total * w["num_steps"] * a.steps
    value = steps_total / dt

    # the same K iterations without the read-back (host free to run ahead), for comparison only
    barrier()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        trainer.run_iteration(log=False)
    barrier()
    dt_nolog = time.perf_counter() - t1

    # device time inside the exchange points of one iteration (all-gather of the env-step record, normaliser / advantage
    # moments, gradient all-reduce, diagnostics): HIP events around every collective cat_envs.parallel issues, on the
    # stream it is issued on.  Collectives captured in the update-phase graph, and gradient buckets reduced inside the
    # library beside the backward launches, cannot be bracketed from here: this pass runs the same iteration with both
    # switched off, i.e. it reports the SERIALISED wire time (what overlap can at most hide), outside the timed region.
    comm = None
    if parallel.active():
        g_was, ov_was = trainer.graph_update, trainer.grad_overlap
        trainer.graph_update = False
        if ov_was:
            trainer.grad_overlap = nat.set_grad_overlap(False)
        trainer.run_iteration(log=True)
        barrier()
        parallel.comm_timing_begin()
        n_comm_it = 2
        for _ in range(n_comm_it):
            trainer.run_iteration(log=True)
        comm = parallel.comm_timing_end()
        comm["iterations"] = n_comm_it
        barrier()
        trainer.graph_update = g_was
        if ov_was:
            trainer.grad_overlap = nat.set_grad_overlap(getattr(trainer, '_grad_overlap_mode', True))

    strong = None
    if world > 1 and not w.get("strong") and not a.no_strong:
        strong = strong_record(a, world, rank, dev_index, shared, max(2, a.steps // 2), max(1, a.warmup // 2))

    if rank == 0:
        M = trainer.M
        if trainer.graph_update or not ev:         # eager replay of the group on the data of the last iteration
            # (also when the eager trainer took the one-call optimiser step, which the wrapper above does not see)
            # rank 0 is alone here: with gradient buckets on, the gradient call itself would all-reduce - and wait for
            # peers that are already at the final barrier
            ov_replay = bool(trainer.grad_overlap)
            if ov_replay:
                nat.set_grad_overlap(False)
            grad_us, n_timed = time_group_eager(trainer)
            if ov_replay:
                nat.set_grad_overlap(getattr(trainer, '_grad_overlap_mode', True))
        else:
            grad_us, n_timed = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3, len(ev)
        macs = fwd_macs(w["obs_dim"], w["hidden"])
        flops_per_launch = 3 * 2 * macs * M                       # fwd + bwd = 3x fwd, per minibatch
        ach = flops_per_launch / grad_us / 1e6
        prec = agent_cfg.mlp_precision
        bf16 = prec != "fp32"
        peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
        # bf16x3 issues three bf16 MFMAs per algorithmic product: the matrix pipe executes 3x the algorithmic FLOPs
        mfma_flops_factor = 3.0 if prec == "bf16x3" else 1.0
        # PMC / trace summaries of a non-default precision carry it in their name: r5_pmc_traffic_cfg2_bf16x3.json
        wl_key = a.workload if prec == w.get("mlp_precision", "fp32") else f"{a.workload}_{prec}"
        traffic, traffic_src, traffic_note = pmc_traffic(wl_key, a.profile_tag)
        if traffic_note:
            print(f"[bench] roofline.traffic = null: {traffic_note}", file=sys.stderr)
        prof_us, prof_src = profiled_group_us(wl_key, a.profile_tag)
        n_mb_steps = int(agent_cfg.updates_epochs) * trainer.n_mb
        it_flops = (w["num_steps"] * trainer.N * 2 * macs) + n_mb_steps * flops_per_launch     # rollout fwd + update
        from cat_envs import native
        out = {
            "metric": "env-steps/s CaT-PPO iteration (rollout + GAE + PPO update)", "value": value,
            "unit": "env-steps/s", "n_gpus": world, "physical_gpus": min(world, n_dev), "ranks_share_gpus": shared,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "strong" if w.get("strong") else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32",
                      "bf16x3": "bf16x3 (split-bf16 GEMM operands = 16 mantissa bits, f32 accumulate/params/activations; "
                                "meets the f32 parity tolerances)",
                      "bf16": "bf16 GEMM operands, f32 accumulate/params (reduced precision: not the headline)"}[prec]
                     + ("" if agent_cfg.rollout_dtype == "fp32" else ", f16 rollout planes"),
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: {w['desc']}", "envs_per_gpu": trainer.N, "envs_total": env_total,
                       "horizon": w["num_steps"], "minibatch_per_gpu": M, "global_minibatch": M * world,
                       "parallelism": f"env-sharded dp{world}, " + ("catppo_allreduce (RCCL)" if
                                      parallel.native_comm_active() or world == 1 else "torch.distributed all_reduce (RCCL)")
                                      + " of the flat gradient",
                       "rccl_world": nat.comm_world if parallel.native_comm_active() else
                                     (world if (world > 1 and not shared) else 0),
                       "collectives": "libcatppo C ABI (librccl)" if parallel.native_comm_active() else
                                      ("gloo, device operands staged through host memory (ranks share a GPU: launcher "
                                       "proof, not a scaling number)" if shared else
                                       "torch.distributed" + (f" (native set-up failed: {parallel.native_comm_error()})"
                                                              if parallel.native_comm_error() else "")
                                       if world > 1 else "none"),
                       "native_comm_error": parallel.native_comm_error(),
                       "comm_env": dict(parallel.comm_env(), launch_attempt=int(os.environ.get("CATPPO_BENCH_LAUNCH_ATTEMPT", "0")),
                                        ipc_mode_set_by=("retry: unset" if os.environ.get("CATPPO_BENCH_IPC_UNSET") == "1" else
                                                         "bench.py default" if os.environ.get("CATPPO_BENCH_IPC_DEFAULTED") == "1"
                                                         else "environment")),
                       "scaling_mode": "strong" if w.get("strong") else "weak",
                       "self_launched": os.environ.get("CATPPO_BENCH_SELF_LAUNCHED") == "1",
                       "grad_overlap": trainer.grad_overlap, "graph_fallback": trainer.graph_fallback,
                       "one_call_optimiser_step": trainer.one_call_step, "readback_in_timed_region": timed_log,
                       "timed_region": "K x run_iteration(log=True): includes the per-iteration diagnostics read-back",
                       "simulated_shard_of_world": a.shard_of if a.shard_of > 0 else None,
                       "rng": trainer.rng, "fused_rollout": trainer.sink is not None,
                       "graph_update": trainer.graph_update, "graph_nodes": trainer.graph_nodes,
                       "overrides": overrides},
            "per_rank_ms": per_rank_ms,
            "ms_per_step_no_readback": 1e3 * dt_nolog / a.steps,
            "comm_ms_per_iteration": None if comm is None else comm["ms"] / comm["iterations"],
            "comm": None if comm is None else {
                "collectives_per_iteration": comm["calls"] / comm["iterations"],
                "bytes_per_iteration": comm["bytes"] / comm["iterations"],
                "ms_by_kind_per_iteration": {k: v[0] / comm["iterations"] for k, v in comm["by_kind"].items()},
                "calls_by_kind_per_iteration": {k: v[1] / comm["iterations"] for k, v in comm["by_kind"].items()},
                "measured": "HIP events around every exchange point of an un-graphed, un-overlapped iteration after the "
                            "timed region (serialised wire time)"},
            "phases_device_ms": phases,
            "iteration_tflops": it_flops / (1e9 * dt / a.steps) / 1e3,
            "roofline": {"bound": "mfma", "kernel": "catppo_ppo_minibatch_grad_packed (row-resident fp32-MFMA forward of the hidden layers "
                         "below the last - or grouped forward GEMM launches - then the last layer with heads + loss + head backward "
                         "in its epilogue from 4096 rows up, else a head+loss launch; paired split-K dW + dX GEMM launches, "
                         "partial fold) per " + str(M) + "-sample minibatch",
                         "achieved": ach * mfma_flops_factor, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach * mfma_flops_factor / peak, "algorithmic_tflops": ach,
                         # what the matrix pipe executes per launch (padded observations, no first-layer data gradient,
                         # 16-output head tiles), x 3 MFMAs per product in the split-bf16 mode
                         "executed_flops": 2 * executed_macs(trainer.Dp, w["hidden"]) * M * mfma_flops_factor,
                         "executed_over_algorithmic_flops": 2 * executed_macs(trainer.Dp, w["hidden"]) * M * mfma_flops_factor
                                                            / flops_per_launch,
                         "traffic": traffic, "avg_launch_us": grad_us, "launches_timed": n_timed,
                         "dominant_kernel_us_profiled": prof_us, "profiled_source": prof_src,
                         # the same FLOPs over the COMMITTED rocprofv3 duration of the group (profiler on, another box)
                         "frac_profiled": None if not prof_us else
                                          flops_per_launch / prof_us / 1e6 * mfma_flops_factor / peak,
                         "timed": "eager replay after the timed region (update phase runs from a hipGraph)"
                                  if trainer.graph_update else "HIP events inside the timed region",
                         "flops_per_launch": flops_per_launch,
                         "traffic_source": traffic_src, "traffic_note": traffic_note, "csrc_hash": csrc_hash()},
            "gae": {"config_size": gae_roofline(nat, w["num_steps"], trainer.N),
                    "config_size_scan": gae_roofline(nat, w["num_steps"], trainer.N, mode=native.GAE_SCAN),
                    "hbm_sweep": [gae_roofline(nat, 24, 1 << 20, 20), gae_roofline(nat, 48, 1 << 22, 10)],
                    "bound": "hbm", "peak_GBps": HBM_PEAK_GBPS, "bytes_per_env_step": 24},
        }
        # (--no-cpu-baseline = the diagnostic form the A/B and profiling scripts use: no secondary record either, so that a
        # kernel trace of that command holds the kernels of ONE precision)
        if prec == "bf16":
            # BASELINE configs[4] (bf16 MFMA operands): the matrix work shrinks 16x while every activation still crosses HBM
            # in fp32 - the group is HBM-bound, and its roofline object says so.  Algorithmic bytes of one
            # minibatch group = a layer-wise step that moves every tensor once each way: the gathered
            # inputs, every hidden activation written and read back, every dZ above the first layer written and read back.
            hid = list(w["hidden"])
            # round 6: hidden activations and dZ are STORED as bf16 from 4096-row minibatches up (csrc/gemm_f32.h "act16")
            el = 2 if (M >= 4096 and os.environ.get("CATPPO_ACT16", "1") != "0") else 4
            act_b = M * sum(hid) * el * 2
            dz_b = M * sum(hid[1:]) * el * 2
            alg_bytes = 2 * act_b + 2 * dz_b + M * (trainer.Dp + trainer.A + 4) * 4
            mf = out["roofline"]
            out["roofline"] = {"bound": "hbm", "kernel": mf["kernel"], "achieved": alg_bytes / grad_us / 1e3, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": alg_bytes / grad_us / 1e3 / HBM_PEAK_GBPS, "algorithmic_bytes": alg_bytes,
                               "activation_element_bytes": el,
                               "traffic": traffic, "traffic_GBps": None if not traffic else traffic / grad_us / 1e3,
                               "hbm_frac": None if not traffic else traffic / grad_us / 1e3 / HBM_PEAK_GBPS,
                               "avg_launch_us": grad_us, "launches_timed": n_timed, "timed": mf["timed"],
                               "traffic_source": traffic_src, "traffic_note": traffic_note, "csrc_hash": mf["csrc_hash"],
                               "mfma": {k: mf[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_tflops", "executed_flops",
                                                          "flops_per_launch", "dominant_kernel_us_profiled", "frac_profiled")}}
        if world == 1 and prec == "fp32" and not a.no_secondary and not a.no_cpu_baseline and a.shard_of == 0:
            out["secondary"] = secondary_record(a, w, dev_index, max(3, a.steps // 2), max(2, a.warmup // 2))
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, trainer, env, agent_cfg)
        if strong is not None:
            out["strong"] = strong
    else:
        out = None
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        parallel.shutdown_native_comm()
        torch.distributed.destroy_process_group()
    if out is not None:
        # RCCL writes a version banner through C stdio: flush it first so that the JSON is the last line
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        assert out["n_gpus"] == a.gpus
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
