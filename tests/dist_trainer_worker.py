"""Two (or more) ranks of the REAL env-sharded ``PPOTrainer`` on one GPU, and the single-process CPU oracle on the
union of their shards ("N ranks = 1 process on the union of shards in exact-parity mode").

One GPU per lease rules out RCCL with more than one rank (RCCL refuses two ranks on one device), so the ranks here
are separate processes that share ``cuda:0`` and exchange through a gloo process group; ``cat_envs.parallel`` stages
device operands through host memory on that backend.  Everything else is the production path: ``PPOTrainer`` with
``parallel.active()``, ragged shards, the minibatch plan, every exchange point of ``dist_exact`` mode.

``rank_main`` (spawned per rank) trains and dumps what the parent needs: its shard's synthetic stream, initial episode
lengths, the actions it took, the permutations it used, its rollout planes / parameters / normalisers / CaT state.
``union_oracle`` (parent) concatenates the shards along the env axis and runs ``oracle.ppo_oracle.PPOOracle`` once.
Test infrastructure only (imports ``oracle/``).
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _paths():
    for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def shard_sizes(n_total: int, world: int):
    base, rem = divmod(n_total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def rank_main(rank: int, world: int, port: int, out_dir: str, spec: dict):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CATPPO_NATIVE_COMM="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _rank_body(rank, world, out_dir, spec)
    finally:
        dist.destroy_process_group()


def _rank_body(rank, world, out_dir, spec):
    import torch
    import smoke_impl
    from cat_envs import parallel
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    assert parallel.active() and parallel.world_size() == world and not parallel.native_comm_active()
    sizes = shard_sizes(spec["n_total"], world)
    off = sum(sizes[:rank])
    N, T, E, iters, seed = sizes[rank], spec["T"], spec["epochs"], spec["iters"], spec["seed"]
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(N, T, spec["minibatch"], E, iters, tuple(spec["hidden"]),
                                                    spec["six_terms"], obs_dim=spec["obs_dim"], seed=seed + rank)
    for k, v in spec.get("overrides", {}).items():
        setattr(agent_cfg, k, v)
    env = make(task, cfg=env_cfg)
    ep0 = env.episode_length_buf.cpu().numpy().copy()      # initial episode lengths (drawn on the device from seed + 17)
    torch.manual_seed(seed + 1000 * rank)        # different initialisations: rank 0's must win (parameter broadcast)
    tr = PPOTrainer(env, agent_cfg)
    assert tr.world == world and tr.rank == rank and tr.n_envs_global == float(spec["n_total"])
    if spec.get("trace"):      # parameters after every optimiser step of the LAST iteration (branch analysis below)
        tr.trace_params = True
    init_sd = {k: v.detach().cpu().numpy().copy() for k, v in tr.agent.state_dict().items()}
    A, B = tr.A, T * N
    rs_eps = np.random.RandomState(seed)                      # the UNION's noise; every rank takes its env columns
    rs_perm = np.random.RandomState(seed + 77 + rank)         # rank-local permutations
    acts, perms_used, stats = [], [], []
    for it in range(iters):
        eps = rs_eps.standard_normal((T, spec["n_total"], A)).astype(np.float32)[:, off:off + N]
        perms = np.stack([rs_perm.permutation(B) for _ in range(E)]).astype(np.int64)
        eps_d, perms_d = torch.from_numpy(np.ascontiguousarray(eps)).cuda(), torch.from_numpy(perms).cuda()
        stats.append(tr.run_iteration(eps_fn=lambda s: eps_d[s], perm_fn=lambda e: perms_d[e]))
        torch.cuda.synchronize()
        acts.append(tr.actions.cpu().numpy().copy())
        perms_used.append(perms)
    cm = env.unwrapped.constraint_manager
    f = lambda t: t.detach().float().cpu().numpy()
    out = dict(
        stream=f(env.sim.stream),
        episode_length0=ep0,
        actions=np.stack(acts), perms=np.stack(perms_used),
        rewards=f(tr.rewards), dones=f(tr.dones), true_dones=f(tr.true_dones), values=f(tr.values),
        logprobs=f(tr.logprobs), advantages=f(tr.advantages), returns=f(tr.returns), flat=f(tr.agent.flat),
        obs_mean=f(tr.agent.obs_rms.running_mean), obs_var=f(tr.agent.obs_rms.running_var),
        obs_count=f(tr.agent.obs_rms.count), val_mean=f(tr.agent.value_rms.running_mean),
        val_var=f(tr.agent.value_rms.running_var), val_count=f(tr.agent.value_rms.count),
        running_maxes=f(cm.cat.get_running_maxes()), M=np.int64(tr.M), n_mb=np.int64(tr.n_mb),
        mb_rows=np.asarray(tr._mb_rows), mb_rows_global=np.asarray(tr._mb_rows_global),
        fused=np.int64(tr.sink is not None), adam_step=np.int64(tr.adam_step),
        loss=np.asarray([s["mean_pg_loss"] for s in stats] + [s["mean_v_loss"] for s in stats]),
        ep_sums=np.stack([f(cm._episode_sums[n]) for n in cm.active_terms]),
        ep_means=np.stack([f(cm._cstr_mean_values[n]) for n in cm.active_terms]))
    if spec.get("trace"):
        # Per optimiser step of the last iteration: this rank's rows of the global minibatch re-evaluated under the parameters
        # BEFORE the step with the loss kernel's clip-branch export on (smoke_impl.device_branches) - what the parent compares
        # with the oracle's branches when the parameter trajectories part; rank 0 also hands over the parameters after every step.
        n_steps, M_r = tr.param_trace.shape[0], int(tr.M)
        codes = np.zeros((n_steps, 2, M_r), np.int8)
        ratio, dl = np.zeros((n_steps, M_r), np.float32), np.zeros((n_steps, M_r), np.float32)
        perms_last = torch.from_numpy(perms_used[-1]).cuda()
        for k in range(n_steps):
            e, j = divmod(k, int(tr.n_mb))
            m = int(tr._mb_rows[j])
            mb = perms_last[e][j * M_r:j * M_r + m]
            theta = tr.param_trace[k - 1] if k > 0 else tr.param_trace_start
            stats_k = tr._adv_stats_all[k] if getattr(tr, "_adv_stats_all", None) is not None and tr.hp.adv_stats_external else None
            c, r_d, d_d = smoke_impl.device_branches(tr, theta, mb, adv_stats=stats_k)
            codes[k, 0, :m], codes[k, 1, :m] = c[:m].numpy(), c[m:].numpy()
            ratio[k, :m], dl[k, :m] = r_d.numpy(), d_d.numpy()
        out["trace/codes"], out["trace/ratio"], out["trace/dl"] = codes, ratio, dl
        if rank == 0:
            out["trace/params"] = np.stack([smoke_impl.logical_params(tr.agent, tr.param_trace[k]).cpu().numpy()
                                            for k in range(n_steps)])
    for k, v in init_sd.items():
        out["init/" + k] = v
    sd = tr.agent.state_dict()
    for k in sd:
        out["final/" + k] = sd[k].detach().cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)


def proto_env(spec):
    """a 4-env instance of the configuration: source of the term table / curriculum / stream layout for the oracle"""
    _paths()
    import smoke_impl
    from cat_envs.shim import make
    task, env_cfg, _ = smoke_impl.make_cfgs(4, spec["T"], spec["minibatch"], spec["epochs"], spec["iters"],
                                            tuple(spec["hidden"]), spec["six_terms"], obs_dim=spec["obs_dim"],
                                            seed=spec["seed"])
    return make(task, cfg=env_cfg)


def union_oracle(spec, world, ranks, proto_env):
    """single-process CPU oracle on the union of the shards; returns (PPOOracle, outputs of its last iteration)"""
    _paths()
    import torch
    from oracle import env_oracle, ppo_oracle
    sizes = shard_sizes(spec["n_total"], world)
    offs = np.cumsum([0] + sizes)
    n_tot, T, E = spec["n_total"], spec["T"], spec["epochs"]
    tmpl = env_oracle.from_device_env(proto_env)                   # terms / curriculum / layout of the configuration
    stream = np.concatenate([r["stream"] for r in ranks], axis=1)
    ep0 = np.concatenate([r["episode_length0"] for r in ranks])
    djp = np.tile(np.asarray(tmpl.default_joint_pos)[:1], (n_tot, 1))
    env = env_oracle.CaTEnvOracle(stream, tmpl.off, tmpl.B, tmpl.H, djp, tmpl.terms, tmpl.curriculum, ep0,
                                  tmpl.max_episode_length, tmpl.step_dt, tau=tmpl.mgr.cat.tau, min_p=tmpl.mgr.cat.min_p)
    hidden = tuple(spec["hidden"])
    over = spec.get("overrides", {})
    ag = ppo_oracle.AgentOracle(spec["obs_dim"], 12, hidden, bf16_hidden=str(over.get("mlp_precision", "fp32")) == "bf16")
    ag.load({k[5:]: torch.from_numpy(v) for k, v in ranks[0].items()
             if k.startswith("init/") and not k.startswith(("init/obs_rms", "init/value_rms"))})
    import smoke_impl
    _, _, agent_cfg = smoke_impl.make_cfgs(n_tot, T, spec["minibatch"] * world, E, spec["iters"], hidden,
                                           spec["six_terms"], obs_dim=spec["obs_dim"], seed=spec["seed"])
    cfg = {k: getattr(agent_cfg, k) for k in ppo_oracle.PPOOracle.DEFAULT_CFG}
    orc = ppo_oracle.PPOOracle(env, n_tot, spec["obs_dim"], 12, cfg=cfg, hidden=hidden, agent=ag,
                               rollout_dtype=str(over.get("rollout_dtype", "fp32")))
    rs_eps = np.random.RandomState(spec["seed"])
    out = None
    orc.trace = bool(spec.get("trace"))          # step_trace of the last iteration (PPOOracle resets it per iteration)
    for it in range(spec["iters"]):
        eps = rs_eps.standard_normal((T, n_tot, 12)).astype(np.float32)
        acts = torch.from_numpy(np.concatenate([r["actions"][it] for r in ranks], axis=1))

        def perm_fn(e, it=it):
            # global minibatch k = union of the ranks' k-th local minibatches, local row t*N_r + i -> t*N + off_r + i
            mbs = []
            n_mb = int(ranks[0]["n_mb"])
            for k in range(n_mb):
                parts = []
                for r, rk in enumerate(ranks):
                    M_r, m = int(rk["M"]), int(rk["mb_rows"][k])
                    loc = rk["perms"][it][e][k * M_r:k * M_r + m]
                    parts.append((loc // sizes[r]) * n_tot + offs[r] + (loc % sizes[r]))
                mbs.append(torch.from_numpy(np.concatenate(parts)))
            return mbs
        out = orc.run_iteration(eps_fn=lambda s: torch.from_numpy(eps[s]), perm_fn=perm_fn,
                                actions_fn=lambda s: acts[s])
    return orc, out


def flat_params_of(sd_arrays, prefix="final/"):
    keys = ["actor_logstd"] + [f"{n}.{i}.{w}" for n in ("critic", "actor_mean") for i in (0, 2, 4, 6)
                               for w in ("weight", "bias")]
    return np.concatenate([np.asarray(sd_arrays[prefix + k]).reshape(-1) for k in keys])
