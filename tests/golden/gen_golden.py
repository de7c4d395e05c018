"""Golden-vector generator: runs the REFERENCE's own code (imported in place from
/root/reference, this container only) on seeded inputs from ``tests/streams.py`` and
freezes inputs-checksums + outputs as small ``.npz`` fixtures next to this file.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Fixtures are data only (arrays); no reference source travels.  What executes:
  cat_*      reference ``ConstraintManager`` + ``CaT`` + ``modify_constraint_p``
  terms      reference ``cat/constraints.py`` term functions on a fake scene
  rms        reference ``RunningMeanStd``
  agent      reference ``Agent`` (weights injected from tests/streams.agent_weights)
  ppo_*      reference ``PPO()`` end to end on a recording open-loop env; locals of the
             running ``PPO`` frame (GAE outputs, value_rms outputs, first-minibatch
             losses/gradients) are captured with ``sys.settrace`` at fixed line numbers
             of the reference file, noise / permutations are injected by patching
             ``Normal.sample`` and ``torch.randperm``.
  envfinish  the six arithmetic lines of ``CaTEnv.step`` (cat_env.py:102-107,118-121)
             evaluated with torch (CaTEnv itself needs Isaac Sim and cannot be built).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _ref_import as R  # noqa: E402
import streams as S  # noqa: E402

torch.set_num_threads(1)      # deterministic reductions for the frozen vectors
torch.use_deterministic_algorithms(True)


#: where save() writes: next to this file, or a scratch directory under ``--check``
OUT_DIR = HERE


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def t2n(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


# ------------------------------------------------------------------------------ CaT streams
class _FakeCatEnv:
    def __init__(self, n):
        self.num_envs = n
        self.device = "cpu"
        self.episode_length_buf = torch.zeros(n, dtype=torch.long)
        self.common_step_counter = 0
        self.cursor = 0
        self.stream = None


def gen_cat(tag, seed, n_envs, terms, max_ps, steps, sub=1, tau=0.95, min_p=0.0,
            curriculum_at=None, reset_at=None):
    cm, cfgmod, _, cu = R.load_ref_cat()
    env = _FakeCatEnv(n_envs)
    env.stream = S.cat_stream(seed, n_envs, terms, steps)

    def make_term(name):
        return lambda e: torch.from_numpy(np.asarray(e.stream[e.cursor][name]))

    cfg = {name: cfgmod.ConstraintTermCfg(func=make_term(name), params={}, max_p=mp)
           for (name, _, _), mp in zip(terms, max_ps)}
    mgr = cm.ConstraintManager(cfg, env, tau=tau, min_p=min_p)
    env.constraint_manager = mgr
    names = [t[0] for t in terms]
    rec = {"cstr_prob": [], "running_maxes": [], "term_max": [], "max_p": []}
    extras_rec = {}
    rs = np.random.RandomState(seed + 1000)
    for t in range(steps):
        env.cursor = t
        env.episode_length_buf += 1
        env.common_step_counter += 1
        if curriculum_at is not None and t == curriculum_at:
            # the reference's own curriculum term rewrites max_p through get/set_term_cfg
            for name in names[:2]:
                cu.modify_constraint_p(env, None, name, num_steps=4 * steps, init_max_p=0.25)
        p = mgr.compute()
        rec["cstr_prob"].append(t2n(p)[::sub].copy())
        rec["running_maxes"].append(t2n(mgr.cat.get_running_maxes())[0].copy())
        rec["term_max"].append(np.stack([t2n(mgr.cat.probs[n].max(1).values)[::sub] for n in names]))
        rec["max_p"].append(np.array([mgr.get_term_cfg(n).max_p for n in names], np.float64))
        if reset_at is not None and t in reset_at:
            ids = np.nonzero(rs.rand(n_envs) < 0.3)[0]
            ex = mgr.reset(torch.from_numpy(ids))
            env.episode_length_buf[torch.from_numpy(ids)] = 0
            extras_rec[f"reset{t}_ids"] = ids
            extras_rec[f"reset{t}_vals"] = np.array([float(ex[k]) for k in sorted(ex)], np.float64)
            extras_rec[f"reset{t}_keys"] = np.array(sorted(ex))
    save(f"cat_{tag}", seed=seed, n_envs=n_envs, steps=steps, sub=sub, tau=tau, min_p=min_p,
         term_names=np.array(names), term_widths=np.array([t[1] for t in terms]),
         term_kinds=np.array([t[2] for t in terms]), init_max_p=np.array(max_ps, np.float64),
         curriculum_at=-1 if curriculum_at is None else curriculum_at,
         reset_at=np.array(sorted(reset_at) if reset_at else [], np.int64),
         input_checksum=S.cat_stream_checksum(env.stream),
         cstr_prob=np.stack(rec["cstr_prob"]), running_maxes=np.stack(rec["running_maxes"]),
         term_max=np.stack(rec["term_max"]), max_p=np.stack(rec["max_p"]),
         episode_sums=np.stack([t2n(mgr._episode_sums[n])[::sub] for n in names]),
         cstr_mean_values=np.stack([t2n(mgr._cstr_mean_values[n])[::sub] for n in names]),
         **extras_rec)


def gen_curriculum():
    _, _, _, cu = R.load_ref_cat()
    env = types.SimpleNamespace(common_step_counter=0)
    cfg = types.SimpleNamespace(max_p=0.25)
    env.constraint_manager = types.SimpleNamespace(get_term_cfg=lambda n: cfg, set_term_cfg=lambda n, c: None)
    steps = np.array([0, 1, 23, 24, 1000, 11999, 12000, 23999, 24000, 24001, 10 ** 6], np.int64)
    table = []
    for init in (0.25, 0.1, 1.0):
        for s in steps:
            env.common_step_counter = int(s)
            table.append(cu.modify_constraint_p(env, None, "x", num_steps=24000, init_max_p=init))
    save("curriculum", steps=steps, inits=np.array([0.25, 0.1, 1.0]), num_steps=24000,
         max_p=np.array(table, np.float64).reshape(3, -1))


# ------------------------------------------------------------------------------ term functions
def gen_terms():
    _, _, cs, _ = R.load_ref_cat()
    from isaaclab.managers import SceneEntityCfg
    n = 257
    st = S.sim_state(77, n)
    tt = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    robot = types.SimpleNamespace(data=types.SimpleNamespace(
        joint_pos=tt["joint_pos"], default_joint_pos=tt["default_joint_pos"], joint_vel=tt["joint_vel"],
        joint_acc=tt["joint_acc"], applied_torque=tt["applied_torque"],
        projected_gravity_b=tt["projected_gravity_b"], root_pos_w=tt["root_pos_w"]))
    sensor = types.SimpleNamespace(
        data=types.SimpleNamespace(net_forces_w_history=tt["net_forces_w_history"],
                                   last_air_time=tt["last_air_time"]),
        compute_first_contact=lambda dt: tt["first_contact"])
    env = types.SimpleNamespace(
        scene={"robot": robot, "contact_forces": sensor},
        command_manager=types.SimpleNamespace(get_command=lambda name: tt["command"]),
        action_manager=types.SimpleNamespace(_action=tt["action"], _prev_action=tt["prev_action"]),
        step_dt=st["step_dt"])
    alljoints = SceneEntityCfg("robot", joint_ids=slice(None))
    hfe = SceneEntityCfg("robot", joint_ids=[1, 4])
    haa = SceneEntityCfg("robot", joint_ids=[0, 3, 6, 9])
    feet = SceneEntityCfg("contact_forces", body_ids=[3, 6, 9, 12])
    upper = SceneEntityCfg("contact_forces", body_ids=[0, 2, 5, 8, 11])
    out = {
        "joint_position": cs.joint_position(env, 1.3, hfe),
        "joint_position_when_moving_forward": cs.joint_position_when_moving_forward(env, 0.2, 0.1, haa),
        "joint_torque": cs.joint_torque(env, 3.0, alljoints),
        "joint_velocity": cs.joint_velocity(env, 16.0, alljoints),
        "joint_acceleration": cs.joint_acceleration(env, 800.0, alljoints),
        "upsidedown": cs.upsidedown(env, 0.0, SceneEntityCfg("robot")),
        "contact": cs.contact(env, upper),
        "base_orientation": cs.base_orientation(env, 0.1, SceneEntityCfg("robot")),
        "air_time": cs.air_time(env, 0.25, 0.1, feet),
        "n_foot_contact": cs.n_foot_contact(env, 2, 0.5, feet),
        "joint_range": cs.joint_range(env, 0.4, alljoints),
        "action_rate": cs.action_rate(env, 80.0, alljoints),
        "foot_contact_force": cs.foot_contact_force(env, 50.0, feet),
        "min_base_height": cs.min_base_height(env, 0.2, SceneEntityCfg("robot")),
        "no_move": cs.no_move(env, 0.1, 4.0, alljoints),
    }
    save("terms", seed=77, n_envs=n, input_checksum=S.checksum(*[np.asarray(v) for v in st.values()]),
         **{k: t2n(v) for k, v in out.items()})


def gen_terms_scale():
    """the norm-based terms (C8, C13) and everything a norm gates (C7 contact, C9 / C10 / C15) at
    4096 envs x 4 steps with the norms planted AT their limits (streams.sim_state_at_the_limits).  Stored: the reference's
    outputs of step 0 in full, and for every step the packed violation masks `c > 0` and a sha256 of the raw fp32 bytes
    (a bit-exact consumer checks the hash; one that is not gets the step-0 values to count ulps against).  Also records,
    on the way, that torch.norm over a short last dimension IS the fp32 FMA chain the oracle / kernels restate."""
    import hashlib
    _, _, cs, _ = R.load_ref_cat()
    from isaaclab.managers import SceneEntityCfg
    n, steps = 4096, 4
    states = S.sim_state_at_the_limits(91, n, steps)
    feet = SceneEntityCfg("contact_forces", body_ids=[3, 6, 9, 12])
    upper = SceneEntityCfg("contact_forces", body_ids=[0, 2, 5, 8, 11])
    alljoints = SceneEntityCfg("robot", joint_ids=slice(None))
    names = ("base_orientation", "foot_contact_force", "contact", "air_time", "n_foot_contact", "no_move")
    rec = {k: [] for k in names}
    chain_mismatch = 0
    for st in states:
        tt = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in st.items()}
        robot = types.SimpleNamespace(data=types.SimpleNamespace(
            joint_pos=tt["joint_pos"], default_joint_pos=tt["default_joint_pos"], joint_vel=tt["joint_vel"],
            joint_acc=tt["joint_acc"], applied_torque=tt["applied_torque"],
            projected_gravity_b=tt["projected_gravity_b"], root_pos_w=tt["root_pos_w"]))
        sensor = types.SimpleNamespace(
            data=types.SimpleNamespace(net_forces_w_history=tt["net_forces_w_history"],
                                       last_air_time=tt["last_air_time"]),
            compute_first_contact=lambda dt, tt=tt: tt["first_contact"])
        env = types.SimpleNamespace(
            scene={"robot": robot, "contact_forces": sensor},
            command_manager=types.SimpleNamespace(get_command=lambda name, tt=tt: tt["command"]),
            action_manager=types.SimpleNamespace(_action=tt["action"], _prev_action=tt["prev_action"]),
            step_dt=st["step_dt"])
        out = {
            "base_orientation": cs.base_orientation(env, 0.1, SceneEntityCfg("robot")),
            "foot_contact_force": cs.foot_contact_force(env, 50.0, feet),
            "contact": cs.contact(env, upper),
            "air_time": cs.air_time(env, 0.25, 0.1, feet),
            "n_foot_contact": cs.n_foot_contact(env, 2, 0.5, feet),
            "no_move": cs.no_move(env, 0.1, 4.0, alljoints),
        }
        for k in names:
            rec[k].append(np.ascontiguousarray(t2n(out[k]).astype(np.float32).reshape(n, -1)))
        # torch.norm == fp32 FMA chain (what oracle/cat_oracle.py:_norm_last and csrc/terms_eval.h:norm3 restate)
        f = st["net_forces_w_history"]
        w = np.longdouble
        acc = (f[..., 0] * f[..., 0]).astype(np.float32)
        for i in (1, 2):
            acc = (f[..., i].astype(w) * f[..., i].astype(w) + acc.astype(w)).astype(np.float32)
        chain_mismatch += int((np.sqrt(acc) != torch.norm(tt["net_forces_w_history"], dim=-1).numpy()).sum())
    assert chain_mismatch == 0, chain_mismatch
    payload = {"seed": 91, "n_envs": n, "steps": steps, "norm_is_fma_chain_mismatches": chain_mismatch,
               "input_checksum": S.checksum(*[np.asarray(v) for st in states for v in st.values()])}
    for k in names:
        payload[k + "_step0"] = rec[k][0]
        payload[k + "_mask_bits"] = np.stack([np.packbits(a > 0) for a in rec[k]])
        payload[k + "_sha256"] = np.array([hashlib.sha256(a.tobytes()).hexdigest() for a in rec[k]])
        payload[k + "_at_zero"] = np.array([int((a == 0).sum()) for a in rec[k]])
    save("terms_scale", **payload)


# ------------------------------------------------------------------------------ env finish
def gen_envfinish():
    rs = np.random.RandomState(5)
    n = 513
    reward = torch.from_numpy(rs.uniform(-0.2, 1.5, n).astype(np.float32))
    p = torch.from_numpy(S.soft_dones(rs, (n,)))
    reset = torch.from_numpy(rs.rand(n) < 0.1)
    r = torch.clip(reward * (1.0 - p), min=0.0, max=None)     # cat_env.py:102-106
    dones = p.clone()                                          # :107
    ids = reset.nonzero(as_tuple=False).squeeze(-1)            # :118
    dones[ids] = 1.0                                           # :121
    save("envfinish", reward_in=t2n(reward), cstr_prob=t2n(p), reset=t2n(reset), reward=t2n(r), dones=t2n(dones))


# ------------------------------------------------------------------------------ RunningMeanStd
def gen_rms():
    ppo = R.load_ref_ppo()
    rs = np.random.RandomState(11)
    vec = ppo.RunningMeanStd(shape=(45,))
    sca = ppo.RunningMeanStd(shape=())
    xs = (rs.standard_normal((30, 64, 45)) * rs.uniform(0.1, 5, 45) + rs.uniform(-2, 2, 45)).astype(np.float32)
    ys = (rs.standard_normal((30, 1536)) * 3 + 1).astype(np.float32)
    vo, so, vstate, sstate = [], [], [], []
    for i in range(30):
        vo.append(t2n(vec(torch.from_numpy(xs[i]))))
        so.append(t2n(sca(torch.from_numpy(ys[i]))))
        vstate.append(np.concatenate([t2n(vec.running_mean), t2n(vec.running_var), [float(vec.count)]]))
        sstate.append(np.array([float(sca.running_mean), float(sca.running_var), float(sca.count)]))
    frozen = t2n(vec(torch.from_numpy(xs[0]), update=False))
    save("rms", seed=11, input_checksum=S.checksum(xs, ys), vec_out_last=vo[-1], vec_out_first=vo[0],
         sca_out_last=so[-1][:64], vec_state=np.stack(vstate).astype(np.float32),
         sca_state=np.stack(sstate).astype(np.float32), frozen_out=frozen)


# ------------------------------------------------------------------------------ Agent
class _SpaceEnv:
    def __init__(self, n, d, a):
        sp = types.SimpleNamespace
        self.unwrapped = sp(num_envs=n, single_observation_space={"policy": sp(shape=(d,))},
                            single_action_space=sp(shape=(a,)))


def _ref_agent(ppo, d, a, weights):
    ag = ppo.Agent(_SpaceEnv(1, d, a))
    sd = ag.state_dict()
    for k, v in weights.items():
        sd[k] = torch.from_numpy(v)
    ag.load_state_dict(sd)
    return ag


def gen_agent():
    ppo = R.load_ref_ppo()
    d, a, b = 45, 12, 96
    w = S.agent_weights(3, d, a)
    ag = _ref_agent(ppo, d, a, w)
    keys = list(ag.state_dict().keys())
    nparam = sum(p.numel() for p in ag.parameters())
    rs = np.random.RandomState(4)
    x = rs.standard_normal((b, d)).astype(np.float32)
    eps = rs.standard_normal((b, a)).astype(np.float32)
    from torch.distributions.normal import Normal
    orig = Normal.sample
    Normal.sample = lambda self, sample_shape=torch.Size(): (self.loc + self.scale * torch.from_numpy(eps)).detach()
    try:
        with torch.no_grad():
            act, logp, ent, val = ag.get_action_and_value(torch.from_numpy(x))
            act2, logp2, ent2, val2 = ag.get_action_and_value(torch.from_numpy(x), torch.from_numpy(eps * 0.3))
            ag.obs_rms(torch.from_numpy(x * 2 + 1))  # move the normaliser off identity
            det = ag(torch.from_numpy(x))
    finally:
        Normal.sample = orig
    save("agent", weight_seed=3, input_seed=4, obs_dim=d, act_dim=a, state_keys=np.array(keys), n_params=nparam,
         weight_checksum=S.checksum(*[w[k] for k in sorted(w)]), input_checksum=S.checksum(x, eps),
         action=t2n(act), logprob=t2n(logp), entropy=t2n(ent), value=t2n(val),
         logprob_given=t2n(logp2), value_given=t2n(val2), deterministic=t2n(det),
         param_order=np.array([n for n, _ in ag.named_parameters()]))


# ------------------------------------------------------------------------------ full PPO() runs
class RecordingEnv:
    """Open-loop env obeying the protocol PPO() consumes (ppo.py:158-161,186,215-230)."""

    def __init__(self, stream, n, d, a):
        self.s, self.t = stream, 0
        sp = types.SimpleNamespace
        self.unwrapped = sp(num_envs=n, single_observation_space={"policy": sp(shape=(d,))},
                            single_action_space=sp(shape=(a,)))

    def reset(self):
        return {"policy": torch.from_numpy(self.s["obs0"])}, {}

    def step(self, action):
        t = self.t
        self.t += 1
        tn = torch.from_numpy
        return ({"policy": tn(self.s["obs"][t])}, tn(self.s["reward"][t]), tn(self.s["dones"][t]),
                tn(self.s["timeouts"][t]), {"log": {"Episode_Reward/track": torch.tensor(float(t))}})


# reference line numbers inside PPO() (cleanrl/ppo.py)
L_AFTER_GAE, L_AFTER_VRMS, L_BEFORE_CLIP, L_AFTER_UPDATE = 280, 290, 353, 356


def run_ref_ppo(tag, seed, N, T, iters, mb, epochs, D=45, A=12, sub=1, keep_grads=True):
    ppo = R.load_ref_ppo()
    stream = S.env_stream(seed, T * iters, N, D)
    weights = S.agent_weights(seed + 1, D, A)
    rs = np.random.RandomState(seed + 2)
    eps_all = rs.standard_normal((iters * T, N, A)).astype(np.float32)
    perms = np.stack([rs.permutation(N * T) for _ in range(iters * epochs)]).astype(np.int64)
    counters = {"eps": 0, "perm": 0}
    cap = {"iters": []}
    cur = {}
    state = {"agent": None}

    from torch.distributions.normal import Normal
    orig_sample, orig_randperm, orig_init = Normal.sample, torch.randperm, ppo.Agent.__init__

    def sample(self, sample_shape=torch.Size()):
        e = torch.from_numpy(eps_all[counters["eps"]])
        counters["eps"] += 1
        return (self.loc + self.scale * e).detach()

    def randperm(n, **kw):
        p = torch.from_numpy(perms[counters["perm"]])
        counters["perm"] += 1
        return p

    def agent_init(self, envs):
        orig_init(self, envs)
        sd = self.state_dict()
        for k, v in weights.items():
            sd[k] = torch.from_numpy(v)
        self.load_state_dict(sd)
        state["agent"] = self

    def tracer(frame, event, arg):
        if frame.f_code.co_name != "PPO":
            return None

        def local(frame, event, arg):
            if event != "line":
                return local
            ln, loc = frame.f_lineno, frame.f_locals
            if ln == L_AFTER_GAE:
                cur.clear()
                cur["advantages"], cur["returns"] = t2n(loc["advantages"]).copy(), t2n(loc["returns"]).copy()
                cur["values"], cur["rewards"] = t2n(loc["values"]).copy(), t2n(loc["rewards"]).copy()
                cur["dones"], cur["true_dones"] = t2n(loc["dones"]).copy(), t2n(loc["true_dones"]).copy()
                cur["next_value"] = t2n(loc["next_value"]).reshape(-1).copy()
                cur["next_done"] = t2n(loc["next_done"]).copy()
                cur["next_true_done"] = t2n(loc["next_true_done"]).copy()
                cur["logprobs"], cur["actions"] = t2n(loc["logprobs"]).copy(), t2n(loc["actions"]).copy()
                cur["obs_last"] = t2n(loc["obs"][-1]).copy()
                ag = loc["agent"]
                cur["obs_rms"] = np.concatenate([t2n(ag.obs_rms.running_mean), t2n(ag.obs_rms.running_var),
                                                 [float(ag.obs_rms.count)]])
            elif ln == L_AFTER_VRMS:
                ag = loc["agent"]
                cur["b_values"], cur["b_returns"] = t2n(loc["b_values"]).copy(), t2n(loc["b_returns"]).copy()
                cur["value_rms"] = np.array([float(ag.value_rms.running_mean), float(ag.value_rms.running_var),
                                             float(ag.value_rms.count)])
            elif ln == L_BEFORE_CLIP and "mb0" not in cur:
                ag = loc["agent"]
                g = torch.cat([p.grad.reshape(-1) for p in ag.parameters()])
                cur["mb0"] = np.array([float(loc["loss"]), float(loc["pg_loss"]), float(loc["v_loss"]),
                                       float(loc["entropy_loss"]), float(loc["approx_kl"]),
                                       float(loc["old_approx_kl"]), float(loc["clipfracs"][-1]),
                                       float(g.norm())])
                if keep_grads:
                    cur["mb0_grad"] = t2n(g).copy()
                cur["mb0_inds"] = t2n(loc["mb_inds"]).copy()
            elif ln == L_AFTER_UPDATE:
                cur["sums"] = np.array([float(loc["sum_pg_loss"]), float(loc["sum_entropy_loss"]),
                                        float(loc["sum_v_loss"]), float(loc["sum_surrogate_loss"])])
                cur["lr"] = loc["optimizer"].param_groups[0]["lr"]
                ag = loc["agent"]
                cur["params_after"] = t2n(torch.cat([p.detach().reshape(-1) for p in ag.parameters()])).copy()
                cap["iters"].append(dict(cur))
            return local
        return local

    cfg = types.SimpleNamespace(
        logger="tensorboard", learning_rate=3.0e-4, num_steps=T, num_iterations=iters, gamma=0.99,
        gae_lambda=0.95, updates_epochs=epochs, minibatch_size=mb, clip_coef=0.2, ent_coef=0.001,
        vf_coef=2.0, max_grad_norm=1.0, norm_adv=True, clip_vloss=True, anneal_lr=True, save_interval=10 ** 9)
    env = RecordingEnv(stream, N, D, A)
    Normal.sample, torch.randperm, ppo.Agent.__init__ = sample, randperm, agent_init
    run_path = f"/tmp/golden_run_{tag}"
    sys.settrace(tracer)
    try:
        ppo.PPO(env, cfg, run_path)
    finally:
        sys.settrace(None)
        Normal.sample, torch.randperm, ppo.Agent.__init__ = orig_sample, orig_randperm, orig_init
    writer = R.tensorboard_stub().last
    scal = {}
    for k, v, it in writer.scalars:
        scal.setdefault(k, []).append(v)
    out = dict(seed=seed, N=N, T=T, iters=iters, minibatch=mb, epochs=epochs, D=D, A=A, sub=sub,
               input_checksum=S.checksum(stream["obs"], stream["reward"], stream["dones"], eps_all, perms),
               scalar_keys=np.array(sorted(scal)),
               scalars=np.array([scal[k] for k in sorted(scal)], np.float64))
    psub = 97  # parameter subsample stride
    for i, c in enumerate(cap["iters"]):
        for k in ("advantages", "returns", "values", "logprobs"):
            out[f"it{i}_{k}"] = c[k][:, ::sub]
        out[f"it{i}_actions"] = c["actions"][:, ::sub][:, :, :3]
        out[f"it{i}_next_value"] = c["next_value"][::sub]
        out[f"it{i}_b_values"] = c["b_values"].reshape(T, N)[:, ::sub]
        out[f"it{i}_b_returns"] = c["b_returns"].reshape(T, N)[:, ::sub]
        out[f"it{i}_obs_last"] = c["obs_last"][::sub]
        for k in ("obs_rms", "value_rms", "mb0", "sums"):
            out[f"it{i}_{k}"] = c[k]
        out[f"it{i}_lr"] = c["lr"]
        out[f"it{i}_params_sub"] = c["params_after"][::psub]
        out[f"it{i}_params_norm"] = float(np.linalg.norm(c["params_after"].astype(np.float64)))
        if keep_grads and i == 0:
            out["it0_mb0_grad_sub"] = c["mb0_grad"][::psub]
            out["it0_mb0_grad_head"] = c["mb0_grad"][:12]
    out["param_sub_stride"] = psub
    save(f"ppo_{tag}", **out)


def gen_skrl_gae():
    """Executes the reference's own nested ``compute_gae`` (skrl/ppo.py:397-442): the function node is lifted
    out of ``PPO._update`` with ``ast`` at generation time (skrl itself is not installed, so the module cannot
    be imported), its free variable ``last_values`` is supplied as a global.  Nothing but arrays is stored."""
    import ast
    src = open(os.path.join(R.REF_ROOT, "exts/cat_envs/cat_envs/tasks/utils/skrl/ppo.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "compute_gae")
    mod = ast.Module(body=[fn], type_ignores=[])
    out = {}
    for tag, (T, N) in {"a": (24, 64), "b": (1, 5), "c": (48, 33)}.items():
        rs = np.random.RandomState(700 + T)
        rew = rs.uniform(0, 1.5, (T, N, 1)).astype(np.float32)
        val = rs.standard_normal((T, N, 1)).astype(np.float32)
        done = np.where(rs.uniform(size=(T, N, 1)) < 0.3, rs.uniform(0, 1, (T, N, 1)), 0.0).astype(np.float32)
        done[rs.uniform(size=(T, N, 1)) < 0.02] = 1.0
        last = rs.standard_normal((N, 1)).astype(np.float32)
        ns = {"torch": torch, "last_values": torch.from_numpy(last)}
        exec(compile(mod, "<reference skrl/ppo.py compute_gae>", "exec"), ns)
        ret, adv = ns["compute_gae"](torch.from_numpy(rew), torch.from_numpy(done), torch.from_numpy(val),
                                     torch.from_numpy(last), discount_factor=0.99, lambda_coefficient=0.95)
        out.update({f"{tag}_rewards": rew, f"{tag}_values": val, f"{tag}_dones": done, f"{tag}_last_values": last,
                    f"{tag}_returns": t2n(ret), f"{tag}_advantages": t2n(adv)})
    save("skrl_gae", **out)


def gen_rlg_play_steps():
    """Executes the reference's own ``CaTA2CAgent.play_steps`` (rl_games/cat_common.py:35-112) - the method node is
    lifted out of the class with ``ast`` (rl_games is not installed, so the module cannot be imported) and called on a
    stub agent object that supplies what the method touches: policy outputs / env answers from
    ``streams.rlg_play_steps_inputs``, a dict-of-planes experience buffer with the float ``dones`` plane of
    cat_experience.py:27-33, recording meters / observer, and rl_games' published ``discount_values`` /
    ``swap_and_flatten01`` / ``AverageMeter.update`` restated (third-party, not in the reference tree).  What the
    golden pins is the reference's float-dones bookkeeping: which dones row lands in which buffer slot, the
    value_bootstrap shaping, ``dones.ge(1.0)`` as the episode end, ``current_* *= 1 - dones``, the length reset.
    Nothing but arrays is stored."""
    import ast
    import time as _time
    src = open(os.path.join(R.REF_ROOT, "exts/cat_envs/cat_envs/tasks/utils/rl_games/cat_common.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "play_steps")
    mod = ast.Module(body=[fn], type_ignores=[])

    def swap_and_flatten01(arr):             # rl_games.common.a2c_common (published)
        if arr is None:
            return arr
        s = arr.size()
        return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])

    ns = {"torch": torch, "time": _time, "swap_and_flatten01": swap_and_flatten01}
    exec(compile(mod, "<reference rl_games/cat_common.py play_steps>", "exec"), ns)
    play_steps = ns["play_steps"]

    N, T, D, A, H = 48, 16, 5, 3, 2
    x = S.rlg_play_steps_inputs(901, N, T, D, A, H)

    class Meter:                             # rl_games torch_ext.AverageMeter (published rule), recording its inputs
        def __init__(self, shape, max_size):
            self.max_size, self.current_size, self.mean, self.log = max_size, 0, torch.zeros(shape), []

        def update(self, values):
            self.log.append(values.clone())
            size = values.size()[0]
            if size == 0:
                return
            new_mean = torch.mean(values.float(), dim=0)
            size = int(np.clip(size, 0, self.max_size))
            old_size = min(self.max_size - size, self.current_size)
            size_sum = old_size + size
            self.current_size = size_sum
            self.mean = (self.mean * old_size + new_mean * size) / size_sum

    class Buffer:                            # the part of rl_games' ExperienceBuffer play_steps uses
        def __init__(self):
            z = lambda *s: torch.zeros(T, N, *s)
            self.tensor_dict = {"obses": z(D), "rewards": z(1), "values": z(1), "neglogpacs": z(), "dones": z(),
                                "actions": z(A), "mus": z(A), "sigmas": z(A)}

        def update_data(self, name, index, val):
            self.tensor_dict[name][index, :] = val

        def get_transformed_list(self, op, names):
            return {k: op(self.tensor_dict[k]) for k in names if self.tensor_dict.get(k) is not None}

    class Observer:
        def __init__(self):
            self.log = []

        def process_infos(self, infos, done_indices):
            self.log.append(done_indices.clone())

    class Agent:
        pass
    ag = Agent()
    ag.horizon_length, ag.use_action_masks, ag.has_central_value, ag.num_agents = T, False, False, 1
    ag.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
    ag.tensor_list = ag.update_list + ["obses", "states", "dones"]
    ag.experience_buffer = Buffer()
    ag.value_bootstrap, ag.gamma, ag.tau, ag.batch_size = True, 0.99, 0.95, N * T
    ag.rewards_shaper = lambda r: r * 0.5                      # DefaultRewardsShaper(scale_value=0.5)
    ag.cast_obs = lambda o: o
    ag.obs = {"obs": torch.from_numpy(x["obs0"])}
    ag.dones = torch.ones(N)
    ag.current_rewards, ag.current_shaped_rewards = torch.zeros(N, 1), torch.zeros(N, 1)
    ag.current_lengths = torch.zeros(N)
    ag.game_rewards, ag.game_shaped_rewards, ag.game_lengths = Meter((1,), 100), Meter((1,), 100), Meter((), 100)
    ag.algo_observer = Observer()
    clock = {"t": 0, "h": 0}

    def get_action_values(obs):
        t = clock["t"]
        return {k: torch.from_numpy(x[k][t]) for k in ("actions", "values", "neglogpacs", "mus", "sigmas")}

    def env_step(actions):
        t = clock["t"]
        clock["t"] += 1
        return ({"obs": torch.from_numpy(x["next_obs"][t])}, torch.from_numpy(x["rewards"][t]),
                torch.from_numpy(x["dones"][t]), {"time_outs": torch.from_numpy(x["time_outs"][t])})

    def discount_values(fdones, last_extrinsic_values, mb_fdones, mb_extrinsic_values, mb_rewards):
        # rl_games A2CBase.discount_values (published; not in the reference tree)
        lastgaelam = 0
        mb_advs = torch.zeros_like(mb_rewards)
        for t in reversed(range(ag.horizon_length)):
            if t == ag.horizon_length - 1:
                nextnonterminal = 1.0 - fdones
                nextvalues = last_extrinsic_values
            else:
                nextnonterminal = 1.0 - mb_fdones[t + 1]
                nextvalues = mb_extrinsic_values[t + 1]
            nextnonterminal = nextnonterminal.unsqueeze(1)
            delta = mb_rewards[t] + ag.gamma * nextvalues * nextnonterminal - mb_extrinsic_values[t]
            mb_advs[t] = lastgaelam = delta + ag.gamma * ag.tau * nextnonterminal * lastgaelam
        return mb_advs
    ag.get_action_values, ag.env_step, ag.discount_values = get_action_values, env_step, discount_values
    ag.get_values = lambda obs: torch.from_numpy(x["last_values"][clock["h"]])
    out = {"N": N, "T": T, "D": D, "A": A, "H": H, "seed": 901, "inputs_checksum": S.checksum(*[x[k] for k in sorted(x)])}
    for h in range(H):
        clock["h"] = h
        batch = play_steps(ag)
        for k in ("returns", "obses", "dones", "values", "actions", "neglogpacs", "mus", "sigmas"):
            out[f"h{h}_batch_{k}"] = t2n(batch[k]).copy()
        out[f"h{h}_played_frames"] = batch["played_frames"]
        out[f"h{h}_buf_rewards"] = t2n(ag.experience_buffer.tensor_dict["rewards"]).copy()
        out[f"h{h}_buf_dones"] = t2n(ag.experience_buffer.tensor_dict["dones"]).copy()
        out[f"h{h}_current_rewards"] = t2n(ag.current_rewards).copy()
        out[f"h{h}_current_shaped_rewards"] = t2n(ag.current_shaped_rewards).copy()
        out[f"h{h}_current_lengths"] = t2n(ag.current_lengths).copy()
        out[f"h{h}_final_dones"] = t2n(ag.dones).copy()
        out[f"h{h}_game_rewards_mean"] = t2n(ag.game_rewards.mean).copy()
        out[f"h{h}_game_shaped_rewards_mean"] = t2n(ag.game_shaped_rewards.mean).copy()
        out[f"h{h}_game_lengths_mean"] = t2n(ag.game_lengths.mean).copy()
        out[f"h{h}_game_size"] = ag.game_rewards.current_size
    done_mask = np.zeros((H * T, N), np.bool_)
    for t, idx in enumerate(ag.algo_observer.log):
        done_mask[t, t2n(idx).reshape(-1)] = True
    out["done_mask"] = done_mask
    out["meter_rewards_in_sum"] = np.array([float(v.sum()) for v in ag.game_rewards.log], np.float64)
    out["meter_lengths_in_sum"] = np.array([float(v.sum()) for v in ag.game_lengths.log], np.float64)
    save("rlg_play_steps", **out)


def check_against_committed(scratch):
    """every fixture regenerated into ``scratch`` equals the committed one array for array (same keys, dtypes, shapes,
    bits): the committed vectors ARE what the reference computes here.  Returns the list of mismatches."""
    bad = []
    committed = sorted(f for f in os.listdir(HERE) if f.endswith(".npz"))
    fresh = sorted(f for f in os.listdir(scratch) if f.endswith(".npz"))
    if committed != fresh:
        bad.append(f"fixture sets differ: committed {committed} vs regenerated {fresh}")
    for f in sorted(set(committed) & set(fresh)):
        a, b = np.load(os.path.join(HERE, f), allow_pickle=False), np.load(os.path.join(scratch, f), allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            bad.append(f"{f}: keys differ: {sorted(set(a.files) ^ set(b.files))}")
            continue
        for k in a.files:
            x, y = a[k], b[k]
            same = x.dtype == y.dtype and x.shape == y.shape and \
                (np.array_equal(x, y, equal_nan=True) if x.dtype.kind in "fc" else np.array_equal(x, y))
            if not same:
                bad.append(f"{f}[{k}]: {x.dtype}{x.shape} vs {y.dtype}{y.shape}")
    return bad


def main():
    assert R.have_reference(), "needs /root/reference (build container only)"
    if sys.argv[1:2] == ["--check"]:      # regenerate everything into a scratch directory and compare with the committed files
        import tempfile
        global OUT_DIR
        with tempfile.TemporaryDirectory() as tmp:
            OUT_DIR = sys.argv[2] if len(sys.argv) > 2 else tmp
            os.makedirs(OUT_DIR, exist_ok=True)
            sys.argv[1:] = []
            main()
            bad = check_against_committed(OUT_DIR)
        for b in bad:
            print("MISMATCH", b)
        print(f"golden --check: {len(bad)} mismatches")
        sys.exit(1 if bad else 0)
    if sys.argv[1:] == ["skrl_gae"]:      # regenerate one fixture without touching the others
        return gen_skrl_gae()
    if sys.argv[1:] == ["rlg_play_steps"]:
        return gen_rlg_play_steps()
    if sys.argv[1:] == ["terms_scale"]:
        return gen_terms_scale()
    print("CaT streams")
    gen_cat("small", 101, 7, S.CAT_TERMS_SMALL, [0.25, 1.0, 0.25, 1.0, 0.5], 16,
            curriculum_at=8, reset_at={5, 11})
    gen_cat("minp", 102, 33, S.CAT_TERMS_SMALL, [0.25, 1.0, 0.25, 1.0, 0.5], 6, min_p=0.05, tau=0.9)
    gen_cat("solo64", 103, 64, S.CAT_TERMS_SOLO12, S.CAT_MAXP_SOLO12, 16, curriculum_at=8, reset_at={7})
    gen_cat("solo4096", 104, 4096, S.CAT_TERMS_SOLO12, S.CAT_MAXP_SOLO12, 8, sub=16, curriculum_at=4)
    gen_curriculum()
    print("terms / envfinish / rms / agent")
    gen_terms()
    gen_terms_scale()
    gen_envfinish()
    gen_rms()
    gen_agent()
    gen_skrl_gae()
    gen_rlg_play_steps()
    print("PPO() runs")
    run_ref_ppo("64x24", 201, N=64, T=24, iters=3, mb=512, epochs=5)
    run_ref_ppo("64x48", 202, N=64, T=48, iters=1, mb=1024, epochs=1, keep_grads=False)
    run_ref_ppo("1x1", 203, N=1, T=1, iters=1, mb=1, epochs=1, keep_grads=False)
    run_ref_ppo("4096x24", 204, N=4096, T=24, iters=1, mb=16384, epochs=1, sub=64, keep_grads=False)
    junk = [p for p, _, f in os.walk(R.REF_ROOT) if p.endswith("__pycache__")]
    assert not junk, f"reference tree polluted: {junk}"


if __name__ == "__main__":
    main()
