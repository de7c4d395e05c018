"""Seeded synthetic input streams shared by the golden generator and the tests.

Everything is drawn from ``numpy.random.RandomState`` (the legacy, frozen bit-stream), so
the generator run in the build container and the tests on the GPU box see byte-identical
inputs; each fixture additionally stores a checksum of its inputs.
"""
from __future__ import annotations

import hashlib

import numpy as np

F32 = np.float32

# (name, width, kind) kind: "float" (N,C) | "float1d" (N,) | "bool1d" (N,) bool | "neg" all <= 0
CAT_TERMS_SMALL = [("torque", 12, "float"), ("upside", 1, "bool1d"), ("orient", 1, "float1d"),
                   ("feet", 4, "float"), ("never", 2, "neg")]
CAT_TERMS_SOLO12 = [("joint_torque", 12, "float"), ("joint_velocity", 12, "float"),
                    ("joint_acceleration", 12, "float"), ("action_rate", 12, "float"),
                    ("contact", 1, "bool1d"), ("foot_contact_force", 4, "float"),
                    ("front_hfe_position", 2, "float"), ("upsidedown", 1, "bool1d"),
                    ("hip_position", 4, "float"), ("base_orientation", 1, "float1d"),
                    ("air_time", 4, "float"), ("no_move", 12, "float"), ("two_foot_contact", 1, "float1d")]
CAT_MAXP_SOLO12 = [0.25, 0.25, 0.25, 0.25, 1.0, 1.0, 1.0, 1.0, 0.25, 0.25, 0.25, 0.1, 0.25]


def checksum(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def cat_stream(seed: int, n_envs: int, terms, steps: int):
    """list over steps of {name: array}.  Mix of violated (>0) and satisfied (<=0) entries,
    exact zeros, a column that is never violated, and scale drifting over time so the
    running-max EMA both grows and decays."""
    rs = np.random.RandomState(seed)
    out = []
    for t in range(steps):
        scale = F32(1.0 + 0.5 * np.sin(0.7 * t))
        step = {}
        for name, width, kind in terms:
            if kind == "float":
                x = (rs.standard_normal((n_envs, width)) * scale - 0.3).astype(F32)
                x[rs.rand(n_envs, width) < 0.05] = 0.0
            elif kind == "float1d":
                x = (rs.standard_normal(n_envs) * scale - 0.8).astype(F32)
            elif kind == "bool1d":
                x = rs.rand(n_envs) < 0.1
            elif kind == "neg":
                x = -np.abs(rs.standard_normal((n_envs, width))).astype(F32)
            else:
                raise ValueError(kind)
            step[name] = x
        out.append(step)
    return out


def cat_stream_checksum(stream) -> str:
    return checksum(*[np.asarray(v) for s in stream for v in s.values()])


def sim_state(seed: int, n_envs: int, n_joints: int = 12, n_bodies: int = 13, history: int = 3):
    """Fake Solo12 sim state for the constraint term functions (SURVEY 8d distributions)."""
    rs = np.random.RandomState(seed)
    n = n_envs
    g = rs.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    cmd = np.stack([rs.uniform(-0.3, 1.0, n), rs.uniform(-0.7, 0.7, n), rs.uniform(-0.78, 0.78, n)], 1)
    cmd[rs.rand(n) < 0.3] *= 0.05       # some envs inside the dead-zone
    forces = np.abs(rs.standard_normal((n, history, n_bodies, 3))) * 20.0
    forces[rs.rand(n, history, n_bodies) > 0.3] = 0.0
    return {
        "joint_pos": (rs.standard_normal((n, n_joints)) * 0.5).astype(F32),
        "default_joint_pos": (rs.standard_normal((n, n_joints)) * 0.1).astype(F32),
        "joint_vel": (rs.standard_normal((n, n_joints)) * 8.0).astype(F32),
        "joint_acc": (rs.standard_normal((n, n_joints)) * 400.0).astype(F32),
        "applied_torque": (rs.standard_normal((n, n_joints)) * 2.0).astype(F32),
        "projected_gravity_b": g.astype(F32),
        "root_pos_w": (rs.standard_normal((n, 3)) * 0.1 + np.array([0, 0, 0.25])).astype(F32),
        "command": cmd.astype(F32),
        "net_forces_w_history": forces.astype(F32),
        "last_air_time": rs.uniform(0, 0.5, (n, n_bodies)).astype(F32),
        "first_contact": rs.rand(n, n_bodies) < 0.2,
        "action": rs.standard_normal((n, n_joints)).astype(F32),
        "prev_action": rs.standard_normal((n, n_joints)).astype(F32),
        "step_dt": 0.02,
    }


def sim_state_at_the_limits(seed: int, n_envs: int, steps: int, force_limit: float = 50.0, grav_limit: float = 0.1,
                            deadzone: float = 0.1, min_command: float = 0.5):
    """`steps` fake Solo12 states of `n_envs` envs each (one dict per step, `sim_state` distributions) in which a large
    share of the vectors that feed a NORM sits at, or within a few ulps of, the limit the reference compares the norm
    with (cat/constraints.py:113-119 `||g_xy|| - limit`, :201-211 `max_h ||F|| - limit`, :129-147 / :163-181 / :226-235
    the command-norm gates) - the only inputs for which a last-bit difference in the norm changes `c > 0`, i.e. the
    termination mask.  Planted per step: exact Pythagorean vectors (the norm is exact in every summation order),
    random directions scaled to the limit in fp64 and rounded to fp32 (the norm lands within ~2 ulp of the limit, on
    either side), and the same nudged by +-1..3 ulp per component."""
    out = []
    for k in range(steps):
        st = sim_state(seed + 1000 * k, n_envs)
        rs = np.random.RandomState(seed + 1000 * k + 7)
        n = n_envs

        def at_norm(shape_prefix, dim, target):
            v = rs.standard_normal(shape_prefix + (dim,))
            v *= target / np.linalg.norm(v, axis=-1, keepdims=True)
            v32 = v.astype(F32)
            nudge = rs.randint(-3, 4, size=v32.shape)
            sel = rs.rand(*v32.shape) < 0.5
            i32 = v32.view(np.int32) + np.where(sel, nudge, 0).astype(np.int32)
            return i32.view(F32)

        # contact forces: feet bodies 3, 6, 9, 12 (and the upper-body ids the contact term reads) in half of the envs
        f = st["net_forces_w_history"]
        H, B = f.shape[1], f.shape[2]
        rows = rs.rand(n) < 0.5
        near = np.abs(at_norm((n, H, B), 3, force_limit))
        f[rows] = near[rows]
        trip = np.array([[30.0, 40.0, 0.0], [0.0, 30.0, 40.0], [40.0, 0.0, 30.0], [0.0, 0.0, 50.0], [14.0, 48.0, 0.0]], F32)
        exact = rs.rand(n) < 0.1
        f[exact] = trip[rs.randint(0, len(trip), size=(int(exact.sum()), H, B))] * F32(force_limit / 50.0)
        # the contact terms compare the same norms with 1.0
        one = rs.rand(n) < 0.15
        f[one] = np.abs(at_norm((n, H, B), 3, 1.0))[one]
        # projected gravity: ||g_xy|| at grav_limit in half of the envs (z completes the unit vector)
        g = st["projected_gravity_b"]
        rows = rs.rand(n) < 0.5
        gxy = at_norm((n,), 2, grav_limit)
        g[rows, :2] = gxy[rows]
        g[rows, 2] = -np.sqrt(np.maximum(0.0, 1.0 - (gxy[rows].astype(np.float64) ** 2).sum(1))).astype(F32)
        exact = rs.rand(n) < 0.05
        g[exact, 0], g[exact, 1] = F32(0.6 * grav_limit), F32(0.8 * grav_limit)
        # command: ||cmd[:3]|| at the two gates
        c = st["command"]
        for target, share in ((deadzone, 0.3), (min_command, 0.3)):
            rows = rs.rand(n) < share
            c[rows, :3] = at_norm((n,), 3, target)[rows]
        out.append(st)
    return out


def soft_dones(rs, shape):
    """float dones in [0,1]: mostly exact 0, some exact 1, some fractional (CaT probabilities)."""
    u = rs.rand(*shape)
    d = np.where(u < 0.7, 0.0, np.where(u < 0.75, 1.0, rs.rand(*shape) * 0.5))
    return d.astype(F32)


def gae_inputs(seed: int, T: int, N: int):
    rs = np.random.RandomState(seed)
    return {
        "rewards": rs.uniform(0, 1.5, (T, N)).astype(F32),
        "values": rs.standard_normal((T, N)).astype(F32),
        "dones": soft_dones(rs, (T, N)),
        "true_dones": (rs.rand(T, N) < 0.02).astype(F32),
        "next_value": rs.standard_normal(N).astype(F32),
        "next_done": soft_dones(rs, (N,)),
        "next_true_done": (rs.rand(N) < 0.02).astype(F32),
    }


def env_stream(seed: int, steps: int, N: int, D: int):
    """Open-loop env stream for PPO runs: obs, reward, float dones, bool time-outs per step."""
    rs = np.random.RandomState(seed)
    return {
        "obs0": (rs.standard_normal((N, D)) * 1.5 + 0.2).astype(F32),
        "obs": (rs.standard_normal((steps, N, D)) * 1.5 + 0.2).astype(F32),
        "reward": rs.uniform(0, 1.5, (steps, N)).astype(F32),
        "dones": soft_dones(rs, (steps, N)),
        "timeouts": rs.rand(steps, N) < 0.02,
    }


def agent_weights(seed: int, obs_dim: int, act_dim: int, hidden=(512, 256, 128)):
    """Deterministic (non-orthogonal) weights under the reference's state_dict keys."""
    rs = np.random.RandomState(seed)
    sd = {"actor_logstd": (rs.standard_normal((1, act_dim)) * 0.1).astype(F32)}
    dims = [obs_dim, *hidden]
    for net, out_dim in (("critic", 1), ("actor_mean", act_dim)):
        sizes = list(zip(dims[:-1], dims[1:])) + [(dims[-1], out_dim)]
        for li, (fi, fo) in enumerate(sizes):
            sd[f"{net}.{2 * li}.weight"] = (rs.standard_normal((fo, fi)) * (1.0 / np.sqrt(fi))).astype(F32)
            sd[f"{net}.{2 * li}.bias"] = (rs.standard_normal(fo) * 0.05).astype(F32)
    return sd


def rlg_play_steps_inputs(seed: int, N: int, T: int, D: int, A: int, horizons: int = 2):
    """Inputs of ``horizons`` consecutive ``CaTA2CAgent.play_steps`` calls (reference rl_games/cat_common.py:35-112):
    per step the policy outputs (actions, values, neglogpacs, mus, sigmas) and the env's answer (next obs, rewards (N,1),
    FLOAT dones mixing exact 0, probabilities in (0,1) and exact 1.0, bool time_outs); plus the first observation and
    the bootstrap values after each horizon."""
    rs = np.random.RandomState(seed)
    S = horizons * T
    d = rs.uniform(0, 1, (S, N)).astype(F32)
    u = rs.rand(S, N)
    dones = np.where(u < 0.55, F32(0), np.where(u < 0.85, d, F32(1))).astype(F32)     # 15 % certain terminations
    return {
        "obs0": rs.standard_normal((N, D)).astype(F32),
        "actions": rs.standard_normal((S, N, A)).astype(F32),
        "values": rs.standard_normal((S, N, 1)).astype(F32),
        "neglogpacs": rs.uniform(5, 20, (S, N)).astype(F32),
        "mus": rs.standard_normal((S, N, A)).astype(F32),
        "sigmas": rs.uniform(0.1, 1.0, (S, N, A)).astype(F32),
        "next_obs": rs.standard_normal((S, N, D)).astype(F32),
        "rewards": rs.uniform(-0.5, 1.5, (S, N, 1)).astype(F32),
        "dones": dones,
        "time_outs": rs.rand(S, N) < 0.07,
        "last_values": rs.standard_normal((horizons, N, 1)).astype(F32),
    }
