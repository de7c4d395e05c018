"""ORACLE (test infrastructure, NOT product code) - CaT constraint path, numpy fp32.

CPU restatement of the reference's constraints-as-terminations arithmetic with every
fp32 rounding step spelled out, so that HIP results can be compared bit-for-bit.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does.

Pinned: checked against vectors produced by the reference's own ``CaT`` /
``ConstraintManager`` / term functions run in the build container
(``tests/golden/gen_golden.py`` -> ``tests/golden/cat_*.npz``; test:
``tests/test_oracle_golden.py``).

Reference locations (relative to /root/reference/exts/cat_envs/cat_envs/tasks/utils):
  cat/constraint_manager.py:39-76   CaT.add
  cat/constraint_manager.py:78-82   CaT.get_probs
  cat/constraint_manager.py:190-229 ConstraintManager.reset / compute
  cat/cat_env.py:102-107,118-121    reward scaling + float dones
  cat/curriculums.py:21-42          modify_constraint_p
  cat/constraints.py:23-235         term functions C1..C15
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _as_2d_f32(c) -> np.ndarray:
    """bool -> float, (N,) -> (N,1)   (constraint_manager.py:46-49)."""
    c = np.asarray(c)
    if c.dtype != np.float32:
        c = c.astype(np.float32)
    if c.ndim == 1:
        c = c[:, None]
    return c


class CaTOracle:
    """State + arithmetic of ``CaT`` (constraint_manager.py:22-116)."""

    def __init__(self, tau: float = 0.95, min_p: float = 0.0):
        self.tau = tau
        self.min_p = min_p
        self.running_maxes: dict[str, np.ndarray] = {}
        self.probs: dict[str, np.ndarray] = {}
        self.max_p: dict[str, np.ndarray] = {}
        self.raw_constraints: dict[str, np.ndarray] = {}

    def add(self, name: str, constraint, max_p: float = 0.1) -> None:
        c = _as_2d_f32(constraint)
        self.raw_constraints[name] = c
        # :55  column max over ALL envs, floored at 1e-6
        cmax = np.maximum(c.max(axis=0, keepdims=True), F32(1e-6)).astype(F32)
        # :58-61  first call assigns; afterwards  rm <- fl(fl(rm*tau) + fl(fl32(1-tau)*cmax)), unfused
        if name in self.running_maxes:
            rm = self.running_maxes[name]
            a = (rm * F32(self.tau)).astype(F32)
            b = (F32(1.0 - self.tau) * cmax).astype(F32)
            self.running_maxes[name] = (a + b).astype(F32)
        else:
            self.running_maxes[name] = cmax
        rm = self.running_maxes[name]
        # :64-72  p = c>0 ? fl(min_p + fl(clamp(fl(c/rm),0,1) * fl32(max_p-min_p))) : 0
        q = (c / rm).astype(F32)
        q = np.minimum(np.maximum(q, F32(0.0)), F32(1.0))
        p = (F32(self.min_p) + (q * F32(max_p - self.min_p)).astype(F32)).astype(F32)
        self.probs[name] = np.where(c > 0, p, F32(0.0)).astype(F32)
        self.max_p[name] = np.full((c.shape[1],), max_p, dtype=F32)

    def get_probs(self) -> np.ndarray:
        if not self.probs:
            return np.zeros((0,), F32)
        return np.concatenate(list(self.probs.values()), axis=1).max(axis=1)

    def get_running_maxes(self) -> np.ndarray:
        if not self.running_maxes:
            return np.zeros((0,), F32)
        return np.concatenate(list(self.running_maxes.values()), axis=1)


class ConstraintManagerOracle:
    """Per-step compute + per-episode statistics (constraint_manager.py:190-229)."""

    def __init__(self, term_names, num_envs: int, tau: float = 0.95, min_p: float = 0.0):
        self.cat = CaTOracle(tau, min_p)
        self.term_names = list(term_names)
        self.episode_sums = {n: np.zeros(num_envs, F32) for n in self.term_names}
        self.cstr_mean_values = {n: np.zeros(num_envs, F32) for n in self.term_names}

    def compute(self, term_values: dict, term_max_p: dict) -> np.ndarray:
        for n in self.term_names:
            self.cat.add(n, term_values[n], term_max_p[n])
        cstr_prob = self.cat.get_probs()
        for n in self.term_names:
            m = self.cat.probs[n].max(axis=1)
            self.episode_sums[n] = (self.episode_sums[n] + (m > 0).astype(F32)).astype(F32)
            self.cstr_mean_values[n] = (self.cstr_mean_values[n] + m).astype(F32)
        return cstr_prob

    def reset(self, env_ids, episode_length_buf) -> dict:
        """:190-211.  ``env_ids`` None = all envs.  Division by a zero episode length gives
        NaN/inf exactly like the reference's first reset."""
        ids = slice(None) if env_ids is None else np.asarray(env_ids)
        out = {}
        with np.errstate(divide="ignore", invalid="ignore"):
            for n in self.term_names:
                L = np.asarray(episode_length_buf)[ids].astype(F32)
                v = (self.episode_sums[n][ids] / L).astype(F32)
                p = (self.cstr_mean_values[n][ids] / L).astype(F32)
                out[f"Episode_Constraint_violation/{n}"] = _torch_like_mean(v) * F32(100)
                out[f"Episode_Constraint_probability/{n}"] = _torch_like_mean(p)
                self.episode_sums[n][ids] = 0
                self.cstr_mean_values[n][ids] = 0
        return out


def _torch_like_mean(x: np.ndarray):
    # mean of an fp32 vector; summation order is implementation defined -> compare with tolerance
    if x.size == 0:
        return F32(np.nan)
    return F32(np.sum(x.astype(np.float64)) / x.size)


def env_finish(reward, cstr_prob, reset_mask):
    """cat_env.py:102-107,118-121:  r = max(fl(r*fl(1-p)), 0);  dones = p;  dones[reset] = 1."""
    reward = np.asarray(reward, F32)
    p = np.asarray(cstr_prob, F32)
    r = np.maximum((reward * (F32(1.0) - p).astype(F32)).astype(F32), F32(0.0))
    dones = p.copy()
    dones[np.asarray(reset_mask, bool)] = F32(1.0)
    return r.astype(F32), dones.astype(F32)


def modify_constraint_p(common_step_counter: int, num_steps: int, init_max_p: float) -> float:
    """curriculums.py:21-42 (Python double arithmetic; result becomes the term's max_p)."""
    progress = min(common_step_counter / num_steps, 1.0)
    t_start = 20
    t_end = 1 / init_max_p
    return 1 / (t_start + progress * (t_end - t_start))


# --------------------------------------------------------------------------------------
# Constraint terms C1..C15 (constraints.py:23-235) on a plain dict of sim-state arrays:
#   joint_pos, default_joint_pos, joint_vel, joint_acc, applied_torque  (N,J)
#   projected_gravity_b (N,3), root_pos_w (N,3), command (N,3)
#   net_forces_w_history (N,H,B,3), last_air_time (N,B), first_contact (N,B) bool
#   action, prev_action (N,J), step_dt float
# ``joints`` / ``bodies`` are index lists (SceneEntityCfg.joint_ids / body_ids).
# --------------------------------------------------------------------------------------
def _sel(x, ids):
    return x if ids is None else x[:, ids]


def _fma32(a, b, c):
    """fp32 fused multiply-add, one rounding: the product of two fp32 values is exact in the 64-bit significand of
    x87 long double (48 bits), the sum with an fp32 addend is rounded there once more only when it needs > 64 bits
    (exponent gap > 16 and a sticky tail: the final rounding to fp32 is then still correct unless the 64-bit result
    lands exactly on an fp32 tie, probability ~2^-40 per operation)"""
    w = np.longdouble
    return (a.astype(w) * b.astype(w) + c.astype(w)).astype(F32)


def _norm_last(x):
    """torch.norm(dim=-1) of a short last dimension as the pinning platform computes it (torch 2.10 CPU, where the
    goldens are generated): ONE fp32 FMA chain, acc = x0*x0 (rounded), acc = fma(x_i, x_i, acc), then sqrt.  Found by
    matching five candidate summation orders against torch on 200 000 random vectors (0 mismatches for the chain) and
    re-checked by tests/test_oracle_golden.py against the reference's own outputs (terms.npz, terms_scale.npz)."""
    x = np.asarray(x, F32)
    acc = (x[..., 0] * x[..., 0]).astype(F32)
    for i in range(1, x.shape[-1]):
        acc = _fma32(x[..., i], x[..., i], acc)
    return np.sqrt(acc).astype(F32)


def _force_peak(s, bodies):
    f = s["net_forces_w_history"]
    f = f if bodies is None else f[:, :, bodies]
    return _norm_last(f).max(axis=1)  # (N,B): max over history of |F|


def joint_position(s, limit, joints=None):                      # C1 :23-31
    return (np.abs(_sel(s["joint_pos"], joints)) - F32(limit)).astype(F32)


def joint_position_when_moving_forward(s, limit, velocity_deadzone, joints=None):  # C2 :34-54
    d = (_sel(s["joint_pos"], joints) - _sel(s["default_joint_pos"], joints)).astype(F32)
    c = (np.abs(d) - F32(limit)).astype(F32)
    gate = (np.abs(s["command"][:, 1]) < F32(velocity_deadzone)).astype(F32)[:, None]
    return (c * gate).astype(F32)


def joint_torque(s, limit, joints=None):                        # C3 :57-65
    return (np.abs(_sel(s["applied_torque"], joints)) - F32(limit)).astype(F32)


def joint_velocity(s, limit, joints=None):                      # C4 :68-75
    return (np.abs(_sel(s["joint_vel"], joints)) - F32(limit)).astype(F32)


def joint_acceleration(s, limit, joints=None):                  # C5 :78-85
    return (np.abs(_sel(s["joint_acc"], joints)) - F32(limit)).astype(F32)


def upsidedown(s, limit):                                       # C6 :88-94 (bool)
    return s["projected_gravity_b"][:, 2] > F32(limit)


def contact(s, bodies=None):                                    # C7 :97-110 (bool)
    return (_force_peak(s, bodies) > F32(1.0)).any(axis=1)


def base_orientation(s, limit):                                 # C8 :113-119
    return (_norm_last(s["projected_gravity_b"][:, :2]) - F32(limit)).astype(F32)


def air_time(s, limit, velocity_deadzone, bodies=None):         # C9 :122-141
    td = _sel(s["first_contact"], bodies).astype(F32)
    la = _sel(s["last_air_time"], bodies)
    gate = (_norm_last(s["command"][:, :3]) > F32(velocity_deadzone)).astype(F32)[:, None]
    return (((F32(limit) - la).astype(F32) * td).astype(F32) * gate).astype(F32)


def n_foot_contact(s, number_of_desired_feet, min_command_value, bodies=None):  # C10 :144-168
    n = (_force_peak(s, bodies) > F32(1.0)).sum(axis=1)
    c = np.abs(n - int(number_of_desired_feet))
    gate = (_norm_last(s["command"][:, :3]) > F32(min_command_value)).astype(F32)
    return (c.astype(F32) * gate).astype(F32)


def joint_range(s, limit, joints=None):                         # C11 :171-181
    d = (_sel(s["joint_pos"], joints) - _sel(s["default_joint_pos"], joints)).astype(F32)
    return (np.abs(d) - F32(limit)).astype(F32)


def action_rate(s, limit, joints=None):                         # C12 :184-198
    d = np.abs((_sel(s["action"], joints) - _sel(s["prev_action"], joints)).astype(F32))
    return ((d / F32(s["step_dt"])).astype(F32) - F32(limit)).astype(F32)


def foot_contact_force(s, limit, bodies=None):                  # C13 :201-211
    return (_force_peak(s, bodies) - F32(limit)).astype(F32)


def min_base_height(s, limit):                                  # C14 :214-220
    return (F32(limit) - s["root_pos_w"][:, 2]).astype(F32)


def no_move(s, velocity_deadzone, joint_vel_limit, joints=None):  # C15 :223-235
    c = (np.abs(_sel(s["joint_vel"], joints)) - F32(joint_vel_limit)).astype(F32)
    gate = (_norm_last(s["command"][:, :3]) < F32(velocity_deadzone)).astype(F32)[:, None]
    return (c * gate).astype(F32)
