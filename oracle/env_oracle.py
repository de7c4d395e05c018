"""ORACLE (test infrastructure, NOT product code) - CPU mirror of the CaT env step.

Replays the same synthetic sim-state stream the device env consumes (a CPU copy of
``SyntheticSolo12Sim.stream``) through the numpy CaT oracle, following the on-path lines of the
reference ``CaTEnv.step`` (cat/cat_env.py:92-121,147) and ``_reset_idx`` (:149-200):
counters, terminations, ``constraint_manager.compute()``, ``reward*(1-p)`` clipped at 0, float
dones with hard resets set to 1, curriculum + manager reset for the envs that reset.

Used by tests / smoke / the CPU baseline of bench.py only.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import cat_oracle as CO

F32 = np.float32

# name -> (oracle term fn, kwargs builder from the cfg params + resolved ids)
_TERMS = {
    "joint_position": lambda s, p, j, b: CO.joint_position(s, p["limit"], j),
    "joint_position_when_moving_forward": lambda s, p, j, b: CO.joint_position_when_moving_forward(
        s, p["limit"], p["velocity_deadzone"], j),
    "joint_torque": lambda s, p, j, b: CO.joint_torque(s, p["limit"], j),
    "joint_velocity": lambda s, p, j, b: CO.joint_velocity(s, p["limit"], j),
    "joint_acceleration": lambda s, p, j, b: CO.joint_acceleration(s, p["limit"], j),
    "upsidedown": lambda s, p, j, b: CO.upsidedown(s, p["limit"]),
    "contact": lambda s, p, j, b: CO.contact(s, b),
    "base_orientation": lambda s, p, j, b: CO.base_orientation(s, p["limit"]),
    "air_time": lambda s, p, j, b: CO.air_time(s, p["limit"], p["velocity_deadzone"], b),
    "n_foot_contact": lambda s, p, j, b: CO.n_foot_contact(s, p["number_of_desired_feet"], p["min_command_value"], b),
    "joint_range": lambda s, p, j, b: CO.joint_range(s, p["limit"], j),
    "action_rate": lambda s, p, j, b: CO.action_rate(s, p["limit"], j),
    "foot_contact_force": lambda s, p, j, b: CO.foot_contact_force(s, p["limit"], b),
    "min_base_height": lambda s, p, j, b: CO.min_base_height(s, p["limit"]),
    "no_move": lambda s, p, j, b: CO.no_move(s, p["velocity_deadzone"], p["joint_vel_limit"], j),
}


class CaTEnvOracle:
    """terms: list of dicts {name, func (key of _TERMS), params, joints, bodies, max_p};
    curriculum: list of dicts {term_name, num_steps, init_max_p}."""

    def __init__(self, stream: np.ndarray, offsets: dict, n_bodies: int, history: int, default_joint_pos, terms,
                 curriculum, episode_length0, max_episode_length: int, step_dt: float, tau=0.95, min_p=0.0):
        self.stream, self.off = stream, offsets
        self.S, self.N, _ = stream.shape
        self.B, self.H = n_bodies, history
        self.default_joint_pos = np.asarray(default_joint_pos, F32)
        self.terms, self.curriculum = terms, curriculum
        self.max_p = {t["name"]: float(t["max_p"]) for t in terms}
        self.mgr = CO.ConstraintManagerOracle([t["name"] for t in terms], self.N, tau, min_p)
        self.episode_length = np.asarray(episode_length0).astype(np.int64).copy()
        self.max_episode_length, self.step_dt = max_episode_length, step_dt
        self.common_step_counter = 0
        self.cursor = -1
        self.action = np.zeros((self.N, 12), F32)
        self.prev_action = np.zeros((self.N, 12), F32)
        self.last_log = {}

    def _f(self, slab, name):
        a, w = self.off[name]
        return slab[:, a:a + w]

    def _state(self, slab):
        f = self._f
        return {"joint_pos": f(slab, "joint_pos"), "default_joint_pos": self.default_joint_pos,
                "joint_vel": f(slab, "joint_vel"), "joint_acc": f(slab, "joint_acc"),
                "applied_torque": f(slab, "applied_torque"), "projected_gravity_b": f(slab, "projected_gravity_b"),
                "root_pos_w": f(slab, "root_pos_w"), "command": f(slab, "command"),
                "last_air_time": f(slab, "last_air_time"), "first_contact": f(slab, "first_contact") > 0.5,
                "net_forces_w_history": f(slab, "forces").reshape(self.N, self.H, self.B, 3),
                "action": self.action, "prev_action": self.prev_action, "step_dt": self.step_dt}

    def reset(self):
        self.cursor = (self.cursor + 1) % self.S
        return {"policy": torch.from_numpy(self._f(self.stream[self.cursor], "obs").copy())}, {}

    def step(self, action: torch.Tensor):
        self.prev_action, self.action = self.action, np.asarray(action.detach().cpu().numpy(), F32)
        self.cursor = (self.cursor + 1) % self.S
        slab = self.stream[self.cursor]
        self.episode_length += 1
        self.common_step_counter += 1
        time_outs = self.episode_length >= self.max_episode_length
        terminated = self._f(slab, "hard_reset")[:, 0] > 0.5
        reset = terminated | time_outs
        st = self._state(slab)
        vals = {t["name"]: _TERMS[t["func"]](st, t["params"], t.get("joints"), t.get("bodies")) for t in self.terms}
        p = self.mgr.compute(vals, self.max_p)
        reward, dones = CO.env_finish(self._f(slab, "reward")[:, 0], p, reset)
        # sync-free device env: curriculum + manager reset run every step with the reset mask
        for c in self.curriculum:
            self.max_p[c["term_name"]] = CO.modify_constraint_p(self.common_step_counter, c["num_steps"],
                                                                 c["init_max_p"])
        ids = np.nonzero(reset)[0]
        if len(ids):
            self.last_log = self.mgr.reset(ids, self.episode_length)
        self.episode_length[ids] = 0
        # IsaacLab ActionManager.reset(env_ids): the action history of the reset envs restarts from zero
        # (self.action may alias the caller's array: copy before writing)
        if len(ids):
            self.action, self.prev_action = self.action.copy(), self.prev_action.copy()
            self.action[ids] = 0
            self.prev_action[ids] = 0
        obs = {"policy": torch.from_numpy(self._f(slab, "obs").copy())}
        return obs, torch.from_numpy(reward), torch.from_numpy(dones), torch.from_numpy(time_outs), {"log": self.last_log}


def from_device_env(env) -> CaTEnvOracle:
    """Build the CPU mirror of a ``cat_envs`` CaTEnv (copies its stream to host memory)."""
    sim = env.sim
    mgr = env.constraint_manager
    terms = []
    for name, cfg in zip(mgr._term_names, mgr._term_cfgs):
        asset = cfg.params.get("asset_cfg")
        ent = env.scene[asset.name] if asset is not None else None

        def ids(sel, n):
            return None if sel is None or isinstance(sel, slice) else [int(i) for i in sel]
        terms.append({"name": name, "func": cfg.func.__name__, "max_p": cfg.max_p,
                      "params": {k: v for k, v in cfg.params.items() if k != "asset_cfg"},
                      "joints": ids(getattr(asset, "joint_ids", None), 12),
                      "bodies": ids(getattr(asset, "body_ids", None), sim.B) if ent is not None else None})
        if terms[-1]["bodies"] is None and "contact_forces" == getattr(asset, "name", ""):
            terms[-1]["bodies"] = None
    cur = [dict(term_name=t.params["term_name"], num_steps=t.params["num_steps"], init_max_p=t.params["init_max_p"])
           for _, t in env.curriculum_manager._terms]
    return CaTEnvOracle(sim.stream.cpu().numpy(), sim.off, sim.B, sim.H, sim.default_joint_pos.cpu().numpy(), terms,
                        cur, env.episode_length_buf.cpu().numpy(), env.max_episode_length, env.step_dt,
                        tau=mgr.cat.tau, min_p=mgr.cat.min_p)
