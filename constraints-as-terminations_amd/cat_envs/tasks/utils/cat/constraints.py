"""Constraint term functions ``f(env, **params) -> (N,) | (N,C)``, positive = violated.

Same names, parameters and return conventions as the reference's cat/constraints.py (C1..C15,
lines 23-235).  Each function carries a ``describe(env, **params)`` attribute that turns the
term into one row of the descriptor table consumed by ``catppo_cat_terms``; the
``ConstraintManager`` uses it to evaluate ALL terms of a config in a single launch that
writes the packed constraint matrix directly.  Calling a function on its own evaluates the
same kernel for that one term.
"""
from __future__ import annotations

import torch

from cat_envs import native
from cat_envs.native import (TERM_ABS_DIFF_LIMIT, TERM_ABS_DIFF_LIMIT_GATE_CMDY, TERM_ABS_LIMIT,
                             TERM_ABS_LIMIT_GATE_CMDNORM_LT, TERM_ACTION_RATE, TERM_AIR_TIME, TERM_CONTACT_ANY,
                             TERM_FORCE_LIMIT, TERM_GREATER, TERM_LIMIT_MINUS, TERM_N_FOOT_CONTACT,
                             TERM_NORM2_LIMIT, TermDesc)
from cat_envs.shim import SceneEntityCfg


MAX_TERM_IDS = native.TERM_MAX_IDS      # catppo_term_desc.ids


class TooManyIds(ValueError):
    """a selection wider than the descriptor's id table (32 ids; Solo12 has 12 joints / 17 bodies).  The manager
    then leaves the fused table and evaluates the config term by term; per-id terms are evaluated in chunks."""


class TermDescription:
    """one descriptor row + the tensors it points to (kept alive while the row is in use).

    ``cacheable``: every pointer in the row aims at the caller's own storage (no dtype / layout conversion had to
    make a copy), so the row stays valid for as long as the simulator updates those buffers in place."""
    __slots__ = ("c", "width", "is_bool", "forces", "command", "cacheable", "_keep")

    def __init__(self, kind, width, ids, limit=0.0, aux=0.0, x=None, y=None, forces=None, command=None,
                 is_bool=False):
        if len(ids) > MAX_TERM_IDS:
            raise TooManyIds(f"term selects {len(ids)} joints/bodies; the fused term kernel takes {MAX_TERM_IDS}")
        d = TermDesc()
        d.kind, d.width, d.n_ids = kind, width, len(ids)
        for i, v in enumerate(ids):
            d.ids[i] = int(v)
        d.limit, d.aux = float(limit), float(aux)
        same = True
        if x is not None:
            x0, x = x, _rowmajor(x)
            same &= x.data_ptr() == x0.data_ptr() and x.dtype == x0.dtype
            d.x, d.x_ld = x.data_ptr(), x.stride(0)
        if y is not None:
            y0, y = y, _rowmajor(y)
            same &= y.data_ptr() == y0.data_ptr() and y.dtype == y0.dtype
            d.y, d.y_ld = y.data_ptr(), y.stride(0)
        self.c, self.width, self.is_bool = d, width, is_bool
        self.forces = None if forces is None else _dense(forces)
        self.command = None if command is None else _dense(command)
        if forces is not None:
            same &= self.forces.data_ptr() == forces.data_ptr() and self.forces.dtype == forces.dtype
        if command is not None:
            same &= self.command.data_ptr() == command.data_ptr() and self.command.dtype == command.dtype
        self.cacheable = bool(same)
        self._keep = (x, y)


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    if t.ndim != 2 or t.stride(1) != 1:
        t = t.reshape(t.shape[0], -1).contiguous()
    return t


def _dense(t: torch.Tensor) -> torch.Tensor:
    """fp32 with dense trailing dims (the env / row stride may be larger: views of a packed state)"""
    if t.dtype == torch.float32 and t[0].is_contiguous():
        return t
    return t.float().contiguous()


def _ids(sel, n: int):
    if isinstance(sel, slice):
        return list(range(n))[sel]
    if isinstance(sel, torch.Tensor):
        return [int(i) for i in sel.tolist()]
    return [int(i) for i in sel]


def _evaluate(env, d: TermDescription) -> torch.Tensor:
    nat = native.get(env.device)
    out = torch.empty(env.num_envs, d.width, device=nat.device)
    H = d.forces.shape[1] if d.forces is not None else 1
    B = d.forces.shape[2] if d.forces is not None else 1
    nat.cat_terms([d.c], env.num_envs, d.forces, H, B, d.command, out)
    return out


def _term(describe, squeeze: bool):
    """build the public function from its descriptor builder"""
    def fn(env, *args, **kwargs):
        d = describe(env, *args, **kwargs)
        out = _evaluate(env, d)
        if squeeze:
            out = out[:, 0]
        return out > 0.5 if d.is_bool else out
    fn.describe = describe
    fn.__wrapped__ = describe     # inspect.signature(fn) reports the term's real parameters
    return fn


def _command(env):
    return env.command_manager.get_command("base_velocity")


# ------------------------------------------------------------------------------------------------
def _d_joint_position(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_pos.shape[1])
    return TermDescription(TERM_ABS_LIMIT, len(ids), ids, limit, x=data.joint_pos)


def _d_joint_position_when_moving_forward(env, limit: float, velocity_deadzone: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_pos.shape[1])
    return TermDescription(TERM_ABS_DIFF_LIMIT_GATE_CMDY, len(ids), ids, limit, native.f32(velocity_deadzone),
                           x=data.joint_pos, y=data.default_joint_pos, command=_command(env))


def _d_joint_torque(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.applied_torque.shape[1])
    return TermDescription(TERM_ABS_LIMIT, len(ids), ids, limit, x=data.applied_torque)


def _d_joint_velocity(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_vel.shape[1])
    return TermDescription(TERM_ABS_LIMIT, len(ids), ids, limit, x=data.joint_vel)


def _d_joint_acceleration(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_acc.shape[1])
    return TermDescription(TERM_ABS_LIMIT, len(ids), ids, limit, x=data.joint_acc)


def _d_upsidedown(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    return TermDescription(TERM_GREATER, 1, [2], limit, x=data.projected_gravity_b, is_bool=True)


def _d_contact(env, asset_cfg: SceneEntityCfg):
    forces = env.scene[asset_cfg.name].data.net_forces_w_history
    ids = _ids(asset_cfg.body_ids, forces.shape[2])
    return TermDescription(TERM_CONTACT_ANY, 1, ids, 1.0, forces=forces, is_bool=True)


def _d_base_orientation(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    return TermDescription(TERM_NORM2_LIMIT, 1, [], limit, x=data.projected_gravity_b)


def _d_air_time(env, limit: float, velocity_deadzone: float, asset_cfg: SceneEntityCfg):
    sensor = env.scene[asset_cfg.name]
    last_air = sensor.data.last_air_time
    ids = _ids(asset_cfg.body_ids, last_air.shape[1])
    touchdown = getattr(sensor, "first_contact_f32", None)
    if touchdown is None:
        touchdown = sensor.compute_first_contact(env.step_dt)
    return TermDescription(TERM_AIR_TIME, len(ids), ids, limit, native.f32(velocity_deadzone), x=last_air,
                           y=touchdown, command=_command(env))


def _d_n_foot_contact(env, number_of_desired_feet: int, min_command_value: float, asset_cfg: SceneEntityCfg):
    forces = env.scene[asset_cfg.name].data.net_forces_w_history
    ids = _ids(asset_cfg.body_ids, forces.shape[2])
    return TermDescription(TERM_N_FOOT_CONTACT, 1, ids, number_of_desired_feet, native.f32(min_command_value),
                           forces=forces, command=_command(env))


def _d_joint_range(env, limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_pos.shape[1])
    return TermDescription(TERM_ABS_DIFF_LIMIT, len(ids), ids, limit, x=data.joint_pos, y=data.default_joint_pos)


def _d_action_rate(env, limit: float, asset_cfg: SceneEntityCfg):
    am = env.action_manager
    ids = _ids(asset_cfg.joint_ids, am._action.shape[1])
    return TermDescription(TERM_ACTION_RATE, len(ids), ids, limit, native.f32(env.step_dt), x=am._action,
                           y=am._prev_action)


def _d_foot_contact_force(env, limit: float, asset_cfg: SceneEntityCfg):
    forces = env.scene[asset_cfg.name].data.net_forces_w_history
    ids = _ids(asset_cfg.body_ids, forces.shape[2])
    return TermDescription(TERM_FORCE_LIMIT, len(ids), ids, limit, forces=forces)


def _d_min_base_height(env, limit: float, asset_cfg: SceneEntityCfg):
    return TermDescription(TERM_LIMIT_MINUS, 1, [2], limit, x=env.scene[asset_cfg.name].data.root_pos_w)


def _d_no_move(env, velocity_deadzone: float, joint_vel_limit: float, asset_cfg: SceneEntityCfg):
    data = env.scene[asset_cfg.name].data
    ids = _ids(asset_cfg.joint_ids, data.joint_vel.shape[1])
    return TermDescription(TERM_ABS_LIMIT_GATE_CMDNORM_LT, len(ids), ids, joint_vel_limit,
                           native.f32(velocity_deadzone), x=data.joint_vel, command=_command(env))


joint_position = _term(_d_joint_position, squeeze=False)                                    # C1  :23-31
joint_position_when_moving_forward = _term(_d_joint_position_when_moving_forward, False)    # C2  :34-54
joint_torque = _term(_d_joint_torque, squeeze=False)                                        # C3  :57-65
joint_velocity = _term(_d_joint_velocity, squeeze=False)                                    # C4  :68-75
joint_acceleration = _term(_d_joint_acceleration, squeeze=False)                            # C5  :78-85
upsidedown = _term(_d_upsidedown, squeeze=True)                                             # C6  :88-94  (bool)
contact = _term(_d_contact, squeeze=True)                                                   # C7  :97-110 (bool)
base_orientation = _term(_d_base_orientation, squeeze=True)                                 # C8  :113-119
air_time = _term(_d_air_time, squeeze=False)                                                # C9  :122-141
n_foot_contact = _term(_d_n_foot_contact, squeeze=True)                                     # C10 :144-168
joint_range = _term(_d_joint_range, squeeze=False)                                          # C11 :171-181
action_rate = _term(_d_action_rate, squeeze=False)                                          # C12 :184-198
foot_contact_force = _term(_d_foot_contact_force, squeeze=False)                            # C13 :201-211
min_base_height = _term(_d_min_base_height, squeeze=True)                                   # C14 :214-220
no_move = _term(_d_no_move, squeeze=False)                                                  # C15 :223-235

for _name, _fn in list(globals().items()):
    if callable(_fn) and hasattr(_fn, "describe") and not _name.startswith("_"):
        _fn.__name__ = _name
        _fn.__qualname__ = _name
