"""CaT environment: manager-based RL env whose ``step`` returns the constraint termination
probability as float ``dones`` and scales the reward by ``(1 - p)``.

Reference: cat/cat_env.py.  Only lines 92-121 and 147 of the reference ``step`` are on the hot
path (counters, terminations, ``constraint_manager.compute()``, reward scaling, float dones,
hard resets, manager resets, return tuple); the physics loop (:61-88) is Isaac Sim / PhysX,
which cannot run on AMD GPUs and is out of scope.  This class therefore drives the same
managers from a *synthetic simulator*: seeded, device-resident, pre-generated streams of Solo12
sim state (joint state, gravity, commands, contact forces, air times), rewards, hard resets
and observations.  Every step the current slab of the stream is moved into persistent state
buffers (what a simulator's ``scene.update`` does) - by one ``copy_`` in ``step``, inside the
first launch of the fused ``step_into`` (``catppo_rollout_step.sim_src``) - and every manager
reads views of those buffers, exactly like IsaacLab's ``scene[...]`` data objects.

No host synchronisation happens inside ``step``: resets are handled with masks
(``exact_reset_sync=True`` restores the reference's ``nonzero()``-based control flow).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import types
from collections.abc import Sequence

import torch

from cat_envs import native
from cat_envs.shim import SceneEntityCfg  # noqa: F401  (re-export for term configs)

from .constraint_manager import ConstraintManager

SOLO12_JOINTS = ["FL_HAA", "FL_HFE", "FL_KFE", "FR_HAA", "FR_HFE", "FR_KFE",
                 "HL_HAA", "HL_HFE", "HL_KFE", "HR_HAA", "HR_HFE", "HR_KFE"]
SOLO12_BODIES = ["base_link"] + [f"{leg}_{part}" for leg in ("FL", "FR", "HL", "HR")
                                 for part in ("SHOULDER", "UPPER_LEG", "LOWER_LEG", "FOOT")]
DEFAULT_JOINT_POS = [0.05, 0.4, -0.8, -0.05, 0.4, -0.8, 0.05, 0.4, -0.8, -0.05, 0.4, -0.8]


class _Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


#: the fused env step advances the simulator state inside catppo_rollout_pre (A/B switch, read once)
_FUSED_SIM_COPY = os.environ.get("CATPPO_FUSED_SIM_COPY", "1") != "0"


class SyntheticSolo12Sim:
    """Pre-generated sim-state streams (SURVEY 8d distributions), resident in HBM.

    Packed layout per env and step (floats): joint_pos 12 | joint_vel 12 | joint_acc 12 |
    applied_torque 12 | projected_gravity_b 3 | root_pos_w 3 | command 3 | last_air_time B |
    first_contact B | net_forces_w_history H*B*3 | reward 1 | hard_reset 1 | obs D.
    """

    H = 3

    def __init__(self, num_envs: int, obs_dim: int, device, seed: int, stream_steps: int):
        self.N, self.D, self.device, self.S = num_envs, obs_dim, torch.device(device), stream_steps
        J, B, H = len(SOLO12_JOINTS), len(SOLO12_BODIES), self.H
        self.J, self.B = J, B
        fields = [("joint_pos", J), ("joint_vel", J), ("joint_acc", J), ("applied_torque", J),
                  ("projected_gravity_b", 3), ("root_pos_w", 3), ("command", 3), ("last_air_time", B),
                  ("first_contact", B), ("forces", H * B * 3), ("reward", 1), ("hard_reset", 1), ("obs", obs_dim)]
        self.off, o = {}, 0
        for name, w in fields:
            self.off[name] = (o, w)
            o += w
        self.F = (o + 3) // 4 * 4
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        N, S, F = num_envs, stream_steps, self.F
        st = torch.zeros(S, N, F, device=self.device)

        def put(name, value):
            a, w = self.off[name]
            st[:, :, a:a + w] = value.reshape(S, N, w)

        def randn(*shape):
            return torch.randn(*shape, device=self.device, generator=g)

        def rand(*shape):
            return torch.rand(*shape, device=self.device, generator=g)

        dq = torch.tensor(DEFAULT_JOINT_POS, device=self.device)
        put("joint_pos", dq + randn(S, N, J) * 0.5)
        put("joint_vel", randn(S, N, J) * 8.0)
        put("joint_acc", randn(S, N, J) * 400.0)
        put("applied_torque", randn(S, N, J) * 2.0)
        grav = torch.cat([randn(S, N, 2) * 0.12, -torch.ones(S, N, 1, device=self.device)], -1)
        grav[..., 2] = torch.where(rand(S, N) < 0.003, 1.0, -1.0)      # rare roll-over
        put("projected_gravity_b", grav / grav.norm(dim=-1, keepdim=True))
        put("root_pos_w", torch.cat([randn(S, N, 2), 0.25 + randn(S, N, 1) * 0.03], -1))
        lo = torch.tensor([-0.3, -0.7, -0.78], device=self.device)
        hi = torch.tensor([1.0, 0.7, 0.78], device=self.device)
        cmd = lo + rand(S, N, 3) * (hi - lo)
        cmd = cmd * (cmd.norm(dim=-1, keepdim=True) > 0.1)              # dead-zone like the reference commands
        cmd = cmd * (rand(S, N, 1) > 0.02)                              # rel_standing_envs = 0.02
        put("command", cmd)
        put("last_air_time", rand(S, N, B) * 0.5)
        put("first_contact", (rand(S, N, B) < 0.1).float())
        is_foot = torch.tensor([n.endswith("_FOOT") for n in SOLO12_BODIES], device=self.device)
        p_contact = torch.where(is_foot, 0.5, 0.002)
        forces = randn(S, N, H, B, 3).abs() * 20.0 * (rand(S, N, H, B, 1) < p_contact.view(1, 1, 1, B, 1))
        put("forces", forces)
        put("reward", rand(S, N, 1) * 1.5)
        put("hard_reset", (rand(S, N, 1) < 0.01).float())
        put("obs", randn(S, N, obs_dim))
        self.stream = st
        self._slabs = list(st.unbind(0))
        self.cur = torch.zeros(N, F, device=self.device)       # persistent "simulator state" buffers
        self.cursor = -1
        self.default_joint_pos = dq.repeat(N, 1).contiguous()
        self._build_scene()

    def view(self, name):
        a, w = self.off[name]
        return self.cur[:, a:a + w]

    def _build_scene(self):
        N, H, B = self.N, self.H, self.B
        a, w = self.off["forces"]
        forces = self.cur.as_strided((N, H, B, 3), (self.F, B * 3, 3, 1), self.cur.storage_offset() + a)
        robot = types.SimpleNamespace(
            joint_names=list(SOLO12_JOINTS), body_names=list(SOLO12_BODIES),
            data=types.SimpleNamespace(
                joint_pos=self.view("joint_pos"), default_joint_pos=self.default_joint_pos,
                joint_vel=self.view("joint_vel"), joint_acc=self.view("joint_acc"),
                applied_torque=self.view("applied_torque"), projected_gravity_b=self.view("projected_gravity_b"),
                root_pos_w=self.view("root_pos_w")))
        fc = self.view("first_contact")
        sensor = types.SimpleNamespace(
            joint_names=[], body_names=list(SOLO12_BODIES),
            data=types.SimpleNamespace(net_forces_w_history=forces, last_air_time=self.view("last_air_time")),
            first_contact_f32=fc, compute_first_contact=lambda dt: fc > 0.5)
        self.scene = {"robot": robot, "contact_forces": sensor}

    def advance(self) -> torch.Tensor:
        """move to the next step of the stream WITHOUT touching the state buffer: returns the slab that holds the new
        state (the fused env step hands it to catppo_rollout_pre, which copies it into ``cur`` inside its own launch)"""
        self.cursor = (self.cursor + 1) % self.S
        return self._slabs[self.cursor]

    def step(self):
        self.cur.copy_(self.advance())                         # "scene.update": one contiguous slab


class _ActionManager:
    def __init__(self, n, a, device):
        self._action = torch.zeros(n, a, device=device)
        self._prev_action = torch.zeros(n, a, device=device)
        self.total_action_dim = a

    def process_action(self, action):
        self._prev_action.copy_(self._action)
        self._action.copy_(action)

    def reset(self, env_ids=None):
        """IsaacLab's ActionManager.reset: the action history of the envs that reset starts from zero, so the
        action-rate constraint of their first step sees ``|a - 0| / dt`` (mask, index list or None = all)."""
        if env_ids is None or isinstance(env_ids, slice):
            self._action.zero_()
            self._prev_action.zero_()
        elif isinstance(env_ids, torch.Tensor) and env_ids.dtype == torch.bool:
            keep = (~env_ids).unsqueeze(1)
            self._action.mul_(keep)
            self._prev_action.mul_(keep)
        else:
            self._action[env_ids] = 0.0
            self._prev_action[env_ids] = 0.0
        return {}


class _CurriculumManager:
    def __init__(self, cfg, env):
        self._env = env
        items = cfg.items() if isinstance(cfg, dict) else (cfg.__dict__.items() if cfg is not None else [])
        self._terms = [(n, t) for n, t in items if t is not None]
        self._keyed = [(f"Curriculum/{n}", t) for n, t in self._terms]
        self._state = {}

    def compute(self, env_ids=None):
        env, state = self._env, self._state
        for key, term in self._keyed:
            state[key] = term.func(env, env_ids, **term.params)

    def reset(self, env_ids=None):
        return {k: v for k, v in self._state.items() if isinstance(v, (int, float))}


class CaTEnv:
    """Drop-in for the reference ``CaTEnv`` on the path PPO consumes (``unwrapped.num_envs``,
    ``single_observation_space["policy"]``, ``single_action_space``, ``reset()``, ``step()``)."""

    persistent_state_buffers = True      # term descriptors may cache raw pointers

    def __init__(self, cfg, render_mode: str | None = None, **kwargs):
        self.cfg = cfg
        self.render_mode = render_mode
        self.num_envs = int(cfg.scene.num_envs)
        dev = getattr(cfg.sim, "device", "cuda:0")
        if not torch.cuda.is_available() or torch.device(dev).type != "cuda":
            raise RuntimeError("CaTEnv needs a HIP device (MI355X): the constraint manager and the synthetic "
                               "simulator are device resident; there is no CPU fallback")
        self.device = torch.device(dev)
        self.physics_dt = cfg.sim.dt
        self.step_dt = cfg.sim.dt * cfg.decimation
        self.max_episode_length_s = cfg.episode_length_s
        self.max_episode_length = math.ceil(cfg.episode_length_s / self.step_dt)
        syn = cfg.synthetic
        self.obs_dim, self.act_dim = int(syn.obs_dim), len(SOLO12_JOINTS)
        seed = int(getattr(cfg, "seed", 0) or 0)
        self.sim = SyntheticSolo12Sim(self.num_envs, self.obs_dim, self.device, seed + int(syn.seed_offset),
                                      int(syn.stream_steps))
        self.scene = self.sim.scene
        self.exact_reset_sync = bool(getattr(syn, "exact_reset_sync", False))
        g = torch.Generator(device=self.device)
        g.manual_seed(seed + 17)
        self.episode_length_buf = torch.randint(0, self.max_episode_length, (self.num_envs,), device=self.device,
                                                generator=g, dtype=torch.long)
        self.common_step_counter = 0
        self._sim_step_counter = 0
        self.extras: dict = {}
        self.single_observation_space = {"policy": _Space((self.obs_dim,))}
        self.single_action_space = _Space((self.act_dim,))
        self.reward_buf = torch.zeros(self.num_envs, device=self.device)
        self._dones = torch.zeros(self.num_envs, device=self.device)
        self.reset_buf = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self.reset_terminated = torch.zeros_like(self.reset_buf)
        self.reset_time_outs = torch.zeros_like(self.reset_buf)
        self.obs_buf = {"policy": self.sim.view("obs")}
        self._rstep = None
        self._nat = None
        self.load_managers()

    # gym-style plumbing ---------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    def close(self):
        pass

    def seed(self, seed: int = -1) -> int:
        return seed

    # managers --------------------------------------------------------------------------------
    def load_managers(self):
        """reference: cat_env.py:18-40 (constraint manager created after the base managers)"""
        self.action_manager = _ActionManager(self.num_envs, self.act_dim, self.device)
        cmd = self.sim.view("command")
        self.command_manager = types.SimpleNamespace(get_command=lambda name: cmd, reset=lambda ids=None: {},
                                                     compute=lambda dt: None)
        self.curriculum_manager = _CurriculumManager(getattr(self.cfg, "curriculum", None), self)
        if hasattr(self.cfg, "constraints"):
            self.constraint_manager = ConstraintManager(self.cfg.constraints, self)
            print("[INFO] Constraint Manager: ", self.constraint_manager)

    # episode control -------------------------------------------------------------------------
    def reset(self, seed: int | None = None, options=None):
        self.sim.step()
        self.extras = {}
        return self.obs_buf, self.extras

    def step(self, action: torch.Tensor):
        """reference: cat_env.py:42-147 with the physics loop replaced by the synthetic stream."""
        self._sim_step_counter += self.cfg.decimation
        self.sim.step()
        self.common_step_counter += 1
        # -- one launch: process_action, episode_length_buf += 1 (:92), terminations (:95-97: hard resets
        #    from the stream, time-outs from the counter), reward_manager output -> reward_buf
        am = self.action_manager
        action = action if (action.dtype == torch.float32 and action.is_contiguous()) else action.float().contiguous()
        native.get(self.device).env_pre_step(action, am._action, am._prev_action, self.episode_length_buf,
                                             self.max_episode_length, self.sim.view("hard_reset"),
                                             self.sim.view("reward"), self.reset_time_outs, self.reset_terminated,
                                             self.reset_buf, self.reward_buf)
        # -- CaT (:99-107,118-121): probability, reward *= (1-p) clipped at 0, dones = p, dones[reset] = 1
        if hasattr(self.cfg, "constraints"):
            self.constraint_manager.compute(reward=self.reward_buf, reset_mask=self.reset_buf, dones=self._dones)
            dones = self._dones
        else:
            dones = self.reset_buf.float()
        # -- resets (:118-125)
        if self.exact_reset_sync:
            ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)   # host sync, reference control flow
            if len(ids) > 0:
                self._reset_idx(ids)
        else:
            self._reset_idx(self.reset_buf)
        # -- observations (:144)
        return self.obs_buf, self.reward_buf, dones, self.reset_time_outs, self.extras

    # fused path ------------------------------------------------------------------------------------
    def can_step_into(self, obs_dim_ok: bool = True) -> bool:
        """``step_into`` is available: constraints configured with describable terms, mask-based resets"""
        cm = getattr(self, "constraint_manager", None)
        return (cm is not None and not self.exact_reset_sync and obs_dim_ok and self.obs_dim <= 512
                and cm.can_fuse_rollout(native.get(self.device)))

    def step_into(self, action: torch.Tensor, sink):
        """``step`` with everything after the simulator update fused into two calls (catppo_rollout_pre /
        catppo_rollout_post: three launches), including the consumer's part: ``sink`` (see ``cleanrl.ppo.RolloutSink``) names the
        rollout-buffer rows of this step and the observation normaliser, so rewards / dones / time-outs and the
        normalised next observation are written where PPO wants them.  Same return tuple as ``step``; ``extras["log"]``
        is the manager's view dict of the ring slot written by this step, the curriculum scalars ride in
        ``extras["log_host"]``."""
        nat = self._nat
        cm = self.constraint_manager
        self._sim_step_counter += self.cfg.decimation
        # "scene.update": the new simulator state.  Fused: catppo_rollout_pre copies the stream's next slab into the state
        # buffer inside its own launch and reads its inputs straight from the slab (catppo_rollout_step.sim_src) - one
        # 5 us copy kernel and a launch boundary less per env step; CATPPO_FUSED_SIM_COPY=0 keeps the separate copy.
        slab = None
        if _FUSED_SIM_COPY:
            slab = self.sim.advance()
        else:
            self.sim.step()
        self.common_step_counter += 1
        st = self._rstep
        if st is None:
            from cat_envs import parallel
            self._parallel = parallel
            nat = self._nat = native.get(self.device)
            st = self._rstep = native.RolloutStep()
            am = self.action_manager
            st.N, st.A, st.D = self.num_envs, self.act_dim, self.obs_dim
            st.action, st.prev_action = am._action.data_ptr(), am._prev_action.data_ptr()
            st.episode_length, st.max_episode_length = self.episode_length_buf.data_ptr(), self.max_episode_length
            hr, rw = self.sim.view("hard_reset"), self.sim.view("reward")
            st.hard_reset, st.hard_reset_stride = hr.data_ptr(), hr.stride(0)
            st.reward_src, st.reward_stride = rw.data_ptr(), rw.stride(0)
            st.time_outs, st.terminated = self.reset_time_outs.data_ptr(), self.reset_terminated.data_ptr()
            st.reset, st.reward, st.dones = self.reset_buf.data_ptr(), self.reward_buf.data_ptr(), self._dones.data_ptr()
            st.zero_action_on_reset = 1
            obs = self.sim.view("obs")
            st.obs_raw, st.obs_ld = obs.data_ptr(), obs.stride(0)
            self._xchg = nat.rollout_xchg_new(cm.cat._p_cstr.shape[1], self.obs_dim)
            st.xchg = self._xchg.data_ptr()
            self._xchg_views = nat.rollout_xchg_views(self._xchg, cm.cat._p_cstr.shape[1], self.obs_dim)
            self._xchg_all = None
            st.sim_state, st.sim_row_bytes = self.sim.cur.data_ptr(), self.sim.F * 4
            self._rstep_ref = C.byref(st)
        if action.dtype != torch.float32 or not action.is_contiguous():
            action = action.float().contiguous()
        st.action_in = action.data_ptr()
        st.sim_src = slab.data_ptr() if slab is not None else None
        cm.fill_rollout_step(st)
        sink.fill(st)
        lib, h, stream = nat.lib, nat.h, nat._stream()
        rc = lib.catppo_rollout_pre(h, self._rstep_ref, stream)
        if rc:
            nat._ok(rc)
        # env-sharded runs exchange the record rollout_pre just wrote {CaT column maxima | observation moment sums}.
        # The maxima travel only in exact mode (cm.dist_group, set by the trainer under dist_exact); the moment sums
        # whenever the consumer's normaliser is global (sink.obs_group) - its divisor obs_rows_total is then the global
        # env count, so the sums MUST be global too (with dist_exact=False they once stayed local while the divisor was
        # global: the running mean shrank by 1/world per update).  Both wanted (exact mode): ONE all-gather of the
        # record, folded by rollout_post; only one of them: an all-reduce of that half.
        par = self._parallel
        group, obs_group = cm.dist_group, getattr(sink, "obs_group", None)
        g_on = group is not None and par.active(group)
        o_on = obs_group is not None and par.active(obs_group)
        if g_on and o_on and group is obs_group:
            # exact mode: ONE collective per env step - all-gather this rank's record {colmax | sums}; rollout_post folds
            # the records of all ranks itself (MAX is exact, the sums run in rank order on every rank)
            w = par.world_size(group)
            if self._xchg_all is None or self._xchg_all.numel() != w * self._xchg.numel():
                self._xchg_all = torch.zeros(w * self._xchg.numel(), dtype=torch.uint8, device=self.device)
            # set on EVERY pass: a step through the other branch (another sink / trainer on the same env with a different
            # dist_exact or obs group) leaves xchg_records = 0, and rollout_post would then fold the local record only
            st.xchg_gathered, st.xchg_records = self._xchg_all.data_ptr(), w
            par.allgather_bytes_(self._xchg, self._xchg_all, group)
        else:
            st.xchg_records = 0
            if g_on:
                par.allreduce_max_(self._xchg_views[0], group)    # exact (max is order independent): masks stay bit-exact
            if o_on:
                par.allreduce_sum_(self._xchg_views[1], obs_group)   # fp64 [sum x | sum x^2]
        rc = lib.catppo_rollout_post(h, self._rstep_ref, stream)
        if rc:
            nat._ok(rc)
        # curriculum AFTER the CaT step, like _reset_idx (host scalars: the new max_p travels with the next step's
        # launch; the reference runs it whenever some env resets - with thousands of envs that is every step)
        cur = self.curriculum_manager
        cur.compute(env_ids=self.reset_buf)
        ex = self.extras
        ex["log"] = cm.latest_log(copy=False)
        ex["log_host"] = cur._state
        ex["log_packed"] = cm.log_packed
        return self.obs_buf, self.reward_buf, self._dones, self.reset_time_outs, ex

    def _reset_idx(self, env_ids: Sequence[int] | torch.Tensor):
        """reference: cat_env.py:149-200.  ``env_ids`` may be a bool mask (sync-free path)."""
        self.curriculum_manager.compute(env_ids=env_ids)
        log = {}
        log.update(self.action_manager.reset(env_ids))
        if hasattr(self.cfg, "constraints"):
            log.update(self.constraint_manager.reset(env_ids))
            self.extras["log_packed"] = self.constraint_manager.log_packed
        log.update(self.curriculum_manager.reset(env_ids))
        self.extras["log"] = log
        if isinstance(env_ids, torch.Tensor) and env_ids.dtype == torch.bool:
            self.episode_length_buf.masked_fill_(env_ids, 0)
        else:
            self.episode_length_buf[env_ids] = 0
