"""Constraint manager: per-env constraint violations -> float termination probability.

Same public surface as the reference (cat/constraint_manager.py): ``CaT`` with
``add/get_probs/get_raw_constraints/get_running_maxes/get_max_p/get_str/log_all/get_names/
get_vals/reset`` and the dict attributes ``running_maxes/probs/max_p/raw_constraints``;
``ConstraintManager(cfg, env, tau, min_p)`` with ``compute/reset/active_terms/get_term_cfg/
set_term_cfg`` raising the same exceptions.

What differs is where the arithmetic runs.  The reference issues ~15 eager launches and a
host sync per term per env step; here ``ConstraintManager.compute()`` is

    catppo_cat_terms   (all term functions of cat/constraints.py -> cstr[N,K], one launch)
    catppo_cat_step    (column max, running-max EMA, probabilities, per-env max, per-term
                        episode statistics, reward scaling, float dones: three launches)

with no host sync.  All terms live side by side in packed (N,K) / (K,) buffers; the per-term
dict entries the reference exposes are views into them.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from typing import Dict, List

import torch

from cat_envs import native, parallel
from cat_envs.shim import ManagerBase, ManagerTermBase

from .manager_constraint_cfg import ConstraintTermCfg


def _as_matrix(constraint: torch.Tensor, device) -> torch.Tensor:
    """device / float / (N,) -> (N,1) normalisation of constraint_manager.py:42-49"""
    if constraint.device != device:
        constraint = constraint.to(device)
    if not torch.is_floating_point(constraint) or constraint.dtype != torch.float32:
        constraint = constraint.float()
    if constraint.ndim == 1:
        constraint = constraint.unsqueeze(1)
    return constraint


class CaT:
    """Termination probabilities from constraint violations (reference: constraint_manager.py:22-116)."""

    def __init__(self, tau: float = 0.95, min_p: float = 0.0):
        self.running_maxes: Dict[str, torch.Tensor] = {}   # (1,C) EMA of the per-column max violation
        self.probs: Dict[str, torch.Tensor] = {}           # (N,C) termination probabilities
        self.max_p: Dict[str, torch.Tensor] = {}           # (C,)  maximum termination probability
        self.raw_constraints: Dict[str, torch.Tensor] = {} # (N,C) raw constraint values
        self.tau = tau
        self.min_p = min_p
        self._device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self._scratch: Dict[str, tuple] = {}

    def reset(self):
        self.probs.clear()
        self.raw_constraints.clear()

    # -- one term at a time (the reference's call pattern; the manager uses the packed path below)
    def add(self, name: str, constraint: torch.Tensor, max_p: float = 0.1):
        nat = native.get(self._device if self._device.type == "cuda" else None)
        c = _as_matrix(constraint, nat.device).contiguous()
        n, width = c.shape
        first = name not in self.running_maxes
        if first or self.running_maxes[name].shape[1] != width:
            self.running_maxes[name] = torch.zeros(1, width, device=nat.device)
            first = True
        sc = self._scratch.get(name)
        if sc is None or sc[0].shape != (n, width):
            sc = (torch.empty(n, width, device=nat.device), torch.empty(n, device=nat.device),
                  torch.zeros(1, n, device=nat.device), torch.zeros(1, n, device=nat.device),
                  (C.c_int32 * 2)(0, width), (C.c_float * 1)())          # host argument arrays, made once per term
            self._scratch[name] = sc
        probs, prob_max, viol, eprob, off, dp = sc
        dp[0] = max_p - self.min_p                                        # double -> fp32 (RNE) in the assignment
        nat.cat_step(c, off, dp, self.min_p, self.tau, first, self.running_maxes[name].view(-1), prob_max, viol,
                     eprob, probs=probs)
        self.raw_constraints[name] = c
        self.probs[name] = probs
        self.max_p[name] = torch.full((width,), max_p, dtype=torch.float, device=nat.device)

    def get_probs(self) -> torch.Tensor:
        if not self.probs:
            return torch.tensor([], device=self._device)
        return torch.cat(list(self.probs.values()), dim=1).max(1).values

    def get_raw_constraints(self) -> torch.Tensor:
        return (torch.cat(list(self.raw_constraints.values()), dim=1) if self.raw_constraints
                else torch.tensor([], device=self._device))

    def get_running_maxes(self) -> torch.Tensor:
        return (torch.cat(list(self.running_maxes.values()), dim=1) if self.running_maxes
                else torch.tensor([], device=self._device))

    def get_max_p(self) -> torch.Tensor:
        return torch.cat(list(self.max_p.values())) if self.max_p else torch.tensor([], device=self._device)

    def get_str(self, names: List[str] | None = None) -> str:
        names = names or list(self.probs.keys())
        return " ".join(f"{n}: {100.0 * self.probs[n].max(1).values.gt(0.0).float().mean().item():.1f}"
                        for n in names)

    def log_all(self, episode_sums: Dict[str, torch.Tensor]):
        for name, probs in self.probs.items():
            key = f"cstr_{name}"
            values = probs.max(1).values.gt(0.0).float()
            if key not in episode_sums:
                episode_sums[key] = torch.zeros_like(values)
            episode_sums[key].add_(values)

    def get_names(self) -> List[str]:
        return list(self.probs.keys())

    def get_vals(self) -> List[float]:
        return [100.0 * p.max(1).values.gt(0.0).float().mean().item() for p in self.probs.values()]

    # -- packed storage shared with ConstraintManager ------------------------------------------
    def _bind_packed(self, names, widths, n_envs: int, device):
        K = sum(widths)
        self._p_cstr = torch.zeros(n_envs, K, device=device)
        self._p_probs = torch.zeros(n_envs, K, device=device)
        self._p_rm = torch.zeros(K, device=device)
        self._p_first = True
        off = 0
        for n, w in zip(names, widths):
            self.raw_constraints[n] = self._p_cstr[:, off:off + w]
            self.probs[n] = self._p_probs[:, off:off + w]
            self.running_maxes[n] = self._p_rm[off:off + w].unsqueeze(0)
            off += w


class ConstraintManager(ManagerBase):
    """Manager of constraint terms (reference: constraint_manager.py:119-265)."""

    LOG_RING = 128  # reset() results stay valid for this many subsequent resets

    def __init__(self, cfg: object, env, tau: float = 0.95, min_p: float = 0.0):
        self.cat = CaT(tau, min_p)
        self._device = torch.device(env.device)
        self._term_names: List[str] = []
        self._term_index: Dict[str, int] = {}
        self._term_cfgs: List[ConstraintTermCfg] = []
        self._class_term_cfgs: List[ConstraintTermCfg] = []
        super().__init__(cfg, env)          # -> _prepare_terms()

        n, nt = self.num_envs, len(self._term_names)
        # packed per-term statistics; the reference's per-name dicts are row views
        self._ep_viol = torch.zeros(max(nt, 1), n, dtype=torch.float, device=self._device)
        self._ep_prob = torch.zeros(max(nt, 1), n, dtype=torch.float, device=self._device)
        self._episode_sums = {name: self._ep_viol[i] for i, name in enumerate(self._term_names)}
        self._cstr_mean_values = {name: self._ep_prob[i] for i, name in enumerate(self._term_names)}
        self._cstr_prob_buf = torch.zeros(n, dtype=torch.float, device=self._device)
        self._log_ring = torch.zeros(self.LOG_RING, 2 * max(nt, 1), device=self._device)
        self._log_views = None
        self._log_keys = self._log_rows = None
        self._log_pos = 0
        self._bound = False
        self._fused = False
        self._desc_cache = None
        self._desc_key = None
        self._desc_dirty, self._desc_age = False, 0
        self._dp = None
        self._rs_table = self._rs_struct = self._rs_ring = None
        self.term_cache = True
        self._widths: List[int] = []
        self._term_off = None
        #: optional torch.distributed process group: envs are sharded over its ranks and the column
        #: maxima are MAX-all-reduced so every shard sees the single-process running maxima
        self.dist_group = None

    # ------------------------------------------------------------------ introspection
    def __str__(self) -> str:
        msg = f"<ConstraintManager> contains {len(self._term_names)} active terms.\n"
        rows = []
        for index, (name, term_cfg) in enumerate(zip(self._term_names, self._term_cfgs)):
            limit_value = term_cfg.params.get("limit", "-")
            names_value = "-"
            asset_cfg = term_cfg.params.get("asset_cfg")
            if asset_cfg is not None:
                names_value = getattr(asset_cfg, "body_names", None) or getattr(asset_cfg, "joint_names", None) or "-"
            elif "names" in term_cfg.params:
                import warnings
                warnings.warn("Using 'names' parameter is deprecated. Use 'asset_cfg' instead.",
                              DeprecationWarning, stacklevel=2)
                names_value = term_cfg.params["names"]
            rows.append([index, name, limit_value, names_value, term_cfg.max_p])
        try:
            from prettytable import PrettyTable
            table = PrettyTable()
            table.title = "Active Constraint Terms"
            table.field_names = ["Index", "Name", "Limit", "Names", "Max p"]
            table.align["Name"] = "l"
            table.align["Limit"] = "r"
            table.align["Max p"] = "r"
            for r in rows:
                table.add_row(r)
            body = table.get_string()
        except ImportError:
            head = ["Index", "Name", "Limit", "Names", "Max p"]
            body = "Active Constraint Terms\n" + "\n".join(
                " | ".join(str(c) for c in r) for r in [head, *rows])
        return msg + body + "\n"

    @property
    def active_terms(self) -> List[str]:
        return self._term_names

    # ------------------------------------------------------------------ reset
    def reset(self, env_ids: Sequence[int] | torch.Tensor | None = None) -> Dict[str, torch.Tensor]:
        """Episode statistics of the envs in ``env_ids`` (indices, a bool mask, or None = all) as
        0-d tensors, then zero their accumulators.  One launch, no host sync; the returned
        tensors are slots of a ring and stay valid for ``LOG_RING`` further resets."""
        nt = len(self._term_names)
        extras: Dict[str, torch.Tensor] = {}
        if nt:
            mask = None
            if env_ids is not None and not isinstance(env_ids, slice):
                ids = env_ids if isinstance(env_ids, torch.Tensor) else torch.as_tensor(list(env_ids), device=self._device)
                if ids.dtype == torch.bool:
                    mask = ids
                else:
                    mask = torch.zeros(self.num_envs, dtype=torch.bool, device=self._device)
                    mask[ids.to(self._device).long()] = True
            nat = native.get(self._device)
            prev, self._log_pos = self._log_pos, (self._log_pos + 1) % self.LOG_RING
            out = self._log_ring[self._log_pos]      # "no env selected" copies the previous slot (in-kernel)
            nat.cat_reset(self._ep_viol, self._ep_prob, self._env.episode_length_buf, mask, out,
                          prev=self._log_ring[prev])
            extras = self.latest_log()
        for term_cfg in self._class_term_cfgs:
            term_cfg.func.reset(env_ids=env_ids)
        return extras

    def ensure_log_ring(self, n_slots: int):
        """make reset() results stay valid for at least ``n_slots`` further resets (a trainer that reads the
        per-step logs after a rollout of T steps needs T + 1)"""
        if n_slots <= self.LOG_RING:
            return
        ring = torch.zeros(n_slots, self._log_ring.shape[1], device=self._device)
        ring[:self.LOG_RING] = self._log_ring
        self.LOG_RING, self._log_ring, self._log_views, self._log_rows = n_slots, ring, None, None

    @property
    def log_packed(self):
        """(keys, tensor[2*n_terms]) of the latest reset() - lets a trainer stack one tensor per
        step instead of 2*n_terms scalars."""
        if self._log_keys is None:
            self._log_keys = []
            for key in self._term_names:
                self._log_keys += [f"Episode_Constraint_violation/{key}", f"Episode_Constraint_probability/{key}"]
        if self._log_rows is None or len(self._log_rows) != self.LOG_RING:
            self._log_rows = list(self._log_ring.unbind(0))
        return self._log_keys, self._log_rows[self._log_pos]

    # ------------------------------------------------------------------ compute
    def _bind(self, nat):
        """first compute(): learn every term's width, allocate the packed buffers"""
        env = self._env
        descs = []
        from .constraints import TooManyIds
        for cfg in self._term_cfgs:
            d = getattr(cfg.func, "describe", None)
            try:
                descs.append(d(env, **cfg.params) if d is not None else None)
            except TooManyIds:                     # wider than the descriptor's id table: per-term path
                descs.append(None)
        self._fused = all(d is not None for d in descs) and len(descs) <= 16
        if self._fused:
            widths = [d.width for d in descs]
        else:
            widths = []
            for cfg in self._term_cfgs:
                out = cfg.func(env, **cfg.params)
                widths.append(1 if out.ndim == 1 else out.shape[1])
        self._widths = widths
        off = [0]
        for w in widths:
            off.append(off[-1] + w)
        self._term_off = (C.c_int32 * len(off))(*off)
        self.cat._bind_packed(self._term_names, widths, self.num_envs, nat.device)
        self._bound = True

    def _describe_terms(self):
        """descriptor table of the fused term kernel.  Built once and reused: the rows hold raw device pointers
        into the simulator's persistent state buffers (IsaacLab's ``data.*`` tensors are allocated once and
        updated in place) plus the term parameters, which only change through ``set_term_cfg`` (that drops
        the cache).  Tables with a converted (copied) input are never cached; ``self.term_cache = False``
        re-describes every step regardless."""
        env = self._env
        # Raw device pointers may only be kept across steps when the env guarantees that its ``data.*`` tensors
        # are allocated once and updated in place (``persistent_state_buffers``).  Any other env (IsaacLab lazy
        # buffers, a simulator that rebinds tensors) is re-described every step, like the reference, which
        # re-evaluates the term inputs on every compute().
        persistent = bool(getattr(env, "persistent_state_buffers", False))
        cache = self._desc_cache if (self.term_cache and persistent) else None
        if cache is not None:
            # parameter snapshot: checked whenever set_term_cfg saw a changed object, and every 64th step to catch
            # in-place edits of term_cfg.params that never went through set_term_cfg
            self._desc_age += 1
            if not self._desc_dirty and self._desc_age < 64:
                return cache
            self._desc_dirty, self._desc_age = False, 0
            if self._desc_key == self._params_key():
                return cache
        rows, forces, command, H, B, keep = [], None, None, 1, 1, []
        for cfg in self._term_cfgs:
            d = cfg.func.describe(env, **cfg.params)
            rows.append(d.c)
            keep.append(d)                          # keeps the tensors behind the pointers alive
            if d.forces is not None:
                forces, H, B = d.forces, d.forces.shape[1], d.forces.shape[2]
            if d.command is not None:
                command = d.command
        arr = (native.TermDesc * len(rows))(*rows)
        self._desc_keep = keep
        table = (arr, forces, H, B, command)
        # cache only tables whose pointers aim at the simulator's own buffers: a term input that needed a dtype /
        # layout conversion was COPIED by describe(), and a cached pointer to that copy would go stale
        self._desc_cache = table if (persistent and all(d.cacheable for d in keep)) else None
        self._desc_key = self._params_key()
        return table

    @staticmethod
    def _term_key(cfg):
        """snapshot of everything ONE descriptor row is built from: the term function and its parameters (scalars by
        value, SceneEntityCfg by its resolved ids); ``max_p`` is not part of it (it travels with every launch)"""
        row = [id(cfg.func)]
        for k, v in cfg.params.items():
            if isinstance(v, (int, float, str, bool)) or v is None:
                row.append((k, v))
            else:
                ids = (getattr(v, "joint_ids", None), getattr(v, "body_ids", None))
                row.append((k, id(v), tuple(str(i) for i in ids)))
        return tuple(row)

    def _params_key(self):
        """per-term snapshots of the whole table.  In-place edits of ``term_cfg.params`` - with or without a
        ``set_term_cfg`` call - change the key and rebuild the table."""
        return tuple(self._term_key(cfg) for cfg in self._term_cfgs)

    def compute(self, reward: torch.Tensor | None = None, reset_mask: torch.Tensor | None = None,
                dones: torch.Tensor | None = None) -> torch.Tensor:
        """Termination probability per env.  The optional arguments fuse the three lines of
        ``CaTEnv.step`` that consume it (cat_env.py:102-107,118-121) into the same launch:
        ``reward`` is scaled in place by (1-p) and clipped at 0, ``dones`` receives p with hard
        resets (``reset_mask``) overwritten by 1."""
        nat = native.get(self._device)
        if not self._term_names:
            return self.cat.get_probs()
        if not self._bound:
            self._bind(nat)
        env, cat = self._env, self.cat
        group = self.dist_group
        sharded = group is not None and parallel.active(group)
        dp = (C.c_float * len(self._term_cfgs))(*[native.f32(c.max_p - cat.min_p) for c in self._term_cfgs])
        args = dict(reward=reward, reset_mask=reset_mask, dones=dones, probs=cat._p_probs)
        have_colmax = False
        if self._fused:
            descs, forces, H, B, command = self._describe_terms()
            if not sharded:
                nat.cat_terms_step(descs, forces, H, B, command, cat._p_cstr, self._term_off, dp, cat.min_p, cat.tau,
                                   cat._p_first, cat._p_rm, self._cstr_prob_buf, self._ep_viol, self._ep_prob, **args)
                cat._p_first = False
                return self._cstr_prob_buf
            if not hasattr(self, "_colmax"):
                self._colmax = torch.zeros_like(cat._p_rm)
            nat.cat_terms_colmax(descs, forces, H, B, command, cat._p_cstr, self._colmax)
            have_colmax = True
        else:
            off = 0
            for cfg, w in zip(self._term_cfgs, self._widths):
                out = cfg.func(env, **cfg.params)
                cat._p_cstr[:, off:off + w].copy_(out.reshape(self.num_envs, w))   # bool -> float in the copy
                off += w
        if sharded:
            if not hasattr(self, "_colmax"):
                self._colmax = torch.zeros_like(cat._p_rm)
            if not have_colmax:
                nat.cat_colmax(cat._p_cstr, self._colmax)
            parallel.allreduce_max_(self._colmax, group)
            nat.cat_apply(cat._p_cstr, self._term_off, dp, cat.min_p, cat.tau, cat._p_first, self._colmax, cat._p_rm,
                          self._cstr_prob_buf, self._ep_viol, self._ep_prob, **args)
        else:
            nat.cat_step(cat._p_cstr, self._term_off, dp, cat.min_p, cat.tau, cat._p_first, cat._p_rm,
                         self._cstr_prob_buf, self._ep_viol, self._ep_prob, **args)
        cat._p_first = False
        return self._cstr_prob_buf

    # ------------------------------------------------------------------ fused rollout step (catppo_rollout_pre/_post)
    def can_fuse_rollout(self, nat) -> bool:
        """the manager's part of an env step can run inside the two-launch fused rollout step: every term has a
        descriptor (``describe``) and the config fits the kernel's tables"""
        if not self._term_names:
            return False
        if not self._bound:
            self._bind(nat)
        return bool(self._fused) and len(self._term_cfgs) <= 16

    def fill_rollout_step(self, st) -> None:
        """write the manager's buffers / parameters of THIS step into a ``native.RolloutStep`` and advance the log
        ring (compute() + reset(reset_mask) of the unfused path).  Runs once per env step on the host: everything
        that does not change from step to step is written only when the descriptor table is (re)built."""
        cat = self.cat
        table = self._describe_terms()
        if table is not self._rs_table or st is not self._rs_struct:
            descs, forces, H, B, command = table
            self._rs_table, self._rs_struct = table, st
            self._dp = (C.c_float * len(self._term_cfgs))()
            st.K, st.n_terms = cat._p_cstr.shape[1], len(self._term_cfgs)
            st.desc = C.cast(descs, C.c_void_p)
            st.forces = forces.data_ptr() if forces is not None else None
            st.forces_env_stride = forces.stride(0) if forces is not None else 0
            st.H, st.B = int(H), int(B)
            st.command = command.data_ptr() if command is not None else None
            st.command_ld = command.stride(0) if command is not None else 0
            st.cstr = cat._p_cstr.data_ptr()
            st.term_off, st.term_dp = C.cast(self._term_off, C.c_void_p), C.cast(self._dp, C.c_void_p)
            st.min_p, st.tau, st.one_minus_tau = native.f32(cat.min_p), native.f32(cat.tau), native.f32(1.0 - cat.tau)
            st.rm, st.cstr_prob = cat._p_rm.data_ptr(), self._cstr_prob_buf.data_ptr()
            st.ep_viol, st.ep_prob = self._ep_viol.data_ptr(), self._ep_prob.data_ptr()
            st.probs = cat._p_probs.data_ptr()
            self._rs_ring = (self._log_ring.data_ptr(), self._log_ring.stride(0) * 4, self.LOG_RING)
        dp, min_p = self._dp, cat.min_p
        for i, c in enumerate(self._term_cfgs):
            dp[i] = c.max_p - min_p              # double -> fp32 (RNE) in the assignment: fl32(max_p - min_p)
        st.first_call = 1 if cat._p_first else 0
        base, stride, ring = self._rs_ring
        if ring != self.LOG_RING:                # ring was enlarged
            self._rs_ring = base, stride, ring = (self._log_ring.data_ptr(), self._log_ring.stride(0) * 4, self.LOG_RING)
        prev, self._log_pos = self._log_pos, (self._log_pos + 1) % ring
        st.log_prev, st.log_out = base + prev * stride, base + self._log_pos * stride
        cat._p_first = False

    def latest_log(self, copy: bool = True) -> Dict[str, torch.Tensor]:
        """the dict ``reset()`` would have returned for the ring slot written last"""
        if self._log_views is None:
            self._log_views = []
            for r in range(self.LOG_RING):
                d = {}
                for t, key in enumerate(self._term_names):
                    d[f"Episode_Constraint_violation/{key}"] = self._log_ring[r, 2 * t]
                    d[f"Episode_Constraint_probability/{key}"] = self._log_ring[r, 2 * t + 1]
                self._log_views.append(d)
        return dict(self._log_views[self._log_pos]) if copy else self._log_views[self._log_pos]

    @property
    def max_p(self) -> Dict[str, torch.Tensor]:
        return {n: torch.full((w,), c.max_p, dtype=torch.float, device=self._device)
                for n, w, c in zip(self._term_names, self._widths, self._term_cfgs)}

    # ------------------------------------------------------------------ term cfg access
    def set_term_cfg(self, term_name: str, cfg: ConstraintTermCfg):
        i = self._term_index.get(term_name)
        if i is None:
            raise ValueError(f"Constraint term '{term_name}' not found.")
        old = self._term_cfgs[i]
        self._term_cfgs[i] = cfg
        # The reference reads ``term_cfg.params`` on every compute() (constraint_manager.py:213-221), so an edit that
        # arrives here - a new cfg object OR the same object after an in-place ``params[...] = x`` - must reach the very
        # next step: compare this term's parameter snapshot with the one the cached descriptor row was built from (one
        # small tuple; the curriculum calls this for every term at every reset with unchanged params: max_p is not in
        # the snapshot, it travels with every launch).  The 64-step sweep in _describe_terms() only remains for edits
        # that never go through set_term_cfg.
        key = self._desc_key
        if cfg is not old or key is None or self._term_key(cfg) != key[i]:
            self._desc_dirty = True
            if cfg.func is not old.func:
                self._desc_cache = None

    def get_term_cfg(self, term_name: str) -> ConstraintTermCfg:
        i = self._term_index.get(term_name)
        if i is None:
            raise ValueError(f"Constraint term '{term_name}' not found.")
        return self._term_cfgs[i]

    def _prepare_terms(self):
        cfg_items = self.cfg.items() if isinstance(self.cfg, dict) else self.cfg.__dict__.items()
        for term_name, term_cfg in cfg_items:
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, ConstraintTermCfg):
                raise TypeError(f"Configuration for term '{term_name}' is not ConstraintTermCfg. "
                                f"Received: '{type(term_cfg)}'.")
            if not isinstance(term_cfg.max_p, (float, int)):
                raise TypeError(f"Limit for term '{term_name}' must be float or int. "
                                f"Received: '{type(term_cfg.max_p)}'.")
            self._resolve_common_term_cfg(term_name, term_cfg, min_argc=1)
            self._term_index[term_name] = len(self._term_names)
            self._term_names.append(term_name)
            self._term_cfgs.append(term_cfg)
            if isinstance(term_cfg.func, ManagerTermBase):
                self._class_term_cfgs.append(term_cfg)
