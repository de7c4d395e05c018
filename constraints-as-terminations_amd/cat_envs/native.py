"""ctypes binding of libcatppo.so (include/catppo.h) - the only way the package reaches the GPU.

There is NO fallback: if the shared library is missing or no gfx950 device is visible,
``Native()`` raises.  Tensors are torch device tensors used purely as device memory
(``data_ptr()``); the stream handed to every call is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcatppo.so")

MAX_HIDDEN = 4
TERM_MAX_IDS = 32     # CATPPO_TERM_MAX_IDS

# catppo_term_kind
TERM_ABS_LIMIT, TERM_ABS_DIFF_LIMIT, TERM_ABS_DIFF_LIMIT_GATE_CMDY, TERM_GREATER = 0, 1, 2, 3
TERM_CONTACT_ANY, TERM_NORM2_LIMIT, TERM_AIR_TIME, TERM_N_FOOT_CONTACT = 4, 5, 6, 7
TERM_ACTION_RATE, TERM_FORCE_LIMIT, TERM_LIMIT_MINUS, TERM_ABS_LIMIT_GATE_CMDNORM_LT = 8, 9, 10, 11


class MlpShape(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MAX_HIDDEN), ("mfma_bf16", C.c_int32)]


class MlpLayout(C.Structure):
    _fields_ = [("obs_pad", C.c_int32), ("n_flat", C.c_int64), ("n_params", C.c_int64),
                ("off_logstd", C.c_int64),
                ("off_w", (C.c_int64 * (MAX_HIDDEN + 1)) * 2), ("off_b", (C.c_int64 * (MAX_HIDDEN + 1)) * 2),
                ("in_dim", C.c_int32 * (MAX_HIDDEN + 1)), ("out_dim", (C.c_int32 * (MAX_HIDDEN + 1)) * 2)]


class PpoHparams(C.Structure):
    _fields_ = [("clip_coef", C.c_float), ("ent_coef", C.c_float), ("vf_coef", C.c_float),
                ("norm_adv", C.c_int32), ("clip_vloss", C.c_int32), ("inv_global_batch", C.c_float),
                ("adv_stats_external", C.c_int32)]


class TermDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("n_ids", C.c_int32), ("ids", C.c_int32 * TERM_MAX_IDS),
                ("limit", C.c_float), ("aux", C.c_float), ("x", C.c_void_p), ("y", C.c_void_p),
                ("x_ld", C.c_int32), ("y_ld", C.c_int32)]


class IterState(C.Structure):
    """catppo_iter_state (device resident; this mirror is only used for its size and for read-backs in tests)"""
    _fields_ = [("seed", C.c_uint64), ("iteration", C.c_int64), ("adam_step", C.c_int64), ("lr", C.c_double),
                ("kl_mark", C.c_double), ("n_mark", C.c_double), ("last_kl", C.c_double), ("adam_step_size", C.c_float),
                ("adam_bc2_sqrt", C.c_float)]


RLG_MAX_VALUE_SIZE = 4


class RlgMeters(C.Structure):
    """catppo_rlg_meters (device resident; this mirror is used for its size and for read-backs)"""
    _fields_ = [("mean_rewards", C.c_float * RLG_MAX_VALUE_SIZE), ("mean_shaped_rewards", C.c_float * RLG_MAX_VALUE_SIZE),
                ("mean_lengths", C.c_float), ("size_rewards", C.c_int32), ("size_shaped", C.c_int32),
                ("size_lengths", C.c_int32), ("max_size", C.c_int32), ("last_done_count", C.c_int32)]


class RolloutStep(C.Structure):
    """catppo_rollout_step: argument block of the fused rollout step (field order = include/catppo.h)"""
    _fields_ = [
        ("N", C.c_int64), ("A", C.c_int32), ("D", C.c_int32), ("K", C.c_int32), ("n_terms", C.c_int32),
        ("action_in", C.c_void_p), ("action", C.c_void_p), ("prev_action", C.c_void_p),
        ("episode_length", C.c_void_p), ("max_episode_length", C.c_int64),
        ("hard_reset", C.c_void_p), ("hard_reset_stride", C.c_int64),
        ("reward_src", C.c_void_p), ("reward_stride", C.c_int64),
        ("time_outs", C.c_void_p), ("terminated", C.c_void_p), ("reset", C.c_void_p), ("reward", C.c_void_p),
        ("desc", C.c_void_p), ("forces", C.c_void_p), ("forces_env_stride", C.c_int64), ("H", C.c_int32),
        ("B", C.c_int32), ("command", C.c_void_p), ("command_ld", C.c_int32), ("cstr", C.c_void_p),
        ("term_off", C.c_void_p), ("term_dp", C.c_void_p), ("min_p", C.c_float), ("tau", C.c_float),
        ("one_minus_tau", C.c_float), ("first_call", C.c_int32),
        ("rm", C.c_void_p), ("cstr_prob", C.c_void_p), ("dones", C.c_void_p), ("ep_viol", C.c_void_p),
        ("ep_prob", C.c_void_p), ("probs", C.c_void_p),
        ("log_prev", C.c_void_p), ("log_out", C.c_void_p), ("zero_action_on_reset", C.c_int32),
        ("rewards_t", C.c_void_p), ("dones_t1", C.c_void_p), ("true_dones_t1", C.c_void_p),
        ("plane_dtype", C.c_int32),
        ("obs_raw", C.c_void_p), ("obs_ld", C.c_int64),
        ("obs_mean", C.c_void_p), ("obs_var", C.c_void_p), ("obs_count", C.c_void_p), ("obs_eps", C.c_float),
        ("obs_rows_total", C.c_double), ("obs_out", C.c_void_p), ("obs_out_ld", C.c_int64),
        ("xchg", C.c_void_p), ("xchg_gathered", C.c_void_p), ("xchg_records", C.c_int32),
        ("sim_src", C.c_void_p), ("sim_state", C.c_void_p), ("sim_row_bytes", C.c_int64)]


F32, F16, F64 = 0, 1, 2                 # CATPPO_F32 / _F16 / _F64
SUM, MAX = 0, 1                         # CATPPO_SUM / _MAX
LR_FIXED, LR_LINEAR, LR_KEEP = 0, 1, 2
GAE_SERIAL, GAE_SCAN = 0, 1
UNIQUE_ID_BYTES = 128

_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_u64 = C.c_uint64

_SIGNATURES = {
    "catppo_version": (C.c_int, []),
    "catppo_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "catppo_destroy": (None, [_vp]),
    "catppo_last_error": (C.c_char_p, [_vp]),
    "catppo_reserve": (C.c_int, [_vp, C.c_uint64]),
    "catppo_cat_step": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _f32, _f32, _f32, _i32, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp]),
    "catppo_cat_colmax": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "catppo_cat_apply": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _f32, _f32, _f32, _i32, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catppo_cat_reset": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "catppo_cat_terms": (C.c_int, [_vp, C.POINTER(TermDesc), _i32, _i64, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _i32,
                                   _vp]),
    "catppo_cat_terms_step": (C.c_int, [_vp, C.POINTER(TermDesc), _i32, _i64, _vp, _i64, _i32, _i32, _vp, _i32, _vp,
                                        _i32, _vp, _vp, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _vp]),
    "catppo_cat_terms_colmax": (C.c_int, [_vp, C.POINTER(TermDesc), _i32, _i64, _vp, _i64, _i32, _i32, _vp, _i32, _vp,
                                          _i32, _vp, _vp]),
    "catppo_env_pre_step": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64,
                                      _vp]),
    "catppo_adv_stats": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "catppo_adv_normalize": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "catppo_value_bootstrap": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _i64, _vp]),
    "catppo_rms_merge": (C.c_int, [_vp, _vp, _f64, _i32, _vp, _vp, _vp, _vp]),
    "catppo_mlp_layout_of": (C.c_int, [C.POINTER(MlpShape), C.POINTER(MlpLayout)]),
    "catppo_mlp_workspace_bytes": (C.c_uint64, [C.POINTER(MlpShape), _i64]),
    "catppo_ppo_minibatch_grad": (C.c_int, [_vp, C.POINTER(MlpShape), C.POINTER(PpoHparams), _vp, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catppo_ppo_minibatch_grad_packed": (C.c_int, [_vp, C.POINTER(MlpShape), C.POINTER(PpoHparams), _vp, _vp, _vp,
                                                   _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catppo_clip_adam": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f64, _f64, _f64, _f64, _i64, _vp]),
    # ABI 0.4: gradient + clip + Adam of a single process in one call
    "catppo_ppo_minibatch_step_packed": (C.c_int, [_vp, C.POINTER(MlpShape), C.POINTER(PpoHparams), _vp, _vp, _vp,
                                                   _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f64, _f64,
                                                   _f64, _vp, _vp]),
    # ---- ABI 0.2
    "catppo_iter_init": (C.c_int, [_vp, _vp, _u64, _f64, _vp]),
    "catppo_iter_begin": (C.c_int, [_vp, _vp, _f64, _i64, C.c_int, _vp]),
    "catppo_clip_adam_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f64, _f64, _f64, _vp, _vp]),
    "catppo_kl_mean": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "catppo_kl_adaptive_lr": (C.c_int, [_vp, _vp, _vp, _f64, _f64, _f64, _f64, _f64, _vp]),
    "catppo_ppo_gather_ex": (C.c_int, [_vp, C.POINTER(MlpShape), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _i32,
                                       _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catppo_rms_moments_ex": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i64, _vp, _vp]),
    "catppo_rollout_store_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _i64, _vp]),
    "catppo_rms_update_ex": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "catppo_rms_normalize_ex": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i64, _vp, _vp, _f32, _vp, _i64, _vp]),
    "catppo_rollout_step_sizeof": (C.c_uint64, []),
    # ---- ABI 0.6: one entry per family (the names of ABI <= 0.5 are inline wrappers in include/catppo_compat.h)
    "catppo_debug_clip_branches": (C.c_int, [_vp, _vp]),
    "catppo_rollout_xchg_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "catppo_gae_planes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp,
                                    _i32, _i64, _vp]),
    "catppo_policy_step": (C.c_int, [_vp, C.POINTER(MlpShape), _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                     C.c_int, _vp]),
    "catppo_rollout_pre": (C.c_int, [_vp, C.POINTER(RolloutStep), _vp]),
    "catppo_rollout_post": (C.c_int, [_vp, C.POINTER(RolloutStep), _vp]),
    "catppo_rollout_defer_tail": (C.c_int, [_vp, C.c_int, _vp]),
    "catppo_graph_begin": (C.c_int, [_vp, _vp]),
    "catppo_graph_end": (C.c_int, [_vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "catppo_graph_launch": (C.c_int, [_vp, C.c_int, _vp]),
    "catppo_graph_destroy": (C.c_int, [_vp, C.c_int]),
    "catppo_comm_unique_id": (C.c_int, [_vp]),
    "catppo_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "catppo_comm_destroy": (C.c_int, [_vp]),
    "catppo_allreduce": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp]),
    "catppo_broadcast": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp]),
    # ---- ABI 0.3
    "catppo_comm_probe": (C.c_int, [_vp]),
    "catppo_allgather": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "catppo_adv_moments_parts": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    "catppo_rlg_meters_init": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "catppo_rlg_episode_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    # ---- ABI 0.5
    "catppo_plan_log": (C.c_char_p, [_vp, C.c_int]),
    "catppo_adv_moments_keyed": (C.c_int, [_vp, _vp, C.c_int, _vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    # ---- ABI 0.4
    "catppo_set_grad_overlap": (C.c_int, [_vp, C.c_int]),
}

EXPORTS = tuple(_SIGNATURES)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_lib = None
_contexts: dict = {}


def get(device=None) -> "Native":
    """process-wide context per HIP device (the manager, the env and PPO share one workspace and
    therefore one stream order)"""
    import torch as _t
    if isinstance(device, _t.device) and device.index is not None:      # fast path: called several times per env step
        ctx = _contexts.get(device.index)
        if ctx is not None:
            return ctx
    if not _t.cuda.is_available():
        raise RuntimeError("no HIP device visible: the CaT-PPO hot path runs on MI355X (gfx950) only; "
                           "there is no CPU fallback")
    idx = _t.cuda.current_device() if device is None or _t.device(device).index is None else _t.device(device).index
    if idx not in _contexts:
        _contexts[idx] = Native(_t.device("cuda", idx))
    return _contexts[idx]


def load_library(path: Optional[str] = None):
    """dlopen libcatppo.so and attach prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("CATPPO_LIB") or LIB_PATH      # CATPPO_LIB: A/B another build of the library
    if not os.path.exists(path):
        raise RuntimeError(
            f"libcatppo.so not found at {path}: build it with "
            "`python constraints-as-terminations_amd/build.py` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.catppo_rollout_step_sizeof() != C.sizeof(RolloutStep):
        raise RuntimeError(f"catppo_rollout_step layout mismatch: library {lib.catppo_rollout_step_sizeof()} B, "
                           f"binding {C.sizeof(RolloutStep)} B (rebuild libcatppo.so)")
    _lib = lib
    return lib


def f32(x: float) -> float:
    """round a Python double to fp32 the way a torch fp32 tensor op consumes a Python scalar"""
    return C.c_float(x).value


def shape_of(obs_dim: int, act_dim: int, hidden, mfma_bf16=False) -> MlpShape:
    """mfma_bf16: False/0 = fp32 MFMA, True/1 = bf16 operands, 2 = split-bf16 (bf16x3)"""
    s = MlpShape()
    s.obs_dim, s.act_dim, s.n_hidden = int(obs_dim), int(act_dim), len(hidden)
    s.mfma_bf16 = int(mfma_bf16)
    for i, h in enumerate(hidden):
        s.hidden[i] = int(h)
    return s


def layout_of(shape: MlpShape) -> MlpLayout:
    lay = MlpLayout()
    rc = load_library().catppo_mlp_layout_of(C.byref(shape), C.byref(lay))
    if rc != 0:
        raise ValueError(f"unsupported MLP shape (obs={shape.obs_dim}, act={shape.act_dim}, "
                         f"hidden={list(shape.hidden)[:shape.n_hidden]}): hidden widths must be multiples of 64, "
                         "the last one in {64,128,256,512}, act_dim <= 15")
    return lay


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str, contiguous: bool = True):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on the GPU (HIP) device")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


class Native:
    """One catppo context bound to one HIP device."""

    def __init__(self, device: Optional[torch.device] = None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the CaT-PPO hot path runs on MI355X (gfx950) only; "
                               "there is no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"device {device} is not a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self._dev_index = int(self.device.index)
        h = _vp()
        rc = self.lib.catppo_create(self.device.index, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"catppo_create(device={self.device.index}) failed with {rc} "
                               "(-3: not a gfx950 device)")
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and getattr(self, "lib", None) is not None:
            try:
                self.lib.catppo_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        # raw hipStream_t of torch's current stream on this device.  The private accessor is ~15x cheaper than
        # torch.cuda.current_stream(): 14 library calls per env step made the public one 40 % of the host time
        if _raw_stream is not None:
            return _raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _ok(self, rc: int):
        if rc != 0:
            msg = self.lib.catppo_last_error(self.h)
            raise RuntimeError(f"libcatppo error {rc}: {msg.decode() if msg else ''}")

    def reserve(self, nbytes: int):
        self._ok(self.lib.catppo_reserve(self.h, int(nbytes)))

    # ------------------------------------------------------------------ CaT
    def cat_step(self, cstr, term_off_host, term_dp_host, min_p, tau, first_call, rm, cstr_prob, ep_viol, ep_prob,
                 reward=None, reset_mask=None, dones=None, probs=None):
        """term_off_host: int32 ctypes array [n_terms+1]; term_dp_host: float ctypes array [n_terms]."""
        N, K = cstr.shape
        n_terms = len(term_dp_host)
        self._ok(self.lib.catppo_cat_step(
            self.h, _p(_chk(cstr, torch.float32, "cstr")), N, K, C.cast(term_off_host, _vp), n_terms,
            C.cast(term_dp_host, _vp), f32(min_p), f32(tau), f32(1.0 - tau), int(bool(first_call)), _p(rm),
            _p(reward), _p(reset_mask), _p(cstr_prob), _p(dones), _p(ep_viol), _p(ep_prob), _p(probs),
            self._stream()))

    def cat_colmax(self, cstr, colmax):
        N, K = cstr.shape
        self._ok(self.lib.catppo_cat_colmax(self.h, _p(_chk(cstr, torch.float32, "cstr")), N, K, _p(colmax),
                                            self._stream()))

    def cat_apply(self, cstr, term_off_host, term_dp_host, min_p, tau, first_call, colmax, rm, cstr_prob, ep_viol,
                  ep_prob, reward=None, reset_mask=None, dones=None, probs=None):
        N, K = cstr.shape
        n_terms = len(term_dp_host)
        self._ok(self.lib.catppo_cat_apply(
            self.h, _p(_chk(cstr, torch.float32, "cstr")), N, K, C.cast(term_off_host, _vp), n_terms,
            C.cast(term_dp_host, _vp), f32(min_p), f32(tau), f32(1.0 - tau), int(bool(first_call)), _p(colmax),
            _p(rm), _p(reward), _p(reset_mask), _p(cstr_prob), _p(dones), _p(ep_viol), _p(ep_prob), _p(probs),
            self._stream()))

    def cat_terms(self, descs, n_envs, forces, H, B, command, cstr):
        """forces: (N,H,B,3) view whose env stride may exceed H*B*3; command: (N,3) view with any row stride"""
        arr = descs if isinstance(descs, C.Array) else (TermDesc * len(descs))(*descs)
        fstride = forces.stride(0) if forces is not None else 0
        cld = command.stride(0) if command is not None else 0
        self._ok(self.lib.catppo_cat_terms(self.h, arr, len(arr), int(n_envs), _p(forces), fstride, int(H), int(B),
                                           _p(command), cld, _p(cstr), cstr.shape[1], self._stream()))

    def cat_terms_step(self, descs, forces, H, B, command, cstr, term_off_host, term_dp_host, min_p, tau, first_call,
                       rm, cstr_prob, ep_viol, ep_prob, reward=None, reset_mask=None, dones=None, probs=None):
        """cat_terms + cat_step in three launches (single-process path)"""
        N, K = cstr.shape
        fstride = forces.stride(0) if forces is not None else 0
        cld = command.stride(0) if command is not None else 0
        self._ok(self.lib.catppo_cat_terms_step(
            self.h, descs, len(descs), N, _p(forces), fstride, int(H), int(B), _p(command), cld, _p(cstr), K,
            C.cast(term_off_host, _vp), C.cast(term_dp_host, _vp), f32(min_p), f32(tau), f32(1.0 - tau),
            int(bool(first_call)), _p(rm), _p(reward), _p(reset_mask), _p(cstr_prob), _p(dones), _p(ep_viol),
            _p(ep_prob), _p(probs), self._stream()))

    def cat_terms_colmax(self, descs, forces, H, B, command, cstr, colmax):
        """terms -> cstr and the local column maxima (env-sharded path: all-reduce MAX, then cat_apply)"""
        N, K = cstr.shape
        fstride = forces.stride(0) if forces is not None else 0
        cld = command.stride(0) if command is not None else 0
        self._ok(self.lib.catppo_cat_terms_colmax(self.h, descs, len(descs), N, _p(forces), fstride, int(H), int(B),
                                                  _p(command), cld, _p(cstr), K, _p(colmax), self._stream()))

    def cat_reset(self, ep_viol, ep_prob, episode_length, mask, out, prev=None):
        n_terms, N = ep_viol.shape
        _chk(episode_length, torch.int64, "episode_length")
        self._ok(self.lib.catppo_cat_reset(self.h, _p(ep_viol), _p(ep_prob), _p(episode_length), _p(mask), n_terms,
                                           N, _p(prev), _p(out), self._stream()))

    # ------------------------------------------------------------------ env bookkeeping
    def env_pre_step(self, action_in, action, prev_action, episode_length, max_len, hard_reset, reward_src,
                     time_outs, terminated, reset, reward_out):
        n, a = action.shape
        self._ok(self.lib.catppo_env_pre_step(
            self.h, _p(_chk(action_in, torch.float32, "action")), _p(action), _p(prev_action), a,
            _p(_chk(episode_length, torch.int64, "episode_length")), int(max_len), _p(hard_reset), hard_reset.stride(0),
            _p(reward_src), reward_src.stride(0), _p(time_outs), _p(terminated), _p(reset), _p(reward_out), n,
            self._stream()))

    def rollout_store(self, reward, dones, time_outs, rewards_t, dones_t1, true_dones_t1):
        self._ok(self.lib.catppo_rollout_store_ex(self.h, _p(reward), _p(dones), _p(time_outs), _p(rewards_t),
                                                  _p(dones_t1), _p(true_dones_t1), F32, reward.numel(), self._stream()))

    def adv_moments_parts(self, adv_part_g, parts_per_mb, total, minibatch, moments):
        """per-minibatch {sum, sum of squares, rows} from the chunk sums of the epoch gather (fp64)"""
        _chk(adv_part_g, torch.float64, "adv_part_g")
        _chk(moments, torch.float64, "moments")
        self._ok(self.lib.catppo_adv_moments_parts(self.h, _p(adv_part_g), int(parts_per_mb), int(total),
                                                   int(minibatch), _p(moments), self._stream()))

    def plan_log(self, enable: int = -1) -> str:
        """ABI 0.5: start (1) / stop (0) / read (-1) the record of launch decisions of the MLP entry points"""
        return (self.lib.catppo_plan_log(self.h, int(enable)) or b"").decode()

    def adv_moments_keyed(self, advantages, st, n_epochs, total, minibatch, parts_scratch, moments):
        """{sum, sum of squares, rows} of every minibatch of ALL ``n_epochs`` keyed permutations of this iteration
        (ABI 0.5): moments (n_epochs * n_mb, 3) fp64; parts_scratch: n_epochs * n_mb * ceil(minibatch / 64) * 2 fp64"""
        _chk(moments, torch.float64, "moments")
        _chk(parts_scratch, torch.float64, "parts_scratch")
        self._ok(self.lib.catppo_adv_moments_keyed(self.h, _p(advantages), self._dt(advantages), _p(st), int(n_epochs),
                                                   int(total), int(minibatch), _p(parts_scratch), _p(moments),
                                                   self._stream()))

    def adv_stats(self, moments, n_minibatches, stats):
        self._ok(self.lib.catppo_adv_stats(self.h, _p(moments), int(n_minibatches), _p(stats), self._stream()))

    # ------------------------------------------------------------------ GAE
    def gae(self, rewards, values, dones, true_dones, next_value, next_done, next_true_done, gamma, gae_lambda,
            advantages, returns):
        T, N = rewards.shape
        for n, t in (("rewards", rewards), ("values", values), ("dones", dones), ("true_dones", true_dones),
                     ("next_value", next_value), ("next_done", next_done), ("next_true_done", next_true_done),
                     ("advantages", advantages), ("returns", returns)):
            _chk(t, torch.float32, n)
        self._ok(self.lib.catppo_gae_planes(self.h, self.GAE_CLEANRL, GAE_SERIAL, F32, _p(rewards), _p(values), _p(dones),
                                            _p(true_dones), _p(next_value), _p(next_done), _p(next_true_done), f32(gamma),
                                            f32(gamma * gae_lambda), _p(advantages), _p(returns), T, N, self._stream()))

    GAE_CLEANRL, GAE_RL_GAMES, GAE_SKRL = 0, 1, 2

    def gae_f16(self, rewards, values, dones, true_dones, next_value, next_done, next_true_done, gamma, gae_lambda,
                advantages, returns, kind=0):
        """fp16 rollout planes; ``gae_lambda`` is lambda (the product with gamma is formed here, except for skrl)"""
        T, N = rewards.shape
        for n, t in (("rewards", rewards), ("values", values), ("dones", dones), ("next_value", next_value),
                     ("advantages", advantages), ("returns", returns)):
            _chk(t, torch.float16, n)
        gl = gae_lambda if kind == self.GAE_SKRL else gamma * gae_lambda
        self._ok(self.lib.catppo_gae_planes(self.h, int(kind), GAE_SERIAL, F16, _p(rewards), _p(values), _p(dones),
                                            _p(true_dones), _p(next_value), _p(next_done), _p(next_true_done), f32(gamma),
                                            f32(gl), _p(advantages), _p(returns), T, N, self._stream()))

    def gae_rl_games(self, fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma, tau, advantages, returns):
        """rl_games discount_values with float dones ((T,N) planes, time major)"""
        T, N = mb_rewards.shape
        for n, t in (("fdones", fdones), ("last_values", last_values), ("mb_fdones", mb_fdones),
                     ("mb_values", mb_values), ("mb_rewards", mb_rewards), ("advantages", advantages),
                     ("returns", returns)):
            _chk(t, torch.float32, n)
        self._ok(self.lib.catppo_gae_planes(self.h, self.GAE_RL_GAMES, GAE_SERIAL, F32, _p(mb_rewards), _p(mb_values),
                                            _p(mb_fdones), None, _p(last_values), _p(fdones), None, f32(gamma),
                                            f32(gamma * tau), _p(advantages), _p(returns), T, N, self._stream()))

    def gae_skrl(self, rewards, dones, values, last_values, discount_factor, lambda_coefficient, advantages,
                 returns):
        """skrl compute_gae recurrence (raw advantages; normalise with adv_normalize)"""
        T, N = rewards.shape
        for n, t in (("rewards", rewards), ("dones", dones), ("values", values), ("last_values", last_values),
                     ("advantages", advantages), ("returns", returns)):
            _chk(t, torch.float32, n)
        self._ok(self.lib.catppo_gae_planes(self.h, self.GAE_SKRL, GAE_SERIAL, F32, _p(rewards), _p(values), _p(dones), None,
                                            _p(last_values), None, None, f32(discount_factor), f32(lambda_coefficient),
                                            _p(advantages), _p(returns), T, N, self._stream()))

    def adv_normalize(self, advantages, out, stats=None):
        self._ok(self.lib.catppo_adv_normalize(self.h, _p(_chk(advantages, torch.float32, "advantages")),
                                               advantages.numel(), _p(_chk(out, torch.float32, "out")), _p(stats),
                                               self._stream()))

    def value_bootstrap(self, rewards, values, time_outs, gamma):
        _chk(time_outs, torch.uint8, "time_outs")
        self._ok(self.lib.catppo_value_bootstrap(self.h, _p(_chk(rewards, torch.float32, "rewards")),
                                                 _p(_chk(values, torch.float32, "values")), _p(time_outs),
                                                 f32(gamma), rewards.numel(), self._stream()))

    # ------------------------------------------------------------------ rl_games episode bookkeeping
    def rlg_meters_new(self, max_size: int) -> torch.Tensor:
        """device-resident catppo_rlg_meters (three AverageMeters over the last ``max_size`` finished episodes)"""
        m = torch.zeros(C.sizeof(RlgMeters), dtype=torch.uint8, device=self.device)
        self._ok(self.lib.catppo_rlg_meters_init(self.h, _p(m), int(max_size), self._stream()))
        return m

    def rlg_meters_read(self, m: torch.Tensor) -> RlgMeters:
        """host copy (synchronises; logging / tests)"""
        return RlgMeters.from_buffer_copy(bytes(m.cpu().numpy().tobytes()))

    def rlg_episode_step(self, rewards, shaped_rewards, dones, current_rewards, current_shaped_rewards,
                         current_lengths, meters, done_mask=None):
        n = dones.numel()
        v = rewards.numel() // n
        for nm, t in (("rewards", rewards), ("shaped_rewards", shaped_rewards), ("dones", dones),
                      ("current_rewards", current_rewards), ("current_shaped_rewards", current_shaped_rewards),
                      ("current_lengths", current_lengths)):
            _chk(t, torch.float32, nm)
        self._ok(self.lib.catppo_rlg_episode_step(self.h, _p(rewards), _p(shaped_rewards), _p(dones), int(v),
                                                  _p(current_rewards), _p(current_shaped_rewards),
                                                  _p(current_lengths), n, _p(meters), _p(done_mask), self._stream()))

    # ------------------------------------------------------------------ running mean / std
    def rms_update(self, x, n_rows, dim, ldx, mean, var, count):
        self._ok(self.lib.catppo_rms_update_ex(self.h, _p(x), F32, int(n_rows), int(dim), int(ldx), _p(mean), _p(var),
                                               _p(count), self._stream()))

    def rms_moments(self, x, n_rows, dim, ldx, sums):
        self._ok(self.lib.catppo_rms_moments_ex(self.h, _p(x), F32, int(n_rows), int(dim), int(ldx),
                                                _p(_chk(sums, torch.float64, "sums")), self._stream()))

    def rms_merge(self, sums, n_total, dim, mean, var, count):
        self._ok(self.lib.catppo_rms_merge(self.h, _p(sums), float(n_total), int(dim), _p(mean), _p(var), _p(count),
                                           self._stream()))

    def rms_normalize(self, x, n_rows, dim, ldx, mean, var, eps, out, ldo):
        self._ok(self.lib.catppo_rms_normalize_ex(self.h, _p(x), F32, int(n_rows), int(dim), int(ldx), _p(mean), _p(var),
                                                  f32(eps), _p(out), int(ldo), self._stream()))

    # ------------------------------------------------------------------ MLP / PPO
    def mlp_reserve(self, shape: MlpShape, rows: int):
        self.reserve(self.lib.catppo_mlp_workspace_bytes(C.byref(shape), int(rows)))

    def policy_act(self, shape, params, x, n_rows, eps, action, logprob, value, given_action=None):
        self._ok(self.lib.catppo_policy_step(self.h, C.byref(shape), _p(params), _p(x), int(n_rows), _p(eps),
                                             _p(given_action), None, 0, None, _p(action), _p(logprob), _p(value), F32,
                                             self._stream()))

    def value(self, shape, params, x, n_rows, value):
        self._ok(self.lib.catppo_policy_step(self.h, C.byref(shape), _p(params), _p(x), int(n_rows), None, None, None, 0,
                                             None, None, None, _p(value), F32, self._stream()))

    def ppo_minibatch_grad(self, shape, hp: PpoHparams, params, b_obs, b_actions, b_logprobs, b_advantages,
                           b_returns_n, b_values_n, mb_inds, vrms_mean, vrms_var, adv_stats, grad, diag):
        _chk(mb_inds, torch.int64, "mb_inds")
        self._ok(self.lib.catppo_ppo_minibatch_grad(
            self.h, C.byref(shape), C.byref(hp), _p(params), _p(b_obs), _p(b_actions), _p(b_logprobs),
            _p(b_advantages), _p(b_returns_n), _p(b_values_n), _p(mb_inds), mb_inds.numel(), _p(vrms_mean),
            _p(vrms_var), _p(adv_stats), _p(grad), _p(diag), self._stream()))

    GATHER_ROWS = 64     # CATPPO_GATHER_ROWS

    def ppo_gather(self, shape, b_obs, b_actions, b_logprobs, b_advantages, b_returns_n, b_values_n, inds, M, x_g,
                   act_g, scal_g, adv_part_g):
        """one launch: the whole permutation ``inds`` -> packed buffers, minibatch m = contiguous slice m"""
        _chk(inds, torch.int64, "inds")
        _chk(adv_part_g, torch.float64, "adv_part_g")
        self._ok(self.lib.catppo_ppo_gather_ex(
            self.h, C.byref(shape), _p(b_obs), _p(b_actions), _p(b_logprobs), _p(b_advantages), F32, _p(b_returns_n),
            _p(b_values_n), _p(inds), None, 0, inds.numel(), int(M), _p(x_g), _p(act_g), _p(scal_g), _p(adv_part_g),
            None, self._stream()))

    def ppo_minibatch_grad_packed(self, shape, hp: PpoHparams, params, x_mb, act_mb, scal_mb, adv_part_mb, M,
                                  vrms_mean, vrms_var, adv_stats, grad, diag):
        self._ok(self.lib.catppo_ppo_minibatch_grad_packed(
            self.h, C.byref(shape), C.byref(hp), _p(params), _p(x_mb), _p(act_mb), _p(scal_mb), _p(adv_part_mb),
            int(M), _p(vrms_mean), _p(vrms_var), _p(adv_stats), _p(grad), _p(diag), self._stream()))

    def ppo_minibatch_step_packed(self, shape, hp: PpoHparams, params, x_mb, act_mb, scal_mb, adv_part_mb, M,
                                  vrms_mean, vrms_var, adv_stats, grad, diag, exp_avg, exp_avg_sq, max_grad_norm,
                                  beta1, beta2, eps, st):
        """``ppo_minibatch_grad_packed`` + ``clip_adam_dev`` in one call (single process: nothing reduces the gradient
        in between); the fold launches emit the squared norm the clip needs"""
        self._ok(self.lib.catppo_ppo_minibatch_step_packed(
            self.h, C.byref(shape), C.byref(hp), _p(params), _p(x_mb), _p(act_mb), _p(scal_mb), _p(adv_part_mb),
            int(M), _p(vrms_mean), _p(vrms_var), _p(adv_stats), _p(grad), _p(diag), _p(exp_avg), _p(exp_avg_sq),
            f32(max_grad_norm), float(beta1), float(beta2), float(eps), _p(st), self._stream()))

    def clip_adam(self, params, grad, exp_avg, exp_avg_sq, n_flat, max_grad_norm, lr, beta1, beta2, eps, step):
        self._ok(self.lib.catppo_clip_adam(self.h, _p(params), _p(grad), _p(exp_avg), _p(exp_avg_sq), int(n_flat),
                                           f32(max_grad_norm), float(lr), float(beta1), float(beta2), float(eps),
                                           int(step), self._stream()))

    # ------------------------------------------------------------------ ABI 0.2: device iteration state
    def iter_state_new(self, seed: int, lr: float) -> torch.Tensor:
        """device-resident catppo_iter_state as a uint8 tensor (the library's kernels own its contents)"""
        st = torch.zeros(C.sizeof(IterState), dtype=torch.uint8, device=self.device)
        self._ok(self.lib.catppo_iter_init(self.h, _p(st), int(seed) & (2 ** 64 - 1), float(lr), self._stream()))
        return st

    def iter_state_read(self, st: torch.Tensor) -> IterState:
        """host copy (synchronises; tests / logging only)"""
        return IterState.from_buffer_copy(bytes(st.cpu().numpy().tobytes()))

    def iter_begin(self, st, lr0, num_iterations, schedule):
        self._ok(self.lib.catppo_iter_begin(self.h, _p(st), float(lr0), int(num_iterations), int(schedule),
                                            self._stream()))

    def clip_adam_dev(self, params, grad, exp_avg, exp_avg_sq, n_flat, max_grad_norm, beta1, beta2, eps, st):
        self._ok(self.lib.catppo_clip_adam_dev(self.h, _p(params), _p(grad), _p(exp_avg), _p(exp_avg_sq), int(n_flat),
                                               f32(max_grad_norm), float(beta1), float(beta2), float(eps), _p(st),
                                               self._stream()))

    def kl_mean(self, st, diag, kl_out):
        self._ok(self.lib.catppo_kl_mean(self.h, _p(st), _p(diag), _p(kl_out), self._stream()))

    def kl_adaptive_lr(self, st, kl, kl_threshold, kl_factor=2.0, lr_factor=1.5, min_lr=1e-6, max_lr=1e-2):
        self._ok(self.lib.catppo_kl_adaptive_lr(self.h, _p(st), _p(kl), float(kl_threshold), float(kl_factor),
                                                float(lr_factor), float(min_lr), float(max_lr), self._stream()))

    # ------------------------------------------------------------------ on-device randomness / fp16 planes
    @staticmethod
    def _dt(t: torch.Tensor) -> int:
        if t.dtype == torch.float32:
            return F32
        if t.dtype == torch.float16:
            return F16
        if t.dtype == torch.float64:
            return F64
        raise TypeError(f"unsupported element type {t.dtype}")

    def policy_act_rng(self, shape, params, x, n_rows, st, step, action, logprob, value, eps_out=None):
        if st is None:
            raise ValueError("policy_act_rng: the iteration state carries the Philox key")
        self._ok(self.lib.catppo_policy_step(self.h, C.byref(shape), _p(params), _p(x), int(n_rows), None, None, _p(st),
                                             int(step), _p(eps_out), _p(action), _p(logprob), _p(value), self._dt(value),
                                             self._stream()))

    def policy_act_ex(self, shape, params, x, n_rows, eps, action, logprob, value, given_action=None):
        self._ok(self.lib.catppo_policy_step(self.h, C.byref(shape), _p(params), _p(x), int(n_rows), _p(eps),
                                             _p(given_action), None, 0, None, _p(action), _p(logprob), _p(value),
                                             self._dt(value), self._stream()))

    def value_ex(self, shape, params, x, n_rows, value):
        self._ok(self.lib.catppo_policy_step(self.h, C.byref(shape), _p(params), _p(x), int(n_rows), None, None, None, 0,
                                             None, None, None, _p(value), self._dt(value), self._stream()))

    def ppo_gather_ex(self, shape, b_obs, b_actions, b_logprobs, b_advantages, b_returns_n, b_values_n, total, M, x_g,
                      act_g, scal_g, adv_part_g, inds=None, st=None, epoch=0, inds_out=None):
        """one gather launch for a whole epoch; the permutation is ``inds`` (int64 [total]) or, with ``st``, the
        keyed on-device bijection of (seed, iteration, epoch)"""
        if inds is not None:
            _chk(inds, torch.int64, "inds")
        _chk(adv_part_g, torch.float64, "adv_part_g")
        self._ok(self.lib.catppo_ppo_gather_ex(
            self.h, C.byref(shape), _p(b_obs), _p(b_actions), _p(b_logprobs), _p(b_advantages),
            self._dt(b_advantages), _p(b_returns_n), _p(b_values_n), _p(inds), None if inds is not None else _p(st),
            int(epoch), int(total), int(M), _p(x_g), _p(act_g), _p(scal_g), _p(adv_part_g), _p(inds_out),
            self._stream()))

    def rms_moments_ex(self, x, n_rows, dim, ldx, sums):
        self._ok(self.lib.catppo_rms_moments_ex(self.h, _p(x), self._dt(x), int(n_rows), int(dim), int(ldx),
                                                _p(_chk(sums, torch.float64, "sums")), self._stream()))

    def rollout_store_ex(self, reward, dones, time_outs, rewards_t, dones_t1, true_dones_t1):
        self._ok(self.lib.catppo_rollout_store_ex(self.h, _p(reward), _p(dones), _p(time_outs), _p(rewards_t),
                                                  _p(dones_t1), _p(true_dones_t1), self._dt(rewards_t), reward.numel(),
                                                  self._stream()))

    def rms_update_ex(self, x, n_rows, dim, ldx, mean, var, count):
        self._ok(self.lib.catppo_rms_update_ex(self.h, _p(x), self._dt(x), int(n_rows), int(dim), int(ldx), _p(mean),
                                               _p(var), _p(count), self._stream()))

    def rms_normalize_ex(self, x, n_rows, dim, ldx, mean, var, eps, out, ldo):
        self._ok(self.lib.catppo_rms_normalize_ex(self.h, _p(x), self._dt(x), int(n_rows), int(dim), int(ldx),
                                                  _p(mean), _p(var), f32(eps), _p(out), int(ldo), self._stream()))

    def gae_mode(self, mode, rewards, values, dones, true_dones, next_value, next_done, next_true_done, gamma,
                 gae_lambda, advantages, returns):
        T, N = rewards.shape
        for n, t in (("rewards", rewards), ("values", values), ("dones", dones), ("true_dones", true_dones),
                     ("next_value", next_value), ("next_done", next_done), ("next_true_done", next_true_done),
                     ("advantages", advantages), ("returns", returns)):
            _chk(t, torch.float32, n)
        self._ok(self.lib.catppo_gae_planes(self.h, self.GAE_CLEANRL, int(mode), F32, _p(rewards), _p(values), _p(dones),
                                            _p(true_dones), _p(next_value), _p(next_done), _p(next_true_done), f32(gamma),
                                            f32(gamma * gae_lambda), _p(advantages), _p(returns), T, N, self._stream()))

    # ------------------------------------------------------------------ fused rollout step
    def rollout_xchg_new(self, K: int, D: int) -> torch.Tensor:
        return torch.zeros(self._xchg_layout(K, D)[0], dtype=torch.uint8, device=self.device)

    def _xchg_layout(self, K: int, D: int):
        b, o = C.c_uint64(0), C.c_uint64(0)
        self._ok(self.lib.catppo_rollout_xchg_layout(int(K), int(D), C.byref(b), C.byref(o)))
        return int(b.value), int(o.value)

    def rollout_xchg_views(self, xchg: torch.Tensor, K: int, D: int):
        """(colmax fp32 [K], sums fp64 [2D]) views of the exchange buffer (the all-reduce operands when sharded)"""
        off = self._xchg_layout(K, D)[1]
        return xchg[:4 * K].view(torch.float32), xchg[off:off + 16 * max(D, 1)].view(torch.float64)

    def rollout_pre(self, step: RolloutStep):
        self._ok(self.lib.catppo_rollout_pre(self.h, C.byref(step), self._stream()))

    def rollout_post(self, step: RolloutStep):
        self._ok(self.lib.catppo_rollout_post(self.h, C.byref(step), self._stream()))

    def rollout_defer_tail(self, on: bool):
        """ABI 0.5: while on, catppo_rollout_post leaves its one-workgroup tail (running maxima / normaliser state /
        episode log) to the next catppo_rollout_pre launch; ``False`` switches it off AND runs a tail still pending"""
        self._ok(self.lib.catppo_rollout_defer_tail(self.h, int(bool(on)), self._stream()))

    def rollout_flush(self):
        self._ok(self.lib.catppo_rollout_defer_tail(self.h, -1, self._stream()))      # -1: flush, the mode stays

    # ------------------------------------------------------------------ hipGraphs
    def debug_clip_branches(self, buf):
        """test hook: ``buf`` (int32 [2 * M], device) receives the clip branch of every sample of the following gradient calls
        (surrogate codes, then value-loss codes: 0 inside / 1 below / 2 above the clip range); ``None`` switches it off"""
        if buf is not None:
            _chk(buf, torch.int32, "buf")
        self._ok(self.lib.catppo_debug_clip_branches(self.h, _p(buf)))

    def graph_begin(self):
        self._ok(self.lib.catppo_graph_begin(self.h, self._stream()))

    def graph_end(self):
        gid, nn = C.c_int(-1), C.c_int(0)
        self._ok(self.lib.catppo_graph_end(self.h, self._stream(), C.byref(gid), C.byref(nn)))
        return gid.value, nn.value

    def graph_abort(self):
        """drop an active capture without keeping its graph (error path)"""
        self.lib.catppo_graph_end(self.h, self._stream(), None, None)       # graph_id == NULL: abort

    def graph_launch(self, gid: int):
        self._ok(self.lib.catppo_graph_launch(self.h, int(gid), self._stream()))

    def graph_destroy(self, gid: int):
        self._ok(self.lib.catppo_graph_destroy(self.h, int(gid)))

    # ------------------------------------------------------------------ collectives (RCCL under the C ABI)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        rc = self.lib.catppo_comm_unique_id(C.cast(buf, _vp))
        if rc != 0:
            raise RuntimeError(f"catppo_comm_unique_id failed ({rc}): librccl could not be loaded")
        return bytes(buf)

    def comm_probe(self):
        """raises unless librccl can be loaded in this process (no communicator, no bootstrap thread is created)"""
        if self.lib.catppo_comm_probe(None) != 0:
            raise RuntimeError("catppo_comm_probe failed: librccl could not be loaded")

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._ok(self.lib.catppo_comm_init(self.h, int(rank), int(world), C.cast(buf, _vp)))

    @property
    def comm_world(self) -> int:
        return max(int(self.lib.catppo_comm_probe(self.h)), 0)      # world size of the context's communicator, 0 = none

    def comm_destroy(self):
        self._ok(self.lib.catppo_comm_destroy(self.h))

    def set_grad_overlap(self, on: bool) -> bool:
        """gradient all-reduce in per-layer buckets on the library's side stream, under the backward launches
        (ABI 0.4); returns whether the next ``ppo_minibatch_grad*`` call will reduce its own buckets"""
        mode = 2 if on in (2, "tail") else int(bool(on))      # 2 / "tail": ABI 0.5, no extra launch
        self._ok(self.lib.catppo_set_grad_overlap(self.h, mode))
        return bool(self.lib.catppo_set_grad_overlap(self.h, -1))       # -1: query

    @property
    def grad_overlap_active(self) -> bool:
        return bool(self.lib.catppo_set_grad_overlap(self.h, -1))       # -1: query

    def allreduce(self, t: torch.Tensor, op: int = SUM):
        self._ok(self.lib.catppo_allreduce(self.h, _p(_chk(t, t.dtype, "allreduce operand")), t.numel(), self._dt(t),
                                           int(op), self._stream()))

    def allgather(self, send: torch.Tensor, recv: torch.Tensor):
        """recv (uint8, world * send.numel() bytes) <- every rank's send (uint8), rank order"""
        _chk(send, torch.uint8, "allgather send")
        _chk(recv, torch.uint8, "allgather recv")
        self._ok(self.lib.catppo_allgather(self.h, _p(send), _p(recv), send.numel(), self._stream()))

    def broadcast(self, t: torch.Tensor, root: int = 0):
        self._ok(self.lib.catppo_broadcast(self.h, _p(_chk(t, t.dtype, "broadcast operand")), t.numel(), self._dt(t),
                                           int(root), self._stream()))
