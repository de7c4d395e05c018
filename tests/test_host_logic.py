"""Host-side logic that needs no GPU: config surface, term parsing / error behaviour, curriculum,
registry, CLI, checkpoint discovery, C oracle vs goldens."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import streams as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Env:
    num_envs, device = 8, "cpu"
    common_step_counter = 0
    episode_length_buf = torch.ones(8, dtype=torch.long)


def _cfg(**terms):
    from cat_envs.shim import configclass
    ns = {k: v for k, v in terms.items()}
    return configclass(type("Cfg", (), ns))()


def test_constraint_term_cfg_and_manager_parsing():
    from cat_envs.tasks.utils.cat import ConstraintManager, ConstraintTermCfg
    f = lambda env, limit: torch.zeros(env.num_envs)
    cfg = {"a": ConstraintTermCfg(func=f, params={"limit": 1.0}, max_p=0.25),
           "skip": None,
           "b": ConstraintTermCfg(func=f, params={"limit": 2.0}, max_p=1)}
    m = ConstraintManager(cfg, _Env())
    assert m.active_terms == ["a", "b"]
    assert m.get_term_cfg("b").max_p == 1
    with pytest.raises(ValueError, match="not found"):
        m.get_term_cfg("zzz")
    with pytest.raises(ValueError, match="not found"):
        m.set_term_cfg("zzz", cfg["a"])
    new = ConstraintTermCfg(func=f, params={"limit": 1.0}, max_p=0.5)
    m.set_term_cfg("a", new)
    assert m.get_term_cfg("a") is new
    assert "contains 2 active terms" in str(m) and "a" in str(m)
    assert set(m._episode_sums) == {"a", "b"} and m._episode_sums["a"].shape == (8,)
    # same exceptions as the reference's _prepare_terms (constraint_manager.py:248-258)
    with pytest.raises(TypeError, match="is not ConstraintTermCfg"):
        ConstraintManager({"a": object()}, _Env())
    with pytest.raises(TypeError, match="must be float or int"):
        ConstraintManager({"a": ConstraintTermCfg(func=f, params={"limit": 1.0}, max_p="0.1")}, _Env())
    with pytest.raises(ValueError, match="mandatory parameters"):
        ConstraintManager({"a": ConstraintTermCfg(func=f, params={}, max_p=0.1)}, _Env())
    # no GPU: compute() must fail loudly instead of falling back
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.compute()


def test_curriculum_matches_reference_table(golden):
    from cat_envs.tasks.utils.cat import ConstraintManager, ConstraintTermCfg
    from cat_envs.tasks.utils.cat.curriculums import modify_constraint_p
    g = golden("curriculum")
    env = _Env()
    env.constraint_manager = ConstraintManager(
        {"x": ConstraintTermCfg(func=lambda e: torch.zeros(8), params={}, max_p=0.25)}, env)
    for i, init in enumerate(g["inits"]):
        for j, step in enumerate(g["steps"]):
            env.common_step_counter = int(step)
            got = modify_constraint_p(env, None, "x", int(g["num_steps"]), float(init))
            assert got == g["max_p"][i, j]
            assert env.constraint_manager.get_term_cfg("x").max_p == got


def test_task_registry_and_configs():
    import cat_envs.tasks  # noqa: F401
    from cat_envs.shim import load_cfg_from_registry, registry
    assert "Isaac-Velocity-CaT-Flat-Solo12-v0" in registry and "Isaac-Velocity-CaT-Flat-Solo12-Play-v0" in registry
    env_cfg = load_cfg_from_registry("Isaac-Velocity-CaT-Flat-Solo12-v0", "env_cfg_entry_point")
    terms = env_cfg.constraints.__dict__
    # reference ConstraintsCfg: 13 terms (cat_flat_env_cfg.py:259-355) with these max_p
    assert list(terms) == [t[0] for t in S.CAT_TERMS_SOLO12]
    assert [terms[k].max_p for k in terms] == S.CAT_MAXP_SOLO12
    assert terms["joint_torque"].params["limit"] == 3.0 and terms["no_move"].params["joint_vel_limit"] == 4.0
    assert len(env_cfg.curriculum.__dict__) == 8
    assert env_cfg.scene.num_envs == 4096 and env_cfg.decimation == 4 and env_cfg.episode_length_s == 10.0
    a = load_cfg_from_registry("Isaac-Velocity-CaT-Flat-Solo12-v0", "clean_rl_cfg_entry_point")
    # reference clean_rl_ppo_cfg.py:12-34
    assert (a.learning_rate, a.num_steps, a.num_iterations, a.gamma, a.gae_lambda) == (3e-4, 24, 2000, 0.99, 0.95)
    assert (a.updates_epochs, a.minibatch_size, a.clip_coef, a.ent_coef, a.vf_coef) == (5, 16384, 0.2, 0.001, 2.0)
    assert a.max_grad_norm == 1.0 and a.norm_adv and a.clip_vloss and a.anneal_lr and a.save_interval == 50
    assert a.to_dict()["experiment_name"] == "solo12_flat" and a.load_checkpoint == "model_.*.pt"
    play = load_cfg_from_registry("Isaac-Velocity-CaT-Flat-Solo12-Play-v0", "env_cfg_entry_point")
    assert play.scene.num_envs == 50


def test_scene_entity_resolution():
    from cat_envs.shim import SceneEntityCfg
    from cat_envs.tasks.utils.cat.cat_env import SOLO12_BODIES, SOLO12_JOINTS
    import types
    scene = {"robot": types.SimpleNamespace(joint_names=SOLO12_JOINTS, body_names=SOLO12_BODIES)}
    c = SceneEntityCfg("robot", joint_names=[".*_HAA", ".*_HFE", ".*_KFE"])
    c.resolve(scene)
    assert c.joint_ids == slice(None)
    c = SceneEntityCfg("robot", joint_names=["FL_HFE", "FR_HFE"])
    c.resolve(scene)
    assert c.joint_ids == [1, 4]
    c = SceneEntityCfg("robot", body_names=["base_link", ".*_UPPER_LEG"])
    c.resolve(scene)
    assert c.body_ids == [0, 2, 6, 10, 14]
    c = SceneEntityCfg("robot", body_names=".*_FOOT")
    c.resolve(scene)
    assert c.body_ids == [4, 8, 12, 16]
    with pytest.raises(ValueError):
        SceneEntityCfg("robot", joint_names=["nope"]).resolve(scene)


def test_cli_surface(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts", "clean_rl"))
    import importlib
    train = importlib.import_module("train")
    play = importlib.import_module("play")
    p = train.build_parser()
    a, rest = p.parse_known_args(["--task", "T", "--num_envs", "64", "--seed", "7", "--num_iterations", "3",
                                  "--experiment_name", "e", "--logger", "tensorboard", "--headless",
                                  "--load_run", "r", "--checkpoint", "c.pt", "--log_project_name", "p",
                                  "agent.minibatch_size=512", "--video_length", "10", "--video_interval", "5"])
    assert (a.task, a.num_envs, a.seed, a.num_iterations, a.headless) == ("T", 64, 7, 3, True)
    assert rest == ["agent.minibatch_size=512"]
    import cli_args
    import cat_envs.tasks  # noqa: F401
    from cat_envs.shim import load_cfg_from_registry
    cfg = cli_args.update_clean_rl_cfg(load_cfg_from_registry("Isaac-Velocity-CaT-Flat-Solo12-v0",
                                                              "clean_rl_cfg_entry_point"), a)
    assert (cfg.seed, cfg.load_run, cfg.load_checkpoint, cfg.logger) == (7, "r", "c.pt", "tensorboard")
    train.apply_overrides({"agent": cfg}, rest)
    assert cfg.minibatch_size == 512
    # the reference's precedence (train.py:57-61,92-107): the key=value overrides are resolved by the hydra_task_config
    # decorator BEFORE main's body, so an explicit flag wins over an override of the same field; a typo is an error
    from cat_envs.shim import hydra_task_config
    argv0 = sys.argv
    try:
        sys.argv = ["train.py", "env.scene.num_envs=128", "agent.hidden=[64,64]", "agent.learning_rate=1e-3"]

        @hydra_task_config("Isaac-Velocity-CaT-Flat-Solo12-v0", "clean_rl_cfg_entry_point")
        def body(env_cfg, agent_cfg, flag_num_envs):
            seen = (env_cfg.scene.num_envs, list(agent_cfg.hidden), agent_cfg.learning_rate)
            env_cfg.scene.num_envs = flag_num_envs if flag_num_envs is not None else env_cfg.scene.num_envs
            return seen, env_cfg.scene.num_envs
        assert body(64) == ((128, [64, 64], 1e-3), 64)
        assert body(None) == ((128, [64, 64], 1e-3), 128)
        sys.argv = ["train.py", "agent.minibatch_sise=512"]
        with pytest.raises(AttributeError, match="minibatch_sise"):
            body(None)
        sys.argv = ["train.py", "agnt.minibatch_size=512"]
        with pytest.raises(KeyError, match="agnt"):
            body(None)
        # a stray token (the value of an unknown `--flag value` pair left by parse_known_args) is reported, not fatal
        sys.argv = ["train.py", "--some_unknown_flag", "7", "agent.minibatch_size=256"]
        body(None)           # must not raise
    finally:
        sys.argv = argv0
    # checkpoint discovery: latest run, highest iteration (reference naming model_<it>.pt)
    for run, files in (("2026-01-01_00-00-00", ["model_49.pt", "model_99.pt"]), ("2026-02-01_00-00-00", ["model_49.pt", "model_149.pt", "model_99.pt"])):
        os.makedirs(tmp_path / run)
        for f in files:
            (tmp_path / run / f).write_bytes(b"")
    assert play.get_checkpoint_path(str(tmp_path), ".*", "model_.*.pt").endswith("2026-02-01_00-00-00/model_149.pt")


def _c_oracle():
    import __graft_entry__ as ge
    ge._build_module().build_oracle_c()
    return C.CDLL(os.path.join(ROOT, "oracle", "liboracle_c.so"))


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("tag", ["small", "minp", "solo64"])
def test_c_oracle_cat_step_vs_reference_golden(golden, tag):
    lib = _c_oracle()
    g = golden(f"cat_{tag}")
    n, steps = int(g["n_envs"]), int(g["steps"])
    terms = list(zip([str(x) for x in g["term_names"]], [int(w) for w in g["term_widths"]], [str(k) for k in g["term_kinds"]]))
    stream = S.cat_stream(int(g["seed"]), n, terms, steps)
    K, nt = sum(w for _, w, _ in terms), len(terms)
    off = np.concatenate([[0], np.cumsum([w for _, w, _ in terms])]).astype(np.int32)
    rm, prob = np.zeros(K, np.float32), np.zeros(n, np.float32)
    viol, eprob = np.zeros((nt, n), np.float32), np.zeros((nt, n), np.float32)
    tau, min_p = float(g["tau"]), float(g["min_p"])
    reset_at = set(int(x) for x in g["reset_at"])
    f32 = lambda x: C.c_float(x)
    for t in range(steps):
        cstr = np.ascontiguousarray(np.concatenate(
            [np.asarray(stream[t][nm]).astype(np.float32).reshape(n, -1) for nm, _, _ in terms], 1))
        dp = np.array([C.c_float(p - min_p).value for p in g["max_p"][t]], np.float32)
        lib.cat_step_oracle(_fp(cstr), C.c_int64(n), K, _fp(off), nt, _fp(dp), f32(min_p), f32(tau), f32(1.0 - tau),
                            int(t == 0), _fp(rm), None, None, _fp(prob), None, _fp(viol), _fp(eprob), None)
        np.testing.assert_array_equal(prob, g["cstr_prob"][t])
        np.testing.assert_array_equal(rm, g["running_maxes"][t])
        if t in reset_at:
            ids = g[f"reset{t}_ids"]
            viol[:, ids] = 0
            eprob[:, ids] = 0
    np.testing.assert_array_equal(viol, g["episode_sums"])
    np.testing.assert_array_equal(eprob, g["cstr_mean_values"])


def test_c_oracle_gae_vs_reference_ppo_run(golden):
    lib = _c_oracle()
    g = golden("ppo_64x24")
    N, T = 64, 24
    s = S.env_stream(int(g["seed"]), T * 3, N, 45)
    dones = np.ascontiguousarray(np.concatenate([np.zeros((1, N), np.float32), s["dones"][:T - 1]]))
    tdones = np.ascontiguousarray(np.concatenate([np.zeros((1, N), np.float32), s["timeouts"][:T - 1].astype(np.float32)]))
    rew, val = np.ascontiguousarray(s["reward"][:T]), np.ascontiguousarray(g["it0_values"])
    nv = np.ascontiguousarray(g["it0_next_value"])
    nd, ntd = np.ascontiguousarray(s["dones"][T - 1]), np.ascontiguousarray(s["timeouts"][T - 1].astype(np.float32))
    adv, ret = np.zeros((T, N), np.float32), np.zeros((T, N), np.float32)
    lib.gae_oracle(_fp(rew), _fp(val), _fp(dones), _fp(tdones), _fp(nv), _fp(nd), _fp(ntd), C.c_float(0.99),
                   C.c_float(0.99 * 0.95), _fp(adv), _fp(ret), T, C.c_int64(N))
    np.testing.assert_array_equal(adv, g["it0_advantages"])
    np.testing.assert_array_equal(ret, g["it0_returns"])


def test_term_descriptor_rows_are_cached_only_when_they_alias_the_simulator_buffers():
    """a descriptor row holds raw pointers: it may be reused across steps only if describe() did not have to copy
    (dtype / layout conversion) any of its inputs"""
    import torch
    from cat_envs.tasks.utils.cat.constraints import TermDescription
    x = torch.zeros(8, 12)
    assert TermDescription(0, 12, list(range(12)), limit=1.0, x=x).cacheable
    assert TermDescription(0, 4, [0, 1, 2, 3], limit=1.0, x=x[:, 2:9]).cacheable            # a column view is fine
    assert not TermDescription(0, 12, list(range(12)), limit=1.0, x=x.double()).cacheable    # converted -> copied
    assert not TermDescription(0, 8, list(range(8)), limit=1.0, x=torch.zeros(12, 8).t()).cacheable   # transposed
    y = torch.zeros(8, 12)
    assert TermDescription(1, 12, list(range(12)), limit=1.0, x=x, y=y).cacheable
    assert not TermDescription(1, 12, list(range(12)), limit=1.0, x=x, y=y.half()).cacheable


def test_update_from_moments_matches_the_reference_running_mean_std(golden):
    """module-level update_mean_var_count_from_moments / RunningMeanStd.update_from_moments (reference ppo.py:33-62)
    fed torch batch moments reproduce the states the reference's RunningMeanStd went through (tests/golden/rms.npz)"""
    import numpy as np
    import torch
    from cat_envs.tasks.utils.cleanrl.ppo import RunningMeanStd, update_mean_var_count_from_moments
    g = golden("rms")
    rs = np.random.RandomState(int(g["seed"]))
    xs = (rs.standard_normal((30, 64, 45)) * rs.uniform(0.1, 5, 45) + rs.uniform(-2, 2, 45)).astype(np.float32)
    rms = RunningMeanStd(shape=(45,), device="cpu")
    m, v, c = torch.zeros(45), torch.ones(45), torch.ones(())
    for i in range(30):
        x = torch.from_numpy(xs[i])
        bm, bv = x.mean(0), x.var(0, correction=0)
        rms.update_from_moments(bm, bv, x.shape[0])
        m, v, c = update_mean_var_count_from_moments(m, v, c, bm, bv, x.shape[0])
        st = np.concatenate([rms.running_mean.numpy(), rms.running_var.numpy(), rms.count.numpy().reshape(1)])
        np.testing.assert_allclose(st, g["vec_state"][i], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(m.numpy(), rms.running_mean.numpy())
    np.testing.assert_array_equal(v.numpy(), rms.running_var.numpy())


def test_index_division_magic_is_exact_over_the_whole_index_range(tmp_path):
    """`catppo_div_magic` / `fast_div` (csrc/common.h, round 5): the env-step kernels divide small indices by run-time widths
    with one multiply-high; exact only because e * d < 2^32 in every use (e < 2^17 rows x columns, d <= 4096 columns).
    Checked exhaustively with the library's OWN host function: a host program built from common.h (hipcc compiles host code
    without a GPU), `(e * magic) >> 32` being what `__umulhi` computes on the device."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = tmp_path / "div_magic.hip"
    src.write_text(r"""
#include "common.h"
#include <cstdio>
int main() {
  long bad = 0;
  for (uint32_t d = 1; d <= 4096; ++d) {
    const uint32_t m = catppo_div_magic(d);
    for (uint32_t e = 0; e < (1u << 17); ++e) {
      const uint32_t q = m ? (uint32_t)(((uint64_t)e * m) >> 32) : e;
      bad += q != e / d;
    }
  }
  printf("bad %ld\n", bad);
  return bad != 0;
}
""")
    exe = tmp_path / "div_magic"
    csrc = os.path.join(ROOT, "constraints-as-terminations_amd", "csrc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-I{csrc}", "-o", str(exe), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "bad 0", r.stdout + r.stderr


@pytest.mark.parametrize("world", [2, 8, 22, 64])
def test_native_comm_known_answer_pass_holds_for_any_world_size(world):
    """`cat_envs.parallel._selfcheck` (the known-answer pass every rank runs on a fresh RCCL communicator before the data
    path may use it): its closed-form answers must hold for ANY world size - ADVICE r4: the fp64 tolerance once failed from
    22 ranks on, which would have dropped every multi-node job to the fallback transport.  `world` ranks as threads of this
    process around an in-memory communicator with RCCL's semantics (SUM in the operand's dtype, MAX, broadcast, all-gather)."""
    import threading
    from cat_envs import parallel

    slots, lock, bar = {}, threading.Lock(), threading.Barrier(world)

    class FakeNat:
        device = torch.device("cpu")

        def __init__(self, r):
            self.r, self.n = r, 0

        def _exchange(self, t):
            k = self.n
            self.n += 1
            with lock:
                slots.setdefault(k, {})[self.r] = t.clone()
            bar.wait()
            out = [slots[k][q] for q in range(world)]
            bar.wait()
            return out

        def allreduce(self, t, op):
            parts = self._exchange(t)
            acc = parts[0].clone()
            for q in parts[1:]:
                acc = torch.maximum(acc, q) if op == 1 else acc + q
            t.copy_(acc)

        def broadcast(self, t, src):
            t.copy_(self._exchange(t)[src])

        def allgather(self, send, recv):
            recv.copy_(torch.cat(self._exchange(send)))

    res = [None] * world

    def rank_main(r):
        res[r] = parallel._selfcheck(FakeNat(r), r, world)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert res == [None] * world, res
    # and a wrong collective is caught: the same pass around a communicator whose SUM drops the last rank
    slots.clear()

    class LossyNat(FakeNat):
        def allreduce(self, t, op):
            parts = self._exchange(t)
            acc = parts[0].clone()
            for q in parts[1:-1] if op == 0 else parts[1:]:
                acc = torch.maximum(acc, q) if op == 1 else acc + q
            t.copy_(acc)

    def lossy_main(r):
        res[r] = parallel._selfcheck(LossyNat(r), r, world)

    th = [threading.Thread(target=lossy_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert all(isinstance(x, str) and "SUM fp32" in x and "SUM fp64" in x for x in res), res


def test_branch_flip_machinery_on_synthetic_traces():
    """The analysis the whole-iteration GPU tests lean on when parameters leave the tight bar (smoke_impl.first_parting_step /
    analyse_branches, cleanrl/ppo.py:320-341), on hand-made data: the parting step is where the distance LEAVES its
    rounding-level plateau (not the first step above the bar), a flipped sample is one whose device clip code differs from the
    code of the oracle's own ratio / value difference, and its margin is the oracle's distance to the boundary."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smoke_impl
    bar = 1.2e-5
    assert smoke_impl.first_parting_step([7e-8] * 5 + [9e-7, 1.6e-6, 2.4e-6, 1.3e-5], bar) == 5      # two-rank cfg3 shape
    assert smoke_impl.first_parting_step([6e-8] * 9 + [1.0e-5, 2.0e-5], bar) == 9                     # cfg4: one jump
    assert smoke_impl.first_parting_step([1e-8] * 4, bar) is None
    assert smoke_impl.first_parting_step([5e-8, 2e-5], bar) == 1
    assert smoke_impl.first_parting_step([3e-5, 4e-5], bar) == 0                                      # apart from the first step
    clip = 0.2
    # four samples: inside, just above 1 + clip (oracle), below 1 - clip, inside; the device has sample 1 INSIDE
    ratio_o = torch.tensor([1.0, 1.2 + 3e-7, 0.7, 1.1])
    ratio_d = torch.tensor([1.0, 1.2 - 2e-7, 0.7, 1.1])
    new_o, old_o, ret_o = torch.tensor([0.0, 0.5, -0.3, 0.1]), torch.zeros(4), torch.tensor([0.1, 0.2, 0.0, 0.0])
    dl_d = new_o - old_o
    code = lambda v, c: ((v < c - clip).long() + 2 * (v > c + clip).long())
    e1, e2 = new_o - ret_o, old_o + (new_o - old_o).clamp(-clip, clip) - ret_o
    vmax = ((e1 * e1 > e2 * e2).long() + 2 * (e1 * e1 < e2 * e2).long()) << 2
    codes = torch.cat([code(ratio_d, 1.0), code(dl_d, 0.0) | vmax])
    step = dict(ratio=ratio_o, newvalue_n=new_o, old_values_n=old_o, returns_n=ret_o)
    rep = smoke_impl.analyse_branches(codes, ratio_d, dl_d, step, clip, True)
    assert rep["flipped_surrogate"] == 1 and rep["flipped_value"] == 0, rep
    assert abs(rep["max_margin_of_flipped"] - abs(abs(float(ratio_o[1]) - 1.0) - clip)) < 2e-8 and rep["max_margin_of_flipped"] < 1e-6
    assert abs(rep["max_device_oracle_ratio_diff"] - abs(float(ratio_o[1] - ratio_d[1]))) < 2e-8
    # no disagreement -> nothing flipped, no margin
    rep0 = smoke_impl.analyse_branches(torch.cat([code(ratio_o, 1.0), code(dl_d, 0.0) | vmax]), ratio_o, dl_d, step, clip, True)
    assert rep0["flipped_surrogate"] == 0 and rep0["flipped_value"] == 0 and rep0["max_margin_of_flipped"] is None
