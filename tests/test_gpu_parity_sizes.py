"""GPU parity at the BASELINE.json configurations AS WRITTEN (whole iterations against the CPU oracle on the same
synthetic stream, action noise and minibatch permutations) and ``ConstraintManager.reset`` / ``catppo_cat_reset``
against the reference's own ``reset()`` outputs (tests/golden/cat_*.npz ``reset{t}_vals``).

Reference: cat/constraint_manager.py:190-211 (reset statistics), cleanrl/ppo.py:251-354 (iteration)."""
import ctypes as C

import numpy as np
import pytest
import torch

import streams as S
from oracle import cat_oracle as CO

pytestmark = pytest.mark.gpu


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ------------------------------------------------------------------------------------------ whole iterations
# Bars of the whole-iteration comparisons.  north_star: "returns/advantages within 1e-5 fp32".  The GAE KERNEL meets
# that bit-exactly on identical inputs (test_gae_bit_exact, test_gae_vs_reference_ppo_run); at ITERATION level GAE's
# inputs are the device's own values (three fp32 GEMM layers in MFMA order vs the oracle's BLAS order) and the first
# recorded run of round 3 (profiles/r3_parity.json: every config-size test writes its achieved errors there) showed
# that the whole iteration stays inside 1e-5 too: advantages <= 4.2e-6, returns <= 2.9e-6, values <= 3.8e-6,
# log-probs <= 7.6e-6, parameters after all optimiser steps <= 5.8e-6 (cfg2, 30 steps) / <= 1e-7 (others).
# Each bar is <= 2x the recorded worst case (round 2 used 1e-4 / 4e-4 and never recorded what it achieved).
BARS = {
    "default": dict(values=8e-6, logprobs=1.6e-5, advantages=1e-5, returns=1e-5, params=1.2e-5, actions=1e-5),
}
# The PARAMETER bar after tens of optimiser steps has a precondition.  The gradients of the clipped surrogate and of the
# clipped value loss (cleanrl/ppo.py:320-341) are DISCONTINUOUS where a sample sits exactly on a clip boundary
# (|ratio - 1| = clip, |newvalue - old value| = clip).  Device and oracle agree on every per-sample quantity to ~1e-6; a
# sample closer to a boundary than that takes different branches on the two sides, its whole gradient contribution flips
# (one sample is ~1 % of the NET critic gradient of a 16384-row minibatch: the per-sample terms largely cancel), and from
# that optimiser step on the two trajectories differ by 1e-5 .. 1e-4 instead of 1e-7.  Round 5 met this at cfg4 (step 9
# of 24, profiles/r5_cfg4_branch_flip.txt) when the head products of fwd_head_kernel moved to another MFMA shape - a
# 1e-7 change of the value head, bit-reproducible, 3e-8 away from the old kernel when no sample sits on a boundary.
# Round 6: the looser bar is no longer granted on the oracle's say-so ("some sample came
# close").  Both sides record their parameters after EVERY optimiser step; when the tight bar fails the test finds the
# first step after which they differ by more than it and RE-RUNS that step's gradient call on the device with
# catppo_debug_clip_branches on - the loss kernel itself exports the clip branch of every sample (smoke_impl.
# branch_flip_report) - and demands
#   (1) until that step the parameters agreed within the tight bar,
#   (2) at least one sample of that minibatch sits on different sides of a clip boundary on device and oracle,
#   (3) every such sample is closer to the boundary (in the oracle's arithmetic) than device and oracle disagree about
#       that per-sample quantity at that step (measured: catppo_policy_step under the same parameters), and that
#       disagreement is itself at rounding level (PER_SAMPLE_DISAGREEMENT_CAP).
# Only then may the parameters use the bar of rounds 1-2; the record names the step, the flipped samples, both numbers.
# What it shows at cfg4 (profiles/r6_parity.json): 4.5e-8 for nine optimiser steps, then 2e-5 in ONE step whose
# minibatch holds one sample with |newvalue - old value| within rounding of clip_coef - round 5's finding, now asserted.
PER_SAMPLE_DISAGREEMENT_CAP = 2e-5
PARAMS_BAR_AFTER_A_BRANCH_FLIP = 4e-4
BOUNDARY_NOISE = 2e-6          # device / oracle disagreement of ratio and value when the parameters are equal (machinery test)


def _iteration(name=None, bars=None, **kw):
    import parity_record
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(trace=True, **kw)
    rep = smoke_impl.compare(trainer, orc, outs[-1], check=False)
    print(name, kw, rep)
    b = dict(BARS["default"], **(BARS.get(name) or {}), **(bars or {}))
    assert rep["rewards"] == 0.0 and rep["dones"] == 0.0, rep          # termination masks are bit-exact
    margin = float(getattr(orc, "clip_boundary_margin", float("inf")))
    flip = None
    if rep["params"] >= b["params"]:
        flip = smoke_impl.branch_flip_report(trainer, orc, b["params"])
        flip_rec = {k: v for k, v in flip.items() if k != "errs"}
        print(f"{name}: parameters left the {b['params']:.1e} bar - per-step analysis: {flip_rec}")
        assert flip["first_step"] is not None, ("the per-step traces do not show the divergence the final parameters do", rep)
        assert flip["err_before"] < b["params"], flip_rec                                     # (1)
        n_flipped = flip["flipped_surrogate"] + flip["flipped_value"]
        assert n_flipped >= 1, ("parameters diverged without a clip-branch disagreement: not a boundary effect", flip_rec)   # (2)
        disagreement = max(flip["max_device_oracle_ratio_diff"] if flip["flipped_surrogate"] else 0.0,
                           flip["max_device_oracle_value_diff"] if flip["flipped_value"] else 0.0)
        assert flip["max_margin_of_flipped"] <= disagreement < PER_SAMPLE_DISAGREEMENT_CAP, flip_rec              # (3)
        flip["errs_up_to_the_step"] = [float(f"{e:.3e}") for e in flip["errs"][:flip["first_step"] + 2]]
        b["params"] = PARAMS_BAR_AFTER_A_BRANCH_FLIP
    if name:
        extra = {} if flip is None else {f"branch_flip.{k}": v for k, v in flip.items() if k != "errs"}
        parity_record.record(name, dict(rep, clip_boundary_margin=margin, params_bar=b["params"], **extra),
                             sizes={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()
                                    if k != "agent_overrides"}, seed=kw.get("seed", 42))
    for k, bar in b.items():
        assert rep[k] < bar, (k, rep[k], bar, rep, margin)
    return trainer, orc, outs, rep


def test_cfg1_exact_64x24_two_terms_48d():
    """BASELINE configs[0]: 64 envs x 24, Solo12 48-d obs, 2 ConstraintTerms (C3 soft + C7 hard), reference MLP"""
    trainer, orc, outs, _ = _iteration("cfg1_64x24_two_terms", num_envs=64, num_steps=24, minibatch=512, epochs=5, iters=2, six_terms="two",
                                       obs_dim=48)
    cm = trainer.envs.constraint_manager
    assert cm.active_terms == ["joint_torque", "contact"] and cm.cat._p_cstr.shape == (64, 13)
    assert trainer.D == 48 and trainer.Dp == 48


def test_cfg2_exact_4096x24_six_terms_3x256_full_update():
    """BASELINE configs[1], the metric's configuration, at its own size: 4096 envs x 24, 48-d obs, 6 terms /
    42 columns, 3x256 MLPs, 5 epochs x 6 minibatches of 16384 (= 30 optimiser steps)."""
    trainer, orc, outs, rep = _iteration("cfg2_4096x24_six_terms_3x256", num_envs=4096, num_steps=24, minibatch=16384,
                                         epochs=5, iters=1, hidden=(256, 256, 256), six_terms=True, obs_dim=48)
    assert trainer.adam_step == 30 and trainer.batch == 98304 and trainer.mb == 16384
    cm = trainer.envs.constraint_manager
    assert len(cm.active_terms) == 6 and cm.cat._p_cstr.shape == (4096, 42)
    np.testing.assert_array_equal(cm.cat.get_running_maxes().cpu().numpy()[0], orc.env.mgr.cat.get_running_maxes()[0])
    for name in cm.active_terms:       # per-term episode statistics after 24 steps with resets: bit-exact
        np.testing.assert_array_equal(cm._episode_sums[name].cpu().numpy(), orc.env.mgr.episode_sums[name])
        np.testing.assert_array_equal(cm._cstr_mean_values[name].cpu().numpy(), orc.env.mgr.cstr_mean_values[name])


def test_cfg4_exact_4096x48_235d_full_update():
    """BASELINE configs[3] at full size: 4096 envs x 48, 235-d observations (48 + 187 height scan; the first layer
    is 240 wide after padding), 3x256 MLPs, minibatches of 16384 (12 per epoch)."""
    trainer, orc, outs, rep = _iteration("cfg4_4096x48_235d", num_envs=4096, num_steps=48, minibatch=16384, epochs=2,
                                         iters=1, hidden=(256, 256, 256), six_terms=True, obs_dim=235)
    assert trainer.Dp == 240 and trainer.adam_step == 24 and trainer.batch == 196608


def test_cfg3_single_process_16384x24_full_constraints_reference_mlp():
    """BASELINE configs[2] as ONE process (the job the 8 ranks shard): 16384 envs x 24, full 13-term ConstraintsCfg,
    reference MLP 512/256/128, 16384-row minibatches (24 per epoch; 2 epochs here = 48 optimiser steps - the CPU
    oracle needs ~1 s per step at this size).  The sharded form of the same job is test_gpu_two_rank_trainer.py."""
    trainer, orc, outs, rep = _iteration("cfg3_single_process_16384x24_13terms", num_envs=16384, num_steps=24,
                                         minibatch=16384, epochs=2, iters=1, hidden=(512, 256, 128), six_terms=False,
                                         obs_dim=45)
    assert trainer.adam_step == 48 and trainer.batch == 393216 and trainer.n_mb == 24
    cm = trainer.envs.constraint_manager
    assert len(cm.active_terms) == 13 and cm.cat._p_cstr.shape == (16384, 78)
    np.testing.assert_array_equal(cm.cat.get_running_maxes().cpu().numpy()[0], orc.env.mgr.cat.get_running_maxes()[0])


# ------------------------------------------------------------------------------------------ reset statistics
def _pack(step, terms):
    cols = []
    for name, width, kind in terms:
        v = np.asarray(step[name]).astype(np.float32)
        cols.append(v.reshape(v.shape[0], -1))
    return np.concatenate(cols, axis=1)


class _StreamEnv:
    """the env surface ConstraintManager touches: num_envs, device, episode_length_buf; terms read ``cursor``"""

    def __init__(self, n, stream):
        self.num_envs, self.device = n, torch.device("cuda", torch.cuda.current_device())
        self.episode_length_buf = torch.zeros(n, dtype=torch.long, device=self.device)
        self.common_step_counter = 0
        self.stream = [{k: torch.from_numpy(np.asarray(v)).to(self.device) for k, v in s.items()} for s in stream]
        self.cursor = 0


@pytest.mark.parametrize("tag,how", [("small", "index"), ("small", "mask"), ("solo64", "index"), ("solo64", "list")])
def test_manager_reset_vs_reference_golden(golden, tag, how):
    """ConstraintManager.compute() / reset(env_ids) driven exactly like the golden generator drove the reference's
    manager; the returned Episode_Constraint_{violation,probability}/* values must match the reference's."""
    from cat_envs.tasks.utils.cat import ConstraintManager, ConstraintTermCfg
    g = golden(f"cat_{tag}")
    n, steps = int(g["n_envs"]), int(g["steps"])
    terms = list(zip([str(x) for x in g["term_names"]], [int(w) for w in g["term_widths"]],
                     [str(k) for k in g["term_kinds"]]))
    env = _StreamEnv(n, S.cat_stream(int(g["seed"]), n, terms, steps))

    def make_term(name):
        return lambda e: e.stream[e.cursor][name]

    cfg = {name: ConstraintTermCfg(func=make_term(name), params={}, max_p=float(mp))
           for (name, _, _), mp in zip(terms, g["init_max_p"])}
    mgr = ConstraintManager(cfg, env, tau=float(g["tau"]), min_p=float(g["min_p"]))
    names = [t[0] for t in terms]
    reset_at = set(int(x) for x in g["reset_at"])
    assert reset_at, "fixture without resets"
    n_checked = 0
    for t in range(steps):
        env.cursor = t
        env.episode_length_buf += 1
        for name, mp in zip(names, g["max_p"][t]):       # the curriculum's max_p of that step
            c = mgr.get_term_cfg(name)
            c.max_p = float(mp)
            mgr.set_term_cfg(name, c)
        p = mgr.compute()
        np.testing.assert_array_equal(p.cpu().numpy()[::int(g["sub"])], g["cstr_prob"][t])
        if t in reset_at:
            ids = np.asarray(g[f"reset{t}_ids"])
            if how == "index":
                sel = torch.from_numpy(ids).cuda()
            elif how == "list":
                sel = [int(i) for i in ids]
            else:
                m = np.zeros(n, bool)
                m[ids] = True
                sel = torch.from_numpy(m).cuda()
            ex = mgr.reset(sel)
            env.episode_length_buf[torch.from_numpy(ids).cuda()] = 0
            keys = [str(k) for k in g[f"reset{t}_keys"]]
            assert sorted(ex) == keys
            got = np.array([float(ex[k]) for k in keys])
            np.testing.assert_allclose(got, g[f"reset{t}_vals"], rtol=1e-5, atol=1e-7)
            # the accumulators of the reset envs are zero afterwards, the others untouched
            for i, name in enumerate(names):
                assert float(mgr._episode_sums[name][torch.from_numpy(ids).cuda()].abs().max()) == 0.0
            n_checked += 1
    assert n_checked == len(reset_at)
    np.testing.assert_array_equal(torch.stack([mgr._episode_sums[k] for k in names]).cpu().numpy()[:, ::int(g["sub"])],
                                  g["episode_sums"])
    np.testing.assert_array_equal(
        torch.stack([mgr._cstr_mean_values[k] for k in names]).cpu().numpy()[:, ::int(g["sub"])], g["cstr_mean_values"])


def test_manager_reset_edge_cases():
    """env_ids=None (all envs), first reset with episode_length == 0 (0/0 -> NaN like the reference's first log,
    constraint_manager.py:196-198), nobody selected (the previous log values are kept), ragged N, and the ring of
    returned 0-d tensors staying valid."""
    from cat_envs.tasks.utils.cat import ConstraintManager, ConstraintTermCfg
    n = 1000
    terms = S.CAT_TERMS_SMALL
    stream = S.cat_stream(5, n, terms, 6)
    env = _StreamEnv(n, stream)
    cfg = {name: ConstraintTermCfg(func=(lambda nm: lambda e: e.stream[e.cursor][nm])(name), params={}, max_p=0.5)
           for name, _, _ in terms}
    mgr = ConstraintManager(cfg, env)
    names = [t[0] for t in terms]
    orc = CO.ConstraintManagerOracle(names, n)
    # (1) reset before any step: lengths are 0, sums are 0 -> NaN means, exactly like the reference
    ex0 = mgr.reset(None)
    assert all(np.isnan(float(v)) for v in ex0.values()) and len(ex0) == 2 * len(names)
    ep_len = np.zeros(n, np.int64)
    rs = np.random.RandomState(0)
    for t in range(6):
        env.cursor = t
        env.episode_length_buf += 1
        ep_len += 1
        mgr.compute()
        orc.compute(stream[t], {nm: 0.5 for nm in names})
        if t == 1:      # (2) nobody selected: the log keeps the previous values (here: the NaNs of ex0)
            ex = mgr.reset(torch.zeros(n, dtype=torch.bool, device="cuda"))
            assert all(np.isnan(float(v)) for v in ex.values())
            for nm in names:
                np.testing.assert_array_equal(mgr._episode_sums[nm].cpu().numpy(), orc.episode_sums[nm])
        if t == 3:      # (3) a subset, some of whose lengths differ
            ids = np.nonzero(rs.rand(n) < 0.4)[0]
            ex = mgr.reset(torch.from_numpy(ids).cuda())
            exp = orc.reset(ids, ep_len)
            for k, v in exp.items():
                np.testing.assert_allclose(float(ex[k]), float(v), rtol=1e-5, atol=1e-7)
            env.episode_length_buf[torch.from_numpy(ids).cuda()] = 0
            ep_len[ids] = 0
            kept = ex
        if t == 4:      # (4) nobody selected again: previous (finite) values survive, earlier dict still valid
            ex = mgr.reset(torch.zeros(n, dtype=torch.bool, device="cuda"))
            for k in kept:
                assert float(ex[k]) == float(kept[k])
    # (5) env_ids=None = every env
    ex = mgr.reset(None)
    exp = orc.reset(None, ep_len)
    for k, v in exp.items():
        np.testing.assert_allclose(float(ex[k]), float(v), rtol=1e-5, atol=1e-7)
    for nm in names:
        assert float(mgr._episode_sums[nm].abs().max()) == 0.0 and float(mgr._cstr_mean_values[nm].abs().max()) == 0.0


def test_cat_reset_c_abi_direct(golden):
    """catppo_cat_reset through the C ABI on the solo64 golden stream (mask form), incl. the `prev` hand-over."""
    from cat_envs import native
    nat = native.Native()
    g = golden("cat_solo64")
    n, steps = int(g["n_envs"]), int(g["steps"])
    terms = list(zip([str(x) for x in g["term_names"]], [int(w) for w in g["term_widths"]],
                     [str(k) for k in g["term_kinds"]]))
    stream = S.cat_stream(int(g["seed"]), n, terms, steps)
    K, nt = sum(w for _, w, _ in terms), len(terms)
    off = np.concatenate([[0], np.cumsum([w for _, w, _ in terms])]).astype(np.int32)
    off_c = (C.c_int32 * len(off))(*off.tolist())
    rm, prob = torch.zeros(K, device="cuda"), torch.zeros(n, device="cuda")
    viol, eprob = torch.zeros(nt, n, device="cuda"), torch.zeros(nt, n, device="cuda")
    ep_len = torch.zeros(n, dtype=torch.long, device="cuda")
    out, prev = torch.full((2 * nt,), -7.0, device="cuda"), torch.full((2 * nt,), 3.0, device="cuda")
    for t in range(steps):
        ep_len += 1
        dp = (C.c_float * nt)(*[native.f32(float(p)) for p in g["max_p"][t]])
        nat.cat_step(dev(_pack(stream[t], terms)), off_c, dp, 0.0, 0.95, t == 0, rm, prob, viol, eprob)
        if t == 2:       # empty mask: out <- prev
            nat.cat_reset(viol, eprob, ep_len, torch.zeros(n, dtype=torch.bool, device="cuda"), out, prev=prev)
            torch.cuda.synchronize()
            assert float((out - 3.0).abs().max()) == 0.0
        if t == 7:
            ids = np.asarray(g["reset7_ids"])
            m = np.zeros(n, bool)
            m[ids] = True
            nat.cat_reset(viol, eprob, ep_len, torch.from_numpy(m).cuda(), out, prev=prev)
            torch.cuda.synchronize()
            keys = [str(k) for k in g["reset7_keys"]]
            exp = dict(zip(keys, g["reset7_vals"]))
            o = out.cpu().numpy()
            for i, (nm, _, _) in enumerate(terms):
                np.testing.assert_allclose(o[2 * i], exp[f"Episode_Constraint_violation/{nm}"], rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(o[2 * i + 1], exp[f"Episode_Constraint_probability/{nm}"], rtol=1e-5,
                                           atol=1e-7)
            ep_len[torch.from_numpy(ids).cuda()] = 0
    np.testing.assert_array_equal(viol.cpu().numpy(), g["episode_sums"])
    np.testing.assert_array_equal(eprob.cpu().numpy(), g["cstr_mean_values"])


def test_set_term_cfg_with_the_same_object_after_an_in_place_edit_acts_on_the_next_step():
    """``get_term_cfg -> params["limit"] = x -> set_term_cfg(same object)`` used to leave the cached descriptor
    table untouched for up to 63 steps.  The reference reads ``term_cfg.params`` on every compute()
    (cat/constraint_manager.py:213-221): the very next compute() must see the new limit."""
    import smoke_impl
    from cat_envs.shim import make
    task, env_cfg, _ = smoke_impl.make_cfgs(64, 8, 256, 1, 1, (256, 256, 256), True, obs_dim=48, seed=4)
    env = make(task, cfg=env_cfg)
    env.reset()
    cm = env.unwrapped.constraint_manager
    act = torch.zeros(64, 12, device="cuda")
    for _ in range(3):
        env.step(act)                                        # the descriptor table is built and cached
    assert cm._desc_cache is not None
    name = "joint_torque"
    off = list(cm.active_terms).index(name)
    col0 = int(cm._term_off[off])
    raw0 = cm.cat._p_cstr[:, col0:col0 + 12].clone()         # |tau| - limit of the last step
    cfg = cm.get_term_cfg(name)
    old = float(cfg.params["limit"])
    cfg.params["limit"] = old + 1000.0                       # in-place edit of the SAME object ...
    cm.set_term_cfg(name, cfg)                               # ... handed back through the manager's API
    env.step(act)
    raw1 = cm.cat._p_cstr[:, col0:col0 + 12]
    assert float(raw1.max()) < -900.0 and float(raw0.max()) > -50.0, (float(raw0.max()), float(raw1.max()))
    # and without set_term_cfg the 64-step sweep still catches it eventually (documented fallback)
    cfg.params["limit"] = old
    for _ in range(70):
        env.step(act)
    assert float(cm.cat._p_cstr[:, col0:col0 + 12].max()) > -50.0


def test_branch_flip_analysis_sees_the_same_per_sample_quantities_on_both_sides():
    """the machinery behind the looser parameter bar, exercised where nothing flips: a 256-env iteration with per-step
    traces; asked about an absurdly tight bar (1e-9) the report names the first optimiser step, and the device's
    re-evaluation of that minibatch (catppo_policy_step under the traced parameters) agrees with the oracle's ratio /
    value difference to BOUNDARY_NOISE - the disagreement level the flip criterion (3) is calibrated on."""
    import parity_record
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=256, num_steps=24, minibatch=2048, epochs=3, iters=1, hidden=(256, 256, 256),
                                             six_terms=True, obs_dim=48, trace=True)
    rep = smoke_impl.compare(trainer, orc, outs[-1], check=False)
    assert rep["params"] < BARS["default"]["params"], rep
    assert smoke_impl.branch_flip_report(trainer, orc, BARS["default"]["params"])["first_step"] is None
    flip = smoke_impl.branch_flip_report(trainer, orc, 1e-9)
    assert flip["first_step"] is not None and flip["n_steps"] == 9 and len(flip["errs"]) == 9
    assert max(flip["errs"]) == pytest.approx(rep["params"], rel=1e-6) or max(flip["errs"]) >= rep["params"]
    assert flip["max_device_oracle_ratio_diff"] < BOUNDARY_NOISE and flip["max_device_oracle_value_diff"] < BOUNDARY_NOISE, flip
    if flip["flipped_surrogate"] + flip["flipped_value"]:
        assert flip["max_margin_of_flipped"] < BOUNDARY_NOISE, flip
    parity_record.record("branch_flip_machinery_256x24", {k: v for k, v in flip.items() if k != "errs"},
                         sizes=dict(num_envs=256, num_steps=24, minibatch=2048, epochs=3), seed=42)
