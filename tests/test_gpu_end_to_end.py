"""Whole hot path on the GPU (env step with fused constraint terms + CaT step, rollout, GAE,
normalisers, epochs x minibatches, clip + Adam) against the CPU oracle on the same streams."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_smoke_pair_reference_arch():
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=64, num_steps=24, minibatch=512, epochs=3, iters=2)
    rep = smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)
    print(rep)
    # normaliser state
    np.testing.assert_allclose(trainer.agent.obs_rms.running_mean.cpu().numpy(), orc.agent.obs_rms.mean.numpy(),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(trainer.agent.value_rms.running_var), float(orc.agent.value_rms.var), rtol=1e-4)
    assert float(trainer.agent.obs_rms.count) == float(orc.agent.obs_rms.count)


def test_full_constraint_set_and_3x256_arch():
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=256, num_steps=8, minibatch=1024, epochs=1, iters=1,
                                             hidden=(256, 256, 256), six_terms=False)
    smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)
    cm = trainer.envs.constraint_manager
    assert len(cm.active_terms) == 13 and cm.cat._p_cstr.shape == (256, 78)
    # running maxima and per-term statistics of the manager are bit-identical to the oracle's
    np.testing.assert_array_equal(cm.cat.get_running_maxes().cpu().numpy()[0], orc.env.mgr.cat.get_running_maxes()[0])
    for i, name in enumerate(cm.active_terms):
        np.testing.assert_array_equal(cm._episode_sums[name].cpu().numpy(), orc.env.mgr.episode_sums[name])
        np.testing.assert_array_equal(cm._cstr_mean_values[name].cpu().numpy(), orc.env.mgr.cstr_mean_values[name])


def test_reference_api_surface_on_device():
    """CaT.add / get_probs, standalone term functions, Agent API, checkpoint round trip."""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cat import CaT, constraints
    from cat_envs.tasks.utils.cleanrl.ppo import Agent
    from oracle import cat_oracle as CO
    import streams as S
    task, env_cfg, _ = smoke_impl.make_cfgs(128, 8, 256, 1, 1)
    env = make(task, cfg=env_cfg)
    env.reset()
    env.step(torch.zeros(128, 12, device="cuda"))
    # single-term functions == their column block in the manager's packed matrix
    cm = env.constraint_manager
    for name, cfg in zip(cm.active_terms, cm._term_cfgs):
        out = cfg.func(env, **cfg.params)
        ref = cm.cat.raw_constraints[name]
        got = out.float().reshape(128, -1)
        np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
    assert constraints.contact(env, **cm.get_term_cfg("contact").params).dtype == torch.bool
    # CaT.add one term at a time (the reference's call pattern) vs the oracle
    cat, orc = CaT(0.95, 0.0), CO.CaTOracle(0.95, 0.0)
    stream = S.cat_stream(5, 100, S.CAT_TERMS_SMALL, 4)
    for step in stream:
        for (name, _, _), mp in zip(S.CAT_TERMS_SMALL, [0.25, 1.0, 0.25, 1.0, 0.5]):
            cat.add(name, torch.from_numpy(np.asarray(step[name])).cuda(), mp)
            orc.add(name, step[name], mp)
        np.testing.assert_array_equal(cat.get_probs().cpu().numpy(), orc.get_probs())
        np.testing.assert_array_equal(cat.get_running_maxes().cpu().numpy(), orc.get_running_maxes())
    assert cat.get_names() == [t[0] for t in S.CAT_TERMS_SMALL] and len(cat.get_vals()) == 5
    # Agent: 23-key state_dict, load/save round trip through the flat buffer
    ag = Agent(env)
    sd = ag.state_dict()
    assert len(sd) == 23 and sd["actor_mean.0.weight"].shape == (512, 45) and sd["critic.6.weight"].shape == (1, 128)
    w = S.agent_weights(3, 45, 12)
    sd2 = {k: (torch.from_numpy(w[k]) if k in w else v) for k, v in sd.items()}
    ag.load_state_dict(sd2)
    x = torch.randn(33, 45, device="cuda")
    a, lp, ent, v = ag.get_action_and_value(x, deterministic=True)
    from oracle import ppo_oracle as PO
    o = PO.AgentOracle(45, 12)
    o.load(w)
    with torch.no_grad():
        a2, lp2, ent2, v2 = o.get_action_and_value(x.cpu(), deterministic=True)
    np.testing.assert_allclose(a.cpu().numpy(), a2.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(v.cpu().numpy(), v2.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), ent2.numpy(), rtol=1e-6)
    _, lp3, _, _ = ag.get_action_and_value(x, action=a2.cuda() * 1.1)
    with torch.no_grad():
        _, lp4, _, _ = o.get_action_and_value(x.cpu(), a2 * 1.1)
    np.testing.assert_allclose(lp3.cpu().numpy(), lp4.numpy(), rtol=1e-5, atol=1e-4)
    assert ag(x).shape == (33, 12)


def test_config4_long_horizon_wide_obs():
    """BASELINE config 4 shapes: horizon 48, 235-d observations (48 + 187 height scan), 3x256 MLP"""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=128, num_steps=48, minibatch=2048, epochs=1, iters=1,
                                             hidden=(256, 256, 256), obs_dim=235)
    assert trainer.Dp == 240 and trainer.obs.shape == (49, 128, 240)
    smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)


def test_config3_per_gpu_shard_full_constraints():
    """BASELINE config 3, one rank's share: 2048 envs, full 13-term ConstraintsCfg, reference MLP"""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=2048, num_steps=4, minibatch=2048, epochs=1, iters=1,
                                             six_terms=False)
    smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)


def test_env_sharded_code_path_on_one_gpu():
    """CATPPO_FORCE_DIST=1: RCCL world of size 1, every exchange point active (two-phase CaT step with the
    MAX all-reduce, fp64 moment sums, external advantage statistics, flat-gradient all-reduce)."""
    import os
    import subprocess
    import sys
    code = (
        "import os, torch, sys\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "import smoke_impl\n"
        "from cat_envs import parallel\n"
        "assert parallel.active()\n"
        "t, o, outs = smoke_impl.run_pair(num_envs=64, num_steps=8, minibatch=256, epochs=2, iters=2)\n"
        "assert t.envs.constraint_manager.dist_group is not None and t.agent.obs_rms.dist_group is not None\n"
        "print(smoke_impl.compare(t, o, outs[-1], tol_scale=2.0))\n"
        "torch.distributed.destroy_process_group()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CATPPO_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([root, os.path.join(root, "constraints-as-terminations_amd")]))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_train_and_play_entry_points(tmp_path):
    """scripts/clean_rl/train.py with the reference's flags, checkpoint written with the reference's naming,
    then scripts/clean_rl/play.py loads it and rolls the policy out."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (like the reference, --experiment_name is accepted but the log directory follows the task's cfg)
    common = ["--task=Isaac-Velocity-CaT-Flat-Solo12-v0", "--headless", "--num_envs", "256"]
    r = subprocess.run([sys.executable, os.path.join(root, "scripts/clean_rl/train.py"), *common, "--num_iterations", "4",
                        "--seed", "3", "agent.save_interval=2", "agent.minibatch_size=2048", "env.synthetic.stream_steps=32"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Starting training for 4 steps" in r.stdout and "Saved model" in r.stdout
    runs = os.listdir(tmp_path / "logs" / "clean_rl" / "solo12_flat")
    assert len(runs) == 1
    run = tmp_path / "logs" / "clean_rl" / "solo12_flat" / runs[0]
    files = sorted(os.listdir(run))
    assert "model_1.pt" in files and "model_3.pt" in files          # (iteration + 1) % save_interval == 0
    assert os.path.exists(run / "params" / "agent.yaml") and os.path.exists(run / "params" / "env.pkl")
    sd = torch.load(run / "model_3.pt", map_location="cpu")
    assert len(sd) == 23 and sd["actor_mean.0.weight"].shape == (512, 45) and float(sd["obs_rms.count"]) > 1
    assert all(torch.isfinite(v).all() for v in sd.values())
    r = subprocess.run([sys.executable, os.path.join(root, "scripts/clean_rl/play.py"), *common, "--video_length", "8"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "model_3.pt" in r.stdout and "mean reward per step" in r.stdout


def test_config5_env_count_full_constraints_7_envs_and_ragged():
    """BASELINE config 5 env count (32768 envs, 13 terms with mixed hard/soft max_p: every term-kernel workgroup walks
    8 tiles and emits one column-maximum partial) and a 7-env run (less than one 16-env tile): termination masks
    bit-exact against the CPU oracle in both."""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=32768, num_steps=2, minibatch=16384, epochs=1, iters=1,
                                             six_terms=False)
    smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)
    trainer, orc, outs = smoke_impl.run_pair(num_envs=7, num_steps=5, minibatch=35, epochs=1, iters=2, six_terms=False)
    smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)


def test_cross_workgroup_handshake_of_the_fused_step_1500_steps():
    """The "last workgroup folds" hand-shake of rollout_pre exchanges its partial rows with device-scope atomic stores /
    loads ordered by a completion wait - no device-scope fences.  A stale read there would show up as
    a wrong column maximum or moment sum in SOME step: 1500 consecutive fused env steps at cfg2's size, the exchange
    record of every step recomputed independently from the constraint tile and the raw observations the same launch
    wrote / read.  Column maxima bit-exact (max is order independent), fp64 moment sums to 1e-12 relative."""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    N, T, S = 4096, 24, 1500
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(N, T, 16384, 1, 1, (256, 256, 256), True, obs_dim=48, stream_steps=64)
    env = make(task, cfg=env_cfg)
    tr = PPOTrainer(env, agent_cfg)
    assert tr.sink is not None
    env_u = env.unwrapped
    cm = env_u.constraint_manager
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.randn(T, N, tr.A, device="cuda", generator=g)
    K, D = cm.cat._p_cstr.shape[1], env_u.obs_dim
    got_max, got_sum = torch.empty(S, K, device="cuda"), torch.empty(S, 2 * D, dtype=torch.float64, device="cuda")
    ref_max, ref_sum = torch.empty_like(got_max), torch.empty_like(got_sum)
    obs_raw = env_u.sim.view("obs")
    for k in range(S):
        tr.sink.step = k % T
        env_u.step_into(acts[k % T], tr.sink)
        got_max[k].copy_(env_u._xchg_views[0])
        got_sum[k].copy_(env_u._xchg_views[1])
        ref_max[k].copy_(cm.cat._p_cstr.max(dim=0).values.clamp_min(1e-6))
        x = obs_raw.double()
        ref_sum[k, :D].copy_(x.sum(0))
        ref_sum[k, D:].copy_((x * x).sum(0))
    torch.cuda.synchronize()
    assert torch.isfinite(got_max).all() and float(ref_max.max()) > 1e-3
    bad = (got_max != ref_max).any(dim=1).nonzero().flatten().tolist()
    assert not bad, f"column maxima differ in steps {bad[:10]} (of {len(bad)})"
    rel = ((got_sum - ref_sum).abs() / ref_sum.abs().clamp_min(1.0)).max(dim=1).values
    assert float(rel.max()) < 1e-12, (float(rel.max()), int(rel.argmax()))
