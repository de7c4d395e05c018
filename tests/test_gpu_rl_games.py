"""rl_games front end on the device against the reference's own ``CaTA2CAgent.play_steps``
(tests/golden/rlg_play_steps.npz: the method executed by gen_golden.py on a stub agent).

Reference: rl_games/cat_common.py:8-112 (float-dones play_steps), rl_games/cat_experience.py:7-33 (float dones in the
experience / replay buffers), rl_games/rl_games.py:9-43 (wrapper returning the termination probability as dones)."""
import numpy as np
import pytest
import torch

import streams as S

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


class _ScriptedVecEnv:
    """vec_env whose answers come from the golden's input stream (what the generator's stub env returned)"""

    def __init__(self, x):
        self.x, self.t = x, 0

    def reset(self):
        return dev(self.x["obs0"])

    def step(self, actions):
        t = self.t
        self.t += 1
        x = self.x
        return dev(x["next_obs"][t]), dev(x["rewards"][t]), dev(x["dones"][t]), {"time_outs": dev(x["time_outs"][t])}


def test_play_steps_vs_reference_golden(golden):
    """two consecutive horizons of play_steps: buffer rows (obs / previous float dones / policy outputs), value_bootstrap
    shaping, returns = discount_values + values, swap_and_flatten01 batch - bit-exact; the episode bookkeeping of
    catppo_rlg_episode_step (dones >= 1.0, current_* *= 1 - dones, length reset) bit-exact; meters 1e-6"""
    from cat_envs import native
    from cat_envs.tasks.utils.rl_games import CaTA2CAgent
    g = golden("rlg_play_steps")
    N, T, D, A, H = (int(g[k]) for k in ("N", "T", "D", "A", "H"))
    x = S.rlg_play_steps_inputs(int(g["seed"]), N, T, D, A, H)
    assert S.checksum(*[x[k] for k in sorted(x)]) == str(g["inputs_checksum"])
    env = _ScriptedVecEnv(x)
    seen = []

    class Observer:
        def process_infos(self, infos, done_indices):
            m = np.zeros(N, np.bool_)
            m[done_indices.reshape(-1).cpu().numpy()] = True
            seen.append(m)

    class Agent(CaTA2CAgent):                      # policy outputs scripted like the golden's stub agent
        def get_action_values(self, obs):
            t = env.t
            return {k: dev(x[k][t]) for k in ("actions", "values", "neglogpacs", "mus", "sigmas")}

        def get_values(self, obs):
            return dev(x["last_values"][self._h])

    ag = Agent(env, N, (D,), A, dict(horizon_length=T, gamma=0.99, tau=0.95, value_bootstrap=True, reward_scale=0.5,
                                     games_to_track=100), algo_observer=Observer())
    assert ag.dones.dtype == torch.float32 and float(ag.dones.min()) == 1.0                       # :30-33
    assert ag.experience_buffer.tensor_dict["dones"].dtype == torch.float32
    for h in range(H):
        ag._h = h
        batch = ag.play_steps()
        torch.cuda.synchronize()
        for k in ("obses", "dones", "values", "actions", "neglogpacs", "mus", "sigmas", "returns"):
            np.testing.assert_array_equal(batch[k].cpu().numpy(), g[f"h{h}_batch_{k}"], err_msg=k)
        assert batch["played_frames"] == int(g[f"h{h}_played_frames"])
        np.testing.assert_array_equal(ag.experience_buffer.tensor_dict["rewards"].cpu().numpy(), g[f"h{h}_buf_rewards"])
        np.testing.assert_array_equal(ag.current_rewards.cpu().numpy(), g[f"h{h}_current_rewards"])
        np.testing.assert_array_equal(ag.current_shaped_rewards.cpu().numpy(), g[f"h{h}_current_shaped_rewards"])
        np.testing.assert_array_equal(ag.current_lengths.cpu().numpy(), g[f"h{h}_current_lengths"])
        np.testing.assert_array_equal(ag.dones.cpu().numpy(), g[f"h{h}_final_dones"])
        np.testing.assert_allclose(ag.game_rewards.get_mean(), g[f"h{h}_game_rewards_mean"].reshape(-1), rtol=1e-6)
        np.testing.assert_allclose(ag.game_shaped_rewards.get_mean(), g[f"h{h}_game_shaped_rewards_mean"].reshape(-1),
                                   rtol=1e-6)
        np.testing.assert_allclose(ag.game_lengths.get_mean(), float(g[f"h{h}_game_lengths_mean"].reshape(-1)[0]), rtol=1e-6)
        assert ag.game_rewards.current_size == int(g[f"h{h}_game_size"]) == ag.game_lengths.current_size
    np.testing.assert_array_equal(np.stack(seen), g["done_mask"])                                 # dones.ge(1.0)
    m = ag.nat.rlg_meters_read(ag._meters)
    assert m.max_size == 100 and m.last_done_count == int(g["done_mask"][-1].sum())


def test_episode_step_kernel_vs_numpy_at_size():
    """catppo_rlg_episode_step at 32768 envs, value_size 2, a small meter window (max_size < finished episodes)"""
    from cat_envs import native
    from oracle import rlg_oracle as RO
    nat = native.get(torch.device("cuda", 0))
    N, V, steps = 32768, 2, 6
    rs = np.random.RandomState(3)
    cr, cs, cl = np.zeros((N, V), np.float32), np.zeros((N, V), np.float32), np.zeros(N, np.float32)
    d_cr, d_cs, d_cl = dev(cr), dev(cs), dev(cl)
    meters = nat.rlg_meters_new(50)
    mr, ms, ml = RO.AverageMeter((V,), 50), RO.AverageMeter((V,), 50), RO.AverageMeter((), 50)
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    for t in range(steps):
        rew = rs.uniform(-1, 2, (N, V)).astype(np.float32)
        shp = (rew * np.float32(0.3)).astype(np.float32)
        u = rs.rand(N)
        dn = np.where(u < 0.6, 0, np.where(u < 0.9, rs.uniform(0, 1, N), 1)).astype(np.float32)
        if t == 3:
            dn[:] = np.minimum(dn, np.float32(0.999))           # a step where nobody finishes: meters untouched
        nat.rlg_episode_step(dev(rew), dev(shp), dev(dn), d_cr, d_cs, d_cl, meters, mask)
        cr, cs, cl = (cr + rew).astype(np.float32), (cs + shp).astype(np.float32), cl + 1
        done = dn >= 1.0
        mr.update(cr[done]), ms.update(cs[done]), ml.update(cl[done])
        nd = (np.float32(1) - dn)[:, None]
        cr, cs = (cr * nd).astype(np.float32), (cs * nd).astype(np.float32)
        cl[done] = 0
        np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), done)
        np.testing.assert_array_equal(d_cr.cpu().numpy(), cr)
        np.testing.assert_array_equal(d_cs.cpu().numpy(), cs)
        np.testing.assert_array_equal(d_cl.cpu().numpy(), cl)
        m = nat.rlg_meters_read(meters)
        np.testing.assert_allclose([m.mean_rewards[v] for v in range(V)], mr.mean, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose([m.mean_shaped_rewards[v] for v in range(V)], ms.mean, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(m.mean_lengths, ml.mean, rtol=2e-6)
        assert m.size_rewards == mr.current_size and m.size_lengths == ml.current_size
        assert m.last_done_count == int(done.sum())


def test_wrapper_and_play_steps_on_the_cat_env():
    """RlGamesVecEnvWrapperCaT around the build's CaTEnv (float ``terminated`` handed over as dones, time-outs in the
    extras, "log" -> "episode") driving CaTA2CAgent with the HIP policy: one horizon, planes consistent with the
    device's own discount_values restated on the host"""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import Agent
    from cat_envs.tasks.utils.rl_games import CaTA2CAgent, RlGamesVecEnvWrapperCaT
    from oracle import rlg_oracle as RO
    task, env_cfg, _ = smoke_impl.make_cfgs(256, 8, 512, 1, 1, (256, 256, 256), True, obs_dim=48, seed=9)
    env = make(task, cfg=env_cfg)
    wrap = RlGamesVecEnvWrapperCaT(env, "cuda:0", clip_obs=100.0, clip_actions=100.0)
    torch.manual_seed(1)
    pol = Agent(env, hidden=(256, 256, 256)).cuda()
    ag = CaTA2CAgent(wrap, 256, (48,), 12, dict(horizon_length=8, gamma=0.99, tau=0.95, value_bootstrap=True), agent=pol)
    obs, rew, dones, extras = wrap.step(torch.zeros(256, 12, device="cuda"))
    assert dones.dtype == torch.float32 and float(dones.max()) <= 1.0 and "time_outs" in extras
    assert extras["time_outs"].dtype == torch.bool and "episode" in extras and "log" not in extras
    assert ((dones > 0) & (dones < 1)).any()                # termination PROBABILITIES reach the agent
    batch = ag.play_steps()
    torch.cuda.synchronize()
    buf = ag.experience_buffer.tensor_dict
    f = lambda t: t.cpu().numpy()
    last_values = f(pol.get_value(ag.obs["obs"]))
    advs = RO.discount_values(f(ag.dones), last_values, f(buf["dones"]), f(buf["values"]), f(buf["rewards"]), 0.99, 0.95)
    np.testing.assert_array_equal(f(batch["returns"]), RO.swap_and_flatten01((advs + f(buf["values"])).astype(np.float32)))
    assert f(buf["dones"])[0].min() == 1.0 and batch["obses"].shape == (256 * 8, 48)
    assert np.isfinite(f(batch["returns"])).all() and float(ag.current_lengths.max()) <= 8


def test_vectorized_replay_buffer_float_dones_and_wraparound():
    from cat_envs.tasks.utils.rl_games import CaTVectorizedReplayBuffer
    rb = CaTVectorizedReplayBuffer((5,), (3,), capacity=10, device="cuda")
    assert rb.dones.dtype == torch.float32 and rb.dones.shape == (10, 1)
    mk = lambda n, s: (torch.full((n, 5), s, device="cuda"), torch.full((n, 3), s, device="cuda"),
                       torch.full((n, 1), s, device="cuda"), torch.full((n, 5), s + 0.5, device="cuda"),
                       torch.full((n, 1), s / 10.0, device="cuda"))
    rb.add(*mk(4, 1.0))
    rb.add(*mk(4, 2.0))
    assert rb.idx == 8 and not rb.full
    rb.add(*mk(4, 3.0))                                   # wraps: rows 8, 9, 0, 1
    assert rb.idx == 2 and rb.full
    np.testing.assert_array_equal(rb.dones.cpu().numpy()[:, 0],
                                  np.float32([0.3, 0.3, 0.1, 0.1, 0.2, 0.2, 0.2, 0.2, 0.3, 0.3]))
    o, a, r, no, d = rb.sample(64)
    assert d.dtype == torch.float32 and o.shape == (64, 5) and {round(float(v), 3) for v in np.unique(d.cpu().numpy())} <= {0.1, 0.2, 0.3}
    np.testing.assert_array_equal((no - o).cpu().numpy(), np.full((64, 5), 0.5, np.float32))
