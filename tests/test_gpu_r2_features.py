"""GPU tests of the ABI 0.2 entry points: device-resident iteration state (lr schedules, Adam step), Philox action
noise and keyed permutation, fused rollout step, hipGraph replay of the update phase, RCCL under the C ABI, GAE scan
mode, fp16 rollout planes (BASELINE config 5)."""
import numpy as np
import pytest
import torch

import streams as S
from oracle import ppo_oracle as PO
from oracle import rng_oracle as RO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from cat_envs import native
    return native.get(torch.device("cuda", 0))


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _params(native, D, A, hidden, seed=3):
    from test_gpu_kernels import flat_params
    shape = native.shape_of(D, A, hidden)
    lay = native.layout_of(shape)
    w = S.agent_weights(seed, D, A, hidden)
    return shape, lay, w, flat_params(native, shape, lay, w)


# ------------------------------------------------------------------------------------------ iteration state
def test_iter_state_linear_schedule_and_device_adam_match_host_path(nat):
    """catppo_iter_begin (linear anneal in double on the device) + catppo_clip_adam_dev == catppo_clip_adam with the
    host-side lr / step count, over several steps and iterations (bit-identical parameters)."""
    from cat_envs import native
    n = 377241 + 7
    rs = np.random.RandomState(0)
    p0 = rs.standard_normal(n).astype(np.float32)
    pa, pb = dev(p0), dev(p0)
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    st = nat.iter_state_new(123, 3e-4)
    step = 0
    for it in range(1, 4):
        nat.iter_begin(st, 3e-4, 2000, native.LR_LINEAR)
        lr = (1.0 - (it - 1.0) / 2000) * 3e-4
        for k in range(3):
            g = (rs.standard_normal(n) * (10.0 if k == 1 else 0.01)).astype(np.float32)   # clipped and unclipped
            ga, gb = dev(g), dev(g)
            step += 1
            nat.clip_adam(pa, ga, ma, va, n, 1.0, lr, 0.9, 0.999, 1e-5, step)
            nat.clip_adam_dev(pb, gb, mb, vb, n, 1.0, 0.9, 0.999, 1e-5, st)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(ga.cpu().numpy(), gb.cpu().numpy())
            # bias corrections: pow() on the device vs the host - the fp32 step size may differ in its last bit
            np.testing.assert_allclose(pb.cpu().numpy(), pa.cpu().numpy(), rtol=0, atol=1e-9)
    s = nat.iter_state_read(st)
    assert s.iteration == 3 and s.adam_step == 9 and s.seed == 123
    assert s.lr == (1.0 - 2.0 / 2000) * 3e-4
    nat.iter_begin(st, 1e-3, 10, native.LR_FIXED)
    assert nat.iter_state_read(st).lr == 1e-3
    nat.iter_begin(st, 5e-3, 10, native.LR_KEEP)
    assert nat.iter_state_read(st).lr == 1e-3 and nat.iter_state_read(st).iteration == 5


def test_kl_adaptive_lr_rule(nat):
    """skrl KLAdaptiveLR / rl_games adaptive schedule (skrl/ppo.py:558-567; skrl_ppo_cfg.yaml:49-51):
    kl > 2*thr -> lr/1.5 (>= 1e-6);  kl < thr/2 -> lr*1.5 (<= 1e-2);  per-epoch mean of the minibatch KLs."""
    from cat_envs import native
    st = nat.iter_state_new(1, 1e-3)
    nat.iter_begin(st, 1e-3, 10, native.LR_KEEP)
    diag = torch.zeros(8, device="cuda")
    kl = torch.zeros(1, device="cuda")
    thr = 0.01
    lr = 1e-3
    tot_kl, tot_n = 0.0, 0.0
    rs = np.random.RandomState(1)
    for epoch in range(40):
        kls = rs.choice([0.0005, 0.004, 0.012, 0.05, 0.3], size=6)
        tot_kl += float(kls.sum())
        tot_n += 6
        diag[4], diag[7] = tot_kl, tot_n
        nat.kl_mean(st, diag, kl)
        m = float(kl)
        np.testing.assert_allclose(m, kls.mean(), rtol=2e-4)
        nat.kl_adaptive_lr(st, kl, thr)
        if m > thr * 2.0:
            lr = max(lr / 1.5, 1e-6)
        elif m < thr / 2.0:
            lr = min(lr * 1.5, 1e-2)
        s = nat.iter_state_read(st)
        assert s.lr == lr, (epoch, s.lr, lr)
        assert abs(s.last_kl - m) < 1e-12
    # clamps
    diag[4], diag[7] = tot_kl + 100.0, tot_n + 1
    for _ in range(40):
        diag[4] += 100.0
        diag[7] += 1
        nat.kl_mean(st, diag, kl)
        nat.kl_adaptive_lr(st, kl, thr)
    assert nat.iter_state_read(st).lr == 1e-6
    for _ in range(60):
        diag[7] += 1
        nat.kl_mean(st, diag, kl)
        nat.kl_adaptive_lr(st, kl, thr)
    assert nat.iter_state_read(st).lr == 1e-2
    # a new iteration restarts the marks (diag is zeroed per iteration)
    nat.iter_begin(st, 1e-3, 10, native.LR_KEEP)
    diag.zero_()
    diag[4], diag[7] = 0.06, 6.0
    nat.kl_mean(st, diag, kl)
    np.testing.assert_allclose(float(kl), 0.01, rtol=1e-5)


# ------------------------------------------------------------------------------------------ randomness
@pytest.mark.parametrize("N,A,hidden", [(4096, 12, (256, 256, 256)), (1000, 12, (512, 256, 128)), (70, 7, (64,))])
def test_philox_action_noise_matches_restatement(nat, N, A, hidden):
    from cat_envs import native
    D = 48
    shape, lay, w, params = _params(native, D, A, hidden)
    x = dev(np.random.RandomState(4).standard_normal((N, lay.obs_pad)).astype(np.float32))
    st = nat.iter_state_new(0x1234567890ABCDEF, 3e-4)
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)          # iteration 2
    nat.mlp_reserve(shape, N)
    act, logp, val = torch.empty(N, A, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    eps = torch.zeros(N, A, device="cuda")
    for step in (0, 5):
        nat.policy_act_rng(shape, params, x, N, st, step, act, logp, val, eps_out=eps)
        torch.cuda.synchronize()
        exp = RO.action_noise(0x1234567890ABCDEF, 2, step, N, A)
        np.testing.assert_allclose(eps.cpu().numpy(), exp, rtol=2e-5, atol=2e-6)
        # the same noise through the supplied-noise entry gives the same action / log-prob / value
        a2, l2, v2 = torch.empty_like(act), torch.empty_like(logp), torch.empty_like(val)
        nat.policy_act_ex(shape, params, x, N, eps, a2, l2, v2)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(act.cpu().numpy(), a2.cpu().numpy())
        np.testing.assert_array_equal(logp.cpu().numpy(), l2.cpu().numpy())
        np.testing.assert_array_equal(val.cpu().numpy(), v2.cpu().numpy())
    e = eps.cpu().numpy()
    assert abs(e.mean()) < 0.05 and abs(e.std() - 1.0) < 0.05
    # another iteration -> another draw
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
    eps2 = torch.zeros(N, A, device="cuda")
    nat.policy_act_rng(shape, params, x, N, st, 5, act, logp, val, eps_out=eps2)
    assert float((eps2 - eps).abs().max()) > 0.1


@pytest.mark.parametrize("B,M", [(98304, 16384), (4096 * 3, 2048), (35, 35), (1000, 300)])
def test_keyed_permutation_gather(nat, B, M):
    from cat_envs import native
    D, A = 45, 12
    shape = native.shape_of(D, A, (64,))
    lay = native.layout_of(shape)
    Dp = lay.obs_pad
    rs = np.random.RandomState(B)
    obs = dev(rs.standard_normal((B, Dp)).astype(np.float32))
    act = dev(rs.standard_normal((B, A)).astype(np.float32))
    sc = [dev(rs.standard_normal(B).astype(np.float32)) for _ in range(4)]   # logp, adv, ret, val
    st = nat.iter_state_new(77, 3e-4)
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
    n_mb = (B + M - 1) // M
    parts = (M + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
    xg, ag = torch.empty(B, Dp, device="cuda"), torch.empty(B, A, device="cuda")
    sg = torch.empty(4 * B, device="cuda")
    ap = torch.empty(n_mb * parts * 2, dtype=torch.float64, device="cuda")
    inds = torch.empty(B, dtype=torch.int64, device="cuda")
    seen = []
    for epoch in (0, 3):
        nat.ppo_gather_ex(shape, obs, act, sc[0], sc[1], sc[2], sc[3], B, M, xg, ag, sg, ap, st=st, epoch=epoch,
                          inds_out=inds)
        torch.cuda.synchronize()
        p = inds.cpu().numpy()
        np.testing.assert_array_equal(p, RO.permutation(77, 1, epoch, B))
        assert sorted(p.tolist()) == list(range(B))
        np.testing.assert_array_equal(xg.cpu().numpy(), obs.cpu().numpy()[p])
        np.testing.assert_array_equal(ag.cpu().numpy(), act.cpu().numpy()[p])
        # the same permutation through the index-array form: identical packed buffers
        xg2, ag2, sg2, ap2 = torch.empty_like(xg), torch.empty_like(ag), torch.empty_like(sg), torch.empty_like(ap)
        nat.ppo_gather_ex(shape, obs, act, sc[0], sc[1], sc[2], sc[3], B, M, xg2, ag2, sg2, ap2, inds=inds)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sg.cpu().numpy(), sg2.cpu().numpy())
        np.testing.assert_array_equal(ap.cpu().numpy(), ap2.cpu().numpy())
        seen.append(p.copy())
    assert (seen[0] != seen[1]).mean() > 0.9 or B < 100


# ------------------------------------------------------------------------------------------ GAE scan mode
@pytest.mark.parametrize("T,N", [(24, 4096), (48, 4096), (24, 64), (5, 1000), (1, 7), (2, 3), (33, 129), (100, 50),
                                 (200, 16)])
def test_gae_scan_mode_vs_serial(nat, T, N):
    """catppo_gae_planes(mode = SCAN): wavefront-shuffle scan over the time axis; <= 1e-5 of the bit-exact serial kernel
    (north_star: returns / advantages within 1e-5)."""
    from cat_envs import native
    x = S.gae_inputs(T * 31 + N, T, N)
    d = {k: dev(v) for k, v in x.items()}
    out = {}
    for mode in (native.GAE_SERIAL, native.GAE_SCAN):
        adv, ret = torch.full((T, N), 7.0, device="cuda"), torch.full((T, N), 7.0, device="cuda")
        nat.gae_mode(mode, d["rewards"], d["values"], d["dones"], d["true_dones"], d["next_value"], d["next_done"],
                     d["next_true_done"], 0.99, 0.95, adv, ret)
        torch.cuda.synchronize()
        out[mode] = (adv.cpu().numpy(), ret.cpu().numpy())
    a, r = PO.gae_numpy_exact(x["rewards"], x["values"], x["dones"], x["true_dones"], x["next_value"], x["next_done"],
                              x["next_true_done"], 0.99, 0.95)
    np.testing.assert_array_equal(out[native.GAE_SERIAL][0], a)
    scale = max(1.0, float(np.abs(a).max()))
    assert np.abs(out[native.GAE_SCAN][0] - a).max() <= 1e-5 * scale
    assert np.abs(out[native.GAE_SCAN][1] - r).max() <= 1e-5 * scale


# ------------------------------------------------------------------------------------------ fp16 planes
def test_fp16_plane_producers(nat):
    """every producer / consumer of the fp16 rollout planes: rollout_store_ex, value in half from the head kernel,
    normaliser over half inputs, gather with half advantages."""
    from cat_envs import native
    N, A, D = 1000, 12, 48
    rs = np.random.RandomState(9)
    reward = rs.uniform(0, 1.5, N).astype(np.float32)
    dones = S.soft_dones(rs, (N,))
    to = rs.rand(N) < 0.1
    r16, d16, t16 = (torch.zeros(N, dtype=torch.float16, device="cuda") for _ in range(3))
    nat.rollout_store_ex(dev(reward), dev(dones), dev(to), r16, d16, t16)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r16.cpu().numpy(), reward.astype(np.float16))
    np.testing.assert_array_equal(d16.cpu().numpy(), dones.astype(np.float16))
    np.testing.assert_array_equal(t16.cpu().numpy(), to.astype(np.float16))
    # head kernel value in half == half(fp32 value)
    shape, lay, w, params = _params(native, D, A, (256, 256, 256))
    x = dev(rs.standard_normal((N, lay.obs_pad)).astype(np.float32))
    eps = dev(rs.standard_normal((N, A)).astype(np.float32))
    nat.mlp_reserve(shape, N)
    act, logp = torch.empty(N, A, device="cuda"), torch.empty(N, device="cuda")
    v32, v16 = torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.float16, device="cuda")
    nat.policy_act_ex(shape, params, x, N, eps, act, logp, v32)
    nat.policy_act_ex(shape, params, x, N, eps, act, logp, v16)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(v16.cpu().numpy(), v32.cpu().numpy().astype(np.float16))
    nv16 = torch.empty(N, dtype=torch.float16, device="cuda")
    nat.value_ex(shape, params, x, N, nv16)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(nv16.cpu().numpy(), v32.cpu().numpy().astype(np.float16))
    # normaliser over a half plane == normaliser over its widened copy
    y = (rs.standard_normal(24 * N) * 3 + 1).astype(np.float16)
    st = [[torch.zeros(1, device="cuda"), torch.ones(1, device="cuda"), torch.ones(1, device="cuda")] for _ in range(2)]
    o = [torch.empty(24 * N, device="cuda") for _ in range(2)]
    for i, t in enumerate((dev(y), dev(y.astype(np.float32)))):
        nat.rms_update_ex(t, 24 * N, 1, 1, *st[i])
        nat.rms_normalize_ex(t, 24 * N, 1, 1, st[i][0], st[i][1], 1e-8, o[i], 1)
    torch.cuda.synchronize()
    for a, b in zip(st[0], st[1]):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_array_equal(o[0].cpu().numpy(), o[1].cpu().numpy())
    # gather with half advantages == gather with their widened copy
    B, M = 24 * N, 4096
    obs = dev(rs.standard_normal((B, lay.obs_pad)).astype(np.float32))
    ac = dev(rs.standard_normal((B, A)).astype(np.float32))
    lp, rt, vl = (dev(rs.standard_normal(B).astype(np.float32)) for _ in range(3))
    perm = dev(rs.permutation(B).astype(np.int64))
    n_mb, parts = (B + M - 1) // M, (M + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
    res = []
    for adv in (dev(y), dev(y.astype(np.float32))):
        xg, ag = torch.empty(B, lay.obs_pad, device="cuda"), torch.empty(B, A, device="cuda")
        sg, ap = torch.empty(4 * B, device="cuda"), torch.empty(n_mb * parts * 2, dtype=torch.float64, device="cuda")
        nat.ppo_gather_ex(shape, obs, ac, lp, adv, rt, vl, B, M, xg, ag, sg, ap, inds=perm)
        torch.cuda.synchronize()
        res.append((sg.cpu().numpy(), ap.cpu().numpy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_cfg5_32768_envs_fp16_planes_bf16_mlp_end_to_end():
    """BASELINE configs[4] as ONE trainer configuration: 32768 envs x 24, 13 ConstraintTerms with mixed hard (1.0) /
    soft (0.25, 0.1) max_p, fp16 rollout planes, bf16-operand MLP GEMMs.  Termination masks bit-exact vs the CPU
    oracle; GAE bit-exact on the device's own planes after widening (fp32 recurrence, RNE to half); whole iteration
    within bf16 distance of the oracle's restatement of that arithmetic."""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=32768, num_steps=24, minibatch=16384, epochs=5, iters=1,
                                             hidden=(256, 256, 256), six_terms=False, obs_dim=48,
                                             agent_overrides={"rollout_dtype": "fp16", "mlp_precision": "bf16"})
    assert trainer.adam_step == 5 * 48           # the configuration's own 5 epochs x 48 minibatches
    T = trainer.T
    assert trainer.rewards.dtype == torch.float16 and trainer.advantages.dtype == torch.float16
    assert trainer.agent.shape.mfma_bf16 == 1 and trainer.sink is not None
    cm = trainer.envs.constraint_manager
    mp = sorted(set(float(c.max_p) for c in cm._term_cfgs))
    assert len(cm.active_terms) == 13 and mp[-1] == 1.0 and mp[0] < 0.25          # mixed hard / soft
    # (1) masks: rewards = reward*(1-p) clipped, dones = p | 1 - bit-exact after the same RNE to half on both sides
    np.testing.assert_array_equal(trainer.rewards.float().cpu().numpy(), orc.rewards.numpy())
    np.testing.assert_array_equal(trainer.dones[1:T].float().cpu().numpy(), orc.dones[1:].numpy())
    np.testing.assert_array_equal(cm.cat.get_running_maxes().cpu().numpy()[0], orc.env.mgr.cat.get_running_maxes()[0])
    # (2) GAE on the device's own half planes: widen, exact fp32 recurrence, round -> bit-exact
    w = lambda t: t.float().cpu().numpy()
    a, r = PO.gae_numpy_exact(w(trainer.rewards), w(trainer.values), w(trainer.dones[:T]), w(trainer.true_dones[:T]),
                              w(trainer.next_value), w(trainer.dones[T]), w(trainer.true_dones[T]), 0.99, 0.95)
    np.testing.assert_array_equal(trainer.advantages.cpu().numpy(), a.astype(np.float16))
    np.testing.assert_array_equal(trainer.returns.cpu().numpy(), r.astype(np.float16))
    # (3) whole iteration vs the oracle's bf16-operand / fp16-plane restatement (bf16 tolerance: a last-bit fp32
    #     difference can flip a bf16 rounding, see test_gpu_bf16.py)
    rep = smoke_impl.compare(trainer, orc, outs[-1], check=False)
    print(rep)
    import parity_record
    parity_record.record("cfg5_32768x24_fp16_planes_bf16_mlp", rep,
                         sizes=dict(num_envs=32768, num_steps=24, minibatch=16384, epochs=5, hidden=[256, 256, 256]),
                         seed=42, note="bf16-operand GEMMs + fp16 planes vs the oracle's restatement of that arithmetic: "
                                       "bf16-distance bars, not the fp32 ones")
    assert rep["values"] < 2e-2 and rep["logprobs"] < 2e-2 and rep["advantages"] < 5e-2, rep
    assert rep["params"] < 4e-3, rep
    assert np.isfinite(trainer.agent.flat.cpu().numpy()).all()


# ------------------------------------------------------------------------------------------ fused rollout / graphs
def _two_trainers(over_a, over_b, num_envs=256, num_steps=8, minibatch=512, epochs=2, iters=3, six_terms=False,
                  hidden=(256, 256, 256), obs_dim=48, inject=False):
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    out = []
    for over in (over_a, over_b):
        task, env_cfg, agent_cfg = smoke_impl.make_cfgs(num_envs, num_steps, minibatch, epochs, 50, hidden, six_terms,
                                                        obs_dim=obs_dim, seed=7)
        for k, v in over.items():
            setattr(agent_cfg, k, v)
        torch.manual_seed(3)
        env = make(task, cfg=env_cfg)
        tr = PPOTrainer(env, agent_cfg)
        g, gp = torch.Generator(device="cuda").manual_seed(5), torch.Generator(device="cuda").manual_seed(6)
        B = num_envs * num_steps
        for _ in range(iters):
            if inject:
                tr.run_iteration(eps_fn=lambda s: torch.randn(num_envs, 12, device="cuda", generator=g),
                                 perm_fn=lambda e: torch.randperm(B, device="cuda", generator=gp), log=False)
            else:
                tr.run_iteration(log=False)
        torch.cuda.synchronize()
        out.append(tr)
    return out


def test_fused_rollout_step_equals_unfused_calls():
    """catppo_rollout_pre/_post (2 launches) vs env_pre_step + cat_terms_step + cat_reset + rollout_store +
    rms_update + rms_normalize (10 launches): same noise / permutations on both sides."""
    a, b = _two_trainers({"fused_rollout": True}, {"fused_rollout": False}, inject=True)
    assert a.sink is not None and b.sink is None
    for name in ("rewards", "dones", "true_dones", "actions", "logprobs", "values", "advantages", "returns"):
        np.testing.assert_array_equal(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy(), err_msg=name)
    ca, cb = a.envs.constraint_manager, b.envs.constraint_manager
    np.testing.assert_array_equal(ca.cat._p_rm.cpu().numpy(), cb.cat._p_rm.cpu().numpy())
    np.testing.assert_array_equal(ca._ep_viol.cpu().numpy(), cb._ep_viol.cpu().numpy())
    np.testing.assert_array_equal(ca._ep_prob.cpu().numpy(), cb._ep_prob.cpu().numpy())
    np.testing.assert_array_equal(a.envs.episode_length_buf.cpu().numpy(), b.envs.episode_length_buf.cpu().numpy())
    np.testing.assert_array_equal(a.envs.action_manager._prev_action.cpu().numpy(),
                                  b.envs.action_manager._prev_action.cpu().numpy())
    # reset statistics of the last step (log ring): same fp64 sums in a different fixed order
    np.testing.assert_allclose(ca._log_ring[ca._log_pos].cpu().numpy(), cb._log_ring[cb._log_pos].cpu().numpy(),
                               rtol=1e-6, atol=1e-9, equal_nan=True)
    # normaliser state / normalised observations: fp64 sums folded in another fixed order (<= 1 ulp of the state)
    np.testing.assert_allclose(a.agent.obs_rms.running_mean.cpu().numpy(), b.agent.obs_rms.running_mean.cpu().numpy(),
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(a.agent.obs_rms.running_var.cpu().numpy(), b.agent.obs_rms.running_var.cpu().numpy(),
                               rtol=1e-6)
    assert float(a.agent.obs_rms.count) == float(b.agent.obs_rms.count)
    np.testing.assert_allclose(a.obs.cpu().numpy(), b.obs.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.agent.flat.cpu().numpy(), b.agent.flat.cpu().numpy(), rtol=0, atol=1e-6)


def test_fused_rollout_ragged_and_wide_obs():
    """N not a multiple of the 16 / 32 env tiles, 235-d observations, reference MLP, 13 terms"""
    a, b = _two_trainers({"fused_rollout": True}, {"fused_rollout": False}, num_envs=1000, num_steps=4,
                         minibatch=1000, epochs=1, iters=2, hidden=(512, 256, 128), obs_dim=235, inject=True)
    for name in ("rewards", "dones", "true_dones"):
        np.testing.assert_array_equal(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy(), err_msg=name)
    np.testing.assert_allclose(a.obs.cpu().numpy(), b.obs.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(a.envs.constraint_manager._ep_prob.cpu().numpy(),
                                  b.envs.constraint_manager._ep_prob.cpu().numpy())


def test_update_phase_graph_replay_is_bit_identical():
    """hipGraph replay of the update phase (catppo_graph_*): 4 iterations with on-device randomness, captured once and
    replayed, against the same run launched kernel by kernel - bit-identical parameters and optimiser state."""
    a, b = _two_trainers({"graph_update": True}, {"graph_update": False}, iters=4)
    # 2 epochs x 4 minibatches, 4 launches per optimiser step since round 6 (512-row minibatches of the 3x256 network take
    # step16_kernel + dw_multi_kernel + fold + clip/Adam; 9 with the layer-wise launches of round 4) + the gathers
    assert a.graph_update and a._graph_id is not None and a.graph_nodes >= 2 * (256 * 8 // 512) * 4
    assert not b.graph_update
    assert a.adam_step == b.adam_step == 4 * 2 * 4
    np.testing.assert_array_equal(a.agent.flat.cpu().numpy(), b.agent.flat.cpu().numpy())
    np.testing.assert_array_equal(a.exp_avg_sq.cpu().numpy(), b.exp_avg_sq.cpu().numpy())
    np.testing.assert_array_equal(a.diag.cpu().numpy(), b.diag.cpu().numpy())
    sa, sb = a.nat.iter_state_read(a.state), b.nat.iter_state_read(b.state)
    assert sa.adam_step == sb.adam_step == 32 and sa.lr == sb.lr and sa.iteration == 4


def test_device_randomness_whole_iteration_vs_oracle():
    """the default product path (Philox noise in the head kernel, keyed permutation in the gather, fused env step):
    the trainer records what it drew, the CPU oracle replays it."""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=256, num_steps=24, minibatch=1024, epochs=3, iters=2,
                                             randomness="device")
    assert trainer.rng == "device" and trainer.sink is not None
    rep = smoke_impl.compare(trainer, orc, outs[-1], tol_scale=2.0)
    print(rep)
    e = trainer.noise_rec.cpu().numpy()
    assert abs(e.mean()) < 0.02 and abs(e.std() - 1) < 0.02
    p = trainer.perm_rec.cpu().numpy()
    assert all(sorted(row.tolist()) == list(range(256 * 24)) for row in p)


def test_long_run_is_stable_and_reproducible():
    """150 iterations of the default product path (Philox noise, keyed permutations, fused env step, graph replay for
    the small minibatches) on the synthetic Solo12 stream: finite everywhere, the policy moves, the value loss falls
    below its start, and a second run from the same seed is bit-identical (no atomics, fixed-order folds, counter-based
    randomness)."""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer

    def run(n_it):
        task, env_cfg, agent_cfg = smoke_impl.make_cfgs(512, 24, 2048, 3, 150, (256, 256, 256), True, obs_dim=48,
                                                        stream_steps=48, seed=11)
        torch.manual_seed(4)
        env = make(task, cfg=env_cfg)
        tr = PPOTrainer(env, agent_cfg)
        hist = []
        for it in range(n_it):
            st = tr.run_iteration(log=(it % 25 == 0 or it == n_it - 1))
            if st is not None:
                hist.append(st)
        torch.cuda.synchronize()
        return tr, hist

    a, ha = run(150)
    assert a.graph_update and a.sink is not None and a.rng == "device"
    flat = a.agent.flat.cpu().numpy()
    assert np.isfinite(flat).all() and np.isfinite(a.exp_avg_sq.cpu().numpy()).all()
    assert all(np.isfinite(list(h.values())).all() for h in ha)
    assert ha[-1]["mean_v_loss"] < ha[0]["mean_v_loss"]                 # the critic learns the synthetic returns
    assert 0.0 < ha[-1]["learning_rate"] < ha[0]["learning_rate"]       # linear anneal (device side)
    s = a.nat.iter_state_read(a.state)
    assert s.iteration == 150 and s.adam_step == 150 * 3 * 6 == a.adam_step
    cm = a.envs.constraint_manager
    assert np.isfinite(cm.cat.get_running_maxes().cpu().numpy()).all()
    b, hb = run(150)
    np.testing.assert_array_equal(b.agent.flat.cpu().numpy(), flat)
    np.testing.assert_array_equal(b.agent.obs_rms.running_var.cpu().numpy(), a.agent.obs_rms.running_var.cpu().numpy())
    assert [h["mean_pg_loss"] for h in hb] == [h["mean_pg_loss"] for h in ha]


def test_adaptive_lr_schedule_end_to_end():
    """lr_schedule='adaptive': the learning rate is driven on the device from the per-epoch KL; it leaves its start
    value by factors of 1.5 only, and the run stays finite."""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(256, 8, 512, 4, 50, (256, 256, 256), True, obs_dim=48)
    agent_cfg.lr_schedule, agent_cfg.kl_threshold = "adaptive", 0.008
    env = make(task, cfg=env_cfg)
    tr = PPOTrainer(env, agent_cfg)
    lrs = []
    for _ in range(3):
        st = tr.run_iteration()
        lrs.append(st["learning_rate"])
    assert np.isfinite(tr.agent.flat.cpu().numpy()).all()
    for lr in lrs:
        k = np.log(lr / 3e-4) / np.log(1.5)
        assert abs(k - round(k)) < 1e-6 and 1e-6 <= lr <= 1e-2, lrs
    assert lrs[-1] != 3e-4          # 12 epochs with KL far from 0.008 at the start: the schedule moved


def test_bf16x3_whole_iteration_vs_fp32_oracle():
    """mlp_precision='bf16x3' (split-bf16 operands on the bf16 matrix pipe) through whole cfg2-shaped iterations against
    the FP32 oracle.  Masks stay bit-exact (the CaT path does not touch the MLP arithmetic); values / advantages within
    1e-4 (fp32-MFMA path: 4e-5), log-probs within 1e-2 (they amplify an error in the mean by |a - mu| / sigma^2 over 12
    action dimensions), parameters within 5e-4 after 18 optimiser steps (fp32-MFMA path: 4e-4).  A 16-bit-operand mode:
    ~10x the fp32 path's error, ~30x below what TF32 (the reference's own GEMM arithmetic on NVIDIA) would give."""
    import smoke_impl
    trainer, orc, outs = smoke_impl.run_pair(num_envs=1024, num_steps=24, minibatch=4096, epochs=3, iters=2,
                                             hidden=(256, 256, 256), six_terms=True, obs_dim=48,
                                             agent_overrides={"mlp_precision": "bf16x3"})
    assert trainer.agent.shape.mfma_bf16 == 2
    rep = smoke_impl.compare(trainer, orc, outs[-1], check=False)
    print(rep)
    assert rep["rewards"] == 0.0 and rep["dones"] == 0.0, rep
    assert rep["values"] < 1e-4 and rep["advantages"] < 1e-4 and rep["returns"] < 1e-4, rep
    assert rep["logprobs"] < 1e-2 and rep["params"] < 5e-4, rep


def test_rl_games_experience_buffer_contract_and_returns():
    """CaTExperienceBuffer driven the way CaTA2CAgent.play_steps drives it (rl_games/cat_common.py:35-112): float dones
    stored per step, value_bootstrap reward shaping with time-outs, discount_values on the buffer planes,
    swap_and_flatten01 of the listed tensors - against the oracle's restatement."""
    from cat_envs.tasks.utils.rl_games import (CaTExperienceBuffer, bootstrap_time_outs, discount_values,
                                               swap_and_flatten01)
    T, N, D, A = 24, 512, 45, 12
    class Box:                                       # gym.spaces.Box stand-in: only .shape is read
        def __init__(self, *shape):
            self.shape = shape
    buf = CaTExperienceBuffer({"observation_space": Box(D), "action_space": Box(A), "agents": 1, "value_size": 1},
                              {"num_actors": N, "horizon_length": T, "has_central_value": False,
                               "use_action_masks": False}, "cuda")
    assert buf.tensor_dict["dones"].dtype == torch.float32 and buf.tensor_dict["dones"].shape == (T, N)
    assert buf.tensor_dict["obses"].shape == (T, N, D) and buf.tensor_dict["values"].shape == (T, N, 1)
    x = S.gae_inputs(77, T, N)
    rs = np.random.RandomState(5)
    obs = rs.standard_normal((T, N, D)).astype(np.float32)
    time_outs = rs.rand(T, N) < 0.05
    dones = torch.ones(N, device="cuda")             # rl_games starts with dones = 1
    dones_seq = np.concatenate([np.ones((1, N), np.float32), x["dones"][:-1]])
    shaped = np.zeros((T, N), np.float32)
    for n in range(T):
        buf.update_data("obses", n, dev(obs[n]))
        buf.update_data("dones", n, dones)
        buf.update_data("values", n, dev(x["values"][n])[:, None])
        rew = dev(x["rewards"][n]).clone()[:, None]
        bootstrap_time_outs(rew, dev(x["values"][n])[:, None], dev(time_outs[n]), 0.99)
        buf.update_data("rewards", n, rew)
        shaped[n] = PO.value_bootstrap(torch.from_numpy(x["rewards"][n]), torch.from_numpy(x["values"][n]),
                                       torch.from_numpy(time_outs[n]), 0.99).numpy()
        dones = dev(x["dones"][n])                   # float termination probability of this step
    np.testing.assert_array_equal(buf.tensor_dict["dones"].cpu().numpy(), dones_seq)
    np.testing.assert_array_equal(buf.tensor_dict["rewards"].cpu().numpy()[..., 0], shaped)
    adv = discount_values(dones, dev(x["next_value"])[:, None], buf.tensor_dict["dones"], buf.tensor_dict["values"],
                          buf.tensor_dict["rewards"], 0.99, 0.95)
    a, r = PO.gae_rl_games(torch.from_numpy(x["dones"][-1]), torch.from_numpy(x["next_value"]),
                           torch.from_numpy(dones_seq), torch.from_numpy(x["values"]), torch.from_numpy(shaped), 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy()[..., 0], a.numpy())
    batch = buf.get_transformed_list(swap_and_flatten01, ["obses", "dones", "values", "not_there"])
    assert sorted(batch) == ["dones", "obses", "values"] and batch["obses"].shape == (N * T, D)
    np.testing.assert_array_equal(batch["obses"].cpu().numpy(), obs.transpose(1, 0, 2).reshape(N * T, D))
    assert set(buf.get_transformed(lambda t: t)) == set(buf.tensor_dict)
    small = CaTExperienceBuffer.from_shapes(8, 4, (3,), 2)
    assert small.tensor_dict["actions"].shape == (4, 8, 2)


def test_skrl_kl_adaptive_scheduler_class(nat):
    """cat_envs.tasks.utils.skrl.KLAdaptiveLR.step(diag): the call site of skrl/ppo.py:558-567 as three device launches"""
    from cat_envs import native
    from cat_envs.tasks.utils.skrl import KLAdaptiveLR
    st = nat.iter_state_new(3, 5e-4)
    nat.iter_begin(st, 5e-4, 10, native.LR_KEEP)
    sch = KLAdaptiveLR(st, kl_threshold=0.01)
    diag = torch.zeros(8, device="cuda")
    lr = 5e-4
    for epoch, kl in enumerate([0.001, 0.001, 0.03, 0.0075, 0.5]):
        diag[4] += kl * 4
        diag[7] += 4
        sch.step(diag)
        if kl > 0.02:
            lr = max(lr / 1.5, 1e-6)
        elif kl < 0.005:
            lr = min(lr * 1.5, 1e-2)
        assert abs(sch.get_last_lr()[0] - lr) < 1e-12, (epoch, sch.get_last_lr(), lr)


# ------------------------------------------------------------------------------------------ RCCL under the C ABI
def test_rccl_c_abi_world_of_one_and_graph_capture():
    """catppo_comm_unique_id / _init / catppo_allreduce / _broadcast on a world of size 1 (the only world a one-GPU
    box has), and an all-reduce captured in a hipGraph together with a kernel."""
    import subprocess
    import sys
    import os
    code = (
        "import torch, numpy as np\n"
        "torch.cuda.set_device(0)\n"
        "from cat_envs import native\n"
        "nat = native.get(torch.device('cuda', 0))\n"
        "assert nat.comm_world == 0\n"
        "uid = nat.comm_unique_id(); assert len(uid) == 128\n"
        "nat.comm_init(0, 1, uid); assert nat.comm_world == 1\n"
        "for dt in (torch.float32, torch.float64):\n"
        "    t = torch.arange(1000, device='cuda', dtype=dt); ref = t.clone()\n"
        "    nat.allreduce(t, native.SUM); nat.allreduce(t, native.MAX); nat.broadcast(t, 0)\n"
        "    torch.cuda.synchronize(); assert torch.equal(t, ref)\n"
        "s = torch.cuda.Stream()\n"
        "g = torch.ones(4096, device='cuda'); m = torch.zeros(4096, device='cuda'); v = torch.zeros(4096, device='cuda')\n"
        "p = torch.zeros(4096, device='cuda'); st = nat.iter_state_new(1, 1e-2)\n"
        "torch.cuda.synchronize()\n"
        "with torch.cuda.stream(s):\n"
        "    nat.graph_begin()\n"
        "    nat.allreduce(g, native.SUM)\n"
        "    nat.clip_adam_dev(p, g, m, v, 4096, 1e9, 0.9, 0.999, 1e-5, st)\n"
        "    gid, nn = nat.graph_end()\n"
        "    assert nn >= 2, nn\n"
        "    for _ in range(3): nat.graph_launch(gid)\n"
        "torch.cuda.synchronize()\n"
        "assert nat.iter_state_read(st).adam_step == 3\n"
        "ref = torch.zeros(4096, requires_grad=True); opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-5)\n"
        "for _ in range(3):\n"
        "    ref.grad = torch.ones(4096); opt.step()\n"
        "np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-9)\n"
        "nat.graph_destroy(gid); nat.comm_destroy(); assert nat.comm_world == 0\n"
        "print('RCCL-ABI-OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([root, os.path.join(root, "constraints-as-terminations_amd")]))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ABI-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
