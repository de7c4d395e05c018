"""bf16-operand MFMA mode of the actor-critic GEMMs (BASELINE config 5: "bf16 MLP MFMA"; `catppo_mlp_shape.mfma_bf16`).

Not a reference code path (the reference's CleanRL trainer is fp32), so there is no golden vector.  The checker is
the oracle's restatement of the mode - both operands of every hidden-layer GEMM rounded to bf16 (RNE), fp32
accumulation, in the forward, the data gradient and the weight gradient - plus the fp32 oracle at bf16-level
tolerance.  Agreement with the bf16 restatement cannot be bit-level: an fp32 activation that differs in its last
bit between the two summation orders can fall on the other side of a bf16 rounding boundary (a 2^-8 relative step
in ONE of the 128..512 products of a dot product), hence a small absolute tolerance next to a tight mean error."""
import math

import numpy as np
import pytest
import torch

import streams as S
from oracle import ppo_oracle as PO
from test_gpu_kernels import _minibatch_case, dev, flat_params, unflatten_grad

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from cat_envs import native
    return native.Native()


@pytest.mark.parametrize("D,A,hidden,B", [(48, 12, (256, 256, 256), 4096), (45, 12, (512, 256, 128), 1000)])
def test_bf16_policy_act_vs_bf16_operand_oracle(nat, D, A, hidden, B):
    from cat_envs import native
    shape = native.shape_of(D, A, hidden, mfma_bf16=True)
    lay = native.layout_of(shape)
    w = S.agent_weights(3, D, A, hidden)
    ag = PO.AgentOracle(D, A, hidden, bf16_hidden=True)
    ag.load(w)
    ag32 = PO.AgentOracle(D, A, hidden)
    ag32.load(w)
    params = flat_params(native, shape, lay, w)
    rs = np.random.RandomState(4)
    x = rs.standard_normal((B, D)).astype(np.float32)
    eps = rs.standard_normal((B, A)).astype(np.float32)
    xp = np.zeros((B, lay.obs_pad), np.float32)
    xp[:, :D] = x
    act, logp, val = torch.empty(B, A, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    nat.mlp_reserve(shape, B)
    nat.policy_act(shape, params, dev(xp), B, dev(eps), act, logp, val)
    torch.cuda.synchronize()
    with torch.no_grad():
        a, lp, _, v = ag.get_action_and_value(torch.from_numpy(x), eps=torch.from_numpy(eps))
        a32, _, _, v32 = ag32.get_action_and_value(torch.from_numpy(x), eps=torch.from_numpy(eps))
    dv = np.abs(val.cpu().numpy() - v.numpy()[:, 0])
    da = np.abs(act.cpu().numpy() - a.numpy())
    assert dv.max() < 5e-3 and dv.mean() < 2e-5, (dv.max(), dv.mean())
    assert da.max() < 5e-3 and da.mean() < 2e-5, (da.max(), da.mean())     # rare boundary flips, tiny mean
    # the mode really rounds: it is measurably away from the fp32 network, by a bf16-sized amount
    d32 = np.abs(val.cpu().numpy() - v32.numpy()[:, 0])
    assert 1e-4 < d32.mean() < 3e-2, d32.mean()
    # and the fp32 mode on the same buffers is unaffected by the flag's existence
    shape32 = native.shape_of(D, A, hidden)
    nat.policy_act(shape32, params, dev(xp), B, dev(eps), act, logp, val)
    np.testing.assert_allclose(val.cpu().numpy(), v32.numpy()[:, 0], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("D,A,hidden,Bsz,M", [(48, 12, (256, 256, 256), 8192, 4096),
                                               (45, 12, (512, 256, 128), 32768, 16384)])
def test_bf16_minibatch_grad_vs_bf16_operand_autograd(nat, D, A, hidden, Bsz, M):
    from cat_envs import native
    shape = native.shape_of(D, A, hidden, mfma_bf16=True)
    lay = native.layout_of(shape)
    w = S.agent_weights(5, D, A, hidden)
    c = _minibatch_case(D, A, hidden, Bsz, M, 6)
    grads = {}
    for name, bf in (("bf16", True), ("fp32", False)):
        ag = PO.AgentOracle(D, A, hidden, bf16_hidden=bf)
        ag.load(w)
        ag.value_rms.mean, ag.value_rms.var = torch.tensor(float(c["vmean"])), torch.tensor(float(c["vvar"]))
        if name == "bf16":
            with torch.no_grad():
                _, lp0, _, _ = ag.get_action_and_value(torch.from_numpy(c["obs"]), torch.from_numpy(c["act"]))
            rs = np.random.RandomState(7)
            c["logp"] = (lp0.numpy() + rs.standard_normal(Bsz).astype(np.float32) * 0.25).astype(np.float32)
        cfg = dict(clip_coef=0.2, ent_coef=0.001, vf_coef=2.0, norm_adv=True, clip_vloss=True)
        [p.requires_grad_(True) for p in ag.parameters()]
        mb = torch.from_numpy(c["inds"])
        loss, st = PO.ppo_minibatch_loss(ag, torch.from_numpy(c["obs"])[mb], torch.from_numpy(c["act"])[mb],
                                         torch.from_numpy(c["logp"])[mb], torch.from_numpy(c["adv"])[mb],
                                         torch.from_numpy(c["ret"])[mb], torch.from_numpy(c["val"])[mb], cfg)
        loss.backward()
        grads[name] = ({k: v.grad.numpy() for k, v in ag.p.items()}, st)

    params = flat_params(native, shape, lay, w)
    obs_p = np.zeros((Bsz, lay.obs_pad), np.float32)
    obs_p[:, :D] = c["obs"]
    grad = torch.zeros(lay.n_flat, device="cuda")
    diag = torch.zeros(8, device="cuda")
    hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / M, 0)
    nat.mlp_reserve(shape, M)
    nat.ppo_minibatch_grad(shape, hp, params, dev(obs_p), dev(c["act"]), dev(c["logp"]), dev(c["adv"]), dev(c["ret"]),
                           dev(c["val"]), dev(c["inds"]), dev(np.array([c["vmean"]])), dev(np.array([c["vvar"]])),
                           None, grad, diag)
    torch.cuda.synchronize()
    got = unflatten_grad(shape, lay, grad.cpu().numpy(), w)
    assert np.isfinite(grad.cpu().numpy()).all()

    def flat(gd):
        return np.concatenate([np.asarray(gd[k], np.float64).reshape(-1) for k in sorted(grads["bf16"][0])])

    g_dev, g_bf, g_32 = flat({k: got[k].reshape(v.shape) for k, v in grads["bf16"][0].items()}), \
        flat(grads["bf16"][0]), flat(grads["fp32"][0])
    cos = lambda a, b: float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))  # noqa: E731
    assert cos(g_dev, g_bf) > 0.9999, cos(g_dev, g_bf)          # same arithmetic up to rare boundary flips
    assert cos(g_dev, g_32) > 0.995, cos(g_dev, g_32)           # and a faithful bf16 approximation of the fp32 gradient
    assert cos(g_dev, g_bf) > cos(g_dev, g_32)
    rel = np.linalg.norm(g_dev - g_bf) / np.linalg.norm(g_bf)
    assert rel < 1e-2, rel
    st = grads["bf16"][1]
    np.testing.assert_allclose(diag.cpu().numpy()[:4], [float(st["pg_loss"]), float(st["v_loss"]),
                                                       float(st["entropy"]), float(st["loss"])], rtol=2e-3, atol=2e-4)


def test_bf16_training_iterations_run_and_stay_close_to_fp32():
    """two full CaT-PPO iterations with mlp_precision='bf16' on the synthetic Solo12 stream: finite, parameters move,
    and stay within bf16 distance of the fp32 run fed the same noise / permutations"""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer

    def run(precision):
        task, env_cfg, agent_cfg = smoke_impl.make_cfgs(256, 8, 512, 2, 4, (256, 256, 256), True, obs_dim=48,
                                                        stream_steps=16, seed=3)
        agent_cfg.mlp_precision = precision
        env = make(task, cfg=env_cfg)
        torch.manual_seed(11)
        tr = PPOTrainer(env, agent_cfg)
        g = torch.Generator(device="cuda").manual_seed(5)
        gp = torch.Generator(device="cuda").manual_seed(6)
        p0 = tr.agent.flat.clone()
        for _ in range(2):
            tr.run_iteration(eps_fn=lambda step: torch.randn(256, 12, device="cuda", generator=g),
                             perm_fn=lambda e: torch.randperm(256 * 8, device="cuda", generator=gp), log=False)
        torch.cuda.synchronize()
        return p0.cpu().numpy(), tr.agent.flat.cpu().numpy(), tr.diag.cpu().numpy()

    p0, p_bf, d_bf = run("bf16")
    p0b, p_32, d_32 = run("fp32")
    np.testing.assert_array_equal(p0, p0b)
    assert np.isfinite(p_bf).all() and np.isfinite(d_bf).all()
    step = np.abs(p_32 - p0).max()
    assert step > 1e-4                                   # the optimiser moved
    assert np.abs(p_bf - p_32).max() < 0.5 * step + 1e-3, (np.abs(p_bf - p_32).max(), step)


def _grad_in_process(tmp_path, tag, D, A, hidden, Bsz, M, **env_over):
    """one minibatch gradient of the bf16-operand mode in a fresh process (the switches are read once per process)"""
    import os
    import subprocess
    import sys
    import test_gpu_kernels as TK
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"{tag}.npz")
    code = TK._FUSED_VS_SPLIT.format(root=root, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=1, out=out)
    code = code.replace("nat.mlp_reserve(shape, M)", "nat.mlp_reserve(shape, M); nat.plan_log(1)")
    code = code.replace("np.savez(", "open({!r}, 'w').write(nat.plan_log(-1)); np.savez(".format(out + ".plan"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_over), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), open(out + ".plan").read()


@pytest.mark.parametrize("D,A,hidden,Bsz,M", [
    (48, 12, (256, 256, 256), 16384, 16384),      # BASELINE configs[4]'s minibatch
    (45, 12, (512, 256, 128), 8192, 4133),        # reference shapes, ragged rows (last tile 37 rows, last split short)
    (235, 12, (256, 256, 256), 8192, 4096),       # wide first layer (cfg4's observations): first-layer dW on 64x64 tiles
])
def test_bf16_stored_activations_equal_the_fp32_stored_path(tmp_path, D, A, hidden, Bsz, M):
    """Round 6 ("act16", gemm_f32.h): hidden activations and dZ STORED as bf16 + bf16 weight copies, against the same mode
    with fp32-stored tensors rounded where a GEMM consumes them (rounds 2-5).  The GEMM operands are the same bf16 values
    either way and every instruction contracts the same 16 k, but a k sits in another operand slot of the instruction (k = 8 h
    + e here, 8 (i / 4) + 4 h + i % 4 there) and the matrix unit's internal summation order follows the slots: the FORWARD -
    every diagnostic of the loss - agrees to fp32 rounding (recorded: 1.2e-7 relative), not bit for bit.  The backward differs
    where the rounded tensors are used outside a GEMM: elu'(H) = H + 1 of a negative activation (2^-9 relative) and the bias
    gradients' column sums of dZ."""
    import parity_record
    new, plan_new = _grad_in_process(tmp_path, "act16", D, A, hidden, Bsz, M, CATPPO_ACT16="1")
    old, plan_old = _grad_in_process(tmp_path, "act32", D, A, hidden, Bsz, M, CATPPO_ACT16="0")
    assert "bf16-stored" in plan_new and "fwd0_w16_kernel" in plan_new and "bf16-stored" not in plan_old, plan_new
    np.testing.assert_allclose(new["diag"][:7], old["diag"][:7], rtol=1e-6, atol=0)
    g1, g0 = new["grad"].astype(np.float64), old["grad"].astype(np.float64)
    assert np.isfinite(g1).all() and np.abs(g0).max() > 0
    scale = np.abs(g0).max()
    rec = {"max_rel_to_largest": float(np.abs(g1 - g0).max() / scale),
           "mean_abs_rel_to_mean_abs": float(np.abs(g1 - g0).mean() / np.abs(g0).mean())}
    parity_record.record(f"act16_vs_fp32_stored_{D}_{'x'.join(map(str, hidden))}_{M}", rec, sizes=dict(D=D, hidden=list(hidden), M=M), seed=6)
    print(rec)
    assert rec["max_rel_to_largest"] < 4e-3 and rec["mean_abs_rel_to_mean_abs"] < 2e-2, rec
