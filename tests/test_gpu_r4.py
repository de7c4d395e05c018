"""Round-4 GPU tests: the deployment forward on the device against the reference's own output, gradient buckets reduced
beside the backward pass (ABI 0.4), the reported eager fallback of a failed graph capture."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import streams as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_agent_forward_on_the_device_vs_reference_golden(golden):
    """``Agent.forward(x)`` = observation normaliser (update=False) + deterministic mean (reference cleanrl/ppo.py:121-123,
    what scripts/clean_rl/play.py:140-144 calls), on the device, against ``agent.npz["deterministic"]`` - the reference
    ``Agent`` itself on these weights / inputs / normaliser state (gen_golden.gen_agent)."""
    from cat_envs.tasks.utils.cleanrl.ppo import Agent
    g = golden("agent")
    d, a = int(g["obs_dim"]), int(g["act_dim"])

    class _Space:
        def __init__(self, shape):
            self.shape = shape

    class _Env:
        num_envs = 96
        single_observation_space = {"policy": _Space((d,))}
        single_action_space = _Space((a,))

        @property
        def unwrapped(self):
            return self

    ag = Agent(_Env()).to("cuda")
    w = S.agent_weights(int(g["weight_seed"]), d, a)
    sd = ag.state_dict()
    ag.load_state_dict({k: (torch.from_numpy(w[k]) if k in w else v) for k, v in sd.items()})
    rs = np.random.RandomState(int(g["input_seed"]))
    x = rs.standard_normal((96, d)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    ag.obs_rms(torch.from_numpy(x * 2 + 1).cuda())          # the golden run moved the normaliser off identity the same way
    got = ag(xd)                                            # forward(): deterministic=True
    assert got.shape == (96, a)
    err = float(np.abs(got.cpu().numpy().astype(np.float64) - g["deterministic"]).max())
    import parity_record
    parity_record.record("agent_forward_deterministic_vs_reference_golden", {"actions": err},
                         sizes=dict(rows=96, obs_dim=d), seed=int(g["input_seed"]))
    assert err <= 1e-5, err                                 # north_star's fp32 bar
    # the normaliser was NOT updated by forward()
    cnt = float(ag.obs_rms.count)
    ag(xd)
    assert float(ag.obs_rms.count) == cnt == 97.0


_OVERLAP_CODE = r"""
import os, sys, torch, numpy as np
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['TEST_PORT'], RANK='0', WORLD_SIZE='1')
import smoke_impl
from cat_envs import parallel
parallel.init_rendezvous(0)          # gloo rendezvous: libcatppo's communicator is the only RCCL communicator of the process
assert torch.distributed.get_backend() == 'gloo'
from cat_envs.shim import make
from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
assert parallel.active()
res = {}
for tag, over in (("off_eager", dict(grad_overlap=False, graph_update=False)),
                  ("on_eager", dict(grad_overlap=True, graph_update=False)),
                  ("on_graph", dict(grad_overlap=True, graph_update=True))):
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(512, 8, 1024, 2, 50, (256, 256, 256), True, obs_dim=48, seed=7)
    for k, v in over.items():
        setattr(agent_cfg, k, v)
    torch.manual_seed(3)
    tr = PPOTrainer(make(task, cfg=env_cfg), agent_cfg)
    assert parallel.native_comm_active() and tr.nat.comm_world == 1
    assert tr.grad_overlap == over["grad_overlap"] == tr.nat.grad_overlap_active, (tag, tr.grad_overlap)
    for _ in range(3):
        tr.run_iteration(log=False)
    torch.cuda.synchronize()
    assert tr.graph_update == over["graph_update"] and tr.graph_fallback is None
    res[tag] = (tr.agent.flat.cpu().numpy().copy(), tr.exp_avg_sq.cpu().numpy().copy(), tr.diag.cpu().numpy().copy(),
                tr.graph_nodes)
for tag in ("on_eager", "on_graph"):
    for i in range(3):
        np.testing.assert_array_equal(res[tag][i], res["off_eager"][i], err_msg=f"{tag}[{i}]")
assert np.abs(res["off_eager"][0]).sum() > 0 and res["on_graph"][3] > 0
parallel.shutdown_native_comm()
torch.distributed.destroy_process_group()
print("OVERLAP-OK", res["on_graph"][3])
"""


def _run_code(code, **env_extra):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CATPPO_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TEST_PORT=str(port),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")]), **env_extra)
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


def test_gradient_buckets_reduced_beside_the_backward_pass_world_of_one():
    """catppo_set_grad_overlap (ABI 0.4): per-layer fold + grouped RCCL all-reduce on the side stream inside
    catppo_ppo_minibatch_grad_packed, eager and captured in the update-phase graph, on a world of one (all a one-GPU box
    has): parameters, Adam state and diagnostics BIT-identical to the single fold launch + one all-reduce."""
    r = _run_code(_OVERLAP_CODE)
    assert r.returncode == 0 and "OVERLAP-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_FALLBACK_CODE = r"""
import os, sys, torch, numpy as np
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['TEST_PORT'], RANK='0', WORLD_SIZE='1')
import smoke_impl
from cat_envs import parallel
parallel.init_rendezvous(0)          # gloo rendezvous: libcatppo's communicator is the only RCCL communicator of the process
assert torch.distributed.get_backend() == 'gloo'
from cat_envs.shim import make
from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
out = []
for broken in (False, True):
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(256, 8, 512, 2, 50, (256, 256, 256), True, obs_dim=48, seed=7)
    agent_cfg.graph_update = True
    torch.manual_seed(3)
    tr = PPOTrainer(make(task, cfg=env_cfg), agent_cfg)
    if broken:                # the first replay fails (as an unsupported capture on some node would): eager from then on
        real = tr.nat.graph_launch
        def bad_launch(gid):
            raise RuntimeError("libcatppo error -2: catppo_graph_launch: injected failure")
        tr.nat.graph_launch = bad_launch
    for _ in range(3):
        tr.run_iteration(log=False)
    torch.cuda.synchronize()
    if broken:
        tr.nat.graph_launch = real
        assert tr.graph_update is False and "injected failure" in tr.graph_fallback, tr.graph_fallback
    else:
        assert tr.graph_update is True and tr.graph_fallback is None
    assert tr.adam_step == 3 * 2 * 4, tr.adam_step
    out.append(tr.agent.flat.cpu().numpy().copy())
np.testing.assert_array_equal(out[0], out[1])
parallel.shutdown_native_comm()
torch.distributed.destroy_process_group()
print("FALLBACK-OK")
"""


def test_failed_graph_replay_falls_back_to_eager_launches_and_says_so():
    """when the capture or the first replay of the update-phase graph fails in an env-sharded run
    the update phase continues eagerly (bit-identical result) and the reason is kept in ``PPOTrainer.graph_fallback``
    (bench.py prints it).  (Round 5: with REAL peers graph + collectives is opt-in, ``CATPPO_GRAPH_COMM=1``; a forced
    world of one - this test - keeps it on.)"""
    r = _run_code(_FALLBACK_CODE)
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "falling back to eager launches" in r.stderr


# ------------------------------------------------------------------------------------------ row-resident forward
@pytest.mark.parametrize("D,A,hidden,Bsz,M", [
    (48, 12, (256, 256, 256), 16384, 16384),      # cfg2's minibatch: 256 row tiles, one workgroup walks both networks
    (48, 12, (256, 256, 256), 8192, 4133),        # ragged, 65 tiles: one workgroup per (tile, network), last tile 37 rows
    (48, 12, (256, 256, 256), 16421, 16421),      # ragged AND >= 256 tiles: both networks per workgroup, 37-row last tile
    (235, 12, (256, 256, 256), 8192, 8192),       # cfg4: 240-wide padded observations (7.5 slabs of 32)
    (48, 12, (256, 256), 4160, 4160),             # two hidden layers: only the first one is computed by the new launch
    (45, 5, (256, 256, 256, 128), 4096, 4096),    # three fused layers below a 128-wide last layer
])
def test_row_resident_forward_equals_the_layerwise_launches(tmp_path, D, A, hidden, Bsz, M):
    """rows_fwd_kernel<64> (fwd_rows.h: the hidden layers below the last one in ONE launch, activation tile resident in
    LDS, stored for the backward) against the layer-wise GEMM launches it replaces (cleanrl/ppo.py:78-96 forward): the
    contraction order per element is the same, so activations - and with them the whole minibatch gradient and the
    diagnostics - must be BIT-identical.  Two processes (the switch is read once per process)."""
    import test_gpu_kernels as TK
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"rows{flag}.npz")
        code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=0, out=out)
        env = dict(os.environ, CATPPO_ROWS_FWD=flag, CATPPO_ROWS_FWD_MIN_ROWS="1", CATPPO_STEP16="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.abs(outs[1]["grad"]).max() > 0
    np.testing.assert_array_equal(outs[0]["grad"], outs[1]["grad"])
    np.testing.assert_array_equal(outs[0]["diag"], outs[1]["diag"])


def test_row_resident_rollout_forward_equals_the_layerwise_path(tmp_path):
    """rows_fwd_kernel<32> with heads (CATPPO_ROWS_FWD_ROLLOUT=1) against the layer-wise rollout forward: hidden layers
    bit-identical by construction, heads sum in another order (2e-6), Philox noise exact."""
    import test_gpu_kernels as TK
    cases = {"cfg2": (48, 12, (256, 256, 256), 4096), "ragged": (45, 12, (256, 256, 256), 2049),
             "tiny": (48, 7, (256, 256), 33), "wide_obs": (235, 12, (256, 256, 256), 300), "one_row": (48, 12, (256,), 1)}
    outs = []
    for env_over in (dict(CATPPO_ROWS_FWD_ROLLOUT="1", CATPPO_FUSED_FWD_MIN_ROWS="1", CATPPO_STEP16_FWD="0"),
                     dict(CATPPO_ROWS_FWD_ROLLOUT="0", CATPPO_FUSED_FWD="0")):
        out = str(tmp_path / f"rr{len(outs)}.npz")
        code = TK._FUSED_FWD_AB.format(root=ROOT, cases=cases, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_over), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    f, l = outs
    for k in f.files:
        if k.endswith("_e3"):
            np.testing.assert_array_equal(f[k], l[k], err_msg=k)
        else:
            np.testing.assert_allclose(f[k], l[k], rtol=0, atol=2e-6 * max(1.0, float(np.abs(l[k]).max())), err_msg=k)
    assert np.abs(f["cfg2_act"]).max() > 0 and np.isfinite(f["ragged_lp"]).all()


@pytest.mark.parametrize("D,A,hidden,Bsz,M", [
    (48, 12, (256, 256, 256), 16384, 16384),      # cfg2
    (45, 12, (512, 256, 128), 4133, 2048),        # reference shapes, small ragged minibatch (head_loss path)
])
def test_first_layer_weight_gradient_launch_carrying_the_fold_is_bit_identical(tmp_path, D, A, hidden, Bsz, M):
    """dw_fold_kernel (the first layer's weight-gradient GEMM + the fold of every other layer's partials in one launch)
    against the separate GEMM and the single fold launch: same partials, same fold order per element - the flat gradient
    and the diagnostics must be BIT-identical.  Two processes (CATPPO_DW0_FOLD is read once)."""
    import test_gpu_kernels as TK
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"dwfold{flag}.npz")
        code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=0, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CATPPO_DW0_FOLD=flag), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.abs(outs[1]["grad"]).max() > 0
    np.testing.assert_array_equal(outs[0]["grad"], outs[1]["grad"])
    np.testing.assert_array_equal(outs[0]["diag"], outs[1]["diag"])


@pytest.mark.parametrize("D", [5, 17, 50, 70, 100, 250])
def test_row_resident_forward_every_slab_count_of_the_first_layer(tmp_path, D):
    """the first layer's contraction length decides which pieces of the slab loop run (padded widths 16 / 32 / 64 / 80 /
    112 / 256 = 1, 1, 2, 3, 4, 8 slabs of 32 k, the last one 2 or 4 blocks long; peeled first refill, refill loop, last but
    one, last): whole minibatch gradient bit-identical to the layer-wise launches at every one of them (4101 ragged rows)"""
    import test_gpu_kernels as TK
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"rows{flag}.npz")
        code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=D, A=12, hidden=(256, 256, 256), Bsz=4101, M=4101, prec=0, out=out)
        env = dict(os.environ, CATPPO_ROWS_FWD=flag, CATPPO_ROWS_FWD_MIN_ROWS="1", CATPPO_STEP16="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.abs(outs[1]["grad"]).max() > 0
    np.testing.assert_array_equal(outs[0]["grad"], outs[1]["grad"])
    np.testing.assert_array_equal(outs[0]["diag"], outs[1]["diag"])


def test_in_launch_fold_tree_of_the_env_step_still_holds_the_fused_step_tests():
    """The default since round 4 folds rollout_pre's partial rows in a launch of its own (rollout_fold_kernel), which the
    fused-vs-unfused tests of test_gpu_r2_features.py exercise as they are; the rounds 2-3 tree inside the launch
    (CATPPO_ROLLOUT_TREE=1, read once per process) is held to the same tests here."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_r2_features.py"), "-k", "fused_rollout"],
                       env=dict(os.environ, CATPPO_ROLLOUT_TREE="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])
    assert "2 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("kw", [
    dict(),                                                                             # 3 x 256: dw_fold + final fold launch
    dict(num_envs=1000, num_steps=4, minibatch=1000, epochs=1, iters=2, hidden=(512, 256, 128), obs_dim=235),
    dict(num_envs=64, num_steps=8, minibatch=128, epochs=2, iters=2, six_terms=True),  # small minibatch (head_loss path)
])
def test_one_call_optimiser_step_equals_gradient_call_plus_clip_adam_call(kw):
    """catppo_ppo_minibatch_step_packed (the fold launches emit the squared norm of the clip) against
    catppo_ppo_minibatch_grad_packed + catppo_clip_adam_dev (one launch more) over whole iterations with the same noise and
    permutations: the clipped gradient, both Adam moments and the parameters can differ only through the summation
    order of the fp64 squared norm (<= 1 ulp of the fp32 clip coefficient), the step count not at all."""
    import test_gpu_r2_features as R2
    a, b = R2._two_trainers({"one_call_step": True}, {"one_call_step": False}, inject=True, **kw)
    assert a.one_call_step and not b.one_call_step
    sa, sb = a.nat.iter_state_read(a.state), b.nat.iter_state_read(b.state)
    assert sa.adam_step == sb.adam_step > 0 and sa.adam_step_size == sb.adam_step_size
    for name in ("grad", "exp_avg", "exp_avg_sq"):
        x, y = getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()
        np.testing.assert_allclose(x, y, rtol=3e-7, atol=0, err_msg=name)
    pa, pb = a.agent.flat.cpu().numpy(), b.agent.flat.cpu().numpy()
    np.testing.assert_allclose(pa, pb, rtol=0, atol=2e-7)
    assert np.abs(a.grad.cpu().numpy()).max() > 0 and np.isfinite(pa).all()
    for name in ("values", "logprobs", "advantages"):
        np.testing.assert_allclose(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy(), rtol=0, atol=2e-6,
                                   err_msg=name)

