"""Round-5 GPU tests: the exchange points folded into fewer collectives (VERDICT r4 item 4 v), the gloo rendezvous with
libcatppo's communicator as the process's only RCCL communicator (item 4 i), the simulator-state copy inside
catppo_rollout_pre against the separate copy, sign agreement of the norm-based constraint terms at scale
(item 6).  Everything goes through the C ABI (ctypes)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run_code(code, **env_extra):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CATPPO_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TEST_PORT=str(port),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "constraints-as-terminations_amd"), HERE]))
    env.update(env_extra)
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


# ------------------------------------------------------------------------------------------ advantage moments, all epochs
@pytest.mark.parametrize("total,M,E,half", [(4096 * 3, 4096, 3, False), (4133, 1000, 5, False), (2048 * 2 + 17, 2048, 2, True)])
def test_adv_moments_of_all_epochs_in_one_call_equal_the_gathers_own_partials(total, M, E, half):
    """catppo_adv_moments_keyed (ABI 0.5) = for every epoch e: catppo_ppo_gather_ex(state, e) -> catppo_adv_moments_parts,
    BIT for bit (same 64-row chunks, same wave butterfly), and = numpy on the recorded permutation to 1e-12; ragged last
    minibatch, fp16 advantage plane (cleanrl/ppo.py:314-318 across epochs)."""
    from cat_envs import native
    nat = native.get(torch.device("cuda", 0))
    dev = nat.device
    D, A = 48, 12
    shape = native.shape_of(D, A, (256, 256, 256))
    Dp = native.layout_of(shape).obs_pad
    g = torch.Generator(device="cpu").manual_seed(5)
    adv32 = torch.randn(total, generator=g).to(dev) * 3 + 0.5
    adv = adv32.half() if half else adv32
    obs, act = torch.zeros(total, Dp, device=dev), torch.zeros(total, A, device=dev)
    sc = [torch.randn(total, generator=g).to(dev) for _ in range(3)]
    st = nat.iter_state_new(1234567, 3e-4)
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
    n_mb, parts = -(-total // M), -(-M // 64)
    scratch = torch.empty(E * n_mb * parts * 2, dtype=torch.float64, device=dev)
    mom = torch.zeros(E * n_mb, 3, dtype=torch.float64, device=dev)
    nat.adv_moments_keyed(adv, st, E, total, M, scratch, mom)
    x_g, act_g, scal_g = torch.empty(total, Dp, device=dev), torch.empty(total, A, device=dev), torch.empty(4 * total, device=dev)
    advp = torch.empty(n_mb * parts * 2, dtype=torch.float64, device=dev)
    ref = torch.zeros(n_mb, 3, dtype=torch.float64, device=dev)
    inds = torch.zeros(total, dtype=torch.int64, device=dev)
    a64 = adv.double().cpu().numpy()
    for e in range(E):
        nat.ppo_gather_ex(shape, obs, act, sc[0], adv, sc[1], sc[2], total, M, x_g, act_g, scal_g, advp, inds=None, st=st,
                          epoch=e, inds_out=inds)
        nat.adv_moments_parts(advp, parts, total, M, ref)
        got = mom[e * n_mb:(e + 1) * n_mb].cpu().numpy()
        np.testing.assert_array_equal(got, ref.cpu().numpy(), err_msg=f"epoch {e}")
        p = inds.cpu().numpy()
        assert sorted(p.tolist()) == list(range(total))
        for k in range(n_mb):
            v = a64[p[k * M:(k + 1) * M]]
            np.testing.assert_allclose(got[k], [v.sum(), (v * v).sum(), len(v)], rtol=1e-12, atol=1e-9)


# ------------------------------------------------------------------------------------------ fewer collectives, same numbers
_FEWER_CODE = r"""
import os, sys, torch, numpy as np
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['TEST_PORT'], RANK='0', WORLD_SIZE='1')
import smoke_impl
from cat_envs import parallel
parallel.init_rendezvous(0)
assert torch.distributed.get_backend() == 'gloo'
from cat_envs.shim import make
from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
task, env_cfg, agent_cfg = smoke_impl.make_cfgs(300, 8, 1000, 3, 50, (256, 256, 256), True, obs_dim=48, seed=11)
agent_cfg.graph_update = os.environ.get('T_GRAPH', '0') == '1'
agent_cfg.rollout_dtype = os.environ.get('T_PLANES', 'fp32')
torch.manual_seed(3)
tr = PPOTrainer(make(task, cfg=env_cfg), agent_cfg)
assert parallel.active() and parallel.native_comm_active() and tr.nat.comm_world == 1
env = parallel.comm_env()
assert env['rendezvous_backend'] == 'gloo' and env['device_transport'] == 'libcatppo RCCL communicator', env
parallel.comm_timing_begin()
for _ in range(3):
    tr.run_iteration(log=True)
rec = parallel.comm_timing_end()
np.savez(os.environ['T_OUT'], flat=tr.agent.flat.cpu().numpy(), m2=tr.exp_avg_sq.cpu().numpy(),
         vmean=tr.agent.value_rms.running_mean.cpu().numpy(), vvar=tr.agent.value_rms.running_var.cpu().numpy(),
         adv_stats=tr._adv_stats_all.cpu().numpy(), vn=tr.values_n.cpu().numpy(), rn=tr.returns_n.cpu().numpy(),
         calls=np.array([rec['calls']]), graph=np.array([int(tr.graph_update)]))
parallel.shutdown_native_comm()
torch.distributed.destroy_process_group()
print('FEWER-OK', rec['calls'], rec['by_kind'])
"""


@pytest.mark.parametrize("planes", ["fp32", "fp16"])
def test_one_exchange_per_iteration_for_value_normaliser_and_advantage_statistics(tmp_path, planes):
    """per iteration the value normaliser exchanges ONE moment record (values and returns together)
    and the advantage statistics of all epochs ONE (before the first gather) instead of 2 + E.  Same arithmetic in the same
    order: parameters, Adam state, normaliser state, per-minibatch statistics BIT-identical to the old exchange pattern
    (CATPPO_VRMS_PAIR=0 CATPPO_ADV_UPFRONT=0), eager and from the replayed graph; and the collective count drops by
    (1 + E - 1) per iteration.  World of one through libcatppo's RCCL communicator, rendezvous on gloo."""
    outs = {}
    for tag, env in (("new", dict()), ("old", dict(CATPPO_VRMS_PAIR="0", CATPPO_ADV_UPFRONT="0")),
                     ("new_graph", dict(T_GRAPH="1"))):
        out = str(tmp_path / f"{tag}.npz")
        r = _run_code(_FEWER_CODE, T_OUT=out, T_PLANES=planes, **env)
        assert r.returncode == 0 and "FEWER-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[tag] = np.load(out)
    for k in ("flat", "m2", "vmean", "vvar", "adv_stats", "vn", "rn"):
        np.testing.assert_array_equal(outs["new"][k], outs["old"][k], err_msg=k)
        np.testing.assert_array_equal(outs["new_graph"][k], outs["old"][k], err_msg="graph " + k)
    assert np.abs(outs["new"]["flat"]).sum() > 0 and outs["new_graph"]["graph"][0] == 1
    # 3 timed iterations, E = 3 epochs: old = 2 (value normaliser) + 3 (advantages), new = 1 + 1 per iteration
    assert int(outs["old"]["calls"][0]) - int(outs["new"]["calls"][0]) == 3 * (1 + 2)


# ------------------------------------------------------------------------------------------ simulator state copy A/B
_SIMCOPY_CODE = r"""
import os, sys, torch, numpy as np
import smoke_impl
from cat_envs.shim import make
from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
task, env_cfg, agent_cfg = smoke_impl.make_cfgs(777, 12, 2048, 2, 50, (256, 256, 256), False, obs_dim=45, seed=5)
torch.manual_seed(3)
env = make(task, cfg=env_cfg)
tr = PPOTrainer(env, agent_cfg)
for _ in range(2):
    tr.run_iteration(log=False)
torch.cuda.synchronize()
eu = env.unwrapped
np.savez(os.environ['T_OUT'], flat=tr.agent.flat.cpu().numpy(), dones=tr.dones.float().cpu().numpy(),
         rewards=tr.rewards.float().cpu().numpy(), obs=tr.obs.cpu().numpy(), act=tr.actions.cpu().numpy(),
         values=tr.values.float().cpu().numpy(), cstr=eu.constraint_manager.cat._p_cstr.cpu().numpy(),
         state=eu.sim.cur.cpu().numpy(), ep_len=eu.episode_length_buf.cpu().numpy())
print('SIMCOPY-OK', tr.sink is not None)
"""


def test_simulator_state_advance_inside_rollout_pre_equals_the_separate_copy(tmp_path):
    """catppo_rollout_step.sim_src (the state advance inside catppo_rollout_pre: every input re-based onto the
    stream slab, rows copied by the tile's workgroup; reference cat/cat_env.py:60-90 `scene.update`) against the separate
    copy kernel (CATPPO_FUSED_SIM_COPY=0): two whole iterations, everything the step produces BIT-identical."""
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"sim{flag}.npz")
        r = _run_code(_SIMCOPY_CODE, T_OUT=out, CATPPO_FUSED_SIM_COPY=flag, CATPPO_FORCE_DIST="0")
        assert r.returncode == 0 and "SIMCOPY-OK True" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    for k in outs[0].files:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    assert np.abs(outs[0]["flat"]).sum() > 0 and outs[0]["dones"].max() > 0


def test_rollout_pre_refuses_outputs_inside_the_simulator_state_block():
    """with sim_src set, an OUTPUT pointer of the step inside [sim_state, sim_state + N * row_bytes) would race
    with the row copy: CATPPO_E_ARG with a message instead of a silently overwritten buffer."""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    task, env_cfg, agent_cfg = smoke_impl.make_cfgs(64, 4, 128, 1, 5, (256, 256), True, obs_dim=48, seed=3)
    env = make(task, cfg=env_cfg)
    tr = PPOTrainer(env, agent_cfg)
    tr.run_iteration(log=False)                        # fills the argument block of the fused step
    torch.cuda.synchronize()
    eu = env.unwrapped
    st = eu._rstep
    assert tr.sink is not None and st is not None
    if not st.sim_src:
        pytest.skip("the env does not hand the state advance to catppo_rollout_pre (CATPPO_FUSED_SIM_COPY=0)")
    nat = tr.nat
    assert nat.lib.catppo_rollout_pre(nat.h, eu._rstep_ref, nat._stream()) == 0       # untouched block: accepted
    torch.cuda.synchronize()
    keep = st.reward
    st.reward = st.sim_state + 64                      # an output inside the state block
    rc = nat.lib.catppo_rollout_pre(nat.h, eu._rstep_ref, nat._stream())
    st.reward = keep
    assert rc != 0
    with pytest.raises(RuntimeError, match="inside the simulator"):
        nat._ok(rc)


# ------------------------------------------------------------------------------------------ norm terms at their limits
def test_norm_based_terms_at_their_limits_at_scale_vs_reference_golden(golden):
    """C8 `base_orientation` / C13 `foot_contact_force` (cat/constraints.py:113-119,201-211) and what a
    norm gates (C7 contact, C9 air_time, C10 n_foot_contact, C15 no_move) at 4096 envs x 4 steps with thousands of norms
    planted exactly at / within a few ulps of their limit (||F|| == 50.0, ||g_xy|| == 0.1, ||cmd|| at the dead-zone):
    the device output is BIT-identical to the reference's (sha256 of the raw fp32 bytes of every step, step 0 element for
    element), so the violation masks agree on every element - including the ~16 000 constraints that are exactly 0."""
    import hashlib
    import parity_record
    import streams as S
    import test_gpu_kernels as TK
    from cat_envs import native
    nat = native.get(torch.device("cuda", 0))
    g = golden("terms_scale")
    n, steps = int(g["n_envs"]), int(g["steps"])
    states = S.sim_state_at_the_limits(int(g["seed"]), n, steps)
    assert S.checksum(*[np.asarray(v) for st in states for v in st.values()]) == str(g["input_checksum"])
    feet, upper = [3, 6, 9, 12], [0, 2, 5, 8, 11]
    want = ("base_orientation", "foot_contact_force", "contact", "air_time", "n_foot_contact", "no_move")
    rec = {k: dict(elements=0, not_bit_equal_step0=0, sign_disagreements=0, exactly_zero=0) for k in want}
    for k, st in enumerate(states):
        sd = {key: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32, device="cuda")
              for key, v in st.items() if isinstance(v, np.ndarray)}
        named = TK._term_descs(native, sd, feet, upper)
        K = sum(t.width for _, t in named)
        cstr = torch.full((n, K), 123.0, device="cuda")
        nat.cat_terms([t for _, t in named], n, sd["net_forces_w_history"], 3, 13, sd["command"], cstr)
        torch.cuda.synchronize()
        out = cstr.cpu().numpy()
        c = 0
        for name, t in named:
            got = np.ascontiguousarray(out[:, c:c + t.width])
            c += t.width
            if name not in want:
                continue
            r = rec[name]
            r["elements"] += got.size
            r["exactly_zero"] += int((got == 0).sum())
            mask_ref = np.unpackbits(g[name + "_mask_bits"][k])[:got.size].astype(bool).reshape(got.shape)
            r["sign_disagreements"] += int(((got > 0) != mask_ref).sum())
            if k == 0:
                r["not_bit_equal_step0"] += int((got != g[name + "_step0"]).sum())
                np.testing.assert_array_equal(got, g[name + "_step0"], err_msg=name)
            np.testing.assert_array_equal(got > 0, mask_ref, err_msg=f"{name}: violation mask, step {k}")
            assert hashlib.sha256(got.tobytes()).hexdigest() == str(g[name + "_sha256"][k]), (name, k)
    parity_record.record("norm_terms_at_their_limits_4096x4_vs_reference_golden",
                         {f"{name}.{key}": v for name, r in rec.items() for key, v in r.items()}, sizes=dict(n_envs=n, steps=steps),
                         seed=int(g["seed"]))
    assert rec["base_orientation"]["exactly_zero"] > 4000 and rec["foot_contact_force"]["exactly_zero"] > 10000


# ------------------------------------------------------------------------------------------ shape-generic row-resident forward
@pytest.mark.parametrize("D,A,hidden,Bsz,M", [
    (45, 12, (512, 256, 128), 16384, 16384),      # the reference's Agent at its minibatch size: one workgroup walks both networks
    (45, 12, (512, 256, 128), 8192, 4133),        # ragged, 65 row tiles: one workgroup per (tile, network), 37-row last tile
    (45, 12, (512, 256, 128), 16421, 16421),      # ragged AND both networks per workgroup
    (48, 12, (128, 128), 4160, 4160),             # one computed layer, 128 wide: waves 4-7 idle
    (48, 12, (256, 128, 128), 8192, 8192),        # 256 then 128 below a 128-wide last layer
    (30, 5, (512, 128, 256, 128), 4096, 4096),    # chunked first layer consumed by a 128-wide layer, three computed layers
    (64, 12, (128, 256, 256), 4099, 4099),        # 64-wide observations (two full slabs), 128 -> 256
    (16, 3, (512, 256, 256), 4096, 4096),         # 16-wide observations (one 2-block slab)
])
def test_shape_generic_row_resident_forward_equals_the_layerwise_launches(tmp_path, D, A, hidden, Bsz, M):
    """rows_fwd_wide_kernel<64> (fwd_rows_wide.h, round 5: a 512-wide first layer produced and consumed in 256-column
    chunks, 128-wide layers on four waves; reference network cleanrl/ppo.py:78-96) against the layer-wise GEMM launches it
    replaces: the contraction order per element is the same, so the activations - and with them the whole minibatch
    gradient and the diagnostics - must be BIT-identical.  Two processes (CATPPO_ROWS_WIDE is read once per process)."""
    import test_gpu_kernels as TK
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"wide{flag}.npz")
        code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=0, out=out)
        env = dict(os.environ, CATPPO_ROWS_WIDE=flag, CATPPO_ROWS_FWD_MIN_ROWS="1", CATPPO_STEP16="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.abs(outs[1]["grad"]).max() > 0 and np.isfinite(outs[0]["grad"]).all()
    np.testing.assert_array_equal(outs[0]["grad"], outs[1]["grad"])
    np.testing.assert_array_equal(outs[0]["diag"], outs[1]["diag"])


def test_shape_generic_row_resident_rollout_forward_equals_the_layerwise_path(tmp_path):
    """rows_fwd_wide_kernel<32> with heads against the layer-wise rollout forward (cleanrl/ppo.py:104-119) on the reference
    network and on mixed widths: hidden layers bit-identical by construction, heads sum in another order (2e-6), Philox
    noise exact.  Window pinned open so that small and ragged batches take the kernel too."""
    import test_gpu_kernels as TK
    cases = {"ref": (45, 12, (512, 256, 128), 4096), "ref_ragged": (45, 12, (512, 256, 128), 2049),
             "ref_tiny": (45, 12, (512, 256, 128), 33), "n128": (48, 7, (128, 128), 300), "one_row": (48, 12, (512, 256), 1),
             "mixed": (30, 5, (256, 128, 256), 1000), "obs64": (64, 12, (512, 128), 257)}
    outs = []
    for env_over in (dict(CATPPO_ROWS_WIDE="1", CATPPO_FUSED_FWD_MIN_ROWS="1", CATPPO_STEP16_FWD="0"),
                     dict(CATPPO_ROWS_WIDE="0", CATPPO_FUSED_FWD="0", CATPPO_ROWS_FWD_ROLLOUT="0")):
        out = str(tmp_path / f"wr{len(outs)}.npz")
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
    assert np.abs(f["ref_act"]).max() > 0 and np.isfinite(f["ref_ragged_lp"]).all()


# ------------------------------------------------------------------------------------------ which kernels does a shape get
def test_plan_log_names_the_kernels_a_shape_gets():
    """catppo_plan_log (ABI 0.5): the dispatch code writes one line per launch decision at the decision
    site; tools/explain_plan.py prints it.  The benchmark shape takes rows_fwd_kernel + fwd_head_kernel, the reference's own
    network (cleanrl/ppo.py:78-96) rows_fwd_wide_kernel in both phases, a 2048-row shard the layer-wise path."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import explain_plan as E
    cfg2 = E.explain(48, 12, (256, 256, 256), 4096, 16384)
    assert "rows_fwd_kernel<32> + heads" in cfg2 and "rows_fwd_kernel<64>" in cfg2 and "fwd_head_kernel<256, prec 0>" in cfg2
    assert cfg2.count("gemm_pair_kernel, ONE launch") == 2 and "dw_fold_kernel" in cfg2 and "clip_adam_dev_kernel" in cfg2
    ref = E.explain(45, 12, (512, 256, 128), 4096, 16384)
    assert "rows_fwd_wide_kernel<32> + heads" in ref and "rows_fwd_wide_kernel<64>" in ref and "in 2 chunk(s)" in ref
    assert "fwd_head_kernel<128, prec 0>" in ref
    shard = E.explain(45, 12, (512, 256, 128), 2048, 2048)       # round 6: an 8-way shard's batch takes the 16-row kernels
    assert "step16_fwd_kernel<48, 512, 256, 128> + heads" in shard and "step16_kernel<48, 512, 256, 128>" in shard
    assert "dw_multi_kernel" in shard and "gemm_pair_kernel" not in shard and "head_loss_kernel" not in shard
    odd = E.explain(45, 12, (512, 128, 128), 2048, 2048)         # a shape without a compiled 16-row kernel: the layer-wise path
    assert "layer-wise" in odd and "head_loss_kernel" in odd and "64x64 weight-gradient tiles" in odd
    from cat_envs import native
    nat = native.get(torch.device("cuda", 0))
    assert nat.plan_log(-1) == nat.plan_log(-1) and "clip + Adam" in nat.plan_log(-1)      # reading does not clear
    assert nat.plan_log(1) == ""                                                             # starting does
    nat.plan_log(0)


def test_weight_gradient_on_64x64_tiles_when_128x128_tiles_underfill_the_chip_is_bit_identical(tmp_path):
    """launch_dw_dx_pair (round 5): the reference's last hidden layer (256 -> 128: two 128x128 weight-gradient tiles x 2
    networks x 32 splits = 128 long workgroups on 256 CUs) takes 64x64 tiles; same splits and contraction order per
    element, so the flat gradient must be BIT-identical to the 128x128 tiling (CATPPO_DW_FILL=0)."""
    import test_gpu_kernels as TK
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"fill{flag}.npz")
        code = TK._FUSED_VS_SPLIT.format(root=ROOT, D=45, A=12, hidden=(512, 256, 128), Bsz=16384, M=16384, prec=0, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CATPPO_DW_FILL=flag), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.abs(outs[1]["grad"]).max() > 0
    np.testing.assert_array_equal(outs[0]["grad"], outs[1]["grad"])
    np.testing.assert_array_equal(outs[0]["diag"], outs[1]["diag"])


def test_gradient_tail_reduced_under_the_final_fold_launch_world_of_one():
    """catppo_set_grad_overlap(ctx, 2) (ABI 0.5): no extra launch - the ranges of the flat gradient that
    are final after dw_fold_kernel are all-reduced on the side stream under the final fold launch, the first layer's behind
    a join; eager and captured, world of one: parameters, Adam state, diagnostics BIT-identical to one all-reduce.  The
    240-wide case takes the shape whose first layer does not share its launch with the fold (one all-reduce inside the
    call)."""
    import test_gpu_r4 as r4
    code = (r4._OVERLAP_CODE.replace('dict(grad_overlap=True, graph_update=False)', 'dict(grad_overlap="tail", graph_update=False)')
            .replace('dict(grad_overlap=True, graph_update=True)', 'dict(grad_overlap="tail", graph_update=True)')
            .replace('tr.grad_overlap == over["grad_overlap"] == tr.nat.grad_overlap_active',
                     'tr.grad_overlap == bool(over["grad_overlap"]) == tr.nat.grad_overlap_active')
            .replace("OVERLAP-OK", "TAIL-OK"))
    assert '"tail"' in code and "TAIL-OK" in code
    for obs in (48, 235):
        r = r4._run_code(code.replace("obs_dim=48", f"obs_dim={obs}"))
        assert r.returncode == 0 and "TAIL-OK" in r.stdout, (obs, r.stdout[-2000:] + r.stderr[-4000:])


# ------------------------------------------------------------------------------------------ deferred post tail
_DEFER_CODE = r"""
import os, sys, torch, numpy as np
dist = os.environ.get('CATPPO_FORCE_DIST') == '1'
if dist:
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['TEST_PORT'], RANK='0', WORLD_SIZE='1')
import smoke_impl
from cat_envs import parallel
if dist:
    parallel.init_rendezvous(0)
from cat_envs.shim import make
from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
task, env_cfg, agent_cfg = smoke_impl.make_cfgs(int(os.environ['T_ENVS']), 12, 2048, 2, 50, (256, 256, 256), False,
                                                obs_dim=int(os.environ['T_OBS']), seed=5)
torch.manual_seed(3)
env = make(task, cfg=env_cfg)
tr = PPOTrainer(env, agent_cfg)
assert tr.defer_tail == (os.environ['CATPPO_ROLLOUT_DEFER_TAIL'] == '1')
logs = []
for _ in range(3):
    logs.append(tr.run_iteration(log=True))
torch.cuda.synchronize()
assert not tr.nat.lib.catppo_rollout_defer_tail(tr.nat.h, -1, tr.nat._stream())           # nothing pending after a rollout
eu = env.unwrapped
cm = eu.constraint_manager
rms = tr.agent.obs_rms
np.savez(os.environ['T_OUT'], flat=tr.agent.flat.cpu().numpy(), dones=tr.dones.float().cpu().numpy(),
         rewards=tr.rewards.float().cpu().numpy(), obs=tr.obs.cpu().numpy(), act=tr.actions.cpu().numpy(),
         values=tr.values.float().cpu().numpy(), cstr=cm.cat._p_cstr.cpu().numpy(), rm=cm.cat._p_rm.cpu().numpy(),
         probs=cm.cat._p_probs.cpu().numpy(), ring=cm._log_ring.cpu().numpy(), ep_viol=cm._ep_viol.cpu().numpy(),
         ep_prob=cm._ep_prob.cpu().numpy(), mean=rms.running_mean.cpu().numpy(), var=rms.running_var.cpu().numpy(),
         count=rms.count.cpu().numpy(), state=eu.sim.cur.cpu().numpy(), ep_len=eu.episode_length_buf.cpu().numpy())
print('DEFER-OK', tr.sink is not None, float(rms.count))
if dist:
    parallel.shutdown_native_comm()
    torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("envs,obs,dist", [(777, 45, "0"), (4096, 48, "0"), (512, 48, "1"), (96, 235, "0")])
def test_post_tail_deferred_into_the_next_pre_launch_is_bit_identical(tmp_path, envs, obs, dist):
    """catppo_rollout_defer_tail (ABI 0.5): the one-workgroup tail of catppo_rollout_post (running
    maxima of cat/constraint_manager.py:58-61, the normaliser state of cleanrl/ppo.py:48-62, the episode log of
    constraint_manager.py:190-211) run by one more workgroup of the NEXT catppo_rollout_pre launch - against the tail
    inside the post launch: three iterations, every buffer and every piece of state BIT-identical; also with every
    exchange point forced on over a world of one (gathered exchange records)."""
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"defer{flag}.npz")
        r = _run_code(_DEFER_CODE, T_OUT=out, CATPPO_ROLLOUT_DEFER_TAIL=flag, CATPPO_FORCE_DIST=dist, T_ENVS=str(envs),
                      T_OBS=str(obs))
        assert r.returncode == 0 and "DEFER-OK True" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    for k in outs[0].files:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    assert np.abs(outs[0]["ring"]).sum() > 0 and outs[0]["dones"].max() > 0 and float(outs[0]["count"]) > 3 * 12 * envs


def test_deferred_post_tail_contract_through_the_c_abi():
    """the state a deferred tail publishes is current after the flush (catppo_rollout_defer_tail(ctx, -1)); a second catppo_rollout_post with the
    tail still pending runs it first (on its own launch); switching the deferral off flushes"""
    import smoke_impl
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPOTrainer
    res = {}
    for mode in ("inline", "deferred"):
        # (a fresh config per env: the curriculum edits the terms' max_p in place)
        task, env_cfg, agent_cfg = smoke_impl.make_cfgs(200, 4, 400, 1, 5, (256, 256), True, obs_dim=48, seed=3)
        torch.manual_seed(3)
        env = make(task, cfg=env_cfg)
        tr = PPOTrainer(env, agent_cfg)
        tr.defer_tail = False
        tr.run_iteration(log=False)                    # fills the argument block of the fused step
        torch.cuda.synchronize()
        eu, nat = env.unwrapped, tr.nat
        cm, rms = eu.constraint_manager, tr.agent.obs_rms
        if mode == "deferred":
            nat.rollout_defer_tail(True)
        snaps = []
        for k in range(3):                             # the same argument block three times: pre, fold, post
            assert nat.lib.catppo_rollout_pre(nat.h, eu._rstep_ref, nat._stream()) == 0
            assert nat.lib.catppo_rollout_post(nat.h, eu._rstep_ref, nat._stream()) == 0
            if k == 1:                                 # post, post: the pending tail of the first runs in front of the second
                assert nat.lib.catppo_rollout_post(nat.h, eu._rstep_ref, nat._stream()) == 0
            if mode == "deferred" and k == 0:
                torch.cuda.synchronize()
                stale = cm.cat._p_rm.clone()           # not yet published ...
                nat.rollout_flush()
                torch.cuda.synchronize()               # ... now it is
                snaps.append(("flush_changed_rm", not torch.equal(stale, cm.cat._p_rm)))
        if mode == "deferred":
            nat.rollout_defer_tail(False)              # = flush
        torch.cuda.synchronize()
        res[mode] = dict(rm=cm.cat._p_rm.cpu().numpy().copy(), mean=rms.running_mean.cpu().numpy().copy(),
                         var=rms.running_var.cpu().numpy().copy(), count=rms.count.cpu().numpy().copy(),
                         ring=cm._log_ring.cpu().numpy().copy(), ep_viol=cm._ep_viol.cpu().numpy().copy(), snaps=snaps)
    for k in ("rm", "mean", "var", "count", "ring", "ep_viol"):
        np.testing.assert_array_equal(res["inline"][k], res["deferred"][k], err_msg=k)
    assert res["deferred"]["snaps"] == [("flush_changed_rm", True)]
