"""GPU parity: every HIP kernel, called through the C ABI (ctypes), against the oracle and
the golden vectors of the reference.  Bit-exact for the CaT masks and GAE; stated tolerances
for the floating-point reductions / GEMMs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import streams as S
from oracle import cat_oracle as CO
from oracle import ppo_oracle as PO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from cat_envs import native
    n = native.Native()
    return n


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def pack_stream_step(step, terms):
    cols = []
    for name, width, kind in terms:
        v = np.asarray(step[name])
        v = v.astype(np.float32)
        cols.append(v.reshape(v.shape[0], -1))
    return np.concatenate(cols, axis=1)


def term_meta(terms, max_p, min_p):
    from cat_envs import native
    off = np.concatenate([[0], np.cumsum([w for _, w, _ in terms])]).astype(np.int32)
    off_c = (C.c_int32 * len(off))(*off.tolist())
    dp = (C.c_float * len(terms))(*[native.f32(p - min_p) for p in max_p])
    return off, off_c, dp


@pytest.mark.parametrize("tag", ["small", "minp", "solo64", "solo4096"])
def test_cat_step_bit_exact_vs_reference_golden(nat, golden, tag):
    """catppo_cat_step on the same streams the reference ConstraintManager consumed."""
    g = golden(f"cat_{tag}")
    n, steps, sub = int(g["n_envs"]), int(g["steps"]), int(g["sub"])
    terms = list(zip([str(x) for x in g["term_names"]], [int(w) for w in g["term_widths"]],
                     [str(k) for k in g["term_kinds"]]))
    tau, min_p = float(g["tau"]), float(g["min_p"])
    stream = S.cat_stream(int(g["seed"]), n, terms, steps)
    K = sum(w for _, w, _ in terms)
    nt = len(terms)
    rm = torch.zeros(K, device="cuda")
    prob = torch.zeros(n, device="cuda")
    viol = torch.zeros(nt, n, device="cuda")
    eprob = torch.zeros(nt, n, device="cuda")
    probs = torch.zeros(n, K, device="cuda")
    ep_len = np.zeros(n, np.int64)
    rs = np.random.RandomState(int(g["seed"]) + 1000)
    reset_at = set(int(x) for x in g["reset_at"])
    orc = CO.ConstraintManagerOracle([t[0] for t in terms], n, tau=tau, min_p=min_p)
    for t in range(steps):
        ep_len += 1
        max_p = [float(x) for x in g["max_p"][t]]
        off, off_c, dp = term_meta(terms, max_p, min_p)
        cstr = dev(pack_stream_step(stream[t], terms))
        nat.cat_step(cstr, off_c, dp, min_p, tau, t == 0, rm, prob, viol, eprob, probs=probs)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(prob.cpu().numpy()[::sub], g["cstr_prob"][t])
        np.testing.assert_array_equal(rm.cpu().numpy(), g["running_maxes"][t])
        pr = probs.cpu().numpy()
        tm = np.stack([pr[:, off[i]:off[i + 1]].max(1)[::sub] for i in range(nt)])
        np.testing.assert_array_equal(tm, g["term_max"][t])
        # and the full (un-subsampled) matrices against the oracle
        po = orc.compute(stream[t], {nm: mp for (nm, _, _), mp in zip(terms, max_p)})
        np.testing.assert_array_equal(prob.cpu().numpy(), po)
        np.testing.assert_array_equal(pr, np.concatenate([orc.cat.probs[nm] for nm, _, _ in terms], 1))
        if t in reset_at:   # ConstraintManager.reset zeroes the stats of the reset envs
            ids = np.nonzero(rs.rand(n) < 0.3)[0]
            idt = torch.from_numpy(ids).cuda()
            L = torch.from_numpy(ep_len).cuda()[idt].float()
            vals = []
            for i, (nm, _, _) in enumerate(terms):
                vals.append((nm, float((viol[i, idt] / L).mean() * 100), float((eprob[i, idt] / L).mean())))
            exp = dict(zip([str(k) for k in g[f"reset{t}_keys"]], g[f"reset{t}_vals"]))
            for nm, v, p in vals:
                np.testing.assert_allclose(v, exp[f"Episode_Constraint_violation/{nm}"], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(p, exp[f"Episode_Constraint_probability/{nm}"], rtol=1e-5, atol=1e-7)
            viol[:, idt] = 0
            eprob[:, idt] = 0
            orc.reset(ids, ep_len)
            ep_len[ids] = 0
    np.testing.assert_array_equal(viol.cpu().numpy()[:, ::sub], g["episode_sums"])
    np.testing.assert_array_equal(eprob.cpu().numpy()[:, ::sub], g["cstr_mean_values"])


def test_cat_two_phase_equals_fused_and_env_finish(nat, golden):
    """colmax -> (all-reduce point) -> apply == fused step; reward/dones epilogue vs golden."""
    terms = S.CAT_TERMS_SOLO12
    n, K = 1000, sum(w for _, w, _ in terms)     # ragged: not a multiple of the 32-env tile
    stream = S.cat_stream(9, n, terms, 3)
    off, off_c, dp = term_meta(terms, S.CAT_MAXP_SOLO12, 0.0)
    nt = len(terms)
    state = []
    for mode in (0, 1):
        rm = torch.zeros(K, device="cuda")
        prob, dones = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        viol, eprob = torch.zeros(nt, n, device="cuda"), torch.zeros(nt, n, device="cuda")
        colmax = torch.zeros(K, device="cuda")
        rs = np.random.RandomState(3)
        outs = []
        for t in range(3):
            cstr = dev(pack_stream_step(stream[t], terms))
            reward = dev(rs.uniform(-0.2, 1.5, n).astype(np.float32))
            r_in = reward.clone()
            reset = dev(rs.rand(n) < 0.1)
            if mode == 0:
                nat.cat_step(cstr, off_c, dp, 0.0, 0.95, t == 0, rm, prob, viol, eprob, reward=reward,
                             reset_mask=reset, dones=dones)
            else:
                nat.cat_colmax(cstr, colmax)
                nat.cat_apply(cstr, off_c, dp, 0.0, 0.95, t == 0, colmax, rm, prob, viol, eprob, reward=reward,
                              reset_mask=reset, dones=dones)
            torch.cuda.synchronize()
            r_exp, d_exp = CO.env_finish(r_in.cpu().numpy(), prob.cpu().numpy(), reset.cpu().numpy())
            np.testing.assert_array_equal(reward.cpu().numpy(), r_exp)
            np.testing.assert_array_equal(dones.cpu().numpy(), d_exp)
            outs.append((prob.cpu().numpy().copy(), rm.cpu().numpy().copy()))
        state.append((outs, viol.cpu().numpy(), eprob.cpu().numpy()))
    for (a, b) in zip(state[0][0], state[1][0]):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(state[0][1], state[1][1])
    np.testing.assert_array_equal(state[0][2], state[1][2])
    g = golden("envfinish")
    n2 = g["reward_in"].shape[0]
    # one-term manager whose probability equals the golden cstr_prob: c = p (rm=1, dp=1)
    p = g["cstr_prob"]
    cstr = dev(np.where(p > 0, p, -1.0).astype(np.float32)[:, None])
    rm = torch.ones(1, device="cuda")
    prob, dones = torch.zeros(n2, device="cuda"), torch.zeros(n2, device="cuda")
    viol, eprob = torch.zeros(1, n2, device="cuda"), torch.zeros(1, n2, device="cuda")
    reward = dev(g["reward_in"])
    off1, dp1 = (C.c_int32 * 2)(0, 1), (C.c_float * 1)(1.0)
    colmax = torch.ones(1, device="cuda")
    nat.cat_apply(cstr, off1, dp1, 0.0, 1.0, False, colmax, rm, prob, viol, eprob, reward=reward,
                  reset_mask=dev(g["reset"]), dones=dones)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(prob.cpu().numpy(), p)
    np.testing.assert_array_equal(reward.cpu().numpy(), g["reward"])
    np.testing.assert_array_equal(dones.cpu().numpy(), g["dones"])


def _term_descs(native, st_dev, feet, upper):
    def d(kind, width, ids, limit=0.0, aux=0.0, x=None, y=None):
        t = native.TermDesc()
        t.kind, t.width, t.n_ids, t.limit, t.aux = kind, width, len(ids), limit, aux
        for i, v in enumerate(ids):
            t.ids[i] = v
        if x is not None:
            t.x, t.x_ld = x.data_ptr(), x.shape[1]
        if y is not None:
            t.y, t.y_ld = y.data_ptr(), y.shape[1]
        return t
    s = st_dev
    alljoints = list(range(12))
    n = native
    return [
        ("joint_position", d(n.TERM_ABS_LIMIT, 2, [1, 4], 1.3, x=s["joint_pos"])),
        ("joint_position_when_moving_forward", d(n.TERM_ABS_DIFF_LIMIT_GATE_CMDY, 4, [0, 3, 6, 9], 0.2, 0.1,
                                                  x=s["joint_pos"], y=s["default_joint_pos"])),
        ("joint_torque", d(n.TERM_ABS_LIMIT, 12, alljoints, 3.0, x=s["applied_torque"])),
        ("joint_velocity", d(n.TERM_ABS_LIMIT, 12, alljoints, 16.0, x=s["joint_vel"])),
        ("joint_acceleration", d(n.TERM_ABS_LIMIT, 12, alljoints, 800.0, x=s["joint_acc"])),
        ("upsidedown", d(n.TERM_GREATER, 1, [2], 0.0, x=s["projected_gravity_b"])),
        ("contact", d(n.TERM_CONTACT_ANY, 1, upper, 1.0)),
        ("base_orientation", d(n.TERM_NORM2_LIMIT, 1, [], 0.1, x=s["projected_gravity_b"])),
        ("air_time", d(n.TERM_AIR_TIME, 4, feet, 0.25, 0.1, x=s["last_air_time"], y=s["first_contact"])),
        ("n_foot_contact", d(n.TERM_N_FOOT_CONTACT, 1, feet, 2, 0.5)),
        ("joint_range", d(n.TERM_ABS_DIFF_LIMIT, 12, alljoints, 0.4, x=s["joint_pos"], y=s["default_joint_pos"])),
        ("action_rate", d(n.TERM_ACTION_RATE, 12, alljoints, 80.0, native.f32(0.02), x=s["action"], y=s["prev_action"])),
        ("foot_contact_force", d(n.TERM_FORCE_LIMIT, 4, feet, 50.0)),
        ("min_base_height", d(n.TERM_LIMIT_MINUS, 1, [2], 0.2, x=s["root_pos_w"])),
        ("no_move", d(n.TERM_ABS_LIMIT_GATE_CMDNORM_LT, 12, alljoints, 4.0, 0.1, x=s["joint_vel"])),
    ]


def test_constraint_terms_vs_reference_golden(nat, golden):
    from cat_envs import native
    g = golden("terms")
    n = int(g["n_envs"])
    st = S.sim_state(int(g["seed"]), n)
    sd = {k: dev(v, torch.float32) for k, v in st.items() if isinstance(v, np.ndarray)}
    feet, upper = [3, 6, 9, 12], [0, 2, 5, 8, 11]
    named = _term_descs(native, sd, feet, upper)
    K = sum(t.width for _, t in named)
    cstr = torch.full((n, K), 123.0, device="cuda")
    nat.cat_terms([t for _, t in named], n, sd["net_forces_w_history"], 3, 13, sd["command"], cstr)
    torch.cuda.synchronize()
    out = cstr.cpu().numpy()
    c = 0
    for name, t in named:
        got = out[:, c:c + t.width]
        c += t.width
        exp = np.asarray(g[name]).astype(np.float32).reshape(n, -1)
        # round 5: ALL fifteen terms bit-exact.  The two norm-based ones (cat/constraints.py:113-119, 201-211) were within
        # 1 / 4 ulp of the norm in rounds 1-4 (unfused sum of squares); terms_eval.h now spells out the fp32 FMA chain
        # torch.norm itself runs, and the record shows zero last-bit differences
        np.testing.assert_array_equal(got, exp, err_msg=name)
        if name in ("base_orientation", "foot_contact_force"):
            import parity_record
            parity_record.record("terms_" + name + "_vs_reference_golden",
                                 {"elements": int(got.size), "not_bit_equal": int((got != exp).sum()),
                                  "max_ulp_of_the_norm": 0.0, "sign_disagreements": int(((got > 0) != (exp > 0)).sum())},
                                 sizes=dict(n_envs=n, width=int(t.width)), seed=int(g["seed"]))


@pytest.mark.parametrize("T,N,seed", [(1, 1, 1), (24, 64, 2), (48, 4096, 3), (5, 1000, 4), (24, 262144, 5)])
def test_gae_bit_exact(nat, T, N, seed):
    x = S.gae_inputs(seed, T, N)
    d = {k: dev(v) for k, v in x.items()}
    adv, ret = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda")
    nat.gae(d["rewards"], d["values"], d["dones"], d["true_dones"], d["next_value"], d["next_done"],
            d["next_true_done"], 0.99, 0.95, adv, ret)
    torch.cuda.synchronize()
    a, r = PO.gae_numpy_exact(x["rewards"], x["values"], x["dones"], x["true_dones"], x["next_value"], x["next_done"],
                              x["next_true_done"], 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), a)
    np.testing.assert_array_equal(ret.cpu().numpy(), r)


@pytest.mark.parametrize("tag", ["64x24", "64x48", "4096x24"])
def test_gae_vs_reference_ppo_run(nat, golden, tag):
    """advantages/returns captured inside the running reference PPO() (float dones, time-outs)."""
    g = golden(f"ppo_{tag}")
    N, T, D, sub, seed = int(g["N"]), int(g["T"]), int(g["D"]), int(g["sub"]), int(g["seed"])
    s = S.env_stream(seed, T * int(g["iters"]), N, D)
    # iteration 0: rewards/dones/timeouts of steps 0..T-1; dones[t] is the done of the PREVIOUS env step
    dones = np.concatenate([np.zeros((1, N), np.float32), s["dones"][:T - 1]])
    tdones = np.concatenate([np.zeros((1, N), np.float32), s["timeouts"][:T - 1].astype(np.float32)])
    # GAE is independent per env: the 4096-env run stores every `sub`-th column of values / next_value / outputs,
    # so the kernel runs on exactly those columns of the (regenerated) reward / done streams
    cols = slice(None, None, sub)
    vals = g["it0_values"]
    Ns = vals.shape[1]
    adv, ret = torch.empty(T, Ns, device="cuda"), torch.empty(T, Ns, device="cuda")
    nat.gae(dev(s["reward"][:T][:, cols]), dev(vals), dev(dones[:, cols]), dev(tdones[:, cols]),
            dev(g["it0_next_value"]), dev(s["dones"][T - 1][cols]),
            dev(s["timeouts"][T - 1].astype(np.float32)[cols]), 0.99, 0.95, adv, ret)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(adv.cpu().numpy(), g["it0_advantages"])
    np.testing.assert_array_equal(ret.cpu().numpy(), g["it0_returns"])


def test_running_mean_std_vs_reference_golden(nat, golden):
    g = golden("rms")
    rs = np.random.RandomState(int(g["seed"]))
    xs = (rs.standard_normal((30, 64, 45)) * rs.uniform(0.1, 5, 45) + rs.uniform(-2, 2, 45)).astype(np.float32)
    ys = (rs.standard_normal((30, 1536)) * 3 + 1).astype(np.float32)
    mean, var, cnt = torch.zeros(45, device="cuda"), torch.ones(45, device="cuda"), torch.ones(1, device="cuda")
    smean, svar, scnt = torch.zeros(1, device="cuda"), torch.ones(1, device="cuda"), torch.ones(1, device="cuda")
    out = torch.zeros(64, 48, device="cuda")     # padded leading dimension, pad columns untouched
    sout = torch.zeros(1536, device="cuda")
    for i in range(30):
        x, y = dev(xs[i]), dev(ys[i])
        nat.rms_update(x, 64, 45, 45, mean, var, cnt)
        nat.rms_normalize(x, 64, 45, 45, mean, var, 1e-8, out, 48)
        nat.rms_update(y, 1536, 1, 1, smean, svar, scnt)
        nat.rms_normalize(y, 1536, 1, 1, smean, svar, 1e-8, sout, 1)
        torch.cuda.synchronize()
        st = np.concatenate([mean.cpu().numpy(), var.cpu().numpy(), cnt.cpu().numpy()])
        np.testing.assert_allclose(st, g["vec_state"][i], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose([float(smean), float(svar), float(scnt)], g["sca_state"][i], rtol=2e-6, atol=1e-6)
        if i == 0:
            np.testing.assert_allclose(out.cpu().numpy()[:, :45], g["vec_out_first"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy()[:, :45], g["vec_out_last"], rtol=1e-5, atol=1e-5)
    assert float(out[:, 45:].abs().max()) == 0.0
    np.testing.assert_allclose(sout.cpu().numpy()[:64], g["sca_out_last"], rtol=1e-5, atol=1e-5)
    # two-phase form (all-reduce point between moments and merge) gives the same state
    m2, v2, c2 = torch.zeros(45, device="cuda"), torch.ones(45, device="cuda"), torch.ones(1, device="cuda")
    sums = torch.zeros(90, device="cuda", dtype=torch.float64)
    for i in range(30):
        nat.rms_moments(dev(xs[i]), 64, 45, 45, sums)
        nat.rms_merge(sums, 64, 45, m2, v2, c2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(m2.cpu().numpy(), mean.cpu().numpy())
    np.testing.assert_array_equal(v2.cpu().numpy(), var.cpu().numpy())


# ------------------------------------------------------------------------------------- MLP
def flat_params(native, shape, lay, sd, device="cuda"):
    """reference state_dict (numpy) -> padded flat fp32 buffer in the library's layout"""
    flat = np.zeros(lay.n_flat, np.float32)
    A = shape.act_dim
    flat[lay.off_logstd:lay.off_logstd + A] = sd["actor_logstd"].reshape(-1)
    for net, pre in ((0, "critic"), (1, "actor_mean")):
        for l in range(shape.n_hidden + 1):
            w, b = sd[f"{pre}.{2 * l}.weight"], sd[f"{pre}.{2 * l}.bias"]
            out, inp = w.shape
            ld = lay.in_dim[l]
            view = flat[lay.off_w[net][l]:lay.off_w[net][l] + out * ld].reshape(out, ld)
            view[:, :inp] = w
            flat[lay.off_b[net][l]:lay.off_b[net][l] + out] = b
    return torch.from_numpy(flat).to(device)


def unflatten_grad(shape, lay, flat, sd_like):
    out = {"actor_logstd": flat[lay.off_logstd:lay.off_logstd + shape.act_dim].reshape(1, -1)}
    for net, pre in ((0, "critic"), (1, "actor_mean")):
        for l in range(shape.n_hidden + 1):
            o, i = sd_like[f"{pre}.{2 * l}.weight"].shape
            ld = lay.in_dim[l]
            out[f"{pre}.{2 * l}.weight"] = flat[lay.off_w[net][l]:lay.off_w[net][l] + o * ld].reshape(o, ld)[:, :i]
            out[f"{pre}.{2 * l}.bias"] = flat[lay.off_b[net][l]:lay.off_b[net][l] + o]
    return out


@pytest.mark.parametrize("D,A,hidden,B", [(45, 12, (512, 256, 128), 96), (48, 12, (256, 256, 256), 4096),
                                           (45, 12, (512, 256, 128), 1000), (45, 12, (128, 512), 300),
                                           (48, 7, (64,), 100), (33, 15, (64, 128, 64, 128), 70),
                                           (16, 1, (128, 64), 1)])
@pytest.mark.parametrize("prec", [0, 2], ids=["fp32mfma", "bf16x3"])
def test_policy_act_vs_oracle_and_golden(nat, golden, D, A, hidden, B, prec):
    """prec 0: fp32-input MFMA, 1e-5 (north_star).  prec 2: split-bf16 operands (three bf16 MFMAs per product, ~16
    mantissa bits per operand: the residual x - hi - lo is <= 2^-17 |x|): measured worst case 2.1e-5 on O(1) outputs
    after three layers, held to 5e-5 against the fp32 oracle and the reference Agent's golden outputs."""
    from cat_envs import native
    shape = native.shape_of(D, A, hidden, mfma_bf16=prec)
    lay = native.layout_of(shape)
    w = S.agent_weights(3, D, A, hidden)
    ag = PO.AgentOracle(D, A, hidden)
    ag.load(w)
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
    tol = dict(rtol=1e-5, atol=2e-5) if prec == 0 else dict(rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(act.cpu().numpy(), a.numpy(), **tol)
    np.testing.assert_allclose(val.cpu().numpy(), v.numpy()[:, 0], **tol)
    np.testing.assert_allclose(logp.cpu().numpy(), lp.numpy(), rtol=1e-5, atol=1e-4)
    val2 = torch.empty(B, device="cuda")
    nat.value(shape, params, dev(xp), B, val2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(val2.cpu().numpy(), val.cpu().numpy())
    if hidden == (512, 256, 128) and B == 96:
        g = golden("agent")      # the reference Agent itself on these weights / inputs
        np.testing.assert_allclose(act.cpu().numpy(), g["action"], **tol)
        np.testing.assert_allclose(logp.cpu().numpy(), g["logprob"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(val.cpu().numpy(), g["value"][:, 0], **tol)
        # deterministic path: eps = NULL -> action = mean
        nat.policy_act(shape, params, dev(xp), B, None, act, logp, val)
        torch.cuda.synchronize()
        with torch.no_grad():
            am, _, _, _ = ag.get_action_and_value(torch.from_numpy(x), deterministic=True)
        np.testing.assert_allclose(act.cpu().numpy(), am.numpy(), **tol)


def _minibatch_case(D, A, hidden, Bsz, M, seed):
    rs = np.random.RandomState(seed)
    return dict(
        obs=rs.standard_normal((Bsz, D)).astype(np.float32),
        act=rs.standard_normal((Bsz, A)).astype(np.float32) * 0.7,
        logp=(rs.standard_normal(Bsz) * 0.5 - 12.0).astype(np.float32),
        adv=rs.standard_normal(Bsz).astype(np.float32) * 2 + 0.3,
        ret=rs.standard_normal(Bsz).astype(np.float32),
        val=rs.standard_normal(Bsz).astype(np.float32),
        inds=rs.permutation(Bsz)[:M].astype(np.int64),
        vmean=np.float32(0.37), vvar=np.float32(2.3))


@pytest.mark.parametrize("D,A,hidden,Bsz,M,flags", [
    (45, 12, (512, 256, 128), 1536, 512, (True, True)),
    (45, 12, (512, 256, 128), 1536, 500, (False, False)),     # ragged minibatch, no adv-norm / no v-clip
    (48, 12, (256, 256, 256), 8192, 4096, (True, True)),
    (45, 12, (512, 256, 128), 98304, 16384, (True, True)),    # the reference's full minibatch
    (45, 12, (128, 512), 2048, 1024, (True, True)),           # widest supported last layer (8 columns per lane)
    (48, 7, (64,), 1024, 512, (True, True)),                  # narrowest: one hidden layer of 64, 7 actions
    (33, 15, (64, 128, 64, 128), 1024, 300, (True, True)),    # deepest (4 hidden layers), widest action (15)
    (16, 1, (128, 64), 512, 65, (True, False)),               # one action dimension, minibatch of 65
    # >= 8192 rows in fp32: last hidden layer + heads + loss run as ONE launch (fwd_head_kernel)
    (48, 12, (256, 256, 256), 24576, 8229, (False, False)),   # 256-wide tile, ragged last 64-row tile, no adv-norm / v-clip
    (45, 3, (256, 128), 16384, 8192, (True, True)),           # 128-wide tile, 3 actions
    (48, 15, (128, 256), 16384, 8200, (True, False)),         # 15 actions (every head slot but one), two hidden layers
])
@pytest.mark.parametrize("prec", [0, 2], ids=["fp32mfma", "bf16x3"])
def test_ppo_minibatch_grad_vs_autograd_oracle(nat, D, A, hidden, Bsz, M, flags, prec):
    from cat_envs import native
    norm_adv, clip_vloss = flags
    shape = native.shape_of(D, A, hidden, mfma_bf16=prec)
    lay = native.layout_of(shape)
    w = S.agent_weights(5, D, A, hidden)
    c = _minibatch_case(D, A, hidden, Bsz, M, 6)
    # make log-probs realistic so that ratios straddle the clip range: old logp = new logp + noise
    ag = PO.AgentOracle(D, A, hidden)
    ag.load(w)
    ag.value_rms.mean, ag.value_rms.var = torch.tensor(float(c["vmean"])), torch.tensor(float(c["vvar"]))
    with torch.no_grad():
        _, lp0, _, _ = ag.get_action_and_value(torch.from_numpy(c["obs"]), torch.from_numpy(c["act"]))
    rs = np.random.RandomState(7)
    c["logp"] = (lp0.numpy() + rs.standard_normal(Bsz).astype(np.float32) * 0.25).astype(np.float32)
    cfg = dict(clip_coef=0.2, ent_coef=0.001, vf_coef=2.0, norm_adv=norm_adv, clip_vloss=clip_vloss)
    params_t = [p.requires_grad_(True) for p in ag.parameters()]
    mb = torch.from_numpy(c["inds"])
    loss, st = PO.ppo_minibatch_loss(ag, torch.from_numpy(c["obs"])[mb], torch.from_numpy(c["act"])[mb],
                                     torch.from_numpy(c["logp"])[mb], torch.from_numpy(c["adv"])[mb],
                                     torch.from_numpy(c["ret"])[mb], torch.from_numpy(c["val"])[mb], cfg)
    loss.backward()
    ref_grad = {k: v.grad.numpy() for k, v in ag.p.items()}

    params = flat_params(native, shape, lay, w)
    obs_p = np.zeros((Bsz, lay.obs_pad), np.float32)
    obs_p[:, :D] = c["obs"]
    grad = torch.full((lay.n_flat,), 7.0, device="cuda")
    grad_pad_probe = grad.clone()
    diag = torch.zeros(8, device="cuda")
    hp = native.PpoHparams(0.2, 0.001, 2.0, int(norm_adv), int(clip_vloss), 1.0 / M, 0)
    nat.mlp_reserve(shape, M)
    args = (shape, hp, params, dev(obs_p), dev(c["act"]), dev(c["logp"]), dev(c["adv"]), dev(c["ret"]),
            dev(c["val"]), dev(c["inds"]), dev(np.array([c["vmean"]])), dev(np.array([c["vvar"]])), None, grad, diag)
    nat.ppo_minibatch_grad(*args)
    torch.cuda.synchronize()
    d = diag.cpu().numpy()
    exp = [float(st["pg_loss"]), float(st["v_loss"]), float(st["entropy"]), float(st["loss"]),
           float(st["approx_kl"]), float(st["old_approx_kl"]), float(st["clipfrac"])]
    np.testing.assert_allclose(d[:7], exp, rtol=2e-4, atol=2e-6)
    assert d[7] == 1.0
    got = unflatten_grad(shape, lay, grad.cpu().numpy(), w)
    gnorm = math.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in ref_grad.values()))
    for k, v in ref_grad.items():
        # <= 1e-4 relative (GEMM reassociation); scale by the tensor's own magnitude
        scale = max(np.abs(v).max(), 1e-8)
        err = np.abs(got[k].reshape(v.shape) - v).max() / scale
        assert err < 2e-4, (k, err)
    gnorm_got = math.sqrt(sum(float((got[k].astype(np.float64) ** 2).sum()) for k in ref_grad))
    assert abs(gnorm_got - gnorm) < 1e-4 * gnorm
    # determinism (no float atomics): a second run is bit-identical
    grad2, diag2 = torch.zeros_like(grad), torch.zeros(8, device="cuda")
    nat.ppo_minibatch_grad(*args[:-2], grad2, diag2)
    torch.cuda.synchronize()
    written = grad.cpu().numpy() != grad_pad_probe.cpu().numpy()
    np.testing.assert_array_equal(grad.cpu().numpy()[written], grad2.cpu().numpy()[written])


def test_clip_adam_vs_torch(nat):
    n = 377_300
    rs = np.random.RandomState(8)
    p0 = rs.standard_normal(n).astype(np.float32)
    p_ref = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=3e-4, eps=1e-5)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        scale = 10.0 if step % 2 else 0.001      # exercise both clip branches
        g0 = (rs.standard_normal(n) * scale / math.sqrt(n)).astype(np.float32)
        lr = 3e-4 * (1 - (step - 1) / 10)
        opt.param_groups[0]["lr"] = lr
        p_ref.grad = torch.from_numpy(g0.copy())
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        g = dev(g0)
        nat.clip_adam(p, g, m, v, n, 1.0, lr, 0.9, 0.999, 1e-5, step)
        torch.cuda.synchronize()
        np.testing.assert_allclose(g.cpu().numpy(), p_ref.grad.numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    st = opt.state[p_ref]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-12)


def test_errors_are_loud(nat):
    from cat_envs import native
    with pytest.raises(ValueError):
        native.layout_of(native.shape_of(45, 12, (100, 64)))
    shape = native.shape_of(45, 12, (512, 256, 128))
    lay = native.layout_of(shape)
    assert lay.n_params == 377241 and lay.obs_pad == 48
    fresh = native.Native()
    params = torch.zeros(lay.n_flat, device="cuda")
    x = torch.zeros(1 << 20, 48, device="cuda")
    a, l, v = torch.zeros(1 << 20, 12, device="cuda"), torch.zeros(1 << 20, device="cuda"), torch.zeros(1 << 20, device="cuda")
    with pytest.raises(RuntimeError, match="workspace"):
        fresh.policy_act(shape, params, x, 1 << 20, None, a, l, v)     # default workspace too small
    with pytest.raises(RuntimeError):
        fresh.gae(a[:, 0].contiguous().view(1, -1), v.view(1, -1), v.view(1, -1), v.view(1, -1), v, v, v, 0.99, 0.95,
                  torch.zeros(1, 1 << 20), torch.zeros(1, 1 << 20, device="cuda"))  # CPU tensor rejected


# ---- the other two trainers' float-done recurrences -----------------------------
@pytest.mark.parametrize("T,N,seed", [(1, 1, 1), (24, 64, 2), (16, 4096, 3), (8, 131072, 4)])
def test_gae_rl_games_variant_bit_exact(nat, T, N, seed):
    """discount_values with float fdones == the CleanRL recurrence with no time-out channel"""
    x = S.gae_inputs(seed, T, N)
    d = {k: dev(v) for k, v in x.items()}
    adv, ret = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda")
    nat.gae_rl_games(d["next_done"], d["next_value"], d["dones"], d["values"], d["rewards"], 0.99, 0.95, adv, ret)
    torch.cuda.synchronize()
    t = {k: torch.from_numpy(v) for k, v in x.items()}
    a, r = PO.gae_rl_games(t["next_done"], t["next_value"], t["dones"], t["values"], t["rewards"], 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), a.numpy())
    np.testing.assert_array_equal(ret.cpu().numpy(), r.numpy())
    z = np.zeros_like(x["dones"])
    a0, _ = PO.gae_numpy_exact(x["rewards"], x["values"], x["dones"], z, x["next_value"], x["next_done"], z[0], 0.99,
                               0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), a0)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gae_skrl_variant_vs_reference_golden(nat, golden, tag):
    """returns bit-exact, whole-batch normalised advantages within 2e-6 of the reference's compute_gae"""
    from cat_envs.tasks.utils.skrl import compute_gae
    g = golden("skrl_gae")
    rew, val, done, last = (g[f"{tag}_{k}"] for k in ("rewards", "values", "dones", "last_values"))
    ret, adv = compute_gae(dev(rew), dev(done), dev(val), dev(last), discount_factor=0.99, lambda_coefficient=0.95)
    torch.cuda.synchronize()
    assert ret.shape == rew.shape and adv.shape == rew.shape
    np.testing.assert_array_equal(ret.cpu().numpy(), g[f"{tag}_returns"])
    if rew.size > 1:
        np.testing.assert_allclose(adv.cpu().numpy(), g[f"{tag}_advantages"], rtol=2e-6, atol=2e-6)
    else:
        assert np.isnan(adv.cpu().numpy()).all() and np.isnan(g[f"{tag}_advantages"]).all()   # std of one sample


def test_gae_skrl_wide_path_and_normalize_stats(nat):
    T, N = 8, 131072
    x = S.gae_inputs(9, T, N)
    d = {k: dev(v) for k, v in x.items()}
    adv, ret = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda")
    nat.gae_skrl(d["rewards"], d["dones"], d["values"], d["next_value"], 0.99, 0.95, adv, ret)
    t = {k: torch.from_numpy(v) for k, v in x.items()}
    r0, n0, a0 = PO.gae_skrl(t["rewards"], t["dones"], t["values"], t["next_value"], 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), a0.numpy())
    np.testing.assert_array_equal(ret.cpu().numpy(), r0.numpy())
    stats = torch.zeros(2, device="cuda")
    out = torch.empty_like(adv)
    nat.adv_normalize(adv, out, stats)
    a64 = a0.numpy().astype(np.float64)
    np.testing.assert_allclose(stats.cpu().numpy(), [a64.mean(), a64.std(ddof=1) + 1e-8], rtol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), n0.numpy(), rtol=1e-5, atol=1e-5)


def test_value_bootstrap_bit_exact(nat):
    from cat_envs.tasks.utils.rl_games import bootstrap_time_outs, discount_values
    rs = np.random.RandomState(3)
    n = 4099
    rew = rs.uniform(0, 1.5, (n, 1)).astype(np.float32)
    val = rs.standard_normal((n, 1)).astype(np.float32)
    to = rs.uniform(size=n) < 0.2
    r = dev(rew)
    bootstrap_time_outs(r, dev(val), dev(to), 0.99)
    want = PO.value_bootstrap(torch.from_numpy(rew), torch.from_numpy(val), torch.from_numpy(to).unsqueeze(1), 0.99)
    np.testing.assert_array_equal(r.cpu().numpy(), want.numpy())
    # (T, N, 1) buffer layout of the rl_games experience buffer through discount_values
    x = S.gae_inputs(5, 12, 100)
    adv = discount_values(dev(x["next_done"]), dev(x["next_value"][:, None]), dev(x["dones"]),
                          dev(x["values"][:, :, None]), dev(x["rewards"][:, :, None]), 0.99, 0.95)
    t = {k: torch.from_numpy(v) for k, v in x.items()}
    a, _ = PO.gae_rl_games(t["next_done"], t["next_value"], t["dones"], t["values"], t["rewards"], 0.99, 0.95)
    assert adv.shape == (12, 100, 1)
    np.testing.assert_array_equal(adv.cpu().numpy()[:, :, 0], a.numpy())


# ---- BASELINE full sizes ------------------------------------------------------------------------------
def test_cat_step_32768_envs_mixed_hard_soft_max_p(nat):
    """config 5 size: 32768 envs x 13 terms (78 columns), hard (max_p = 1) and soft (0.1 / 0.25) terms mixed,
    every matrix bit-exact against the oracle over 4 steps with a max_p curriculum change at step 2."""
    terms, n = S.CAT_TERMS_SOLO12, 32768
    K, nt = sum(w for _, w, _ in terms), len(S.CAT_TERMS_SOLO12)
    stream = S.cat_stream(77, n, terms, 4)
    rm, prob = torch.zeros(K, device="cuda"), torch.zeros(n, device="cuda")
    viol, eprob = torch.zeros(nt, n, device="cuda"), torch.zeros(nt, n, device="cuda")
    probs = torch.zeros(n, K, device="cuda")
    orc = CO.ConstraintManagerOracle([t[0] for t in terms], n, tau=0.95, min_p=0.0)
    for t in range(4):
        max_p = [p if t < 2 else min(1.0, 2 * p) for p in S.CAT_MAXP_SOLO12]
        _, off_c, dp = term_meta(terms, max_p, 0.0)
        nat.cat_step(dev(pack_stream_step(stream[t], terms)), off_c, dp, 0.0, 0.95, t == 0, rm, prob, viol, eprob,
                     probs=probs)
        po = orc.compute(stream[t], {nm: mp for (nm, _, _), mp in zip(terms, max_p)})
        np.testing.assert_array_equal(prob.cpu().numpy(), po)
        np.testing.assert_array_equal(probs.cpu().numpy(), np.concatenate([orc.cat.probs[nm] for nm, _, _ in terms], 1))
    assert (prob.cpu().numpy() == 1.0).any() and ((prob.cpu().numpy() > 0) & (prob.cpu().numpy() < 1)).any()


def test_gae_full_size_slices_and_linearity(nat):
    """N = 2^22 envs x T = 48 (the 4.8 GB roofline-sweep size): slices of envs (the recurrence is independent per
    env) bit-exact against the oracle, and the whole result checked through a size-independent property:
    with dones == 0 the advantages are linear in the rewards, A(r1 + r2, v) + A(0, v) == A(r1, v) + A(r2, v)
    up to fp32 rounding."""
    T, N = 48, 1 << 22
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda *s: torch.rand(*s, device="cuda", generator=g)  # noqa: E731
    rew, val = mk(T, N) * 1.5, mk(T, N) * 2 - 1
    done = torch.where(mk(T, N) < 0.3, mk(T, N), torch.zeros((), device="cuda"))
    done = torch.where(mk(T, N) < 0.02, torch.ones((), device="cuda"), done)
    td = (mk(T, N) < 0.01).float()
    nv, nd, ntd = mk(N) * 2 - 1, done[0].clone(), td[0].clone()
    adv, ret = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda")
    nat.gae(rew, val, done, td, nv, nd, ntd, 0.99, 0.95, adv, ret)
    for sl in (slice(0, 2048), slice(N // 2 - 1000, N // 2 + 1000), slice(N - 2048, N)):
        c = lambda x: x[..., sl].cpu().numpy()  # noqa: E731
        a, r = PO.gae_numpy_exact(c(rew), c(val), c(done), c(td), c(nv), c(nd), c(ntd), 0.99, 0.95)
        np.testing.assert_array_equal(c(adv), a)
        np.testing.assert_array_equal(c(ret), r)
    assert bool((ret == adv + val).all())
    z, zr = torch.zeros(T, N, device="cuda"), torch.zeros(N, device="cuda")
    r2 = mk(T, N)
    out = []
    for r_ in (rew + r2, z, rew, r2):
        a_ = torch.empty(T, N, device="cuda")
        nat.gae(r_, val, z, z, nv, zr, zr, 0.99, 0.95, a_, ret)
        out.append(a_)
    err = (out[0] + out[1] - out[2] - out[3]).abs().max().item()
    assert err < 2e-4, err


@pytest.mark.parametrize("T,N,seed", [(24, 64, 2), (5, 1000, 4), (24, 262144, 5)])
def test_gae_fp16_planes_bit_exact(nat, T, N, seed):
    """fp16 rollout planes (BASELINE config 5): fp32 recurrence on the widened values, RNE to half on store"""
    x = {k: v.astype(np.float16) for k, v in S.gae_inputs(seed, T, N).items()}
    d = {k: dev(v) for k, v in x.items()}
    adv = torch.empty(T, N, device="cuda", dtype=torch.float16)
    ret = torch.empty_like(adv)
    nat.gae_f16(d["rewards"], d["values"], d["dones"], d["true_dones"], d["next_value"], d["next_done"],
                d["next_true_done"], 0.99, 0.95, adv, ret)
    f = {k: v.astype(np.float32) for k, v in x.items()}
    a, r = PO.gae_numpy_exact(f["rewards"], f["values"], f["dones"], f["true_dones"], f["next_value"], f["next_done"],
                              f["next_true_done"], 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), a.astype(np.float16))
    np.testing.assert_array_equal(ret.cpu().numpy(), r.astype(np.float16))


def test_epoch_gather_plus_packed_grad_equals_gathering_grad(nat):
    """catppo_ppo_gather (one launch per epoch) + catppo_ppo_minibatch_grad_packed on slice k == catppo_ppo_minibatch_grad
    on inds[k*M:(k+1)*M], bit for bit, including the short last minibatch (B = 2.5 minibatches)"""
    from cat_envs import native
    D, A, hidden, M = 45, 12, (256, 128), 1000
    B = 2500
    shape = native.shape_of(D, A, hidden)
    lay = native.layout_of(shape)
    c = _minibatch_case(D, A, hidden, B, B, 11)
    params = torch.randn(lay.n_flat, device="cuda") * 0.05
    obs_p = np.zeros((B, lay.obs_pad), np.float32)
    obs_p[:, :D] = c["obs"]
    bufs = [dev(obs_p), dev(c["act"]), dev(c["logp"]), dev(c["adv"]), dev(c["ret"]), dev(c["val"])]
    inds = dev(c["inds"])
    vm, vv = dev(np.array([c["vmean"]])), dev(np.array([c["vvar"]]))
    parts = (M + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
    n_mb = (B + M - 1) // M
    x_g, act_g = torch.empty(B, lay.obs_pad, device="cuda"), torch.empty(B, A, device="cuda")
    scal_g = torch.empty(4 * B, device="cuda")
    advp = torch.empty(n_mb * parts * 2, dtype=torch.float64, device="cuda")
    nat.mlp_reserve(shape, M)
    nat.ppo_gather(shape, *bufs, inds, M, x_g, act_g, scal_g, advp)
    np.testing.assert_array_equal(x_g.cpu().numpy(), obs_p[c["inds"]])
    for k in range(n_mb):
        start, m = k * M, min(M, B - k * M)
        hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / m, 0)
        g1, d1 = torch.zeros(lay.n_flat, device="cuda"), torch.zeros(8, device="cuda")
        g2, d2 = torch.zeros(lay.n_flat, device="cuda"), torch.zeros(8, device="cuda")
        nat.ppo_minibatch_grad(shape, hp, params, *bufs, inds[start:start + m], vm, vv, None, g1, d1)
        nat.ppo_minibatch_grad_packed(shape, hp, params, x_g[start:], act_g[start:], scal_g[4 * start:],
                                      advp[2 * k * parts:], m, vm, vv, None, g2, d2)
        np.testing.assert_array_equal(g1.cpu().numpy(), g2.cpu().numpy())
        np.testing.assert_array_equal(d1.cpu().numpy(), d2.cpu().numpy())
        assert np.abs(g1.cpu().numpy()).max() > 0


_FUSED_VS_SPLIT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "constraints-as-terminations_amd"))
import streams as S
import test_gpu_kernels as T
from cat_envs import native
D, A, hidden, Bsz, M = {D}, {A}, {hidden}, {Bsz}, {M}
nat = native.Native()
shape = native.shape_of(D, A, hidden, mfma_bf16={prec})
lay = native.layout_of(shape)
w = S.agent_weights(5, D, A, hidden)
c = T._minibatch_case(D, A, hidden, Bsz, M, 6)
c["logp"] = (c["logp"] * 0.0 - 11.0 + np.random.RandomState(3).standard_normal(Bsz) * 0.3).astype(np.float32)
params = T.flat_params(native, shape, lay, w)
obs_p = np.zeros((Bsz, lay.obs_pad), np.float32); obs_p[:, :D] = c["obs"]
grad = torch.zeros(lay.n_flat, device="cuda"); diag = torch.zeros(8, device="cuda")
hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / M, 0)
nat.mlp_reserve(shape, M)
nat.ppo_minibatch_grad(shape, hp, params, T.dev(obs_p), T.dev(c["act"]), T.dev(c["logp"]), T.dev(c["adv"]),
                       T.dev(c["ret"]), T.dev(c["val"]), T.dev(c["inds"]), T.dev(np.array([c["vmean"]])),
                       T.dev(np.array([c["vvar"]])), None, grad, diag)
torch.cuda.synchronize()
np.savez({out!r}, grad=grad.cpu().numpy(), diag=diag.cpu().numpy())
"""


@pytest.mark.parametrize("D,A,hidden,Bsz,M,prec", [
    (48, 12, (256, 256, 256), 16384, 16384, 0),      # cfg2's minibatch
    (45, 12, (512, 256, 128), 8192, 4133, 0),        # reference shapes, ragged, just above the fused threshold
    (48, 12, (256, 256, 256), 8192, 8192, 1),        # bf16 operands (cfg5)
])
def test_fused_head_launch_equals_the_two_launch_path(tmp_path, D, A, hidden, Bsz, M, prec):
    """fwd_head_kernel (last hidden layer + heads + loss + head backward in one launch) against the forward GEMM +
    head_loss_kernel pair it replaces: same minibatch, two processes (CATPPO_FUSED_HEAD is read once per process).
    Different summation orders in the heads only: agreement far inside the oracle tolerance of the tests above."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / f"fused{flag}.npz")
        code = _FUSED_VS_SPLIT.format(root=root, D=D, A=A, hidden=hidden, Bsz=Bsz, M=M, prec=prec, out=out)
        # (bf16 operands: fp32-STORED activations on both sides - what the bf16-stored mode changes in the backward has its own
        # test, tests/test_gpu_bf16.py::test_bf16_stored_activations_equal_the_fp32_stored_path)
        env = dict(os.environ, CATPPO_FUSED_HEAD=flag, CATPPO_ACT16="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    g1, g0 = outs[0]["grad"], outs[1]["grad"]
    scale = np.abs(g0).max()
    assert np.abs(g1 - g0).max() <= 2e-5 * scale, (np.abs(g1 - g0).max(), scale)
    np.testing.assert_allclose(outs[0]["diag"][:7], outs[1]["diag"][:7], rtol=2e-5, atol=1e-7)


_FUSED_FWD_AB = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "constraints-as-terminations_amd"))
import streams as S
import test_gpu_kernels as T
from cat_envs import native
nat = native.Native()
res = {{}}
for tag, (D, A, hidden, B) in {cases!r}.items():
    shape = native.shape_of(D, A, hidden)
    lay = native.layout_of(shape)
    params = T.flat_params(native, shape, lay, S.agent_weights(9, D, A, hidden))
    rs = np.random.RandomState(B)
    x = np.zeros((B, lay.obs_pad), np.float32); x[:, :D] = rs.standard_normal((B, D)) * 1.5
    eps = rs.standard_normal((B, A)).astype(np.float32)
    act, lp, val = (torch.empty(B, A, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda"))
    nat.mlp_reserve(shape, B)
    nat.policy_act(shape, params, T.dev(x), B, T.dev(eps), act, lp, val)
    v2 = torch.empty(B, device="cuda")
    nat.value(shape, params, T.dev(x), B, v2)
    st = nat.iter_state_new(1234, 3e-4)
    a3, l3, v3, e3 = torch.empty(B, A, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty(B, A, device="cuda")
    nat.policy_act_rng(shape, params, T.dev(x), B, st, 5, a3, l3, v3, eps_out=e3)
    torch.cuda.synchronize()
    for k, t in (("act", act), ("lp", lp), ("val", val), ("v2", v2), ("a3", a3), ("l3", l3), ("e3", e3)):
        res[tag + "_" + k] = t.cpu().numpy()
np.savez({out!r}, **res)
"""


def test_fused_forward_launch_equals_the_layerwise_path(tmp_path):
    """fused_fwd_kernel (all hidden layers + head of one network per 32-row workgroup) against the layer-wise launches it
    replaces between 2049 and 4096 rows: two processes (the switches are read once), the window pinned open so that
    small, ragged and multi-chunk shapes run fused too.  Hidden activations are bit-identical by construction (same
    contraction order); the head sums in another order: values / means / log-probs agree to 2e-6, Philox noise exactly."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = {"cfg2": (48, 12, (256, 256, 256), 4096), "ref_ragged": (45, 12, (512, 256, 128), 2049),
             "tiny": (45, 12, (512, 256, 128), 33), "wide_head": (33, 7, (128, 512), 300), "one_row": (48, 12, (256, 128), 1)}
    outs = []
    for env_over in (dict(CATPPO_FUSED_FWD="1", CATPPO_FUSED_FWD_MIN_ROWS="1", CATPPO_ROWS_FWD_ROLLOUT="0", CATPPO_ROWS_WIDE="0", CATPPO_STEP16_FWD="0"),
                     dict(CATPPO_FUSED_FWD="0", CATPPO_ROWS_FWD_ROLLOUT="0", CATPPO_ROWS_WIDE="0", CATPPO_STEP16_FWD="0")):
        out = str(tmp_path / f"ff{len(outs)}.npz")
        code = _FUSED_FWD_AB.format(root=root, cases=cases, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_over), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    f, l = outs
    for k in f.files:
        if k.endswith("_e3"):
            np.testing.assert_array_equal(f[k], l[k], err_msg=k)             # same Philox counters -> same noise
        else:
            np.testing.assert_allclose(f[k], l[k], rtol=0, atol=2e-6 * max(1.0, float(np.abs(l[k]).max())), err_msg=k)
    assert np.abs(f["cfg2_act"]).max() > 0 and np.isfinite(f["ref_ragged_lp"]).all()
