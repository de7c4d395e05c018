"""Env-sharded path on CPU: world_size 2, gloo.  The exchange helpers of cat_envs.parallel are run
on two ranks holding the two halves of the envs / of a minibatch and must reproduce the
single-process oracle on the union of the shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import streams as S
    from cat_envs import parallel
    from oracle import cat_oracle as CO
    from oracle import ppo_oracle as PO
    res = {}
    assert parallel.world_size() == world and parallel.rank() == rank and parallel.active()

    # ---- 1. CaT: local column max -> MAX all-reduce -> each shard's probabilities == its rows of the
    #         single-process result (bit-exact: max is order independent)
    N, terms = 101, S.CAT_TERMS_SMALL            # odd N: shards of 51 / 50 envs
    stream = S.cat_stream(21, N, terms, 5)
    sl = parallel.shard_slice(N, rank, world)
    full = CO.CaTOracle(0.95, 0.0)
    rm = {}
    for step in stream:
        for (name, width, _), mp_ in zip(terms, [0.25, 1.0, 0.25, 1.0, 0.5]):
            full.add(name, step[name], mp_)
            c = CO._as_2d_f32(np.asarray(step[name]))[sl]
            cmax = torch.from_numpy(np.maximum(c.max(0), np.float32(1e-6)))
            parallel.allreduce_max_(cmax)
            cmax = cmax.numpy()
            rm[name] = cmax if name not in rm else (rm[name] * np.float32(0.95) + np.float32(1.0 - 0.95) * cmax).astype(np.float32)
            q = np.clip((c / rm[name]).astype(np.float32), 0, 1)
            p = np.where(c > 0, (np.float32(0.0) + (q * np.float32(mp_ - 0.0)).astype(np.float32)).astype(np.float32), np.float32(0))
            assert np.array_equal(rm[name], full.running_maxes[name][0])
            assert np.array_equal(p, full.probs[name][sl])
    res["cat"] = True

    # ---- 2. normaliser: fp64 moment sums all-reduced == single-process update on all rows
    rs = np.random.RandomState(5)
    x = (rs.standard_normal((64, 45)) * 3 + 1).astype(np.float32)
    xs = torch.from_numpy(x[parallel.shard_slice(64, rank, world)]).double()
    sums = torch.cat([xs.sum(0), (xs * xs).sum(0), torch.tensor([float(xs.shape[0])], dtype=torch.float64)])
    parallel.global_moment_sums(sums)
    n = float(sums[-1])
    mean = sums[:45] / n
    var = sums[45:90] / n - mean * mean
    ref = PO.RMSOracle((45,))
    ref.update(torch.from_numpy(x))
    bm, bv = torch.mean(torch.from_numpy(x), 0), torch.var(torch.from_numpy(x), correction=0, dim=0)
    assert n == 64 and torch.allclose(mean.float(), bm, rtol=1e-6, atol=1e-6) and torch.allclose(var.float(), bv, rtol=1e-5)
    res["rms"] = True

    # ---- 3. minibatch: per-rank gradient with 1/M_global scaling + global advantage statistics,
    #         SUM all-reduce of the flat gradient == single-process gradient on the whole minibatch
    D, A, hidden, M = 45, 12, (64, 64, 64), 256
    w = S.agent_weights(9, D, A, hidden)
    rs = np.random.RandomState(10)
    batch = dict(obs=rs.standard_normal((M, D)).astype(np.float32), act=rs.standard_normal((M, A)).astype(np.float32) * 0.5,
                 adv=rs.standard_normal(M).astype(np.float32), ret=rs.standard_normal(M).astype(np.float32),
                 val=rs.standard_normal(M).astype(np.float32))
    cfg = dict(clip_coef=0.2, ent_coef=0.001, vf_coef=2.0, norm_adv=True, clip_vloss=True)

    def make_agent():
        ag = PO.AgentOracle(D, A, hidden)
        ag.load(w)
        return ag
    ag0 = make_agent()
    with torch.no_grad():
        _, lp0, _, _ = ag0.get_action_and_value(torch.from_numpy(batch["obs"]), torch.from_numpy(batch["act"]))
    batch["logp"] = (lp0.numpy() + rs.standard_normal(M).astype(np.float32) * 0.2).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    # single process
    ag = make_agent()
    ps = [p.requires_grad_(True) for p in ag.parameters()]
    loss, _ = PO.ppo_minibatch_loss(ag, t["obs"], t["act"], t["logp"], t["adv"], t["ret"], t["val"], cfg)
    loss.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ps])
    # sharded
    sl = parallel.shard_slice(M, rank, world)
    stats = parallel.global_adv_stats(t["adv"][sl])
    a_all = t["adv"].double()
    assert abs(float(stats[0]) - float(a_all.mean())) < 1e-6 and abs(float(stats[1]) - (float(a_all.std()) + 1e-8)) < 1e-6
    ag = make_agent()
    ps = [p.requires_grad_(True) for p in ag.parameters()]
    adv_n = (t["adv"][sl] - stats[0]) / stats[1]
    cfg_local = dict(cfg, norm_adv=False)
    loss, _ = PO.ppo_minibatch_loss(ag, t["obs"][sl], t["act"][sl], t["logp"][sl], adv_n, t["ret"][sl], t["val"][sl], cfg_local)
    (loss * (adv_n.numel() / M)).backward()       # mean over the local shard -> 1/M_global scaling
    g = torch.cat([p.grad.reshape(-1) for p in ps])
    parallel.allreduce_sum_(g)
    assert float((g - g_ref).abs().max()) < 1e-6 * max(1.0, float(g_ref.abs().max())), float((g - g_ref).abs().max())
    res["grad"] = True

    # ---- 3b. ragged shards (51 / 50 rows): the TRUE global row count travels with the sums (round 1 assumed equal
    #          shards); exchange buffer of the fused rollout step = {colmax MAX | moment sums SUM}
    xr = (rs.standard_normal((101, 7)) * 2 - 0.5).astype(np.float32)
    slr = parallel.shard_slice(101, rank, world)
    loc = torch.from_numpy(xr[slr]).double()
    sums = torch.cat([loc.sum(0), (loc * loc).sum(0)])
    cnt = torch.tensor([float(loc.shape[0])], dtype=torch.float64)
    colmax = torch.from_numpy(np.maximum(xr[slr].max(0), np.float32(1e-6)))
    parallel.allreduce_sum_(sums)
    parallel.allreduce_sum_(cnt)
    parallel.allreduce_max_(colmax)
    assert float(cnt) == 101.0 and np.array_equal(colmax.numpy(), np.maximum(xr.max(0), np.float32(1e-6)))
    allx = torch.from_numpy(xr).double()
    assert torch.allclose(sums[:7] / cnt, allx.mean(0), rtol=1e-12) and \
        torch.allclose(sums[7:] / cnt - (sums[:7] / cnt) ** 2, allx.var(0, correction=0), rtol=1e-9)
    # ---- 3c. KL of the adaptive schedule: each rank's diag carries sum_i kl_i / (M_local * world); the SUM over ranks is
    #          the global mean (skrl: all_reduce(kl, SUM) / world_size, skrl/ppo.py:562-564)
    kl_rows = rs.uniform(0, 0.05, 256).astype(np.float64)
    slk = parallel.shard_slice(256, rank, world)
    kl_local = torch.tensor([kl_rows[slk].sum() / (len(kl_rows[slk]) * world)], dtype=torch.float64)
    parallel.allreduce_sum_(kl_local)
    assert abs(float(kl_local) - kl_rows.mean()) < 1e-12
    # no HIP device here: the native RCCL communicator is not used, torch.distributed carries the exchange
    assert not parallel.native_comm_active()

    # ---- 3d. the native communicator set-up is all-or-nothing: rank 1 cannot join -> BOTH ranks keep
    #          torch.distributed and know why; then rank 0 cannot even make an id -> same outcome, nobody hangs
    class _Nat:
        comm_world = 0
        device = torch.device("cpu")

        def __init__(self, fail_id=False, fail_init=False, wrong_max=False):
            self.fail_id, self.fail_init, self.destroyed, self.init_calls = fail_id, fail_init, False, 0
            self.wrong_max = wrong_max

        # the collectives of a communicator that did come up (here: carried by the test's gloo group)
        def allreduce(self, t, op=0):
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
            if op == 1 and self.wrong_max:
                t[3] += 1.0

        def broadcast(self, t, root=0):
            dist.broadcast(t, src=root)

        def allgather(self, send, recv):
            dist.all_gather_into_tensor(recv, send)

        def comm_unique_id(self):
            if self.fail_id:
                raise RuntimeError("catppo_comm_unique_id failed (-6): librccl could not be loaded")
            return bytes(128)

        def comm_probe(self):
            if self.fail_id:
                raise RuntimeError("catppo_comm_probe failed: librccl could not be loaded")

        def comm_init(self, r, w, uid):
            assert len(uid) == 128 and w == world
            self.init_calls += 1
            if self.fail_init:
                raise RuntimeError("libcatppo error -6: ncclCommInitRank: unhandled system error")

        def comm_destroy(self):
            self.destroyed = True

    nat_a = _Nat(fail_init=(rank == 1))
    assert parallel.init_native_comm(nat_a) is False and not parallel.native_comm_active()
    assert "rank 1" in parallel.native_comm_error() and "rank 0" not in parallel.native_comm_error()
    assert nat_a.destroyed == (rank == 0)          # the rank that HAD joined left again
    nat_b = _Nat(fail_id=(rank == 0))
    assert parallel.init_native_comm(nat_b) is False and "unique id" in parallel.native_comm_error() or \
        "librccl" in parallel.native_comm_error()
    assert nat_b.init_calls == 0                   # nobody entered the (blocking) communicator set-up
    # a rank that cannot load librccl at all (precondition vote): the healthy rank must NOT enter ncclCommInitRank,
    # where it would wait for the missing peer forever
    nat_c = _Nat(fail_id=(rank == 1))
    assert parallel.init_native_comm(nat_c) is False and "rank 1" in parallel.native_comm_error()
    assert nat_c.init_calls == 0 and not nat_c.destroyed
    # a communicator that comes up but returns a wrong MAX on one rank (known-answer pass after the set-up, round 4):
    # both ranks leave it again and say which collective on which rank
    nat_d = _Nat(wrong_max=(rank == 1))
    assert parallel.init_native_comm(nat_d) is False and nat_d.destroyed
    assert parallel.native_comm_error() == "rank 1: wrong result of MAX fp32", parallel.native_comm_error()
    # ... and one whose collectives all answer correctly is taken
    nat_e = _Nat()
    assert parallel.init_native_comm(nat_e) is True and parallel.native_comm_active()
    parallel.shutdown_native_comm()
    assert nat_e.destroyed and not parallel.native_comm_active()
    assert parallel.gather_counts(50 + rank) == [50, 51]
    chk = torch.ones(1)
    parallel.allreduce_sum_(chk)                   # the fallback transport still works
    assert float(chk) == world

    # ---- 3b. the fused env step's ONE exchange per step: every rank's byte record, in rank order, on every rank
    rec = torch.full((24,), rank + 1, dtype=torch.uint8)
    allr = torch.zeros(24 * world, dtype=torch.uint8)
    parallel.allgather_bytes_(rec, allr)
    assert allr.view(world, 24).eq(torch.arange(1, world + 1, dtype=torch.uint8)[:, None]).all()

    # ---- 4. broadcast of the flat parameters from rank 0
    flat = torch.full((10,), float(rank))
    parallel.broadcast_(flat, src=0)
    assert float(flat.sum()) == 0.0
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(",".join(sorted(res)))
    dist.destroy_process_group()


def test_env_sharding_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ok{r}").read() == "cat,grad,rms"


def test_shard_slices_cover_everything():
    sys.path.insert(0, os.path.join(ROOT, "constraints-as-terminations_amd"))
    from cat_envs import parallel
    for n in (1, 7, 4096, 16385):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                s = parallel.shard_slice(n, r, w)
                got += list(range(n))[s]
            assert got == list(range(n))
    assert parallel.world_size() == 1 and not parallel.active()


def test_minibatch_plan_is_identical_on_every_rank_for_ragged_shards():
    """ceil(T*N_r / M) differs between ranks when the env shards differ by one (cfg3 over 3 ranks:
    5462/5461/5461 envs, per-rank minibatch 5461 -> 25 vs 24 minibatches): the plan must give every rank the same
    number of optimiser steps, non-empty minibatches, and the true global row count of each."""
    sys.path.insert(0, os.path.join(ROOT, "constraints-as-terminations_amd"))
    import pytest
    from cat_envs import parallel
    rows = [24 * 5462, 24 * 5461, 24 * 5461]
    n_mb, m_r, per_rank, glob = parallel.minibatch_plan(rows, 16384 // 3)
    assert n_mb == 24 and m_r == [5462, 5461, 5461]
    assert all(len(pr) == n_mb and min(pr) >= 1 and sum(pr) == b for pr, b in zip(per_rank, rows))
    assert glob == [sum(pr[k] for pr in per_rank) for k in range(n_mb)] and sum(glob) == sum(rows)
    # equal shards: exactly the single-process schedule of each rank
    n_mb, m_r, per_rank, glob = parallel.minibatch_plan([98304, 98304], 16384)
    assert n_mb == 6 and m_r == [16384, 16384] and glob == [32768] * 6
    # one rank: ceil(B / M) minibatches, the last one short
    n_mb, m_r, per_rank, glob = parallel.minibatch_plan([1000], 384)
    assert n_mb == 3 and m_r == [384] and per_rank == [[384, 384, 232]] and glob == [384, 384, 232]
    # 2049 envs over 2 ranks x 24 steps, 4096-row minibatches
    n_mb, m_r, per_rank, glob = parallel.minibatch_plan([24 * 1025, 24 * 1024], 4096)
    assert n_mb == 6 and m_r == [4100, 4096] and glob == [8196] * 6
    with pytest.raises(ValueError):
        parallel.minibatch_plan([59, 58], 2)      # 29 minibatches of ceil(59/29) = 3 rows: rank 0 runs out after 20


# ------------------------------------------------------------------ communicator set-up votes (round 5)
class _FakeNat:
    """libcatppo's communicator calls as cat_envs.parallel.init_native_comm uses them, over the gloo group itself:
    `scenario` injects the failure one rank sees"""

    def __init__(self, rank, scenario):
        self.rank, self.scenario, self.comm_world = rank, scenario, 0
        self.device = torch.device("cpu")
        self.inits = self.destroys = 0

    def comm_unique_id(self):
        return b"u" * 128

    def comm_probe(self):
        if self.scenario == "probe_fails_on_rank1" and self.rank == 1:
            raise OSError("librccl.so: cannot open shared object file")

    def comm_init(self, r, w, uid):
        assert uid == b"u" * 128
        if self.scenario == "init_fails_on_rank0" and r == 0:
            raise RuntimeError("libcatppo error -3: ncclCommInitRank: unhandled system error")
        self.inits += 1
        self.comm_world = w

    def comm_destroy(self):
        self.destroys += 1
        self.comm_world = 0

    def allreduce(self, t, op):
        if self.scenario == "wrong_sum_on_rank1" and self.rank == 1 and t.dtype == torch.float32 and op == 0:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t += 1.0
            return
        if t.dtype == torch.float16:                     # gloo has no fp16 SUM: widen
            w = t.float()
            dist.all_reduce(w, op=dist.ReduceOp.SUM)
            t.copy_(w.half())
            return
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)

    def broadcast(self, t, src):
        dist.broadcast(t, src)

    def allgather(self, send, recv):
        parts = [torch.empty_like(send) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, send)
        recv.copy_(torch.cat(parts))


def _vote_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cat_envs import parallel
    parallel.init_rendezvous(None, timeout_s=120)
    assert dist.get_backend() == "gloo" and parallel.active()
    done = []
    for scenario in ("probe_fails_on_rank1", "init_fails_on_rank0", "wrong_sum_on_rank1", "healthy"):
        nat = _FakeNat(rank, scenario)
        ok = parallel.init_native_comm(nat)
        err = parallel.native_comm_error()
        if scenario == "healthy":
            assert ok and parallel.native_comm_active() and nat.inits == 1 and nat.destroys == 0
            assert err is None, err                      # the earlier scenarios' reasons do not stick to a healthy set-up
            assert parallel.reinit_native_comm() and nat.inits == 2 and nat.destroys == 1     # fresh communicator, both ranks
            parallel.shutdown_native_comm()
            assert not parallel.native_comm_active() and nat.destroys == 2
        else:
            # EVERY rank ends up on the fallback, with the failing rank named - nobody is left waiting inside the set-up
            assert not ok and not parallel.native_comm_active() and nat.comm_world == 0, (scenario, ok)
            want = {"probe_fails_on_rank1": "rank 1: librccl.so", "init_fails_on_rank0": "rank 0: libcatppo error -3",
                    "wrong_sum_on_rank1": "rank 1: wrong result of SUM fp32"}[scenario]
            assert err is not None and want in err, (scenario, err)
            if scenario == "probe_fails_on_rank1":
                assert nat.inits == 0                    # nobody entered the blocking communicator set-up
            if scenario == "init_fails_on_rank0":
                assert nat.destroys == (1 if rank == 1 else 0)      # the rank whose set-up succeeded gives it back
        done.append(scenario)
    open(os.path.join(out_dir, f"votes{rank}"), "w").write(",".join(done))
    dist.destroy_process_group()


def test_communicator_setup_votes_keep_every_rank_on_one_transport_world2_gloo(tmp_path):
    """cat_envs.parallel.init_native_comm (VERDICT r4 item 4 / ADVICE r4): a failure that only ONE rank sees - librccl not
    loadable, ncclCommInitRank failing, a wrong known-answer result - must leave BOTH ranks on the fallback transport with
    the failing rank named, and never one rank inside the blocking set-up waiting for the other; a healthy set-up is kept,
    can be re-created collectively (after an aborted graph capture) and torn down.  Two real processes on gloo."""
    port = _free_port()
    mp.spawn(_vote_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"votes{r}").read() == "probe_fails_on_rank1,init_fails_on_rank0,wrong_sum_on_rank1,healthy"
