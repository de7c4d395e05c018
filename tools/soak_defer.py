"""Soak of the deferred post tail (catppo_rollout_defer_tail): N iterations of a bench workload twice - tail inside the post
launch / tail deferred into the next rollout_pre launch - and a digest of everything the env step and the trainer hold at
the end.  The two digests must be equal: a rare ordering bug between the post launch, the policy forward and the tail (three
iterations in tests/test_gpu_r5.py would not see a 1-in-1000 event) shows up as a mismatch after thousands of env steps.

    python tools/soak_defer.py [workload] [iterations]        (on the GPU box)
"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "constraints-as-terminations_amd"))
import bench  # noqa: E402


def run(workload, iters, defer):
    torch.manual_seed(3)
    env, tr, _ = bench.build(workload, 1, 0, None, 1, 0, {"defer_rollout_tail": defer})
    assert tr.sink is not None and tr.defer_tail == defer
    for _ in range(iters):
        tr.run_iteration(log=False)
    torch.cuda.synchronize()
    eu = env.unwrapped
    cm, rms = eu.constraint_manager, tr.agent.obs_rms
    parts = dict(flat=tr.agent.flat, rm=cm.cat._p_rm, ring=cm._log_ring, ep_viol=cm._ep_viol, ep_prob=cm._ep_prob,
                 mean=rms.running_mean, var=rms.running_var, count=rms.count, obs=tr.obs, dones=tr.dones.float(),
                 rewards=tr.rewards.float(), ep_len=eu.episode_length_buf, state=eu.sim.cur)
    return {k: hashlib.sha256(v.contiguous().cpu().numpy().tobytes()).hexdigest()[:16] for k, v in parts.items()}


def main():
    if os.environ.get("CATPPO_FORCE_DIST") == "1":      # every exchange point on over a world of one (gathered records)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.update(RANK="0", WORLD_SIZE="1")
        from cat_envs import parallel
        parallel.init_rendezvous(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        a = run(wl, iters, False)
        b = run(wl, iters, True)
    bad = [k for k in a if a[k] != b[k]]
    T = bench.WORKLOADS[wl]["num_steps"]
    mode = "every exchange point forced on (world of one), " if os.environ.get("CATPPO_FORCE_DIST") == "1" else ""
    print(f"{wl}: {mode}{iters} iterations = {iters * T} env steps per mode; digests", "EQUAL" if not bad else f"DIFFER in {bad}")
    for k in a:
        print(f"  {k:8s} {a[k]} {b[k]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
