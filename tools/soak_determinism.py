"""Determinism soak of the round-6 kernels: N iterations of a bench workload TWICE from the same seeds; the sha256 digests of
the parameters, the Adam state and the rollout buffers at the end must be equal and every value finite.  Every kernel of the
path sums in a fixed order (no atomics on data), so a digest mismatch means a race (an LDS hazard, a missing barrier, a read of
memory another workgroup is still writing) - what a three-iteration parity test can miss.

    python tools/soak_determinism.py [workload ...] [--iters 100]          (on the GPU box)
    workloads of interest: cfg3_shard (step16_kernel + dw_multi_kernel + step16_fwd_kernel), cfg5 (bf16-stored activations,
    IILoop16, fwd0_w16_kernel), cfg1 (step16 at 512-row minibatches), cfg2 (lean critic epilogue of fwd_head_kernel)
"""
import contextlib
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "constraints-as-terminations_amd"))
import bench  # noqa: E402


def run(workload, iters):
    torch.manual_seed(11)
    env, tr, _ = bench.build(workload, 5, 0, None, 1, 0, {})
    stats = None
    for i in range(iters):
        stats = tr.run_iteration(log=(i == iters - 1))
    torch.cuda.synchronize()
    parts = dict(flat=tr.agent.flat, exp_avg=tr.exp_avg, exp_avg_sq=tr.exp_avg_sq, obs=tr.obs, values=tr.values.float(),
                 logprobs=tr.logprobs.float(), rewards=tr.rewards.float(), advantages=tr.advantages.float())
    finite = all(bool(torch.isfinite(v.float()).all()) for v in parts.values())
    dig = {k: hashlib.sha256(v.contiguous().cpu().numpy().tobytes()).hexdigest()[:16] for k, v in parts.items()}
    return dig, finite, stats


def main():
    argv = sys.argv[1:]
    iters = 100
    if "--iters" in argv:
        i = argv.index("--iters")
        iters = int(argv[i + 1])
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    rc = 0
    for wl in (args or ["cfg3_shard", "cfg5", "cfg1", "cfg2"]):
        with contextlib.redirect_stdout(sys.stderr):
            a, fa, sa = run(wl, iters)
            b, fb, _ = run(wl, iters)
        bad = [k for k in a if a[k] != b[k]]
        w = bench.WORKLOADS[wl]
        steps = iters * 5 * max(1, (w["num_steps"] * w["num_envs"]) // w.get("minibatch", 16384))
        print(f"{wl}: {iters} iterations twice (~{steps} optimiser steps, {iters * w['num_steps']} env steps each); finite {fa and fb}; "
              f"digests {'EQUAL' if not bad else 'DIFFER in ' + str(bad)}; last iteration: "
              + ", ".join(f"{k} {v:.4g}" for k, v in (sa or {}).items() if isinstance(v, float)))
        rc |= int(bool(bad) or not (fa and fb))
    return rc


if __name__ == "__main__":
    sys.exit(main())
