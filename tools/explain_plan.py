"""Which kernels does a shape get?  

    python tools/explain_plan.py [--obs 48] [--act 12] [--hidden 256,256,256] [--envs 4096] [--minibatch 16384]
                                 [--precision fp32|bf16|bf16x3]          (needs the MI355X: the plan is RECORDED, not re-derived)

Runs ONE rollout forward (catppo_policy_step on `--envs` rows) and ONE optimiser step
(catppo_ppo_minibatch_step_packed on `--minibatch` rows) on random data with catppo_plan_log switched on and prints what
the library's dispatch code wrote at its decision sites: kernel, grid, and the rule that selected it.  `--all` walks the
BASELINE shapes (profiles/r5_explain_plan.txt is this output)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def explain(obs, act, hidden, envs, mb, precision="fp32"):
    from cat_envs import native
    from cat_envs.tasks.utils.cleanrl.ppo import MLP_PRECISIONS
    nat = native.get(torch.device("cuda", 0))
    dev = nat.device
    shape = native.shape_of(obs, act, hidden, mfma_bf16=MLP_PRECISIONS[precision])
    lay = native.layout_of(shape)
    nat.mlp_reserve(shape, max(envs, mb))
    g = torch.Generator(device="cpu").manual_seed(1)
    flat = (torch.randn(lay.n_flat, generator=g) * 0.05).to(dev)
    x = torch.zeros(max(envs, mb), lay.obs_pad, device=dev)
    x[:, :obs] = torch.randn(max(envs, mb), obs, generator=g).to(dev)
    st = nat.iter_state_new(7, 3e-4)
    nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
    a, lp, v = torch.empty(envs, act, device=dev), torch.empty(envs, device=dev), torch.empty(envs, device=dev)
    out = [f"== obs {obs} (padded {lay.obs_pad}), act {act}, hidden {tuple(hidden)}, {envs} envs, minibatch {mb}, {precision}"]
    nat.plan_log(1)
    nat.policy_act_rng(shape, flat, x, envs, st, 0, a, lp, v)
    actg = torch.randn(mb, act, generator=g).to(dev)
    scal = torch.randn(4 * mb, generator=g).to(dev)
    scal[:mb] = -11.0
    parts = (mb + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
    advp = torch.zeros(2 * parts, dtype=torch.float64, device=dev)
    advp[0], advp[1] = float(scal[mb:2 * mb].double().sum()), float((scal[mb:2 * mb].double() ** 2).sum())
    hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / mb, 0)
    grad, diag = torch.zeros(lay.n_flat, device=dev), torch.zeros(8, device=dev)
    m1, m2 = torch.zeros(lay.n_flat, device=dev), torch.zeros(lay.n_flat, device=dev)
    one, zero = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    nat.ppo_minibatch_step_packed(shape, hp, flat, x, actg, scal, advp, mb, zero, one, None, grad, diag, m1, m2, 1.0, 0.9,
                                  0.999, 1e-5, st)
    torch.cuda.synchronize()
    out.append(nat.plan_log(0).rstrip())
    return "\n".join(out)


BASELINE = [
    ("cfg1: 64 envs, 2 terms, reference MLP", 48, 12, (512, 256, 128), 64, 512, "fp32"),
    ("cfg2: the bench default", 48, 12, (256, 256, 256), 4096, 16384, "fp32"),
    ("reference shapes", 45, 12, (512, 256, 128), 4096, 16384, "fp32"),
    ("cfg3: one rank's share at 8 GPUs", 45, 12, (512, 256, 128), 2048, 2048, "fp32"),
    ("cfg4: 235-d observations", 235, 12, (256, 256, 256), 4096, 16384, "fp32"),
    ("cfg5: bf16 MLP", 48, 12, (256, 256, 256), 32768, 16384, "bf16"),
    ("cfg2, split-bf16 (the secondary record)", 48, 12, (256, 256, 256), 4096, 16384, "bf16x3"),
]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs", type=int, default=48)
    ap.add_argument("--act", type=int, default=12)
    ap.add_argument("--hidden", default="256,256,256")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--minibatch", type=int, default=16384)
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16", "bf16x3"))
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    if a.all:
        for name, *spec in BASELINE:
            print("#", name)
            print(explain(*spec))
            print()
    else:
        print(explain(a.obs, a.act, tuple(int(h) for h in a.hidden.split(",")), a.envs, a.minibatch, a.precision))
