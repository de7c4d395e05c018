"""Per-kernel-group timings on one MI355X (HIP events on torch's current stream, which is the
stream every libcatppo call is enqueued on).  Prints one JSON object per line.

    python tools/microbench.py [--reps 50]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraints-as-terminations_amd"))
from cat_envs import native  # noqa: E402


def peak_frac(algorithmic_tflops, prec):
    """roofline fraction against the peak of the matrix instruction the mode executes: fp32 MFMA 157.3 TFLOP/s; bf16 MFMA
    2500 TFLOP/s dense, with the split-bf16 mode executing three MFMAs per algorithmic product (never an fp32 fraction)"""
    if prec == 0:
        return {"frac_157TF_fp32_mfma": algorithmic_tflops / 157.3}
    factor = 3.0 if prec == 2 else 1.0
    return {"frac_2500TF_bf16_mfma": algorithmic_tflops * factor / 2500.0, "executed_over_algorithmic_flops": factor}


def timeit(fn, reps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    nat = native.Native()
    dev = "cuda"
    out = []

    torch.cuda.synchronize(); print("[microbench] next: GAE: config size and HBM-roofline sweep (24 B per env-step)", file=sys.stderr, flush=True)
    # ---- GAE: config size and HBM-roofline sweep (24 B per env-step)
    for T, N in [(24, 4096), (48, 4096), (24, 32768), (24, 1 << 18), (24, 1 << 20), (48, 1 << 22)]:
        x = [torch.rand(T, N, device=dev) for _ in range(4)]
        nv, nd, ntd = (torch.rand(N, device=dev) for _ in range(3))
        adv, ret = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
        us = timeit(lambda: nat.gae(x[0], x[1], x[2], x[3], nv, nd, ntd, 0.99, 0.95, adv, ret), a.reps)
        byt = 24 * T * N + 12 * N
        out.append(dict(kernel="gae", T=T, N=N, us=us, GBps=byt / us / 1e3, frac_8TBps=byt / us / 1e3 / 8000))
        del x, adv, ret

    torch.cuda.synchronize(); print("[microbench] next: GAE on fp16 planes (12 B per env-step)", file=sys.stderr, flush=True)
    # ---- GAE on fp16 planes (12 B per env-step)
    for T, N in [(24, 32768), (24, 1 << 20), (48, 1 << 22)]:
        x = [torch.rand(T, N, device=dev).half() for _ in range(4)]
        nv, nd, ntd = (torch.rand(N, device=dev).half() for _ in range(3))
        adv, ret = (torch.empty(T, N, device=dev, dtype=torch.float16) for _ in range(2))
        us = timeit(lambda: nat.gae_f16(x[0], x[1], x[2], x[3], nv, nd, ntd, 0.99, 0.95, adv, ret), a.reps)
        byt = 12 * T * N + 6 * N
        out.append(dict(kernel="gae[fp16 planes]", T=T, N=N, us=us, GBps=byt / us / 1e3, frac_8TBps=byt / us / 1e3 / 8000))
        del x, adv, ret

    torch.cuda.synchronize(); print("[microbench] next: GAE scan mode (wavefront-shuffle scan over time) at the size", file=sys.stderr, flush=True)
    # ---- GAE scan mode (wavefront-shuffle scan over time) at the sizes where the serial kernel is latency bound
    for T, N in [(24, 4096), (48, 4096), (24, 2048), (24, 32768)]:
        x = [torch.rand(T, N, device=dev) for _ in range(4)]
        nv, nd, ntd = (torch.rand(N, device=dev) for _ in range(3))
        adv, ret = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
        for mode, name in ((native.GAE_SERIAL, "gae serial"), (native.GAE_SCAN, "gae scan")):
            us = timeit(lambda: nat.gae_mode(mode, x[0], x[1], x[2], x[3], nv, nd, ntd, 0.99, 0.95, adv, ret), a.reps)
            out.append(dict(kernel=name + " (launch loop: launch-rate bound below ~7 us)", T=T, N=N, us=us,
                            GBps=(24 * T * N + 12 * N) / us / 1e3))

    torch.cuda.synchronize(); print("[microbench] next: CaT step", file=sys.stderr, flush=True)
    # ---- CaT step
    for N, widths in [(4096, [12, 12, 1, 4, 12, 1]), (4096, [12, 12, 12, 12, 1, 4, 2, 1, 4, 1, 4, 12, 1]),
                      (32768, [12, 12, 12, 12, 1, 4, 2, 1, 4, 1, 4, 12, 1])]:
        K, nt = sum(widths), len(widths)
        off = (C.c_int32 * (nt + 1))(*np.concatenate([[0], np.cumsum(widths)]).tolist())
        dp = (C.c_float * nt)(*([0.25] * nt))
        cstr = torch.randn(N, K, device=dev)
        rm, prob, dones = torch.ones(K, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        viol, eprob = torch.zeros(nt, N, device=dev), torch.zeros(nt, N, device=dev)
        rew = torch.rand(N, device=dev)
        reset = torch.rand(N, device=dev) < 0.01
        us = timeit(lambda: nat.cat_step(cstr, off, dp, 0.0, 0.95, False, rm, prob, viol, eprob, reward=rew,
                                         reset_mask=reset, dones=dones), a.reps)
        byt = N * (4 * K + 4 + 4 + 8 + 16 * nt)
        out.append(dict(kernel="cat_step(3 launches)", N=N, K=K, n_terms=nt, us=us, GBps=byt / us / 1e3))

    torch.cuda.synchronize(); print("[microbench] next: obs normaliser", file=sys.stderr, flush=True)
    # ---- obs normaliser
    for N, D in [(4096, 45), (4096, 235), (98304, 1)]:
        x = torch.randn(N, D, device=dev)
        ldo = (D + 15) // 16 * 16 if D > 1 else 1
        o = torch.zeros(N, ldo, device=dev)
        m, v, c = torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.ones(1, device=dev)

        def f():
            nat.rms_update(x, N, D, D, m, v, c)
            nat.rms_normalize(x, N, D, D, m, v, 1e-8, o, ldo)
        us = timeit(f, a.reps)
        out.append(dict(kernel="rms update+normalize(4 launches)", N=N, D=D, us=us, GBps=N * D * 12 / us / 1e3))

    torch.cuda.synchronize(); print("[microbench] next: MLP", file=sys.stderr, flush=True)
    # ---- MLP
    for D, hidden, bf16 in [(45, (512, 256, 128), 0), (48, (256, 256, 256), 0), (48, (256, 256, 256), 2),
                            (48, (256, 256, 256), 1)]:
        A = 12
        shape = native.shape_of(D, A, hidden, mfma_bf16=bf16)
        tag = {0: "", 1: "[bf16 operands]", 2: "[bf16x3 split operands]"}[bf16]
        lay = native.layout_of(shape)
        dims = [lay.obs_pad, *hidden]
        macs_true = sum(i * o for i, o in zip([D, *hidden[:-1]], hidden)) * 2 + hidden[-1] * (A + 1)
        params = torch.randn(lay.n_flat, device=dev) * 0.05
        for N in (4096,):
            x = torch.randn(N, lay.obs_pad, device=dev)
            eps = torch.randn(N, A, device=dev)
            act, lp, val = torch.empty(N, A, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
            nat.mlp_reserve(shape, 16384)
            us = timeit(lambda: nat.policy_act(shape, params, x, N, eps, act, lp, val), a.reps)
            fl = 2 * macs_true * N
            out.append(dict(kernel="policy_act" + tag, arch=list(hidden), D=D, N=N, us=us, TFLOPs=fl / us / 1e6,
                            **peak_frac(fl / us / 1e6, bf16)))
        B, M = 98304, 16384
        obs, acts = torch.randn(B, lay.obs_pad, device=dev), torch.randn(B, A, device=dev)
        logp, adv, ret, val = (torch.randn(B, device=dev) for _ in range(4))
        logp -= 14.0
        inds = torch.randperm(B, device=dev)[:M].contiguous()
        vm, vv = torch.zeros(1, device=dev), torch.ones(1, device=dev)
        grad, diag = torch.zeros(lay.n_flat, device=dev), torch.zeros(8, device=dev)
        hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / M, 0)
        us = timeit(lambda: nat.ppo_minibatch_grad(shape, hp, params, obs, acts, logp, adv, ret, val, inds, vm, vv,
                                                   None, grad, diag), max(a.reps // 2, 5))
        fl = 6 * macs_true * M
        out.append(dict(kernel="ppo_minibatch_grad" + tag, arch=list(hidden), D=D, M=M, us=us, TFLOPs=fl / us / 1e6,
                        **peak_frac(fl / us / 1e6, bf16)))
        m1, m2 = torch.zeros(lay.n_flat, device=dev), torch.zeros(lay.n_flat, device=dev)
        us = timeit(lambda: nat.clip_adam(params, grad, m1, m2, lay.n_flat, 1.0, 3e-4, 0.9, 0.999, 1e-5, 3), a.reps)
        out.append(dict(kernel="clip_adam(2 launches)", n=int(lay.n_flat), us=us, GBps=lay.n_flat * 28 / us / 1e3))
        if bf16 == 0:
            st = nat.iter_state_new(1, 3e-4)
            nat.iter_begin(st, 3e-4, 10, native.LR_FIXED)
            us = timeit(lambda: nat.clip_adam_dev(params, grad, m1, m2, lay.n_flat, 1.0, 0.9, 0.999, 1e-5, st), a.reps)
            out.append(dict(kernel="clip_adam_dev(2 launches, lr/step on the device)", n=int(lay.n_flat), us=us))
            x = torch.randn(4096, lay.obs_pad, device=dev)
            act_r, lp_r, val_r = (torch.empty(4096, A, device=dev), torch.empty(4096, device=dev),
                                  torch.empty(4096, device=dev))
            us = timeit(lambda: nat.policy_act_rng(shape, params, x, 4096, st, 3, act_r, lp_r, val_r), a.reps)
            out.append(dict(kernel="policy_act_rng (Philox noise in the head kernel)", arch=list(hidden), N=4096, us=us))
            n_mb, parts = B // M, (M + nat.GATHER_ROWS - 1) // nat.GATHER_ROWS
            xg, ag = torch.empty(B, lay.obs_pad, device=dev), torch.empty(B, A, device=dev)
            sg, ap = torch.empty(4 * B, device=dev), torch.empty(n_mb * parts * 2, dtype=torch.float64, device=dev)
            perm = torch.randperm(B, device=dev)
            us = timeit(lambda: nat.ppo_gather_ex(shape, obs, acts, logp, adv, ret, val, B, M, xg, ag, sg, ap, st=st), a.reps)
            us2 = timeit(lambda: nat.ppo_gather_ex(shape, obs, acts, logp, adv, ret, val, B, M, xg, ag, sg, ap, inds=perm),
                         a.reps)
            us3 = timeit(lambda: torch.randperm(B, device=dev), a.reps)
            out.append(dict(kernel="epoch gather, keyed permutation inside", B=B, us=us, with_index_array_us=us2,
                            torch_randperm_alone_us=us3))

    for o in out:
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items()}))


if __name__ == "__main__":
    main()
