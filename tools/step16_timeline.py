"""Phase timeline of step16_kernel (round 6; wall-clock stamps of thread 0 of every workgroup).  Needs a library built with
-DSTEP16_TL:
    python constraints-as-terminations_amd/build.py --variant s16tl -DSTEP16_TL
    CATPPO_LIB=$PWD/tools/bin/libcatppo_s16tl.so python tools/step16_timeline.py [rows] [ref|cfg2]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from cat_envs import native  # noqa: E402

NAMES = ["entry", "X tile in LDS, requests out", "L0 contraction", "L0 epilogue + barrier", "L1 contraction", "L1 epilogue + barrier",
         "L2 contraction", "L2 epilogue + barrier", "head outputs (A) + combine", "row math + barrier", "head dW (C) + dZ2 (B) + barrier",
         "dX through W2", "dZ1 epilogue + barrier", "dX through W1", "dZ0 epilogue + scalars"]


def main():
    torch.cuda.set_device(0)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    net = sys.argv[2] if len(sys.argv) > 2 else "ref"
    D, hidden = (45, (512, 256, 128)) if net == "ref" else (48, (256, 256, 256))
    A = 12
    nat = native.get(torch.device("cuda", 0))
    shape = native.shape_of(D, A, hidden)
    lay = native.layout_of(shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    flat = rn(lay.n_flat) * 0.05
    obs, act = rn(M, lay.obs_pad), rn(M, A)
    logp, adv, ret, val = rn(M) * 0.3 - 11.0, rn(M), rn(M), rn(M)
    inds = torch.randperm(M, device="cuda")
    grad, diag = torch.zeros(lay.n_flat, device="cuda"), torch.zeros(8, device="cuda")
    one, zero = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    hp = native.PpoHparams(0.2, 0.001, 2.0, 1, 1, 1.0 / M, 0)
    nat.mlp_reserve(shape, M)
    run = lambda: nat.ppo_minibatch_grad(shape, hp, flat, obs, act, logp, adv, ret, val, inds, zero, one, None, grad, diag)
    lib = nat.lib
    lib.catppo_debug_step16_tl.restype, lib.catppo_debug_step16_tl.argtypes = C.c_int, [C.c_void_p]
    buf = torch.zeros(2 * 1024 * 8 * 16, dtype=torch.int64, device="cuda")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    assert lib.catppo_debug_step16_tl(buf.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(2, 1024, 8, 16).astype(np.float64) * 0.01     # 100 MHz -> us
    n_t = min(-(-M // 16), 1024)
    t0 = raw[:, :n_t, :, 0][raw[:, :n_t, :, 0] > 0].min()
    for ni, nm in ((0, "critic"), (1, "actor")):
        t = raw[ni, :n_t, :, :15]                                  # [tile][wave][stamp]
        print("%s workgroups (%d): entry %.2f .. %.2f us after the first wave of the launch, exit median %.2f max %.2f"
              % (nm, n_t, t[:, :, 0].min() - t0, t[:, :, 0].max() - t0, np.median(t[:, :, 14]) - t0, t[:, :, 14].max() - t0))
        # per phase: duration seen by wave 0, and the spread of the waves' ARRIVAL at the end of the phase
        d = np.diff(t, axis=2)
        for i in range(14):
            arr = t[:, :, i + 1]
            skew = arr.max(axis=1) - arr.min(axis=1)
            print("   %-40s wave0 %6.2f us | fastest wave %6.2f  slowest wave %6.2f | arrival skew across the 8 waves %5.2f us"
                  % (NAMES[i + 1], np.median(d[:, 0, i]), np.median(d[:, :, i].min(axis=1)), np.median(d[:, :, i].max(axis=1)), np.median(skew)))
        print("   %-40s median %6.2f us" % ("workgroup total", np.median(t[:, :, 14].max(axis=1) - t[:, :, 0].min(axis=1))))
        w = t[n_t // 2]                                            # one workgroup, every wave: cumulative times
        print("   one workgroup (tile %d), us since its first wave started; rows = waves 0..7" % (n_t // 2))
        for wv in range(8):
            print("      " + " ".join("%6.2f" % (x - w[:, 0].min()) for x in w[wv]))


if __name__ == "__main__":
    main()
