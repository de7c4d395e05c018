"""Phase timeline of fused_fwd_kernel (wall-clock stamps per workgroup).  Needs a library built with -DFUSED_TL:
    python constraints-as-terminations_amd/build.py --variant fftl -DFUSED_TL
    CATPPO_LIB=$PWD/tools/bin/libcatppo_fftl.so python tools/fused_fwd_timeline.py [rows] [hidden...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")):
    sys.path.insert(0, p)
from cat_envs import native  # noqa: E402


def main():
    torch.cuda.set_device(0)
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    hidden = tuple(int(x) for x in sys.argv[2:]) or (256, 256, 256)
    nat = native.get(torch.device("cuda", 0))
    shape = native.shape_of(48, 12, hidden)
    lay = native.layout_of(shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    flat = torch.randn(lay.n_flat, device="cuda", generator=g) * 0.05
    x = torch.randn(rows, lay.obs_pad, device="cuda", generator=g)
    eps = torch.randn(rows, 12, device="cuda", generator=g)
    act, lp, val = torch.empty(rows, 12, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    nat.mlp_reserve(shape, rows)
    lib = nat.lib
    lib.catppo_debug_fused_tl.restype, lib.catppo_debug_fused_tl.argtypes = C.c_int, [C.c_void_p]
    buf = torch.zeros(2 * 1024 * 16, dtype=torch.int64, device="cuda")
    for _ in range(5):
        nat.policy_act(shape, flat, x, rows, eps, act, lp, val)
    torch.cuda.synchronize()
    assert lib.catppo_debug_fused_tl(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.policy_act(shape, flat, x, rows, eps, act, lp, val)
    e1.record()
    torch.cuda.synchronize()
    print("rows %d hidden %s: launch %.1f us (event)" % (rows, hidden, e0.elapsed_time(e1) * 1e3))
    t = buf.cpu().numpy().reshape(2, 1024, 16).astype(np.float64) * 0.01     # 100 MHz wall clock -> us
    names = ["entry", "x tile in LDS"] + ["layer %d done" % l for l in range(len(hidden))]
    idx = [0, 1] + [2 + l for l in range(len(hidden))] + [8]
    names.append("head done")
    raw = buf.cpu().numpy().reshape(2, 1024, 16).astype(np.float64)
    ok = raw[0, :, 2] > 0
    dw = (raw[0, ok, 3] - raw[0, ok, 2]) * 0.01          # us (100 MHz wall clock) spent in layer 1
    dc = raw[0, ok, 11] - raw[0, ok, 10]                  # shader-clock ticks (s_memtime) over the same span
    print("layer 1: %.2f us wall, %.0f shader ticks -> %.2f GHz effective" % (np.median(dw), np.median(dc), np.median(dc / dw) / 1e3))
    t0 = t[:, :, 0][t[:, :, 0] > 0].min()
    for net, nm in ((0, "critic"), (1, "actor")):
        a = t[net]
        a = a[a[:, 0] > 0]
        print("%s workgroups: %d (us since the first workgroup of the launch started)" % (nm, len(a)))
        for i, name in zip(idx, names):
            v = a[:, i] - t0
            print("  %-16s min %6.1f p50 %6.1f max %6.1f" % (name, v.min(), np.median(v), v.max()))


if __name__ == "__main__":
    main()
