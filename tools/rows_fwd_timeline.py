"""Phase timeline of rows_fwd_kernel<64> (training launch; wall-clock stamps of thread 0 of every workgroup).  Needs a
library built with -DFUSED_TL:
    python constraints-as-terminations_amd/build.py --variant fftl -DFUSED_TL
    CATPPO_LIB=$PWD/tools/bin/libcatppo_fftl.so python tools/rows_fwd_timeline.py [rows] [obs_dim] [n_layers]"""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    obs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    nl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    nat = native.get(torch.device("cuda", 0))
    shape = native.shape_of(obs, 12, (256, 256, 256))
    lay = native.layout_of(shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    flat = torch.randn(lay.n_flat, device="cuda", generator=g) * 0.05
    x = torch.randn(rows, lay.obs_pad, device="cuda", generator=g)
    nat.mlp_reserve(shape, rows)
    lib = nat.lib
    lib.catppo_debug_fused_tl.restype, lib.catppo_debug_fused_tl.argtypes = C.c_int, [C.c_void_p]
    lib.catppo_debug_rows_fwd.restype = C.c_int
    lib.catppo_debug_rows_fwd.argtypes = [C.c_void_p, C.POINTER(native.MlpShape), C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_void_p]
    run = lambda: nat._ok(lib.catppo_debug_rows_fwd(nat.h, C.byref(shape), flat.data_ptr(), x.data_ptr(), rows, nl,
                                                    nat._stream()))
    buf = torch.zeros(2 * 1024 * 16 + 2 * 1024 * 4, dtype=torch.int64, device="cuda")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    assert lib.catppo_debug_fused_tl(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * rows * 2 * (lay.obs_pad * 256 + (nl - 1) * 256 * 256)
    us = e0.elapsed_time(e1) * 1e3
    print("rows %d obs %d layers %d: launch %.1f us (event), %.1f TFLOP/s, matrix-pipe time at 2.4 GHz %.1f us"
          % (rows, obs, nl, us, fl / us / 1e6, fl / 157.3e6))
    allraw = buf.cpu().numpy()
    raw = allraw[:2 * 1024 * 16].reshape(2, 1024, 16).astype(np.float64)
    ck = allraw[2 * 1024 * 16:].reshape(2, 1024, 4).astype(np.float64)[0]
    ck = ck[ck[:, 0] > 0]
    dt_us, ticks = (ck[:, 3] - ck[:, 2]) * 0.01, ck[:, 1] - ck[:, 0]
    print("last layer of net 0, wave 0: %.2f us wall, %.0f shader ticks -> %.3f GHz; 256 MFMAs of this wave = %.1f ticks each"
          % (np.median(dt_us), np.median(ticks), np.median(ticks / dt_us) / 1e3, np.median(ticks) / 256))
    t = raw * 0.01                                                   # 100 MHz wall clock -> us
    wg = t[0][t[0][:, 0] > 0]
    t0 = wg[:, 0].min()
    names = {0: "entry"}
    for ni in range(2):
        names[1 + 8 * ni] = "net %d: x tile in LDS" % ni
        for l in range(nl):
            names[2 + 8 * ni + 2 * l] = "net %d layer %d: contraction done" % (ni, l)
            names[3 + 8 * ni + 2 * l] = "net %d layer %d: tile written, stores issued" % (ni, l)
    if nl == 2:
        names[6] = "   net 0 layer 1: every wave done (barrier 1)"
        names[7] = "   net 0 layer 1: tile written (barrier 2)"
        names[15] = "   net 0 layer 1: tile re-read for the stores"
        names[14] = "   net 0 layer 0: ring filled (begin)"
    print("%d workgroups (us since the first workgroup of the launch started)" % len(wg))
    prev = None
    order = sorted(names, key=lambda i: np.median(wg[:, i]))
    for i in order:
        v = wg[:, i]
        if not (v > 0).all():
            continue
        v = v - t0
        d = "" if prev is None else "   (+%.2f us median since the previous stamp)" % (np.median(v) - prev)
        print("  %-44s min %6.2f p50 %6.2f max %6.2f%s" % (names[i], v.min(), np.median(v), v.max(), d))
        prev = np.median(v)


if __name__ == "__main__":
    main()
