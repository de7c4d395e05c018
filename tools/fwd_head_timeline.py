"""Step timeline of fwd_head_kernel (wall-clock stamps per workgroup) inside a real update phase.

Needs a library whose mlp.hip was compiled with -DFWD_HEAD_TL (same recipe as tools/rollout_timeline.py with
csrc/mlp.hip / -DFWD_HEAD_TL); on the GPU box:
    CATPPO_LIB=$PWD/tools/bin/libcatppo_fhtl.so python tools/fwd_head_timeline.py [workload]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    env, trainer, agent_cfg = bench.build(wl, 1, 0, "fp32", 1, 0, {})
    lib = trainer.nat.lib
    lib.catppo_debug_fwd_head_tl.restype = C.c_int
    lib.catppo_debug_fwd_head_tl.argtypes = [C.c_void_p]
    buf = torch.zeros(2 * 1024 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        trainer.run_iteration(log=False)
    torch.cuda.synchronize()
    assert lib.catppo_debug_fwd_head_tl(buf.data_ptr()) == 0
    trainer.run_iteration(log=False)          # the stamps of the LAST minibatch survive
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(2, 1024, 8).astype(np.float64) * 0.01     # us
    names = ["entry", "main loop + H tile done", "A (head outputs) done", "row math done", "C (dWh) done",
             "B (dZ in LDS) done", "rows streamed out", "end"]
    for net, nm in ((0, "critic"), (1, "actor")):
        a = t[net]
        a = a[a[:, 0] > 0]
        t0 = t[:, :, 0][t[:, :, 0] > 0].min()
        print("%s workgroups: %d (us since the first workgroup of the launch started)" % (nm, len(a)))
        for i, name in enumerate(names):
            v = a[:, i] - t0
            print("  %-26s min %.1f p50 %.1f max %.1f" % (name, v.min(), np.median(v), v.max()))


if __name__ == "__main__":
    main()
