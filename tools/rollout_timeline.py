"""Phase timeline of rollout_pre / rollout_post (wall clock stamps per workgroup) on the cfg2 rollout.

Needs a library whose rollout.hip was compiled with -DROLLOUT_TL, e.g. (from constraints-as-terminations_amd/, after
build.py):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
            -DROLLOUT_TL -c csrc/rollout.hip -o /tmp/rollout_tl.o
            hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/bin/libcatppo_tl.so /tmp/rollout_tl.o <the other build/*.o>
then on the GPU box:  CATPPO_LIB=$PWD/tools/bin/libcatppo_tl.so python tools/rollout_timeline.py [workload]
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
    nat = trainer.nat
    lib = nat.lib
    lib.catppo_debug_rollout_tl.restype = C.c_int
    lib.catppo_debug_rollout_tl.argtypes = [C.c_void_p]
    buf = torch.zeros(2 * 1024 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        trainer.run_iteration(log=False)
    torch.cuda.synchronize()
    assert lib.catppo_debug_rollout_tl(buf.data_ptr()) == 0
    trainer.run_iteration(log=False)          # the stamps of the LAST env step of this iteration survive
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(2, 1024, 16).astype(np.float64) * 0.01     # us
    pre, post = t[0], t[1]
    npre = int((pre[:, 0] > 0).sum())
    npost = int((post[:, 0] > 0).sum())
    pre, post = pre[:npre], post[:npost]
    t0 = pre[:, 0].min()
    def col(a, i):
        v = a[:, i]
        v = v[v > 0] - t0
        return "n=%d min %.1f p50 %.1f max %.1f" % (len(v), v.min(), np.median(v), v.max()) if len(v) else "-"
    print("rollout_pre: %d workgroups (us since its first workgroup started)" % npre)
    for i, name in enumerate(["entry", "ids staged", "tile done", "partials written", "L1 ticket resolved",
                              "L1 fold done (group-last)", "L2 ticket resolved", "end (last workgroup)"]):
        print("  %-28s %s" % (name, col(pre, i)))
    for i, name in [(9, "  tile: terms done"), (10, "  tile: counters done"),
                    (11, "  tile: obs moments done"), (12, "  tile: barrier passed")]:
        print("  %-28s %s" % (name, col(pre, i)))
    print("rollout_post: %d workgroups" % npost)
    for i, name in enumerate(["entry", "state derived", "tile done", "ticket resolved", "end (last workgroup)"]):
        print("  %-28s %s" % (name, col(post, i)))


if __name__ == "__main__":
    main()
