"""Wall time of the phases of one CaT-PPO iteration (cfg2 shapes): host enqueue time (Python + ctypes, before the
sync) and device-complete time.  python tools/phase_times.py [--workload cfg2|reference] [--iters 10]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")]
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if os.environ.get("CATPPO_FORCE_DIST") == "1":        # every exchange point active, RCCL world of size 1
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29534"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", 0))
    env, trainer, agent_cfg = bench.build(a.workload, 42, 0)
    for _ in range(3):
        trainer.run_iteration(log=False)
    torch.cuda.synchronize()
    acc = {}
    for _ in range(a.iters):
        for name, fn in (("rollout", trainer.rollout), ("compute_returns", trainer.compute_returns),
                         ("update", trainer.update)):
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            e = acc.setdefault(name, [0.0, 0.0])
            e[0] += t1 - t0
            e[1] += t2 - t0
        trainer.obs[0].copy_(trainer.obs[trainer.T])
        trainer.dones[0].copy_(trainer.dones[trainer.T])
        trainer.true_dones[0].copy_(trainer.true_dones[trainer.T])
    print(json.dumps({k: {"host_enqueue_ms": 1e3 * v[0] / a.iters, "device_done_ms": 1e3 * v[1] / a.iters}
                      for k, v in acc.items()}, indent=1))


if __name__ == "__main__":
    main()
