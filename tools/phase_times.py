"""Device time of the phases of one CaT-PPO iteration (HIP events on the trainer's stream) and the host time the
iteration takes to enqueue.  python tools/phase_times.py [--workload cfg2|reference|cfg3_shard|...] [--iters 10]"""
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
    trainer.time_phases = True
    t0 = time.perf_counter()
    for _ in range(a.iters):
        trainer.run_iteration(log=False)
    t1 = time.perf_counter()
    out = trainer.phase_summary()
    t2 = time.perf_counter()
    out["host_enqueue_ms_per_iteration"] = 1e3 * (t1 - t0) / a.iters
    out["wall_ms_per_iteration"] = 1e3 * (t2 - t0) / a.iters
    out["fused_rollout"] = trainer.sink is not None
    out["graph_update"] = trainer.graph_update
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
