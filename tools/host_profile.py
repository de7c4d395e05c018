"""cProfile of the host side of PPOTrainer.rollout() (where the Python/ctypes time of an env step goes)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "constraints-as-terminations_amd")]
import bench  # noqa: E402

torch.cuda.set_device(0)
env, trainer, agent_cfg = bench.build("cfg2", 42, 0)
for _ in range(3):
    trainer.run_iteration(log=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    trainer.run_iteration(log=False)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
