"""Phase times of several trainers built one after the other in ONE process (python tools/two_trainers_probe.py
cfg2,reference): found the workspace-regrowth slowdown fixed in catppo_reserve."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"constraints-as-terminations_amd")]
import bench
order = sys.argv[1].split(",")
for wl in order:
    env, trainer, _ = bench.build(wl, 42, 0, overrides={"graph_update": False})
    for _ in range(2): trainer.run_iteration(log=False)
    trainer.time_phases=True
    t0=time.perf_counter()
    for _ in range(5): trainer.run_iteration(log=False)
    t1=time.perf_counter()
    ph=trainer.phase_summary()
    print(wl, "rollout_us_per_step", round(1e3*ph["rollout_ms"]/trainer.T,1), "update_ms", round(ph["update_ms"],2), "host_ms_per_it", round(1e3*(t1-t0)/5,2), flush=True)
