import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "constraints-as-terminations_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import bench
from cat_envs import native
nat = native.Native()
for T, N in [(24,4096),(24,1<<18),(24,1<<20),(48,1<<22)]:
    r = bench.gae_roofline(nat, T, N, 20)
    print(T, N, round(r["us"],1), round(r["GBps"]), round(r["frac"],3))
