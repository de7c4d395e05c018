"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher around it must run N
ranks and print ONE line with n_gpus == N; a launcher that started a different number of ranks is refused.

Reference precedent for a self-contained distributed entry point: scripts/skrl/train.py:28-30,116-117,
scripts/rl_games/train.py:23-25,100-107 (`--distributed` flags, no wrapper script needed)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT", "CATPPO_FORCE_DIST")}
    env.update(kw)
    return env


def test_print_launch_is_the_drivers_command_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1", "--print-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    line = r.stdout.strip().splitlines()[-1]
    for tok in ("-m torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr 127.0.0.1",
                "bench.py --gpus 4 --steps 3 --warmup 1"):
        assert tok in line, (tok, line)
    assert "--print-launch" not in line


def test_scaling_strong_is_the_cfg3_workload():
    """`--scaling strong` = BASELINE configs[2] as written (`--workload cfg3`); a contradiction is refused"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--scaling", "strong", "--print-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0 and "--scaling strong" in r.stdout, r.stderr
    r = subprocess.run([sys.executable, BENCH, "--scaling", "strong", "--workload", "cfg2"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 2 and "contradicts" in r.stderr


def test_world_size_that_differs_from_gpus_is_refused():
    """no JSON line whose n_gpus differs from --gpus: a launcher that started 2 ranks for `--gpus 1` gets exit code 2"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2
    assert r.stdout.strip() == "" and "refusing to run" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_itself(tmp_path):
    """the launcher proven on whatever box this runs on: with fewer than two GPUs the two ranks share cuda:0 and exchange
    through gloo with host staging (the transport of test_gpu_two_rank_trainer.py); with two or more they use RCCL"""
    import torch
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=_env(), timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["self_launched"] is True
    assert d["config"]["envs_total"] == 2 * d["config"]["envs_per_gpu"] and d["scaling"] == "weak"
    # round 5: 24 env-step all-gathers + 30 gradient all-reduces + 3 small ones per iteration (value-normaliser moments of
    # values and returns together, the advantage moments of all epochs together, the diagnostics); 61-62 before
    assert d["comm_ms_per_iteration"] > 0 and 24 + 30 <= d["comm"]["collectives_per_iteration"] <= 24 + 30 + 3
    # per-rank times (stragglers), what the communicator set-up saw, why native RCCL is / is not carrying the data path
    pr = d["per_rank_ms"]
    assert len(pr["by_rank"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-6 * pr["max"]
    ce = d["config"]["comm_env"]
    assert ce["rendezvous_backend"] == "gloo" and ce["launch_attempt"] == 1 and "HSA_ENABLE_IPC_MODE_LEGACY" in ce
    assert ce["ipc_mode_set_by"] in ("environment", "bench.py default")
    assert d["config"]["scaling_mode"] == "weak" and "native_comm_error" in d["config"]
    if torch.cuda.device_count() < 2:
        assert d["ranks_share_gpus"] is True and d["physical_gpus"] == 1
        assert "gloo" in d["config"]["collectives"] and d["config"]["rccl_world"] == 0
    else:
        assert d["ranks_share_gpus"] is False and d["config"]["rccl_world"] == 2
        assert d["config"]["graph_fallback"] is None or isinstance(d["config"]["graph_fallback"], str)
    assert d["value"] > 0 and abs(d["value"] - d["config"]["envs_total"] * 24 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-3 * d["value"]
    # round 6: the weak line carries BASELINE configs[2] as a `strong` record of the same process group
    st = d["strong"]
    assert st["scaling"] == "strong" and st["workload"].startswith("cfg3") and st["envs_total"] == 16384
    assert st["envs_per_gpu"] == 8192 and st["minibatch_per_gpu"] == 8192 and st["global_minibatch"] == 16384
    assert st["value"] > 0 and abs(st["value"] - 16384 * 24 * st["steps"] / (st["ms_per_step"] * st["steps"] * 1e-3)) < 1e-3 * st["value"]
    assert len(st["per_rank_ms"]["by_rank"]) == 2 and st["comm_ms_per_iteration"] > 0
    assert st["collectives_per_iteration"] >= 24 + 5 * 24 and 0 < st["roofline"]["frac"] < 1
    assert st["rccl_world"] == d["config"]["rccl_world"]
