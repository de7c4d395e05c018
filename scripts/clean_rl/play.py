"""Load a CleanRL checkpoint into ``Agent`` and roll the policy out (reference: scripts/clean_rl/play.py).

Checkpoints written by ``PPO()`` (``model_<it>.pt``, the reference's 23-key ``state_dict``) load
here and in the reference's play.py alike.  Like the reference (play.py:107-135) the script writes
``<run>/exported/model.onnx`` and ``<run>/exported/model.pt`` (deterministic policy with the frozen
observation normaliser) before rolling the policy out on the device kernels.
"""
import argparse
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, os.path.join(ROOT, "constraints-as-terminations_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cli_args  # noqa: E402  isort: skip


def get_checkpoint_path(log_root: str, run_dir: str = ".*", checkpoint: str = "model_.*.pt") -> str:
    """latest run matching ``run_dir`` and, inside it, the highest-numbered file matching ``checkpoint``"""
    runs = sorted(d for d in os.listdir(log_root) if os.path.isdir(os.path.join(log_root, d)) and re.match(run_dir, d))
    if not runs:
        raise ValueError(f"no run matching '{run_dir}' in {log_root}")
    run = os.path.join(log_root, runs[-1])
    files = [f for f in os.listdir(run) if re.match(checkpoint, f)]
    if not files:
        raise ValueError(f"no checkpoint matching '{checkpoint}' in {run}")
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", f)] or [0])
    return os.path.join(run, files[-1])


def main(argv=None):
    parser = argparse.ArgumentParser(description="Play an RL agent trained with CleanRL.")
    parser.add_argument("--video", action="store_true", default=False)
    parser.add_argument("--video_length", type=int, default=200, help="Number of roll-out steps.")
    parser.add_argument("--num_envs", type=int, default=None)
    parser.add_argument("--task", type=str, default=None)
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument("--device", type=str, default=None)
    parser.add_argument("--headless", action="store_true", default=False)
    cli_args.add_clean_rl_args(parser)
    args_cli = parser.parse_args(argv)
    import torch

    import cat_envs.tasks  # noqa: F401
    from cat_envs.shim import load_cfg_from_registry, make
    from cat_envs.tasks.utils.cleanrl.ppo import Agent

    env_cfg = load_cfg_from_registry(args_cli.task, "env_cfg_entry_point")
    if args_cli.num_envs is not None:
        env_cfg.scene.num_envs = args_cli.num_envs
    if args_cli.device is not None:
        env_cfg.sim.device = args_cli.device
    agent_cfg = cli_args.parse_clean_rl_cfg(args_cli.task, args_cli)
    log_root_path = os.path.abspath(os.path.join("logs", "clean_rl", agent_cfg.experiment_name))
    print(f"[INFO] Loading experiment from directory: {log_root_path}")
    resume_path = get_checkpoint_path(log_root_path, agent_cfg.load_run, agent_cfg.load_checkpoint)
    print(f"[INFO] Loading model: {resume_path}")

    env = make(args_cli.task, cfg=env_cfg)
    actor = Agent(env, hidden=tuple(agent_cfg.hidden), mlp_precision=str(getattr(agent_cfg, "mlp_precision", "fp32")))
    actor.load_state_dict(torch.load(resume_path, map_location=actor.flat.device))
    actor.eval()

    from cat_envs.tasks.utils.cleanrl.export import export_policy
    exported = export_policy(actor.state_dict(), os.path.join(os.path.dirname(resume_path), "exported"))
    print(f"[INFO] Exported ONNX model to {exported['onnx']}")
    print(f"[INFO] Exported .pt model to {exported['jit']}")

    obs = env.reset()[0]["policy"]
    ret = 0.0
    for _ in range(args_cli.video_length):
        with torch.no_grad():
            actions, _, _, _ = actor.get_action_and_value(actor.obs_rms(obs, update=False))
        obs, rewards, dones, timeouts, info = env.step(actions)
        obs = obs["policy"]
        ret += float(rewards.mean())
    print(f"[INFO] mean reward per step over {args_cli.video_length} steps: {ret / args_cli.video_length:.4f}")
    env.close()


if __name__ == "__main__":
    main()
