"""Train an RL agent with CleanRL PPO on a CaT task (entry point of the hot path).

Same command line as the reference (scripts/clean_rl/train.py):

    python scripts/clean_rl/train.py --task=Isaac-Velocity-CaT-Flat-Solo12-v0 --headless

Isaac Sim cannot run on an AMD box, so the task registry and config classes of ``cat_envs.shim``
stand in for gymnasium / hydra and the env is driven by the device-resident synthetic Solo12 stream;
only AppLauncher's argument definitions are picked up when IsaacLab happens to be importable.
``key=value`` arguments (``env.scene.num_envs=1024 agent.minibatch_size=8192``) are the reference's hydra overrides:
like there (train.py:57-61,92) they are moved to ``sys.argv`` after argparse and applied by the ``hydra_task_config``
decorator BEFORE ``main``'s body, so the explicit flags (--num_envs, --seed, --num_iterations, --device) win over them.
``cat_envs.shim.hydra_task_config`` resolves them without hydra; IsaacLab's own decorator is used when it is importable.
"""
import argparse
import os
import pickle
import sys
from datetime import datetime

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, os.path.join(ROOT, "constraints-as-terminations_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cli_args  # noqa: E402  isort: skip


def build_parser():
    parser = argparse.ArgumentParser(description="Train an RL agent with CleanRL.")
    parser.add_argument("--video", action="store_true", default=False, help="Record videos during training.")
    parser.add_argument("--video_length", type=int, default=200, help="Length of the recorded video (in steps).")
    parser.add_argument("--video_interval", type=int, default=2000, help="Interval between video recordings (in steps).")
    parser.add_argument("--num_envs", type=int, default=None, help="Number of environments to simulate.")
    parser.add_argument("--task", type=str, default=None, help="Name of the task.")
    parser.add_argument("--seed", type=int, default=None, help="Seed used for the environment")
    parser.add_argument("--num_iterations", type=int, default=None, help="RL Policy training iterations.")
    cli_args.add_clean_rl_args(parser)
    try:  # AppLauncher contributes --headless / --device / ... when Isaac Sim exists
        from isaaclab.app import AppLauncher
        AppLauncher.add_app_launcher_args(parser)
    except ImportError:
        parser.add_argument("--headless", action="store_true", default=False, help="(no-op without Isaac Sim)")
        parser.add_argument("--device", type=str, default=None, help="Device, e.g. cuda:0")
    return parser


def apply_overrides(cfgs: dict, overrides):
    """hydra-style dotted overrides (kept as a module-level name for callers of round 1/2)"""
    from cat_envs.shim import apply_overrides as _apply
    _apply(cfgs, overrides)


def dump_cfg(path, cfg):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if path.endswith(".yaml"):
        import yaml
        with open(path, "w") as f:
            yaml.safe_dump(cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg), f, default_flow_style=False)
    else:
        with open(path, "wb") as f:
            pickle.dump(cfg.to_dict() if hasattr(cfg, "to_dict") else cfg, f)


def main(argv=None):
    args_cli, hydra_args = build_parser().parse_known_args(argv)
    if args_cli.video:
        args_cli.enable_cameras = True
    # clear out sys.argv for the override resolver, like the reference does for Hydra (train.py:57-58); an in-process
    # caller (tests, notebooks) gets its own argv back when main returns
    argv_before = sys.argv
    sys.argv = [sys.argv[0]] + hydra_args
    try:
        return _main(args_cli)
    finally:
        sys.argv = argv_before


def _main(args_cli):
    import torch

    import cat_envs.tasks  # noqa: F401  registers the tasks
    from cat_envs.shim import make
    from cat_envs.tasks.utils.cleanrl.ppo import PPO
    try:
        from isaaclab_tasks.utils.hydra import hydra_task_config
    except ImportError:
        from cat_envs.shim import hydra_task_config

    if torch.distributed.is_available() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # dmabuf IPC for RCCL on this platform; must be in the environment before the first HIP call of the process
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # rendezvous on gloo (CPU): the process's ONLY RCCL communicator is libcatppo's own, created by PPOTrainer
        # (cat_envs.parallel.init_native_comm); cf. scripts/skrl/train.py:116-117 for the reference's distributed set-up
        from cat_envs import parallel
        parallel.init_rendezvous(local)
        if args_cli.device is None:
            args_cli.device = f"cuda:{local}"

    @hydra_task_config(args_cli.task, "clean_rl_cfg_entry_point")
    def run(env_cfg, agent_cfg):
        # override configurations with the non-hydra CLI arguments (reference train.py:98-107)
        agent_cfg = cli_args.update_clean_rl_cfg(agent_cfg, args_cli)
        env_cfg.scene.num_envs = args_cli.num_envs if args_cli.num_envs is not None else env_cfg.scene.num_envs
        agent_cfg.num_iterations = (args_cli.num_iterations if args_cli.num_iterations is not None
                                    else agent_cfg.num_iterations)
        # env-sharded runs (torchrun): every rank owns its own block of --num_envs environments with its own stream
        # (seed + rank, like the reference's distributed front-ends: scripts/rl_games/train.py:100-107), so the
        # all-reduces combine distinct shards; rank 0 alone writes the run directory
        rank = int(os.environ.get("RANK", "0")) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0
        env_cfg.seed = agent_cfg.seed + rank
        env_cfg.sim.device = args_cli.device if args_cli.device is not None else env_cfg.sim.device

        log_root_path = os.path.abspath(os.path.join("logs", "clean_rl", agent_cfg.experiment_name))
        print(f"[INFO] Logging experiment in directory: {log_root_path}")
        log_dir = os.path.join(log_root_path, datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
        if rank == 0:
            dump_cfg(os.path.join(log_dir, "params", "env.yaml"), env_cfg)
            dump_cfg(os.path.join(log_dir, "params", "agent.yaml"), agent_cfg)
            dump_cfg(os.path.join(log_dir, "params", "env.pkl"), env_cfg)
            dump_cfg(os.path.join(log_dir, "params", "agent.pkl"), agent_cfg)

        env = make(args_cli.task, cfg=env_cfg, render_mode="rgb_array" if args_cli.video else None)
        if args_cli.video:
            print("[WARN] video recording needs Isaac Sim rendering; ignored with the synthetic simulator")
        PPO(env, agent_cfg, log_dir)
        env.close()

    run()


if __name__ == "__main__":
    main()
