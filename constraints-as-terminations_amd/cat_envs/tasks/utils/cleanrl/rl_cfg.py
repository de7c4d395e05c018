"""Hyper-parameter surface of the CleanRL PPO runner (reference: cleanrl/rl_cfg.py:12-38)."""
from typing import Literal

from cat_envs.shim import MISSING, configclass


@configclass
class CleanRlPpoActorCriticCfg:
    seed: int = 42

    save_interval: int = MISSING

    learning_rate: float = MISSING
    num_steps: int = MISSING
    num_iterations: int = MISSING
    gamma: float = MISSING
    gae_lambda: float = MISSING
    updates_epochs: int = MISSING
    minibatch_size: int = MISSING
    clip_coef: float = MISSING
    ent_coef: float = MISSING
    vf_coef: float = MISSING
    max_grad_norm: float = MISSING
    norm_adv: bool = MISSING
    clip_vloss: bool = MISSING
    anneal_lr: bool = MISSING

    experiment_name: str = MISSING
    logger: Literal["tensorboard", "wandb"] = "tensorboard"
    wandb_project: str = MISSING

    load_run: str = MISSING
    load_checkpoint: str = MISSING

    # --- extensions of this implementation (defaults reproduce the reference) -----------------
    #: hidden widths of both MLPs (reference Agent hard-codes 512/256/128, ppo.py:78-95)
    hidden: tuple = (512, 256, 128)
    #: env-sharded runs: all-reduce the cross-env statistics (CaT column max, normaliser moments,
    #: minibatch advantage mean/std) so that N ranks reproduce one process on the union of shards
    dist_exact: bool = True
    #: "fp32" = the reference's numerics (fp32-input MFMA).  "bf16" = hidden-layer GEMM operands rounded to bf16,
    #: fp32 accumulation, fp32 master weights / activations / optimiser (BASELINE config 5); not a parity mode.
    mlp_precision: str = "fp32"
