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
    #: element type of the (T,N) rollout planes rewards / values / dones / true_dones / advantages / returns:
    #: "fp32" (reference) or "fp16" (BASELINE config 5: halves the GAE traffic; the recurrence still runs in fp32)
    rollout_dtype: str = "fp32"
    #: where the action noise (ppo.py:111) and the minibatch permutation (ppo.py:295) come from: "device" =
    #: counter-based Philox4x32-10 / keyed Feistel permutation inside the kernels (no extra launches, no index
    #: array); "torch" = torch.randn / torch.randperm like the reference
    rng: str = "device"
    #: learning-rate schedule: None = the reference's ``anneal_lr`` switch (linear anneal or fixed), "adaptive" =
    #: KL-adaptive (skrl KLAdaptiveLR / rl_games lr_schedule: adaptive), updated on the device after every epoch
    lr_schedule: object = None
    kl_threshold: float = 0.01
    #: "serial" = one lane per env (bit-exact with the reference loop), "scan" = wavefront-shuffle scan over time
    gae_mode: str = "serial"
    #: fuse the post-simulator part of an env step (terms, CaT, resets, buffer rows, obs normaliser) into two
    #: launches when the env offers ``step_into`` (CaTEnv does)
    fused_rollout: bool = True
    #: replay the update phase of an iteration from a hipGraph (needs rng="device"); None = automatic: on for
    #: minibatches <= 4096 samples, where the optimiser step is launch bound
    graph_update: object = None
