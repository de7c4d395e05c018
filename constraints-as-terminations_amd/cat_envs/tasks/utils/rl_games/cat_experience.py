"""Float-dones experience buffer (reference: rl_games/cat_experience.py:20-33 overrides the uint8 ``dones``
plane of rl_games' ``ExperienceBuffer`` with fp32).  Time-major (horizon, num_actors, ...) planes, the layout
the GAE kernel scans."""
from __future__ import annotations

from typing import Dict, Sequence

import torch


class CaTExperienceBuffer:
    def __init__(self, num_actors: int, horizon_length: int, obs_shape: Sequence[int], actions_num: int,
                 device="cuda"):
        T, N = int(horizon_length), int(num_actors)
        dev = torch.device(device)
        z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
        self.horizon_length, self.num_actors = T, N
        self.tensor_dict: Dict[str, torch.Tensor] = {
            "obses": z(T, N, *obs_shape), "rewards": z(T, N, 1), "values": z(T, N, 1), "neglogpacs": z(T, N),
            "dones": z(T, N),                     # fp32, not uint8: the CaT termination probability
            "actions": z(T, N, actions_num), "mus": z(T, N, actions_num), "sigmas": z(T, N, actions_num),
        }

    def update_data(self, name: str, index: int, val: torch.Tensor):
        self.tensor_dict[name][index].copy_(val.reshape(self.tensor_dict[name][index].shape))

    def get_transformed_list(self, transform_op, tensor_list):
        return {k: transform_op(self.tensor_dict[k]) for k in tensor_list if k in self.tensor_dict}
