"""Learning-rate schedule of the position parameters -- the mirror of utils/general_utils.py:39-74
(`get_expon_lr_func`), used by `training_setup` / `update_learning_rate` (scene/mesh_gaussian_model.py:376-379,
scene/gaussian_model.py:171-177): log-linear interpolation from lr_init to lr_final over max_steps, optionally eased
in over the first lr_delay_steps by a sine ramp that starts at lr_delay_mult."""
from __future__ import annotations

import math
from typing import Callable


def get_expon_lr_func(lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
                      max_steps: int = 1000000) -> Callable[[int], float]:
    def rate(step: int) -> float:
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0                                      # parameter switched off
        ease = 1.0
        if lr_delay_steps > 0:
            frac = min(max(step / lr_delay_steps, 0.0), 1.0)
            ease = lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * frac)
        t = min(max(step / max_steps, 0.0), 1.0)
        return ease * math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)

    return rate
