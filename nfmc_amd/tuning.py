"""Step-size dual averaging and the refit-buffer split (nfmc/algorithms/sampling/tuning.py)."""
import math
from dataclasses import dataclass

import torch


@dataclass
class DualAveragingParams:
    target_acceptance_rate: float = 0.651
    kappa: float = 0.75
    gamma: float = 0.05
    t0: int = 10


class DualAveraging:
    """tuning.py:15-41: log-step dual averaging driven by (target - observed) acceptance."""

    def __init__(self, initial_step_size, params: DualAveragingParams):
        self.t = params.t0
        self.error_sum = 0.0
        self.log_step_averaged = math.log(initial_step_size)
        self.log_step = math.inf
        self.mu = math.log(10 * initial_step_size)
        self.p = params

    def step(self, acceptance_rate_error):
        self.error_sum += float(acceptance_rate_error)
        self.log_step = self.mu - self.error_sum / (math.sqrt(self.t) * self.p.gamma)
        eta = self.t ** -self.p.kappa
        self.log_step_averaged = eta * self.log_step + (1 - eta) * self.log_step_averaged
        self.t += 1

    @property
    def value(self):
        return math.exp(self.log_step_averaged)

    def __repr__(self):
        return f'DA error: {self.error_sum:.2f}'


def train_val_split(x: torch.Tensor, train_pct: float, max_train_size: int, max_val_size: int, shuffle: bool = True,
                    shard=None):
    """tuning.py:44-65.  x: (n_iterations, n_chains, *event).  With `shard` (chains split over GPUs) every
    rank contributes an equal share of the capped buffer and the shares are all-gathered (collective C1), so
    each rank fits the flow on the same rows."""
    rows = x.flatten(0, 1)
    if shard is not None and shard.world > 1:
        per_rank = -(-(max_train_size + max_val_size) // shard.world)
        if shuffle:
            rows = rows[torch.randperm(len(rows), device=rows.device)]
        rows = shard.all_gather_rows(rows[:per_rank].contiguous())
        shuffle = False  # already shuffled per rank; gathered order is deterministic
    if shuffle:
        rows = rows[torch.randperm(len(rows), device=rows.device)]
    n_train = int(train_pct * len(rows))
    x_train, x_val = rows[:n_train], rows[n_train:]
    return x_train[:max_train_size], x_val[:max_val_size]
