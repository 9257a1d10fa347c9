"""Warmup helpers next to the hot path: the step-size controller and the refit-buffer split
(what nfmc/algorithms/sampling/tuning.py provides to the samplers)."""
import math
from dataclasses import dataclass

import torch


@dataclass
class DualAveragingParams:
    """Constants of the controller; the defaults are the reference's (tuning.py:8-12)."""
    target_acceptance_rate: float = 0.651
    kappa: float = 0.75    # decay exponent of the averaging weight t^-kappa
    gamma: float = 0.05    # shrinkage towards the anchor log(10 h0)
    t0: int = 10           # iteration count the schedule starts from


class DualAveraging:
    """Nesterov dual averaging of log(step size), driven by `step(target_rate - observed_rate)`.

    State: the running sum E of the errors fed so far and the iteration count t (starting at t0).
      raw      log h_t   = log(10 h0) - E / (gamma sqrt(t))
      smoothed log hbar  <- w log h_t + (1 - w) log hbar,   w = t^-kappa
    `value` is exp(log hbar), the step size the sampler uses next (same recurrences, constants and update order
    as tuning.py:15-41: `test_mala_warmup_tuning_golden` replays the reference's tuned step sizes through it)."""

    def __init__(self, initial_step_size, params: DualAveragingParams):
        self.params = params
        self.iteration = int(params.t0)
        self.error_sum = 0.0
        self.anchor = math.log(10 * initial_step_size)
        self.log_raw = math.inf                          # no raw estimate before the first error
        self.log_smooth = math.log(initial_step_size)

    def step(self, acceptance_rate_error):
        p, t = self.params, self.iteration
        self.error_sum += float(acceptance_rate_error)
        self.log_raw = self.anchor - self.error_sum / (math.sqrt(t) * p.gamma)
        weight = t ** -p.kappa
        self.log_smooth = weight * self.log_raw + (1 - weight) * self.log_smooth
        self.iteration = t + 1

    @property
    def value(self) -> float:
        return math.exp(self.log_smooth)

    def __repr__(self):
        return 'DualAveraging(step=%.4g, error_sum=%.2f, t=%d)' % (self.value, self.error_sum, self.iteration)


def _shuffled(rows: torch.Tensor) -> torch.Tensor:
    return rows[torch.randperm(rows.shape[0], device=rows.device)]


def _sampled_rows(rows: torch.Tensor, m: int) -> torch.Tensor:
    """The first m rows of a shuffle of `rows`, without materialising the shuffled buffer: on the GPU one launch of
    nfmc_rows_sample_f32 (rows pi(0 .. m - 1) of a keyed pseudo-random permutation, key from torch's CPU generator)."""
    total = rows.shape[0]
    if rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous() and 0 < m <= total:
        from . import hip
        flat = rows.reshape(total, -1)
        out = torch.empty(m, flat.shape[1], dtype=torch.float32, device=rows.device)
        seed = int(torch.randint(0, 2 ** 62, ()).item())
        hip.check(hip.lib().nfmc_rows_sample_f32(hip.ptr(flat), total, flat.shape[1], seed, 0, hip.ptr(out), m, None, hip.stream()),
                  'nfmc_rows_sample_f32')
        return out.reshape(m, *rows.shape[1:])
    return rows[torch.randperm(total, device=rows.device)[:m]].contiguous()


def train_val_split(x: torch.Tensor, train_pct: float, max_train_size: int, max_val_size: int, shuffle: bool = True,
                    shard=None):
    """(n_iterations, n_chains, *event) -> (x_train, x_val): all (step, chain) rows pooled, shuffled, cut at
    `train_pct` and capped (tuning.py:44-65).

    With `shard` (chains split over GPUs; collective C1 of SURVEY 8e) every rank contributes the same number of rows
    of the capped buffer, drawn from its own shuffled rows, and the shares are all-gathered.  The row count is agreed
    first (all-reduce MIN of what every rank can give: block sizes differ by one when the chain count does not divide,
    and a time limit can leave ranks with different numbers of kept steps), so the collective always sees equal
    sizes.  The gathered rows are then shuffled AGAIN with a seed broadcast from rank 0 before the `train_pct` cut:
    train and validation both mix rows of every rank (a cut in rank order would validate on the last ranks' chains
    only), and every rank fits the flow on the same rows in the same order."""
    rows = x.flatten(0, 1)
    if shard is not None and shard.world > 1:
        share = -(-(max_train_size + max_val_size) // shard.world)
        share = shard.all_reduce_min_int(min(share, rows.shape[0]))
        if share <= 0:
            raise ValueError('train_val_split: a rank has no rows to contribute to the refit buffer')
        if shuffle:
            local = _sampled_rows(rows, share)      # this rank's share: the first rows of a shuffle of its own rows
        else:
            local = rows[:share].contiguous()
        rows = shard.all_gather_rows(local)
        if shuffle:
            seed = shard.broadcast_int(int(torch.randint(0, 2 ** 62, ()).item()))
            perm = torch.randperm(rows.shape[0], generator=torch.Generator().manual_seed(seed))
            rows = rows[perm.to(rows.device)]
    elif shuffle:
        # the rows of the shuffled buffer that survive the cut and the caps, without materialising the shuffled buffer (at
        # the C5 shape it is 160 MiB per refit for 8192 kept rows)
        total = rows.shape[0]
        cut = int(train_pct * total)
        n_train, n_val = min(cut, int(max_train_size)), min(total - cut, int(max_val_size))
        if rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous() and total > 0 and n_train + n_val > 0:
            # ONE launch (csrc/fit_support.hip: nfmc_rows_sample_f32): rows pi(0 .. n_train + n_val - 1) of a keyed
            # pseudo-random permutation pi of the pooled rows -- positions [0, n_train) and [cut, cut + n_val) of a uniform
            # shuffle are, in distribution, any n_train + n_val distinct positions of it.  The key comes from torch's CPU
            # generator, like the reference's randperm (tuning.py:58-59); no sort of all rows, no index tensors.
            out = _sampled_rows(rows, n_train + n_val)
            return out[:n_train], out[n_train:]
        perm = torch.randperm(total, device=rows.device)
        return rows[perm[:cut][:max_train_size]], rows[perm[cut:][:max_val_size]]
    cut = int(train_pct * rows.shape[0])
    return rows[:cut][:max_train_size], rows[cut:][:max_val_size]
