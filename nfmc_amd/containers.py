"""Kernel / parameter dataclasses and the result containers of the path, name-for-name with
nfmc/algorithms/sampling/base.py, re-designed for a device-resident run:

* `MCMCStatistics` keeps fp64 per-coordinate sums and integer counters that the HIP kernels accumulate
  on the GPU (hip.DeviceStats); `running_first_moment`, `running_second_moment`, `acceptance_rate`
  are derived on access.  The streaming formula of `MCMCExpectation.update` (base.py:88-95) equals the
  plain mean over everything seen, which is what sums / n_seen gives.
* `MCMCSamples` stores kept states in one pre-sized device tensor written by the kernels, with the
  thinning / max_samples / last_sample semantics of base.py:234-263; `.as_tensor()` hands back a CPU
  tensor like the reference (`as_device_tensor()` avoids the copy).
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Union

import torch


@dataclass
class MCMCKernel:
    def __repr__(self):
        raise NotImplementedError

    def __post_init__(self):
        pass


@dataclass
class NFMCKernel(MCMCKernel):
    event_shape: Union[Tuple[int, ...], torch.Size]
    flow: Any = None

    def __post_init__(self):
        super().__post_init__()
        if self.flow is None:  # base.py:24-26
            from .flows import Flow, RealNVP
            self.flow = Flow(RealNVP(self.event_shape))

    def __repr__(self):
        return f'NFMCKernel(event_shape={tuple(self.event_shape)})'


@dataclass
class MCMCParameters:
    n_iterations: int = 100
    n_warmup_iterations: int = 100
    tuning: bool = False
    store_samples: bool = True

    def __post_init__(self):
        pass

    def tuning_mode(self):
        self.tuning = True

    def sampling_mode(self):
        self.tuning = False


@dataclass
class NFMCParameters(MCMCParameters):
    train_pct: float = 0.7
    max_train_size: int = 4096
    max_val_size: int = 4096
    flow_fit_kwargs: Dict[str, Any] = None

    def __post_init__(self):
        super().__post_init__()
        if self.flow_fit_kwargs is None:
            self.flow_fit_kwargs = {
                'early_stopping': True,
                'early_stopping_threshold': 50,
                'batch_size': 'adaptive',
                'show_progress': False
            }


class MCMCExpectation:
    """E[f(x)] over everything seen, from a running fp64 sum (f = identity or square)."""

    def __init__(self, event_shape, power: int):
        self.event_shape = tuple(event_shape)
        self.power = power
        self.n_seen = 0
        self.total = torch.zeros(self.event_shape, dtype=torch.float64)

    def update(self, x: torch.Tensor):
        if x.dim() == len(self.event_shape) + 1:
            x = x[None]
        elif x.dim() != len(self.event_shape) + 2:
            raise ValueError
        v = x.detach().to('cpu', torch.float64)
        self.total = self.total + (v ** self.power).sum(dim=(0, 1))
        self.n_seen += x.shape[0] * x.shape[1]

    def add_sums(self, total: torch.Tensor, n_new: int):
        self.total = self.total + total.detach().to('cpu', torch.float64).reshape(self.event_shape)
        self.n_seen += int(n_new)

    def reset(self):
        self.n_seen = 0
        self.total = torch.zeros(self.event_shape, dtype=torch.float64)

    @property
    def running_value(self):
        return self.as_tensor()

    def as_tensor(self):
        if self.n_seen == 0:
            return torch.zeros(self.event_shape)
        return (self.total / self.n_seen).float()


class MCMCExpectationDict:
    def __init__(self, expectations: Dict[str, MCMCExpectation], data_transform=lambda v: v):
        self.expectations = expectations
        self.data_transform = data_transform

    def update(self, x: torch.Tensor):
        xt = self.data_transform(x)
        for e in self.expectations.values():
            e.update(xt)

    def reset(self):
        for e in self.expectations.values():
            e.reset()

    def as_tensor(self):
        return {k: v.as_tensor() for k, v in self.expectations.items()}

    def __getitem__(self, key):
        return self.expectations[key]


@dataclass
class MCMCStatistics:
    event_shape: Union[Tuple[int, ...], torch.Size]
    n_accepted_trajectories: Optional[int] = 0
    n_attempted_trajectories: Optional[int] = 0
    n_divergences: Optional[int] = 0
    n_target_gradient_calls: Optional[int] = 0
    n_target_calls: Optional[int] = 0
    elapsed_time_seconds: Optional[float] = 0.0
    n_nonfinite_log_ratios: int = 0  # chains whose log acceptance ratio was NaN/inf (rejected, langevin.py:106)

    data_transform: Any = None
    expectations: MCMCExpectationDict = None

    def __post_init__(self):
        self.event_shape = tuple(self.event_shape)
        if self.data_transform is None:
            self.data_transform = lambda v: v
        # the reference builds the dict with the identity transform and never rebinds it (App. C #1)
        self.expectations = MCMCExpectationDict({
            'first_moment': MCMCExpectation(self.event_shape, 1),
            'second_moment': MCMCExpectation(self.event_shape, 2),
        })

    def update_counters(self, n_accepted_trajectories: int = 0, n_attempted_trajectories: int = 0,
                        n_divergences: int = 0, n_target_gradient_calls: int = 0, n_target_calls: int = 0):
        self.n_accepted_trajectories = int(self.n_accepted_trajectories + n_accepted_trajectories)
        self.n_attempted_trajectories = int(self.n_attempted_trajectories + n_attempted_trajectories)
        self.n_divergences = int(self.n_divergences + n_divergences)
        self.n_target_gradient_calls = int(self.n_target_gradient_calls + n_target_gradient_calls)
        self.n_target_calls = int(self.n_target_calls + n_target_calls)

    def update_elapsed_time(self, delta_time_seconds: float):
        self.elapsed_time_seconds = float(self.elapsed_time_seconds + delta_time_seconds)

    def absorb_device_sums(self, sum_x: torch.Tensor, sum_x2: torch.Tensor, n_new: int):
        """Fold kernel-accumulated sums (hip.DeviceStats) into the expectations."""
        self.expectations['first_moment'].add_sums(sum_x, n_new)
        self.expectations['second_moment'].add_sums(sum_x2, n_new)

    @property
    def running_first_moment(self):
        return self.expectations['first_moment'].as_tensor()

    @property
    def running_second_moment(self):
        return self.expectations['second_moment'].as_tensor()

    @property
    def running_variance(self):
        return self.running_second_moment - self.running_first_moment ** 2

    @property
    def acceptance_rate(self):
        if self.n_attempted_trajectories == 0:
            return torch.nan
        return self.n_accepted_trajectories / self.n_attempted_trajectories

    @property
    def calls_per_second(self):
        if self.elapsed_time_seconds > 0:
            return self.n_target_calls / self.elapsed_time_seconds
        return torch.nan

    @property
    def grads_per_second(self):
        if self.elapsed_time_seconds > 0:
            return self.n_target_gradient_calls / self.elapsed_time_seconds
        return torch.nan

    def __repr__(self):
        return (f"acc-rate: {self.acceptance_rate:.2f}, "
                f"kcalls/s: {self.calls_per_second / 1000:.2f}, "
                f"kgrads/s: {self.grads_per_second / 1000:.2f}, "
                f"divergences: {self.n_divergences}")

    def as_dict(self):
        return {
            'n_accepted_trajectories': self.n_accepted_trajectories,
            'n_attempted_trajectories': self.n_attempted_trajectories,
            'n_divergences': self.n_divergences,
            'n_target_gradient_calls': self.n_target_gradient_calls,
            'n_target_calls': self.n_target_calls,
            'elapsed_time_seconds': self.elapsed_time_seconds,
            'grads_per_second': self.grads_per_second,
            'acceptance_rate': self.acceptance_rate,
            'calls_per_second': self.calls_per_second,
        }


class MCMCSamples:
    """Kept states `(n_kept, n_chains, *event_shape)`; semantics of base.py:215-271.

    The samplers let the kernels write every step of a launch straight into a device buffer `(k, n, d)` and
    `add` it once per launch; rows stay on the device (`as_device_tensor`) until `.samples` is read, which is the
    only device-to-host copy (the reference does one `.cpu()` per step, base.py:257).
    """

    def __init__(self, event_shape, store_samples: bool = True, thinning: int = 1, max_samples: int = None):
        self.event_shape = tuple(event_shape)
        self.store_samples = store_samples
        self.thinning = thinning
        self.max_samples = max_samples
        self.reset()

    def reset(self):
        self._chunks: List[torch.Tensor] = []  # each (k, n, *event)
        self.n_samples = 0
        self.seen_samples = 0
        self.last_sample = None

    def __getitem__(self, index):
        if index == -1 or index == self.n_samples - 1:
            return self.last_sample
        return self.as_device_tensor()[index]

    def add(self, x: torch.Tensor):
        nd = len(self.event_shape)
        if x.dim() == nd + 1 and tuple(x.shape[1:]) == self.event_shape:
            x = x[None]
        elif x.dim() == nd + 2 and tuple(x.shape[2:]) == self.event_shape:
            pass
        else:
            raise ValueError(f"Expected x.shape[1:] or x.shape[2:] to be {self.event_shape}, got {x.shape = }")
        self.last_sample = x[-1].detach().clone()
        if not self.store_samples:
            return
        idx = torch.arange(self.seen_samples, self.seen_samples + len(x))
        keep = (idx % self.thinning) == 0
        self.seen_samples += len(x)
        kept = x.detach()[keep.to(x.device)] if self.thinning != 1 else x.detach()
        if len(kept):
            self._chunks.append(kept)
            self.n_samples += len(kept)
        if self.max_samples is not None and self.n_samples > self.max_samples:
            full = torch.cat(self._chunks, dim=0)[-self.max_samples:]
            self._chunks = [full]
            self.n_samples = len(full)

    def as_device_tensor(self) -> torch.Tensor:
        if not self._chunks:   # nothing kept yet (n_iterations = 0): an empty (0, n_chains, *event) tensor
            if self.last_sample is not None:
                return self.last_sample.new_empty((0,) + tuple(self.last_sample.shape))
            return torch.empty((0, 0) + self.event_shape)
        if len(self._chunks) > 1:
            self._chunks = [torch.cat(self._chunks, dim=0)]
        return self._chunks[0]

    def as_tensor(self) -> torch.Tensor:
        return self.as_device_tensor().cpu()


@dataclass
class MCMCOutput:
    event_shape: Union[Tuple[int, ...], torch.Size]
    running_samples: MCMCSamples = None
    statistics: Optional[MCMCStatistics] = None
    kernel: Optional[MCMCKernel] = None
    store_samples: bool = True
    max_samples: int = None

    def __post_init__(self):
        self.event_shape = tuple(self.event_shape)
        if self.running_samples is None:
            self.running_samples = MCMCSamples(self.event_shape, store_samples=self.store_samples,
                                               max_samples=self.max_samples)
        if self.statistics is None:
            self.statistics = MCMCStatistics(self.event_shape)

    @property
    def samples(self) -> Union[torch.Tensor, None]:
        if not self.store_samples:
            return None
        return self.running_samples.as_tensor()

    @property
    def samples_device(self) -> Union[torch.Tensor, None]:
        if not self.store_samples:
            return None
        return self.running_samples.as_device_tensor()

    def resample(self, n: int) -> torch.Tensor:
        flat = self.samples.flatten(0, 1)
        mask = torch.randint(low=0, high=len(flat), size=(n,))
        return flat[mask]

    @property
    def mean(self):
        return self.statistics.running_first_moment

    @property
    def variance(self):
        return self.statistics.running_second_moment - self.statistics.running_first_moment ** 2

    @property
    def second_moment(self):
        return self.statistics.running_second_moment


class Sampler:
    """Sampler protocol of base.py:317-348: `.warmup(x0, ...)` / `.sample(x0, ...)` -> MCMCOutput."""

    def __init__(self, event_shape, target, kernel: MCMCKernel, params: MCMCParameters):
        self.event_shape = tuple(event_shape)
        self.target = target
        self.kernel = kernel
        self.params = params
        self.event_size = int(torch.prod(torch.as_tensor(self.event_shape)))
        self.seed = None          # native-stream seed; None -> drawn from torch's global RNG per sample() call
        self.shard = None         # dist.Shard when chains are split over GPUs
        self.replay = None        # ReplayNoise-like object for parity tests (see samplers/common.py)

    @property
    def name(self):
        return "Generic sampler"

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        raise NotImplementedError

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        raise NotImplementedError
