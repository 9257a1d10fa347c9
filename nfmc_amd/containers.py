"""Host-side containers of a device-resident run.  The public names and attributes are the ones callers of
nfmc/algorithms/sampling/base.py use (so results read the same), the implementation is this package's own:

* moments are fp64 per-coordinate SUMS that the HIP kernels accumulate on the GPU (hip.DeviceStats) plus a count;
  `running_first_moment` etc. divide on access.  The reference's streaming update (base.py:88-95) is the plain mean
  over everything seen, which is what sum / count gives (tests/test_host_cpu.py checks the two against each other).
* counters live in one table per statistics class (`COUNTERS`), so `update_counters`, `as_dict` and the multi-GPU
  merge (dist.Shard.merge_statistics) are generic over them; subclasses extend the table (jump counters).
* kept states are slabs `(k, n_chains, *event)` that the kernels write straight into device memory, one slab per
  launch; thinning / `max_samples` / `last_sample` follow base.py:234-263; the only device-to-host copy happens
  when `.samples` is read (`samples_device` avoids it).
"""
import math
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch


def _shape(event_shape) -> tuple:
    return tuple(int(s) for s in event_shape)


# ------------------------------------------------------------------------------------------------ kernels / parameters
@dataclass
class MCMCKernel:
    """State a sampler tunes and returns with its output (`MCMCOutput.kernel`)."""

    def __post_init__(self):   # subclasses chain through here
        return None

    def __repr__(self):
        raise NotImplementedError('kernels describe themselves')


@dataclass
class NFMCKernel(MCMCKernel):
    event_shape: Sequence[int]
    flow: Any = None   # duck-typed: anything with the Flow methods; default: this package's RealNVP

    def __post_init__(self):
        super().__post_init__()
        if self.flow is not None:
            return
        from .flows import Flow, RealNVP
        self.flow = Flow(RealNVP(self.event_shape))

    def __repr__(self):
        return 'NFMCKernel(event_shape=%s)' % (_shape(self.event_shape),)


@dataclass
class MCMCParameters:
    n_iterations: int = 100
    n_warmup_iterations: int = 100
    tuning: bool = False          # True while warmup adapts the kernel
    store_samples: bool = True
    # MCMCSamples.thinning / .max_samples (base.py:222-224), which the reference's samplers never set (their output object
    # is built inside sample()); here they are parameters, applied on the device before every launch
    thinning: int = 1
    max_samples: Optional[int] = None
    spill_to_host: bool = False   # start the device-to-host copy of the kept states at the end of sample(), asynchronously

    def __post_init__(self):
        return None

    def _set_tuning(self, flag: bool):
        self.tuning = bool(flag)

    def tuning_mode(self):
        self._set_tuning(True)

    def sampling_mode(self):
        self._set_tuning(False)


# defaults of the refit inside a run (base.py:46-61): short, early-stopped, quiet
_REFIT_DEFAULTS = (('early_stopping', True), ('early_stopping_threshold', 50), ('batch_size', 'adaptive'),
                   ('show_progress', False))


@dataclass
class NFMCParameters(MCMCParameters):
    train_pct: float = 0.7
    max_train_size: int = 4096
    max_val_size: int = 4096
    flow_fit_kwargs: Optional[Dict[str, Any]] = None

    def __post_init__(self):
        super().__post_init__()
        if self.flow_fit_kwargs is None:
            self.flow_fit_kwargs = dict(_REFIT_DEFAULTS)


# ------------------------------------------------------------------------------------------------ moments
class MCMCExpectation:
    """E[x^power] per coordinate over every (step, chain) seen: an fp64 sum and a count."""

    def __init__(self, event_shape, power: int):
        self.event_shape = _shape(event_shape)
        self.power = int(power)
        self.reset()

    def reset(self):
        self.total = torch.zeros(self.event_shape, dtype=torch.float64)
        self.n_seen = 0

    def add_sums(self, total: torch.Tensor, n_new: int):
        """Sums the kernels accumulated over `n_new` (step, chain) pairs."""
        self.total = self.total + total.detach().to('cpu', torch.float64).reshape(self.event_shape)
        self.n_seen += int(n_new)

    def update(self, x: torch.Tensor):
        """Host-side path: x is (n_chains, *event) or (n_steps, n_chains, *event)."""
        lead = x.dim() - len(self.event_shape)
        if lead not in (1, 2):
            raise ValueError('expected (n_chains, *event) or (n_steps, n_chains, *event), got %s' % (tuple(x.shape),))
        rows = x.detach().to('cpu', torch.float64).reshape(-1, *self.event_shape)
        self.add_sums(rows.pow(self.power).sum(dim=0), rows.shape[0])

    def as_tensor(self) -> torch.Tensor:
        if not self.n_seen:
            return torch.zeros(self.event_shape)
        return (self.total / self.n_seen).float()

    running_value = property(as_tensor)


class MCMCExpectationDict:
    """Named expectations fed from the same (optionally transformed) states."""

    def __init__(self, expectations: Dict[str, MCMCExpectation], data_transform: Callable = None):
        self.expectations = expectations
        self.data_transform = data_transform if data_transform is not None else (lambda v: v)

    def __getitem__(self, name):
        return self.expectations[name]

    def update(self, x: torch.Tensor):
        seen = self.data_transform(x)
        for member in self.expectations.values():
            member.update(seen)

    def reset(self):
        for member in self.expectations.values():
            member.reset()

    def as_tensor(self) -> Dict[str, torch.Tensor]:
        return {name: member.as_tensor() for name, member in self.expectations.items()}


def _rate(numerator, denominator):
    return numerator / denominator if denominator > 0 else math.nan


@dataclass
class MCMCStatistics:
    event_shape: Sequence[int]
    n_accepted_trajectories: Optional[int] = 0
    n_attempted_trajectories: Optional[int] = 0
    n_divergences: Optional[int] = 0
    n_target_gradient_calls: Optional[int] = 0
    n_target_calls: Optional[int] = 0
    elapsed_time_seconds: Optional[float] = 0.0
    n_nonfinite_log_ratios: int = 0   # transitions whose log acceptance ratio was NaN / inf: rejected (langevin.py:106)
    data_transform: Any = None
    expectations: MCMCExpectationDict = None

    # integer counters `update_counters` accepts and `as_dict` reports; subclasses extend the tuple
    COUNTERS = ('n_accepted_trajectories', 'n_attempted_trajectories', 'n_divergences', 'n_target_gradient_calls',
                'n_target_calls')

    def __post_init__(self):
        self.event_shape = _shape(self.event_shape)
        if self.data_transform is None:
            self.data_transform = lambda v: v
        # moments of the UNtransformed state, like the reference, whose dict never sees a transform
        self.expectations = MCMCExpectationDict({name: MCMCExpectation(self.event_shape, power)
                                                 for name, power in (('first_moment', 1), ('second_moment', 2))})

    # -- accumulation
    def update_counters(self, **increments):
        for name, inc in increments.items():
            if name not in self.COUNTERS:
                raise TypeError('unknown counter %r' % name)
            setattr(self, name, int(getattr(self, name) + inc))

    def update_elapsed_time(self, delta_time_seconds: float):
        self.elapsed_time_seconds = float(self.elapsed_time_seconds) + float(delta_time_seconds)

    def absorb_device_sums(self, sum_x: torch.Tensor, sum_x2: torch.Tensor, n_new: int):
        """Per-coordinate sums of x and x^2 over `n_new` (step, chain) pairs, from hip.DeviceStats."""
        for name, total in (('first_moment', sum_x), ('second_moment', sum_x2)):
            self.expectations[name].add_sums(total, n_new)

    # -- derived quantities
    @property
    def running_first_moment(self):
        return self.expectations['first_moment'].as_tensor()

    @property
    def running_second_moment(self):
        return self.expectations['second_moment'].as_tensor()

    @property
    def running_variance(self):
        mean = self.running_first_moment
        return self.running_second_moment - mean * mean

    @property
    def acceptance_rate(self):
        return _rate(self.n_accepted_trajectories, self.n_attempted_trajectories)

    @property
    def calls_per_second(self):
        return _rate(self.n_target_calls, self.elapsed_time_seconds)

    @property
    def grads_per_second(self):
        return _rate(self.n_target_gradient_calls, self.elapsed_time_seconds)

    RATES = ('grads_per_second', 'acceptance_rate', 'calls_per_second')

    def as_dict(self) -> Dict[str, Any]:
        names = self.COUNTERS + ('elapsed_time_seconds',) + self.RATES
        return {name: getattr(self, name) for name in names}

    def _summary(self):
        return [('acc-rate', '%.2f' % self.acceptance_rate), ('kcalls/s', '%.2f' % (self.calls_per_second / 1000)),
                ('kgrads/s', '%.2f' % (self.grads_per_second / 1000)), ('divergences', str(self.n_divergences))]

    def __repr__(self):
        return ', '.join('%s: %s' % pair for pair in self._summary())


# ------------------------------------------------------------------------------------------------ kept states
class DeviceSampleStore:
    """The device slab behind `NfmcSampleStore` (include/nfmc_hip.h): `MCMCSamples.add`'s thinning and `max_samples`
    window (base.py:249-263) decided BEFORE each launch, so the kernels write only states that are still kept at the end
    and the slab never has more than min(ceil(total / thinning), max_samples) rows -- the reference appends every
    state to a host list and slices it afterwards.  Row j mod rows holds the j-th kept state (a ring)."""

    def __init__(self, n, d, device, total, thinning=1, max_samples=None):
        self.n, self.d = int(n), int(d)
        self.thinning = max(1, int(thinning))
        kept_total = -(-int(total) // self.thinning)
        self.rows = kept_total if not max_samples else min(kept_total, int(max_samples))
        self.buf = torch.empty(self.rows, self.n, self.d, dtype=torch.float32, device=device) if self.rows > 0 else None
        self.seen = 0   # transitions offered so far

    @property
    def n_kept(self):
        """Kept states that survive: min(#kept so far, rows)."""
        return min(-(-self.seen // self.thinning), self.rows)

    def plan(self, k):
        """(stride, countdown, ring_rows, row) of NfmcSampleStore for a launch that offers the next k transitions
        (advances the store): countdown = transitions to skip before the launch's first kept one, row = its ring row."""
        t = self.thinning
        desc = (t, (-self.seen) % t, max(self.rows, 1), (-(-self.seen // t)) % max(self.rows, 1))
        self.seen += int(k)
        return desc

    def struct(self, k):
        """The NfmcSampleStore itself (device pointer + plan)."""
        from . import hip
        stride, countdown, rows, row = self.plan(k)
        if self.buf is None:
            return hip.dense_store(None)
        return hip.NfmcSampleStore(hip.ptr(self.buf), stride, countdown, rows, row)

    def add_dense(self, x):
        """Offer (k, n, d) states that already sit in a dense device tensor (host-driven paths)."""
        k = int(x.shape[0])
        if self.buf is not None and k > 0:
            t = self.thinning
            idx = torch.arange(self.seen, self.seen + k)
            keep = idx[idx % t == 0][-self.rows:]           # more kept than rows: only the newest `rows` survive
            if keep.numel():
                rows = ((keep // t) % self.rows).to(x.device)
                self.buf.index_copy_(0, rows, x.reshape(k, self.n, self.d)[(keep - self.seen).to(x.device)])
        self.seen += k

    def ordered(self):
        """(n_kept, n, d) in chronological order (a view when the ring has not wrapped)."""
        if self.buf is None:
            return torch.empty(0, self.n, self.d)
        kept = -(-self.seen // self.thinning)
        if kept <= self.rows:
            return self.buf[:kept]
        start = kept % self.rows                           # oldest surviving row
        return self.buf if start == 0 else torch.cat([self.buf[start:], self.buf[:start]])


class MCMCSamples:
    """States kept from a run, `(n_kept, n_chains, *event_shape)`."""

    def __init__(self, event_shape, store_samples: bool = True, thinning: int = 1, max_samples: int = None):
        self.event_shape = _shape(event_shape)
        self.store_samples = store_samples
        self.thinning = thinning
        self.max_samples = max_samples
        self.reset()

    def reset(self):
        self._host = None                       # host copy of the kept states (spill / as_tensor)
        self._slabs: List[torch.Tensor] = []   # device tensors (k, n_chains, *event), in order
        self.n_samples = 0                      # rows kept
        self.seen_samples = 0                   # rows offered (thinning counts these)
        self.last_sample = None                 # newest state, kept even when nothing is stored

    def _as_steps(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.dim() - len(self.event_shape)
        if lead in (1, 2) and tuple(x.shape[lead:]) == self.event_shape:
            return x if lead == 2 else x.unsqueeze(0)
        raise ValueError('states must be (n_chains, *%s) or (n_steps, n_chains, *%s), got %s'
                         % (self.event_shape, self.event_shape, tuple(x.shape)))

    def add(self, x: torch.Tensor):
        steps = self._as_steps(x).detach()
        self.last_sample = steps[-1].clone()
        if not self.store_samples:
            return
        first = self.seen_samples
        self.seen_samples += len(steps)
        if self.thinning != 1:
            offset = (-first) % self.thinning          # first offered row whose global index is kept
            steps = steps[offset::self.thinning]
        if len(steps):
            self._slabs.append(steps)
            self.n_samples += len(steps)
        if self.max_samples is not None and self.n_samples > self.max_samples:
            newest = self.as_device_tensor()[-self.max_samples:]
            self._slabs, self.n_samples = [newest], len(newest)

    def adopt_store(self, store: DeviceSampleStore, spill: bool = False):
        """Take over what the kernels kept (thinning and window already applied on the device); `spill` starts the
        asynchronous copy to pinned host memory right away."""
        kept = store.ordered()
        self.seen_samples += store.seen
        self.thinning = store.thinning
        self._slabs = [kept.reshape(kept.shape[0], kept.shape[1], *self.event_shape)] if kept.shape[0] else []
        self.n_samples = int(kept.shape[0])
        self._host = None
        if spill:
            self.spill()

    def as_device_tensor(self) -> torch.Tensor:
        if len(self._slabs) > 1:
            self._slabs = [torch.cat(self._slabs, dim=0)]
        if self._slabs:
            return self._slabs[0]
        # nothing kept (e.g. n_iterations = 0): an empty stack over the chains seen so far
        if self.last_sample is None:
            return torch.empty((0, 0) + self.event_shape)
        return self.last_sample.new_empty((0,) + tuple(self.last_sample.shape))

    def spill(self):
        """Start copying the kept states to pinned host memory on a side stream and return at once (the asynchronous
        spill of SURVEY 8f3); `as_tensor()` waits for it.  A no-op for host-resident or empty stores."""
        dev_t = self.as_device_tensor()
        if not dev_t.is_cuda or dev_t.numel() == 0 or getattr(self, '_host', None) is not None:
            return
        stream = torch.cuda.Stream(device=dev_t.device)
        stream.wait_stream(torch.cuda.current_stream(dev_t.device))     # after the kernels that wrote the slab
        host = torch.empty(dev_t.shape, dtype=dev_t.dtype, pin_memory=True)
        with torch.cuda.stream(stream):
            host.copy_(dev_t, non_blocking=True)
        self._host, self._host_stream, self._host_src = host, stream, dev_t

    def as_tensor(self) -> torch.Tensor:
        """The kept states on the host.  Copied once: later reads return the same host tensor until states are added."""
        dev_t = self.as_device_tensor()
        if not dev_t.is_cuda:
            return dev_t
        if getattr(self, '_host', None) is None or self._host_src is not dev_t:
            self._host = None
            self.spill()
            if getattr(self, '_host', None) is None:      # empty store
                return dev_t.cpu()
        self._host_stream.synchronize()
        return self._host

    def __getitem__(self, index):
        if index in (-1, self.n_samples - 1):
            return self.last_sample
        return self.as_device_tensor()[index]


@dataclass
class MCMCOutput:
    event_shape: Sequence[int]
    running_samples: MCMCSamples = None
    statistics: Optional[MCMCStatistics] = None
    kernel: Optional[MCMCKernel] = None
    store_samples: bool = True
    max_samples: int = None
    kernel_events: Any = None   # bench.py: (label, start, end) HIP events of the launches, when kernel timing was on

    def __post_init__(self):
        self.event_shape = _shape(self.event_shape)
        if self.statistics is None:
            self.statistics = MCMCStatistics(self.event_shape)
        if self.running_samples is None:
            self.running_samples = MCMCSamples(self.event_shape, store_samples=self.store_samples,
                                               max_samples=self.max_samples)

    def _kept(self, on_device: bool):
        if not self.store_samples:
            return None
        store = self.running_samples
        return store.as_device_tensor() if on_device else store.as_tensor()

    @property
    def samples(self) -> Optional[torch.Tensor]:
        """(n_kept, n_chains, *event) on the host, or None when the run did not store samples."""
        return self._kept(False)

    @property
    def samples_device(self) -> Optional[torch.Tensor]:
        return self._kept(True)

    def spill(self):
        """Begin the asynchronous device-to-host copy of the kept states (see MCMCSamples.spill)."""
        if self.store_samples:
            self.running_samples.spill()
        return self

    def resample(self, n: int) -> torch.Tensor:
        """n draws with replacement from all kept (step, chain) states."""
        pool = self.samples.flatten(0, 1)
        return pool[torch.randint(len(pool), (n,))]

    @property
    def mean(self):
        return self.statistics.running_first_moment

    @property
    def second_moment(self):
        return self.statistics.running_second_moment

    @property
    def variance(self):
        return self.statistics.running_variance


# ------------------------------------------------------------------------------------------------ sampler protocol
class Sampler:
    """`.warmup(x0, ...)` / `.sample(x0, ...)` -> MCMCOutput (the protocol `sample()` drives, base.py:317-348)."""

    name = 'Generic sampler'

    def __init__(self, event_shape, target, kernel: MCMCKernel, params: MCMCParameters):
        self.event_shape = _shape(event_shape)
        self.event_size = math.prod(self.event_shape)
        self.target = target
        self.kernel = kernel
        self.params = params
        self.fuse = 'auto'    # 'auto': a plain callable target that potentials.recognize() reproduces as a quadratic
                              # is evaluated in closed form inside the kernels; False / 'never': always call it
        self.seed = None      # native-stream seed; None: drawn from torch's global RNG per sample() call
        self.rng_rounds = 10  # 10: Philox4x32-10 (the library's stream); 7: the opt-in Philox4x32-7 stream of the exact-fit kernels
        self.shard = None     # dist.Shard when the chains are split over GPUs
        self.replay = None    # (normals, uniforms) to replay instead of the native streams (parity tests)

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        raise NotImplementedError

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        raise NotImplementedError
