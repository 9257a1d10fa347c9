"""Independent Metropolis-Hastings with the flow as proposal (nfmc/algorithms/sampling/nfmc/imh.py).
`FixedIMH`: all iterations run inside `nfmc_flow_mh_steps_f32` with the per-chain log q(x) cached on
the device (imh.py:214,233).  `AdaptiveIMH`: one kernel launch per iteration (log q recomputed, the flow
moves), the one-epoch refit between launches on the device (imh.py:147-175)."""
import time
from copy import deepcopy
from dataclasses import dataclass
from typing import Optional

import torch
from tqdm import tqdm

from .. import hip
from ..containers import DeviceSampleStore, MCMCOutput, NFMCKernel, NFMCParameters, Sampler
from .common import Run, chunks, progress, resolve_target
from .jump import (flow_is_native, flow_mh_supported, imh_parallel_ok, launch_flow_mh, launch_imh_parallel,
                   split_flow_mh)


@dataclass
class IMHKernel(NFMCKernel):
    pass


@dataclass
class IMHParameters(NFMCParameters):
    train_distribution: str = 'uniform'
    adaptation_dropoff: float = 0.9999
    warmup_fit_kwargs: dict = None

    def __post_init__(self):
        super().__post_init__()
        if self.train_distribution not in ['bounded_geom_approx', 'bounded_geom', 'uniform']:
            raise ValueError
        if self.warmup_fit_kwargs is None:
            self.warmup_fit_kwargs = {
                'early_stopping': True,
                'early_stopping_threshold': 50,
                'keep_best_weights': True,
                'n_samples': 1,
                'n_epochs': 500,
                'lr': 0.05,
                'check_for_divergences': True
            }


def _accepts_potential(flow) -> bool:
    """Only the package's own Flow takes the extra keyword; foreign flow objects get the reference's call."""
    from ..flows import Flow
    return isinstance(flow, Flow)


class AbstractIMH(Sampler):
    def __init__(self, event_shape, target, kernel: Optional[IMHKernel] = None,
                 params: Optional[IMHParameters] = None):
        if kernel is None:
            kernel = IMHKernel(event_shape)
        if params is None:
            params = IMHParameters()
        super().__init__(event_shape, target, kernel, params)

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """imh.py:60-75: variational fit of the flow to the target, then one flow sample as state."""
        # the closed-form descriptor of the target, when there is one, lets the fit step run on the device (flow_training.py)
        pot = resolve_target(self.target, tuple(x0.shape[1:]), getattr(self, 'fuse', 'auto'), x0)
        extra = {'potential': pot} if pot is not None and _accepts_potential(self.kernel.flow) else {}
        self.kernel.flow.variational_fit(lambda v: -self.target(v), **self.params.warmup_fit_kwargs,
                                         show_progress=show_progress, time_limit_seconds=time_limit_seconds, **extra)
        out = MCMCOutput(event_shape=tuple(x0.shape[1:]), store_samples=self.params.store_samples)
        out.running_samples.add(self.kernel.flow.sample(x0.shape[0]).detach())
        return out

    @property
    def name(self):
        return "Abstract IMH"


class FixedIMH(AbstractIMH):
    @property
    def name(self):
        return "Fixed IMH"

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """imh.py:200-255 on the device."""
        run = Run(self, x0)
        n, d, event_shape = run.n, run.d, run.event_shape
        out = MCMCOutput(event_shape, store_samples=self.params.store_samples,
                         max_samples=getattr(self.params, 'max_samples', None))
        flow = self.kernel.flow
        T = int(self.params.n_iterations)
        pot = resolve_target(self.target, event_shape, self.fuse, run.x)
        fused = pot is not None and flow_is_native(flow)
        store = DeviceSampleStore(n, d, run.dev, T, getattr(self.params, 'thinning', 1),
                                  getattr(self.params, 'max_samples', None)) if (self.params.store_samples and T > 0) else None
        logq = torch.empty(n, dtype=torch.float32, device=run.dev)
        fused = fused and flow_mh_supported(run, flow, pot, logq)
        t0 = time.time()
        done = 0
        unlimited = time_limit_seconds is None and not show_progress
        limit = hip.MAX_STEPS_PER_CALL if unlimited else 16
        parallel = fused and imh_parallel_ok(run, flow, pot, logq)
        bar = progress(show_progress, total=T, desc=self.name)
        if not fused:
            logq.copy_(flow.log_prob(run.x.reshape(n, *event_shape)).detach().to(run.dev, torch.float32))  # imh.py:214
        while done < T:
            if run.time_is_up(t0, time_limit_seconds):
                break
            k = min(limit, T - done) if fused else 1
            if fused and parallel and unlimited:
                # all proposals of the chunk at once: as many steps as 2^26 work items (1 GiB of work arrays) allow
                k = min(T - done, hip.IMH_PARALLEL_MAX_STEPS, max(16, (1 << 26) // max(n, 1)))
            if fused and parallel:
                launch_imh_parallel(run, flow, pot, logq, k, done, done > 0,
                                    run.stats.struct(defer=True, attempted=n * k), store)
            elif fused:
                launch_flow_mh(run, flow, pot, logq, k, done, done > 0, True,
                               run.stats.struct(defer=True, attempted=n * k), store)
            else:
                split_flow_mh(run, flow, self.target, event_shape, done, True, run.stats.struct(), logq=logq)
                if store is not None:
                    store.add_dense(run.x[None])
            done += k
            bar.update(k)
        bar.close()
        # the final-state copy and the statistics fold go out right behind the last kernel; the one device-to-host
        # copy of the totals is the only synchronisation of the call
        last_sample = run.x.reshape(n, *event_shape).clone()
        sum_x, sum_x2, cnt, _jc = run.stats.host_totals()
        st = out.statistics
        st.update_counters(n_target_calls=2 * n * done, n_accepted_trajectories=int(cnt[hip.CNT_ACCEPTED]),
                           n_attempted_trajectories=int(cnt[hip.CNT_ATTEMPTED]))
        st.n_nonfinite_log_ratios = int(cnt[hip.CNT_NONFINITE])
        st.absorb_device_sums(sum_x.reshape(event_shape), sum_x2.reshape(event_shape), n * done)
        if store is not None:
            out.running_samples.adopt_store(store, getattr(self.params, 'spill_to_host', False))
        out.running_samples.last_sample = last_sample
        st.update_elapsed_time(time.time() - t0)
        out.kernel = self.kernel
        out.kernel_events = run.kernel_events
        if run.shard is not None:
            run.shard.merge_statistics(st)
        return out


class HostDraws:
    """The scalar host-side draws of AdaptiveIMH (imh.py:148,156-160).  Default: torch's global generator,
    like the reference; tests replay recorded values.  With sharded chains rank 0 draws for everybody."""

    def __init__(self, shard=None, uniforms=None, ints=None):
        self.shard = shard
        self.uniforms = list(uniforms) if uniforms is not None else None
        self.ints = list(ints) if ints is not None else None

    def rand(self) -> float:
        u = float(self.uniforms.pop(0)) if self.uniforms is not None else float(torch.rand(size=()))
        if self.shard is not None and self.shard.world > 1:
            u = self.shard.broadcast_int(int(u * 2 ** 53)) / 2 ** 53
        return u

    def randint(self, low: int, high: int) -> int:
        k = int(self.ints.pop(0)) if self.ints is not None else int(torch.randint(low=low, high=high, size=()))
        if self.shard is not None and self.shard.world > 1:
            k = self.shard.broadcast_int(k)
        return k


def bounded_geom_index(p: float, max_val: int, u: float) -> int:
    """imh.py:39-45 (`sample_bounded_geom`) with the uniform passed in."""
    v = torch.arange(0, max_val + 1)
    pdf = p * (1 - p) ** (max_val - v) / (1 - (1 - p) ** (max_val + 1))
    cdf = torch.cumsum(pdf, dim=0)
    return int(torch.searchsorted(cdf, torch.tensor(u, dtype=cdf.dtype), right=True))


class AdaptiveIMH(AbstractIMH):
    """imh.py:82-181.  The proposal flow is refitted (one epoch of maximum likelihood) on ONE stored state,
    picked by `train_distribution`, with probability adaptation_dropoff^i after iteration i."""

    host_draws = None   # tests: (uniforms, ints) to replay instead of torch's global generator

    def __init__(self, event_shape, target, kernel: Optional[IMHKernel] = None,
                 params: Optional[IMHParameters] = None):
        if params is None:
            params = IMHParameters()
        if not params.store_samples:
            # imh.py:92-95 means to do this (it assigns through self.params before it exists)
            print('Warning: params.store_samples is False')
            print('Warning: setting params.store_samples to True')
            params.store_samples = True
        super().__init__(event_shape, target, kernel, params)

    @property
    def name(self):
        return "Adaptive IMH"

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        if not self.params.store_samples:
            print("WARNING: params.store_samples is False")
            print("WARNING: cannot adapt IMH kernel without storing samples - params.store_samples")
            print("WARNING: setting params.store_samples to True")
            self.params.store_samples = True
        run = Run(self, x0)
        n, d, event_shape = run.n, run.d, run.event_shape
        if run.shard is not None and run.shard.world > 1 and run.n_global % run.shard.world != 0:
            raise ValueError('adaptive_imh with sharded chains needs n_chains divisible by the number of ranks')
        out = MCMCOutput(event_shape, store_samples=True)
        flow = self.kernel.flow
        T = int(self.params.n_iterations)
        pot = resolve_target(self.target, event_shape, self.fuse, run.x)
        fused = pot is not None and flow_is_native(flow)
        host = HostDraws(run.shard, *(self.host_draws or (None, None)))
        buf = torch.empty(max(T, 1), n, d, dtype=torch.float32, device=run.dev)
        logq = torch.empty(n, dtype=torch.float32, device=run.dev)
        fused = fused and flow_mh_supported(run, flow, pot, logq)
        t0 = time.time()
        done, n_refits = 0, 0
        bar = progress(show_progress, total=T, desc=self.name)
        for i in range(T):
            if run.time_is_up(t0, time_limit_seconds):
                break
            if fused:
                launch_flow_mh(run, flow, pot, logq, 1, i, False, True,
                               run.stats.struct(defer=True, attempted=n), buf[i:i + 1])              # :121-134
            else:
                split_flow_mh(run, flow, self.target, event_shape, i, True, run.stats.struct())
                buf[i].copy_(run.x)
            done += 1
            if host.rand() < self.params.adaptation_dropoff ** i:                       # :147-149
                dist = self.params.train_distribution
                if dist == 'uniform':
                    k = host.randint(0, done)                                           # :156
                elif dist == 'bounded_geom_approx':
                    k = host.randint(max(0, done - 100), done)                          # :158
                elif dist == 'bounded_geom':
                    k = bounded_geom_index(0.025, done - 1, host.rand())                # :160
                else:
                    raise ValueError
                x_train = buf[k]
                if run.shard is not None:
                    x_train = run.shard.all_gather_rows(x_train)    # C1: every rank fits the same rows
                weights = deepcopy(flow.state_dict())                                   # :166
                try:
                    flow.fit(x_train.reshape(-1, *event_shape), n_epochs=1, show_progress=False)   # :168
                    n_refits += 1
                except ValueError:
                    flow.load_state_dict(weights)                                       # :170
            bar.update(1)
        bar.close()
        # the final-state copy and the statistics fold go out right behind the last kernel; the one device-to-host
        # copy of the totals is the only synchronisation of the call
        last_sample = run.x.reshape(n, *event_shape).clone()
        sum_x, sum_x2, cnt, _jc = run.stats.host_totals()
        st = out.statistics
        # imh.py:140-144 books the 2n target evaluations as gradient calls; kept
        st.update_counters(n_target_gradient_calls=2 * n * done, n_accepted_trajectories=int(cnt[hip.CNT_ACCEPTED]),
                           n_attempted_trajectories=int(cnt[hip.CNT_ATTEMPTED]))
        st.n_nonfinite_log_ratios = int(cnt[hip.CNT_NONFINITE])
        st.absorb_device_sums(sum_x.reshape(event_shape), sum_x2.reshape(event_shape), n * done)
        if done > 0:
            out.running_samples.add(buf[:done].reshape(done, n, *event_shape))
        out.running_samples.last_sample = last_sample
        st.update_elapsed_time(time.time() - t0)
        self.n_refits = n_refits
        out.kernel = self.kernel
        if run.shard is not None:
            run.shard.merge_statistics(st)
        return out
