"""Inner MCMC samplers: Langevin (MALA/ULA) and HMC/UHMC, with the reference's kernel/parameter
dataclasses (nfmc/algorithms/sampling/mcmc/{base,langevin,hmc}.py).

`sample()` runs all `n_iterations` transitions inside HIP kernels (nfmc_mala_steps_f32 /
nfmc_hmc_steps_f32) when the target is a closed-form potential; `propose()` -- the reference's plug-in
seam (mcmc/base.py:27-34) -- stays callable and serves arbitrary Python targets: U and grad U come from
torch autograd on the GPU, proposal / log-ratio / accept-select / moments from the HIP kernels.
"""
import ctypes as C
import math
import os
import time
from copy import deepcopy
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
from tqdm import tqdm

from .. import hip
from ..containers import DeviceSampleStore, MCMCKernel, MCMCOutput, MCMCParameters, Sampler
from ..tuning import DualAveraging, DualAveragingParams
from .common import Run, chunks, imd_tensor, progress, resolve_target


@dataclass
class MetropolisKernel(MCMCKernel):
    event_size: int
    inv_mass_diag: torch.Tensor = None
    step_size: float = 0.01
    da: DualAveraging = None
    da_params: DualAveragingParams = None

    def __post_init__(self):
        super().__post_init__()
        if self.inv_mass_diag is None:
            self.inv_mass_diag = torch.ones(self.event_size)
        else:
            self.inv_mass_diag = torch.as_tensor(self.inv_mass_diag, dtype=torch.float32)   # lists / fp64 / any device
            if tuple(self.inv_mass_diag.shape) != (self.event_size,):
                raise ValueError('inv_mass_diag must have event_size = %d entries, got shape %s'
                                 % (self.event_size, tuple(self.inv_mass_diag.shape)))
        if self.da_params is None:
            self.da_params = DualAveragingParams()
        if self.da is None:
            self.da = DualAveraging(self.step_size, self.da_params)


@dataclass
class MetropolisParameters(MCMCParameters):
    tune_inv_mass_diag: bool = True
    tune_step_size: bool = True
    adjustment: bool = True
    imd_adjustment: float = 1e-3
    # warmup on the device: transitions per controller update.  1 = the reference's schedule (mcmc/base.py:92-96: the
    # kernel is updated after every transition); K > 1 updates once per K transitions with the acceptance rate and the
    # per-coordinate variance pooled over them, K transitions per launch
    tune_every: int = 1


class DeviceTuning:
    """Step size, mass diagonal and dual-averaging state in device memory for the length of a warmup (NfmcTune,
    include/nfmc_hip.h): the controller (mcmc/base.py:142-161, tuning.py:15-41) runs inside each call's statistics fold
    and the next call reads what it wrote, so the warmup loop never waits for the GPU."""

    def __init__(self, sampler, run):
        k, p, da = sampler.kernel, sampler.params, sampler.kernel.da
        st = torch.zeros(int(hip.lib().nfmc_tune_state_doubles(run.d)), dtype=torch.float64)
        st[hip.TUNE_STEP_SIZE] = float(k.step_size)
        st[hip.TUNE_LOG_SMOOTH] = da.log_smooth
        st[hip.TUNE_ERROR_SUM] = da.error_sum
        st[hip.TUNE_ITERATION] = da.iteration
        st[hip.TUNE_ANCHOR] = da.anchor
        st[hip.TUNE_LOG_RAW] = da.log_raw if math.isfinite(da.log_raw) else 0.0
        st[hip.TUNE_TARGET] = da.params.target_acceptance_rate
        st[hip.TUNE_KAPPA] = da.params.kappa
        st[hip.TUNE_GAMMA] = da.params.gamma
        st[hip.TUNE_IMD_ADJUSTMENT] = p.imd_adjustment
        self.state = st.to(run.dev)
        self.imd = k.inv_mass_diag.detach().to(run.dev, torch.float32).contiguous().clone()
        self.tune_step = bool(p.tune_step_size and p.adjustment)            # mcmc/base.py:153
        self.tune_imd = bool(p.tune_inv_mass_diag and run.n > 1)            # mcmc/base.py:146
        self.updates = 0

    def struct(self, every=0, n_steps=1):
        """NfmcTune for a call of n_steps transitions with one controller update per `every` of them."""
        every = int(every) if every and every < n_steps else 0
        self.updates += -(-n_steps // every) if every else 1
        return hip.NfmcTune(hip.ptr(self.state, torch.float64), hip.ptr(self.imd), int(self.tune_step), int(self.tune_imd),
                            every, 0)

    def download(self, kernel):
        """The tuned kernel back on the host (one copy, at the end of the warmup)."""
        st = self.state.cpu()
        if self.tune_imd:
            kernel.inv_mass_diag = self.imd.cpu().to(kernel.inv_mass_diag.dtype)
        if self.tune_step and self.updates:
            da = kernel.da
            da.log_smooth, da.error_sum = float(st[hip.TUNE_LOG_SMOOTH]), float(st[hip.TUNE_ERROR_SUM])
            da.iteration, da.log_raw = int(round(float(st[hip.TUNE_ITERATION]))), float(st[hip.TUNE_LOG_RAW])
            kernel.step_size = float(st[hip.TUNE_STEP_SIZE])


@dataclass
class LangevinKernel(MetropolisKernel):
    event_size: int
    step_size: Optional[float] = None

    def __post_init__(self):
        if self.step_size is None:
            self.step_size = self.event_size ** (-1 / 3)  # langevin.py:16-18
        super().__post_init__()

    def __repr__(self):
        return (f'log step: {math.log(self.step_size):.2f}, '
                f'mass norm: {torch.max(torch.abs(self.inv_mass_diag)):.2f}')


@dataclass
class LangevinParameters(MetropolisParameters):
    pass


@dataclass
class HMCKernel(MetropolisKernel):
    event_size: int
    n_leapfrog_steps: int = 20

    def __repr__(self):
        return (f'log step: {math.log(self.step_size):.2f}, '
                f'leapfrogs: {self.n_leapfrog_steps}, '
                f'mass norm: {torch.max(torch.abs(self.inv_mass_diag)):.2f}')


@dataclass
class HMCParameters(MetropolisParameters):
    pass


class TargetFailure(Exception):
    """The target raised ValueError: the reference's failure channel (langevin.py:111-114, hmc.py:117-120,
    mh.py:63-66): every chain rejects this step and the step counts as ONE divergence."""


def _guarded(fn, *args):
    """Call into user code (`target`, autograd through it).  A ValueError from there becomes TargetFailure; the
    library's own argument errors (hip.NfmcArgumentError, also a ValueError) are never swallowed."""
    try:
        return fn(*args)
    except hip.NfmcArgumentError:
        raise
    except ValueError as e:
        raise TargetFailure(str(e)) from e


def _value_and_grad(target, x, event_shape):
    """U(x), grad U(x) by autograd on the GPU (the reference's recipe, langevin.py:66-70)."""
    with torch.enable_grad():
        xr = x.detach().reshape(x.shape[0], *event_shape).clone().requires_grad_(True)
        u = target(xr)
        g, = torch.autograd.grad(u.sum(), xr)
    return u.detach().reshape(-1).float().contiguous(), g.detach().reshape(x.shape).float().contiguous()


class MCMCSampler(Sampler):

    def __init__(self, event_shape, target, kernel: MCMCKernel, params: MCMCParameters,
                 data_transform=lambda v: v):
        super().__init__(event_shape, target, kernel, params)
        self.data_transform = data_transform

    @property
    def name(self):
        return "Generic MCMC"

    def propose(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, int, int, int]:
        raise NotImplementedError

    def update_kernel(self, data: Dict[str, Any]):
        raise NotImplementedError

    # ---- fused launch of k transitions; implemented by Langevin / HMC
    def _launch(self, run: Run, pot, k, step0, samples, masks_out=None, log_ratio_out=None, jump=None, rng=None, tune=None):
        raise NotImplementedError

    def _counts(self, n, k):
        raise NotImplementedError

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        # mcmc/base.py:39-54
        with torch.no_grad():
            warmup_copy = deepcopy(self)
        warmup_copy.params.tuning_mode()
        warmup_copy.params.n_iterations = self.params.n_warmup_iterations
        out = warmup_copy.sample(x0, show_progress=show_progress, time_limit_seconds=time_limit_seconds)
        self.kernel = warmup_copy.kernel
        new_params = warmup_copy.params
        new_params.n_iterations = self.params.n_iterations
        self.params = new_params
        self.params.sampling_mode()
        return out

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """mcmc/base.py:56-102 on the device."""
        run = Run(self, x0)
        step0 = 0
        n, d = run.n, run.d
        event_shape = run.event_shape
        out = MCMCOutput(event_shape, store_samples=self.params.store_samples,
                         max_samples=getattr(self.params, 'max_samples', None))
        out.statistics.data_transform = self.data_transform
        K = int(self.params.n_iterations)
        pot = resolve_target(self.target, event_shape, self.fuse, run.x)
        # kept states: thinning / max_samples window applied on the device, slab bounded by max_samples (f3)
        store = DeviceSampleStore(n, d, run.dev, K, getattr(self.params, 'thinning', 1),
                                  getattr(self.params, 'max_samples', None)) if (self.params.store_samples and K > 0) else None
        run.stats.zero_()
        self._n_divergences = 0
        t0 = time.time()
        done = 0
        label = f'{self.name} (tuning)' if self.params.tuning else self.name
        bar = progress(show_progress, total=K, desc=label)
        stepwise = self.params.tuning or pot is None
        limit = 1 if stepwise else (hip.MAX_STEPS_PER_CALL if time_limit_seconds is None and not show_progress else 32)
        # warmup of a fused sampler on one GPU: kernel state and controller on the device, no host round trip per step
        # (sharded chains keep the host controller: its statistics are all-reduced over the ranks every step)
        tune = None
        if self.params.tuning:
            # warmup ALWAYS draws from the default Philox4x32-10 stream, whichever controller runs (device, host,
            # sharded): tuning launches use the general kernels, which carry that stream only, and a warmup must not
            # depend on where its controller lives.  `rng_rounds=7` applies to the sampling launches (documented on sample())
            run.rounds = 10
        if (self.params.tuning and pot is not None and isinstance(self, MetropolisSampler)
                and (run.shard is None or run.shard.world == 1) and os.environ.get('NFMC_TUNE_DEVICE', '1') != '0'):
            tune = DeviceTuning(self, run)
            # one ABI call enqueues every (kernel, controller) pair of up to 512 transitions: no host work per update
            tune.every = max(1, int(getattr(self.params, 'tune_every', 1)))
            limit = hip.MAX_STEPS_PER_CALL if time_limit_seconds is None and not show_progress else max(32, tune.every)
        while done < K:
            if run.time_is_up(t0, time_limit_seconds):
                break
            k = min(limit, K - done)
            if tune is not None:
                self._launch(run, pot, k, step0 + done, store, tune=tune)
            elif pot is not None:
                mask_buf = torch.empty(k, n, dtype=torch.uint8, device=run.dev) if self.params.tuning else None
                self._launch(run, pot, k, step0 + done, store, masks_out=mask_buf)
                mask = mask_buf[-1].bool() if mask_buf is not None else None
            else:
                mask = self._split_step(run, step0 + done, store)
            if self.params.tuning and tune is None:
                with torch.no_grad():
                    self.update_kernel({'x': run.x.reshape(n, *event_shape), 'mask': mask})
            done += k
            bar.update(k)
        bar.close()
        # the final-state copy and the statistics fold go out right behind the last kernel; the one device-to-host
        # copy of the totals is the only synchronisation of the call
        last_sample = run.x.reshape(n, *event_shape).clone()
        if tune is not None:
            tune.download(self.kernel)
        sum_x, sum_x2, cnt, _jc = run.stats.host_totals()
        calls, grads = self._counts(n, done)
        out.statistics.update_counters(n_target_calls=calls, n_target_gradient_calls=grads,
                                       n_divergences=self._n_divergences,
                                       n_accepted_trajectories=int(cnt[hip.CNT_ACCEPTED]),
                                       n_attempted_trajectories=int(cnt[hip.CNT_ATTEMPTED]))
        out.statistics.n_nonfinite_log_ratios = int(cnt[hip.CNT_NONFINITE])
        out.statistics.absorb_device_sums(sum_x.reshape(event_shape), sum_x2.reshape(event_shape),
                                          n * done)
        rs = out.running_samples
        if store is not None:
            rs.adopt_store(store, getattr(self.params, 'spill_to_host', False))
        rs.last_sample = last_sample
        out.statistics.update_elapsed_time(time.time() - t0)
        out.kernel = self.kernel
        out.kernel_events = run.kernel_events
        self._cur_run = None
        if run.shard is not None:
            run.shard.merge_statistics(out.statistics)
        return out

    def _rejected_step(self, xf, n_calls, n_grads):
        """What propose() returns when the target raised ValueError: x' = x, nobody accepts (log ratio -inf, also for
        unadjusted kernels, whose select would otherwise accept everything), one divergence; the call counters are the
        reference's, which books them after its try block whatever happened inside."""
        n = xf.shape[0]
        self._last_log_ratio = torch.full((n,), -math.inf, dtype=torch.float32, device=xf.device)
        self._last_uniforms = None
        return xf, torch.zeros(n, dtype=torch.bool, device=xf.device), n_calls, n_grads, 1

    # ---- one transition through the propose() seam (arbitrary targets)
    def _split_step(self, run: Run, step, sample_view):
        n, d = run.n, run.d
        self._cur_run, self._cur_step = run, step
        x_prime, mask_or_lr, n_calls, n_grads, n_divs = self.propose(run.x)
        self._n_divergences = getattr(self, '_n_divergences', 0) + int(n_divs)
        st = hip.NfmcSelectArgs()
        st.x, st.x_prime, st.n, st.d = hip.ptr(run.x), hip.ptr(x_prime), n, d
        st.n_carry = 0
        lr = self._last_log_ratio
        st.log_ratio = hip.ptr(lr) if lr is not None else None
        un = self._last_uniforms
        st.uniforms = hip.ptr(un) if un is not None else None
        st.rng = hip.make_rng(run.seed, run.chain_offset, step, rounds=run.rounds)
        st.rng_tag = hip.TAG_ACCEPT
        st.stats = run.stats.struct()
        mask = torch.empty(n, dtype=torch.uint8, device=run.dev)
        st.mask_out = hip.ptr(mask, torch.uint8)
        hip.check(hip.lib().nfmc_mh_accept_select_f32(C.byref(st), hip.stream()), 'nfmc_mh_accept_select_f32')
        if sample_view is not None:
            if hasattr(sample_view, 'add_dense'):
                sample_view.add_dense(run.x[None])
            else:
                sample_view[0].copy_(run.x)
        return mask.bool()


class MetropolisSampler(MCMCSampler):
    def update_kernel(self, data: Dict[str, Any]):
        """mcmc/base.py:142-161: EMA of the per-coordinate variance + dual-averaging step size.  With sharded chains
        the variance and the acceptance rate are those of ALL chains (one all-reduce of [sum x, sum x^2, n, accepted]),
        so every rank tunes the same kernel."""
        x = data['x']
        mask = data['mask']
        shard = getattr(self, 'shard', None)
        sharded = shard is not None and shard.world > 1
        n_chains = x.shape[0]
        flat = x.flatten(1, -1)
        acc_rate = None
        if sharded:
            d = flat.shape[1]
            xd = flat.double()
            pack = torch.cat([xd.sum(0), (xd * xd).sum(0),
                              xd.new_tensor([float(n_chains), float(mask.sum()) if mask is not None else 0.0])])
            shard.all_reduce_sum_(pack)
            n_all = float(pack[2 * d])
            n_chains = int(round(n_all))
            var_all = ((pack[d:2 * d] - pack[:d] ** 2 / n_all) / max(n_all - 1.0, 1.0)).float()
            acc_rate = float(pack[2 * d + 1]) / n_all
        if n_chains > 1 and self.params.tune_inv_mass_diag:
            var = (var_all if sharded else torch.var(flat, dim=0)).to(self.kernel.inv_mass_diag)
            self.kernel.inv_mass_diag = (self.params.imd_adjustment * var
                                         + (1 - self.params.imd_adjustment) * self.kernel.inv_mass_diag)
        if self.params.tune_step_size and self.params.adjustment:
            if acc_rate is None:
                acc_rate = float(torch.mean(mask.float()))
            error = self.kernel.da_params.target_acceptance_rate - acc_rate
            self.kernel.da.step(error)
            self.kernel.step_size = self.kernel.da.value


class Langevin(MetropolisSampler):
    def __init__(self, event_shape, target, kernel: Optional[LangevinKernel] = None,
                 params: Optional[LangevinParameters] = None):
        if kernel is None:
            kernel = LangevinKernel(event_size=int(torch.prod(torch.as_tensor(event_shape))))
        if params is None:
            params = LangevinParameters()
        super().__init__(event_shape, target, kernel, params)

    @property
    def name(self):
        return 'LMC'

    def _counts(self, n, k):
        per = 2 * n if self.params.adjustment else n  # langevin.py:116-120
        return per * k, per * k

    def _launch(self, run, pot, k, step0, samples, masks_out=None, log_ratio_out=None, jump=None, rng=None, tune=None):
        a = hip.NfmcMalaArgs()
        a.x, a.n, a.d, a.n_steps = hip.ptr(run.x), run.n, run.d, k
        a.step_size = float(self.kernel.step_size)
        a.adjust = (1 if self.params.adjustment else 0) | (2 if getattr(self, 'random_walk', False) else 0)
        imd = tune.imd if tune is not None else imd_tensor(self.kernel, run.dev)
        a.inv_mass_diag = hip.ptr(imd)
        a.pot = pot.descriptor(run.dev)
        # `rng`: an NfmcRng prepared by the caller (replay bookkeeping of fused jump tails)
        a.rng = rng if rng is not None else run.rng(step0, k, adjusted=self.params.adjustment)
        if tune is not None:   # the controller rides on the per-call statistics fold
            a.stats = run.stats.struct()
            a.tune = tune.struct(getattr(tune, 'every', 0), k)
        else:
            a.stats = run.stats.struct(defer=True, attempted=run.n * k, jump_attempted=run.n if jump is not None else 0)
        a.samples = hip.store_struct(samples, k + (1 if jump is not None else 0))
        a.masks_out = hip.ptr(masks_out, torch.uint8) if masks_out is not None else None
        a.log_ratio_out = hip.ptr(log_ratio_out) if log_ratio_out is not None else None
        a.jump = C.pointer(jump) if jump is not None else None
        with run.timed('mala_steps'):
            hip.check(hip.lib().nfmc_mala_steps_f32(C.byref(a), hip.stream()), 'nfmc_mala_steps_f32')

    def propose(self, x):
        """langevin.py:61-122 for an arbitrary target: autograd U/grad U + HIP proposal and log-ratio.
        Returns (x_prime, mask, n_calls, n_grads, n_divergences); mask is evaluated lazily by the
        accept-select kernel when called from `sample()`."""
        dev = hip.require_gpu()
        n = x.shape[0]
        xf = x.detach().to(dev, torch.float32).reshape(n, -1).contiguous()
        d = xf.shape[1]
        run = getattr(self, '_cur_run', None)
        step = getattr(self, '_cur_step', 0)
        seed = run.seed if run is not None else (self.seed or 0)
        off = run.chain_offset if run is not None else 0
        h = float(self.kernel.step_size)
        imd = imd_tensor(self.kernel, dev)
        lib = hip.lib()
        nz = un = None
        if run is not None and run.replay is not None:   # the noise is drawn before the first target call (langevin.py:63)
            nz, _ = run.replay.take(1, with_uniforms=False)
        per = 2 * n if self.params.adjustment else n     # langevin.py:116-120
        try:
            u, g = _guarded(_value_and_grad, self.target, xf, self.event_shape)
        except TargetFailure:
            return self._rejected_step(xf, per, per)
        x_prime = torch.empty_like(xf)
        rng = hip.make_rng(seed, off, step, nz, None, rounds=run.rounds if run is not None else 0)
        hip.check(lib.nfmc_langevin_propose_f32(hip.ptr(xf), hip.ptr(g), hip.ptr(imd), h, n, d, C.byref(rng),
                                                hip.ptr(x_prime), hip.stream()), 'nfmc_langevin_propose_f32')
        n_calls = n_grads = n
        self._last_log_ratio = None
        self._last_uniforms = None
        mask = torch.ones(n, dtype=torch.bool, device=dev)
        if self.params.adjustment:
            try:
                up, gp = _guarded(_value_and_grad, self.target, x_prime, self.event_shape)
            except TargetFailure:
                return self._rejected_step(xf, per, per)
            if nz is not None:
                un = run.replay.take_uniforms(1)              # drawn after the second target call (langevin.py:106)
                self._last_uniforms = un[0].contiguous()
            lr = torch.empty(n, dtype=torch.float32, device=dev)
            hip.check(lib.nfmc_langevin_log_ratio_f32(hip.ptr(xf), hip.ptr(x_prime), hip.ptr(u), hip.ptr(up),
                                                      hip.ptr(g), hip.ptr(gp), hip.ptr(imd), h, n, d, hip.ptr(lr),
                                                      hip.stream()), 'nfmc_langevin_log_ratio_f32')
            self._last_log_ratio = lr
            n_calls += n
            n_grads += n
            if run is None:  # stand-alone use of the seam: evaluate the mask here
                unif = torch.empty(n, dtype=torch.float32, device=dev)
                hip.check(lib.nfmc_philox_uniforms_f32(C.byref(rng), hip.TAG_ACCEPT, n, hip.ptr(unif), hip.stream()),
                          'nfmc_philox_uniforms_f32')
                mask = torch.log(unif) < lr
        return x_prime, mask, n_calls, n_grads, 0


class MALA(Langevin):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.adjustment = True


class ULA(Langevin):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.adjustment = False


@dataclass
class MHKernel(MetropolisKernel):
    event_size: int

    def __repr__(self):
        return (f'log step: {math.log(self.step_size):.2f}, '
                f'mass norm: {torch.max(torch.abs(self.inv_mass_diag)):.2f}')


@dataclass
class MHParameters(MetropolisParameters):
    imd_adjustment: float = 1e-5

    def __post_init__(self):
        self.tune_step_size = False       # mh.py:23-25
        self.tune_inv_mass_diag = True


class MH(Langevin):
    """Random-walk Metropolis (nfmc/algorithms/sampling/mcmc/mh.py): x' = x + eps * inv_mass_diag.  Runs on the
    Langevin kernel with a random-walk proposal (NfmcMalaArgs.adjust bit 1)."""
    random_walk = True

    def __init__(self, event_shape, target, kernel: Optional[MHKernel] = None, params: Optional[MHParameters] = None):
        if kernel is None:
            kernel = MHKernel(event_size=int(torch.prod(torch.as_tensor(event_shape))))
        if params is None:
            params = MHParameters()
        MetropolisSampler.__init__(self, event_shape, target, kernel, params)

    @property
    def name(self):
        return "MH"

    def _counts(self, n, k):
        return (2 * n * k if self.params.adjustment else 0), 0   # mh.py:67-71

    def propose(self, x):
        """mh.py:44-73 for an arbitrary target (split path)."""
        dev = hip.require_gpu()
        n = x.shape[0]
        xf = x.detach().to(dev, torch.float32).reshape(n, -1).contiguous()
        d = xf.shape[1]
        run = getattr(self, '_cur_run', None)
        step = getattr(self, '_cur_step', 0)
        seed = run.seed if run is not None else (self.seed or 0)
        off = run.chain_offset if run is not None else 0
        replayed = run is not None and run.replay is not None
        if replayed:
            nz, _ = run.replay.take(1, with_uniforms=False)
            noise = nz[0]
        else:
            noise = torch.empty(n, d, dtype=torch.float32, device=dev)
            rng = hip.make_rng(seed, off, step, rounds=run.rounds if run is not None else 0)
            hip.check(hip.lib().nfmc_philox_normals_f32(C.byref(rng), hip.TAG_NOISE, n, d, hip.ptr(noise), hip.stream()),
                      'nfmc_philox_normals_f32')
        x_prime = (xf + noise * self.kernel.inv_mass_diag.to(dev, torch.float32)[None]).contiguous()
        self._last_log_ratio = None
        self._last_uniforms = None
        mask = torch.ones(n, dtype=torch.bool, device=dev)
        n_calls = 0
        if self.params.adjustment:
            try:
                with torch.no_grad():
                    lr = (_guarded(self.target, xf.reshape(n, *self.event_shape)).reshape(-1)
                          - _guarded(self.target, x_prime.reshape(n, *self.event_shape)).reshape(-1))
            except TargetFailure:
                return self._rejected_step(xf, 2 * n, 0)      # mh.py:63-71
            self._last_log_ratio = lr.float().contiguous()
            if replayed:
                self._last_uniforms = run.replay.take_uniforms(1)[0].contiguous()   # mh.py:59, after both target calls
            n_calls = 2 * n
            if run is None:
                unif = torch.empty(n, dtype=torch.float32, device=dev)
                rng = hip.make_rng(seed, off, step, rounds=run.rounds if run is not None else 0)
                hip.check(hip.lib().nfmc_philox_uniforms_f32(C.byref(rng), hip.TAG_ACCEPT, n, hip.ptr(unif), hip.stream()),
                          'nfmc_philox_uniforms_f32')
                mask = torch.log(unif) < self._last_log_ratio
        return x_prime, mask, n_calls, 0, 0


class RandomWalk(MH):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.adjustment = False


class HMC(MetropolisSampler):
    def __init__(self, event_shape, target, kernel: Optional[HMCKernel] = None,
                 params: Optional[HMCParameters] = None):
        if kernel is None:
            kernel = HMCKernel(event_size=int(torch.prod(torch.as_tensor(event_shape))))
        if params is None:
            params = HMCParameters()
        super().__init__(event_shape, target, kernel, params)

    @property
    def name(self):
        return "HMC"

    def _counts(self, n, k):
        grads = 2 * self.kernel.n_leapfrog_steps * n  # hmc.py:122-125
        calls = grads + (2 * n if self.params.adjustment else 0)
        return calls * k, grads * k

    def _launch(self, run, pot, k, step0, samples, masks_out=None, log_ratio_out=None, jump=None, rng=None, tune=None):
        a = hip.NfmcHmcArgs()
        a.x, a.n, a.d, a.n_steps = hip.ptr(run.x), run.n, run.d, k
        a.step_size = float(self.kernel.step_size)
        a.n_leapfrog = int(self.kernel.n_leapfrog_steps)
        a.adjust = 1 if self.params.adjustment else 0
        imd = tune.imd if tune is not None else imd_tensor(self.kernel, run.dev)
        a.inv_mass_diag = hip.ptr(imd)
        a.pot = pot.descriptor(run.dev)
        # `rng`: an NfmcRng prepared by the caller (replay bookkeeping of fused jump tails)
        a.rng = rng if rng is not None else run.rng(step0, k, adjusted=self.params.adjustment)
        if tune is not None:   # the controller rides on the per-call statistics fold
            a.stats = run.stats.struct()
            a.tune = tune.struct(getattr(tune, 'every', 0), k)
        else:
            a.stats = run.stats.struct(defer=True, attempted=run.n * k, jump_attempted=run.n if jump is not None else 0)
        a.samples = hip.store_struct(samples, k + (1 if jump is not None else 0))
        a.masks_out = hip.ptr(masks_out, torch.uint8) if masks_out is not None else None
        a.log_ratio_out = hip.ptr(log_ratio_out) if log_ratio_out is not None else None
        a.jump = C.pointer(jump) if jump is not None else None
        with run.timed('hmc_steps'):
            hip.check(hip.lib().nfmc_hmc_steps_f32(C.byref(a), hip.stream()), 'nfmc_hmc_steps_f32')

    def propose(self, x):
        """hmc.py:96-126 for an arbitrary target: leapfrog with autograd gradients on the GPU."""
        dev = hip.require_gpu()
        n = x.shape[0]
        xf = x.detach().to(dev, torch.float32).reshape(n, -1).contiguous()
        d = xf.shape[1]
        run = getattr(self, '_cur_run', None)
        step = getattr(self, '_cur_step', 0)
        seed = run.seed if run is not None else (self.seed or 0)
        off = run.chain_offset if run is not None else 0
        h = float(self.kernel.step_size)
        m = self.kernel.inv_mass_diag.to(dev, torch.float32)
        lib = hip.lib()
        replayed = run is not None and run.replay is not None
        if replayed:
            nz, _ = run.replay.take(1, with_uniforms=False)
            noise = nz[0]
        else:
            noise = torch.empty(n, d, dtype=torch.float32, device=dev)
            rng = hip.make_rng(seed, off, step, rounds=run.rounds if run is not None else 0)
            hip.check(lib.nfmc_philox_normals_f32(C.byref(rng), hip.TAG_NOISE, n, d, hip.ptr(noise), hip.stream()),
                      'nfmc_philox_normals_f32')
        p = noise * (1 / m.sqrt())
        p0 = p
        q = xf
        L = self.kernel.n_leapfrog_steps
        n_grads = 2 * L * n                                   # hmc.py:122-125
        n_calls = n_grads + (2 * n if self.params.adjustment else 0)
        # hmc.py:69,71 evaluates the gradient twice at every interior position (the second half step of one leapfrog, the
        # first of the next) and hmc.py:103-110 the target again at both ends: 2 L + 2 calls, in that order -- kept for a
        # user's callable, which may count or fail per call (tests/golden/hmc_fail_d5).  A caller that knows its target is a
        # deterministic function (NeuTra's adjusted target on the flow kernels) sets `one_evaluation_per_position`: L + 1
        # value-and-gradient calls return the same numbers (the fused kernels do the same; the reported n_calls / n_grads
        # stay the reference's formulas).
        merged = bool(getattr(self, 'one_evaluation_per_position', False))
        u0 = u1 = None
        try:
            if merged:
                u0, g = _guarded(_value_and_grad, self.target, q, self.event_shape)
                u1 = u0
                for _ in range(L):
                    p = p - h / 2 * g
                    q = q + h * (p * m)
                    u1, g = _guarded(_value_and_grad, self.target, q, self.event_shape)
                    p = p - h / 2 * g
            else:
                for _ in range(L):
                    p = p - h / 2 * _guarded(_value_and_grad, self.target, q, self.event_shape)[1]
                    q = q + h * (p * m)
                    p = p - h / 2 * _guarded(_value_and_grad, self.target, q, self.event_shape)[1]
        except TargetFailure:
            return self._rejected_step(xf, n_calls, n_grads)  # hmc.py:117-120
        n_calls = n_grads
        self._last_log_ratio = None
        self._last_uniforms = None
        mask = torch.ones(n, dtype=torch.bool, device=dev)
        if self.params.adjustment:
            try:
                with torch.no_grad():
                    if not merged:
                        u0 = _guarded(self.target, xf.reshape(n, *self.event_shape)).reshape(-1)
                        u1 = _guarded(self.target, q.reshape(n, *self.event_shape)).reshape(-1)
                    h0 = u0.reshape(-1) + 0.5 * (p0 ** 2 * m).sum(-1)
                    h1 = u1.reshape(-1) + 0.5 * (p ** 2 * m).sum(-1)
            except TargetFailure:
                return self._rejected_step(xf, n_grads + 2 * n, n_grads)
            self._last_log_ratio = (h0 - h1).float().contiguous()
            if replayed:
                self._last_uniforms = run.replay.take_uniforms(1)[0].contiguous()   # hmc.py:112, after both target calls
            n_calls += 2 * n
            if run is None:
                unif = torch.empty(n, dtype=torch.float32, device=dev)
                rng = hip.make_rng(seed, off, step, rounds=run.rounds if run is not None else 0)
                hip.check(lib.nfmc_philox_uniforms_f32(C.byref(rng), hip.TAG_ACCEPT, n, hip.ptr(unif), hip.stream()),
                          'nfmc_philox_uniforms_f32')
                mask = torch.log(unif) < self._last_log_ratio
        return q.contiguous(), mask, n_calls, n_grads, 0


class UHMC(HMC):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.adjustment = False
