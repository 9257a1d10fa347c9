"""NeuTra HMC: HMC in the flow's latent space on U~(z) = U(f^-1(z)) - log|det J_{f^-1}(z)|
(nfmc/algorithms/sampling/nfmc/neutra.py:58-68,109-129).  The adjusted potential, its gradient (a
hand-written VJP through the coupling stack), the leapfrog integrator and the accept step run inside
`nfmc_neutra_hmc_steps_f32`.

Like the reference, the stored samples and the running moments are those of the LATENT state z: the
`data_transform` assigned at neutra.py:122 never reaches the moments;
`NeuTraParameters.transform_output=True` opts into x-space samples/moments instead.
"""
import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import Optional, Type

import torch
from tqdm import tqdm

from .. import hip
from ..containers import DeviceSampleStore, MCMCOutput, NFMCKernel, NFMCParameters, Sampler
from .common import Run, chunks, imd_tensor, progress, resolve_target
from .mcmc import (HMC, MH, HMCKernel, HMCParameters, MHKernel, MHParameters, MetropolisKernel, MetropolisParameters,
                   MetropolisSampler)


@dataclass
class NeuTraKernel(NFMCKernel):
    pass


@dataclass
class NeuTraParameters(NFMCParameters):
    batch_inverse_size: int = 128
    warmup_fit_kwargs: dict = None

    def __post_init__(self):
        super().__post_init__()
        if self.warmup_fit_kwargs is None:
            self.warmup_fit_kwargs = {
                'early_stopping': True,
                'early_stopping_threshold': 5000,
                'keep_best_weights': True,
                'n_samples': 1,
                'n_epochs': 50000,
                'lr': 0.05
            }


class NeuTra(Sampler):
    def __init__(self, event_shape, target, inner_sampler_class: Type[MetropolisSampler],
                 inner_kernel: MetropolisKernel, inner_params: MetropolisParameters,
                 kernel: NeuTraKernel = None, params: NeuTraParameters = None):
        if kernel is None:
            kernel = NeuTraKernel(event_shape)
        if params is None:
            params = NeuTraParameters()
        super().__init__(event_shape, target, kernel, params)
        inner_params.n_iterations = self.params.n_iterations
        self.inner_sampler = inner_sampler_class(event_shape, self.adjusted_target, inner_kernel, inner_params)
        self.inner_sampler.fuse = False  # adjusted_target is a Python closure; the fused route is sample() below
        self._grad_kernel_ok = True

    def adjusted_target(self, _z, return_data: bool = False):
        """neutra.py:58-68 through the flow's API (split path and external callers)."""
        n = _z.shape[0]
        dev = hip.require_gpu()
        grad_needed = torch.is_grad_enabled() and _z.requires_grad
        if grad_needed and self._grad_kernel_ok and self._closed_form() is not None and self._flow_on_kernels():
            try:
                return _AdjustedPotential.apply(_z, self)
            except hip.NfmcArgumentError as e:
                if not e.no_kernel:
                    raise
                self._grad_kernel_ok = False   # e.g. d > ~200 (three wave tiles exceed the LDS), H > 32 off the MFMA shapes, wide spline conditioners
        if grad_needed:
            # shapes without a reverse-sweep kernel: differentiate the torch restatement of the
            # flow (on the GPU; the target is the user's callable)
            from ..flow_training import inverse_torch
            flow = self.kernel.flow
            if flow.get_device() != dev:
                flow.to(dev)
            x, log_det_inverse = inverse_torch(flow.bijection, _z.to(dev, torch.float32))
            x = x.reshape(n, *self.event_shape)
            adjusted_potential = self.target(x).reshape(-1) - log_det_inverse.reshape(-1)
            return (adjusted_potential, x) if return_data else adjusted_potential
        x, log_det_inverse = self.kernel.flow.bijection.inverse(_z)
        log_prob = -self.target(x)
        adjusted_potential = -(log_prob.reshape(-1) + log_det_inverse.to(log_prob).reshape(-1))
        return (adjusted_potential, x) if return_data else adjusted_potential

    def _flow_on_kernels(self):
        bij = getattr(self.kernel.flow, 'bijection', None)
        return hasattr(bij, 'packed') and not bij.beyond_kernels()

    def _min_hidden(self):
        """Conditioner width to present to the kernels: d = 64 / 128 with one or two hidden layers runs on the
        matrix cores (csrc/neutra_mfma.hip) also when the flow's own conditioner is narrow (zero-padded)."""
        bij = self.kernel.flow.bijection
        ok = (getattr(bij, 'd', 0) in (64, 128) and getattr(bij, 'n_hidden_layers', 0) in (1, 2)
              and not getattr(bij, 'n_bins', 0) and os.environ.get('NFMC_NEUTRA_VALU', '0') != '1')
        return 64 if ok else 0

    def _closed_form(self):
        """The target as a closed-form potential descriptor (None: an arbitrary callable, differentiated by autograd)."""
        if getattr(self, '_pot_cache', None) is None or self._pot_cache[0] is not self.target:
            self._pot_cache = (self.target, resolve_target(self.target, self.event_shape, self.fuse))
        return self._pot_cache[1]

    def _potential_grad(self, z):
        """U~(z), grad U~(z) from nfmc_neutra_potential_grad_f32 (closed-form targets only)."""
        dev = hip.require_gpu()
        n = z.shape[0]
        pot = self._closed_form()
        if pot is None:
            raise ValueError('NeuTra needs a closed-form potential (nfmc_amd.potentials) for the gradient kernel')
        zf = z.detach().to(dev, torch.float32).reshape(n, -1).contiguous()
        st, _keep = self.kernel.flow.bijection.packed(dev, self._min_hidden())
        pd = pot.descriptor(dev)
        u = torch.empty(n, dtype=torch.float32, device=dev)
        g = torch.empty_like(zf)
        hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zf), n, hip.ptr(u),
                                                           hip.ptr(g), hip.stream()), 'nfmc_neutra_potential_grad_f32')
        return u, g.reshape(z.shape)

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """neutra.py:70-107: variational fit, then tune the inner sampler."""
        fit_limit = 0.3 * time_limit_seconds if time_limit_seconds is not None else None
        t0 = time.time()
        from .imh import _accepts_potential
        pot = resolve_target(self.target, tuple(x0.shape[1:]), getattr(self, 'fuse', 'auto'), x0)
        extra = {'potential': pot} if pot is not None and _accepts_potential(self.kernel.flow) else {}
        self.kernel.flow.variational_fit(lambda v: -self.target(v),
                                         **{**dict(time_limit_seconds=fit_limit), **self.params.warmup_fit_kwargs},
                                         show_progress=show_progress, **extra)
        inner_limit = time_limit_seconds - (time.time() - t0) if time_limit_seconds is not None else None
        self.inner_sampler.params.tuning_mode()
        self.inner_sampler.params.store_samples = self.params.store_samples
        self.inner_sampler.params.n_warmup_iterations = self.params.n_warmup_iterations
        return self.inner_sampler.warmup(x0, show_progress=show_progress, time_limit_seconds=inner_limit)

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """neutra.py:109-129 on the device."""
        inner = self.inner_sampler
        inner.params.n_iterations = self.params.n_iterations
        inner.params.sampling_mode()
        inner.params.store_samples = self.params.store_samples
        inner.params.thinning = getattr(self.params, 'thinning', 1)
        inner.params.max_samples = getattr(self.params, 'max_samples', None)
        run = Run(self, x0)
        n, d, event_shape = run.n, run.d, run.event_shape
        pot = resolve_target(self.target, event_shape, self.fuse)
        def split():
            # NeuTraMH / arbitrary targets / shapes without a fused kernel: the inner sampler's split path on the
            # adjusted target (neutra.py:116-127)
            inner.seed, inner.shard, inner.replay = self.seed, self.shard, self.replay
            inner.rng_rounds = self.rng_rounds
            # a closed-form target on the gradient kernel is a deterministic function of z: the split path's HMC evaluates
            # the adjusted target once per position (a user's callable keeps the reference's 2 L + 2 calls per trajectory)
            inner.one_evaluation_per_position = pot is not None and self._grad_kernel_ok and self._flow_on_kernels()
            out = inner.sample(x0, show_progress=show_progress, time_limit_seconds=time_limit_seconds)
            out.kernel.flow = self.kernel.flow
            return out

        if not isinstance(inner, HMC) or pot is None or not self._flow_on_kernels():
            return split()
        out = MCMCOutput(event_shape, store_samples=self.params.store_samples,
                         max_samples=getattr(self.params, 'max_samples', None))
        T = int(self.params.n_iterations)
        store = DeviceSampleStore(n, d, run.dev, T, getattr(self.params, 'thinning', 1),
                                  getattr(self.params, 'max_samples', None)) if (self.params.store_samples and T > 0) else None
        st_flow, _keep = self.kernel.flow.bijection.packed(run.dev, self._min_hidden())
        imd = imd_tensor(inner.kernel, run.dev)
        bij = self.kernel.flow.bijection
        sbytes = int(hip.lib().nfmc_neutra_scratch_bytes(n, d, max(bij.n_hidden, self._min_hidden()), int(st_flow.n_hidden_layers),
                                                         int(st_flow.n_coupling)))
        scratch = torch.empty(max(sbytes // 4, 1), dtype=torch.float32, device=run.dev)
        if os.environ.get('NFMC_KEEP_SCRATCH'):   # diagnostics only (tools/trace_c4.py reads the marks of a trace build from its tail)
            self._scratch = scratch
        t0 = time.time()
        done = 0
        limit = hip.MAX_STEPS_PER_CALL if (time_limit_seconds is None and not show_progress) else 4
        bar = progress(show_progress, total=T, desc='NeuTra HMC')
        while done < T:
            if run.time_is_up(t0, time_limit_seconds):
                break
            k = min(limit, T - done)
            a = hip.NfmcNeutraHmcArgs()
            a.z, a.n, a.n_steps = hip.ptr(run.x), n, k
            a.n_leapfrog = int(inner.kernel.n_leapfrog_steps)
            a.step_size = float(inner.kernel.step_size)
            a.adjust = 1 if inner.params.adjustment else 0
            a.inv_mass_diag = hip.ptr(imd)
            a.flow = st_flow
            a.pot = pot.descriptor(run.dev)
            a.rng = run.rng(done, k, adjusted=inner.params.adjustment)
            a.stats = run.stats.struct()
            seen_before = store.seen if store is not None else 0
            a.samples = hip.store_struct(store, k)
            a.scratch, a.scratch_bytes = hip.ptr(scratch), sbytes
            try:
                with run.timed('neutra_hmc_steps'):
                    hip.check(hip.lib().nfmc_neutra_hmc_steps_f32(C.byref(a), hip.stream()), 'nfmc_neutra_hmc_steps_f32')
            except hip.NfmcArgumentError as e:
                if done == 0 and e.no_kernel and run.rounds == 10:   # validation precedes every launch: nothing has run yet
                    bar.close()
                    return split()
                if store is not None:
                    store.seen = seen_before
                raise
            done += k
            bar.update(k)
        bar.close()
        # the final-state copy and the statistics fold go out right behind the last kernel; the one device-to-host
        # copy of the totals is the only synchronisation of the call
        last_sample = run.x.reshape(n, *event_shape).clone()
        sum_x, sum_x2, cnt, _jc = run.stats.host_totals()
        calls, grads = inner._counts(n, done)
        st = out.statistics
        st.update_counters(n_target_calls=calls, n_target_gradient_calls=grads,
                           n_accepted_trajectories=int(cnt[hip.CNT_ACCEPTED]),
                           n_attempted_trajectories=int(cnt[hip.CNT_ATTEMPTED]))
        st.n_nonfinite_log_ratios = int(cnt[hip.CNT_NONFINITE])
        st.absorb_device_sums(sum_x.reshape(event_shape), sum_x2.reshape(event_shape), n * done)
        if store is not None:
            out.running_samples.adopt_store(store, getattr(self.params, 'spill_to_host', False))
        out.running_samples.last_sample = last_sample
        st.update_elapsed_time(time.time() - t0)
        out.kernel = inner.kernel
        out.kernel.flow = self.kernel.flow  # neutra.py:128
        out.kernel_events = run.kernel_events
        if run.shard is not None:
            run.shard.merge_statistics(st)
        return out


class _AdjustedPotential(torch.autograd.Function):
    """Autograd view of the HIP adjusted-potential kernel, so `torch.autograd.grad(adjusted_target(z))`
    (hmc.py:40-48) works for external callers."""

    @staticmethod
    def forward(ctx, z, sampler):
        u, g = sampler._potential_grad(z)
        ctx.save_for_backward(g)
        return u

    @staticmethod
    def backward(ctx, grad_out):
        g, = ctx.saved_tensors
        return grad_out.reshape(-1, *([1] * (g.dim() - 1))) * g, None


class NeuTraHMC(NeuTra):
    def __init__(self, event_shape, target, inner_kernel: HMCKernel = None, inner_params: HMCParameters = None,
                 kernel: NeuTraKernel = None, params: NeuTraParameters = None):
        if inner_kernel is None:
            inner_kernel = HMCKernel(event_size=int(torch.prod(torch.as_tensor(event_shape))))
        if inner_params is None:
            inner_params = HMCParameters()
        super().__init__(event_shape, target, HMC, inner_kernel, inner_params, kernel, params)


class NeuTraMH(NeuTra):
    """neutra.py:147-159: random-walk MH in latent space (the adjusted potential comes from the HIP kernel,
    the proposal/test through the inner sampler's split path)."""

    def __init__(self, event_shape, target, inner_kernel: MHKernel = None, inner_params: MHParameters = None,
                 kernel: NeuTraKernel = None, params: NeuTraParameters = None):
        if inner_kernel is None:
            inner_kernel = MHKernel(event_size=int(torch.prod(torch.as_tensor(event_shape))))
        if inner_params is None:
            inner_params = MHParameters()
        super().__init__(event_shape, target, MH, inner_kernel, inner_params, kernel, params)
