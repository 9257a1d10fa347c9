"""Jump NFMC: K inner MCMC transitions, then one independent-MH "jump" proposed by the flow
(nfmc/algorithms/sampling/nfmc/jump.py).  On the device an outer iteration is two kinds of launch:
the fused inner kernel (K transitions, state in registers) and `nfmc_flow_mh_steps_f32`
(flow.sample + flow.log_prob + 2 target calls + MH test + masked update + moments in one kernel).
"""
import ctypes as C
import math
import os
import time
from copy import deepcopy
from dataclasses import dataclass
from typing import Optional

import torch
from tqdm import tqdm

from .. import hip
from ..containers import (DeviceSampleStore, MCMCKernel, MCMCOutput, MCMCParameters, MCMCStatistics, NFMCKernel,
                          NFMCParameters, Sampler)
from ..flows import Flow, RealNVP
from ..tuning import train_val_split
from ..util import metropolis_acceptance_log_ratio
from .common import Run, chunks, progress, resolve_target
from .mcmc import HMC, MALA, MH, UHMC, ULA, TargetFailure, _guarded


@dataclass
class JumpNFMCParameters(NFMCParameters):
    adjusted_jumps: bool = True
    fit_nf: bool = False
    warmup_fit_kwargs: dict = None
    n_jumps_before_training: int = 10

    def __post_init__(self):
        super().__post_init__()
        if self.warmup_fit_kwargs is None:
            self.warmup_fit_kwargs = {
                'early_stopping': True,
                'early_stopping_threshold': 50,
                'keep_best_weights': True,
                'n_samples': 1,
                'n_epochs': 500,
                'lr': 0.05
            }


@dataclass
class JumpNFMCStatistics(MCMCStatistics):
    n_accepted_jumps: int = 0
    n_attempted_jumps: int = 0

    COUNTERS = MCMCStatistics.COUNTERS + ('n_accepted_jumps', 'n_attempted_jumps')
    RATES = MCMCStatistics.RATES + ('jump_acceptance_rate',)

    @property
    def jump_acceptance_rate(self):
        return self.n_accepted_jumps / self.n_attempted_jumps if self.n_attempted_jumps else math.nan

    def _summary(self):
        rows = super()._summary()
        return [('MCMC acc-rate', rows[0][1]), ('Jump acc-rate', '%.2f' % self.jump_acceptance_rate)] + rows[1:]


class JumpNFMCOutput(MCMCOutput):
    def __init__(self, event_shape, *args, **kwargs):
        kwargs['statistics'] = JumpNFMCStatistics(event_shape)
        super().__init__(event_shape, *args, **kwargs)


def flow_is_native(flow) -> bool:
    """True when the flow is the build's RealNVP within the fused kernels' limits."""
    bij = getattr(flow, 'bijection', None)
    if not isinstance(bij, RealNVP):
        return False
    return not bij.beyond_kernels()


def flow_fits_jump_tail(flow) -> bool:
    """The jump can ride at the end of the inner sampler's launch (NfmcJumpTail): narrow conditioner, d <= 512."""
    bij = getattr(flow, 'bijection', None)
    return (isinstance(bij, RealNVP) and bij.n_hidden <= 8 and bij.d <= 512 and bij.n_hidden_layers <= 4
            and bij.n_bins == 0)


def make_jump_tail(run: Run, flow, adjusted):
    """NfmcJumpTail for the transition right after an inner launch (+ keep-alive references)."""
    st, keep = flow.bijection.packed(run.dev)
    t = hip.NfmcJumpTail()
    t.flow = st
    t.adjusted = 1 if adjusted else 0
    t.counters = hip.ptr(run.stats._jump_counters, torch.int64)
    t._keep = [keep]
    return t


def flow_mh_supported(run: Run, flow, pot, logq, adjusted=True) -> bool:
    """Whether nfmc_flow_mh_steps_f32 has a kernel for this flow / shape (nfmc_flow_mh_supported_f32).  When it has
    not (e.g. ragged d > ~300: neither the weight image nor the wave tiles fit the LDS) the samplers compose the
    transition from the flow's own kernels (split_flow_mh), like for a foreign flow object."""
    a, _keep = _flow_mh_probe_args(run, flow, pot, logq, adjusted)
    return _supported(int(hip.lib().nfmc_flow_mh_supported_f32(C.byref(a))), 'nfmc_flow_mh_supported_f32')


def launch_flow_mh(run: Run, flow, pot, logq, k, step0, cached, adjusted, stats_struct, samples=None,
                   masks_out=None, log_ratio_out=None):
    a = hip.NfmcFlowMhArgs()
    st, _keep = flow.bijection.packed(run.dev)
    a.x, a.logq, a.n, a.n_steps = hip.ptr(run.x), hip.ptr(logq), run.n, k
    a.logq_cached = 1 if cached else 0
    a.adjusted = 1 if adjusted else 0
    a.flow = st
    a.pot = pot.descriptor(run.dev)
    a.rng = run.rng(step0, k, adjusted=adjusted)
    a.stats = stats_struct
    a.samples = hip.store_struct(samples, k)
    a.masks_out = hip.ptr(masks_out, torch.uint8) if masks_out is not None else None
    a.log_ratio_out = hip.ptr(log_ratio_out) if log_ratio_out is not None else None
    with run.timed('flow_mh_steps'):
        hip.check(hip.lib().nfmc_flow_mh_steps_f32(C.byref(a), hip.stream()), 'nfmc_flow_mh_steps_f32')


def _flow_mh_probe_args(run: Run, flow, pot, logq, adjusted):
    a = hip.NfmcFlowMhArgs()
    st, _keep = flow.bijection.packed(run.dev)
    a.x, a.logq, a.n, a.n_steps = hip.ptr(run.x), hip.ptr(logq), run.n, 1
    a.adjusted = 1 if adjusted else 0
    a.flow = st
    a.pot = pot.descriptor(run.dev)
    a.rng = hip.make_rng(run.seed, run.chain_offset, 0, rounds=run.rounds)
    a.stats = hip.null_stats()
    return a, _keep


def _supported(rc, what) -> bool:
    if rc in (hip.EUNSUPPORTED, hip.ESHAPE):   # a valid request no fused kernel covers (NfmcArgumentError.no_kernel)
        return False
    hip.check(rc, what)
    return True


def imh_parallel_ok(run: Run, flow, pot, logq) -> bool:
    """FixedIMH as a data-parallel problem (csrc/imh_parallel.hip): register-layout flows whose weight image fits the
    LDS (nfmc_imh_parallel_supported_f32 decides).  It wins most when the chains alone do not fill the GPU (sequential
    transitions are latency-bound there), and still by 10-15 % when they do.  NFMC_IMH_PARALLEL=0 turns it off."""
    bij = getattr(flow, 'bijection', None)
    if not flow_is_native(flow) or bij.n_hidden > 8 or bij.d > 512:   # affine and spline couplings (imh_parallel_rqs.hip)
        return False
    # d = 64, 1000 steps, parallel vs sequential: 1.0 vs 3.0 ms at n = 1000, 1.98 vs 3.0 at 8192, 5.7 vs 6.7 at 32768,
    # 10.6 vs 12.3 at 65536 (one accept-uniform draw per row instead of per lane, no per-step select / moments)
    if os.environ.get('NFMC_IMH_PARALLEL') == '0':
        return False
    a, _keep = _flow_mh_probe_args(run, flow, pot, logq, True)
    return _supported(int(hip.lib().nfmc_imh_parallel_supported_f32(C.byref(a))), 'nfmc_imh_parallel_supported_f32')


def launch_imh_parallel(run: Run, flow, pot, logq, k, step0, cached, stats_struct, samples=None, masks_out=None,
                        log_ratio_out=None):
    """k IMH transitions of every chain through nfmc_imh_parallel_f32 (same contract as launch_flow_mh)."""
    a = hip.NfmcFlowMhArgs()
    st, _keep = flow.bijection.packed(run.dev)
    a.x, a.logq, a.n, a.n_steps = hip.ptr(run.x), hip.ptr(logq), run.n, k
    a.logq_cached = 1 if cached else 0
    a.adjusted = 1
    a.flow = st
    a.pot = pot.descriptor(run.dev)
    a.rng = run.rng(step0, k, adjusted=True)
    a.stats = stats_struct
    a.samples = hip.store_struct(samples, k)
    a.masks_out = hip.ptr(masks_out, torch.uint8) if masks_out is not None else None
    a.log_ratio_out = hip.ptr(log_ratio_out) if log_ratio_out is not None else None
    nbytes = int(hip.lib().nfmc_imh_parallel_work_bytes(run.n, run.d, k))
    work = torch.empty(nbytes, dtype=torch.uint8, device=run.dev)
    with run.timed('imh_parallel'):
        hip.check(hip.lib().nfmc_imh_parallel_f32(C.byref(a), hip.ptr(work, torch.uint8), nbytes, hip.stream()),
                  'nfmc_imh_parallel_f32')
    return work   # kept alive by the caller until the stream has consumed it (torch's allocator is stream-ordered)


def split_flow_mh(run: Run, flow, target, event_shape, step, adjusted, stats_struct, logq=None):
    """One flow-proposal MH transition for an arbitrary target / foreign flow object
    (jump.py:205-231, imh.py:221-233): flow passes through the flow's own API, target calls in torch on
    the GPU, test + masked update + moments in nfmc_mh_accept_select_f32."""
    n, d = run.n, run.d
    with torch.no_grad():
        if flow_is_native(flow) or isinstance(flow, Flow):   # this package's flow: the run's own noise streams
            rng = hip.make_rng(run.seed, run.chain_offset, step, rounds=run.rounds)
            if run.replay is not None:
                nz, _ = run.replay.take(1, with_uniforms=False)
                x_prime, ld = flow.bijection.inverse(nz[0].reshape(n, *event_shape))
                zz = nz[0]
                f_xp = (-0.5 * (zz * zz).sum(-1) - 0.5 * d * 1.8378770664093453) - ld
                unif = None   # taken below, once the target calls have returned (jump.py:225)
            else:
                x_prime, f_xp = flow.sample(n, return_log_prob=True, rng=rng)
                unif = None
        else:
            x_prime, f_xp = flow.sample(n, return_log_prob=True)
            unif = None
        x_prime = x_prime.detach().to(run.dev, torch.float32).reshape(n, d).contiguous()
        f_xp = f_xp.detach().to(run.dev, torch.float32).contiguous()
        lr = None
        f_x = None
        target_calls = 0
        if adjusted:
            # jump.py:210-227 / imh.py:222-237: a ValueError from the target or from flow.log_prob rejects every chain
            # (no divergence is counted for a jump); the two target calls are booked once both have returned
            try:
                u_x = _guarded(target, run.x.reshape(n, *event_shape)).reshape(-1)
                u_xp = _guarded(target, x_prime.reshape(n, *event_shape)).reshape(-1)
                target_calls = 2 * n
                f_x = logq if logq is not None else _guarded(flow.log_prob, run.x.reshape(n, *event_shape))
                f_x = f_x.detach().to(run.dev, torch.float32).contiguous()
                lr = metropolis_acceptance_log_ratio(-u_x, -u_xp, f_x, f_xp).float().contiguous()
                if run.replay is not None and (flow_is_native(flow) or isinstance(flow, Flow)):
                    unif = run.replay.take_uniforms(1)[0].contiguous()
            except TargetFailure:
                lr = torch.full((n,), -math.inf, dtype=torch.float32, device=run.dev)
    st = hip.NfmcSelectArgs()
    st.x, st.x_prime, st.n, st.d = hip.ptr(run.x), hip.ptr(x_prime), n, d
    st.log_ratio = hip.ptr(lr) if lr is not None else None
    st.uniforms = hip.ptr(unif) if unif is not None else None
    st.n_carry = 0
    if logq is not None and adjusted:
        st.n_carry = 1
        st.carry[0] = hip.ptr(logq)
        st.carry_prime[0] = hip.ptr(f_xp)
    st.rng = hip.make_rng(run.seed, run.chain_offset, step, rounds=run.rounds)
    st.rng_tag = hip.TAG_JUMP
    st.stats = stats_struct
    st.mask_out = None
    hip.check(hip.lib().nfmc_mh_accept_select_f32(C.byref(st), hip.stream()), 'nfmc_mh_accept_select_f32')
    return target_calls


class JumpNFMC(Sampler):
    """Requires a flow with an efficient inverse (and forward, for adjusted jumps)."""

    # Run the jump as the tail of the last inner launch (NfmcJumpTail) instead of its own kernel.  Correct and
    # tested, but off by default: the flow code raises the fused kernel's VGPR allocation from 91 to ~200
    # (occupancy 5 -> 2 waves/SIMD), which costs the 100 inner transitions more than the separate 58 us jump
    # launch does (measured 0.474 vs 0.45 ms per outer iteration at C3).
    fuse_jump_tail = False

    def __init__(self, event_shape, target, inner_sampler: Sampler, kernel: NFMCKernel = None,
                 params: JumpNFMCParameters = None):
        if kernel is None:
            kernel = NFMCKernel(event_shape)
        if params is None:
            params = JumpNFMCParameters()
        super().__init__(event_shape, target, kernel, params)
        self.inner_sampler = inner_sampler

    @property
    def name(self):
        return 'Jump MCMC'

    def _refit(self, flow, x_train, x_val):
        """jump.py:201 `self.kernel.flow.fit(x_train=..., x_val=..., **flow_fit_kwargs)`.  The build's own Flow takes
        `defer_check`: the run's kernels are enqueued and the divergence check (ValueError) is made when the next refit
        starts / when sampling ends, so that the next inner launch is queued behind the fit without a host round trip; a
        foreign flow object is called exactly as the reference calls it."""
        from ..flows import Flow
        if isinstance(flow, Flow):
            return flow.fit(x_train=x_train, x_val=x_val, **{'defer_check': True, **self.params.flow_fit_kwargs})
        flow.fit(x_train=x_train, x_val=x_val, **self.params.flow_fit_kwargs)
        return None

    def warmup(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """jump.py:104-154: tune the inner sampler, then MLE-fit the flow on its samples (rollback on ValueError)."""
        inner_limit = 0.7 * time_limit_seconds if time_limit_seconds is not None else None
        t0 = time.time()
        self.inner_sampler.params.store_samples = True
        self.inner_sampler.shard = self.shard
        warmup_output = self.inner_sampler.warmup(x0, show_progress=show_progress, time_limit_seconds=inner_limit)
        x_train, x_val = train_val_split(warmup_output.samples_device, train_pct=self.params.train_pct,
                                         max_train_size=self.params.max_train_size,
                                         max_val_size=self.params.max_val_size, shard=self.shard)
        flow_params = deepcopy(self.kernel.flow.state_dict())
        fit_limit = time_limit_seconds - (time.time() - t0) if time_limit_seconds is not None else None
        try:
            self.kernel.flow.fit(x_train=x_train, x_val=x_val,
                                 **{**self.params.flow_fit_kwargs,
                                    **dict(show_progress=show_progress, time_limit_seconds=fit_limit)})
        except ValueError:
            self.kernel.flow.load_state_dict(flow_params)
        return warmup_output

    def sample(self, x0, show_progress: bool = True, time_limit_seconds=None) -> MCMCOutput:
        """jump.py:156-246 on the device."""
        inner = self.inner_sampler
        if not inner.params.store_samples:
            raise ValueError("Inner sampler in jump HMC must store samples")
        run = Run(self, x0)
        n, d, event_shape = run.n, run.d, run.event_shape
        flow = self.kernel.flow
        T, K = int(self.params.n_iterations), int(inner.params.n_iterations)
        off = (False, 'never')
        pot = resolve_target(self.target, event_shape,
                             'never' if (self.fuse in off or getattr(inner, 'fuse', 'auto') in off) else 'auto', run.x)
        fused = pot is not None and flow_is_native(flow)
        tail_ok = (self.fuse_jump_tail and fused and flow_fits_jump_tail(flow) and not self.params.fit_nf
                   and isinstance(inner, (MALA, ULA, HMC, UHMC)))
        inner._cur_run = run
        inner._n_divergences = 0
        jump_target_calls = 0

        # kept states of the whole run, T * (K + 1) offered: thinning / max_samples applied on the device (f3).  With
        # fit_nf the inner states of one outer iteration are also needed as a dense block for the refit (fit_buf).
        store = DeviceSampleStore(n, d, run.dev, T * (K + 1), getattr(self.params, 'thinning', 1),
                                  getattr(self.params, 'max_samples', None)) if (self.params.store_samples and T > 0) else None
        fit_buf = torch.empty(K, n, d, dtype=torch.float32, device=run.dev) if self.params.fit_nf else None
        pending_fit = None     # the latest refit's deferred check (flow_training.PendingFit)
        logq = torch.empty(n, dtype=torch.float32, device=run.dev)
        # Can the jump run on the flow-MH kernels?  The answer needs the packed weights (host work, a small upload): asked
        # before the first launch only when the jump is to ride behind the inner kernel, otherwise after the first inner
        # launches are queued, so that the GPU already works while the host packs
        probed = not fused

        def probe():
            nonlocal fused, tail_ok, probed
            if not probed and not flow_mh_supported(run, flow, pot, logq, self.params.adjusted_jumps):
                fused = tail_ok = False   # the jump through the flow's own kernels (split_flow_mh)
            probed = True

        if tail_ok:
            probe()

        t0 = time.time()
        done = 0
        bar = progress(show_progress, range(T), desc='Jump MCMC')
        for i in bar:
            if run.time_is_up(t0, time_limit_seconds):
                break
            base = i * (K + 1)
            dense = fit_buf is not None   # inner states into the refit block (offered to the store afterwards), else straight into the store
            # ---- K inner transitions (jump.py:178); when the flow is narrow the jump rides at the end of
            # the last inner launch, on the same registers (NfmcJumpTail)
            tail_done = False
            if pot is not None:
                for off, k in chunks(K):
                    last = off + k == K
                    tail = None
                    if last and tail_ok:
                        tail = make_jump_tail(run, flow, self.params.adjusted_jumps)
                    view = fit_buf[off:off + k] if dense else store
                    if tail is not None and run.replay is not None:
                        # replay order: K inner fields, then the jump's latent + uniform
                        rng_inner = run.rng(base + off, k, adjusted=inner.params.adjustment)
                        lat, un = run.replay.take(1, with_uniforms=self.params.adjusted_jumps)
                        tail.replay_latent = hip.ptr(lat[0].contiguous())
                        tail.replay_uniform = hip.ptr(un[0].contiguous()) if un is not None else None
                        tail._keep += [lat, un]
                        inner._launch(run, pot, k, base + off, view, jump=tail, rng=rng_inner)
                    else:
                        inner._launch(run, pot, k, base + off, view, jump=tail)
                    tail_done = tail is not None
            else:
                for off in range(K):
                    inner._split_step(run, base + off, fit_buf[off:off + 1] if dense else store)
            probe()
            if dense and store is not None:
                store.add_dense(fit_buf)
            # ---- optional refit on this iteration's inner samples (jump.py:193-201)
            if self.params.fit_nf and i >= self.params.n_jumps_before_training:
                x_train, x_val = train_val_split(fit_buf.reshape(K, n, *event_shape),
                                                 train_pct=self.params.train_pct,
                                                 max_train_size=self.params.max_train_size,
                                                 max_val_size=self.params.max_val_size, shard=self.shard)
                if pending_fit is not None:
                    pending_fit.result()     # the previous refit: a diverged run raises here (jump.py:201 raises at once)
                pending_fit = self._refit(flow, x_train, x_val)
            # ---- the jump (jump.py:205-243)
            if tail_done:
                jump_target_calls += 2 * n if self.params.adjusted_jumps else 0
            elif fused:
                launch_flow_mh(run, flow, pot, logq, 1, base + K, False, self.params.adjusted_jumps,
                               run.stats.struct(defer=True, attempted=n, jump=True), store)
                jump_target_calls += 2 * n if self.params.adjusted_jumps else 0
            else:
                jump_target_calls += split_flow_mh(run, flow, self.target, event_shape, base + K,
                                                   self.params.adjusted_jumps, run.stats.struct(jump=True))
                if store is not None:
                    store.add_dense(run.x[None])
            done = i + 1
            if show_progress:
                run.sync()
                bar.set_postfix_str(f'acc {int(run.stats.counters[hip.CNT_ACCEPTED])}/'
                                    f'{int(run.stats.counters[hip.CNT_ATTEMPTED])}')
        # end of the call, ordered for the GPU: the copy of the final state and the statistics fold are enqueued right
        # behind the last kernel, and the one device-to-host copy of the totals is the only synchronisation (with a
        # synchronize first, then the fold, then host work, then the clone, the stream sat idle ~110 us per call)
        if pending_fit is not None:
            pending_fit.result()
        last_sample = run.x.reshape(n, *event_shape).clone()
        inner._cur_run = None
        sum_x, sum_x2, cnt, jc = run.stats.host_totals()
        calls, grads = inner._counts(n, K * done)
        out = JumpNFMCOutput(event_shape, store_samples=self.params.store_samples,
                             max_samples=getattr(self.params, 'max_samples', None))
        st = out.statistics
        st.update_counters(n_accepted_trajectories=int(cnt[hip.CNT_ACCEPTED]),
                           n_attempted_trajectories=int(cnt[hip.CNT_ATTEMPTED]),
                           n_divergences=inner._n_divergences,      # jump.py:183: the inner sampler's failed steps
                           n_target_calls=calls + jump_target_calls,
                           n_target_gradient_calls=grads,
                           n_accepted_jumps=int(jc[hip.CNT_ACCEPTED]), n_attempted_jumps=n * done)
        st.n_nonfinite_log_ratios = int(cnt[hip.CNT_NONFINITE]) + int(jc[hip.CNT_NONFINITE])
        st.absorb_device_sums(sum_x.reshape(event_shape), sum_x2.reshape(event_shape), n * done * (K + 1))
        if store is not None:
            out.running_samples.adopt_store(store, getattr(self.params, 'spill_to_host', False))
        out.running_samples.last_sample = last_sample
        st.update_elapsed_time(time.time() - t0)
        out.kernel = self.kernel
        out.kernel_events = run.kernel_events
        if run.shard is not None:
            run.shard.merge_statistics(st)
        return out


def _make(inner_cls):
    class _Jump(JumpNFMC):
        def __init__(self, event_shape, target, kernel: NFMCKernel = None, params: JumpNFMCParameters = None,
                     inner_kernel: MCMCKernel = None, inner_params: MCMCParameters = None):
            super().__init__(event_shape, target, inner_cls(event_shape, target, inner_kernel, inner_params),
                             kernel, params)

    return _Jump


class JumpHMC(_make(HMC)):
    pass


class JumpUHMC(_make(UHMC)):
    pass


class JumpMALA(_make(MALA)):
    pass


class JumpULA(_make(ULA)):
    pass


class JumpMH(_make(MH)):
    pass
