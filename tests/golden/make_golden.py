#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box: nothing at test time imports this script).

How the reference is driven: `import nfmc` fails only because the
third-party `torchflows` and `potentials` packages are absent.  The five names the
nfmc-owned modules touch at import time are registered in `sys.modules` first
(`torchflows.{Flow,RealNVP}`, `torchflows.flows.Flow`, `torchflows.utils.{sum_except_batch,
get_batch_shape}`, `potentials.base.Potential`); `sum_except_batch(x, event_shape)` is given
its documented meaning (sum over the event axes) because hmc.py:103-110 calls it.  The
reference's langevin/hmc/mcmc-base/jump/imh/neutra/tuning/util code then runs unmodified.
Where the reference needs a flow object (`NFMCKernel.flow` is duck-typed,
nfmc/algorithms/sampling/base.py:18-26) the build's own CPU RealNVP (oracle/flow.py) is
plugged in, so the fixtures pin the complete outer loops *given* that flow.

Every random draw of the reference run is recorded by wrapping torch.randn/randn_like/
rand/rand_like for the duration of the run, so the fixtures carry the exact noise the
reference consumed (the HIP kernels' replay mode and the oracle's ReplayNoise read it back).

A fixture is data only: inputs (x0, noise, uniforms, flow weights, scalars) and the
reference's outputs (samples, counters, moments).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class Flow:  # only used for isinstance checks in nfmc/sample.py
        pass

    class RealNVP:
        pass

    class Potential:
        pass

    def sum_except_batch(x, event_shape):
        return x.flatten(start_dim=x.dim() - len(event_shape)).sum(-1)

    def get_batch_shape(x, event_shape):
        return x.shape[:x.dim() - len(event_shape)]

    mod('torchflows', Flow=Flow, RealNVP=RealNVP)
    mod('torchflows.flows', Flow=Flow)
    mod('torchflows.utils', sum_except_batch=sum_except_batch, get_batch_shape=get_batch_shape)
    mod('potentials')
    mod('potentials.base', Potential=Potential)
    sys.path.insert(0, '/root/reference')


class DrawRecorder:
    """Records every torch.randn/randn_like (normals) and rand/rand_like (uniforms) call."""

    def __enter__(self):
        self.normals, self.uniforms = [], []
        self._orig = {k: getattr(torch, k) for k in ('randn', 'randn_like', 'rand', 'rand_like', 'randperm', 'randint')}
        self.perms, self.ints = [], []

        def wrap(name, store):
            fn = self._orig[name]

            def inner(*a, **k):
                v = fn(*a, **k)
                store.append(v.detach().clone())
                return v

            return inner

        torch.randn = wrap('randn', self.normals)
        torch.randn_like = wrap('randn_like', self.normals)
        torch.rand = wrap('rand', self.uniforms)
        torch.rand_like = wrap('rand_like', self.uniforms)
        torch.randperm = wrap('randperm', self.perms)
        torch.randint = wrap('randint', self.ints)
        return self

    def __exit__(self, *exc):
        for k, v in self._orig.items():
            setattr(torch, k, v)


def flow_arrays(flow):
    return {f'flow/{k}': v.detach().numpy().copy() for k, v in flow.state_dict().items()}


def out_arrays(out, jump=False):
    st = out.statistics
    d = {
        'exp/samples': out.samples.numpy(),
        'exp/first_moment': st.running_first_moment.detach().numpy(),
        'exp/second_moment': st.running_second_moment.detach().numpy(),
        'exp/counters': np.array([st.n_accepted_trajectories, st.n_attempted_trajectories, st.n_divergences,
                                  st.n_target_calls, st.n_target_gradient_calls], dtype=np.int64),
    }
    if jump:
        d['exp/jump_counters'] = np.array([st.n_accepted_jumps, st.n_attempted_jumps], dtype=np.int64)
    return d


ONLY = set(sys.argv[1:])   # `python make_golden.py NAME...` rewrites only those fixtures


def save(name, rec, **arrays):
    if ONLY and name not in ONLY:
        return
    if rec is not None:
        chain_u = [v for v in rec.uniforms if v.dim() > 0]
        host_u = [v for v in rec.uniforms if v.dim() == 0]      # scalar host-side draws (AdaptiveIMH)
        arrays['noise/normals'] = np.stack([v.numpy() for v in rec.normals]) if rec.normals else np.zeros((0,), np.float32)
        arrays['noise/uniforms'] = np.stack([v.numpy() for v in chain_u]) if chain_u else np.zeros((0,), np.float32)
        if host_u or rec.ints:
            arrays['noise/host_uniforms'] = np.array([float(v) for v in host_u], dtype=np.float64)
            arrays['noise/host_ints'] = np.array([int(v) for v in rec.ints], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in arrays.items() if not k.startswith('flow/')})


def main():
    _install_standins()
    from nfmc.algorithms.sampling.mcmc.langevin import MALA, ULA, LangevinKernel
    from nfmc.algorithms.sampling.mcmc.hmc import HMC, UHMC, HMCKernel, HMCParameters
    from nfmc.algorithms.sampling.mcmc.langevin import LangevinParameters
    from nfmc.algorithms.sampling.base import NFMCKernel
    from nfmc.algorithms.sampling.nfmc.jump import JumpMALA, JumpHMC, JumpNFMCParameters
    from nfmc.algorithms.sampling.nfmc.imh import FixedIMH, IMHKernel, IMHParameters
    from nfmc.algorithms.sampling.nfmc.neutra import NeuTraHMC, NeuTraKernel, NeuTraParameters
    from nfmc.algorithms.sampling.tuning import train_val_split, DualAveraging, DualAveragingParams
    from nfmc.util import metropolis_acceptance_log_ratio, parse_flow_string

    from oracle import flow as oflow
    from oracle import potentials as opot

    sumsq = opot.sum_squares

    # ---------------------------------------------------------------- MALA / ULA (a5, a6)
    for name, cls, d, n, k, imd in [('mala_d6', MALA, 6, 16, 8, None),
                                    ('mala_d7_mass', MALA, 7, 12, 6, 'ramp'),
                                    ('ula_d6', ULA, 6, 16, 5, None)]:
        torch.manual_seed(11)
        x0 = torch.randn(n, d)
        kern = LangevinKernel(event_size=d)
        if imd == 'ramp':
            kern.inv_mass_diag = torch.linspace(0.7, 1.4, d)
        s = cls((d,), sumsq, kern, LangevinParameters(n_iterations=k))
        with DrawRecorder() as rec:
            out = s.sample(x0.clone(), show_progress=False)
        save(name, rec, x0=x0.numpy(), step_size=np.float64(kern.step_size),
             inv_mass_diag=kern.inv_mass_diag.numpy(), **out_arrays(out))

    # funnel (non-separable) MALA
    torch.manual_seed(12)
    d, n, k = 5, 16, 6
    x0 = 0.5 * torch.randn(n, d)
    kern = LangevinKernel(event_size=d, step_size=0.05)
    s = MALA((d,), opot.funnel(3.0), kern, LangevinParameters(n_iterations=k))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    save('mala_funnel_d5', rec, x0=x0.numpy(), step_size=np.float64(kern.step_size),
         inv_mass_diag=kern.inv_mass_diag.numpy(), **out_arrays(out))

    # ---------------------------------------------------------------- the failure channel: a target that raises
    # ValueError on chosen calls (langevin.py:111-114, hmc.py:117-120, mh.py:63-66, jump.py:226-227)
    class FailingTarget:
        def __init__(self, base, fail_calls):
            self.base, self.fail_calls, self.calls = base, set(fail_calls), 0

        def __call__(self, x):
            self.calls += 1
            if self.calls in self.fail_calls:
                raise ValueError('target failed on call %d' % self.calls)
            return self.base(x)

    from nfmc.algorithms.sampling.mcmc.mh import MH as _MH, MHKernel as _MHKernel, MHParameters as _MHParameters
    for name, kind, d, n, k, fails in [('mala_fail_d5', 'mala', 5, 12, 6, (4, 9)), ('hmc_fail_d5', 'hmc', 5, 8, 4, (5,)),
                                       ('mh_fail_d5', 'mh', 5, 12, 5, (3,)), ('ula_fail_d5', 'ula', 5, 8, 4, (2,))]:
        torch.manual_seed(41)
        x0 = torch.randn(n, d)
        tgt = FailingTarget(sumsq, fails)
        extra = {}
        if kind in ('mala', 'ula'):
            kern = LangevinKernel(event_size=d)
            s = (MALA if kind == 'mala' else ULA)((d,), tgt, kern, LangevinParameters(n_iterations=k))
            extra['step_size'] = np.float64(kern.step_size)
        elif kind == 'hmc':
            kern = HMCKernel(event_size=d, n_leapfrog_steps=2, step_size=0.1)
            s = HMC((d,), tgt, kern, HMCParameters(n_iterations=k))
            extra.update(step_size=np.float64(0.1), n_leapfrog=np.int64(2))
        else:
            kern = _MHKernel(event_size=d)
            kern.inv_mass_diag = torch.linspace(0.2, 0.5, d)
            s = _MH((d,), tgt, kern, _MHParameters(n_iterations=k))
        with DrawRecorder() as rec:
            out = s.sample(x0.clone(), show_progress=False)
        assert out.statistics.n_divergences == len(fails), out.statistics.n_divergences
        save(name, rec, x0=x0.numpy(), inv_mass_diag=kern.inv_mass_diag.numpy(), fail_calls=np.array(fails, dtype=np.int64),
             **extra, **out_arrays(out))

    # ---------------------------------------------------------------- random-walk MH (f2)
    from nfmc.algorithms.sampling.mcmc.mh import MH, RandomWalk, MHKernel, MHParameters
    from nfmc.algorithms.sampling.nfmc.jump import JumpMH
    for name, cls, d, n, k, imd in [('mh_d5', MH, 5, 16, 8, 'small'), ('rw_d6', RandomWalk, 6, 8, 4, 'small')]:
        torch.manual_seed(31)
        x0 = torch.randn(n, d)
        kern = MHKernel(event_size=d)
        kern.inv_mass_diag = torch.linspace(0.2, 0.5, d)
        s = cls((d,), sumsq, kern, MHParameters(n_iterations=k))
        with DrawRecorder() as rec:
            out = s.sample(x0.clone(), show_progress=False)
        save(name, rec, x0=x0.numpy(), inv_mass_diag=kern.inv_mass_diag.numpy(), **out_arrays(out))

    # ---------------------------------------------------------------- HMC / UHMC (a7)
    for name, cls, d, n, k, L, h, imd in [('hmc_d5', HMC, 5, 8, 4, 3, 0.1, None),
                                          ('hmc_d6_mass', HMC, 6, 8, 4, 4, 0.08, 'ramp'),
                                          ('uhmc_d5', UHMC, 5, 8, 3, 3, 0.1, None)]:
        torch.manual_seed(13)
        x0 = torch.randn(n, d)
        kern = HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h)
        if imd == 'ramp':
            kern.inv_mass_diag = torch.linspace(0.6, 1.5, d)
        s = cls((d,), sumsq, kern, HMCParameters(n_iterations=k))
        with DrawRecorder() as rec:
            out = s.sample(x0.clone(), show_progress=False)
        save(name, rec, x0=x0.numpy(), step_size=np.float64(h), n_leapfrog=np.int64(L),
             inv_mass_diag=kern.inv_mass_diag.numpy(), **out_arrays(out))

    # ---------------------------------------------------------------- jump (a4) with the build's CPU flow
    def make_flow(d, seed, n_layers=2, ck=None, target_std=None):
        torch.manual_seed(seed)
        f = oflow.Flow(oflow.RealNVP((d,), n_layers=n_layers, conditioner_kwargs=ck))
        return oflow.perturb_(f, seed + 100, 0.3, target_std)

    d, n, T, K = 6, 12, 3, 4
    flow = make_flow(d, 21, target_std=0.7)
    torch.manual_seed(14)
    x0 = torch.randn(n, d)
    ik = LangevinKernel(event_size=d)
    s = JumpMALA((d,), sumsq, NFMCKernel((d,), flow=flow), JumpNFMCParameters(n_iterations=T),
                 ik, LangevinParameters(n_iterations=K))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    # normals: per outer iteration K inner (n,d) then one latent (n,d); uniforms: K inner + 1 jump
    save('jump_mala_d6', rec, x0=x0.numpy(), step_size=np.float64(ik.step_size), n_outer=np.int64(T),
         n_inner=np.int64(K), **flow_arrays(flow), **out_arrays(out, jump=True))

    # jump with a failing target: call 5 is an inner MALA call of outer iteration 0 (a divergence; the step makes no second call), call 7 the jump's
    # second target call of outer iteration 0 (all chains rejected, no divergence, its 2n calls not booked)
    d, n, T, K = 6, 10, 2, 3
    flow = make_flow(d, 28, target_std=0.7)
    torch.manual_seed(42)
    x0 = torch.randn(n, d)
    ik = LangevinKernel(event_size=d)
    tgt = FailingTarget(sumsq, (5, 7))
    s = JumpMALA((d,), tgt, NFMCKernel((d,), flow=flow), JumpNFMCParameters(n_iterations=T),
                 ik, LangevinParameters(n_iterations=K))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    assert out.statistics.n_divergences == 1
    save('jump_mala_fail_d6', rec, x0=x0.numpy(), step_size=np.float64(ik.step_size), n_outer=np.int64(T),
         n_inner=np.int64(K), fail_calls=np.array((5, 7), dtype=np.int64), **flow_arrays(flow), **out_arrays(out, jump=True))

    d, n, T, K, L = 8, 10, 2, 2, 3
    flow = make_flow(d, 22, n_layers=3, ck={'n_hidden': 5, 'n_layers': 3}, target_std=0.7)
    torch.manual_seed(15)
    x0 = torch.randn(n, d)
    ik = HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=0.1)
    s = JumpHMC((d,), sumsq, NFMCKernel((d,), flow=flow), JumpNFMCParameters(n_iterations=T),
                ik, HMCParameters(n_iterations=K))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    save('jump_hmc_d8', rec, x0=x0.numpy(), step_size=np.float64(0.1), n_leapfrog=np.int64(L),
         n_outer=np.int64(T), n_inner=np.int64(K), flow_n_layers=np.int64(3), flow_n_hidden=np.int64(5),
         flow_cond_layers=np.int64(3), **flow_arrays(flow), **out_arrays(out, jump=True))

    # ---------------------------------------------------------------- FixedIMH (a9)
    d, n, T = 6, 32, 6
    flow = make_flow(d, 23, target_std=0.7)
    torch.manual_seed(16)
    x0 = torch.randn(n, d)
    s = FixedIMH((d,), sumsq, IMHKernel((d,), flow=flow), IMHParameters(n_iterations=T))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    save('imh_d6', rec, x0=x0.numpy(), n_iterations=np.int64(T), **flow_arrays(flow), **out_arrays(out))

    # odd d (d_a != d_b) IMH
    d, n, T = 7, 16, 4
    flow = make_flow(d, 24, n_layers=3, target_std=0.7)
    torch.manual_seed(17)
    x0 = torch.randn(n, d)
    s = FixedIMH((d,), sumsq, IMHKernel((d,), flow=flow), IMHParameters(n_iterations=T))
    with DrawRecorder() as rec:
        out = s.sample(x0.clone(), show_progress=False)
    save('imh_d7_odd', rec, x0=x0.numpy(), n_iterations=np.int64(T), flow_n_layers=np.int64(3),
         **flow_arrays(flow), **out_arrays(out))

    # ---------------------------------------------------------------- AdaptiveIMH (f2); the flow's `fit` is the
    # build's own one-step AdamW refit (oracle/flow.py fit_), torchflows' is not available
    from nfmc.algorithms.sampling.nfmc.imh import AdaptiveIMH
    for name, dist, d, n, T in [('adaptive_imh_d6', 'uniform', 6, 24, 6), ('adaptive_imh_geom_d5', 'bounded_geom', 5, 16, 5)]:
        flow = make_flow(d, 26, target_std=0.7)
        w0 = flow_arrays(flow)
        torch.manual_seed(27)
        x0 = torch.randn(n, d)
        s = AdaptiveIMH((d,), sumsq, IMHKernel((d,), flow=flow),
                        IMHParameters(n_iterations=T, train_distribution=dist, adaptation_dropoff=0.8))
        with DrawRecorder() as rec:
            out = s.sample(x0.clone(), show_progress=False)
        save(name, rec, x0=x0.numpy(), n_iterations=np.int64(T), adaptation_dropoff=np.float64(0.8),
             **w0, **{k.replace('flow/', 'flow_final/'): v for k, v in flow_arrays(flow).items()}, **out_arrays(out))

    # ---------------------------------------------------------------- NeuTra HMC (a10)
    d, n, T, L = 6, 8, 3, 3
    flow = make_flow(d, 25, target_std=0.7)
    torch.manual_seed(18)
    z0 = torch.randn(n, d)
    ik = HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=0.05)
    s = NeuTraHMC((d,), sumsq, ik, HMCParameters(), NeuTraKernel((d,), flow=flow), NeuTraParameters(n_iterations=T))
    with DrawRecorder() as rec:
        out = s.sample(z0.clone(), show_progress=False)
    save('neutra_hmc_d6', rec, x0=z0.numpy(), step_size=np.float64(0.05), n_leapfrog=np.int64(L),
         n_iterations=np.int64(T), **flow_arrays(flow), **out_arrays(out))

    # ---------------------------------------------------------------- train_val_split (a12)
    torch.manual_seed(19)
    xs = torch.randn(5, 7, 3)
    with DrawRecorder() as rec:
        tr, va = train_val_split(xs, 0.7, 16, 4)
    save('train_val_split', None, x=xs.numpy(), perm=rec.perms[0].numpy(), train=tr.numpy(), val=va.numpy())

    # ---------------------------------------------------------------- DualAveraging + update_kernel (A.7)
    da = DualAveraging(0.25, DualAveragingParams())
    errs = np.linspace(-0.3, 0.4, 12)
    vals = []
    for e in errs:
        da.step(float(e))
        vals.append(da.value)
    torch.manual_seed(20)
    d, n, k = 5, 32, 6
    x0 = torch.randn(n, d)
    kern = LangevinKernel(event_size=d)
    s = MALA((d,), sumsq, kern, LangevinParameters(n_iterations=k, n_warmup_iterations=k))
    with DrawRecorder() as rec:
        wout = s.warmup(x0.clone(), show_progress=False)
    save('tuning', rec, da_errors=errs, da_values=np.array(vals), x0=x0.numpy(),
         step_size0=np.float64(d ** (-1 / 3)), tuned_step_size=np.float64(s.kernel.step_size),
         tuned_inv_mass_diag=s.kernel.inv_mass_diag.numpy(), **out_arrays(wout))

    # ---------------------------------------------------------------- util (a3, a8)
    lr = metropolis_acceptance_log_ratio(torch.tensor([1.0, -2.0]), torch.tensor([0.5, 3.0]),
                                         torch.tensor([0.25, 0.0]), torch.tensor([-1.0, 4.0]))
    p = parse_flow_string('realnvp%{"n_layers": 10, "conditioner_kwargs": {"n_layers": 5, "n_hidden": 100}}')
    save('util', None, log_ratio=lr.numpy(), parsed_n_layers=np.int64(p['kwargs']['n_layers']),
         parsed_n_hidden=np.int64(p['kwargs']['conditioner_kwargs']['n_hidden']))


if __name__ == '__main__':
    main()
