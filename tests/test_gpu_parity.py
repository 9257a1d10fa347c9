"""GPU parity: the HIP path (through the C ABI, via the nfmc_amd samplers) against
  (1) the golden vectors recorded from the reference itself (tests/golden), in replay mode, and
  (2) the CPU oracle on seeded inputs, in replay and in native-Philox mode.

Tolerances (fp32 path; kernels contract to FMA and use the gfx950 hardware log/exp/sin/cos):
  states / moments   atol 3e-5 on O(1) values
  log-ratios         atol 2e-4 (sums of d terms of O(1..30))
  counters           exact, except chains whose |log u - log ratio| < 1e-4 (accept decision ill-conditioned)
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_flow, FailingTarget

pytestmark = pytest.mark.gpu

ATOL = 3e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda', 0)


def _noise(fx):
    nz = fx['noise/normals'] if fx['noise/normals'].ndim > 1 else None
    un = fx['noise/uniforms'] if fx['noise/uniforms'].ndim > 1 else None
    return nz, un


def _check_out(out, fx, jump=False, atol=ATOL):
    np.testing.assert_allclose(out.samples.numpy().reshape(fx['exp/samples'].shape), fx['exp/samples'], atol=atol, rtol=0)
    np.testing.assert_allclose(out.statistics.running_first_moment.numpy(), fx['exp/first_moment'], atol=atol, rtol=0)
    np.testing.assert_allclose(out.statistics.running_second_moment.numpy(), fx['exp/second_moment'], atol=atol, rtol=0)
    c = fx['exp/counters']
    st = out.statistics
    assert (st.n_accepted_trajectories, st.n_attempted_trajectories, st.n_divergences, st.n_target_calls,
            st.n_target_gradient_calls) == tuple(int(v) for v in c)
    if jump:
        assert (st.n_accepted_jumps, st.n_attempted_jumps) == tuple(int(v) for v in fx['exp/jump_counters'])


def _amd_flow(fx, d, n_layers=2, n_hidden=None, cond_layers=2):
    from nfmc_amd.flows import Flow, RealNVP
    ck = {'n_layers': cond_layers}
    if n_hidden is not None:
        ck['n_hidden'] = n_hidden
    f = Flow(RealNVP((d,), n_layers=n_layers, conditioner_kwargs=ck))
    f.load_state_dict({k[len('flow/'):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('flow/')})
    return f


# ------------------------------------------------------------------------------------------ native streams
def test_philox_streams_match_oracle(dev):
    from nfmc_amd import hip
    from oracle import philox
    n, d, seed, step, off = 1000, 37, 0x1234567890ABCDEF, 77, 5000
    rng = hip.make_rng(seed, off, step)
    out = torch.empty(n, d, device=dev)
    for tag in (philox.TAG_NOISE, philox.TAG_LATENT):
        hip.check(hip.lib().nfmc_philox_normals_f32(C.byref(rng), tag, n, d, hip.ptr(out), hip.stream()), 'normals')
        want = philox.normal_field(seed, np.arange(off, off + n), step, d, tag)
        np.testing.assert_allclose(out.cpu().numpy(), want, atol=4e-6, rtol=0)
    u = torch.empty(n, device=dev)
    hip.check(hip.lib().nfmc_philox_uniforms_f32(C.byref(rng), philox.TAG_ACCEPT, n, hip.ptr(u), hip.stream()), 'unif')
    np.testing.assert_array_equal(u.cpu().numpy(), philox.accept_uniform(seed, np.arange(off, off + n), step))
    hip.check(hip.lib().nfmc_philox_uniforms_f32(C.byref(rng), philox.TAG_JUMP, n, hip.ptr(u), hip.stream()), 'unif')
    np.testing.assert_array_equal(u.cpu().numpy(), philox.jump_uniform(seed, np.arange(off, off + n), step))


# ------------------------------------------------------------------------------------------ golden: inner samplers
@pytest.mark.parametrize('name,cls,pot', [('mala_d6', 'MALA', 'sumsq'), ('mala_d7_mass', 'MALA', 'sumsq'),
                                          ('ula_d6', 'ULA', 'sumsq'), ('mala_funnel_d5', 'MALA', 'funnel')])
def test_langevin_golden(dev, name, cls, pot):
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares, Funnel
    fx = load_golden(name)
    d = fx['x0'].shape[1]
    target = SumOfSquares((d,)) if pot == 'sumsq' else Funnel((d,), 3.0)
    kern = mcmc.LangevinKernel(event_size=d, step_size=float(fx['step_size']),
                               inv_mass_diag=torch.from_numpy(fx['inv_mass_diag']))
    s = getattr(mcmc, cls)((d,), target, kern, mcmc.LangevinParameters(n_iterations=fx['exp/samples'].shape[0]))
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


@pytest.mark.parametrize('name,cls', [('hmc_d5', 'HMC'), ('hmc_d6_mass', 'HMC'), ('uhmc_d5', 'UHMC')])
def test_hmc_golden(dev, name, cls):
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden(name)
    d = fx['x0'].shape[1]
    kern = mcmc.HMCKernel(event_size=d, step_size=float(fx['step_size']), n_leapfrog_steps=int(fx['n_leapfrog']),
                          inv_mass_diag=torch.from_numpy(fx['inv_mass_diag']))
    s = getattr(mcmc, cls)((d,), SumOfSquares((d,)), kern, mcmc.HMCParameters(n_iterations=fx['exp/samples'].shape[0]))
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


@pytest.mark.parametrize('name,cls,fuse', [('mh_d5', 'MH', 'auto'), ('rw_d6', 'RandomWalk', 'auto'), ('mh_d5', 'MH', False)])
def test_random_walk_mh_golden(dev, name, cls, fuse):
    """mh.py:44-73 on the Langevin kernel with a random-walk proposal (fused) and through propose() (split)."""
    from nfmc_amd.samplers import mcmc
    fx = load_golden(name)
    d = fx['x0'].shape[1]
    kern = mcmc.MHKernel(event_size=d, inv_mass_diag=torch.from_numpy(fx['inv_mass_diag']))
    s = getattr(mcmc, cls)((d,), lambda x: torch.sum(x ** 2, dim=-1), kern, mcmc.MHParameters(n_iterations=fx['exp/samples'].shape[0]))
    if cls == 'RandomWalk':
        s.params.adjustment = False
    s.fuse = fuse
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


def test_langevin_golden_python_callable_split_path(dev):
    """Arbitrary Python target (fuse disabled): autograd U/grad U + HIP proposal/log-ratio/select kernels."""
    from nfmc_amd.samplers import mcmc
    fx = load_golden('mala_d7_mass')
    d = fx['x0'].shape[1]
    kern = mcmc.LangevinKernel(event_size=d, step_size=float(fx['step_size']),
                               inv_mass_diag=torch.from_numpy(fx['inv_mass_diag']))
    s = mcmc.MALA((d,), lambda x: torch.sum(x ** 2, dim=-1), kern,
                  mcmc.LangevinParameters(n_iterations=fx['exp/samples'].shape[0]))
    s.fuse = False
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


def test_hmc_golden_python_callable_split_path(dev):
    from nfmc_amd.samplers import mcmc
    fx = load_golden('hmc_d6_mass')
    d = fx['x0'].shape[1]
    kern = mcmc.HMCKernel(event_size=d, step_size=float(fx['step_size']), n_leapfrog_steps=int(fx['n_leapfrog']),
                          inv_mass_diag=torch.from_numpy(fx['inv_mass_diag']))
    s = mcmc.HMC((d,), lambda x: torch.sum(x ** 2, dim=-1), kern, mcmc.HMCParameters(n_iterations=fx['exp/samples'].shape[0]))
    s.fuse = False
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


@pytest.mark.parametrize('name,cls', [('mala_fail_d5', 'MALA'), ('ula_fail_d5', 'ULA'), ('hmc_fail_d5', 'HMC'), ('mh_fail_d5', 'MH')])
def test_target_failure_channel_golden(dev, name, cls):
    """The reference's only error channel on the propose() seam (langevin.py:111-114, hmc.py:117-120, mh.py:63-66): a
    ValueError raised by the target rejects every chain for that step, counts one divergence, and the run goes on.
    Fixtures recorded from the reference with a target that raises on chosen calls."""
    from nfmc_amd.samplers import mcmc
    fx = load_golden(name)
    d = fx['x0'].shape[1]
    k = fx['exp/samples'].shape[0]
    target = FailingTarget(lambda x: torch.sum(x ** 2, dim=-1), fx['fail_calls'])
    imd = torch.from_numpy(fx['inv_mass_diag'])
    if cls in ('MALA', 'ULA'):
        s = getattr(mcmc, cls)((d,), target, mcmc.LangevinKernel(event_size=d, step_size=float(fx['step_size']), inv_mass_diag=imd),
                               mcmc.LangevinParameters(n_iterations=k))
    elif cls == 'HMC':
        s = mcmc.HMC((d,), target, mcmc.HMCKernel(event_size=d, step_size=float(fx['step_size']),
                                                   n_leapfrog_steps=int(fx['n_leapfrog']), inv_mass_diag=imd),
                     mcmc.HMCParameters(n_iterations=k))
    else:
        s = mcmc.MH((d,), target, mcmc.MHKernel(event_size=d, inv_mass_diag=imd), mcmc.MHParameters(n_iterations=k))
    s.fuse = False          # probing would consume the target's call counter (and a failing target is not a quadratic)
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)
    assert out.statistics.n_divergences == len(fx['fail_calls'])


def test_jump_with_failing_target_golden(dev):
    """jump.py:183 (the inner sampler's divergences are carried into the jump statistics) and :226-227 (a jump whose
    target call raises rejects every chain, counts no divergence and books no target calls)."""
    from nfmc_amd.samplers import jump, mcmc
    from nfmc_amd.containers import NFMCKernel
    fx = load_golden('jump_mala_fail_d6')
    d = fx['x0'].shape[1]
    target = FailingTarget(lambda x: torch.sum(x ** 2, dim=-1), fx['fail_calls'])
    s = jump.JumpMALA((d,), target, NFMCKernel((d,), flow=_amd_flow(fx, d)),
                      jump.JumpNFMCParameters(n_iterations=int(fx['n_outer'])),
                      mcmc.LangevinKernel(event_size=d, step_size=float(fx['step_size'])),
                      mcmc.LangevinParameters(n_iterations=int(fx['n_inner'])))
    s.fuse = False
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx, jump=True)
    assert out.statistics.n_divergences == 1


def test_library_argument_errors_are_not_swallowed_by_the_failure_channel(dev):
    """NfmcArgumentError subclasses ValueError (the reference's error type) but must never be mistaken for a failing
    target: a sampler handed a bad shape still raises."""
    from nfmc_amd import hip
    from nfmc_amd.samplers import mcmc
    assert issubclass(hip.NfmcArgumentError, ValueError)

    def bad_target(x):
        raise hip.NfmcArgumentError('library error inside user code', hip.EINVAL)
    s = mcmc.MALA((4,), bad_target, None, mcmc.LangevinParameters(n_iterations=2))
    s.fuse = False
    with pytest.raises(hip.NfmcArgumentError):
        s.sample(torch.randn(8, 4), show_progress=False)


# ------------------------------------------------------------------------------------------ golden: flow samplers
@pytest.mark.parametrize('fuse_tail', [False, True])
def test_jump_mala_golden(dev, fuse_tail):
    from nfmc_amd.samplers import jump, mcmc
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden('jump_mala_d6')
    d = 6
    s = jump.JumpMALA((d,), SumOfSquares((d,)), NFMCKernel((d,), flow=_amd_flow(fx, d)),
                      jump.JumpNFMCParameters(n_iterations=int(fx['n_outer'])),
                      mcmc.LangevinKernel(event_size=d, step_size=float(fx['step_size'])),
                      mcmc.LangevinParameters(n_iterations=int(fx['n_inner'])))
    s.replay = _noise(fx)
    s.fuse_jump_tail = fuse_tail   # jump as the tail of the inner launch (NfmcJumpTail) or as its own kernel
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx, jump=True)


@pytest.mark.parametrize('fuse_tail', [False, True])
def test_jump_hmc_golden(dev, fuse_tail):
    from nfmc_amd.samplers import jump, mcmc
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden('jump_hmc_d8')
    d = 8
    flow = _amd_flow(fx, d, int(fx['flow_n_layers']), int(fx['flow_n_hidden']), int(fx['flow_cond_layers']))
    s = jump.JumpHMC((d,), SumOfSquares((d,)), NFMCKernel((d,), flow=flow),
                     jump.JumpNFMCParameters(n_iterations=int(fx['n_outer'])),
                     mcmc.HMCKernel(event_size=d, step_size=float(fx['step_size']), n_leapfrog_steps=int(fx['n_leapfrog'])),
                     mcmc.HMCParameters(n_iterations=int(fx['n_inner'])))
    s.replay = _noise(fx)
    s.fuse_jump_tail = fuse_tail
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx, jump=True)


@pytest.mark.parametrize('name,d,nl', [('imh_d6', 6, 2), ('imh_d7_odd', 7, 3)])
def test_imh_golden(dev, name, d, nl):
    from nfmc_amd.samplers import imh
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden(name)
    s = imh.FixedIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=_amd_flow(fx, d, nl)),
                     imh.IMHParameters(n_iterations=int(fx['n_iterations'])))
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)


@pytest.mark.parametrize('name,dist,d', [('adaptive_imh_d6', 'uniform', 6), ('adaptive_imh_geom_d5', 'bounded_geom', 5)])
def test_adaptive_imh_golden(dev, name, dist, d):
    """AdaptiveIMH (imh.py:82-181) vs the reference's own run: chain noise and host draws replayed; the one-epoch
    refits run on the GPU (AdamW through the differentiable flow), so weights agree to fp32 training noise:
    tolerance 5e-5 on samples / final weights, counters exact."""
    from nfmc_amd.samplers import imh
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden(name)
    flow = _amd_flow(fx, d, 2)
    s = imh.AdaptiveIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=flow),
                        imh.IMHParameters(n_iterations=int(fx['n_iterations']), train_distribution=dist,
                                          adaptation_dropoff=float(fx['adaptation_dropoff'])))
    s.replay = _noise(fx)
    s.host_draws = (fx['noise/host_uniforms'], fx['noise/host_ints'])
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    np.testing.assert_allclose(out.samples.numpy(), fx['exp/samples'], atol=5e-5, rtol=0)
    np.testing.assert_allclose(out.mean.numpy(), fx['exp/first_moment'], atol=5e-5, rtol=0)
    np.testing.assert_allclose(out.second_moment.numpy(), fx['exp/second_moment'], atol=5e-5, rtol=0)
    st = out.statistics
    c = fx['exp/counters']
    assert (st.n_accepted_trajectories, st.n_attempted_trajectories, st.n_target_calls,
            st.n_target_gradient_calls) == (c[0], c[1], c[3], c[4])
    assert s.n_refits > 0
    for k, v in flow.state_dict().items():
        np.testing.assert_allclose(v.detach().cpu().numpy(), fx['flow_final/' + k], atol=5e-5, rtol=0)


def test_adaptive_imh_strategy_string(dev):
    from nfmc_amd import sample
    torch.manual_seed(0)
    out = sample(lambda x: torch.sum(x ** 2, dim=-1), event_shape=(6,), strategy='adaptive_imh', n_chains=64,
                 n_iterations=12, show_progress=False)
    assert out.samples.shape == (12, 64, 6) and torch.isfinite(out.samples).all()
    assert out.statistics.n_attempted_trajectories == 12 * 64


def test_imh_golden_python_callable_split_path(dev):
    from nfmc_amd.samplers import imh
    fx = load_golden('imh_d6')
    d = 6
    s = imh.FixedIMH((d,), lambda x: torch.sum(x ** 4, dim=-1) ** 0.5 * 0 + torch.sum(x ** 2, dim=-1),
                     imh.IMHKernel((d,), flow=_amd_flow(fx, d, 2)), imh.IMHParameters(n_iterations=int(fx['n_iterations'])))
    s.replay = _noise(fx)
    import nfmc_amd.samplers.imh as m
    orig = m.resolve_target
    m.resolve_target = lambda *a, **k: None   # force the split path
    try:
        out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    finally:
        m.resolve_target = orig
    _check_out(out, fx)


# ------------------------------------------------------------------------------------------ flow known answers
FLOW_CASES = [(6, 2, None, 2), (7, 3, 5, 3), (25, 2, None, 2), (64, 2, None, 2), (64, 4, 16, 2), (100, 3, 7, 1),
              (8, 1, 32, 2), (256, 2, 4, 2), (2, 2, 4, 2), (3, 5, 9, 2),
              (100, 4, 100, 5), (64, 2, 64, 1), (128, 2, 128, 2), (30, 2, 40, 3),   # wide conditioners
              (256, 2, 128, 2), (256, 3, 40, 1), (512, 2, 128, 2), (512, 3, 64, 2), (512, 1, 100, 1),   # ... at d = 32 k: mfma_wide.hip
              (32, 2, 64, 2), (96, 3, 128, 1), (160, 2, 100, 2), (320, 3, 64, 2), (448, 2, 128, 2)]


@pytest.mark.parametrize('d,nl,nh,cl', [(6, 2, None, 2), (64, 3, 16, 2), (64, 2, 64, 2), (33, 4, 7, 1), (96, 3, 64, 1), (256, 2, 128, 2)])
def test_nice_flow_matches_oracle(dev, d, nl, nh, cl):
    """'nice' (nfmc/util.py:13): additive couplings through the same kernels (min_scale = 1): forward / inverse /
    log_prob vs the oracle, exact volume preservation of the couplings, round trip, sampling and NeuTra gradient."""
    from nfmc_amd.flows import NICE, Flow
    from nfmc_amd.util import create_flow_object
    from oracle import flow as oflow
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    torch.manual_seed(d + nl)
    of = oflow.perturb_(oflow.Flow(oflow.NICE((d,), n_layers=nl, conditioner_kwargs=ck)), 9, 0.4, 0.8)
    f = create_flow_object('nice', (d,), n_layers=nl, conditioner_kwargs=ck)
    assert isinstance(f.bijection, NICE)
    f.load_state_dict(of.state_dict())
    x = torch.randn(300, d) * 0.8
    with torch.no_grad():
        z0, ld0 = of.bijection.forward(x)
        lp0 = of.log_prob(x)
    z, ld = f.bijection.forward(x)
    np.testing.assert_allclose(z.cpu().numpy(), z0.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(ld.cpu().numpy(), ld0.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), lp0.numpy(), atol=1e-4, rtol=1e-5)
    # couplings preserve volume exactly: logdet is the two elementwise layers' constant
    const = float((of.bijection.layers[0].log_scale.sum() + of.bijection.layers[-1].log_scale.sum()).detach())
    assert float((ld.cpu() - const).abs().max()) < 1e-5 * max(1.0, abs(const))
    assert float((ld.cpu() - ld.cpu()[0]).abs().max()) == 0.0   # bit-identical for every row
    xb, ldi = f.bijection.inverse(z)
    np.testing.assert_allclose(xb.cpu().numpy(), x.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(ldi.cpu().numpy(), -ld.cpu().numpy(), atol=1e-5, rtol=0)
    xs, lq = f.sample(500, return_log_prob=True)
    np.testing.assert_allclose(lq.cpu().numpy(), f.log_prob(xs).cpu().numpy(), atol=2e-4, rtol=1e-5)


def test_nice_neutra_and_jump_strategies(dev):
    """The adjusted-target gradient through additive couplings vs autograd of the oracle, and a jump run."""
    from nfmc_amd import sample
    from nfmc_amd.flows import NICE, Flow
    from nfmc_amd.potentials import SumOfSquares
    from nfmc_amd.samplers.neutra import NeuTraHMC, NeuTraKernel, NeuTraParameters
    from oracle import flow as oflow
    from oracle import potentials as opot
    from oracle import samplers as osamp
    d = 8
    torch.manual_seed(5)
    of = oflow.perturb_(oflow.Flow(oflow.NICE((d,))), 3, 0.4, 0.8)
    f = Flow(NICE((d,)))
    f.load_state_dict(of.state_dict())
    s = NeuTraHMC((d,), SumOfSquares((d,)), kernel=NeuTraKernel((d,), flow=f), params=NeuTraParameters(n_iterations=2))
    z = torch.randn(64, d)
    u, g = s._potential_grad(z.to(dev))
    zz = z.clone().requires_grad_(True)
    u0 = osamp.neutra_adjusted_target(of, opot.sum_squares, (d,))(zz)
    g0, = torch.autograd.grad(u0.sum(), zz)
    np.testing.assert_allclose(u.cpu().numpy(), u0.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), g0.numpy(), atol=5e-5, rtol=1e-4)
    torch.manual_seed(0)
    out = sample(lambda x: torch.sum(x ** 2, dim=-1), event_shape=(d,), strategy='jump_mala', flow='nice', n_chains=256,
                 n_iterations=5, show_progress=False, inner_param_kwargs={'n_iterations': 10})
    assert torch.isfinite(out.samples).all() and out.statistics.n_attempted_jumps == 5 * 256


@pytest.mark.parametrize('d,nl,nh,cl', FLOW_CASES)
def test_flow_matches_oracle_and_known_answers(dev, d, nl, nh, cl):
    from nfmc_amd.flows import Flow, RealNVP
    from oracle import flow as oflow
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    torch.manual_seed(d * 100 + nl)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 7, 0.4, 0.8)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    n = 333
    x = torch.randn(n, d)
    with torch.no_grad():
        z_o, ld_o = of.bijection.forward(x)
        xi_o, ldi_o = of.bijection.inverse(x)
        lp_o = of.log_prob(x)
    z, ld = f.bijection.forward(x)
    np.testing.assert_allclose(z.cpu().numpy(), z_o.numpy(), atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(ld.cpu().numpy(), ld_o.numpy(), atol=1e-4, rtol=1e-5)
    xi, ldi = f.bijection.inverse(x)
    np.testing.assert_allclose(xi.cpu().numpy(), xi_o.numpy(), atol=5e-5, rtol=2e-5)
    np.testing.assert_allclose(ldi.cpu().numpy(), ldi_o.numpy(), atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), lp_o.numpy(), atol=2e-4, rtol=1e-5)
    # (1) inverse(forward(x)) == x ; (2) logdets cancel
    xr, ldr = f.bijection.inverse(z)
    np.testing.assert_allclose(xr.cpu().numpy(), x.numpy(), atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose((ld + ldr).cpu().numpy(), 0, atol=1e-4)
    # (5) sample's log-prob equals log_prob of the returned x
    xs, lq = f.sample(n, return_log_prob=True)
    np.testing.assert_allclose(f.log_prob(xs).cpu().numpy(), lq.cpu().numpy(), atol=3e-4, rtol=1e-5)
    assert len(f.bijection.layers) == 2 + 2 * nl


def test_flow_identity_is_standard_normal(dev):
    from nfmc_amd.flows import Flow, RealNVP
    d = 12
    f = Flow(RealNVP((d,)))
    with torch.no_grad():
        for p in f.parameters():
            p.zero_()
    x = torch.randn(50, d)
    want = -0.5 * (x ** 2).sum(-1) - 0.5 * d * np.log(2 * np.pi)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), want.numpy(), atol=2e-5)
    z, ld = f.bijection.forward(x)
    np.testing.assert_allclose(z.cpu().numpy(), x.numpy(), atol=1e-6)
    np.testing.assert_allclose(ld.cpu().numpy(), 0, atol=1e-6)


def test_flow_logdet_is_jacobian_slogdet(dev):
    """(3): logdet_forward equals slogdet of the autograd Jacobian of the CPU restatement (d <= 8)."""
    from nfmc_amd.flows import Flow, RealNVP
    from oracle import flow as oflow
    d = 6
    torch.manual_seed(3)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=3)), 11, 0.5)
    f = Flow(RealNVP((d,), n_layers=3))
    f.load_state_dict(of.state_dict())
    x = torch.randn(5, d)
    _, ld = f.bijection.forward(x)
    for i in range(5):
        J = torch.autograd.functional.jacobian(lambda v: of.bijection.forward(v[None])[0][0], x[i])
        np.testing.assert_allclose(float(ld[i]), float(torch.linalg.slogdet(J)[1]), atol=1e-4)


# ------------------------------------------------------------------------------------------ native mode vs oracle
def _flips_ok(tr_lr, tr_lu, got_masks, want_masks, tol=1e-4):
    """mask disagreements are allowed only where the oracle's test was within `tol` of a tie"""
    bad = got_masks != want_masks
    if not bad.any():
        return True
    margin = (torch.stack(tr_lu) - torch.stack(tr_lr)).abs().numpy()
    return bool((margin[bad] < tol).all())


@pytest.mark.parametrize('d,n,k', [(64, 300, 12), (25, 257, 8), (128, 130, 6)])
def test_mala_native_stream_matches_oracle(dev, d, n, k):
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    from oracle import samplers as osamp, potentials as opot
    torch.manual_seed(5)
    x0 = torch.randn(n, d)
    seed = 4242
    s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_iterations=k))
    s.seed = seed
    out = s.sample(x0, show_progress=False)
    tr = osamp.mcmc_sample(x0, opot.sum_squares, 'langevin', k, d ** (-1 / 3), noise=osamp.PhiloxNoise(seed))
    got = out.samples.reshape(k, n, d)
    want = tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 1e-4
    assert same.float().mean() > 0.98          # a near-tie flip changes the whole later trajectory of that chain
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=1e-4, rtol=0)
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= max(2, int(0.01 * n * k))


def test_hmc_native_stream_matches_oracle(dev):
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    from oracle import samplers as osamp, potentials as opot
    d, n, k, L, h = 32, 200, 5, 7, 0.07
    torch.manual_seed(6)
    x0 = torch.randn(n, d)
    s = mcmc.HMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h),
                 mcmc.HMCParameters(n_iterations=k))
    s.seed = 99
    out = s.sample(x0, show_progress=False)
    tr = osamp.mcmc_sample(x0, opot.sum_squares, 'hmc', k, h, n_leapfrog=L, noise=osamp.PhiloxNoise(99))
    got, want = out.samples.reshape(k, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 1e-4
    assert same.float().mean() > 0.98
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=1e-4, rtol=0)


def test_jump_mala_native_stream_matches_oracle(dev):
    from nfmc_amd.samplers import jump, mcmc
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import samplers as osamp, potentials as opot, flow as oflow
    d, n, T, K = 16, 192, 3, 5
    torch.manual_seed(8)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,))), 5, 0.2, 0.7071)
    f = Flow(RealNVP((d,)))
    f.load_state_dict(of.state_dict())
    x0 = 0.7 * torch.randn(n, d)
    s = jump.JumpMALA((d,), SumOfSquares((d,)), NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T),
                      None, mcmc.LangevinParameters(n_iterations=K))
    s.seed = 31337
    out = s.sample(x0, show_progress=False)
    s.fuse_jump_tail = True
    out_fused = s.sample(x0, show_progress=False)
    assert torch.allclose(out_fused.samples, out.samples, atol=1e-5)   # same stream, same transitions
    assert out_fused.statistics.n_accepted_jumps == out.statistics.n_accepted_jumps
    tr = osamp.jump_sample(x0, opot.sum_squares, of, 'langevin', T, K, d ** (-1 / 3), noise=osamp.PhiloxNoise(31337))
    got, want = out.samples.reshape(T * (K + 1), n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 2e-4
    assert same.float().mean() > 0.97
    assert out.statistics.n_attempted_jumps == n * T
    assert abs(out.statistics.n_accepted_jumps - tr.n_accepted_jumps) <= max(2, int(0.02 * n * T))
    assert out.statistics.n_accepted_jumps > 0.1 * n * T   # the scaled flow proposes well: jumps do get accepted


def test_philox_7_round_stream(dev):
    """The opt-in Philox4x32-7 stream (NfmcRng.rounds = 7, `sample(..., rng_rounds=7)`): words and normals equal the
    oracle's (oracle/philox.py with rounds=7, itself pinned by the Random123 known-answer vectors), MALA and jump_mala
    runs on it equal the oracle fed the same stream, launches without a 7-round kernel raise ValueError."""
    from nfmc_amd import hip, sample
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import Funnel, SumOfSquares
    from nfmc_amd.samplers import jump, mcmc
    from oracle import philox, samplers as osamp, potentials as opot, flow as oflow
    n, d = 70, 22
    out = torch.empty(n, d, device=dev)
    rng = hip.make_rng(0xDEADBEEF12345, 1000, 7, rounds=7)
    hip.check(hip.lib().nfmc_philox_normals_f32(C.byref(rng), hip.TAG_NOISE, n, d, hip.ptr(out), hip.stream()), 'normals')
    want = philox.normal_field(0xDEADBEEF12345, np.arange(1000, 1000 + n, dtype=np.uint32), 7, d, philox.TAG_NOISE, rounds=7)
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=4e-6)
    un = torch.empty(n, device=dev)
    hip.check(hip.lib().nfmc_philox_uniforms_f32(C.byref(rng), hip.TAG_ACCEPT, n, hip.ptr(un), hip.stream()), 'uniforms')
    assert np.array_equal(un.cpu().numpy(), philox.accept_uniform(0xDEADBEEF12345, np.arange(1000, 1000 + n, dtype=np.uint32), 7, rounds=7))
    # MALA on the exact-fit kernel
    d, n, k = 64, 300, 12
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(5))
    s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_iterations=k))
    s.seed, s.rng_rounds = 4242, 7
    o7 = s.sample(x0, show_progress=False)
    tr = osamp.mcmc_sample(x0, opot.sum_squares, 'langevin', k, d ** (-1 / 3), noise=osamp.PhiloxNoise(4242, rounds=7))
    got, want = o7.samples.reshape(k, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 1e-4
    assert same.float().mean() > 0.98
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=1e-4, rtol=0)
    s.rng_rounds = 10
    o10 = s.sample(x0, show_progress=False)
    assert not torch.allclose(o10.samples, o7.samples)          # a different stream
    # HMC and jump_mala (register flow-MH kernel)
    h = mcmc.HMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=4, step_size=0.05), mcmc.HMCParameters(n_iterations=4))
    h.seed, h.rng_rounds = 9, 7
    oh = h.sample(x0, show_progress=False)
    trh = osamp.mcmc_sample(x0, opot.sum_squares, 'hmc', 4, 0.05, n_leapfrog=4, noise=osamp.PhiloxNoise(9, rounds=7))
    sameh = (oh.samples.reshape(4, n, d) - trh.stacked()).abs().amax(dim=(0, 2)) < 1e-4
    assert sameh.float().mean() > 0.98
    torch.manual_seed(8)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,))), 5, 0.1, 0.7071)
    f = Flow(RealNVP((d,)))
    f.load_state_dict(of.state_dict())
    T, K = 2, 5
    j = jump.JumpMALA((d,), SumOfSquares((d,)), NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T), None,
                      mcmc.LangevinParameters(n_iterations=K))
    j.seed, j.rng_rounds = 31337, 7
    oj = j.sample(x0, show_progress=False)
    trj = osamp.jump_sample(x0, opot.sum_squares, of, 'langevin', T, K, d ** (-1 / 3), noise=osamp.PhiloxNoise(31337, rounds=7))
    samej = (oj.samples.reshape(T * (K + 1), n, d) - trj.stacked()).abs().amax(dim=(0, 2)) < 2e-4
    assert samej.float().mean() > 0.97
    assert abs(oj.statistics.n_accepted_jumps - trj.n_accepted_jumps) <= max(2, int(0.03 * n * T))
    # no 7-round kernel: fail loudly, never fall back to the other stream
    with pytest.raises(ValueError):
        sample(Funnel((d,), 3.0), strategy='mala', x0=x0, n_iterations=2, show_progress=False, rng_rounds=7)
    with pytest.raises(ValueError):
        sample(SumOfSquares((d,)), strategy='neutra_hmc', x0=x0, n_iterations=2, show_progress=False, rng_rounds=7)
    with pytest.raises(ValueError):
        sample(SumOfSquares((d,)), strategy='mala', x0=x0, n_iterations=2, show_progress=False, rng_rounds=8)


@pytest.mark.parametrize('d,nl,nh,cl,pot', [(6, 2, None, 2, 'sumsq'), (7, 3, 5, 1, 'funnel'), (24, 2, 16, 2, 'sumsq'),
                                             (33, 1, 32, 2, 'funnel')])
def test_neutra_spline_potential_and_gradient_match_autograd(dev, d, nl, nh, cl, pot):
    """f4: the reverse sweep through rational-quadratic spline couplings ('c-rqnsf'; hand-written adjoints of the
    spline's forward formulas + the implicit-function theorem for the inverse direction, flow_device.hpp
    rqs_inverse_backward) equals torch autograd through the CPU restatement of NeuTra.adjusted_target
    (neutra.py:58-68), including chains whose coordinates leave the spline's interval (identity tails)."""
    from nfmc_amd import hip
    from nfmc_amd.util import create_flow_object
    from nfmc_amd.potentials import SumOfSquares, Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    torch.manual_seed(d * 3 + nl)
    of = oflow.perturb_(oflow.Flow(oflow.CRQNSF((d,), n_layers=nl, conditioner_kwargs=ck)), 9, 1.0 if d <= 8 else 0.3, 0.8)
    f = create_flow_object('c-rqnsf', (d,), n_layers=nl, conditioner_kwargs=ck)
    f.load_state_dict(of.state_dict())
    target_cpu = opot.sum_squares if pot == 'sumsq' else opot.funnel(3.0)
    target = SumOfSquares((d,)) if pot == 'sumsq' else Funnel((d,), 3.0)
    n = 130
    z = 1.2 * torch.randn(n, d)
    z[:6] *= 5.0                       # some coordinates beyond the bound B = 5
    z = z.requires_grad_(True)
    u_ref = osamp.neutra_adjusted_target(of, target_cpu, (d,))(z)
    g_ref, = torch.autograd.grad(u_ref.sum(), z)
    st, _keep = f.bijection.packed(dev)
    pd = target.descriptor(dev)
    zd = z.detach().to(dev).contiguous()
    u = torch.empty(n, device=dev)
    g = torch.empty(n, d, device=dev)
    hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zd), n, hip.ptr(u), hip.ptr(g),
                                                       hip.stream()), 'neutra_potential_grad')
    ur, gr = u_ref.detach(), g_ref
    np.testing.assert_allclose(u.cpu().numpy(), ur.numpy(), atol=2e-4 * (1 + float(ur.abs().max())), rtol=0)
    err = (g.cpu() - gr).abs().amax(dim=1) / (1 + gr.abs().amax(dim=1))
    # a coordinate within rounding of a knot may sit in the neighbouring bin on the other side: the value is continuous
    # there, the gradient of the log-derivative term is not
    assert (err < 2e-3).float().mean() > 0.97, float((err < 2e-3).float().mean())
    assert float(err.median()) < 2e-4


def test_neutra_hmc_with_spline_flow_on_the_fused_kernel(dev):
    """neutra_hmc with a 'c-rqnsf' flow runs inside nfmc_neutra_hmc_steps_f32 (no split path) and matches the oracle on
    the native stream."""
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.util import create_flow_object
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    d, n, T, L, h = 10, 120, 3, 4, 0.03
    torch.manual_seed(4)
    of = oflow.perturb_(oflow.Flow(oflow.CRQNSF((d,))), 9, 0.5, 0.8)
    f = create_flow_object('c-rqnsf', (d,))
    f.load_state_dict(of.state_dict())
    z0 = 0.7 * torch.randn(n, d)
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))

    def boom(*a, **k):
        raise AssertionError('split path taken')
    s.inner_sampler.sample = boom
    s.seed = 12
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.sum_squares, of, T, h, None, L, noise=osamp.PhiloxNoise(12))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 1e-3
    assert same.float().mean() > 0.93, float(same.float().mean())
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 6


# ------------------------------------------------------------------------------------------ size-independent properties
def test_moments_of_sum_squares_target_large(dev):
    """U = sum x^2 => N(0, I/2): mean 0, variance 0.5 (README.md:45-46) at n=65536, d=64."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    d, n = 64, 65536
    torch.manual_seed(0)
    out = sample(SumOfSquares((d,)), strategy='mala', n_chains=n, n_iterations=300, show_progress=False,
                 x0=torch.randn(n, d) * 0.7071, param_kwargs={'store_samples': False})
    assert out.samples is None
    assert abs(float(out.mean.abs().max())) < 3e-3
    np.testing.assert_allclose(out.variance.numpy(), 0.5, rtol=4e-3)
    assert 0.2 < out.statistics.acceptance_rate < 0.9


def test_run_twice_is_bitwise_identical(dev):
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    d, n = 64, 4096
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(1))
    outs = []
    for _ in range(2):
        torch.manual_seed(3)  # same flow initialisation in both runs
        o = sample(SumOfSquares((d,)), strategy='jump_mala', n_iterations=3, show_progress=False, x0=x0, seed=7,
                   inner_param_kwargs={'n_iterations': 20}, param_kwargs={'store_samples': False})
        outs.append(o)
    assert torch.equal(outs[0].running_samples.last_sample, outs[1].running_samples.last_sample)
    assert torch.equal(outs[0].statistics.expectations['first_moment'].total, outs[1].statistics.expectations['first_moment'].total)
    assert outs[0].statistics.n_accepted_trajectories == outs[1].statistics.n_accepted_trajectories


def test_sharded_chains_equal_single_run(dev):
    """Global-chain-id keyed noise: simulating rows [lo, hi) alone equals the slice of the full run."""
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    from nfmc_amd.dist import Shard
    d, n, k = 64, 1000, 10
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(2))
    full = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_iterations=k, store_samples=False))
    full.seed = 5
    a = full.sample(x0, show_progress=False).running_samples.last_sample
    parts = []
    for r in range(3):
        s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_iterations=k, store_samples=False))
        s.seed = 5
        s.shard = Shard(rank=r, world=3)
        s.shard.merge_statistics = lambda st: st
        parts.append(s.sample(x0, show_progress=False).running_samples.last_sample)
    assert torch.equal(torch.cat(parts), a)


# ------------------------------------------------------------------------------------------ API shape tests (reference test/)
@pytest.mark.parametrize('strategy', ['mala', 'ula', 'hmc', 'uhmc', 'mh', 'imh', 'jump_mala', 'jump_ula', 'jump_hmc',
                                      'jump_uhmc', 'jump_mh', 'neutra_mh'])
def test_sample_api_shapes(dev, strategy):
    """mirror of test/test_samplers.py:175-249 + test_moment_estimation.py:31-49 for the path's strategies."""
    from nfmc_amd import sample
    torch.manual_seed(0)
    n_iterations, n_chains, event_shape = 3, 20, (10,)
    out = sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=event_shape, strategy=strategy, n_chains=n_chains,
                 n_iterations=n_iterations, n_warmup_iterations=3, show_progress=False,
                 inner_param_kwargs={'n_iterations': 4} if 'jump' in strategy else None)
    k = n_iterations * 5 if 'jump' in strategy else n_iterations
    assert out.samples.shape == (k, n_chains, *event_shape)
    assert torch.isfinite(out.samples).all()
    for m in (out.mean, out.second_moment, out.variance):
        assert m.shape == event_shape and m.isfinite().all()


def test_custom_event_shape(dev):
    """test/test_custom_shapes.py: 2-D events are flattened row-major."""
    from nfmc_amd import sample
    out = sample(lambda x: torch.sum(x ** 2, dim=(1, 2)), event_shape=(8, 8), strategy='jump_hmc', n_chains=7,
                 n_iterations=2, show_progress=False)
    assert out.samples.shape == (2 * 6, 7, 8, 8)
    out = sample(lambda x: torch.sum(x ** 2, dim=(1, 2)), event_shape=(8, 8), strategy='imh', n_chains=7,
                 n_iterations=2, show_progress=False)
    assert out.samples.shape == (2, 7, 8, 8)


def test_no_sample_storing(dev):
    """test/test_no_sample_storing.py."""
    from nfmc_amd.sample import create_sampler
    for strategy in ['mala', 'hmc', 'imh', 'jump_mala']:
        s = create_sampler(target=lambda x: torch.sum(x ** 2, dim=-1), event_shape=(10,), strategy=strategy,
                           param_kwargs={'store_samples': False})
        out = s.sample(torch.randn(20, 10), show_progress=False, time_limit_seconds=5.0)
        assert out.samples is None
        assert out.running_samples.last_sample.shape == (20, 10)


def test_flow_kwargs(dev):
    """test/test_flow_kwargs.py (test_basic and test_advanced)."""
    from nfmc_amd import sample
    t = lambda x: torch.sum(x ** 2, dim=-1)
    basic = sample(event_shape=(100,), target=t, flow='realnvp', strategy='imh', n_iterations=3, show_progress=False)
    adv = sample(event_shape=(100,), target=t, flow='realnvp%{"n_layers": 10}', strategy='imh', n_iterations=3,
                 show_progress=False)
    assert len(adv.kernel.flow.bijection.layers) > len(basic.kernel.flow.bijection.layers)
    adv2 = sample(event_shape=(100,), target=t, strategy='imh', n_iterations=3, show_progress=False,
                  flow='realnvp%{"n_layers": 10, "conditioner_kwargs": {"n_layers": 5, "n_hidden": 100}}')
    assert len(adv2.kernel.flow.bijection.layers) > len(basic.kernel.flow.bijection.layers)
    assert adv2.samples.shape == (3, 100, 100) and torch.isfinite(adv2.samples).all()


# ------------------------------------------------------------------------------------------ NeuTra (K5)
@pytest.mark.parametrize('d,nl,nh,cl,pot', [(6, 2, None, 2, 'sumsq'), (7, 3, 5, 3, 'sumsq'), (16, 2, 16, 1, 'funnel'),
                                             (64, 2, None, 2, 'sumsq'), (128, 2, 32, 2, 'funnel'), (9, 1, 8, 4, 'sumsq')])
def test_neutra_potential_and_gradient_match_autograd(dev, d, nl, nh, cl, pot):
    """(7): the hand-written VJP equals torch autograd through the CPU restatement of
    NeuTra.adjusted_target (neutra.py:58-68)."""
    from nfmc_amd import hip
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares, Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    torch.manual_seed(d + nl)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 3, 0.4, 0.8)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    target_cpu = opot.sum_squares if pot == 'sumsq' else opot.funnel(3.0)
    target = SumOfSquares((d,)) if pot == 'sumsq' else Funnel((d,), 3.0)
    n = 150
    z = (0.6 * torch.randn(n, d)).requires_grad_(True)
    u_ref = osamp.neutra_adjusted_target(of, target_cpu, (d,))(z)
    g_ref, = torch.autograd.grad(u_ref.sum(), z)
    st, _keep = f.bijection.packed(dev)
    pd = target.descriptor(dev)
    zd = z.detach().to(dev).contiguous()
    u = torch.empty(n, device=dev)
    g = torch.empty(n, d, device=dev)
    hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zd), n, hip.ptr(u), hip.ptr(g),
                                                       hip.stream()), 'neutra_potential_grad')
    scale = 1 + float(g_ref.abs().max())
    np.testing.assert_allclose(u.cpu().numpy(), u_ref.detach().numpy(), atol=2e-4 * (1 + float(u_ref.detach().abs().max())), rtol=0)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=2e-4 * scale, rtol=0)


def test_neutra_hmc_golden(dev):
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden('neutra_hmc_d6')
    d = 6
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)),
                         mcmc.HMCKernel(event_size=d, n_leapfrog_steps=int(fx['n_leapfrog']), step_size=float(fx['step_size'])),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=_amd_flow(fx, d)),
                         neutra.NeuTraParameters(n_iterations=int(fx['n_iterations'])))
    s.replay = _noise(fx)
    out = s.sample(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx, atol=1e-4)


def test_neutra_hmc_native_stream_matches_oracle(dev):
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    d, n, T, L, h = 12, 130, 4, 5, 0.05
    torch.manual_seed(12)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), conditioner_kwargs={'n_hidden': 8})), 9, 0.3)
    f = Flow(RealNVP((d,), conditioner_kwargs={'n_hidden': 8}))
    f.load_state_dict(of.state_dict())
    z0 = 0.5 * torch.randn(n, d)
    imd = torch.linspace(0.8, 1.3, d)
    s = neutra.NeuTraHMC((d,), Funnel((d,), 3.0), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h,
                                                                 inv_mass_diag=imd.clone()),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
    s.seed = 77
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.funnel(3.0), of, T, h, imd, L, noise=osamp.PhiloxNoise(77))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 3e-4
    assert same.float().mean() > 0.97
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=3e-4, rtol=0)
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 4
    assert out.statistics.n_target_calls == (2 * L + 2) * n * T and out.statistics.n_target_gradient_calls == 2 * L * n * T


def test_neutra_sample_api(dev):
    from nfmc_amd import sample
    out = sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(10,), strategy='neutra_hmc', n_chains=20,
                 n_iterations=3, show_progress=False, inner_kernel_kwargs={'n_leapfrog_steps': 4, 'step_size': 0.1})
    assert out.samples.shape == (3, 20, 10) and torch.isfinite(out.samples).all()
    assert out.mean.shape == (10,) and out.kernel.flow is not None


# ------------------------------------------------------------------------------------------ warmup (f1)
def test_mala_warmup_tuning_golden(dev):
    """MCMCSampler.warmup + MetropolisSampler.update_kernel (mcmc/base.py:39-54,142-161) against the reference."""
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    fx = load_golden('tuning')
    d = 5
    s = mcmc.MALA((d,), SumOfSquares((d,)), mcmc.LangevinKernel(event_size=d),
                  mcmc.LangevinParameters(n_iterations=6, n_warmup_iterations=6))
    s.replay = _noise(fx)
    out = s.warmup(torch.from_numpy(fx['x0']), show_progress=False)
    _check_out(out, fx)
    np.testing.assert_allclose(s.kernel.step_size, float(fx['tuned_step_size']), rtol=1e-5)
    np.testing.assert_allclose(s.kernel.inv_mass_diag.numpy(), fx['tuned_inv_mass_diag'], atol=1e-6)
    assert s.params.tuning is False and s.params.n_iterations == 6


@pytest.mark.parametrize('kind', ['mala', 'hmc', 'mh', 'mala_mass'])
def test_device_tuning_equals_the_host_controller(dev, kind, monkeypatch):
    """f1: the warmup controller on the device (NfmcTune: mass-diagonal EMA + dual averaging inside the statistics fold,
    no host round trip per step) against the host-side `update_kernel` (mcmc/base.py:142-161, tuning.py:15-41) fed the
    same transitions: tuned step size, mass diagonal, controller state, samples and counters."""
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import DiagonalGaussian
    d, n, k = 24, 500, 40
    sig = torch.linspace(0.5, 2.0, d)
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(8)) * sig

    def make():
        pot = DiagonalGaussian((d,), 0.0, sig)
        if kind.startswith('mala'):
            kern = mcmc.LangevinKernel(event_size=d, inv_mass_diag=torch.linspace(0.8, 1.2, d) if kind == 'mala_mass' else None)
            s = mcmc.MALA((d,), pot, kern, mcmc.LangevinParameters(n_iterations=5, n_warmup_iterations=k))
        elif kind == 'hmc':
            s = mcmc.HMC((d,), pot, mcmc.HMCKernel(event_size=d, n_leapfrog_steps=4, step_size=0.05),
                         mcmc.HMCParameters(n_iterations=5, n_warmup_iterations=k))
        else:
            s = mcmc.MH((d,), pot, mcmc.MHKernel(event_size=d, inv_mass_diag=torch.full((d,), 0.3)),
                        mcmc.MHParameters(n_iterations=5, n_warmup_iterations=k))
        s.seed = 77
        return s

    a = make()
    out_a = a.warmup(x0, show_progress=False)
    monkeypatch.setenv('NFMC_TUNE_DEVICE', '0')
    b = make()
    out_b = b.warmup(x0, show_progress=False)
    np.testing.assert_allclose(a.kernel.step_size, b.kernel.step_size, rtol=2e-5)
    np.testing.assert_allclose(a.kernel.inv_mass_diag.numpy(), b.kernel.inv_mass_diag.numpy(), rtol=1e-5, atol=1e-7)
    if kind != 'mh':
        assert a.kernel.step_size != make().kernel.step_size            # the controller did move it
        assert a.kernel.da.iteration == b.kernel.da.iteration == 10 + k
        np.testing.assert_allclose(a.kernel.da.error_sum, b.kernel.da.error_sum, atol=1e-4)
    same = ((out_a.samples - out_b.samples).abs().amax(dim=(0, 2)) < 1e-3).float().mean()
    assert same > 0.95      # identical streams; a chain parts only through an accept test within rounding of a tie
    assert abs(out_a.statistics.n_accepted_trajectories - out_b.statistics.n_accepted_trajectories) <= 0.01 * n * k


def test_device_tuning_batched_controller(dev):
    """`tune_every = K`: one controller update per K transitions (K transitions per launch), statistics pooled over
    them; the tuned step size still lands where the per-transition schedule puts it for a well-conditioned target."""
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares
    d, n = 64, 4096
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(9)) * 0.7071
    tuned = {}
    for every in (1, 10):
        s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_warmup_iterations=300, tune_every=every))
        s.seed = 5
        out = s.warmup(x0, show_progress=False)
        tuned[every] = s.kernel.step_size
        assert out.statistics.n_attempted_trajectories == n * 300
        assert s.kernel.da.iteration == 10 + 300 // every
    assert 0.6 < tuned[10] / tuned[1] < 1.6, tuned


@pytest.mark.parametrize('strategy', ['mala', 'hmc', 'imh', 'neutra_hmc'])
def test_warmup_api(dev, strategy):
    """test/test_warmup.py: warmup with store_samples=False returns last_sample only."""
    from nfmc_amd.sample import create_sampler
    torch.manual_seed(0)
    s = create_sampler(target=lambda x: torch.sum(x ** 2, dim=-1), event_shape=(10,), strategy=strategy,
                       param_kwargs={'store_samples': False, 'n_warmup_iterations': 5},
                       inner_kernel_kwargs={'n_leapfrog_steps': 3} if strategy == 'neutra_hmc' else None)
    if hasattr(s.params, 'warmup_fit_kwargs') and s.params.warmup_fit_kwargs:
        s.params.warmup_fit_kwargs.update(n_epochs=20, n_samples=64)
    out = s.warmup(torch.randn(20, 10), show_progress=False, time_limit_seconds=20.0)
    assert out.samples is None
    assert out.running_samples.last_sample.shape == (20, 10)


def test_jump_warmup_then_sample_improves_jump_acceptance(dev):
    """JumpNFMC.warmup (jump.py:104-154): tune the inner sampler, MLE-fit the flow on its samples; afterwards
    the jumps of the hot path are accepted far more often than with the untrained flow."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    torch.manual_seed(0)
    d, n = 8, 512
    kw = dict(strategy='jump_mala', n_chains=n, n_iterations=4, show_progress=False,
              inner_param_kwargs={'n_iterations': 20}, param_kwargs={'store_samples': False}, seed=3)
    cold = sample(SumOfSquares((d,)), **kw)
    torch.manual_seed(0)
    warm = sample(SumOfSquares((d,)), warmup=True, n_warmup_iterations=100, **kw)
    assert warm.statistics.jump_acceptance_rate > cold.statistics.jump_acceptance_rate + 0.2
    assert warm.statistics.jump_acceptance_rate > 0.4
    np.testing.assert_allclose(warm.variance.numpy(), 0.5, atol=0.08)


# ------------------------------------------------------------------------------------------ matrix-core path (wide conditioners)
@pytest.mark.parametrize('d,nl,nh,cl,pot', [(64, 2, 64, 1, 'sumsq'), (64, 3, 40, 2, 'funnel'), (128, 2, 128, 2, 'funnel'),
                                             (128, 1, 100, 1, 'sumsq'), (128, 2, 64, 2, 'sumsq'), (64, 2, 128, 2, 'sumsq')])
def test_neutra_mfma_potential_and_gradient_match_autograd(dev, d, nl, nh, cl, pot):
    """Wide conditioners run on v_mfma_f32_32x32x2_f32; same check as the VALU path: VJP == autograd of the
    CPU restatement.  fp32 MFMA is an exact fp32 fma chain, so the tolerance stays at the fp32 level."""
    from nfmc_amd import hip
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares, Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    ck = {'n_layers': cl, 'n_hidden': nh}
    torch.manual_seed(d + nl + nh)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 3, 0.15, 0.8)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    target_cpu = opot.sum_squares if pot == 'sumsq' else opot.funnel(3.0)
    target = SumOfSquares((d,)) if pot == 'sumsq' else Funnel((d,), 3.0)
    n = 200   # not a multiple of the 128-chain workgroup tile
    z = (0.6 * torch.randn(n, d)).requires_grad_(True)
    u_ref = osamp.neutra_adjusted_target(of, target_cpu, (d,))(z)
    g_ref, = torch.autograd.grad(u_ref.sum(), z)
    st, _keep = f.bijection.packed(dev)
    pd = target.descriptor(dev)
    zd = z.detach().to(dev).contiguous()
    u = torch.empty(n, device=dev)
    g = torch.empty(n, d, device=dev)
    hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zd), n, hip.ptr(u), hip.ptr(g),
                                                       hip.stream()), 'neutra_potential_grad')
    np.testing.assert_allclose(u.cpu().numpy(), u_ref.detach().numpy(), atol=3e-4 * (1 + float(u_ref.detach().abs().max())), rtol=0)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=3e-4 * (1 + float(g_ref.abs().max())), rtol=0)


@pytest.mark.parametrize('d,nl,nh,cl,pot', [(256, 2, 128, 2, 'funnel'), (256, 3, 64, 1, 'sumsq'), (256, 1, 100, 2, 'sumsq'),
                                             (512, 2, 128, 2, 'sumsq'), (512, 3, 40, 1, 'funnel'), (512, 2, 64, 2, 'funnel'),
                                             (32, 2, 64, 2, 'funnel'), (96, 3, 128, 1, 'sumsq'), (160, 2, 100, 2, 'funnel'),
                                             (192, 3, 64, 2, 'sumsq'), (320, 2, 128, 2, 'funnel'), (448, 3, 40, 1, 'sumsq')])
def test_neutra_wide_event_mfma_potential_and_gradient_match_autograd(dev, d, nl, nh, cl, pot):
    """Round 3: every d that is a multiple of 32 (other than 64 / 128, which the register-resident kernels serve) with
    conditioners of width 33..128 on the matrix cores with the state and gradient STREAMED through a scratch slab
    (csrc/mfma_wide.hip; these shapes used to fall back to torch autograd through the restatement).  Odd and even numbers
    of coupling layers (latent order reversed or not), one and two hidden layers, padded widths, halves that are not whole
    128-coordinate slices or 64-coordinate groups, a batch that is not a multiple of the 128-chain workgroup tile."""
    from nfmc_amd import hip
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares, Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    ck = {'n_layers': cl, 'n_hidden': nh}
    torch.manual_seed(d + nl + nh)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 3, 0.1, 0.8)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    target_cpu = opot.sum_squares if pot == 'sumsq' else opot.funnel(3.0)
    target = SumOfSquares((d,)) if pot == 'sumsq' else Funnel((d,), 3.0)
    n = 300
    z = (0.6 * torch.randn(n, d)).requires_grad_(True)
    u_ref = osamp.neutra_adjusted_target(of, target_cpu, (d,))(z)
    g_ref, = torch.autograd.grad(u_ref.sum(), z)
    st, _keep = f.bijection.packed(dev)
    pd = target.descriptor(dev)
    zd = z.detach().to(dev).contiguous()
    u = torch.full((n,), float('nan'), device=dev)
    g = torch.full((n, d), float('nan'), device=dev)
    hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zd), n, hip.ptr(u), hip.ptr(g),
                                                       hip.stream()), 'neutra_potential_grad')
    np.testing.assert_allclose(u.cpu().numpy(), u_ref.detach().numpy(), atol=3e-4 * (1 + float(u_ref.detach().abs().max())), rtol=0)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=3e-4 * (1 + float(g_ref.abs().max())), rtol=0)
    # twice the same: nothing depends on what the slab held
    u2 = torch.empty_like(u)
    g2 = torch.empty_like(g)
    hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(zd), n, hip.ptr(u2), hip.ptr(g2),
                                                       hip.stream()), 'neutra_potential_grad')
    assert torch.equal(u, u2) and torch.equal(g, g2)


@pytest.mark.parametrize('d,nh,cl,nl', [(64, 64, 1, 2), (128, 128, 2, 2), (64, 128, 2, 3)])
def test_neutra_hmc_mfma_native_stream_matches_oracle(dev, d, nh, cl, nl):
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import Funnel, SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    n, T, L, h = 150, 3, 4, 0.03
    ck = {'n_hidden': nh, 'n_layers': cl}
    torch.manual_seed(d + nh)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 9, 0.1)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    z0 = 0.5 * torch.randn(n, d)
    imd = torch.linspace(0.8, 1.3, d)
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h,
                                                                    inv_mass_diag=imd.clone()),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
    s.seed = 78
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.sum_squares, of, T, h, imd, L, noise=osamp.PhiloxNoise(78))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 5e-4
    assert same.float().mean() > 0.96
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=5e-4, rtol=0)
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 5
    np.testing.assert_allclose(out.mean.numpy(), tr.moments.first.numpy(), atol=2e-3)
    np.testing.assert_allclose(out.second_moment.numpy(), tr.moments.second.numpy(), atol=3e-3)


# ------------------------------------------------------------------------------------------ N > 1 on one card
SHARD_WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world, port, outdir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0)
from nfmc_amd import sample
from nfmc_amd.dist import Shard
from nfmc_amd.potentials import SumOfSquares
d, n = 64, 4096
x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(3)) * 0.7
torch.manual_seed(1)   # identical flow weights on every rank
out = sample(SumOfSquares((d,)), strategy='jump_mala', n_iterations=3, show_progress=False, x0=x0, seed=11,
             shard=Shard(), inner_param_kwargs={'n_iterations': 10}, param_kwargs={'store_samples': False})
st = out.statistics
# warmup with a flow fit: every rank contributes its share of the refit buffer (all-gather C1) and fits the same rows
torch.manual_seed(2)
out2 = sample(SumOfSquares((d,)), strategy='jump_mala', n_iterations=2, n_warmup_iterations=6, warmup=True,
              show_progress=False, x0=x0[:600], seed=12, shard=Shard(), inner_param_kwargs={'n_iterations': 4},
              param_kwargs={'flow_fit_kwargs': {'n_epochs': 3}})
weights = torch.cat([p.detach().flatten().cpu() for p in out2.kernel.flow.parameters()])
torch.save({'last': out.running_samples.last_sample.cpu(), 'mean': out.mean, 'second': out.second_moment,
            'acc': st.n_accepted_trajectories, 'att': st.n_attempted_trajectories, 'jacc': st.n_accepted_jumps,
            'jatt': st.n_attempted_jumps, 'calls': st.n_target_calls, 'weights': weights,
            'att2': out2.statistics.n_attempted_trajectories, 'mean2': out2.mean},
           os.path.join(outdir, f'r{rank}.pt'))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharded_sampling_matches_single_process(dev, tmp_path):
    """One process per rank (gloo rendezvous, both ranks on this card): every rank simulates its block of the
    global chains and ends with the statistics of ALL chains (collective C2); together they reproduce the
    single-process run bit for bit (global-chain-id keyed noise)."""
    import socket, subprocess, sys, os
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(SHARD_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = str(s.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, str(script), root, str(r), '2', port, str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    d, n = 64, 4096
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(3)) * 0.7
    torch.manual_seed(1)
    ref = sample(SumOfSquares((d,)), strategy='jump_mala', n_iterations=3, show_progress=False, x0=x0, seed=11,
                 inner_param_kwargs={'n_iterations': 10}, param_kwargs={'store_samples': False})
    r0, r1 = (torch.load(tmp_path / f'r{r}.pt') for r in range(2))
    assert torch.equal(torch.cat([r0['last'], r1['last']]), ref.running_samples.last_sample.cpu())
    for r in (r0, r1):   # both ranks hold the merged (global) statistics
        assert r['acc'] == ref.statistics.n_accepted_trajectories and r['att'] == ref.statistics.n_attempted_trajectories
        assert r['jacc'] == ref.statistics.n_accepted_jumps and r['jatt'] == ref.statistics.n_attempted_jumps
        assert r['calls'] == ref.statistics.n_target_calls
        np.testing.assert_allclose(r['mean'].numpy(), ref.mean.numpy(), atol=1e-6)
        np.testing.assert_allclose(r['second'].numpy(), ref.second_moment.numpy(), atol=1e-6)
    # the warmup fit: both ranks hold the same fitted flow (same gathered rows, same optimiser steps) and the merged
    # statistics of all 600 chains
    assert torch.isfinite(r0['weights']).all() and torch.equal(r0['weights'], r1['weights'])
    assert r0['att2'] == r1['att2'] == 600 * 2 * 4 and torch.equal(r0['mean2'], r1['mean2'])


def test_integration_stub_runs_and_matches_the_package(dev):
    """INTEGRATION.md's reference-side binding, executed verbatim, equals nfmc_amd's MALA with the same seed."""
    from test_host_cpu import _integration_stub_namespace
    from nfmc_amd.potentials import SumOfSquares
    from nfmc_amd.samplers import mcmc
    ns = _integration_stub_namespace()
    n, d, k, h, seed = 512, 64, 25, 0.25, 1234
    torch.manual_seed(0)
    x0 = torch.randn(n, d)
    x = x0.to(dev).clone()
    sum_x, sum_x2, counters = ns['mala_steps_sum_squares'](x, k, h, seed)
    torch.cuda.synchronize()
    s = mcmc.MALA((d,), SumOfSquares((d,)), mcmc.LangevinKernel(event_size=d, step_size=h),
                  mcmc.LangevinParameters(n_iterations=k, store_samples=False))
    s.seed = seed
    out = s.sample(x0, show_progress=False)
    assert torch.equal(out.running_samples.last_sample.reshape(n, d).cpu(), x.cpu())
    assert int(counters[0]) == out.statistics.n_accepted_trajectories and int(counters[1]) == n * k
    np.testing.assert_allclose((sum_x / (n * k)).cpu().numpy(), out.mean.numpy().astype(np.float64), atol=1e-6)


@pytest.mark.parametrize('d,cl', [(64, 2), (128, 1)])
def test_neutra_narrow_conditioner_on_matrix_cores_equals_valu_path(dev, d, cl, monkeypatch):
    """NeuTra presents a narrow conditioner zero-padded to 64 hidden units so that d = 64 / 128 runs on the
    matrix-core kernels: the gradient equals the VALU kernels' (and autograd of the oracle) and a run agrees."""
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import Funnel
    from nfmc_amd.samplers import mcmc, neutra
    from oracle import flow as oflow
    from oracle import potentials as opot
    from oracle import samplers as osamp
    ck = {'n_hidden': 6, 'n_layers': cl}
    torch.manual_seed(d + cl)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), conditioner_kwargs=ck)), 5, 0.3, 0.8)
    f = Flow(RealNVP((d,), conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())

    def make():
        return neutra.NeuTraHMC((d,), Funnel((d,), 3.0), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=4, step_size=0.02),
                                mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f),
                                neutra.NeuTraParameters(n_iterations=3))
    z = 0.5 * torch.randn(300, d)
    s = make()
    assert s._min_hidden() == 64
    u1, g1 = s._potential_grad(z.to(dev))
    s.seed = 11
    out1 = s.sample(z, show_progress=False)
    monkeypatch.setenv('NFMC_NEUTRA_VALU', '1')
    s2 = make()
    assert s2._min_hidden() == 0
    u2, g2 = s2._potential_grad(z.to(dev))
    s2.seed = 11
    out2 = s2.sample(z, show_progress=False)
    zz = z.clone().requires_grad_(True)
    u0 = osamp.neutra_adjusted_target(of, opot.funnel(3.0), (d,))(zz)
    g0, = torch.autograd.grad(u0.sum(), zz)
    for u, g in ((u1, g1), (u2, g2)):
        np.testing.assert_allclose(u.cpu().numpy(), u0.detach().numpy(), atol=2e-4, rtol=2e-5)
        np.testing.assert_allclose(g.cpu().numpy(), g0.numpy(), atol=2e-4, rtol=2e-4)
    a, b = out1.samples[-1], out2.samples[-1]
    assert float(((a - b).abs().amax(dim=1) < 1e-3).float().mean()) > 0.97
    assert out1.statistics.n_attempted_trajectories == out2.statistics.n_attempted_trajectories == 900


# ------------------------------------------------------------------------------------------ edge cases
def _sumsq(x):
    return torch.sum(x ** 2, dim=-1)


def test_edge_smallest_and_largest_shapes(dev):
    """One chain of one coordinate; the widest event the kernels take (d = 1024) and one beyond (ValueError, the
    reference's error channel); a chain count that is not a multiple of any tile."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    torch.manual_seed(0)
    out = sample(_sumsq, event_shape=(1,), strategy='mala', n_chains=1, n_iterations=50, show_progress=False)
    assert out.samples.shape == (50, 1, 1) and torch.isfinite(out.samples).all()
    out = sample(_sumsq, event_shape=(1,), strategy='hmc', n_chains=1, n_iterations=5, show_progress=False)
    assert out.samples.shape == (5, 1, 1) and out.statistics.n_attempted_trajectories == 5
    out = sample(SumOfSquares((1024,)), strategy='mala', n_chains=64, n_iterations=300, show_progress=False,
                 param_kwargs={'store_samples': False})
    assert abs(float(out.variance.mean()) - 0.5) < 0.02 and out.mean.shape == (1024,)
    with pytest.raises(ValueError, match='supported range'):
        sample(SumOfSquares((1025,)), strategy='mala', n_chains=4, n_iterations=2, show_progress=False)
    # a flow over ONE coordinate has no pass-through coordinate (every coupling's conditioner is its last bias): the kernels
    # start at d = 2, the flow's passes are composed from torch ops and the sampler takes the split path
    out = sample(_sumsq, event_shape=(1,), strategy='imh', n_chains=8, n_iterations=2, show_progress=False)
    assert out.samples.shape == (2, 8, 1) and torch.isfinite(out.samples).all()
    out = sample(SumOfSquares((64,)), strategy='jump_mala', n_chains=100003, n_iterations=2, show_progress=False,
                 param_kwargs={'store_samples': False}, inner_param_kwargs={'n_iterations': 20})
    assert out.statistics.n_attempted_trajectories == 100003 * 40 and out.statistics.n_attempted_jumps == 100003 * 2


def test_edge_zero_iterations_and_long_runs(dev):
    """n_iterations = 0 returns empty containers; more transitions than one launch takes (512) are chunked."""
    from nfmc_amd import sample
    for strategy in ('mala', 'jump_mala', 'imh'):
        out = sample(_sumsq, event_shape=(4,), strategy=strategy, n_chains=4, n_iterations=0, show_progress=False)
        assert out.samples.shape[0] == 0 and out.statistics.n_attempted_trajectories == 0
    torch.manual_seed(1)
    out = sample(_sumsq, event_shape=(8,), strategy='mala', n_chains=1000, n_iterations=1300, show_progress=False,
                 param_kwargs={'store_samples': False})
    assert out.statistics.n_attempted_trajectories == 1300 * 1000
    assert abs(float(out.variance.mean()) - 0.5) < 0.01 and float(out.mean.abs().max()) < 0.01


def test_edge_nonfinite_states_are_rejected_and_counted(dev):
    """A NaN / inf coordinate makes the log ratio non-finite: the proposal is rejected (langevin.py:106 compares
    log u < NaN -> False), the chain keeps its state, the event is counted; healthy chains are unaffected."""
    from nfmc_amd import sample
    torch.manual_seed(2)
    x0 = torch.randn(16, 8)
    x0[3, 2] = float('nan')
    x0[5, 0] = float('inf')
    out = sample(_sumsq, event_shape=(8,), strategy='mala', x0=x0, n_iterations=10, show_progress=False)
    last = out.samples[-1]
    assert torch.isnan(last[3, 2]) and torch.isinf(last[5, 0])
    assert out.statistics.n_nonfinite_log_ratios == 20
    ok = [i for i in range(16) if i not in (3, 5)]
    assert torch.isfinite(last[ok]).all() and out.statistics.n_accepted_trajectories > 0


@pytest.mark.parametrize('strategy', ['mala', 'jump_mala', 'imh', 'neutra_hmc'])
def test_progress_bar_on_and_off_give_the_same_run(dev, strategy):
    """show_progress=True drives a tqdm bar (and, for the jump samplers, reads the counters every outer step);
    show_progress=False uses a no-op stand-in.  Same chains either way."""
    from nfmc_amd import sample
    outs = []
    for show in (False, True):
        torch.manual_seed(3)
        kw = dict(inner_param_kwargs={'n_iterations': 5}) if strategy == 'jump_mala' else {}
        if strategy in ('jump_mala', 'imh', 'neutra_hmc'):
            kw['flow'] = 'realnvp'
        outs.append(sample(_sumsq, event_shape=(8,), strategy=strategy, n_chains=40, n_iterations=6, show_progress=show,
                           seed=5, **kw))
    a, b = outs
    assert torch.equal(a.samples, b.samples)
    assert a.statistics.n_accepted_trajectories == b.statistics.n_accepted_trajectories


def test_edge_c_abi_argument_errors(dev):
    """Negative status -> ValueError with the library's message; nothing is launched."""
    import ctypes as C
    from nfmc_amd import hip
    a = hip.NfmcMalaArgs()
    with pytest.raises(ValueError, match='invalid argument'):
        hip.check(hip.lib().nfmc_mala_steps_f32(C.byref(a), hip.stream()), 'nfmc_mala_steps_f32')
    x = torch.zeros(4, 8, device=dev)
    a.x, a.n, a.d, a.n_steps, a.step_size, a.adjust = hip.ptr(x), 4, 8, 100000, 0.1, 1
    a.pot = hip.NfmcPotential(0, 0, None, None, 1.0, 0.0)
    with pytest.raises(ValueError):
        hip.check(hip.lib().nfmc_mala_steps_f32(C.byref(a), hip.stream()), 'nfmc_mala_steps_f32')
    st = hip.DeviceStats(8, dev)
    raw = st._raw()
    raw.scratch_bytes = 16   # too small
    a.n_steps, a.stats = 4, raw
    with pytest.raises(ValueError, match='scratch'):
        hip.check(hip.lib().nfmc_mala_steps_f32(C.byref(a), hip.stream()), 'nfmc_mala_steps_f32')


@pytest.mark.parametrize('d,nl,nh,cl', [(6, 2, None, 2), (7, 3, 5, 1), (64, 2, 16, 2), (33, 2, 32, 2)])
def test_rqs_flow_matches_oracle_and_known_answers(dev, d, nl, nh, cl):
    """'c-rqnsf' through the kernels (nfmc_realnvp_forward/inverse_f32 with n_bins = 8): forward, inverse, log_prob
    and sampling against the oracle's spline; tolerances: fp32 kernels with hardware exp/log/rcp/sqrt."""
    from nfmc_amd.util import create_flow_object
    from nfmc_amd.flows import CRQNSF
    from oracle import flow as oflow
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    torch.manual_seed(d * 7 + nl)
    # wide inputs with O(1) weights saturate the softmaxes (bins at the minimum width, slopes ~1e3) and make the
    # fp32 round trip meaningless; keep the splines moderately bent at large d
    of = oflow.perturb_(oflow.Flow(oflow.CRQNSF((d,), n_layers=nl, conditioner_kwargs=ck)), 9, 1.0 if d <= 8 else 0.25, 0.8)
    f = create_flow_object('c-rqnsf', (d,), n_layers=nl, conditioner_kwargs=ck)
    assert isinstance(f.bijection, CRQNSF)
    f.load_state_dict(of.state_dict())
    x = torch.randn(400, d) * 1.5
    x[:5] *= 6.0    # beyond the spline bound in some coordinates: identity there
    with torch.no_grad():
        z0, ld0 = of.bijection.forward(x)
        lp0 = of.log_prob(x)
        xi0, ldi0 = of.bijection.inverse(x)
    z, ld = f.bijection.forward(x)
    np.testing.assert_allclose(z.cpu().numpy(), z0.numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(ld.cpu().numpy(), ld0.numpy(), atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), lp0.numpy(), atol=5e-3, rtol=2e-4)
    xi, ldi = f.bijection.inverse(x)
    np.testing.assert_allclose(xi.cpu().numpy(), xi0.numpy(), atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(ldi.cpu().numpy(), ldi0.numpy(), atol=1e-3, rtol=1e-4)
    xb, ldb = f.bijection.inverse(z)
    np.testing.assert_allclose(xb.cpu().numpy(), x.numpy(), atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(ldb.cpu().numpy(), -ld.cpu().numpy(), atol=2e-2, rtol=2e-3)   # fp32 round trip
    xs, lq = f.sample(500, return_log_prob=True)
    err = (lq - f.log_prob(xs)).abs().cpu()        # inverse pass vs forward pass of the same point: an fp32 round trip
    assert float(err.max()) < 5e-2 and float((err < 3e-3).float().mean()) > 0.98


def test_rqs_flow_in_jump_and_imh_strategies(dev):
    """Spline flows drive the flow-proposal Metropolis kernels (one chain per lane): jump_mala and imh run, moments
    of the N(0, I/2) target come out, a fit lifts the jump acceptance; NeuTra runs through the split path."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    from nfmc_amd.util import create_flow_object
    d = 8
    torch.manual_seed(0)
    flow = create_flow_object('c-rqnsf', (d,))
    data = torch.randn(4096, d) * 0.7071
    flow.fit(data[:3000], x_val=data[3000:], n_epochs=150, lr=0.02)
    out = sample(SumOfSquares((d,)), strategy='jump_mala', flow=flow, n_chains=2048, n_iterations=10, show_progress=False,
                 inner_param_kwargs={'n_iterations': 20}, param_kwargs={'store_samples': False}, seed=3)
    st = out.statistics
    assert st.n_attempted_jumps == 2048 * 10 and st.jump_acceptance_rate > 0.3
    assert abs(float(out.variance.mean()) - 0.5) < 0.03 and float(out.mean.abs().max()) < 0.05
    out = sample(SumOfSquares((d,)), strategy='imh', flow=flow, n_chains=2048, n_iterations=50, show_progress=False, seed=4)
    assert out.samples.shape == (50, 2048, d) and out.statistics.acceptance_rate > 0.3
    assert abs(float(out.variance.mean()) - 0.5) < 0.05
    # NeuTra: no reverse-sweep kernel for splines -> the split path differentiates the torch restatement of the flow
    out = sample(SumOfSquares((d,)), strategy='neutra_hmc', flow=flow, n_chains=256, n_iterations=20, show_progress=False,
                 inner_kernel_kwargs={'n_leapfrog_steps': 5, 'step_size': 0.2}, seed=5)
    assert out.samples.shape == (20, 256, d) and torch.isfinite(out.samples).all()
    assert out.statistics.acceptance_rate > 0.5
    x_last, _ = flow.bijection.inverse(out.samples[-1])       # NeuTra samples live in latent space (neutra.py:122)
    assert abs(float(x_last.var(dim=0).mean()) - 0.5) < 0.1


@pytest.mark.parametrize('strategy', ['jump_mala', 'jump_hmc'])
def test_fused_jump_tail_exact_fit_equals_separate_jump_launch(dev, strategy):
    """d = 64 (exact-fit layout): the jump riding at the end of the inner launch (NfmcJumpTail) and the separate
    flow-MH launch simulate the same chains from the same Philox streams."""
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.potentials import SumOfSquares
    outs = []
    for fuse in (False, True):
        torch.manual_seed(5)
        s = create_sampler(SumOfSquares((64,)), strategy=strategy, flow='realnvp', param_kwargs={'n_iterations': 4},
                           inner_param_kwargs={'n_iterations': 6})
        s.fuse_jump_tail = fuse
        s.seed = 21
        torch.manual_seed(6)
        outs.append(s.sample(torch.randn(700, 64) * 0.7, show_progress=False))
    a, b = outs
    assert a.samples.shape == b.samples.shape == (4 * 7, 700, 64)
    close = ((a.samples - b.samples).abs().amax(dim=(0, 2)) < 1e-4).float().mean()
    assert float(close) > 0.99
    assert abs(a.statistics.n_accepted_jumps - b.statistics.n_accepted_jumps) <= 2
    assert a.statistics.n_attempted_jumps == b.statistics.n_attempted_jumps == 4 * 700


@pytest.mark.parametrize('d,n,strategy', [(256, 1001, 'jump_hmc'), (64, 700, 'jump_mala'), (128, 333, 'imh'), (512, 130, 'jump_mala')])
def test_two_chains_per_lane_group_equals_one_chain_kernel(dev, d, n, strategy, monkeypatch):
    """flow_mh_b2_kernel (two chains per lane group share every weight row read from LDS; the production jump kernel
    at d >= 256) against flow_mh_b_kernel: the same chains, bit for bit -- per chain the arithmetic and the Philox
    stream are the same.  Odd chain counts leave the last lane group with one chain."""
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.potentials import SumOfSquares
    outs = []
    for dual in ('0', '1'):
        monkeypatch.setenv('NFMC_FLOWB_DUAL', dual)
        torch.manual_seed(5)
        kw = {'inner_param_kwargs': {'n_iterations': 3}} if strategy != 'imh' else {}
        s = create_sampler(SumOfSquares((d,)), strategy=strategy, flow='realnvp',
                           param_kwargs={'n_iterations': 4, 'store_samples': False}, **kw)
        if hasattr(s, 'fuse_jump_tail'):
            s.fuse_jump_tail = False
        s.seed = 33
        torch.manual_seed(6)
        outs.append(s.sample(torch.randn(n, d) * 0.7, show_progress=False))
    a, b = outs
    assert torch.equal(a.running_samples.last_sample, b.running_samples.last_sample)
    key = 'n_accepted_jumps' if strategy != 'imh' else 'n_accepted_trajectories'
    assert getattr(a.statistics, key) == getattr(b.statistics, key)
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), atol=1e-5)
    np.testing.assert_allclose(a.second_moment.numpy(), b.second_moment.numpy(), atol=1e-5)


@pytest.mark.parametrize('d,nh,cl,strategy', [(64, 64, 1, 'imh'), (128, 128, 2, 'jump_mala'), (64, 40, 2, 'jump_hmc'),
                                              (256, 128, 2, 'jump_mala'), (160, 64, 1, 'imh')])
def test_wide_flow_metropolis_on_matrix_cores_equals_valu_kernels(dev, d, nh, cl, strategy, monkeypatch):
    """Wide conditioners at d = 64 / 128: forward / inverse / flow-MH run on the matrix cores (flow_mfma.hip); at the other
    multiples of 32 -- round 4 -- on the streamed matrix-core flow-MH kernel (mfma_wide.hip: flow_mh_wide_kernel, one launch
    per step; until then the step was composed from three launches).  The one-chain-per-lane kernels (NFMC_FLOW_NO_MFMA=1)
    simulate the same chains from the same Philox streams.  The matrix-core leg must go through nfmc_flow_mh_steps_f32."""
    from nfmc_amd.samplers import imh as imh_mod, jump as jump_mod
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow
    ck = {'n_hidden': nh, 'n_layers': cl}
    torch.manual_seed(d + nh)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=2, conditioner_kwargs=ck)), 4, 0.05, 0.75)
    outs = []
    fused_calls = []
    orig = jump_mod.launch_flow_mh

    def spy(*args, **kwargs):
        fused_calls.append(1)
        return orig(*args, **kwargs)
    monkeypatch.setattr(jump_mod, 'launch_flow_mh', spy)
    monkeypatch.setattr(imh_mod, 'launch_flow_mh', spy)
    for no_mfma in (False, True):
        if no_mfma:
            monkeypatch.setenv('NFMC_FLOW_NO_MFMA', '1')
        f = Flow(RealNVP((d,), n_layers=2, conditioner_kwargs=ck))
        f.load_state_dict(of.state_dict())
        kw = {'inner_param_kwargs': {'n_iterations': 3}} if strategy != 'imh' else {}
        s = create_sampler(SumOfSquares((d,)), strategy=strategy, flow=f, param_kwargs={'n_iterations': 5}, **kw)
        s.seed = 17
        torch.manual_seed(9)
        outs.append(s.sample(torch.randn(300, d) * 0.7, show_progress=False))
        if not no_mfma:
            assert fused_calls, 'the matrix-core leg did not go through the fused flow-MH entry point'
    a, b = outs
    assert a.samples.shape == b.samples.shape
    close = ((a.samples - b.samples).abs().amax(dim=(0, 2)) < 2e-3).float().mean()
    assert float(close) > 0.97, float(close)
    sa, sb = a.statistics, b.statistics
    key = 'n_accepted_jumps' if strategy != 'imh' else 'n_accepted_trajectories'
    assert abs(getattr(sa, key) - getattr(sb, key)) <= 3
    assert getattr(sa, key) > 20          # the flow is close to the target: jumps do get accepted
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), atol=2e-3)


def test_mixed_deferred_and_immediate_statistics(dev):
    """Closed-form potential (fused inner launches: deferred statistics) with a FOREIGN flow object (jump through the
    split path: statistics folded per call): the two modes alternate on one scratch and must not contaminate each
    other -- the run equals the all-fused run of the same chains."""
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    d = 16

    class Foreign(torch.nn.Module):   # duck-typed flow (sampling/base.py:18-26): delegates to a native one
        def __init__(self, inner):
            super().__init__()
            self.inner = inner
            self.bijection = _Bij(inner.bijection)
        event_shape = (d,)
        def sample(self, n, return_log_prob=False, no_grad=False, rng=None):
            return self.inner.sample(n, return_log_prob=return_log_prob, rng=rng)
        def log_prob(self, x):
            return self.inner.log_prob(x)

    class _Bij:
        def __init__(self, b):
            self._b = b
            self.event_shape, self.layers = b.event_shape, b.layers
        def forward(self, x):
            return self._b.forward(x)
        def inverse(self, z):
            return self._b.inverse(z)

    outs = []
    for foreign in (False, True):
        torch.manual_seed(3)
        f = Flow(RealNVP((d,)))
        f.seed = 99
        s = create_sampler(SumOfSquares((d,)), strategy='jump_mala', flow=Foreign(f) if foreign else f,
                           param_kwargs={'n_iterations': 6, 'store_samples': False}, inner_param_kwargs={'n_iterations': 7})
        s.seed = 5
        torch.manual_seed(4)
        outs.append(s.sample(torch.randn(3000, d) * 0.7, show_progress=False))
    a, b = outs
    sa, sb = a.statistics, b.statistics
    assert sa.n_attempted_trajectories == sb.n_attempted_trajectories == 3000 * 42
    assert sa.n_attempted_jumps == sb.n_attempted_jumps == 3000 * 6
    # the inner transitions see the same noise; the jumps draw their latents from different streams, so compare moments
    assert abs(sa.acceptance_rate - sb.acceptance_rate) < 0.02
    np.testing.assert_allclose(a.second_moment.numpy(), b.second_moment.numpy(), atol=0.03)
    assert abs(float(b.variance.mean()) - 0.5) < 0.03 and abs(float(a.variance.mean()) - 0.5) < 0.03


@pytest.mark.parametrize('strategy', ['mala', 'hmc', 'mh', 'imh', 'adaptive_imh', 'jump_mala', 'jump_hmc', 'neutra_hmc'])
def test_sampling_time_limit(dev, strategy):
    """Mirror of the reference's test/test_time_limit.py (skipped there: "may not terminate"): a million iterations
    under a one-second sampling limit return promptly with whatever was done."""
    import time
    from nfmc_amd import sample
    torch.manual_seed(0)
    t0 = time.time()
    out = sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(10,), strategy=strategy, n_chains=64,
                 n_iterations=1_000_000, sampling_time_limit_seconds=1.0, warmup=False, show_progress=False,
                 param_kwargs={'store_samples': False} if strategy != 'adaptive_imh' else {'n_iterations': 200})
    assert time.time() - t0 < 15.0
    assert out.statistics.n_attempted_trajectories > 0 and torch.isfinite(out.mean).all()


@pytest.mark.parametrize('strategy', ['mala', 'jump_mala', 'imh'])
def test_warmup_time_limit(dev, strategy):
    """Second half of the reference's test/test_time_limit.py: warmup under a time limit, then sampling under one."""
    import time
    from nfmc_amd import sample
    torch.manual_seed(0)
    t0 = time.time()
    out = sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(10,), strategy=strategy, n_chains=64,
                 n_iterations=1_000_000, n_warmup_iterations=1_000_000, sampling_time_limit_seconds=1.0,
                 warmup_time_limit_seconds=1.0, warmup=True, show_progress=False, param_kwargs={'store_samples': False})
    assert time.time() - t0 < 25.0
    assert out.statistics.n_attempted_trajectories > 0 and torch.isfinite(out.mean).all()


def test_moment_estimation_full(dev):
    """Mirror of the reference's test/test_moment_estimation.py::test_full over every supported strategy (the
    `negative_log_likelihood=` keyword of the likelihood-based samplers is accepted and unused here)."""
    from nfmc_amd import sample
    from nfmc_amd.util import get_supported_samplers
    for strategy in get_supported_samplers():
        torch.manual_seed(0)
        out = sample(target=lambda x: torch.sum(x ** 2, dim=1), negative_log_likelihood=lambda x: torch.sum(x ** 2, dim=1),
                     event_shape=(10,), strategy=strategy, n_iterations=3, n_warmup_iterations=3, show_progress=False)
        for v in (out.mean, out.second_moment, out.variance):
            assert v.shape == (10,) and torch.isfinite(v).all(), strategy


def test_moment_estimation_basic_direct_construction(dev):
    """Mirror of test_moment_estimation.py::test_basic: `Sampler(event_shape, target)` with every default."""
    from nfmc_amd.potentials import DiagonalGaussian
    from nfmc_amd.samplers.imh import AdaptiveIMH
    from nfmc_amd.samplers.jump import JumpHMC
    from nfmc_amd.samplers.mcmc import HMC
    from nfmc_amd.samplers.neutra import NeuTraHMC
    torch.manual_seed(0)
    target = DiagonalGaussian((12,), mu=torch.zeros(12), sigma=torch.linspace(0.5, 2.0, 12))
    for cls in (HMC, NeuTraHMC, JumpHMC, AdaptiveIMH):
        torch.manual_seed(0)
        s = cls(target.event_shape, target)
        s.params.n_iterations = 3
        if isinstance(s, JumpHMC):
            s.inner_sampler.params.n_iterations = 3
        out = s.sample(torch.randn(100, *target.event_shape), show_progress=False)
        st = out.statistics
        assert st.running_first_moment.shape == tuple(target.event_shape), cls
        assert st.running_second_moment.shape == tuple(target.event_shape)
        assert torch.isfinite(st.running_first_moment).all() and torch.isfinite(st.running_second_moment).all()


def _std_gauss(x):
    return 0.5 * torch.sum(x ** 2, dim=-1)


def test_warmup_outputs_like_reference(dev):
    """Mirror of the reference's test/test_warmup.py: default-constructed samplers, shape of what `.warmup()` returns."""
    from nfmc_amd.samplers.imh import AdaptiveIMH, FixedIMH
    from nfmc_amd.samplers.jump import JumpHMC, JumpMALA, JumpMH, JumpUHMC, JumpULA
    from nfmc_amd.samplers.mcmc import HMC, MALA, MH, UHMC, ULA, RandomWalk
    from nfmc_amd.samplers.neutra import NeuTraHMC
    n_dim, n_chains = 5, 3
    for cls in (MALA, MH, UHMC, HMC, ULA, RandomWalk):                       # test_warmup_mcmc
        torch.manual_seed(0)
        s = cls(event_shape=(n_dim,), target=_std_gauss)
        s.params.n_warmup_iterations = 7
        out = s.warmup(torch.randn(n_chains, n_dim), show_progress=False)
        assert out.samples.shape == (7, n_chains, n_dim) and torch.isfinite(out.samples).all(), cls
    for cls in (JumpMH, JumpULA, JumpHMC, JumpUHMC, JumpMALA):               # test_warmup_jump_nfmc
        torch.manual_seed(0)
        s = cls(event_shape=(n_dim,), target=_std_gauss)
        s.params.flow_fit_kwargs = {**(s.params.flow_fit_kwargs or {}), 'n_epochs': 10}
        s.inner_sampler.params.n_warmup_iterations = 5
        out = s.warmup(torch.randn(n_chains, n_dim), show_progress=False)
        assert out.samples.dim() == 3 and out.samples.shape[1:] == (n_chains, n_dim), cls
        assert torch.isfinite(out.samples).all()
    for cls in (AdaptiveIMH, FixedIMH):                                      # test_warmup_imh
        torch.manual_seed(0)
        s = cls(event_shape=(n_dim,), target=_std_gauss)
        s.params.warmup_fit_kwargs.update(n_epochs=20, n_samples=64)
        out = s.warmup(torch.randn(n_chains, n_dim), show_progress=False)
        assert out.samples.shape == (1, n_chains, n_dim) and torch.isfinite(out.samples).all(), cls
    torch.manual_seed(0)                                                      # test_warmup_neutra
    s = NeuTraHMC(event_shape=(n_dim,), target=_std_gauss)
    s.params.warmup_fit_kwargs.update(n_epochs=20, n_samples=64)
    s.params.n_warmup_iterations = 6
    out = s.warmup(torch.randn(n_chains, n_dim), show_progress=False)
    assert out.samples.shape == (s.inner_sampler.params.n_warmup_iterations, n_chains, n_dim)
    assert torch.isfinite(out.samples).all()


@pytest.mark.parametrize('d,n,T,store', [(64, 300, 40, True), (25, 1000, 700, False), (7, 5, 33, True)])
def test_imh_data_parallel_equals_sequential_transitions(dev, d, n, T, store, monkeypatch):
    """FixedIMH as a data-parallel problem (imh_parallel.hip: all proposals at once, per-chain scan, weighted replay)
    against the sequential flow-MH kernel: same Philox streams -> states, samples and counters bit for bit; the
    moments are sums in another order (tolerance 1e-5)."""
    from nfmc_amd.samplers import imh
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow
    torch.manual_seed(d)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,))), 6, 0.2, 0.75)   # close to the target: accepts happen
    outs = []
    for par in ('1', '0'):
        monkeypatch.setenv('NFMC_IMH_PARALLEL', par)
        f = Flow(RealNVP((d,)))
        f.load_state_dict(of.state_dict())
        s = imh.FixedIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=f),
                         imh.IMHParameters(n_iterations=T, store_samples=store))
        s.seed = 77
        torch.manual_seed(1)
        outs.append(s.sample(torch.randn(n, d) * 0.7, show_progress=False))
    a, b = outs
    assert a.statistics.n_accepted_trajectories == b.statistics.n_accepted_trajectories > 0
    assert a.statistics.n_attempted_trajectories == b.statistics.n_attempted_trajectories == n * T
    assert torch.equal(a.running_samples.last_sample, b.running_samples.last_sample)
    if store:
        assert torch.equal(a.samples, b.samples)
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), atol=1e-5)
    np.testing.assert_allclose(a.second_moment.numpy(), b.second_moment.numpy(), atol=1e-5)


# (d, H, hidden layers, coupling layers): every lane layout of the register flow kernels -- LPC = 1 .. 64 lanes per
# chain, HP = 4 / 8, the distributed hidden stack (LPC >= HP) and the redundant one (LPC < HP), exact-fit and ragged d
REGISTER_FLOW_LAYOUTS = [(4, 3, 1, 2), (8, 8, 2, 2), (12, 8, 2, 3), (16, 4, 2, 2), (24, 3, 1, 2), (32, 8, 2, 2),
                         (64, 8, 2, 3), (100, 8, 2, 2), (128, 4, 2, 2), (128, 8, 3, 2), (256, 8, 2, 2), (256, 4, 1, 3),
                         (500, 8, 2, 2), (512, 8, 2, 2), (512, 4, 2, 2)]


@pytest.mark.parametrize('d,nh,cl,nl', REGISTER_FLOW_LAYOUTS)
@pytest.mark.parametrize('parallel', [False, True])
def test_register_flow_metropolis_matches_oracle_in_every_layout(dev, d, nh, cl, nl, parallel):
    """IMH transitions (flow inverse pass + log q + potential + test) on the register-layout kernels, sequential
    (flow_mh_b_kernel) and data-parallel (imh_parallel.hip), against the oracle on the same Philox streams: the
    log-ratios of every (step, chain) whose chain followed the oracle's accept decisions so far, and the decisions."""
    from nfmc_amd.samplers import imh, jump
    from nfmc_amd.samplers.common import Run
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import samplers as osamp, potentials as opot, flow as oflow
    n, T = 70, 5
    torch.manual_seed(d + nh)
    ck = {'n_hidden': nh, 'n_layers': cl}
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 3, 0.15, 0.75)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    x0 = 0.7 * torch.randn(n, d)
    pot = SumOfSquares((d,))
    s = imh.FixedIMH((d,), pot, imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=T))
    s.seed = 4242
    run = Run(s, x0)
    logq = torch.empty(n, dtype=torch.float32, device=dev)
    masks = torch.zeros(T, n, dtype=torch.uint8, device=dev)
    lr = torch.zeros(T, n, dtype=torch.float32, device=dev)
    samples = torch.zeros(T, n, d, dtype=torch.float32, device=dev)
    supported = jump.imh_parallel_ok(run, f, pot, logq) if parallel else jump.flow_mh_supported(run, f, pot, logq)
    if not supported:
        # no fused kernel for this shape (ragged d = 500: neither the weight image nor two wave tiles fit the LDS):
        # the sampler composes the transition from the flow's own kernels -- same streams, same results
        assert d == 500
        out = s.sample(x0, show_progress=False)
        tr = osamp.imh_sample(x0, opot.sum_squares, of, T, noise=osamp.PhiloxNoise(4242))
        same = (out.samples.reshape(T, n, d) - tr.stacked()).abs().amax(dim=(0, 2)) < 3e-4
        assert same.float().mean() > 0.95
        assert out.statistics.n_attempted_trajectories == n * T
        return
    launch = jump.launch_imh_parallel if parallel else jump.launch_flow_mh
    args = (run, f, pot, logq, T, 0, False) + (() if parallel else (True,))
    keep = launch(*args, run.stats.struct(), samples, masks, lr)
    torch.cuda.synchronize()
    tr = osamp.imh_sample(x0, opot.sum_squares, of, T, noise=osamp.PhiloxNoise(4242))
    want_lr = torch.stack(tr.log_ratios).numpy()
    want_m = torch.stack(tr.masks).numpy()
    got_lr, got_m = lr.cpu().numpy(), masks.cpu().numpy().astype(bool)
    # rows before (and including) a chain's first disagreement with the oracle's decisions are comparable
    agree = np.logical_and.accumulate(np.vstack([np.ones((1, n), bool), (got_m == want_m)[:-1]]), axis=0)
    assert agree.mean() > 0.97
    tol = 2e-4 * max(1.0, d / 64) + 2e-5 * np.abs(want_lr)
    assert (np.abs(got_lr - want_lr)[agree] <= tol[agree]).all(), float(np.abs(got_lr - want_lr)[agree].max())
    assert ((got_m == want_m) | ~agree).mean() > 0.97
    follows = agree[-1] & (got_m[-1] == want_m[-1])
    np.testing.assert_allclose(samples[-1].cpu().numpy()[follows], tr.samples[-1].numpy()[follows], atol=3e-5 * max(1.0, d / 64))
    assert want_m.any() or d >= 100   # small flows are close enough to the target for acceptances to happen


@pytest.mark.parametrize('accepting', [True, False])
@pytest.mark.parametrize('n,k,d', [(1, 1, 16), (1, 57, 16), (63, 27, 16), (64, 28, 16), (65, 29, 16), (129, 55, 16), (200, 56, 16),
                                   (70, 113, 16), (300, 1, 16), (70, 57, 25), (33, 85, 6)])
def test_data_parallel_imh_equals_sequential_kernel_at_block_and_tile_edges(dev, n, k, d, accepting, monkeypatch):
    """The data-parallel independence sampler (imh_parallel.hpp) at the edges of its blocking: step counts around the scan's
    28-step mask words and two-block iterations (1, 27, 28, 29, 55, 56, 57, 113), chain counts around the 64-lane scan
    waves and the 64-row tiles of the proposal / replay kernels (1, 63, 64, 65, 129, ...).  With a flow close to the target
    most proposals are accepted (the replay CORRECTS the proposal pass's sums), with a mismatched one few are (it SUMS the
    accepted rows); exact-fit (d = 16) and ragged (d = 25, 6) register layouts.  Final states, kept states, masks, log-ratios and counters equal the sequential kernel's bit for bit;
    moments to summation order."""
    from nfmc_amd.samplers import imh
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow
    torch.manual_seed(n * 1000 + k)
    of = oflow.Flow(oflow.RealNVP((d,)))
    if accepting:   # identity couplings + the target's scale, slightly perturbed: the proposals are almost draws of N(0, I/2)
        with torch.no_grad():
            for prm in of.parameters():
                prm.zero_()
        of = oflow.perturb_(of, 5, 0.05, 0.5 ** 0.5)
    else:
        of = oflow.perturb_(of, 5, 0.3, 1.5)
    x0 = 0.7 * torch.randn(n, d)
    outs = []
    for par in ('1', '0'):
        monkeypatch.setenv('NFMC_IMH_PARALLEL', par)
        f = Flow(RealNVP((d,)))
        f.load_state_dict(of.state_dict())
        s = imh.FixedIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=k))
        s.seed = 31
        outs.append(s.sample(x0, show_progress=False))
    a, b = outs
    assert torch.equal(a.running_samples.last_sample, b.running_samples.last_sample)
    assert torch.equal(a.samples, b.samples)
    assert a.statistics.n_accepted_trajectories == b.statistics.n_accepted_trajectories
    assert a.statistics.n_attempted_trajectories == b.statistics.n_attempted_trajectories == n * k
    rate = a.statistics.n_accepted_trajectories / (n * k)
    if n * k >= 1000:   # above 0.62 the replay corrects the proposal pass's sums, below it sums the accepted rows
        assert (rate > 0.65) if accepting else (rate < 0.5), rate
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), atol=2e-5)
    np.testing.assert_allclose(a.second_moment.numpy(), b.second_moment.numpy(), atol=2e-5)
    # without the sample store (the path the bench takes: the correcting replay is allowed)
    outs = []
    for par in ('1', '0'):
        monkeypatch.setenv('NFMC_IMH_PARALLEL', par)
        f = Flow(RealNVP((d,)))
        f.load_state_dict(of.state_dict())
        s = imh.FixedIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=k, store_samples=False))
        s.seed = 31
        outs.append(s.sample(x0, show_progress=False))
    a, b = outs
    assert torch.equal(a.running_samples.last_sample, b.running_samples.last_sample)
    assert a.statistics.n_accepted_trajectories == b.statistics.n_accepted_trajectories
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), atol=2e-5)
    np.testing.assert_allclose(a.second_moment.numpy(), b.second_moment.numpy(), atol=2e-5)


@pytest.mark.parametrize('d,nh,cl,nl', [(4, 3, 1, 2), (16, 4, 2, 2), (24, 3, 1, 2), (64, 4, 2, 2), (64, 8, 2, 3), (100, 8, 2, 2),
                                        (128, 4, 2, 2)])
def test_spline_flow_metropolis_on_the_register_layout_matches_oracle(dev, d, nh, cl, nl, monkeypatch):
    """f4 (util.py:288-289, 'c-rqnsf'): the flow-proposal Metropolis step with rational-quadratic spline couplings on the
    register-layout kernel (flow_mh_b_kernel<..., NB = 8>: LPC lanes per chain, each lane evaluates the 23 spline parameters
    of its own target coordinates) against the oracle on the same Philox streams, exact-fit and ragged layouts, and against
    the one-chain-per-lane kernel it replaces for these widths (NFMC_FLOW_TILE_PATH=1): same log-ratios to rounding."""
    from nfmc_amd.samplers import imh, jump
    from nfmc_amd.samplers.common import Run
    from nfmc_amd.flows import Flow, CRQNSF
    from nfmc_amd.potentials import SumOfSquares
    from oracle import samplers as osamp, potentials as opot, flow as oflow
    n, T = 70, 5
    torch.manual_seed(d + nh)
    ck = {'n_hidden': nh, 'n_layers': cl}
    of = oflow.perturb_(oflow.Flow(oflow.CRQNSF((d,), n_layers=nl, conditioner_kwargs=ck)), 3, 0.3, 0.75)
    f = Flow(CRQNSF((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    x0 = 0.7 * torch.randn(n, d)
    pot = SumOfSquares((d,))
    tr = osamp.imh_sample(x0, opot.sum_squares, of, T, noise=osamp.PhiloxNoise(4242))
    want_lr = torch.stack(tr.log_ratios).numpy()
    want_m = torch.stack(tr.masks).numpy()
    got = {}
    for tile_path in ('', '1'):
        if tile_path:
            monkeypatch.setenv('NFMC_FLOW_TILE_PATH', '1')
        s = imh.FixedIMH((d,), pot, imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=T))
        s.seed = 4242
        run = Run(s, x0)
        logq = torch.empty(n, dtype=torch.float32, device=dev)
        masks = torch.zeros(T, n, dtype=torch.uint8, device=dev)
        lr = torch.zeros(T, n, dtype=torch.float32, device=dev)
        samples = torch.zeros(T, n, d, dtype=torch.float32, device=dev)
        assert jump.flow_mh_supported(run, f, pot, logq)
        jump.launch_flow_mh(run, f, pot, logq, T, 0, False, True, run.stats.struct(), samples, masks, lr)
        torch.cuda.synchronize()
        got[tile_path] = (lr.cpu().numpy(), masks.cpu().numpy().astype(bool), samples.cpu().numpy())
    got_lr, got_m, got_x = got['']
    agree = np.logical_and.accumulate(np.vstack([np.ones((1, n), bool), (got_m == want_m)[:-1]]), axis=0)
    assert agree.mean() > 0.95
    tol = 3e-4 * max(1.0, d / 64) + 3e-5 * np.abs(want_lr)
    assert (np.abs(got_lr - want_lr)[agree] <= tol[agree]).all(), float(np.abs(got_lr - want_lr)[agree].max())
    follows = agree[-1] & (got_m[-1] == want_m[-1])
    np.testing.assert_allclose(got_x[-1][follows], tr.samples[-1].numpy()[follows], atol=5e-5 * max(1.0, d / 64))
    # the two kernel families agree wherever they made the same decisions
    t_lr, t_m, _t_x = got['1']
    both = np.logical_and.accumulate(np.vstack([np.ones((1, n), bool), (got_m == t_m)[:-1]]), axis=0)
    assert both.mean() > 0.97
    assert (np.abs(got_lr - t_lr)[both] <= 2 * tol[both]).all()
    # the data-parallel independence sampler with the same spline flow (imh_parallel_rqs.hip): the sequential register
    # kernel's decisions, and its log-ratios and states to rounding (the spline arithmetic is compiled in two contexts;
    # fused multiply-adds are contracted differently, unlike the affine flow, whose two paths agree bit for bit)
    monkeypatch.delenv('NFMC_FLOW_TILE_PATH')
    s = imh.FixedIMH((d,), pot, imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=T))
    s.seed = 4242
    run = Run(s, x0)
    logq = torch.empty(n, dtype=torch.float32, device=dev)
    masks = torch.zeros(T, n, dtype=torch.uint8, device=dev)
    lr = torch.zeros(T, n, dtype=torch.float32, device=dev)
    samples = torch.zeros(T, n, d, dtype=torch.float32, device=dev)
    if not jump.imh_parallel_ok(run, f, pot, logq):
        assert d >= 100 and nh == 8   # the spline weight image of the 128-coordinate layout at width 8 exceeds the LDS
        return
    jump.launch_imh_parallel(run, f, pot, logq, T, 0, False, run.stats.struct(), samples, masks, lr)
    torch.cuda.synchronize()
    p_m = masks.cpu().numpy().astype(bool)
    both = np.logical_and.accumulate(np.vstack([np.ones((1, n), bool), (got_m == p_m)[:-1]]), axis=0)
    assert both.mean() > 0.99 and ((got_m == p_m) | ~both).mean() > 0.99
    assert (np.abs(lr.cpu().numpy() - got_lr)[both] <= 0.2 * tol[both]).all()
    follows = both[-1] & (got_m[-1] == p_m[-1])
    np.testing.assert_allclose(samples.cpu().numpy()[-1][follows], got_x[-1][follows], atol=2e-5 * max(1.0, d / 64))


@pytest.mark.parametrize('d,ck,kind', [(200, {}, 'realnvp'), (24, {'n_hidden': 40}, 'realnvp'), (10, {}, 'c-rqnsf')])
def test_neutra_hmc_shapes_without_a_fused_kernel_match_oracle(dev, d, ck, kind):
    """NeuTra HMC where nfmc_neutra_hmc_steps_f32 has no kernel (d > ~156: four wave tiles exceed the LDS; conditioners
    wider than 32 off the matrix-core shapes; spline couplings): the sampler composes the transition from the inner
    HMC's split path on the adjusted target -- same Philox streams as the fused kernel, so the oracle still applies."""
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.flows import Flow, RealNVP, CRQNSF
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    n, T, L, h = 40, 3, 3, 0.05
    torch.manual_seed(d)
    ocls, cls = (oflow.CRQNSF, CRQNSF) if kind == 'c-rqnsf' else (oflow.RealNVP, RealNVP)
    of = oflow.perturb_(oflow.Flow(ocls((d,), conditioner_kwargs=ck)), 9, 0.2)
    f = Flow(cls((d,), conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    z0 = 0.5 * torch.randn(n, d)
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
    s.seed = 78
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.sum_squares, of, T, h, None, L, noise=osamp.PhiloxNoise(78))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 5e-4
    assert same.float().mean() > 0.9, float(same.float().mean())
    assert out.statistics.n_attempted_trajectories == n * T
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 4


@pytest.mark.parametrize('d,nh,cl,nl,mass', [(256, 128, 2, 2, False), (512, 64, 1, 3, True), (96, 100, 2, 2, True)])
def test_neutra_hmc_wide_events_and_wide_conditioners_stay_off_torch_autograd(dev, d, nh, cl, nl, mass, monkeypatch):
    """NeuTra HMC at d = 256 / 512 (any multiple of 32) with a conditioner wider than 32: `nfmc_neutra_hmc_steps_f32` composes
    the trajectory on the stream from the streamed matrix-core gradient kernel and three elementwise kernels
    (csrc/mfma_wide.hip: no Python between the leapfrog steps, no torch autograd -- the test fails if the split path or the
    torch restatement is touched), one gradient per position, and follows the oracle on the Philox streams: kept states,
    acceptance count, moments; with and without a mass diagonal."""
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd import flow_training
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    n, T, L, h = 150, 3, 10, 0.02
    ck = {'n_hidden': nh, 'n_layers': cl}
    torch.manual_seed(d + nh)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), 9, 0.05)
    f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    z0 = 0.5 * torch.randn(n, d)
    imd = torch.linspace(0.8, 1.3, d) if mass else None

    def boom(*a, **k):
        raise AssertionError('the split path / the torch restatement of the flow was used: no kernel served this shape')
    monkeypatch.setattr(flow_training, 'inverse_torch', boom)
    kern = mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h, **({'inv_mass_diag': imd.clone()} if mass else {}))
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)), kern, mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f),
                         neutra.NeuTraParameters(n_iterations=T))
    s.inner_sampler.sample = boom
    s.seed = 78
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.sum_squares, of, T, h, imd, L, noise=osamp.PhiloxNoise(78))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 5e-4
    assert same.float().mean() > 0.9, float(same.float().mean())
    np.testing.assert_allclose(got[:, same].numpy(), want[:, same].numpy(), atol=5e-4, rtol=0)
    assert out.statistics.n_attempted_trajectories == n * T
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 4
    np.testing.assert_allclose(out.mean.numpy(), tr.moments.first.numpy(), atol=3e-3)
    np.testing.assert_allclose(out.second_moment.numpy(), tr.moments.second.numpy(), atol=5e-3)


@pytest.mark.parametrize('d,nh', [(200, None), (256, None), (300, 8), (512, None), (511, 16)])
def test_neutra_hmc_wide_events_run_on_the_fused_kernel(dev, d, nh):
    """Round 3: the VALU NeuTra kernels hold 32 or 16 chains per wave when 64 rows of the event do not fit the LDS, so every
    event size up to 512 with a conditioner of width <= 32 has a fused trajectory kernel and a gradient kernel (before,
    d > ~156 / ~208 went to the split path and torch autograd).  The fused kernel must run (the test fails on the split
    path), match the oracle on the Philox streams, and `nfmc_neutra_potential_grad_f32` must match autograd of the oracle."""
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    n, T, L, h = 70, 2, 2, 0.03
    torch.manual_seed(d)
    ck = {'n_hidden': nh} if nh else {}
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), conditioner_kwargs=ck)), 9, 0.1)
    f = Flow(RealNVP((d,), conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    z0 = 0.5 * torch.randn(n, d)
    s = neutra.NeuTraHMC((d,), SumOfSquares((d,)), mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))

    def boom(*a, **k):
        raise AssertionError('NeuTra took the split path: no fused kernel ran')
    s.inner_sampler.sample = boom
    s.seed = 78
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, opot.sum_squares, of, T, h, None, L, noise=osamp.PhiloxNoise(78))
    got, want = out.samples.reshape(T, n, d), tr.stacked()
    same = (got - want).abs().amax(dim=(0, 2)) < 5e-4
    assert same.float().mean() > 0.9, float(same.float().mean())
    assert out.statistics.n_attempted_trajectories == n * T
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 4
    # the standalone gradient kernel against autograd through the CPU restatement (neutra.py:58-68)
    u, g = s._potential_grad(z0)
    zz = z0.clone().requires_grad_(True)
    x, ld = of.bijection.inverse(zz)
    ut = opot.sum_squares(x) - ld
    gt, = torch.autograd.grad(ut.sum(), zz)
    np.testing.assert_allclose(u.cpu().numpy(), ut.detach().numpy(), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(g.cpu().numpy(), gt.numpy(), atol=2e-4 * max(1.0, float(gt.abs().max())))


@pytest.mark.parametrize('strategy', ['jump_mala', 'jump_hmc', 'jump_mh', 'imh', 'adaptive_imh', 'neutra_hmc', 'neutra_mh'])
def test_every_flow_strategy_runs_on_awkward_shapes(dev, strategy):
    """Event sizes around every tile / layout boundary (2 ... 511, odd, ragged), each flow kind, a conditioner width off
    the matrix-core shapes: whichever kernel (fused, register, tile, matrix-core, composed) serves the shape, the call
    returns finite samples of the right shape."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    kw = {}
    if strategy.startswith('jump'):
        kw = dict(inner_param_kwargs={'n_iterations': 3})
    elif strategy == 'neutra_hmc':
        kw = dict(inner_kernel_kwargs={'n_leapfrog_steps': 2})
    wide = 'realnvp%{"conditioner_kwargs":{"n_hidden":40}}'
    cases = [(d, 'realnvp') for d in (2, 3, 5, 31, 65, 129, 257, 300, 511, 512)]
    cases += [(d, fl) for d in (5, 65, 300) for fl in ('nice', 'c-rqnsf', wide)]
    cases += [(1, fl) for fl in ('realnvp', 'nice', 'c-rqnsf')]   # one coordinate: every coupling's source half is empty
    for d, fl in cases:
        torch.manual_seed(0)
        out = sample(SumOfSquares((d,)), strategy=strategy, flow=fl, n_chains=37, n_iterations=3, show_progress=False, **kw)
        assert out.samples.shape[1:] == (37, d) and torch.isfinite(out.samples).all(), (strategy, d, fl)
    # beyond the flow kernels (d > 512): the flow passes are composed from torch ops on the GPU, the samplers take
    # the split path
    out = sample(SumOfSquares((513,)), strategy=strategy, flow='realnvp', n_chains=8, n_iterations=2, show_progress=False, **kw)
    assert out.samples.shape[1:] == (8, 513) and torch.isfinite(out.samples).all()


def test_neutra_hmc_with_an_arbitrary_callable_target_matches_oracle(dev):
    """neutra.py:58-68 with a target the library has no closed form for (a two-mode mixture): the adjusted potential is
    differentiated by autograd through the torch restatement of the flow; momenta and accept uniforms are the fused
    kernel's Philox streams."""
    from nfmc_amd.samplers import neutra, mcmc
    from nfmc_amd.flows import Flow, RealNVP
    from oracle import flow as oflow, samplers as osamp

    def target(x):
        c = torch.tensor([-2.0, 2.0], device=x.device)
        return -torch.logsumexp(-0.5 * (x.unsqueeze(-1) - c) ** 2, dim=-1).sum(-1)

    d, n, T, L, h = 6, 48, 3, 4, 0.1
    torch.manual_seed(5)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,))), 4, 0.2)
    f = Flow(RealNVP((d,)))
    f.load_state_dict(of.state_dict())
    z0 = torch.randn(n, d)
    s = neutra.NeuTraHMC((d,), target, mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h),
                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
    s.seed = 79
    out = s.sample(z0, show_progress=False)
    tr = osamp.neutra_hmc_sample(z0, target, of, T, h, None, L, noise=osamp.PhiloxNoise(79))
    same = (out.samples.reshape(T, n, d) - tr.stacked()).abs().amax(dim=(0, 2)) < 5e-4
    assert same.float().mean() > 0.9, float(same.float().mean())
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 3


def test_edge_every_entry_point_rejects_null_and_zeroed_arguments(dev):
    """Each C-ABI entry point called with NULL / zero-initialised arguments returns a negative status (the caller's
    ValueError) -- it neither launches nor dereferences anything."""
    from nfmc_amd import hip
    L, st = hip.lib(), hip.stream()
    z = lambda cls: C.byref(cls())
    calls = {
        'mala': lambda: L.nfmc_mala_steps_f32(z(hip.NfmcMalaArgs), st),
        'hmc': lambda: L.nfmc_hmc_steps_f32(z(hip.NfmcHmcArgs), st),
        'forward': lambda: L.nfmc_realnvp_forward_f32(z(hip.NfmcRealNVP), None, 4, None, None, None, st),
        'inverse': lambda: L.nfmc_realnvp_inverse_f32(z(hip.NfmcRealNVP), None, 4, None, None, None, None, st),
        'flow_mh': lambda: L.nfmc_flow_mh_steps_f32(z(hip.NfmcFlowMhArgs), st),
        'flow_mh_supported': lambda: L.nfmc_flow_mh_supported_f32(z(hip.NfmcFlowMhArgs)),
        'imh_parallel_supported': lambda: L.nfmc_imh_parallel_supported_f32(z(hip.NfmcFlowMhArgs)),
        'imh_parallel': lambda: L.nfmc_imh_parallel_f32(z(hip.NfmcFlowMhArgs), None, 0, st),
        'neutra_hmc': lambda: L.nfmc_neutra_hmc_steps_f32(z(hip.NfmcNeutraHmcArgs), st),
        'neutra_grad': lambda: L.nfmc_neutra_potential_grad_f32(z(hip.NfmcRealNVP), z(hip.NfmcPotential), None, 4, None, None, st),
        'select': lambda: L.nfmc_mh_accept_select_f32(z(hip.NfmcSelectArgs), st),
        'langevin_propose': lambda: L.nfmc_langevin_propose_f32(None, None, None, 0.1, 4, 8, z(hip.NfmcRng), None, st),
        'langevin_log_ratio': lambda: L.nfmc_langevin_log_ratio_f32(None, None, None, None, None, None, None, 0.1, 4, 8, None, st),
        'moments': lambda: L.nfmc_moments_update_f32(None, 4, 8, z(hip.NfmcStats), st),
        'stats_fold': lambda: L.nfmc_stats_fold_f32(z(hip.NfmcStats), 8, 0, None, 0, st),
        'normals': lambda: L.nfmc_philox_normals_f32(z(hip.NfmcRng), hip.TAG_NOISE, 4, 8, None, st),
        'uniforms': lambda: L.nfmc_philox_uniforms_f32(z(hip.NfmcRng), hip.TAG_ACCEPT, 4, None, st),
        'limits': lambda: L.nfmc_limits(None),
        'fit_step': lambda: L.nfmc_flow_fit_step_f32(z(hip.NfmcFlowFit), None, 4, z(hip.NfmcAdamW), st),
        'fit_variational': lambda: L.nfmc_flow_variational_fit_step_f32(z(hip.NfmcFlowFit), z(hip.NfmcPotential), None, 4, z(hip.NfmcAdamW), st),
        'fit_epochs': lambda: L.nfmc_flow_fit_epochs_f32(z(hip.NfmcFlowFit), None, None, 4, 0, z(hip.NfmcAdamW), z(hip.NfmcFitControl), 0, 1, st),
        'fit_epochs_no_control': lambda: L.nfmc_flow_fit_epochs_f32(z(hip.NfmcFlowFit), None, None, 4, 0, z(hip.NfmcAdamW), None, 0, 1, st),
        'rows_sample': lambda: L.nfmc_rows_sample_f32(None, 4, 8, 1, 0, None, 2, None, st),
        'blob_copy': lambda: L.nfmc_flow_blob_copy_f32(None, None, 1, 1, st),
    }
    for name, call in calls.items():
        rc = int(call())
        assert rc < 0, (name, rc)
        with pytest.raises(ValueError):
            hip.check(rc, name)
    # size queries answer 0 for what they have no kernel / no workspace for
    assert int(L.nfmc_flow_scratch_bytes(None, 4, 1)) == 0 and int(L.nfmc_flow_scratch_bytes(z(hip.NfmcRealNVP), 4, 1)) == 0
    assert int(L.nfmc_flow_fit_workspace(None, 4, 0, 16, None)) == 0 and int(L.nfmc_flow_fit_supported_f32(None)) == 0
    for name in ('mala', 'hmc', 'flow_mh', 'neutra_hmc', 'select', 'imh_parallel'):   # NULL struct pointers
        fn = {'mala': L.nfmc_mala_steps_f32, 'hmc': L.nfmc_hmc_steps_f32, 'flow_mh': L.nfmc_flow_mh_steps_f32,
              'neutra_hmc': L.nfmc_neutra_hmc_steps_f32, 'select': L.nfmc_mh_accept_select_f32}.get(name)
        rc = int(fn(None, st)) if fn else int(L.nfmc_imh_parallel_f32(None, None, 0, st))
        assert rc < 0, (name, rc)
    torch.cuda.synchronize()


@pytest.mark.parametrize('strategy', ['hmc', 'mh', 'imh', 'jump_mala', 'jump_hmc', 'neutra_hmc', 'neutra_mh'])
def test_sharded_chains_equal_single_run_for_every_strategy(dev, strategy, monkeypatch):
    """noise keyed by the GLOBAL chain id -- three ranks' blocks of chains are bit for bit the rows of the
    single-process run (sequential flow-MH here; the data-parallel IMH has its own bitwise test against it)."""
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.potentials import SumOfSquares
    from nfmc_amd.dist import Shard
    monkeypatch.setenv('NFMC_IMH_PARALLEL', '0' if strategy == 'imh' else '1')
    d, n = 16, 301
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(3))

    def make():
        torch.manual_seed(11)   # same flow weights in every "process"
        kw = {}
        if strategy.startswith('jump'):
            kw = dict(inner_param_kwargs={'n_iterations': 4})
        elif strategy == 'neutra_hmc':
            kw = dict(inner_kernel_kwargs={'n_leapfrog_steps': 3})
        s = create_sampler(SumOfSquares((d,)), strategy=strategy, flow='realnvp', param_kwargs={'n_iterations': 5}, **kw)
        s.seed = 9
        return s

    full = make().sample(x0, show_progress=False).running_samples.last_sample
    parts = []
    for r in range(3):
        s = make()
        s.shard = Shard(rank=r, world=3)
        s.shard.merge_statistics = lambda st: st
        parts.append(s.sample(x0, show_progress=False).running_samples.last_sample)
    assert torch.equal(torch.cat(parts), full)


def test_edge_state_with_more_than_2_to_31_elements(dev):
    """34 M chains x 64 coordinates (2.2e9 floats, 8.7 GB): row offsets are 64-bit everywhere -- the last chains move
    like the first ones, and the moments cover all of them."""
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    n, d = 34_000_000, 64
    x0 = torch.randn(n, d, device=dev) * 0.7071
    for strategy, kw in (('mala', {}), ('jump_mala', {'inner_param_kwargs': {'n_iterations': 2}}), ('imh', {})):
        out = sample(SumOfSquares((d,)), strategy=strategy, x0=x0, n_iterations=2, show_progress=False,
                     param_kwargs={'store_samples': False}, seed=1, **kw)
        last = out.running_samples.last_sample
        assert last.shape == (n, d)
        moved_head = (last[:1000].to(dev) != x0[:1000]).any(dim=1).float().mean()
        moved_tail = (last[-1000:].to(dev) != x0[-1000:]).any(dim=1).float().mean()
        if strategy != 'imh':   # the unfitted flow's independence proposals are almost never accepted
            assert moved_head > 0.2 and moved_tail > 0.2, (strategy, float(moved_head), float(moved_tail))
        assert abs(float(out.variance.mean()) - 0.5) < 5e-3
        del out, last
    x2 = torch.randn(50, 4, 4, device=dev)
    out = sample(lambda x: (x ** 2).flatten(1).sum(1), strategy='hmc', x0=x2, n_iterations=2, show_progress=False)
    assert out.samples.shape == (2, 50, 4, 4)   # event shape taken from x0


@pytest.mark.parametrize('d,ck', [(600, {}), (20, {'n_hidden': 200})])
def test_flows_beyond_the_kernel_shapes_match_oracle(dev, d, ck):
    """Events wider than 512 / conditioners wider than 128: forward, inverse, log_prob and sampling are composed from
    torch ops on the GPU (same spec), latents from the same Philox stream; IMH through the split path follows the
    oracle on native streams."""
    from nfmc_amd import hip
    from nfmc_amd.samplers import imh
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares
    from oracle import flow as oflow, potentials as opot, samplers as osamp, philox
    torch.manual_seed(d)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), conditioner_kwargs=ck)), 3, 0.1, 0.75)
    f = Flow(RealNVP((d,), conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    assert f.bijection.beyond_kernels()
    x = 0.7 * torch.randn(50, d)
    z, ld = f.bijection.forward(x)
    zo, ldo = of.bijection.forward(x)
    np.testing.assert_allclose(z.cpu().numpy(), zo.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(ld.cpu().numpy(), ldo.detach().numpy(), atol=2e-4 * max(1, d / 64))
    xb, _ = f.bijection.inverse(z)
    np.testing.assert_allclose(xb.cpu().numpy(), x.numpy(), atol=2e-5)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), of.log_prob(x).detach().numpy(), atol=3e-4 * max(1, d / 64), rtol=1e-5)
    xs, lq = f.sample(40, return_log_prob=True, rng=hip.make_rng(5, 0, 7))
    zz = philox.normal_field(5, np.arange(40), 7, d, philox.TAG_LATENT)
    xo, ldi = of.bijection.inverse(torch.from_numpy(zz))
    np.testing.assert_allclose(xs.cpu().numpy(), xo.detach().numpy(), atol=3e-5)
    n, T = 60, 4
    x0 = 0.7 * torch.randn(n, d)
    s = imh.FixedIMH((d,), SumOfSquares((d,)), imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=T))
    s.seed = 21
    out = s.sample(x0, show_progress=False)
    tr = osamp.imh_sample(x0, opot.sum_squares, of, T, noise=osamp.PhiloxNoise(21))
    same = (out.samples.reshape(T, n, d) - tr.stacked()).abs().amax(dim=(0, 2)) < 3e-4
    assert same.float().mean() > 0.95
    assert abs(out.statistics.n_accepted_trajectories - tr.n_accepted) <= 3


def test_flow_objects_copy_and_pickle_after_use(dev):
    """A flow that has been used (weights packed for the kernels) still deep-copies and pickles: the device-side
    caches hold ctypes structs with raw pointers and are dropped from the copied state."""
    import copy, io
    from nfmc_amd import sample
    from nfmc_amd.potentials import SumOfSquares
    out = sample(SumOfSquares((8,)), strategy='imh', n_chains=16, n_iterations=2, show_progress=False)
    flow = out.kernel.flow
    twin = copy.deepcopy(flow)
    x = torch.randn(5, 8)
    assert torch.equal(flow.log_prob(x), twin.log_prob(x))
    buf = io.BytesIO()
    torch.save(flow, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert torch.equal(flow.log_prob(x), back.log_prob(x))


def test_randomized_flow_kernel_parity(dev):
    """Forty seeded random flows (kind, d in 2..512, conditioner width 1..128, hidden layers, coupling layers, chain
    counts off and on the tile sizes): forward, inverse, log-determinants and log_prob of whichever kernel serves the
    shape (register / one-chain-per-lane / matrix-core) against the oracle."""
    import random
    from nfmc_amd.flows import Flow, RealNVP, NICE, CRQNSF
    from oracle import flow as oflow
    rng = random.Random(7)
    for trial in range(40):
        kind = rng.choice(['realnvp', 'realnvp', 'nice', 'c-rqnsf'])
        d = rng.choice([2, 3, 5, 8, 17, 32, 63, 64, 65, 100, 128, 129, 200, 256, 384, 511, 512])
        widths = [1, 3, 4, 5, 8, 9, 16, 17, 32]
        H = rng.choice(widths if kind == 'c-rqnsf' else widths + [33, 64, 100, 128])
        cl, nl, n = rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 4]), rng.choice([1, 7, 64, 65, 300])
        ck = {'n_hidden': H, 'n_layers': cl}
        ocls, cls = {'realnvp': (oflow.RealNVP, RealNVP), 'nice': (oflow.NICE, NICE), 'c-rqnsf': (oflow.CRQNSF, CRQNSF)}[kind]
        torch.manual_seed(trial)
        of = oflow.perturb_(oflow.Flow(ocls((d,), n_layers=nl, conditioner_kwargs=ck)), trial, 0.1 if kind == 'c-rqnsf' else 0.2)
        f = Flow(cls((d,), n_layers=nl, conditioner_kwargs=ck))
        f.load_state_dict(of.state_dict())
        x = torch.randn(n, d)
        z, ld = f.bijection.forward(x)
        zo, ldo = of.bijection.forward(x)
        xb, ldb = f.bijection.inverse(zo.detach())
        xo, ldbo = of.bijection.inverse(zo.detach())
        lp, lpo = f.log_prob(x).cpu(), of.log_prob(x).detach()
        tol = 5e-5 * max(1, d / 64) * (4 if kind == 'c-rqnsf' else 1)
        case = (kind, d, H, cl, nl, n)
        assert float((z.cpu() - zo.detach()).abs().max()) < tol, case
        assert float((xb.cpu() - xo.detach()).abs().max()) < 4 * tol, case
        assert float((ld.cpu() - ldo.detach()).abs().max()) < 40 * tol and float((ldb.cpu() - ldbo.detach()).abs().max()) < 40 * tol, case
        assert float((lp - lpo).abs().max() / (1 + lpo.abs().max())) < 1e-4, case


def test_randomized_sampler_parity(dev):
    """Forty seeded random sampler runs (MALA / ULA / HMC / UHMC / MH; d in 1..1024 over every lane layout; unit or
    random mass diagonal; sum of squares / diagonal Gaussian / funnel; chain counts off the tile sizes) on the native
    Philox streams against the oracle."""
    import random
    from nfmc_amd.samplers import mcmc
    from nfmc_amd.potentials import SumOfSquares, DiagonalGaussian, Funnel
    from oracle import samplers as osamp, potentials as opot
    rng = random.Random(11)
    for trial in range(40):
        kind = rng.choice(['langevin', 'langevin', 'hmc', 'mh'])
        d = rng.choice([1, 2, 3, 4, 7, 8, 16, 25, 31, 32, 64, 65, 100, 128, 200, 256, 300, 512, 700, 1024])
        n, k = rng.choice([1, 5, 64, 130, 257]), rng.choice([1, 3, 7])
        adjust, mass = rng.random() > 0.2, rng.random() > 0.5
        potk = rng.choice(['sumsq', 'gauss', 'funnel']) if d >= 2 else 'sumsq'
        torch.manual_seed(trial)
        imd = torch.rand(d) + 0.5 if mass else None
        if potk == 'sumsq':
            pot, opt = SumOfSquares((d,)), opot.sum_squares
        elif potk == 'gauss':
            mu, sig = torch.randn(d) * 0.3, torch.rand(d) + 0.7
            pot, opt = DiagonalGaussian((d,), mu, sig), (lambda v, mu=mu, sig=sig: (((v - mu) ** 2) / (2 * sig ** 2)).sum(-1))
        else:
            pot, opt = Funnel((d,), 3.0), opot.funnel(3.0)
        h = 0.05 if kind == 'hmc' else 0.5 * d ** (-1 / 3)
        x0 = 0.5 * torch.randn(n, d)
        if kind == 'langevin':
            s = (mcmc.MALA if adjust else mcmc.ULA)((d,), pot, mcmc.LangevinKernel(event_size=d, step_size=h, inv_mass_diag=imd),
                                                    mcmc.LangevinParameters(n_iterations=k))
        elif kind == 'hmc':
            s = (mcmc.HMC if adjust else mcmc.UHMC)((d,), pot, mcmc.HMCKernel(event_size=d, step_size=h, n_leapfrog_steps=3,
                                                                              inv_mass_diag=imd), mcmc.HMCParameters(n_iterations=k))
        else:
            adjust = True
            s = mcmc.MH((d,), pot, mcmc.MHKernel(event_size=d, inv_mass_diag=(imd * 0.1 if mass else torch.full((d,), 0.1))),
                        mcmc.MHParameters(n_iterations=k))
        s.seed = 1000 + trial
        out = s.sample(x0, show_progress=False)
        tr = osamp.mcmc_sample(x0, opt, kind, k, None if kind == 'mh' else h, inv_mass_diag=s.kernel.inv_mass_diag.clone(),
                               adjustment=adjust, noise=osamp.PhiloxNoise(1000 + trial), **(dict(n_leapfrog=3) if kind == 'hmc' else {}))
        err = (out.samples.reshape(k, n, d) - tr.stacked()).abs().amax(dim=(0, 2))
        same = err < 3e-4 * max(1, d / 64)   # chains whose accept decision sits on the boundary may differ
        assert same.float().mean() >= (0.9 if n > 20 else 0.6), (kind, d, n, k, adjust, mass, potk, float(err.max()))


def test_randomized_neutra_and_jump_parity(dev):
    """Twenty-four seeded random NeuTra-HMC / JumpMALA / JumpHMC runs (d = 2..200, conditioner widths 3..128, one or two
    hidden layers, one to three couplings, Gaussian and funnel targets) on the native Philox streams against the oracle:
    whichever kernel serves the shape -- fused VALU, matrix-core, register flow, tile flow, or the composed path."""
    import random
    from nfmc_amd.samplers import neutra, mcmc, jump
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import SumOfSquares, Funnel
    from oracle import flow as oflow, potentials as opot, samplers as osamp
    # NFMC_TEST_SWEEP="<seed>,<trials>" runs a longer sweep from another seed (used after kernel changes)
    sweep_seed, trials = (int(v) for v in os.environ.get('NFMC_TEST_SWEEP', '5,24').split(','))
    rng = random.Random(sweep_seed)
    for trial in range(trials):
        which = rng.choice(['neutra', 'neutra', 'jump_mala', 'jump_hmc'])
        d = rng.choice([2, 5, 12, 33, 64, 64, 100, 128, 128, 150, 200])
        H, cl, nl = rng.choice([3, 8, 16, 32, 40, 64, 128]), rng.choice([1, 2]), rng.choice([1, 2, 3])
        n, T, K = rng.choice([3, 40, 70]), 2, 3
        potk = rng.choice(['sumsq', 'funnel'])
        pot, opt = (SumOfSquares((d,)), opot.sum_squares) if potk == 'sumsq' else (Funnel((d,), 3.0), opot.funnel(3.0))
        ck = {'n_hidden': H, 'n_layers': cl}
        torch.manual_seed(trial)
        of = oflow.perturb_(oflow.Flow(oflow.RealNVP((d,), n_layers=nl, conditioner_kwargs=ck)), trial, 0.15,
                            None if which == 'neutra' else 0.75)
        f = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs=ck))
        f.load_state_dict(of.state_dict())
        x0 = 0.5 * torch.randn(n, d)
        seed = 50 + trial
        if which == 'neutra':
            s = neutra.NeuTraHMC((d,), pot, mcmc.HMCKernel(event_size=d, n_leapfrog_steps=3, step_size=0.03), mcmc.HMCParameters(),
                                 neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
            s.seed = seed
            got = s.sample(x0, show_progress=False).samples.reshape(T, n, d)
            want = osamp.neutra_hmc_sample(x0, opt, of, T, 0.03, None, 3, noise=osamp.PhiloxNoise(seed)).stacked()
        elif which == 'jump_mala':
            s = jump.JumpMALA((d,), pot, NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T), None,
                              mcmc.LangevinParameters(n_iterations=K))
            s.seed = seed
            got = s.sample(x0, show_progress=False).samples.reshape(T * (K + 1), n, d)
            want = osamp.jump_sample(x0, opt, of, 'langevin', T, K, d ** (-1 / 3), noise=osamp.PhiloxNoise(seed)).stacked()
        else:
            s = jump.JumpHMC((d,), pot, NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T),
                             mcmc.HMCKernel(event_size=d, n_leapfrog_steps=3, step_size=0.05), mcmc.HMCParameters(n_iterations=K))
            s.seed = seed
            got = s.sample(x0, show_progress=False).samples.reshape(T * (K + 1), n, d)
            want = osamp.jump_sample(x0, opt, of, 'hmc', T, K, 0.05, n_leapfrog=3, noise=osamp.PhiloxNoise(seed)).stacked()
        err = (got - want).abs().amax(dim=(0, 2))
        same = err < 5e-4 * max(1, d / 64)
        assert same.float().mean() >= (0.85 if n > 20 else 0.6), (which, d, H, cl, nl, n, potk, float(err.max()))
